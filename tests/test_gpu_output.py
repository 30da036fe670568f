"""-m gpu: output unit + classifier (model.py:512-576) and the end-to-end parity bar of the task:
classifier logits within 1e-4 (fp32) of the oracle and identical answer argmax."""
import pytest
import torch

from oracle import dropout_hash as dh
from oracle import mac_oracle as mo
from helpers import make_case, oracle_run, rel_err, max_abs
from test_gpu_cell import build_cell

pytestmark = pytest.mark.gpu


def oracle_logits(cfg, ref_params, memory, vq, keep, seed, b0=0, dtype=torch.float64, need_grad=False):
    prm = {k: v.detach().cpu().to(dtype).clone().requires_grad_(need_grad) for k, v in ref_params.items()}
    vs = mo.VarStore(params=prm, dtype=dtype)
    B, d = memory.shape
    masks = None
    if keep < 1.0:
        H = cfg.outClassifierDims[0]
        masks = [torch.from_numpy(dh.mask_for(seed, 7, 0, keep, (B, 2 * d), b0=b0)).to(dtype),
                 torch.from_numpy(dh.mask_for(seed, 8, 0, keep, (B, H), b0=b0)).to(dtype)]
    return mo.output_classifier(cfg, vs, memory, vq, output_keep=keep, masks=masks), prm


@pytest.mark.parametrize("B,d,H,A,train", [(64, 512, 512, 28, False), (64, 512, 512, 28, True), (5, 128, 64, 7, True), (3, 64, 32, 33, False)])
def test_classifier_matches_oracle(macx, dev, B, d, H, A, train):
    cfg = mo.flag_file_config("args", memDim=d, ctrlDim=d, attDim=d, outClassifierDims=[H], answerWordsNum=A)
    out = macx.OutputClassifier(cfg, generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        for f in ("outQuestion_b", "fc0_b", "fc1_b"):
            getattr(out, f).copy_(torch.rand_like(getattr(out, f)) - 0.5)
    g = torch.Generator().manual_seed(2)
    mem = torch.randn(B, d, generator=g)
    vq = torch.rand(B, d, generator=g) * 2 - 1
    memd, vqd = mem.to(dev).requires_grad_(True), vq.to(dev).requires_grad_(True)
    logits = out(memd, vqd, train=train, seed=9, b0=4)
    dl = torch.randn(B, A, generator=g)
    (logits * dl.to(dev)).sum().backward()
    torch.cuda.synchronize()
    keep = cfg.outputDropout if train else 1.0
    mr, vr = mem.double().requires_grad_(True), vq.double().requires_grad_(True)
    ref, prm = oracle_logits(cfg, out.to_reference_dict(), mr, vr, keep, 9, b0=4, need_grad=True)
    (ref * dl.double()).sum().backward()
    assert max_abs(logits, ref) < 2e-5
    assert rel_err(memd.grad, mr.grad) < 1e-4 and rel_err(vqd.grad, vr.grad) < 1e-4
    for f, name in macx.output.REF_NAMES.items():
        assert rel_err(getattr(out, f).grad, prm[name].grad) < 1e-4, f


def test_end_to_end_logits_and_argmax_p12(macx, dev):
    """The task's parity bar: logits within 1e-4 of the fp32 oracle after p = 12 cell steps, argmax identical
    (B=64, S=50, N=196, d=512, evaluation mode; oracle in fp32 AND fp64)."""
    B, S, N, d, p, A = 64, 50, 196, 512, 12, 28
    cfg, vq, words, lengths, kb = make_case("args", B, S, N, d, p)
    cfg.answerWordsNum = A
    cell, params, (vqd, wd, kbd) = build_cell(macx, dev, cfg, vq, words, lengths, kb, False)
    out = macx.OutputClassifier(cfg, generator=torch.Generator().manual_seed(3)).to(dev)
    with torch.no_grad():
        state = cell.run()
        logits = out(state.memory, vqd)
    torch.cuda.synchronize()
    for dtype, tol in ((torch.float32, 1e-4), (torch.float64, 1e-4)):
        ref = oracle_run(cfg, params.to_reference_dict(), vq, words, lengths, kb, dtype=dtype)
        rl, _ = oracle_logits(cfg, out.to_reference_dict(), ref["memory"].detach(), vq.to(dtype), 1.0, 0, dtype=dtype)
        assert max_abs(logits, rl) < tol, (dtype, max_abs(logits, rl))
        assert torch.equal(logits.argmax(-1).cpu(), rl.argmax(-1))


def test_end_to_end_logits_and_argmax_p12_train_mode(macx, dev):
    """The same bar in TRAINING mode at the metric's size: dropout .85/.85/1.0 in the cell (memory, read sites)
    and the classifier's output dropout, masks from the stateless stream handed identically to the fp64 oracle (chunks of 8
    questions: same global question indices, same masks).  Logits within 1e-4, argmax identical."""
    B, S, N, d, p, A = 64, 50, 196, 512, 12, 28
    cfg, vq, words, lengths, kb = make_case("args", B, S, N, d, p)
    cfg.answerWordsNum = A
    seed = 77
    cell, params, (vqd, wd, kbd) = build_cell(macx, dev, cfg, vq, words, lengths, kb, True, seed=seed)
    out = macx.OutputClassifier(cfg, generator=torch.Generator().manual_seed(3)).to(dev)
    with torch.no_grad():
        state = cell.run()
        logits = out(state.memory, vqd, train=True, seed=seed)
    torch.cuda.synchronize()
    mems = []
    for lo in range(0, B, 8):
        sl = slice(lo, lo + 8)
        r = oracle_run(cfg, params.to_reference_dict(), vq[sl], words[sl], lengths[sl], kb[sl], train=True, seed=seed, b0=lo,
                       dtype=torch.float64)
        mems.append(r["memory"].detach())
    mem = torch.cat(mems)
    assert rel_err(state.memory, mem) < 2e-5
    rl, _ = oracle_logits(cfg, out.to_reference_dict(), mem, vq.double(), cfg.outputDropout, seed, dtype=torch.float64)
    assert max_abs(logits, rl) < 1e-4, max_abs(logits, rl)
    assert torch.equal(logits.argmax(-1).cpu(), rl.argmax(-1))


@pytest.mark.parametrize("clip", [8.0, 0.0])
def test_adam_ema_step_matches_oracle(macx, dev, clip):
    """model.py:639-669 over a flat buffer: 4 steps, clip active (large grads) and inactive."""
    from oracle import optim_oracle as oo
    g = torch.Generator().manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(sh, generator=g).to(dev)) for sh in [(300, 70), (513,), (64, 64), (1,)]]
    ref_p = torch.cat([p.detach().cpu().double().reshape(-1) for p in ps]).numpy()
    opt = macx.optim.FlatAdamEMA(ps, lr=1e-3, clip_norm=clip, ema_decay=0.999)
    ref_m = ref_p * 0; ref_v = ref_p * 0; ref_e = ref_p.copy()
    for t in range(1, 5):
        scale = 20.0 if t % 2 else 0.01          # alternate above / below the clip threshold
        grads = [torch.randn(p.shape, generator=g) * scale for p in ps]
        for p, gr in zip(ps, grads):
            p.grad = gr.to(dev)
        norm = opt.step()
        gflat = torch.cat([gr.double().reshape(-1) for gr in grads]).numpy()
        ref_p, ref_m, ref_v, ref_e, ref_norm = oo.adam_ema_step(ref_p, gflat, ref_m, ref_v, ref_e, 1e-3, t, clip=clip)
        torch.cuda.synchronize()
        assert abs(float(norm) - ref_norm) < 1e-4 * ref_norm
        got = torch.cat([p.detach().cpu().double().reshape(-1) for p in ps]).numpy()
        assert abs(got - ref_p).max() < 2e-6
        ema = torch.cat([e.detach().cpu().double().reshape(-1) for e in opt.ema_state()]).numpy()
        assert abs(ema - ref_e).max() < 2e-6
    # parameters are views of the flat buffer
    assert ps[0].data_ptr() == opt.flat.data_ptr()


def test_adam_ema_step_follows_the_reference_training_op(macx, dev):
    """macx_adam_ema_step (optim.FlatAdamEMA), fed the gradients the reference's computeGradients produced, against what the
    reference's addTrainingOp (model.py:639-669) left in the variables, the EMA shadows and the norm -- five steps, clip active
    on three (tests/golden/reference/training_steps.npz, generated by executing model.py)."""
    from helpers import load_training_fixture
    hyper, names, init, steps = load_training_fixture()
    ps = [torch.nn.Parameter(torch.as_tensor(init[n], dtype=torch.float32).to(dev)) for n in names]
    opt = macx.optim.FlatAdamEMA(ps, lr=hyper["lr"], beta1=hyper["beta1"], beta2=hyper["beta2"], eps=hyper["eps"],
                                 clip_norm=hyper["clip"], ema_decay=hyper["decay"])
    for t, s in enumerate(steps):
        for p, n in zip(ps, names):
            p.grad = torch.as_tensor(s["g"][n], dtype=torch.float32).to(dev).reshape(p.shape)
        norm = opt.step()
        torch.cuda.synchronize()
        assert abs(float(norm) - s["norm"]) < 1e-5 * s["norm"]
        for p, e, n in zip(ps, opt.ema_state(), names):
            assert float((p.detach().cpu().double() - torch.as_tensor(s["p"][n]).reshape(p.shape)).abs().max()) < 2e-6, (t, n)
            assert float((e.detach().cpu().double() - torch.as_tensor(s["e"][n]).reshape(e.shape)).abs().max()) < 2e-6, (t, n)
        off = 0


@pytest.mark.parametrize("B,A", [(64, 28), (5, 3), (130, 100), (2, 1)])
def test_answer_loss_and_pred_kernel(macx, dev, B, A):
    """macx_answer_loss (addAnswerLossOp + addPredOp, model.py:593-612): mean sparse CE, first-maximum argmax and the
    gradient of the mean loss in one kernel vs fp64 torch; ties resolve to the first index like tf.argmax."""
    g = torch.Generator().manual_seed(B * 131 + A)
    logits = torch.randn(B, A, generator=g) * 4
    if A > 2:
        logits[0, 1] = logits[0, 2] = logits[0].max() + 1.0           # a tie: argmax must be 1
    answers = torch.randint(0, A, (B,), generator=g)
    ld = logits.to(dev).requires_grad_(True)
    loss, pred = macx.output.answer_loss_and_pred(ld, answers.to(dev))
    (loss * 3.0).backward()
    torch.cuda.synchronize()
    ref = logits.double().requires_grad_(True)
    rloss = torch.nn.functional.cross_entropy(ref, answers)
    (rloss * 3.0).backward()
    assert abs(float(loss) - float(rloss)) < 1e-5
    assert torch.equal(pred.cpu().long(), logits.argmax(-1)) and (A <= 2 or int(pred[0]) == 1)
    assert float((ld.grad.cpu().double() - ref.grad).abs().max()) < 1e-6
    bad = answers.clone()
    bad[0] = A
    l2, _ = macx.output.answer_loss_and_pred(logits.to(dev), bad.to(dev))
    assert not torch.isfinite(l2)                                      # out-of-range label: NaN loss, as TF's GPU kernel


@pytest.mark.parametrize("variant", ["no_question", "question_mul", "deep", "no_hidden", "prelu", "std_relu_mul_deep"])
@pytest.mark.parametrize("train", [False, True])
def test_generic_output_unit_matches_oracle(macx, dev, variant, train):
    """macx.OutputClassifier on the option sets the fused kernels refuse (no --outQuestion, --outQuestionMul, 0 / 2 / 3 hidden
    layers, --relu PRM / STD): one HIP kernel per reference op, against the fp64 oracle with identical dropout masks --
    logits <= 2e-5, every parameter / input gradient <= 1e-4."""
    from test_generic_output_host import out_cfg, run_pair, check
    out, logits, ref, mem, vq, prm = run_pair(macx, out_cfg(variant), train, dev=dev)
    torch.cuda.synchronize()
    check(out, logits, ref, mem, vq, prm, gtol=1e-4)
