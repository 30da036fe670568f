"""-m gpu: whole-cell parity (forward, attentions, histories, gradients) of the HIP path against
the oracle on identical parameters, inputs and dropout masks."""
import pytest
import torch

from oracle import mac_oracle as mo
from helpers import make_case, oracle_run, rel_err, max_abs

pytestmark = pytest.mark.gpu

FWD_TOL = 2e-5     # relative, per state tensor (fp32 kernels vs fp64 oracle)
GRAD_TOL = 2e-4    # relative to the largest entry of each gradient tensor


def build_cell(macx, dev, cfg, vq, words, lengths, kb, train, seed=0, b0=0, requires_grad=False, gen_seed=5, tune=None):
    p = cfg.netLength
    params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(gen_seed)).to(dev)
    # non-zero biases so that bias paths are exercised
    g = torch.Generator().manual_seed(gen_seed + 1)
    with torch.no_grad():
        for f in params.fields:
            t = getattr(params, f)
            if f.endswith("_b"):
                t.copy_((torch.rand(t.shape, generator=g) - 0.5) * 0.2)
    vqd, wd, kbd = [t.to(dev).requires_grad_(requires_grad) for t in (vq, words, kb)]
    cell = macx.MACCell(vecQuestions=vqd, questionWords=wd, questionCntxWords=wd, questionLengths=lengths.to(dev),
                        knowledgeBase=kbd, memoryDropout=cfg.memoryDropout, readDropout=cfg.readDropout,
                        writeDropout=cfg.writeDropout, batchSize=vq.shape[0], train=train, config=cfg, params=params,
                        seed=seed, b0=b0, tune=tune)
    return cell, params, (vqd, wd, kbd)


@pytest.mark.parametrize("name,B,S,N,d,p,train", [
    ("args", 4, 9, 196, 128, 3, False),
    ("args", 4, 9, 196, 128, 3, True),
    ("args", 3, 50, 49, 256, 2, True),
    ("args", 2, 7, 14, 128, 2, False),
    ("args2", 5, 12, 100, 128, 2, True),
    ("args3", 4, 9, 49, 128, 4, True),
    ("args4", 4, 9, 49, 128, 3, True),
    ("args4", 3, 9, 196, 128, 2, False),
    ("args1", 4, 9, 49, 128, 4, True),
    ("args", 3, 9, 196, 512, 2, True),        # d = 512, 588 rows: the chain kernels' 16-row tiles
    ("args", 24, 9, 196, 512, 2, True),       # 4704 rows: 32-row tiles
])
def test_forward_stepwise_matches_oracle(macx, dev, name, B, S, N, d, p, train):
    cfg, vq, words, lengths, kb = make_case(name, B, S, N, d, p)
    cell, params, _ = build_cell(macx, dev, cfg, vq, words, lengths, kb, train, seed=77, b0=2)
    with torch.no_grad():
        state = cell.zero_state(B)
        for i in range(p):
            cell.iteration = i
            _, state = cell(cell.none, state)
    torch.cuda.synchronize()
    ref = oracle_run(cfg, params.to_reference_dict(), vq, words, lengths, kb, train=train, seed=77, b0=2)
    rc = ref["cell"]
    assert rel_err(state.memory, ref["memory"]) < FWD_TOL
    assert rel_err(state.control, ref["control"]) < FWD_TOL
    assert rel_err(cell.controls, rc.controls) < FWD_TOL
    assert rel_err(cell.memories, rc.memories) < FWD_TOL
    assert rel_err(cell.infos, rc.infos) < FWD_TOL
    for i in range(p):
        assert max_abs(cell.attentions["kb"][i], rc.attentions["kb"][i]) < 2e-6
        assert max_abs(cell.attentions["question"][i], rc.attentions["question"][i]) < 2e-6
        a = cell.attentions["kb"][i]
        assert float(a.min()) >= 0 and max_abs(a.sum(-1), torch.ones(B)) < 1e-5
        if cfg.writeSelfAtt:
            assert cell.attentions["self"][i].shape == (B, i + 1)
            assert max_abs(cell.attentions["self"][i], rc.attentions["self"][i]) < 2e-6
        if cfg.writeGate:
            assert max_abs(cell.attentions["gate"][i], rc.attentions["gate"][i]) < 2e-6


@pytest.mark.parametrize("name,B,S,N,d,p,train", [
    ("args", 3, 9, 196, 128, 2, False),
    ("args", 3, 9, 196, 128, 3, True),
    ("args", 2, 11, 49, 256, 2, True),
    ("args3", 3, 9, 49, 128, 4, True),
    ("args3", 2, 7, 30, 128, 3, False),
    ("args4", 3, 9, 49, 128, 3, True),
    ("args1", 3, 9, 49, 128, 4, True),
    ("args1", 2, 7, 30, 128, 3, False),
    ("args", 2, 7, 30, 128, 18, True),        # more than 16 steps: the deferred dKB launch adds att (x) dinfo in two chunks
    ("args", 2, 5, 20, 128, 1, True),         # a single step
    ("args", 3, 9, 196, 512, 2, True),        # d = 512, 588 rows: the chain kernels' 16-row tiles (14 tiles per question: dy through dc_reduce)
    ("args", 24, 7, 196, 512, 2, True),       # 4704 rows: 32-row tiles (dy partials summed by the linear)
    ("args1", 5, 7, 49, 512, 3, True),        # 245 rows, recurrent control: dc inside the loop
])
def test_backward_matches_oracle_autograd(macx, dev, name, B, S, N, d, p, train):
    cfg, vq, words, lengths, kb = make_case(name, B, S, N, d, p)
    cell, params, (vqd, wd, kbd) = build_cell(macx, dev, cfg, vq, words, lengths, kb, train, seed=5, requires_grad=True)
    g = torch.Generator().manual_seed(9)
    dmem = torch.randn(B, d, generator=g) / B
    dctl = torch.randn(B, d, generator=g) / B
    state = cell.run()
    loss = (state.memory * dmem.to(dev)).sum() + (state.control * dctl.to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()
    ref = oracle_run(cfg, params.to_reference_dict(), vq, words, lengths, kb, train=train, seed=5, need_grad=True,
                     d_memory=dmem, d_control=dctl)
    assert rel_err(state.memory, ref["memory"]) < FWD_TOL
    rvq, rwords, rkb = ref["inputs"]
    errs = {"vecQuestions": rel_err(vqd.grad, rvq.grad), "words": rel_err(wd.grad, rwords.grad),
            "knowledgeBase": rel_err(kbd.grad, rkb.grad)}
    names = macx.params.reference_names(cfg, p)
    for f in params.fields:
        gt = getattr(params, f).grad
        assert gt is not None, f
        for refname, idx in names[f]:
            rg = ref["params"][refname].grad
            got = gt if idx is None else gt[idx]
            # d/d(logit bias) of a softmax is analytically zero: compare absolutely (fp32 round-off ~1e-7)
            floor = 5e-2 if refname.endswith("linearLayerlogits/biases/bias") else 1e-6
            errs[refname] = rel_err(got.reshape(rg.shape), rg, floor=floor)
    bad = {k: v for k, v in errs.items() if not (v < GRAD_TOL)}
    assert not bad, bad


@pytest.mark.parametrize("name,B,S,N,d,p,gemm", [
    ("args", 3, 9, 49, 128, 3, None),          # the H2 family below the chain kernels' width
    ("args", 3, 9, 196, 512, 2, None),         # the chain kernels
    ("args4", 3, 9, 49, 128, 3, "split"),      # kb_dropout_kernel + the write unit's sites
    ("args1", 3, 9, 49, 128, 3, "native"),
])
def test_mask_word_matches_oracle(macx, dev, name, B, S, N, d, p, gemm):
    """macx_dropout.mask_word: one device word XORed into every site key when the kernels run.  Forward, every gradient
    against the oracle on the masks of (seed, word); word 0 is bit for bit the run without a word."""
    cfg, vq, words, lengths, kb = make_case(name, B, S, N, d, p, writeDropout=0.9)
    word = 0xC0FFEE11
    wt = torch.tensor([word - (1 << 32)], dtype=torch.int32, device=dev)

    def run(mask_word):
        params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(5)).to(dev)
        vqd, wd, kbd = [t.to(dev).requires_grad_(True) for t in (vq, words, kb)]
        cell = macx.MACCell(vqd, wd, wd, lengths.to(dev), kbd, cfg.memoryDropout, cfg.readDropout, cfg.writeDropout, B, True,
                            config=cfg, params=params, seed=5, gemm=gemm, mask_word=mask_word)
        state = cell.run()
        (state.memory * dmem.to(dev)).sum().backward()
        torch.cuda.synchronize()
        return state, params, (vqd, wd, kbd)

    dmem = torch.randn(B, d, generator=torch.Generator().manual_seed(9)) / B
    state, params, (vqd, wd, kbd) = run(wt)
    ref = oracle_run(cfg, params.to_reference_dict(), vq, words, lengths, kb, train=True, seed=5, need_grad=True, d_memory=dmem, word=word)
    assert rel_err(state.memory, ref["memory"]) < FWD_TOL
    rvq, rwords, rkb = ref["inputs"]
    assert rel_err(kbd.grad, rkb.grad) < GRAD_TOL and rel_err(wd.grad, rwords.grad) < GRAD_TOL and rel_err(vqd.grad, rvq.grad) < GRAD_TOL
    names = macx.params.reference_names(cfg, p)
    bad = {}
    for f in params.fields:
        for refname, idx in names[f]:
            rg = ref["params"][refname].grad
            gt = getattr(params, f).grad
            got = gt if idx is None else gt[idx]
            e = rel_err(got.reshape(rg.shape), rg, floor=5e-2 if refname.endswith("linearLayerlogits/biases/bias") else 1e-6)
            if not e < GRAD_TOL:
                bad[refname] = e
    assert not bad, bad
    # a different stream than the plain seed's ...
    plain, pparams, (_, _, pkb) = run(None)
    assert not torch.equal(plain.memory, state.memory)
    # ... and word 0 IS the plain seed's stream
    zero, zparams, (_, _, zkb) = run(torch.zeros(1, dtype=torch.int32, device=dev))
    assert torch.equal(zero.memory, plain.memory) and torch.equal(zkb.grad, pkb.grad)
    assert all(torch.equal(a.grad, b.grad) for a, b in zip(zparams.tensors(), pparams.tensors()))


def test_stepwise_final_state_carries_gradient(macx, dev):
    """The model.py:453-458 loop, unchanged, trains: the last step's state has the autograd edge."""
    cfg, vq, words, lengths, kb = make_case("args", 2, 6, 49, 128, 2)
    cell, params, (vqd, wd, kbd) = build_cell(macx, dev, cfg, vq, words, lengths, kb, False, requires_grad=True)
    state = cell.zero_state(2)
    for i in range(cfg.netLength):
        cell.iteration = i
        _, state = cell(cell.none, state)
    state.memory.sum().backward()
    torch.cuda.synchronize()
    cell2, params2, (vq2, w2, kb2) = build_cell(macx, dev, cfg, vq, words, lengths, kb, False, requires_grad=True)
    s2 = cell2.run()
    s2.memory.sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(state.memory, s2.memory)
    assert torch.equal(kbd.grad, kb2.grad)
    assert torch.equal(params.projX_W.grad, params2.projX_W.grad)


def test_native_gemm_mode_matches_oracle(macx, dev, native_gemm):
    """The same forward/backward parity with the knowledge-base GEMMs on the native f32 MFMA kernel."""
    test_backward_matches_oracle_autograd(macx, dev, "args", 3, 7, 196, 128, 2, True)
    test_forward_stepwise_matches_oracle(macx, dev, "args4", 2, 5, 49, 128, 2, True)


def test_headline_shape_forward_backward(macx, dev):
    """BASELINE configs[1] shape (B=64,S=50,N=196,d=512,p=4): parity vs the fp32 oracle + invariants."""
    cfg, vq, words, lengths, kb = make_case("args", 64, 50, 196, 512, 4)
    cell, params, (vqd, wd, kbd) = build_cell(macx, dev, cfg, vq, words, lengths, kb, True, seed=1234, requires_grad=True)
    state = cell.run()
    dmem = torch.randn(64, 512, generator=torch.Generator().manual_seed(1)) / 64
    (state.memory * dmem.to(dev)).sum().backward()
    torch.cuda.synchronize()
    sl = slice(16, 24)            # fp64 oracle on a slice (questions are independent); the full-size all-gradient check is
    ref = oracle_run(cfg, params.to_reference_dict(), vq[sl], words[sl], lengths[sl], kb[sl], train=True, seed=1234, b0=16,
                     dtype=torch.float64, need_grad=True, d_memory=dmem[sl])     # tests/test_gpu_configs.py
    assert rel_err(state.memory[sl], ref["memory"]) < 1e-4
    assert rel_err(kbd.grad[sl], ref["inputs"][2].grad) < 2e-4
    # determinism: a second identical run is bit-identical (no float atomics anywhere)
    cell2, params2, (vq2, w2, kb2) = build_cell(macx, dev, cfg, vq, words, lengths, kb, True, seed=1234, requires_grad=True)
    s2 = cell2.run()
    (s2.memory * dmem.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(state.memory, s2.memory)
    assert torch.equal(kbd.grad, kb2.grad)
    assert torch.equal(params.memKbProj_W.grad, params2.memKbProj_W.grad)


@pytest.mark.parametrize("over", [
    dict(controlFeedPrevAtt=False),                                   # feed the continuous control instead
    dict(controlFeedInputs=False, controlContAct="NON"),              # single contControl layer on the previous control only
    dict(controlFeedPrevAtt=False, writeSelfAtt=True, writeSelfAttMod="CONT", writeGate=True),
    dict(writeSelfAtt=True, writeSelfAttMod="NON"),
])
def test_recurrent_control_variants(macx, dev, over):
    B, S, N, d, p = 3, 8, 30, 128, 3
    cfg, vq, words, lengths, kb = make_case("args1", B, S, N, d, p, **over)
    cell, params, (vqd, wd, kbd) = build_cell(macx, dev, cfg, vq, words, lengths, kb, True, seed=3, requires_grad=True)
    g = torch.Generator().manual_seed(2)
    dmem = torch.randn(B, d, generator=g) / B
    dctl = torch.randn(B, d, generator=g) / B
    state = cell.run()
    ((state.memory * dmem.to(dev)).sum() + (state.control * dctl.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    ref = oracle_run(cfg, params.to_reference_dict(), vq, words, lengths, kb, train=True, seed=3, need_grad=True,
                     d_memory=dmem, d_control=dctl)
    assert rel_err(state.memory, ref["memory"]) < FWD_TOL and rel_err(state.control, ref["control"]) < FWD_TOL
    zero_if_none = lambda g, like: torch.zeros_like(like) if g is None else g      # an input this variant never reads
    rvq, rw, rkb = ref["inputs"]
    errs = {"vecQuestions": rel_err(vqd.grad, zero_if_none(rvq.grad, rvq)), "words": rel_err(wd.grad, zero_if_none(rw.grad, rw)),
            "knowledgeBase": rel_err(kbd.grad, rkb.grad)}
    names = macx.params.reference_names(cfg, p)
    for f in params.fields:
        for refname, idx in names[f]:
            rg = zero_if_none(ref["params"][refname].grad, ref["params"][refname])
            got = getattr(params, f).grad
            got = got if idx is None else got[idx]
            floor = 5e-2 if refname.endswith("linearLayerlogits/biases/bias") else 1e-6
            errs[refname] = rel_err(got.reshape(rg.shape), rg, floor=floor)
    bad = {k: v for k, v in errs.items() if not (v < GRAD_TOL)}
    assert not bad, bad


def test_kernel_family_is_a_per_cell_option(macx, dev):
    """macx_opts.gemm_family: three cells of ONE process on the three kernel families, forward + backward, without touching
    the process default; every family within the parity tolerance of the oracle."""
    cfg, vq, words, lengths, kb = make_case("args", 3, 9, 49, 128, 2)
    L = macx._lib.lib()
    default = L.macx_gemm_mode(-1)
    params = macx.MACCellParams(cfg, 2, generator=torch.Generator().manual_seed(5)).to(dev)
    ref = oracle_run(cfg, params.to_reference_dict(), vq, words, lengths, kb, train=True, seed=3, need_grad=True,
                     d_memory=torch.ones(3, 128))
    for fam in ("h2", "split", "native"):
        for t in params.tensors():
            t.grad = None
        kbd = kb.to(dev).requires_grad_(True)
        cell = macx.MACCell(vq.to(dev), words.to(dev), words.to(dev), lengths.to(dev), kbd, cfg.memoryDropout, cfg.readDropout,
                            cfg.writeDropout, 3, True, config=cfg, params=params, seed=3, gemm=fam)
        assert cell.opts.gemm_family == {"native": 1, "split": 2, "h2": 3}[fam]
        state = cell.run()
        state.memory.sum().backward()
        torch.cuda.synchronize()
        assert L.macx_gemm_mode(-1) == default
        assert rel_err(state.memory, ref["memory"]) < FWD_TOL, fam
        assert rel_err(kbd.grad, ref["inputs"][2].grad) < GRAD_TOL, fam
        assert rel_err(params.memKbProj_W.grad, ref["params"]["MACnetwork/MACCell/read/linearLayermemKbProj/weights/weight"].grad) < GRAD_TOL, fam
