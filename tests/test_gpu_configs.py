"""-m gpu: the configurations BASELINE.json names, at full size, against the fp64 oracle: final state, attentions and EVERY
gradient (all parameters, knowledge base, words, question vectors).  Questions are independent inside the cell, so the oracle
runs the batch in chunks of questions (same global question indices -> same dropout masks) and adds the parameter gradients
up.  Every tensor's observed error is recorded (helpers.check_margin -> gpurun_out/parity_margins.json, committed as
profiles/rNN_parity_margins.json) and bounded by min(tolerance, 3 x the smallest committed observation of that tensor over all
rounds); a tensor no record knows fails."""
import pytest
import torch

from oracle import mac_oracle as mo
from helpers import make_case, oracle_run, rel_err, max_abs, check_margin
from test_gpu_cell import build_cell

pytestmark = pytest.mark.gpu

FWD_TOL = 2e-5      # final memory / control, relative to the largest entry
ATT_TOL = 2e-6      # attention weights, absolute
GRAD_TOL = 3e-5     # every gradient, relative to the tensor's largest entry (observed worst: 1.12e-5, the bias of projX at B = 128, p = 16)


def oracle_chunked_grads(cfg, ref_params, vq, words, lengths, kb, train, seed, d_memory, chunk=8, dtype=torch.float64, b0=0):
    """fp64 oracle over the whole batch in chunks of questions (questions are independent; parameter gradients add up;
    each chunk sees the masks of its global question indices).  Bounded memory: one chunk's graph at a time."""
    B = vq.shape[0]
    pg = {k: torch.zeros(v.shape, dtype=dtype) for k, v in ref_params.items()}
    mem, ctl, gvq, gw, gkb, att = [], [], [], [], [], []
    for lo in range(0, B, chunk):
        sl = slice(lo, min(B, lo + chunk))
        r = oracle_run(cfg, ref_params, vq[sl], words[sl], lengths[sl], kb[sl], train=train, seed=seed, b0=b0 + lo, dtype=dtype,
                       need_grad=True, d_memory=d_memory[sl])
        for k, v in r["params"].items():
            if v.grad is not None:
                pg[k] += v.grad
        mem.append(r["memory"].detach()); ctl.append(r["control"].detach())
        att.append(torch.stack([a.detach() for a in r["cell"].attentions["kb"]]))
        a, b, c = r["inputs"]
        gvq.append(a.grad); gw.append(b.grad); gkb.append(c.grad)
        del r
    return dict(memory=torch.cat(mem), control=torch.cat(ctl), vq=torch.cat(gvq), words=torch.cat(gw), kb=torch.cat(gkb), params=pg,
                att_kb=torch.cat(att, dim=1))


def check_full(macx, dev, tag, name, B, S, N, d, p, seed, b0=0):
    """one configuration at full size: whole batch through the HIP cell, whole batch through the chunked fp64 oracle"""
    cfg, vq, words, lengths, kb = make_case(name, B, S, N, d, p)
    cell, params, (vqd, wd, kbd) = build_cell(macx, dev, cfg, vq, words, lengths, kb, True, seed=seed, b0=b0, requires_grad=True)
    state = cell.run()
    gm = torch.randn(B, d, generator=torch.Generator().manual_seed(1)) / B
    (state.memory * gm.to(dev)).sum().backward()
    torch.cuda.synchronize()
    # invariants over the whole batch
    assert torch.isfinite(state.memory).all() and torch.isfinite(kbd.grad).all()
    for i in range(p):
        a = cell.attentions["kb"][i]
        assert float(a.min()) >= 0 and float((a.sum(-1) - 1).abs().max()) < 1e-5
        q = cell.attentions["question"][i]
        pad = (torch.arange(S, device=dev).unsqueeze(0) >= lengths.to(dev).unsqueeze(1))
        assert float(q[pad].abs().sum()) == 0.0
    assert cell.controls.shape == (B, p + 1, d) and cell.memories.shape == (B, p + 1, d)
    ref = oracle_chunked_grads(cfg, params.to_reference_dict(), vq, words, lengths, kb, True, seed, gm, b0=b0)
    errs = {"memory": (rel_err(state.memory, ref["memory"]), FWD_TOL), "control": (rel_err(state.control, ref["control"]), FWD_TOL),
            "att_kb(abs)": (max_abs(torch.stack(list(cell.attentions["kb"])), ref["att_kb"]), ATT_TOL),
            "d_vecQuestions": (rel_err(vqd.grad, ref["vq"]), GRAD_TOL), "d_words": (rel_err(wd.grad, ref["words"]), GRAD_TOL),
            "d_knowledgeBase": (rel_err(kbd.grad, ref["kb"]), GRAD_TOL)}
    names = macx.params.reference_names(cfg, p)
    for f in params.fields:
        gt = getattr(params, f).grad
        for refname, idx in names[f]:
            rg = ref["params"][refname]
            got = gt if idx is None else gt[idx]
            # (the logits biases shift every logit of a softmax alike: their gradient is round-off around zero, compared absolutely)
            floor = 5e-2 if refname.endswith("linearLayerlogits/biases/bias") else 1e-6
            errs["d_" + refname] = (rel_err(got.reshape(rg.shape), rg, floor=floor), GRAD_TOL)
    bad = {}
    for k, (e, tol) in errs.items():
        ok, bound = check_margin("%s/%s" % (tag, k), e, tol, require_record=True)
        if not ok:
            bad[k] = (e, bound)
    assert not bad, bad
    return cell, cfg


def test_metric_configuration_all_gradients_fp64(macx, dev):
    """BASELINE.json's metric configuration at FULL size -- B=64, S=50, N=196, d=512, p=12, training-mode dropout: final state and
    EVERY gradient (all parameters incl. the deferred all-steps dW2 / dWx contractions over p*B*N = 150 528 rows)."""
    check_full(macx, dev, "metric_b64_p12", "args", 64, 50, 196, 512, 12, seed=1234)


def test_clevr_training_b128_p12(macx, dev):
    """BASELINE configs[2]: netLength 12, batch 128, fwd+bwd in training mode."""
    check_full(macx, dev, "config2_b128_p12", "args", 128, 50, 196, 512, 12, seed=31)


def test_dp_shard_b128_p16(macx, dev):
    """BASELINE configs[3] per-GPU shard: 128 questions, netLength 16; the shard's b0 selects the global masks."""
    check_full(macx, dev, "config3_shard_b128_p16", "args", 128, 50, 196, 512, 16, seed=9, b0=384)


@pytest.mark.parametrize("name", ["args3", "args4"])
def test_gqa_shape_7x7_p4(macx, dev, name):
    """BASELINE configs[4]: KB = 7x7 (N = 49), netLength 4, write self-attention / memory gate, B = 64, d = 512."""
    cell, cfg = check_full(macx, dev, "config4_gqa_%s_p4" % name, name, 64, 50, 49, 512, 4, seed=31)
    if cfg.writeSelfAtt:
        assert cell.attentions["self"][3].shape == (64, 4)
    if cfg.writeGate:
        z = cell.attentions["gate"][0]
        assert float(z.min()) > 0 and float(z.max()) < 1
