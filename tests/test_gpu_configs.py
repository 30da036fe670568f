"""-m gpu: the configurations BASELINE.json names, at full size.  Questions are independent inside the
cell, so the oracle checks a SLICE of the batch (same global question indices -> same dropout masks) while
the HIP path runs the whole batch; the rest of the batch is covered by size-independent properties."""
import pytest
import torch

from oracle import mac_oracle as mo
from helpers import make_case, oracle_run, rel_err, max_abs
from test_gpu_cell import build_cell

pytestmark = pytest.mark.gpu


def run_and_check_slice(macx, dev, name, B, S, N, d, p, lo, hi, train=True, seed=31, tol=2e-4):
    cfg, vq, words, lengths, kb = make_case(name, B, S, N, d, p)
    cell, params, (vqd, wd, kbd) = build_cell(macx, dev, cfg, vq, words, lengths, kb, train, seed=seed, requires_grad=True)
    state = cell.run()
    gm = torch.randn(B, d, generator=torch.Generator().manual_seed(3)) / B
    (state.memory * gm.to(dev)).sum().backward()
    torch.cuda.synchronize()
    # whole batch: invariants
    assert torch.isfinite(state.memory).all() and torch.isfinite(kbd.grad).all()
    for i in range(p):
        a = cell.attentions["kb"][i]
        assert float(a.min()) >= 0 and float((a.sum(-1) - 1).abs().max()) < 1e-5
        q = cell.attentions["question"][i]
        pad = (torch.arange(S, device=dev).unsqueeze(0) >= lengths.to(dev).unsqueeze(1))
        assert float(q[pad].abs().sum()) == 0.0
    assert cell.controls.shape == (B, p + 1, d) and cell.memories.shape == (B, p + 1, d)
    # slice [lo, hi) against the fp64 oracle with the masks of global questions lo..hi-1
    sl = slice(lo, hi)
    ref = oracle_run(cfg, params.to_reference_dict(), vq[sl], words[sl], lengths[sl], kb[sl], train=train, seed=seed, b0=lo,
                     dtype=torch.float64, need_grad=True, d_memory=gm[sl])
    assert rel_err(state.memory[sl], ref["memory"]) < tol
    assert rel_err(state.control[sl], ref["control"]) < tol
    assert rel_err(kbd.grad[sl], ref["inputs"][2].grad) < tol
    assert rel_err(wd.grad[sl], ref["inputs"][1].grad) < tol
    for i in range(p):
        assert max_abs(cell.attentions["kb"][i][sl], ref["cell"].attentions["kb"][i]) < 1e-5
    return cell, cfg


@pytest.mark.parametrize("name", ["args3", "args4"])
def test_gqa_shape_7x7_p4(macx, dev, name):
    """BASELINE configs[4]: KB = 7x7 (N = 49), netLength 4, write self-attention / memory gate, B = 64, d = 512."""
    cell, cfg = run_and_check_slice(macx, dev, name, 64, 50, 49, 512, 4, 10, 14)
    if cfg.writeSelfAtt:
        assert cell.attentions["self"][3].shape == (64, 4)
    if cfg.writeGate:
        z = cell.attentions["gate"][0]
        assert float(z.min()) > 0 and float(z.max()) < 1


def test_clevr_training_b128_p12(macx, dev):
    """BASELINE configs[2]: netLength 12, batch 128, fwd+bwd in training mode."""
    run_and_check_slice(macx, dev, "args", 128, 50, 196, 512, 12, 100, 102)


def test_dp_shard_b128_p16(macx, dev):
    """BASELINE configs[3] per-GPU shard: 128 questions, netLength 16; the shard's b0 selects the global masks."""
    cfg, vq, words, lengths, kb = make_case("args", 128, 50, 196, 512, 16)
    cell, params, (vqd, wd, kbd) = build_cell(macx, dev, cfg, vq, words, lengths, kb, True, seed=9, b0=384, requires_grad=True)
    state = cell.run()
    state.memory.sum().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(state.memory).all() and torch.isfinite(params.memKbProj_W.grad).all()
    ref = oracle_run(cfg, params.to_reference_dict(), vq[:2], words[:2], lengths[:2], kb[:2], train=True, seed=9, b0=384,
                     dtype=torch.float32)
    assert rel_err(state.memory[:2], ref["memory"]) < 3e-4


def oracle_chunked_grads(cfg, ref_params, vq, words, lengths, kb, train, seed, d_memory, chunk=8, dtype=torch.float64):
    """fp64 oracle over the whole batch in chunks of questions (questions are independent; parameter gradients add up;
    each chunk sees the masks of its global question indices).  Bounded memory: one chunk's graph at a time."""
    B = vq.shape[0]
    pg = {k: torch.zeros(v.shape, dtype=dtype) for k, v in ref_params.items()}
    mem, ctl, gvq, gw, gkb = [], [], [], [], []
    for lo in range(0, B, chunk):
        sl = slice(lo, min(B, lo + chunk))
        r = oracle_run(cfg, ref_params, vq[sl], words[sl], lengths[sl], kb[sl], train=train, seed=seed, b0=lo, dtype=dtype,
                       need_grad=True, d_memory=d_memory[sl])
        for k, v in r["params"].items():
            if v.grad is not None:
                pg[k] += v.grad
        mem.append(r["memory"].detach()); ctl.append(r["control"].detach())
        a, b, c = r["inputs"]
        gvq.append(a.grad); gw.append(b.grad); gkb.append(c.grad)
        del r
    return dict(memory=torch.cat(mem), control=torch.cat(ctl), vq=torch.cat(gvq), words=torch.cat(gw), kb=torch.cat(gkb), params=pg)


def test_metric_configuration_all_gradients_fp64(macx, dev):
    """BASELINE.json's metric configuration at FULL size -- B=64, S=50, N=196, d=512, p=12, training-mode dropout -- against
    the fp64 oracle: final state and EVERY gradient (all parameters incl. the deferred all-steps dW2 / dWx contractions
    over p*B*N = 150 528 rows, knowledge base, words, question vectors) within 2e-4 of the largest entry."""
    B, S, N, d, p = 64, 50, 196, 512, 12
    cfg, vq, words, lengths, kb = make_case("args", B, S, N, d, p)
    cell, params, (vqd, wd, kbd) = build_cell(macx, dev, cfg, vq, words, lengths, kb, True, seed=1234, requires_grad=True)
    state = cell.run()
    gm = torch.randn(B, d, generator=torch.Generator().manual_seed(1)) / B
    (state.memory * gm.to(dev)).sum().backward()
    torch.cuda.synchronize()
    ref = oracle_chunked_grads(cfg, params.to_reference_dict(), vq, words, lengths, kb, True, 1234, gm)
    assert rel_err(state.memory, ref["memory"]) < 1e-4 and rel_err(state.control, ref["control"]) < 1e-4
    errs = {"vecQuestions": rel_err(vqd.grad, ref["vq"]), "words": rel_err(wd.grad, ref["words"]),
            "knowledgeBase": rel_err(kbd.grad, ref["kb"])}
    names = macx.params.reference_names(cfg, p)
    for f in params.fields:
        gt = getattr(params, f).grad
        for refname, idx in names[f]:
            rg = ref["params"][refname]
            got = gt if idx is None else gt[idx]
            floor = 5e-2 if refname.endswith("linearLayerlogits/biases/bias") else 1e-6
            errs[refname] = rel_err(got.reshape(rg.shape), rg, floor=floor)
    bad = {k: v for k, v in errs.items() if not (v < 2e-4)}
    assert not bad, bad
