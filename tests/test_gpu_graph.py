"""The captured-graph inference path (macx.CapturedForward) must reproduce the eager run bit for bit, follow parameter
updates, and accept new inputs -- whether the process replays the graph or (self-check failed, see graph.py) falls back to
eager launches behind the same interface."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_captured_forward_equals_eager(macx, dev):
    B, S, N, d, p = 6, 7, 40, 128, 3
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
    params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(0)).to(dev)
    cap = macx.CapturedForward(cfg, params, B, S, N)

    def eager(vq, words, lengths, kb):
        with torch.no_grad():
            cell = macx.MACCell(vq, words, words, lengths, kb, 1.0, 1.0, 1.0, B, False, config=cfg, params=params)
            st = cell.run()
            return st.memory.clone(), [a.clone() for a in cell.attentions["kb"]]

    for seed in (1, 2):
        vq, words, lengths, kb = [t.to(dev) for t in macx.configs.synthetic_inputs(B, S, N, d, seed=seed)]
        ref, ref_att = eager(vq, words, lengths, kb)
        got = cap(vq, words, lengths, kb)
        torch.cuda.synchronize()
        assert torch.equal(ref, got)
        assert all(torch.equal(a, b) for a, b in zip(ref_att, cap.attentions["kb"]))
    # parameters are read at replay time
    with torch.no_grad():
        params.projX_W.mul_(1.5)
    ref, _ = eager(vq, words, lengths, kb)
    assert torch.equal(ref, cap.replay())


def test_eager_fallback_behind_the_same_interface(macx, dev):
    B, S, N, d, p = 4, 6, 33, 128, 2
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
    params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(0)).to(dev)
    cap = macx.CapturedForward(cfg, params, B, S, N, verify=False)
    cap.captured = False                       # what a failed self-check sets
    vq, words, lengths, kb = [t.to(dev) for t in macx.configs.synthetic_inputs(B, S, N, d, seed=3)]
    with torch.no_grad():
        cell = macx.MACCell(vq, words, words, lengths, kb, 1.0, 1.0, 1.0, B, False, config=cfg, params=params)
        ref = cell.run().memory.clone()
    got = cap(vq, words, lengths, kb)
    torch.cuda.synchronize()
    assert torch.equal(ref, got)
    assert all(torch.equal(a, b) for a, b in zip(cell.attentions["kb"], cap.attentions["kb"]))
