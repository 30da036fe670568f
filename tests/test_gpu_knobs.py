"""-m gpu: the A/B knobs of macx_debug_set select different kernels for the same mathematics -- every route must give the
gradients of the default one (summation orders differ, so not bit-for-bit):
  key 4 = 0   the read unit as per-product launches instead of the chain kernels
  key 5 = 0   S_b = X_b^T dI1_b once per step (delivering dy) instead of dy from the chain kernel + one deferred launch
  key 6 = 1   the per-step dKB contraction on the internal side queue (fork / join by events) with accumulation in HBM
  key 8 = 2 / 0  dW1a / dW1b from the dual-A contraction over the kept X * y (d % 256 == 0: the d = 256 and d = 512 cases) / from the
              128 x 128 per-question S_b kernel instead of the 128 x 256 one
  key 14 = 0  the merged dKB launch folding its block accumulator per 128-wide K block instead of once per step (d = 512)
  key 12 = 1  the long-reduction [B,d] linears on 8 waves per workgroup instead of 4 (the cross-wave sum has a different order)"""
import pytest
import torch

from helpers import make_case, rel_err
from test_gpu_cell import build_cell

pytestmark = pytest.mark.gpu


def run(macx, dev, name, B, S, N, d, p):
    cfg, vq, words, lengths, kb = make_case(name, B, S, N, d, p)
    cell, params, (vqd, wd, kbd) = build_cell(macx, dev, cfg, vq, words, lengths, kb, True, seed=11, requires_grad=True)
    state = cell.run()
    gm = torch.randn(B, d, generator=torch.Generator().manual_seed(3)).to(dev)
    (state.memory * gm).sum().backward()
    torch.cuda.synchronize()
    out = {"memory": state.memory.detach().clone(), "d_kb": kbd.grad.clone(), "d_words": wd.grad.clone(), "d_vq": vqd.grad.clone()}
    for f in params.fields:
        out["d_" + f] = getattr(params, f).grad.clone()
    return out


@pytest.mark.parametrize("key,value", [(4, 0), (5, 0), (6, 1), (8, 2), (8, 0), (10, 2), (12, 1), (14, 0)])
@pytest.mark.parametrize("name,B,S,N,d,p", [("args", 5, 9, 196, 128, 3), ("args1", 4, 9, 49, 256, 4), ("args", 3, 7, 196, 512, 3)])
def test_knob_routes_agree(macx, dev, key, value, name, B, S, N, d, p):
    lib = macx._lib.lib()
    defaults = {4: 1, 5: 1, 6: 0, 8: 1, 10: 3, 12: 0, 14: 1}
    ref = run(macx, dev, name, B, S, N, d, p)
    assert lib.macx_debug_set(key, value) == 0
    try:
        got = run(macx, dev, name, B, S, N, d, p)
    finally:
        assert lib.macx_debug_set(key, defaults[key]) == 0
    for k in ref:
        # (the logits bias shifts every logit of a softmax alike: its gradient is round-off around zero)
        floor = 0.2 if k.endswith("Logits_b") else 1e-6
        assert rel_err(got[k], ref[k], floor=floor) < 2e-5, k


@pytest.mark.parametrize("key,values,default", [(10, (0, 1), 2), (11, (32, 128, 256), 0), (13, (0,), 1)])
@pytest.mark.parametrize("name,B,S,N,d,p", [("args", 3, 7, 196, 512, 2), ("args1", 4, 9, 49, 256, 3), ("args", 2, 5, 33, 512, 5),
                                            ("args3", 3, 6, 49, 128, 4), ("args", 64, 7, 20, 128, 3)])
def test_launch_shape_knobs_are_bit_identical(macx, dev, key, values, default, name, B, S, N, d, p):
    """key 10: the all-steps weight-gradient contractions with round 4's loop (0), with a buffer's halves re-requested inside the
    iteration (1) and with dW2 / dWx as one launch on top (2, the default) multiply the same fragments in the same order.
    key 11: two dependent [B,d] linears as one launch with a device-scope barrier between them (32 / 128 / 256 workgroups) or as two
    launches (0, the default: the pairs measured slower) compute the same tiles with the same code.
    key 13: sb_h2w_kernel as one stage stream over all steps (1) or drained and re-primed per step (0): same products, folds, order.
    Final memory and every gradient bit for bit."""
    lib = macx._lib.lib()
    if key == 10:
        assert lib.macx_debug_set(10, 2) == 0       # (the default, 3, halves the reduction splits: another summation order)
    ref = run(macx, dev, name, B, S, N, d, p)
    try:
        for v in values:
            assert lib.macx_debug_set(key, v) == 0
            got = run(macx, dev, name, B, S, N, d, p)
            for k in ref:
                assert torch.equal(got[k], ref[k]), (v, k)
    finally:
        assert lib.macx_debug_set(key, 3 if key == 10 else default) == 0
