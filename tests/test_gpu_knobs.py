"""-m gpu: the A/B hooks of macx_opts.tune (include/macx.h MACX_TUNE_*, per call) select different kernels for the same mathematics --
every route must give the gradients of the default one (summation orders differ, so not bit-for-bit):
  chain = 0       the read unit as per-product launches instead of the chain kernels
  sb_defer = 0    S_b = X_b^T dI1_b once per step (delivering dy) instead of dy from the chain kernel + one deferred launch
  sb_wide = 0     dW1a / dW1b from the 128 x 128 per-question S_b kernel instead of the 128 x 256 one
  wgrad_pipe = 2  dW2 / dWx with the full number of reduction splits (the default, 3, halves them)
  dkb_uni = 0     the merged dKB launch folding its block accumulator per 128-wide K block instead of once per step (d = 512)
  chain_kv = 0    the chain kernels' K loop with the activation fragments requested in front of a slice's products
The table travels in the opts of the call: two cells with different tables in one process never see each other's."""
import pytest
import torch

from helpers import make_case, rel_err
from test_gpu_cell import build_cell

pytestmark = pytest.mark.gpu


def run(macx, dev, name, B, S, N, d, p, tune=None):
    cfg, vq, words, lengths, kb = make_case(name, B, S, N, d, p)
    cell, params, (vqd, wd, kbd) = build_cell(macx, dev, cfg, vq, words, lengths, kb, True, seed=11, requires_grad=True, tune=tune)
    state = cell.run()
    gm = torch.randn(B, d, generator=torch.Generator().manual_seed(3)).to(dev)
    (state.memory * gm).sum().backward()
    torch.cuda.synchronize()
    out = {"memory": state.memory.detach().clone(), "d_kb": kbd.grad.clone(), "d_words": wd.grad.clone(), "d_vq": vqd.grad.clone()}
    for f in params.fields:
        out["d_" + f] = getattr(params, f).grad.clone()
    return out


@pytest.mark.parametrize("key,value", [("chain", 0), ("sb_defer", 0), ("sb_wide", 0), ("wgrad_pipe", 2), ("dkb_uni", 0), ("chain_kv", 0)])
@pytest.mark.parametrize("name,B,S,N,d,p", [("args", 5, 9, 196, 128, 3), ("args1", 4, 9, 49, 256, 4), ("args", 3, 7, 196, 512, 3)])
def test_knob_routes_agree(macx, dev, key, value, name, B, S, N, d, p):
    ref = run(macx, dev, name, B, S, N, d, p)
    got = run(macx, dev, name, B, S, N, d, p, tune={key: value})
    for k in ref:
        # (the logits bias shifts every logit of a softmax alike: its gradient is round-off around zero)
        floor = 0.2 if k.endswith("Logits_b") else 1e-6
        assert rel_err(got[k], ref[k], floor=floor) < 2e-5, k


@pytest.mark.parametrize("key,values,base", [("wgrad_pipe", (0, 1), {"wgrad_pipe": 2}), ("sb_cont", (0,), {})])
@pytest.mark.parametrize("name,B,S,N,d,p", [("args", 3, 7, 196, 512, 2), ("args1", 4, 9, 49, 256, 3), ("args", 2, 5, 33, 512, 5),
                                            ("args3", 3, 6, 49, 128, 4), ("args", 64, 7, 20, 128, 3)])
def test_launch_shape_knobs_are_bit_identical(macx, dev, key, values, base, name, B, S, N, d, p):
    """wgrad_pipe: the all-steps weight-gradient contractions with round 4's loop (0), with a buffer's halves re-requested inside the
    iteration (1) and with dW2 / dWx as one launch on top (2) multiply the same fragments in the same order (the default, 3, halves the
    reduction splits: another summation order, so the comparison is made against 2).
    sb_cont: sb_h2w_kernel as one stage stream over all steps (1, the default) or drained and re-primed per step (0): same products,
    folds, order -- the bit-identity gate of that kernel's hand-counted waits: mandatory on a compiler upgrade.
    Final memory and every gradient bit for bit."""
    ref = run(macx, dev, name, B, S, N, d, p, tune=dict(base))
    for v in values:
        got = run(macx, dev, name, B, S, N, d, p, tune=dict(base, **{key: v}))
        for k in ref:
            assert torch.equal(got[k], ref[k]), (v, k)


def test_tables_of_two_cells_do_not_leak(macx, dev):
    """A cell on a non-default route and a default cell run alternately in one process: each call reads the table in ITS opts (there is
    no state in the library), so the default cell's results are bit for bit those of a process that never saw the other table."""
    shape = ("args", 3, 7, 196, 512, 2)
    ref = run(macx, dev, *shape)
    other = run(macx, dev, *shape, tune={"chain": 0, "sb_defer": 0})
    again = run(macx, dev, *shape)
    for k in ref:
        assert torch.equal(again[k], ref[k]), k
    assert any(not torch.equal(other[k], ref[k]) for k in ref)      # (the other route rounds differently: it really ran)


@pytest.mark.parametrize("name,B,S,N,d,p", [("args", 64, 7, 196, 512, 3), ("args", 43, 5, 196, 512, 4), ("args3", 64, 5, 196, 512, 2),
                                            ("args", 50, 5, 170, 512, 5)])
def test_dkb_on_idle_cus_agrees_with_merged_launch(macx, dev, name, B, S, N, d, p):
    """dkb_fill: dKB of step i + 1 as jobs on the CUs chain_bwd's launch of step i leaves idle + a closing launch (default, 3 jobs per
    filler workgroup) against ONE merged launch over all steps (0).  1 and 2 jobs per workgroup move the window of left-out tiles
    (nskip) and with it which launch stores a tile first.  Only dKB may differ (one fp32 rounding per step instead of one per
    launch); everything else bit for bit -- the fillers share a launch with the chain tiles and must not disturb them."""
    ref = run(macx, dev, name, B, S, N, d, p, tune={"dkb_fill": 0})
    for v in (3, 1, 2):
        got = run(macx, dev, name, B, S, N, d, p, tune={"dkb_fill": v})
        for k in ref:
            if k == "d_kb":
                assert rel_err(got[k], ref[k], floor=1e-6) < 2e-6, (v, k)
            else:
                assert torch.equal(got[k], ref[k]), (v, k)


@pytest.mark.parametrize("name,B,S,N,d,p", [("args", 64, 7, 196, 512, 3), ("args", 43, 5, 196, 512, 2), ("args3", 50, 5, 170, 512, 4),
                                            ("args", 8, 7, 196, 512, 3), ("args", 24, 5, 196, 512, 3), ("args", 5, 5, 49, 512, 4)])
def test_fillers_of_chain_fwd_are_bit_identical(macx, dev, name, B, S, N, d, p):
    """pre_fill: the filler workgroups of chain_fwd's launch of step i run step i - 1's write unit, step i's y = md Wy + by (results
    handed to the tiles of the SAME launch through counters and agent-scope loads / stores) and stage 0 of step i + 1 (default 1;
    2: without the write unit) against launches / stages of their own (0): the same
    arithmetic in the same order -- final memory and every gradient bit for bit, several times over (a stale read would show)."""
    ref = run(macx, dev, name, B, S, N, d, p, tune={"pre_fill": 0})
    for rep in range(2):
        for v in (1, 2):
            got = run(macx, dev, name, B, S, N, d, p, tune={"pre_fill": v})
            for k in ref:
                assert torch.equal(got[k], ref[k]), (rep, v, k)


@pytest.mark.parametrize("name,B,S,N,d,p", [("args", 64, 7, 196, 512, 4), ("args", 128, 5, 196, 512, 2)])
def test_fillers_of_chain_fwd_inference_and_two_rounds(macx, dev, name, B, S, N, d, p):
    """Evaluation runs (nothing kept, no dropout: the fillers have no stage 0 to do, only the write linear and y) and B = 128 (392 tiles
    = two rounds of the chip, 120 fillers dispatched in front of them): final state and attentions bit for bit against pre_fill = 0,
    and the run's fail word (a tile or filler that gave up waiting) is zero."""
    cfg, vq, words, lengths, kb = make_case(name, B, S, N, d, p)
    out = {}
    for v in (0, 1):
        with torch.no_grad():
            cell, params, _ = build_cell(macx, dev, cfg, vq, words, lengths, kb, False, seed=11, tune={"pre_fill": v})
            state = cell.run()
            torch.cuda.synchronize()
            out[v] = [state.memory.clone(), state.control.clone()] + [a.clone() for a in cell.attentions["kb"]]
            assert int(cell._run.saved.view(torch.int32)[-576 + 63]) == 0
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)
