"""-m gpu: the documented maxima and minima of the fused cell (include/macx.h, macx_check: S <= 256, N <= 1024, p <= 32,
d <= 1024; a single question, a single knowledge-base cell, a single step) -- forward state and every gradient against
the fp64 oracle, and the first size past each limit is rejected with MACX_EINVAL rather than computed wrongly."""
import ctypes as C

import pytest
import torch

from helpers import make_case
from test_gpu_cell import test_backward_matches_oracle_autograd as parity

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,B,S,N,d,p,train", [
    ("args", 2, 5, 1024, 1024, 2, True),       # widest cell the kernels take: N = 1024 cells, d = 1024
    ("args4", 2, 256, 16, 128, 2, True),       # longest question
    ("args", 2, 6, 9, 128, 32, True),          # most reasoning steps (deferred dKB: 32 steps in chunks of 16)
    ("args3", 2, 6, 9, 128, 32, False),        # ... with self-attention over 32 histories
    ("args1", 1, 3, 1, 128, 1, True),          # one question, one knowledge-base cell, one step
    ("args", 1, 3, 209, 128, 2, True),         # one question whose knowledge base crosses a 208-row tile
    ("args", 64, 8, 196, 128, 1, False),       # the metric's batch and grid at the narrowest width
])
def test_extreme_shapes_match_the_oracle(macx, dev, name, B, S, N, d, p, train):
    parity(macx, dev, name, B, S, N, d, p, train)


@pytest.mark.parametrize("name,B,S,N,d,p,train", [
    ("args", 3, 6, 49, 144, 2, True),          # 144 -> 256 columns: chain kernels at d = 256, dropout indices at the logical width
    ("args", 5, 7, 196, 400, 3, True),         # 400 -> 512: the d = 512 tile geometry (980 rows: 16-row tiles)
    ("args4", 2, 6, 20, 200, 2, True),         # self-attention + gate (sigmoid(0) = 0.5 in the padded columns, times zeros)
    ("args1", 2, 5, 33, 72, 2, False),         # narrower than one granule, recurrent control, evaluation
])
def test_widths_that_are_not_multiples_of_128(macx, dev, name, B, S, N, d, p, train):
    """config.py:294-296 takes any width: a cell whose width is a multiple of 8 runs zero-padded to the kernels' 128-column
    granule (macx.cell.PaddedMACCell, macx_shapes.d_logical) with the UNPADDED cell's dropout masks -- states and every
    gradient against the fp64 oracle at the logical width."""
    parity(macx, dev, name, B, S, N, d, p, train)


def test_padded_cell_runs_on_the_h2_family_whatever_the_default(macx, dev):
    """the logical-width dropout index lives in the H2 kernels: a padded cell selects that family for itself (process default
    split / native included) and refuses an explicit other one"""
    cfg, vq, words, lengths, kb = make_case("args", 2, 5, 20, 144, 2)
    L = macx._lib.lib()
    before = L.macx_gemm_mode(-1)
    try:
        L.macx_gemm_mode(1)                       # process default: the 6-term bf16 split
        vqd, wd, kbd = [t.to(dev) for t in (vq, words, kb)]
        mk = lambda **kw: macx.MACCell(vqd, wd, wd, lengths.to(dev), kbd, 0.85, 0.85, 1.0, 2, True, config=cfg, seed=3, **kw)
        cell = mk()
        assert cell.run().memory.shape == (2, 144)
        with pytest.raises(macx.UnsupportedOptions):
            mk(gemm="split")
    finally:
        L.macx_gemm_mode(before)


def test_padded_cell_refuses_a_sigmoid_unit_activation(macx, dev):
    """sigmoid(0) = 0.5 would fill the padded columns (PaddedMACCell's 'exact zeros' argument does not hold): refused"""
    cfg, vq, words, lengths, kb = make_case("args", 2, 5, 20, 144, 2, readMemAct="SIGMOID")
    vqd, wd, kbd = [t.to(dev) for t in (vq, words, kb)]
    with pytest.raises(macx.UnsupportedOptions):
        macx.MACCell(vqd, wd, wd, lengths.to(dev), kbd, 0.85, 0.85, 1.0, 2, True, config=cfg, seed=3)


@pytest.mark.parametrize("over", [dict(S=257), dict(N=1025), dict(p=33), dict(d=1152), dict(d=192), dict(B=0), dict(d=256, d_logical=100),
                                  dict(d=256, d_logical=132)])
def test_first_size_past_each_limit_is_rejected(macx, over):
    cfg, *_ = make_case("args", 2, 4, 8, 128, 2)
    opts = macx.options.freeze(cfg)
    kw = dict(B=2, S=4, N=8, d=128, p=2, b0=0, d_logical=0)
    kw.update(over)
    sh = macx._lib.MacxShapes(**kw)
    L = macx._lib.lib()
    assert L.macx_check(C.byref(opts), C.byref(sh)) == -1            # MACX_EINVAL
    assert L.macx_saved_floats(C.byref(opts), C.byref(sh), 1) == 0
