"""-m gpu: unit-level parity of the HIP kernels, called through the C ABI, against the oracle / torch."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import dropout_hash as dh
from oracle import mac_oracle as mo
from helpers import default_gemm_mode, rel_err, max_abs

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr())


def test_dropout_stream_matches_numpy(macx, dev):
    L = macx._lib.lib()
    n = 100003
    for (seed, site, step, keep, first) in [(1234, 3, 0, 0.85, 0), (7, 5, 11, 0.5, 4000000000), (0, 1, 0, 1.0, 17)]:
        out = torch.empty(n, device=dev)
        macx._lib.check(L.macx_dropout_mask(seed, site, step, keep, first, n, _p(out), None), "mask")
        torch.cuda.synchronize()
        ref = dh.keep_mask(seed, site, step, keep, first, n)
        assert np.array_equal(out.cpu().numpy(), ref)
        if keep < 1:
            assert abs(ref.mean() - keep) < 0.01
        # under a run's mask word (macx_dropout.mask_word): read from device memory when the kernel runs
        for word in (0, 0x9E3779B9, 0xFFFFFFFF):
            wt = torch.tensor([word - (1 << 32) if word >= (1 << 31) else word], dtype=torch.int32, device=dev)
            macx._lib.check(L.macx_dropout_mask_w(seed, site, step, keep, first, n, _p(wt), _p(out), None), "mask_w")
            torch.cuda.synchronize()
            refw = dh.keep_mask(seed, site, step, keep, first, n, word=word)
            assert np.array_equal(out.cpu().numpy(), refw)
            assert (word == 0) == np.array_equal(refw, ref) or keep >= 1


@pytest.mark.parametrize("rows,k1,k2,nout,act", [(64, 512, 0, 512, "NON"), (64, 512, 512, 512, "TANH"),
                                                  (5, 128, 0, 128, "ELU"), (37, 256, 128, 208, "SIGMOID"), (64, 512, 0, 512, "RELU")])
def test_linear(macx, dev, rows, k1, k2, nout, act):
    L = macx._lib.lib()
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn(rows, k1, generator=g)
    x2 = torch.randn(rows, k2, generator=g) if k2 else None
    W = torch.randn(k1 + k2, nout, generator=g) / (k1 + k2) ** 0.5
    b = torch.randn(nout, generator=g)
    xin = x1 if x2 is None else torch.cat([x1, x2], dim=1)
    ref = xin.double() @ W.double() + b.double() + 0.25
    ref = {"NON": lambda t: t, "TANH": torch.tanh, "ELU": torch.nn.functional.elu, "SIGMOID": torch.sigmoid,
           "RELU": torch.relu}[act](ref)
    out = torch.empty(rows, nout, device=dev)
    x1d, Wd, bd = x1.to(dev), W.to(dev), b.to(dev)
    x2d = x2.to(dev) if x2 is not None else None
    wp = torch.empty_like(Wd)
    macx._lib.check(L.macx_pack_weight(_p(Wd), k1 + k2, nout, 0, _p(wp), None), "pack")
    macx._lib.check(L.macx_linear(_p(x1d), k1, _p(x2d) if x2d is not None else None, k2, rows, _p(wp), _p(bd), 0.25, nout,
                                  macx._lib.ACT[act], _p(out), None), "linear")
    torch.cuda.synchronize()
    assert max_abs(out, ref) < 2e-5


@pytest.mark.parametrize("B,N,d,keep", [(4, 196, 128, 1.0), (3, 196, 256, 0.85), (5, 49, 128, 0.85), (2, 14, 128, 0.5),
                                        (2, 100, 128, 0.85), (2, 300, 128, 1.0), (2, 30, 128, 0.85), (1, 250, 256, 0.85)])
@pytest.mark.parametrize("mode", [1, 0])
def test_kb_project(macx, dev, B, N, d, keep, mode):
    """X = dropout(KB) Wx + bx through the MFMA kernels (split-bf16 and native f32) vs fp64, asymmetric W (transpose-detecting)."""
    L = macx._lib.lib()
    L.macx_gemm_mode(mode)
    try:
        _kb_project_case(macx, dev, B, N, d, keep)
    finally:
        L.macx_gemm_mode(default_gemm_mode())


def _kb_project_case(macx, dev, B, N, d, keep):
    L = macx._lib.lib()
    g = torch.Generator().manual_seed(2)
    kb = torch.randn(B, N, d, generator=g)
    W = torch.randn(d, d, generator=g) / d ** 0.5
    b = torch.randn(d, generator=g)
    sh = macx._lib.MacxShapes(B=B, S=1, N=N, d=d, p=1, b0=3)
    dp = macx._lib.MacxDropout(keep_memory=1.0, keep_read=keep, keep_write=1.0, seed=99)
    out = torch.empty(B, N, d, device=dev)
    wp = torch.empty(2 * d * d, device=dev)
    kbd, Wd, bd = kb.to(dev), W.to(dev), b.to(dev)
    macx._lib.check(L.macx_pack_weight(_p(Wd), d, d, macx._lib.kb_pack_flags(), _p(wp), None), "pack")
    bits = torch.empty(B * N * d + B * N * d // 32 + 4, device=dev)
    macx._lib.check(L.macx_kb_project(C.byref(sh), C.byref(dp), 5, _p(kbd), _p(wp), _p(bd), _p(out), _p(bits), None), "kb_project")
    torch.cuda.synchronize()
    mask = torch.from_numpy(dh.mask_for(99, dh.SITE_READ_KB, 5, keep, (B, N, d), b0=3)).double()
    ref = ((kb.double() / keep) * mask) @ W.double() + b.double()
    assert rel_err(out, ref) < 2e-6


def test_split_gemm_error_is_fp32_class(macx, dev):
    """The split-bf16 kernel (3 exact bf16 pieces per operand, 6 MFMA terms, fp32 accumulate) against fp64, next to the
    native f32-MFMA kernel on the same data: error per unit of sum|a*b| no larger than 1.5x the native kernel's, over wide
    dynamic range (1e-30 .. 3e20 entries, fp16-overflowing values, exact zeros)."""
    L = macx._lib.lib()
    B, N, d = 6, 196, 512
    g = torch.Generator().manual_seed(4)
    kb = torch.randn(B, N, d, generator=g) * torch.exp(4 * torch.randn(B, N, 1, generator=g))
    kb[0, 0, :8] = torch.tensor([1e-30, -3e20, 1.0, -1.0, 65504.0, 1e-8, 3.14159274, 0.0])
    W = torch.randn(d, d, generator=g) / 22
    b = torch.randn(d, generator=g)
    ref = kb.double().reshape(-1, d) @ W.double() + b.double()
    scale = kb.double().abs().reshape(-1, d) @ W.double().abs() + b.double().abs() + 1e-300
    sh = macx._lib.MacxShapes(B=B, S=1, N=N, d=d, p=1, b0=0)
    dp = macx._lib.MacxDropout(keep_memory=1.0, keep_read=1.0, keep_write=1.0, seed=1)
    err = {}
    Wd, kbd, bd = W.to(dev), kb.to(dev), b.to(dev)      # alive across the asynchronous calls
    try:
        for mode in (0, 1):
            L.macx_gemm_mode(mode)
            wp = torch.zeros(2 * d * d, device=dev)
            out = torch.empty(B, N, d, device=dev)
            macx._lib.check(L.macx_pack_weight(_p(Wd), d, d, macx._lib.kb_pack_flags(), _p(wp), None), "pack")
            macx._lib.check(L.macx_kb_project(C.byref(sh), C.byref(dp), 0, _p(kbd), _p(wp), _p(bd), _p(out), None, None), "proj")
            torch.cuda.synchronize()
            e = (out.cpu().double().reshape(-1, d) - ref).abs() / scale
            err[mode] = (float(e.max()), float(e.mean()))
    finally:
        L.macx_gemm_mode(default_gemm_mode())
    assert err[1][0] < 1e-6 and err[1][1] < 5e-8, err
    assert err[1][0] <= 1.5 * err[0][0] and err[1][1] <= 1.5 * err[0][1], err


def test_control_attend(macx, dev):
    L = macx._lib.lib()
    B, S, d = 6, 13, 256
    g = torch.Generator().manual_seed(3)
    cc = torch.randn(B, d, generator=g)
    words = torch.randn(B, S, d, generator=g)
    lengths = torch.tensor([13, 1, 5, 7, 12, 3], dtype=torch.int32)
    w = torch.randn(d, generator=g) / d ** 0.5
    b = torch.tensor([0.3])
    sh = macx._lib.MacxShapes(B=B, S=S, N=1, d=d, p=1, b0=0)
    att = torch.empty(B, S, device=dev)
    ctl = torch.empty(B, d, device=dev)
    args = [t.to(dev) for t in (cc, words, lengths, w, b)]
    macx._lib.check(L.macx_control_attend(C.byref(sh), *[_p(t) for t in args], _p(att), _p(ctl), None), "control_attend")
    torch.cuda.synchronize()
    inter = cc.double().unsqueeze(1) * words.double()
    logits = (inter * w.double()).sum(-1) + 0.3
    ref_att = torch.softmax(mo.Ops.expMask(logits, lengths), dim=-1)
    ref_ctl = (ref_att.unsqueeze(-1) * words.double()).sum(1)
    assert max_abs(att, ref_att) < 1e-6
    assert max_abs(ctl, ref_ctl) < 1e-5
    # padded words get exactly zero attention
    for bi in range(B):
        assert float(att[bi, lengths[bi]:].abs().sum()) == 0.0
    # and the unit's backward (macx_control_attend_bwd) for a caller-chosen d_control, against autograd in fp64
    ccd, wdd, wwd = [t.double().requires_grad_(True) for t in (cc, words, w)]
    bd = torch.tensor([0.3], dtype=torch.float64, requires_grad=True)
    lg = ((ccd.unsqueeze(1) * wdd) * wwd).sum(-1) + bd
    a2 = torch.softmax(mo.Ops.expMask(lg, lengths), dim=-1)
    c2 = (a2.unsqueeze(-1) * wdd).sum(1)
    dctl = torch.randn(B, d, generator=g)
    (c2 * dctl.double()).sum().backward()
    n_ws = L.macx_control_attend_bwd_ws_floats(C.byref(sh))
    ws = torch.empty(n_ws, device=dev)
    dcc, dwords, dw, db = torch.empty(B, d, device=dev), torch.empty(B, S, d, device=dev), torch.empty(d, device=dev), torch.empty(1, device=dev)
    dctl_d = dctl.to(dev)
    macx._lib.check(L.macx_control_attend_bwd(C.byref(sh), _p(dctl_d), _p(args[0]), _p(att), _p(args[1]), _p(args[3]), _p(ws), n_ws,
                                              _p(dcc), _p(dwords), _p(dw), _p(db), None), "control_attend_bwd")
    torch.cuda.synchronize()
    assert rel_err(dcc, ccd.grad) < 2e-5 and rel_err(dwords, wdd.grad) < 2e-5 and rel_err(dw, wwd.grad) < 2e-5
    assert abs(float(db) - float(bd.grad)) < 1e-5          # analytically zero (a bias in front of a softmax)
    assert L.macx_control_attend_bwd(C.byref(sh), _p(dctl_d), _p(args[0]), _p(att), _p(args[1]), _p(args[3]), _p(ws), 8,
                                     _p(dcc), _p(dwords), _p(dw), _p(db), None) == -4


@pytest.mark.parametrize("mode", [1, 0])
@pytest.mark.parametrize("M,Kd,Jd", [(12544, 512, 512), (768, 512, 512), (64, 128, 256), (1001, 256, 128), (49, 128, 128),
                                     (1, 128, 128), (3, 128, 256), (6, 256, 512), (31, 128, 128), (33, 128, 384), (97, 384, 256)])
def test_wgrad(macx, dev, M, Kd, Jd, mode):
    """C = A^T G on the split-bf16 (128x128 and 128x256 tiles, producer/consumer waves) and the native f32 TN kernel, from a
    single reduction row up: ragged last stages, fewer rows than one 32-row stage, row counts below a wave's 8-row group."""
    L = macx._lib.lib()
    L.macx_gemm_mode(mode)
    try:
        _wgrad_case(macx, dev, M, Kd, Jd)
    finally:
        L.macx_gemm_mode(default_gemm_mode())


def _wgrad_case(macx, dev, M, Kd, Jd):
    L = macx._lib.lib()
    g = torch.Generator().manual_seed(4)
    A = torch.randn(M, Kd, generator=g)
    G = torch.randn(M, Jd, generator=g)
    ns = L.macx_wgrad_splits(M, Kd, Jd)
    assert ns >= 1
    out = torch.empty(Kd, Jd, device=dev)
    ws = torch.empty(ns * Kd * Jd, device=dev)
    Ad, Gd = A.to(dev), G.to(dev)
    macx._lib.check(L.macx_wgrad(_p(Ad), Kd, _p(Gd), Jd, M, Kd, Jd, _p(out), _p(ws), None), "wgrad")
    torch.cuda.synchronize()
    ref = A.double().t() @ G.double()
    assert rel_err(out, ref) < 3e-6
    # deterministic: a second call is bit-identical
    out2 = torch.empty_like(out)
    macx._lib.check(L.macx_wgrad(_p(Ad), Kd, _p(Gd), Jd, M, Kd, Jd, _p(out2), _p(ws), None), "wgrad")
    torch.cuda.synchronize()
    assert torch.equal(out, out2)


@pytest.mark.parametrize("B,N,d", [(3, 196, 128), (2, 49, 256), (4, 1, 128), (2, 250, 512)])
def test_kb_attend_unit_forward_and_backward(macx, dev, B, N, d):
    """macx_kb_attend_fwd / _bwd: the attention unit on its own (softmax over the knowledge-base cells + att2Smry,
    ops.py:140-150) against fp64 autograd, <= 1e-5 per the per-unit contract of SURVEY 8b."""
    L = macx._lib.lib()
    g = torch.Generator().manual_seed(B * 1000 + N)
    logits = torch.randn(B, N, generator=g) * 3
    bias = torch.tensor([0.3])
    kb = torch.randn(B, N, d, generator=g)
    dinfo = torch.randn(B, d, generator=g)
    lg, kbd_ = logits.double().requires_grad_(True), kb.double().requires_grad_(True)
    att_ref = torch.softmax(lg + 0.3, dim=-1)
    info_ref = (att_ref.unsqueeze(-1) * kbd_).sum(1)
    (info_ref * dinfo.double()).sum().backward()
    lo, bi, kbt, di = logits.to(dev), bias.to(dev), kb.to(dev), dinfo.to(dev)
    att, info = torch.empty(B, N, device=dev), torch.empty(B, d, device=dev)
    macx._lib.check(L.macx_kb_attend_fwd(B, N, d, _p(lo), _p(bi), _p(kbt), _p(att), _p(info), None), "fwd")
    nws = L.macx_kb_attend_bwd_ws_floats(B, N, d)
    ws = torch.empty(nws, device=dev)
    dl = torch.empty(B, N, device=dev)
    dkb = torch.full((B, N, d), 2.0, device=dev)
    macx._lib.check(L.macx_kb_attend_bwd(B, N, d, _p(att), _p(kbt), _p(di), _p(dl), _p(dkb), 1, _p(ws), nws, None), "bwd")
    torch.cuda.synchronize()
    rel = lambda a, b: float((a.cpu().double() - b).abs().max() / max(float(b.abs().max()), 1e-6))
    assert rel(att, att_ref.detach()) < 1e-5 and rel(info, info_ref.detach()) < 1e-5
    assert rel(dl, lg.grad) < 1e-5
    assert rel(dkb - 2.0, kbd_.grad) < 1e-5
    assert L.macx_kb_attend_bwd(B, N, d, _p(att), _p(kbt), _p(di), _p(dl), None, 0, _p(ws), 1, None) == macx._lib.MACX_ESMALL
