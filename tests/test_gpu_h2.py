"""-m gpu: the H2 tensor format (mac-network_amd/csrc/macx_h2.hip.h) and the GEMM family on it in isolation: conversion
round trip, and the three-term fp16 product against fp64 next to the native f32-MFMA kernel on the same data."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import default_gemm_mode, rel_err

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("B,N,C_", [(3, 196, 512), (2, 49, 128), (1, 1, 128), (2, 209, 256), (1, 450, 128)])
def test_h2_round_trip(macx, dev, B, N, C_):
    """fp32 -> H2 -> fp32: every element within 2^-23 of its (row, 128-column block) maximum, elements within 2^-16 of that
    maximum to fp32's own rounding unit; zeros, tiny rows, huge rows and non-finite values survive."""
    L = macx._lib.lib()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, N, C_, generator=g) * torch.exp(6 * torch.randn(B, N, 1, generator=g))
    x[0, 0, :6] = torch.tensor([0.0, 1e-30, -3e20, 65504.0, -1e-8, 1.0])
    if N > 2:
        x[0, 1, :] = 0.0                      # an all-zero row
        x[0, 2, 5] = float("inf")
    xd = x.to(dev)
    n = L.macx_h2_floats(B * N, C_)
    assert n > 0
    h2 = torch.empty(n, device=dev)
    out = torch.empty(B * N, C_, device=dev)
    macx._lib.check(L.macx_h2_from_f32(_p(xd), B, N, C_, _p(h2), None), "from")
    macx._lib.check(L.macx_h2_to_f32(_p(h2), B * N, C_, _p(out), None), "to")
    torch.cuda.synchronize()
    y = out.cpu().reshape(B, N, C_)
    xb = x.reshape(B, N, C_ // 128, 128).double()
    yb = y.reshape(B, N, C_ // 128, 128).double()
    fin = torch.isfinite(xb).all(dim=-1, keepdim=True)
    blockmax = xb.abs().amax(dim=-1, keepdim=True)
    err = (yb - xb).abs()
    ok = fin.expand_as(xb)
    assert float((err[ok] / blockmax.expand_as(xb)[ok].clamp_min(1e-300)).max()) < 2.0 ** -23
    big = ok & (xb.abs() >= blockmax * 2.0 ** -15) & (xb != 0)
    assert float((err[big] / xb.abs()[big]).max()) <= 2.0 ** -23
    assert torch.equal(y[0, 1], torch.zeros(C_)) if N > 2 else True
    if N > 2:
        assert not torch.isfinite(y[0, 2, 5])


@pytest.mark.parametrize("B,N,K,n_out", [(6, 196, 512, 512), (3, 49, 128, 256), (2, 1, 128, 128), (2, 209, 256, 128),
                                          (1, 14, 384, 128), (5, 100, 256, 512), (1, 250, 128, 128)])
def test_h2_gemm_error_is_fp32_class(macx, dev, B, N, K, n_out):
    """act(A W + b) on the H2 kernels (2 fp16 planes per operand, 3 MFMA terms, fp32 accumulate, per-row-block exponents)
    against fp64: error per unit of sum |a w| no larger than 1.5x the native f32-MFMA kernel's on the same data, over a wide
    dynamic range across rows (e^(+-4) row scales, tiny and huge entries)."""
    L = macx._lib.lib()
    g = torch.Generator().manual_seed(4)
    A = torch.randn(B, N, K, generator=g) * torch.exp(4 * torch.randn(B, N, 1, generator=g))
    A[0, 0, :8] = torch.tensor([1e-30, -3e20, 1.0, -1.0, 65504.0, 1e-8, 3.14159274, 0.0])
    W = torch.randn(K, n_out, generator=g) / 22
    b = torch.randn(n_out, generator=g)
    ref = A.double().reshape(-1, K) @ W.double() + b.double()
    scale = A.double().abs().reshape(-1, K) @ W.double().abs() + b.double().abs() + 1e-300
    n = L.macx_h2_floats(B * N, K) + L.macx_h2_floats(B * N, n_out) + K * n_out + 64
    ws = torch.empty(n, device=dev)
    out = torch.empty(B * N, n_out, device=dev)
    Ad, Wd, bd = A.to(dev), W.to(dev), b.to(dev)        # keep the device copies alive across the asynchronous call
    macx._lib.check(L.macx_h2_gemm(_p(Ad), B, N, K, _p(Wd), n_out, _p(bd), 0, _p(out), _p(ws), n, None), "h2_gemm")
    torch.cuda.synchronize()
    e = (out.cpu().double() - ref).abs() / scale
    emax, emean = float(e.max()), float(e.mean())
    assert emax < 1e-6 and emean < 5e-8, (emax, emean)
    if K == n_out and B * N >= 100:      # the native kernel (macx_kb_project) is square-only; a ratio of maxima needs a sample
        try:
            L.macx_gemm_mode(0)
            sh = macx._lib.MacxShapes(B=B, S=1, N=N, d=K, p=1, b0=0)
            dp = macx._lib.MacxDropout(keep_memory=1.0, keep_read=1.0, keep_write=1.0, seed=1)
            wp = torch.zeros(2 * K * K, device=dev)
            o2 = torch.empty(B, N, K, device=dev)
            macx._lib.check(L.macx_pack_weight(_p(Wd), K, K, macx._lib.kb_pack_flags(), _p(wp), None), "pack")
            macx._lib.check(L.macx_kb_project(C.byref(sh), C.byref(dp), 0, _p(Ad), _p(wp), _p(bd), _p(o2), None, None), "proj")
            torch.cuda.synchronize()
            e0 = (o2.cpu().double().reshape(-1, K) - ref).abs() / scale
            assert emax <= 1.5 * float(e0.max()) and emean <= 1.5 * float(e0.mean()), (emax, emean, float(e0.max()), float(e0.mean()))
        finally:
            L.macx_gemm_mode(default_gemm_mode())


def test_h2_gemm_activation_and_determinism(macx, dev):
    L = macx._lib.lib()
    B, N, K = 3, 196, 256
    g = torch.Generator().manual_seed(9)
    A = torch.randn(B, N, K, generator=g)
    W = torch.randn(K, K, generator=g) / 16
    b = torch.randn(K, generator=g)
    n = 2 * L.macx_h2_floats(B * N, K) + K * K + 64
    ws = torch.empty(n, device=dev)
    outs = []
    Ad, Wd, bd = A.to(dev), W.to(dev), b.to(dev)
    for _ in range(2):
        out = torch.empty(B * N, K, device=dev)
        macx._lib.check(L.macx_h2_gemm(_p(Ad), B, N, K, _p(Wd), K, _p(bd), macx._lib.ACT["ELU"], _p(out), _p(ws), n, None), "h2_gemm")
        torch.cuda.synchronize()
        outs.append(out.cpu())
    ref = torch.nn.functional.elu(A.double().reshape(-1, K) @ W.double() + b.double())
    assert rel_err(outs[0], ref) < 2e-6
    assert torch.equal(outs[0], outs[1])


def _wide(shape, g, span=12.0):
    """N(0,1) entries times 2^u, u uniform in [-span, span], independently per ELEMENT: every exponent block (a row, a matrix, a
    whole tensor) holds entries 2^(2 span) apart"""
    return torch.randn(shape, generator=g) * torch.exp2((torch.rand(shape, generator=g) * 2 - 1) * span)


@pytest.mark.parametrize("B,N,d", [(3, 196, 512), (2, 49, 256)])
def test_chain_kernel_products_on_wide_dynamic_range(macx, dev, B, N, d):
    """The chain kernels' coarser exponent granularities -- ONE exponent per ROW of d columns for the LDS-resident activation
    tile, ONE per MATRIX for the packed weights (pack format 3) -- on data whose entries span 2^+-12 inside every such block
    (they had only ever seen N(0,1)-scale data).  Stage 1 (X = KB Wx + bx) and stage 3 (I2 = H1 W2 + b2, with
    the kept H1 as the operand the kernel multiplied) against fp64: error per unit of the row's largest sum |a w| + |b| below 1e-6
    and at most 1.5x the native v_mfma_f32_16x16x4_f32 kernel's on the same data (+ 2^-22 for the H2 rounding of the stored result)."""
    L = macx._lib.lib()
    p = 1
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
    g = torch.Generator().manual_seed(31)
    vq, words, lengths, _ = macx.configs.synthetic_inputs(B, 5, N, d, seed=3)
    kb = _wide((B, N, d), g)
    params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(2)).to(dev)
    with torch.no_grad():
        params.projX_W.copy_(_wide((d, d), g) / 22)
        params.memKbProj2_W.copy_(_wide((d, d), g) / 22)
        params.projX_b.copy_(torch.randn(d, generator=g))
        params.memKbProj2_b.copy_(torch.randn(d, generator=g))
    vqd, wd, kbd, ld = vq.to(dev), words.to(dev), kb.to(dev), lengths.to(dev)
    cell = macx.MACCell(vqd, wd, wd, ld, kbd, 1.0, 1.0, 1.0, B, True, config=cfg, params=params, seed=1, gemm="h2")
    run = macx.cell._Run(cell, True)
    run.begin()
    run.step(0)
    outs = []
    for which in (1, 2, 3):
        o = torch.empty(B * N, d, device=dev)
        macx._lib.check(L.macx_saved_activation(C.byref(run.opts), C.byref(run.shapes), which, 0, _p(run.saved), run.saved_floats, _p(o),
                                                run.stream), "macx_saved_activation")
        outs.append(o)
    torch.cuda.synchronize()
    X, H1, I2 = [o.cpu().double() for o in outs]
    Wx, bx = params.projX_W.detach().cpu().double(), params.projX_b.detach().cpu().double()
    W2, b2 = params.memKbProj2_W.detach().cpu().double(), params.memKbProj2_b.detach().cpu().double()
    A1 = kb.double().reshape(-1, d)
    cases = [("X", X, A1, Wx, bx, kb.reshape(-1, d), params.projX_W, params.projX_b),
             ("I2", I2, H1, W2, b2, outs[1].cpu(), params.memKbProj2_W, params.memKbProj2_b)]
    for name, got, A, W, b, A32, Wt, bt in cases:
        ref = A @ W + b
        scale = (A.abs() @ W.abs() + b.abs()).amax(dim=1, keepdim=True) + 1e-300
        e = ((got - ref).abs() / scale)
        emax, emean = float(e.max()), float(e.mean())
        assert emax < 1e-6, (name, emax)
        # the native f32-MFMA kernel on the same operands
        try:
            L.macx_gemm_mode(0)
            sh = macx._lib.MacxShapes(B=B, S=1, N=N, d=d, p=1, b0=0)
            dp = macx._lib.MacxDropout(keep_memory=1.0, keep_read=1.0, keep_write=1.0, seed=1)
            wp = torch.zeros(2 * d * d, device=dev)
            o2 = torch.empty(B, N, d, device=dev)
            Ad = A32.to(dev).contiguous()
            macx._lib.check(L.macx_pack_weight(_p(Wt.detach()), d, d, macx._lib.kb_pack_flags(), _p(wp), None), "pack")
            macx._lib.check(L.macx_kb_project(C.byref(sh), C.byref(dp), 0, _p(Ad), _p(wp), _p(bt.detach()), _p(o2), None, None), "proj")
            torch.cuda.synchronize()
            e0 = (o2.cpu().double().reshape(-1, d) - ref).abs() / scale
        finally:
            L.macx_gemm_mode(default_gemm_mode())
        assert emax <= 1.5 * float(e0.max()) + 2.0 ** -22 and emean <= 1.5 * float(e0.mean()) + 2.0 ** -24, (name, emax, emean, float(e0.max()), float(e0.mean()))
