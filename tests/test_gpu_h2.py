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
