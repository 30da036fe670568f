#!/usr/bin/env python3
"""Split-bf16 (3 planes, 6 MFMA terms) knowledge-base GEMM vs the native f32-MFMA kernel: time and error against fp64.
    python tools/gemm6_probe.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import macx

def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def main():
    L = macx._lib.lib(); dev = torch.device("cuda:0")
    p = lambda t: C.c_void_p(t.data_ptr())
    for B, N, d in ((64, 196, 512), (8, 196, 512), (3, 49, 256)):
        g = torch.Generator().manual_seed(1)
        kb = torch.randn(B, N, d, generator=g).to(dev); W = (torch.randn(d, d, generator=g) / 22).to(dev); b = torch.randn(d, generator=g).to(dev)
        kb[0, 0, :8] = torch.tensor([1e-30, -3e20, 1.0, -1.0, 65504.0, 1e-8, 3.14159274, 0.0])
        ref = (kb.double().reshape(-1, d) @ W.double() + b.double()).reshape(B, N, d)
        scale = (kb.double().abs().reshape(-1, d) @ W.double().abs()).reshape(B, N, d) + 1e-300
        sh = macx._lib.MacxShapes(B=B, S=50, N=N, d=d, p=12, b0=0)
        dp = macx._lib.MacxDropout(keep_memory=1.0, keep_read=1.0, keep_write=1.0, seed=1)
        flops = 2.0 * B * N * d * d
        for mode in (0, 1):
            L.macx_gemm_mode(mode)
            wp = torch.zeros(2 * d * d, device=dev); out = torch.empty(B, N, d, device=dev)
            macx._lib.check(L.macx_pack_weight(p(W), d, d, 2 * mode, p(wp), None), "pack")
            macx._lib.check(L.macx_kb_project(C.byref(sh), C.byref(dp), 0, p(kb), p(wp), p(b), p(out), None, None), "proj")
            torch.cuda.synchronize()
            err = ((out.double() - ref).abs() / scale)
            us = timeit(lambda: L.macx_kb_project(C.byref(sh), C.byref(dp), 0, p(kb), p(wp), p(b), p(out), None, None))
            line = "B=%d N=%d d=%d %s: %7.1f us %6.1f TF(f32-equiv)  err/sum|ab| max %.2e mean %.2e" % (
                B, N, d, "split-bf16x6" if mode else "native f32  ", us, flops / us / 1e6, float(err.max()), float(err.mean()))
            if B == 64:
                for dbg in (1, 2, 3, 32, 64, 96):     # 1 no epilogue, 2 no staging, 32 / 64 A / B re-read slice 0 (cache-hot)
                    L.macx_debug_set(1, dbg)
                    line += " | dbg%d %.1f" % (dbg, timeit(lambda: L.macx_kb_project(C.byref(sh), C.byref(dp), 0, p(kb), p(wp), p(b), p(out), None, None)))
                L.macx_debug_set(1, 0)
            print(line)
        L.macx_gemm_mode(1)

if __name__ == "__main__":
    main()
