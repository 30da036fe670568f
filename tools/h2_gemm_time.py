#!/usr/bin/env python3
"""Times the H2 knowledge-base GEMM alone (B=64, N=196, d=512) under the measurement knobs of macx_debug_set(1, mask).
python tools/h2_gemm_time.py"""
import ctypes as C, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(mask, reps):
    import macx
    L = macx._lib.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    dev = torch.device("cuda:0")
    B, N, K = 64, 196, 512
    g = torch.Generator().manual_seed(1)
    A = torch.randn(B, N, K, generator=g).to(dev)
    W = (torch.randn(K, K, generator=g) / 22).to(dev)
    b = torch.randn(K, generator=g).to(dev)
    n = 2 * L.macx_h2_floats(B * N, K) + K * K + 64
    ws = torch.zeros(n, device=dev)
    out = torch.zeros(B * N, K, device=dev)
    L.macx_debug_set(1, mask)
    ts = []
    for r in (1, reps + 1):
        os.environ["MACX_H2_DEBUG_REPS"] = str(r)
        for it in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            macx._lib.check(L.macx_h2_gemm(p(A), B, N, K, p(W), K, p(b), 0, p(out), p(ws), n, None), "g")
            e1.record()
            torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("mask %3d: %.2f us per launch" % (mask, (ts[1] - ts[0]) / reps * 1e3), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]), 50)
    else:
        for mask in (0, 1, 2, 32, 2 | 32, 64, 64 | 2 | 32, 1 | 64 | 2 | 32, 1 | 64, 1 | 2 | 32):
            subprocess.run([sys.executable, __file__, str(mask)])
