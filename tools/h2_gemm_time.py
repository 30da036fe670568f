#!/usr/bin/env python3
"""Times the H2 knowledge-base GEMM alone (B=64, N=196, d=512) under the measurement knobs of macx_debug_set(1, mask).
python tools/h2_gemm_time.py"""
import ctypes as C, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(mask, reps, rt=0):
    os.environ["MACX_H2_DEBUG_REPS"] = "10"        # launches per C call: the host's per-call cost (tens of us on some boxes) drops out
    import macx
    L = macx._lib.lib()
    if rt:
        L.macx_debug_set(2, rt)
    p = lambda t: C.c_void_p(t.data_ptr())
    dev = torch.device("cuda:0")
    B, N, K = int(os.environ.get('MACX_TIME_B', '64')), 196, 512
    g = torch.Generator().manual_seed(1)
    A = torch.randn(B, N, K, generator=g).to(dev)
    W = (torch.randn(K, K, generator=g) / 22).to(dev)
    b = torch.randn(K, generator=g).to(dev)
    hf = L.macx_h2_floats(B * N, K)
    NBUF = 4
    hin = [torch.zeros(hf, device=dev) for _ in range(NBUF)]
    hout = [torch.zeros(hf, device=dev) for _ in range(NBUF)]
    wh = torch.zeros(K * K + 64, device=dev)
    macx._lib.check(L.macx_h2_pack_weight(p(W), K, K, 0, p(wh), None), "pack")
    for i in range(NBUF):
        macx._lib.check(L.macx_h2_from_f32(p(A), B, N, K, p(hin[i]), None), "from")
    L.macx_debug_set(1, mask)
    for i in range(8):
        macx._lib.check(L.macx_h2_gemm_planes(p(hin[i % NBUF]), B, N, K, p(wh), K, p(b), 0, p(hout[i % NBUF]), None), "g")
    torch.cuda.synchronize()
    best = 1e9
    for it in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            L.macx_h2_gemm_planes(p(hin[i % NBUF]), B, N, K, p(wh), K, p(b), 0, p(hout[i % NBUF]), None)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3 / 10)
    ts = [0.0, best * reps / 1e3]
    print("mask %3d rt %2d: %.2f us per launch" % (mask, rt, (ts[1] - ts[0]) / reps * 1e3), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2:
        child(int(sys.argv[1]), 50, int(sys.argv[2]))
    else:
        masks = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else (0, 128, 1, 1 | 128, 64, 1 | 64, 1 | 2 | 32, 1 | 64 | 2 | 32)
        for rt in (0, 7):
            for mask in masks:
                subprocess.run([sys.executable, __file__, str(mask), str(rt)])
