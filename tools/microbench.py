#!/usr/bin/env python3
"""GPU micro-benchmarks of the individual C-ABI entry points (HIP events on torch's current stream).
    python tools/microbench.py [--nw 4|8]"""
import argparse, ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import macx

def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--B", type=int, default=64); ap.add_argument("--stem", action="store_true"); ap.add_argument("--encoder", action="store_true"); args = ap.parse_args()
    L = macx._lib.lib(); dev = torch.device("cuda:0")
    B, N, d = args.B, 196, 512
    p = lambda t: C.c_void_p(t.data_ptr())
    kb = torch.randn(B, N, d, device=dev); W = torch.randn(d, d, device=dev) / 22; b = torch.randn(d, device=dev)
    wp = torch.empty(2 * d * d, device=dev); out = torch.empty(B, N, d, device=dev); bits = torch.empty(B * N * d + B * N * d // 32, device=dev)
    L.macx_pack_weight(p(W), d, d, macx._lib.kb_pack_flags(), p(wp), None)
    sh = macx._lib.MacxShapes(B=B, S=50, N=N, d=d, p=12, b0=0)
    flops = 2.0 * B * N * d * d
    for keep in (1.0, 0.85):
        dp = macx._lib.MacxDropout(keep_memory=1.0, keep_read=keep, keep_write=1.0, seed=1)
        us = timeit(lambda: L.macx_kb_project(C.byref(sh), C.byref(dp), 0, p(kb), p(wp), p(b), p(out), p(bits), None))
        print("kb_project keep=%.2f: %8.1f us  %6.1f TF" % (keep, us, flops / us / 1e6))
    M = B * N
    A = torch.randn(M, d, device=dev); G = torch.randn(M, d, device=dev)
    ns = L.macx_wgrad_splits(M, d, d); ws = torch.empty(ns * d * d, device=dev); o = torch.empty(d, d, device=dev)
    us = timeit(lambda: L.macx_wgrad(p(A), d, p(G), d, M, d, d, p(o), p(ws), None))
    print("wgrad M=%d (%d splits, incl. slab reduce): %8.1f us  %6.1f TF" % (M, ns, us, flops / us / 1e6))
    Mb = 12 * M
    A2 = torch.randn(Mb, d, device=dev); G2 = torch.randn(Mb, d, device=dev)
    ns2 = L.macx_wgrad_splits(Mb, d, d); ws2 = torch.empty(ns2 * d * d, device=dev)
    us = timeit(lambda: L.macx_wgrad(p(A2), d, p(G2), d, Mb, d, d, p(o), p(ws2), None), n=5, warm=2)
    print("wgrad M=%d (%d splits): %8.1f us  %6.1f TF" % (Mb, ns2, us, 12 * flops / us / 1e6))
    x = torch.randn(64, 512, device=dev); o2 = torch.empty(64, 512, device=dev)
    us = timeit(lambda: L.macx_linear(p(x), 512, None, 0, 64, p(wp), p(b), 0.0, 512, 0, p(o2), None))
    print("linear 64x512x512: %8.1f us" % us)

if __name__ == "__main__" and "--stem" not in sys.argv:
    main()


def stem_bench():
    import time
    dev = torch.device("cuda:0")
    cfg = macx.configs.flag_file_config("args")
    stem = macx.Stem(cfg).to(dev)
    B = 64
    img = torch.relu(torch.randn(B, 196, 1024, device=dev))
    dkb = torch.randn(B, 196, 512, device=dev)
    def fwd():
        with torch.no_grad():
            return stem(img, train=True, seed=1)
    us_f = timeit(fwd, n=10, warm=3)
    def fb():
        kb = stem(img, train=True, seed=1)
        kb.backward(dkb)
    us_fb = timeit(fb, n=10, warm=3)
    fl_f = 2.0 * B * 196 * 9 * (1024 * 512 + 512 * 512)
    fl_b = 2.0 * B * 196 * 9 * (1024 * 512 + 2 * 512 * 512)
    print("stem fwd B=64: %8.1f us  %6.1f TF ; fwd+bwd: %8.1f us  %6.1f TF" % (us_f, fl_f / us_f / 1e6, us_fb, (fl_f + fl_b) / us_fb / 1e6))


def encoder_bench():
    dev = torch.device("cuda:0")
    cfg = macx.configs.flag_file_config("args")
    enc = macx.QuestionEncoder(cfg, vocab=90).to(dev)
    for B, S in ((64, 50), (64, 30), (8, 50)):
        g = torch.Generator().manual_seed(1)
        lengths = torch.randint(3, S + 1, (B,), generator=g, dtype=torch.int32)
        q = torch.randint(1, 91, (B, S), generator=g, dtype=torch.int32) * (torch.arange(S).unsqueeze(0) < lengths.unsqueeze(1)).to(torch.int32)
        q, lengths = q.to(dev), lengths.to(dev)
        dW = torch.randn(B, S, 512, device=dev); dQ = torch.randn(B, 512, device=dev)
        def fwd():
            with torch.no_grad():
                return enc(q, lengths, train=True, seed=1, check_ids=False)
        def fb():
            w, v = enc(q, lengths, train=True, seed=1, check_ids=False)
            torch.autograd.backward([w, v], [dW, dQ])
        print("encoder B=%d S=%d: fwd %8.1f us ; fwd+bwd %8.1f us" % (B, S, timeit(fwd, n=10, warm=3), timeit(fb, n=10, warm=3)))


if __name__ == "__main__" and "--stem" in sys.argv:
    stem_bench()
if __name__ == "__main__" and "--encoder" in sys.argv:
    encoder_bench()
