"""Stem forward (two 3x3 convolutions as implicit GEMMs, kb_gemm3h) under the phase knobs of macx_debug_set(1, mask):
2 the products alone (no loads, no split), 16 no forced interleave of split and products, 32 every K slice re-reads slice 0 of A (cache-hot), 64 ... of the weights.  Timing only
(results are wrong under a non-zero mask)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import macx

dev = torch.device("cuda:0")
cfg = macx.configs.flag_file_config("args")
stem = macx.Stem(cfg).to(dev)
B = 64
img = torch.relu(torch.randn(B, 196, 1024, device=dev))
L = macx._lib.lib()


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def fwd():
    with torch.no_grad():
        return stem(img, train=True, seed=1)


fl = 2.0 * B * 196 * 9 * (1024 * 512 + 512 * 512)
masks = [int(m) for m in sys.argv[1:]] or [0, 32, 64, 96]
res = {m: [] for m in masks}
for rnd in range(6):                      # rotate: boxes drift by several per cent within a run (clocks, power state)
    for m in masks:
        L.macx_debug_set(1, m)
        res[m].append(timeit(fwd, n=8, warm=2))
L.macx_debug_set(1, 0)
for m in masks:
    v = sorted(res[m])
    print("mask %3d: stem fwd min %7.1f us  median %7.1f us  (%.2f PF executed on 3 fp16 terms at the minimum)" % (m, v[0], v[len(v) // 2], 3 * fl / v[0] / 1e9))
