"""Same-process A/B of the stem's convolution kernels (macx_debug_set(9, v): 0 kb_gemm3h_kernel, 1 kb_conv_chain_kernel): forward and
forward + backward of the CLEVR-shape stem at B = 64, rotating over the two, min / median."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import macx

dev = torch.device("cuda:0")
cfg = macx.configs.flag_file_config("args")
stem = macx.Stem(cfg).to(dev)
B = 64
img = torch.relu(torch.randn(B, 196, 1024, device=dev))
dkb = torch.randn(B, 196, 512, device=dev)
L = macx._lib.lib()


def timeit(fn, n=8, warm=2):
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


def fb():
    for t in stem.parameters():
        t.grad = None
    stem(img, train=True, seed=1).backward(dkb)


res = {(v, k): [] for v in (0, 1) for k in ("fwd", "fwd+bwd")}
for rnd in range(int(os.environ.get("ROUNDS", "9"))):
    for v in (0, 1):
        L.macx_debug_set(9, v)
        res[(v, "fwd")].append(timeit(fwd))
        res[(v, "fwd+bwd")].append(timeit(fb))
L.macx_debug_set(9, 1)
for (v, k), t in sorted(res.items()):
    t = sorted(t)
    print("%-22s %-8s min %7.1f us  median %7.1f us" % ("kb_conv_chain_kernel" if v else "kb_gemm3h_kernel", k, t[0], t[len(t) // 2]))
