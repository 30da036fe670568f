#!/usr/bin/env python3
"""Same-process A/B of one key of the per-call tuning table (macx_opts.tune, _lib.TUNE; default: chain_kv, the K-loop variant of the
chain kernels) on the bench step: for every value, (1) one training step compared with the kv = 0 step on the same seed (final memory, dKB, three weight
gradients: the variants reorder fp32 sums, so the comparison is relative to the tensor's largest entry), (2) ms per bench
step, (3) the forward chain kernel alone (macx_read_chain_time).
    python tools/kv_sweep.py --key wgrad_pipe 3 0 1 2 [--steps 20] [--batch 64] [--rounds 2]
The first value is the parity reference.  (Every step freezes a new cell, which picks options.SESSION_TUNE up.)"""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, macx

ap = argparse.ArgumentParser()
ap.add_argument("kvs", nargs="+", type=int)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--p", type=int, default=12)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--key", default="chain_kv", help="name of the tuning key (macx._lib.TUNE)")
ap.add_argument("--zero", action="store_true", help="all-zero knowledge base and weights: the matrix pipe's data-dependent power "
                "draw (tools/probes/mfma_probe.hip) taken out of the kernel times -- timing only")
a = ap.parse_args()
dev = torch.device("cuda:0")
L = macx._lib.lib()
step, params, kbd, bl = bench.make_step(macx, dev, None, 1, 0, a.batch, a.p, 1234)


def snapshot():
    step(0)
    torch.cuda.synchronize()
    names = ["dKB"] + [n for n in ("memKbProj_W", "projX_W", "memKbProj2_W", "newMemory_W", "projY_W") if hasattr(params, n)]
    out = {"dKB": kbd.grad.detach().clone()}
    for n in names[1:]:
        out[n] = getattr(params, n).grad.detach().clone()
    return out


macx.options.SESSION_TUNE[a.key] = a.kvs[0]
for i in range(4):
    step(i)
ref = snapshot()
for k, v in ref.items():
    print("ref %-14s max %.3e" % (k, float(v.abs().max())), flush=True)
for kv in a.kvs:
    macx.options.SESSION_TUNE[a.key] = kv
    got = snapshot()
    worst = max(float((got[k] - ref[k]).abs().max() / ref[k].abs().max()) for k in ref)
    bad = any(not torch.isfinite(v).all() for v in got.values())
    print("kv %2d  parity vs the first value: worst rel-to-max %.2e%s" % (kv, worst, "  NON-FINITE" if bad else ""), flush=True)
if a.zero:
    with torch.no_grad():
        kbd.zero_()
        for t in params.tensors():
            t.zero_()
for r in range(a.rounds):
    for kv in a.kvs:
        macx.options.SESSION_TUNE[a.key] = kv
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(i)
        torch.cuda.synchronize()
        print("round %d kv %2d  %.3f ms/step" % (r, kv, (time.perf_counter() - t0) / a.steps * 1e3), flush=True)
macx.options.SESSION_TUNE.pop(a.key, None)
