#!/usr/bin/env python3
"""ms per bench step under a list of macx_opts.tune[MACX_TUNE_PHASE_MASK] masks, all in ONE process (a fresh gpurun box can take a minute
per python start): differences between masks are stage times.  Results under a non-zero mask are wrong; only time counts.
    python tools/mask_sweep.py 0 131072 262144 [--chain 0] [--steps 20] [--batch 64]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, macx

ap = argparse.ArgumentParser()
ap.add_argument("masks", nargs="+", type=int)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--p", type=int, default=12)
ap.add_argument("--chain", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
L = macx._lib.lib()
macx.options.SESSION_TUNE["chain"] = a.chain
step, params, kbd, bl = bench.make_step(macx, dev, None, 1, 0, a.batch, a.p, 1234)
for i in range(8):
    step(i)
torch.cuda.synchronize()
for m in a.masks:
    macx.options.SESSION_TUNE["phase_mask"] = m
    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    torch.cuda.synchronize()
    print("mask %8d  %.3f ms/step" % (m, (time.perf_counter() - t0) / a.steps * 1e3), flush=True)
macx.options.SESSION_TUNE.pop("phase_mask", None)
