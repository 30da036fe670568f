#!/usr/bin/env python3
"""A few eager training steps of the metric's shape under a phase-knob mask of macx_opts.tune[MACX_TUNE_PHASE_MASK] -- for a kernel trace per mask
(results are WRONG under a non-zero mask; only kernel durations mean something):
    rocprofv3 --kernel-trace --stats -d out -o r -- python tools/mask_steps.py 1024
masks: 512 sb: skip the per-question fold, 1024 sb / wgrad: skip fragments + MFMAs, 2048 sb / wgrad: skip the in-loop DMA,
       32 / 64 kb_gemm_h2 (dKB): skip the in-loop staging / the fragment reads + MFMAs"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, macx

mask = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
L = macx._lib.lib()
step, params, kbd, bl = bench.make_step(macx, dev, None, 1, 0, 64, 12, 1234)
macx.options.SESSION_TUNE["phase_mask"] = mask      # (every step freezes a new cell: it picks the table up) -- the warm-up steps too,
for i in range(3):                                  # so that a per-kernel mean of the trace is the masked kernel's
    step(i)
torch.cuda.synchronize()
for i in range(steps):
    step(3 + i)
torch.cuda.synchronize()
macx.options.SESSION_TUNE.pop("phase_mask", None)
print("mask", mask, "done")
