#!/usr/bin/env python3
"""Where the filler workgroups of chain_fwd's launch (ChainPreP) spend their time, and how long the tiles wait for y: a library built
with MACX_BUILD_DEFINES=MACX_FILL_PROF leaves timestamps (s_memtime-class cycle counter, as reported by __builtin_readcyclecounter)
of step 5's filler 0 and first tile in the run's sync words.   MACX_BUILD_DEFINES=MACX_FILL_PROF python tools/fill_prof.py [pre_fill]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench, macx

dev = torch.device("cuda:0")
v = int(sys.argv[1]) if len(sys.argv) > 1 else 1
macx.options.SESSION_TUNE["pre_fill"] = v
D, S, N, B, P = 512, 50, 196, 64, 12
cfg = macx.configs.flag_file_config("args", netLength=P, memDim=D, ctrlDim=D, attDim=D)
vq, words, lengths, kb = macx.configs.synthetic_inputs(B, S, N, D, seed=1234)
params = macx.MACCellParams(cfg, P, generator=torch.Generator().manual_seed(1234)).to(dev)
vqd, wd, kbd = [t.to(dev).requires_grad_(True) for t in (vq, words, kb)]
ld = lengths.to(dev)
from importlib import import_module
cellmod = import_module(macx.MACCell.__module__)
RunCls = cellmod._Run
for i in range(4):
    cell = macx.MACCell(vecQuestions=vqd, questionWords=wd, questionCntxWords=wd, questionLengths=ld, knowledgeBase=kbd,
                        memoryDropout=cfg.memoryDropout, readDropout=cfg.readDropout, writeDropout=cfg.writeDropout, batchSize=B,
                        train=True, config=cfg, params=params, seed=1234 + i, b0=0)
    state = cell.run()
    (state.memory.sum()).backward()
    torch.cuda.synchronize()
import gc
saved = cell._run.saved if cell._run is not None else max((o for o in gc.get_objects() if isinstance(o, RunCls)), key=lambda o: o.saved.numel()).saved
w = saved.view(torch.int32)[-576:].cpu().numpy().astype("uint32")
t = w[32:48].astype("int64")
print("pre_fill", v, "raw", t.tolist())
f0, t0 = t[0], t[8]
names = {0: "filler start", 1: "filler: write tiles done", 2: "filler: y signalled", 3: "filler: stage-0 jobs done",
         8: "tile start", 9: "tile: second product done (needs y)", 10: "tile: y there", 11: "tile: last stage done"}
base = min(f0, t0)
for k in sorted(names):
    print("%-40s %10d ticks  (+%d)" % (names[k], t[k], (t[k] - base) & 0xFFFFFFFF))
d = w[48:63].astype("int64")
print("chain_bwd filler 0 of step 5 (dKB jobs): start | per job: tile in LDS, K loop done, staged, row pass issued -- ticks since start")
print([int((v - d[0]) & 0xFFFFFFFF) if v else 0 for v in d])
b = w[48 + 336:48 + 336 + 8].astype("int64")
print("chain_bwd tile 0 of step 5: start, B0 done, B1 product, B1 epilogue, B2 product 1, dy + y scaling, B2 product 2, dX emitted -- ticks since start")
print([int((v - b[0]) & 0xFFFFFFFF) if v else 0 for v in b])

