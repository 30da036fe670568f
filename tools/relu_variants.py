#!/usr/bin/env python3
"""The metric's step (B=64, p=12, fwd+bwd, train-mode dropout) under the activation variants that select OTHER chain_bwd kernels than the
published configurations' ELU: --relu STD (plain ReLU for readMemAct / readCtrlAct), readCtrlAct = TANH, readCtrlAct = NON.
    python tools/relu_variants.py            (under rocprofv3 --kernel-trace --stats for the per-kernel rows)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import macx

dev = torch.device("cuda:0")
B, S, N, D, P = 64, 50, 196, 512, 12
vq, words, lengths, kb = macx.configs.synthetic_inputs(B, S, N, D, seed=1234)
vqd, wd, kbd = [t.to(dev).requires_grad_(True) for t in (vq, words, kb)]
ld = lengths.to(dev)
gm = (torch.randn(B, D, generator=torch.Generator().manual_seed(1)) / B).to(dev)
for name, over in (("relu=ELU (args.txt)", {}), ("relu=STD", {"relu": "STD"}), ("readCtrlAct=TANH", {"readCtrlAct": "TANH"}),
                   ("readCtrlAct=NON", {"readCtrlAct": "NON"})):
    cfg = macx.configs.flag_file_config("args", netLength=P, memDim=D, ctrlDim=D, attDim=D, **over)
    params = macx.MACCellParams(cfg, P, generator=torch.Generator().manual_seed(1234)).to(dev)

    def step(i):
        cell = macx.MACCell(vecQuestions=vqd, questionWords=wd, questionCntxWords=wd, questionLengths=ld, knowledgeBase=kbd,
                            memoryDropout=cfg.memoryDropout, readDropout=cfg.readDropout, writeDropout=cfg.writeDropout, batchSize=B,
                            train=True, config=cfg, params=params, seed=1234 + i)
        st = cell.run()
        for t in [vqd, wd, kbd] + params.tensors():
            t.grad = None
        torch.autograd.backward([st.memory], [gm])

    for i in range(4):
        step(i)
    best = None
    for blk in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            step(4 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10 * 1e3
        best = dt if best is None else min(best, dt)
    print("%-22s %.3f ms per step (eager, best of 3 blocks of 10)" % (name, best), flush=True)
