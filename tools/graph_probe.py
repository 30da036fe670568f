#!/usr/bin/env python3
"""Feasibility probe: capture one cell fwd+bwd (fixed dropout seed) into a HIP graph through torch.cuda.CUDAGraph and
compare replay time with eager launches.  python tools/graph_probe.py [B ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import macx

def main():
    dev = torch.device("cuda:0")
    S, N, D, P = 50, 196, 512, 12
    cfg = macx.configs.flag_file_config("args", netLength=P, memDim=D, ctrlDim=D, attDim=D)
    for B in [int(a) for a in sys.argv[1:]] or [64, 8]:
        vq, words, lengths, kb = macx.configs.synthetic_inputs(B, S, N, D, seed=1)
        params = macx.MACCellParams(cfg, P, generator=torch.Generator().manual_seed(1)).to(dev)
        vqd, wd, kbd = [t.to(dev).requires_grad_(True) for t in (vq, words, kb)]
        ld = lengths.to(dev)
        gmem = (torch.randn(B, D) / B).to(dev)
        def step(seed=7):
            cell = macx.MACCell(vecQuestions=vqd, questionWords=wd, questionCntxWords=wd, questionLengths=ld, knowledgeBase=kbd,
                                memoryDropout=cfg.memoryDropout, readDropout=cfg.readDropout, writeDropout=cfg.writeDropout,
                                batchSize=B, train=True, config=cfg, params=params, seed=seed, b0=0)
            state = cell.run()
            for t in [vqd, wd, kbd] + params.tensors():
                t.grad = None
            torch.autograd.backward([state.memory], [gmem])
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 10 * 1e3
        ref = params.memKbProj_W.grad.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        gref = params.memKbProj_W.grad
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 10 * 1e3
        print("B=%d: eager %.3f ms/step, graph replay %.3f ms/step, grads equal: %s" % (B, eager, graph, torch.equal(ref, gref)))

if __name__ == "__main__":
    main()
