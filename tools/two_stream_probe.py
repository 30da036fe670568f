#!/usr/bin/env python3
"""Two towers on ONE GPU: the batch of 64 questions as two half-batches on two HIP streams (independent chains of kernels
that run concurrently and out of phase), gradients summed -- against the single-chain step.  python tools/two_stream_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import macx

def main():
    dev = torch.device("cuda:0")
    B, S, N, D, P = 64, 50, 196, 512, 12
    cfg = macx.configs.flag_file_config("args", netLength=P)
    vq, words, lengths, kb = macx.configs.synthetic_inputs(B, S, N, D)
    params = macx.MACCellParams(cfg, P, generator=torch.Generator().manual_seed(1)).to(dev)
    gmem = (torch.randn(B, D, generator=torch.Generator().manual_seed(1)) / B).to(dev)
    vqd, wd, kbd, ld = [t.to(dev) for t in (vq, words, kb, lengths)]

    def chain(lo, hi, seed, stream):
        with torch.cuda.stream(stream):
            a, w, k = [t[lo:hi].detach().requires_grad_(True) for t in (vqd, wd, kbd)]
            cell = macx.MACCell(vecQuestions=a, questionWords=w, questionCntxWords=w, questionLengths=ld[lo:hi], knowledgeBase=k,
                                memoryDropout=cfg.memoryDropout, readDropout=cfg.readDropout, writeDropout=cfg.writeDropout,
                                batchSize=hi - lo, train=True, config=cfg, params=params, seed=seed, b0=lo)
            st = cell.run()
            grads = torch.autograd.grad([st.memory], [k] + params.tensors(), [gmem[lo:hi]], allow_unused=True)
        return grads

    def step(nchain, seed, streams):
        cur = torch.cuda.current_stream()
        for s in streams[:nchain]:
            s.wait_stream(cur)
        per = B // nchain
        outs = [chain(c * per, (c + 1) * per, seed, streams[c]) for c in range(nchain)]
        for s in streams[:nchain]:
            cur.wait_stream(s)
        # parameter gradients of the towers add up to the full-batch gradient
        total = [g for g in outs[0][1:]]
        for o in outs[1:]:
            total = [a + b if (a is not None and b is not None) else (a if b is None else b) for a, b in zip(total, o[1:])]
        return outs, total

    streams = [torch.cuda.Stream() for _ in range(4)]
    ref = None
    for nchain in (1, 2, 4):
        for i in range(6):
            outs, total = step(nchain, 7, streams)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            outs, total = step(nchain, 7, streams)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        gW = [g for g in total if g is not None and g.numel() == D * D][0]
        if ref is None:
            ref = gW.clone()
        err = float((gW - ref).abs().max() / ref.abs().max())
        print("%d chain(s): %.3f ms per 64 questions  %.0f q/s   (gradient vs 1 chain: rel %.1e)" % (nchain, dt * 1e3, B / dt, err))

if __name__ == "__main__":
    main()
