#!/usr/bin/env python3
"""Run-to-run determinism of the metric step (B=64, p=12): every gradient of two eager runs on the same inputs must be bit-identical;
which tensors differ, under macx_debug_set(8, 0/1) (S_b kernel: 128x128 / 128x256).  Fresh (garbage-filled) buffers between runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import macx
dev = torch.device("cuda:0")
B, S, N, d, p = 64, 50, 196, 512, 12
cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(0)).to(dev)
g = torch.Generator().manual_seed(20240520)
vq = torch.randn(B, d, generator=g).to(dev).requires_grad_(True)
words = torch.randn(B, S, d, generator=g).to(dev).requires_grad_(True)
kb = torch.randn(B, N, d, generator=g).to(dev).requires_grad_(True)
lengths = torch.full((B,), S, dtype=torch.int32, device=dev)
gm = torch.randn(B, d, generator=g).to(dev)
L = macx._lib.lib()
leaves = [vq, words, kb] + params.tensors()
names = ["vecQ", "words", "kb"] + list(params.fields)

def run():
    for t in leaves:
        t.grad = None
    junk = torch.full((300_000_000,), float("nan"), device=dev)      # what the next run's torch.empty buffers will hold
    del junk
    cell = macx.MACCell(vq, words, words, lengths, kb, cfg.memoryDropout, cfg.readDropout, cfg.writeDropout, B, True, config=cfg,
                        params=params, seed=5)
    st = cell.run()
    torch.autograd.backward([st.memory], [gm])
    torch.cuda.synchronize()
    return [st.memory.detach().clone()] + [t.grad.clone() for t in leaves]

for mode in (0, 1):
    L.macx_debug_set(8, mode)
    a = run(); b = run(); c = run()
    bad = [n for n, x, y, z in zip(["memory"] + names, a, b, c) if not (torch.equal(x, y) and torch.equal(x, z))]
    nonfinite = [n for n, x in zip(["memory"] + names, a) if not torch.isfinite(x).all()]
    print("sb wide =", mode, " tensors that differ between runs:", bad, " non-finite:", nonfinite, flush=True)
