"""The B = 128 training step of bench.py's train_b128 leg, 10 times, wall time per piece (see tools/slow_host_probe.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, macx
dev = torch.device("cuda:0")
B, S, N, D, P = 128, 50, 196, 512, 12
cfg = macx.configs.flag_file_config("args", netLength=P, memDim=D, ctrlDim=D, attDim=D)
vq, words, lengths, kb = macx.configs.synthetic_inputs(B, S, N, D, seed=1)
params = macx.MACCellParams(cfg, P, generator=torch.Generator().manual_seed(1)).to(dev)
vqd, wd, kbd = [t.to(dev).requires_grad_(True) for t in (vq, words, kb)]
ld = lengths.to(dev)
gm = (torch.randn(B, D) / B).to(dev)
out = []
for i in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cell = macx.MACCell(vqd, wd, wd, ld, kbd, cfg.memoryDropout, cfg.readDropout, cfg.writeDropout, B, True, config=cfg, params=params, seed=i)
    st = cell.run()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for t in [vqd, wd, kbd] + list(params.tensors()):
        t.grad = None
    torch.autograd.backward([st.memory], [gm])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    out.append("%.1f/%.1f@%x" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, cell._run.saved.data_ptr() >> 20 if cell._run is not None else 0))
    del cell, st
print("fwd/bwd ms @ saved address (MB):", " ".join(out))
