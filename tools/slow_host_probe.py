"""Why is the eager step slow on some boxes of the pool?  Host-side rates, one line each: kernel launches through the C ABI and
through torch, big allocations through the caching allocator, and the pieces of the B = 128 training step (bench.py's
train_b128_p12_adam_ema leg) with a synchronize behind each."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import macx

dev = torch.device("cuda:0")
L = macx._lib.lib()
p_ = lambda t: C.c_void_p(t.data_ptr())


def rate(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


print("cpus", os.cpu_count(), "loadavg", os.getloadavg())
x = torch.zeros(256, device=dev)
y = torch.zeros(256, device=dev)
a, b = rate(lambda: L.macx_op_act(1, p_(x), None, 256, 1, p_(y), None), 3000)
print("C-ABI tiny kernel: %.1f us per call issued, %.1f us incl. drain" % (a, b))
a, b = rate(lambda: x.add_(1.0), 3000)
print("torch tiny kernel: %.1f us per call issued, %.1f us incl. drain" % (a, b))
t0 = time.perf_counter()
for _ in range(10):
    big = torch.empty(700_000_000, device=dev)
    del big
print("torch.empty(2.8 GB) + del: %.1f us each (cached after the first)" % ((time.perf_counter() - t0) / 10 * 1e6))

B, S, N, D, P = 128, 50, 196, 512, 12
cfg = macx.configs.flag_file_config("args", netLength=P, memDim=D, ctrlDim=D, attDim=D)
vq, words, lengths, kb = macx.configs.synthetic_inputs(B, S, N, D, seed=1)
params = macx.MACCellParams(cfg, P, generator=torch.Generator().manual_seed(1)).to(dev)
vqd, wd, kbd = [t.to(dev).requires_grad_(True) for t in (vq, words, kb)]
ld = lengths.to(dev)
gm = (torch.randn(B, D) / B).to(dev)
opt = macx.optim.FlatAdamEMA(params.tensors(), lr=1e-4, clip_norm=8.0, ema_decay=0.999)
acc = {}


def tick(name, t0):
    torch.cuda.synchronize()
    acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)


for i in range(7):
    t0 = time.perf_counter()
    cell = macx.MACCell(vqd, wd, wd, ld, kbd, cfg.memoryDropout, cfg.readDropout, cfg.writeDropout, B, True, config=cfg, params=params, seed=i)
    tick("ctor", t0)
    t0 = time.perf_counter()
    st = cell.run()
    tick("forward", t0)
    for t in [vqd, wd, kbd] + list(params.tensors()):
        t.grad = None
    t0 = time.perf_counter()
    torch.autograd.backward([st.memory], [gm])
    tick("backward", t0)
    t0 = time.perf_counter()
    opt.step()
    tick("opt.step (gather path)", t0)
    t0 = time.perf_counter()
    del cell, st
    tick("release", t0)
for k, v in acc.items():
    print("B=128 %-24s %s ms" % (k, " ".join("%.2f" % t for t in v)))
free, total = torch.cuda.mem_get_info()
ms = torch.cuda.memory_stats()
print("mem_get_info free %.1f GB of %.1f GB; allocator: device_alloc %d device_free %d alloc_retries %d ooms %d reserved %.1f GB" % (
    free / 2**30, total / 2**30, ms.get("num_device_alloc", -1), ms.get("num_device_free", -1), ms.get("num_alloc_retries", -1),
    ms.get("num_ooms", -1), ms.get("reserved_bytes.all.current", 0) / 2**30))
print("env:", {k: v for k, v in os.environ.items() if "ALLOC" in k or "PYTORCH" in k or k.startswith("HSA") or k.startswith("HIP")})
