#!/usr/bin/env python3
"""debug aid: where are the non-finite outputs of macx_h2_gemm?  python tools/h2_nan_probe.py B N K n_out"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import macx
L = macx._lib.lib()
if os.environ.get("MACX_DBG"):
    L.macx_debug_set(1, int(os.environ["MACX_DBG"]))
p = lambda t: C.c_void_p(t.data_ptr())
dev = torch.device("cuda:0")
B, N, K, n_out = [int(v) for v in sys.argv[1:5]]
g = torch.Generator().manual_seed(4)
A = torch.randn(B, N, K, generator=g)
W = torch.randn(K, n_out, generator=g) / 22
b = torch.randn(n_out, generator=g)
if int(os.environ.get("MACX_DBG", "0")) & 8:
    b = torch.zeros(n_out)
ref = A.double().reshape(-1, K) @ W.double() + b.double()
n = L.macx_h2_floats(B * N, K) + L.macx_h2_floats(B * N, n_out) + K * n_out + 64
ws = torch.zeros(n, device=dev)
out = torch.zeros(B * N, n_out, device=dev)
Ad, Wd, bd = A.to(dev), W.to(dev), b.to(dev)
macx._lib.check(L.macx_h2_gemm(p(Ad), B, N, K, p(Wd), n_out, p(bd), 0, p(out), p(ws), n, None), "h2_gemm")
torch.cuda.synchronize()
o = out.cpu().double()
bad = ~torch.isfinite(o)
print("non-finite:", int(bad.sum()), "of", o.numel())
if bad.any():
    rows = bad.any(1).nonzero().flatten()
    cols = bad.any(0).nonzero().flatten()
    print("rows", rows[:20].tolist(), "... n", len(rows)); print("cols", cols[:20].tolist(), "... n", len(cols))
e = ((o - ref).abs() / (ref.abs() + 1))
e[bad] = 0
print("max rel err of finite", float(e.max()))
import math
if int(os.environ.get("MACX_DBG", "0")) & 8:
    fr = (o - b.double()).abs().clamp_min(1e-300).log2()
    print("log2 f per row:", [round(float(v), 1) for v in fr[:24, 0]], " col spread", float((fr.max(1).values - fr.min(1).values).max()))
rat = (o / ref).abs().clamp_min(1e-300).log2()
print("log2 |out/ref| per row (median over cols), first rows:", [round(float(v), 1) for v in rat.median(1).values[:12]])
r, c = divmod(int(e.argmax()), n_out)
print("worst at", r, c, float(o[r, c]), float(ref[r, c]))
