#!/usr/bin/env python3
"""Debug aid: where does the H2 GEMM differ from fp64?  python tools/h2_debug.py [B N K n_out ...]"""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import macx
L = macx._lib.lib()
p = lambda t: C.c_void_p(t.data_ptr())
dev = torch.device("cuda:0")


def run(B, N, K, n_out, mode="rand"):
    g = torch.Generator().manual_seed(4)
    if mode == "rand":
        A = torch.randn(B, N, K, generator=g)
        W = torch.randn(K, n_out, generator=g) / 22
    else:  # A[r][k] = 1 if k == kk else 0  -> out[r][j] = W[kk][j]
        kk = int(mode)
        A = torch.zeros(B, N, K); A[:, :, kk] = 1.0
        W = (torch.arange(K).float().unsqueeze(1) * 1000 + torch.arange(n_out).float().unsqueeze(0))
    b = torch.zeros(n_out)
    ref = A.double().reshape(-1, K) @ W.double() + b.double()
    n = L.macx_h2_floats(B * N, K) + L.macx_h2_floats(B * N, n_out) + K * n_out + 64
    ws = torch.zeros(n, device=dev)
    out = torch.zeros(B * N, n_out, device=dev)
    macx._lib.check(L.macx_h2_gemm(p(A.to(dev)), B, N, K, p(W.to(dev)), n_out, p(b.to(dev)), 0, p(out), p(ws), n, None), "h2_gemm")
    torch.cuda.synchronize()
    e = (out.cpu().double() - ref).abs() / ref.abs().max()
    return float(e.max()), out.cpu(), ref


for B, N, K, n_out in [(2, 196, 512, 512), (2, 196, 256, 256), (2, 196, 128, 128), (2, 196, 384, 128), (2, 16, 128, 128), (2, 16, 256, 128),
                       (8, 196, 128, 128), (64, 196, 128, 128), (64, 196, 256, 128), (2, 209, 256, 128), (5, 100, 256, 512)]:
    e, _, _ = run(B, N, K, n_out)
    print("B %3d N %3d K %3d n_out %3d  rel err %.2e" % (B, N, K, n_out, e), flush=True)
for kk in (0, 5, 8, 31, 32, 40, 64, 100, 127):
    e, out, ref = run(2, 16, 128, 128, str(kk))
    print("one-hot k=%3d: err %.2e  out[0,:4]=%s (expect %s)  out[3,:2]=%s" % (kk, e, out[0, :4].tolist(), ref[0, :4].tolist(), out[3, :2].tolist()), flush=True)
print("---- which rows are right (one-hot k=5, expect 5000+j)")
for B, N, K, n_out in [(2, 16, 128, 128), (2, 32, 128, 128), (2, 196, 128, 128), (8, 196, 512, 512)]:
    e, out, ref = run(B, N, K, n_out, "5")
    good = ((out.double() - ref).abs().max(1).values < 1e-2 * 5000)
    zero = (out.abs().max(1).values == 0)
    print("B %d N %d K %d: good rows %d / %d, all-zero rows %d; first good %s first bad %s" % (
        B, N, K, int(good.sum()), B * N, int(zero.sum()), good.nonzero().flatten()[:12].tolist(), (~good).nonzero().flatten()[:12].tolist()))
    r = int((~good & ~zero).nonzero()[0]) if (~good & ~zero).any() else None
    if r is not None:
        print("   bad row", r, "out[:4]", out[r, :4].tolist(), "cols good in that row:", int(((out[r].double() - ref[r]).abs() < 50).sum()))
