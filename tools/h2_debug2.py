#!/usr/bin/env python3
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import macx
L = macx._lib.lib()
p = lambda t: C.c_void_p(t.data_ptr())
dev = torch.device("cuda:0")
B, N, K, n_out = 2, 196, 128, 128
A = torch.zeros(B, N, K); A[:, :, 5] = 1.0
W = (torch.arange(K).float().unsqueeze(1) * 1000 + torch.arange(n_out).float().unsqueeze(0))
b = torch.zeros(n_out)
fa = L.macx_h2_floats(B * N, K)
n = fa + L.macx_h2_floats(B * N, n_out) + K * n_out + 64
ws = torch.zeros(n, device=dev)
out = torch.zeros(B * N, n_out, device=dev)
macx._lib.check(L.macx_h2_gemm(p(A.to(dev)), B, N, K, p(W.to(dev)), n_out, p(b.to(dev)), 0, p(out), p(ws), n, None), "h2_gemm")
back = torch.zeros(B * N, K, device=dev)
macx._lib.check(L.macx_h2_to_f32(p(ws), B * N, K, p(back), None), "to")
torch.cuda.synchronize()
aok = (back.cpu() == A.reshape(-1, K)).all(1)
print("stage", os.environ.get("MACX_H2_DEBUG_STAGE"), ": intact A rows", int(aok.sum()), "bad", (~aok).nonzero().flatten()[:6].tolist())
raw = ws.cpu().numpy().view("uint8")
Rp = B * N + 64
pb = (K // 8) * Rp * 16
ex = raw[2 * pb: 2 * pb + Rp].view("int8")
hi = raw[:pb].view("float16").reshape(K // 8, Rp, 8)
lo = raw[pb:2 * pb].view("float16").reshape(K // 8, Rp, 8)
print("   exps rows 0..3", ex[:4].tolist(), " hi[0,0]", hi[0, 0].tolist(), " lo[0,0]", lo[0, 0].tolist(), "hi[3,5]", hi[3, 5].tolist())
