#!/usr/bin/env python3
"""Randomised cross-check of the two GEMM kernel families on the whole cell: for random (flag file, B, S, N, d, p, train)
the split-bf16 run and the native f32-MFMA run must agree on the final state and on every gradient to fp32 round-off
(they share everything but the large contractions).  python tools/mode_fuzz.py [n_cases] [seed]"""
import os, random, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import macx
from test_gpu_cell import build_cell
from helpers import make_case


def run(mode, name, B, S, N, d, p, train, seed):
    macx._lib.lib().macx_gemm_mode(mode)
    cfg, vq, words, lengths, kb = make_case(name, B, S, N, d, p)
    cell, params, (vqd, wd, kbd) = build_cell(macx, torch.device("cuda:0"), cfg, vq, words, lengths, kb, train, seed=seed, b0=3,
                                              requires_grad=True)
    st = cell.run()
    g = torch.Generator().manual_seed(1)
    gm = torch.randn(B, d, generator=g).cuda() / B
    torch.autograd.backward([st.memory, st.control], [gm, gm * 0.5])
    torch.cuda.synchronize()
    out = {"memory": st.memory.detach().clone(), "control": st.control.detach().clone(), "dKB": kbd.grad.clone(), "dwords": wd.grad.clone(),
           "dvq": vqd.grad.clone()}
    for f in params.fields:
        out["d" + f] = getattr(params, f).grad.clone()
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    worst = 0.0
    for case in range(n):
        name = rnd.choice(["args", "args1", "args3", "args4"])
        B = rnd.choice([1, 2, 3, 5, 8, 17]); S = rnd.randint(3, 12); N = rnd.choice([1, 7, 16, 30, 49, 100, 113, 196, 209, 250])
        d = rnd.choice([128, 256]); p = rnd.randint(1, 4); train = rnd.random() < 0.7
        a = run(1, name, B, S, N, d, p, train, case)
        b = run(0, name, B, S, N, d, p, train, case)
        bad = []
        for k in a:
            den = float(b[k].abs().max()) + 1e-20
            e = float((a[k] - b[k]).abs().max()) / den
            if den > 1e-6:          # gradients that are identically ~0 (e.g. unused logit bias) carry no signal
                worst = max(worst, e)
                if e > 2e-4:
                    bad.append((k, e))
        print("case %2d %-5s B=%-2d S=%-2d N=%-3d d=%d p=%d train=%d  %s" % (case, name, B, S, N, d, p, train, "OK" if not bad else bad[:4]), flush=True)
        assert not bad
    macx._lib.lib().macx_gemm_mode(1)
    print("all %d cases agree; worst relative difference %.2e" % (n, worst))


if __name__ == "__main__":
    main()
