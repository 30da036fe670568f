"""GPU box: one fused-cell configuration against the fp64 oracle in each kernel family -- per-tensor relative errors.
python tests/case_probe.py name B S N d p train seed key=value ...   (test infrastructure: it uses the oracle, so it lives
under tests/)"""
import ast
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import macx  # noqa: E402
from helpers import make_case, oracle_run, rel_err  # noqa: E402
from test_gpu_cell import build_cell  # noqa: E402

name, B, S, N, d, p, train, seed = sys.argv[1], *[int(x) for x in sys.argv[2:7]], sys.argv[7] == "1", int(sys.argv[8])
over = {}
for kv in sys.argv[9:]:
    k, v = kv.split("=", 1)
    try:
        over[k] = ast.literal_eval(v)
    except Exception:
        over[k] = v
dev = torch.device("cuda:0")
cfg, vq, words, lengths, kb = make_case(name, B, S, N, d, p, **over)
g = torch.Generator().manual_seed(9)
dmem, dctl = torch.randn(B, d, generator=g) / B, torch.randn(B, d, generator=g) / B
ref = None
for fam in ("h2", "split", "native"):
    cell, params, (vqd, wd, kbd) = build_cell(macx, dev, cfg, vq, words, lengths, kb, train, seed=seed, requires_grad=True)
    cell2 = macx.MACCell(vqd, wd, wd, lengths.to(dev), kbd, cfg.memoryDropout, cfg.readDropout, cfg.writeDropout, B, train, config=cfg,
                         params=params, seed=seed, gemm=fam)
    st = cell2.run()
    ((st.memory * dmem.to(dev)).sum() + (st.control * dctl.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    if ref is None:
        ref = oracle_run(cfg, params.to_reference_dict(), vq, words, lengths, kb, train=train, seed=seed, need_grad=True, d_memory=dmem, d_control=dctl)
    names = macx.params.reference_names(cfg, p)
    errs = {"memory": rel_err(st.memory, ref["memory"]), "dKB": rel_err(kbd.grad, ref["inputs"][2].grad)}
    for f in params.fields:
        gt = getattr(params, f).grad
        for refname, idx in names[f]:
            rg = ref["params"][refname].grad
            if rg is not None and float(rg.abs().max()) > 1e-9:
                errs[refname.split("MACCell/")[-1]] = rel_err((gt if idx is None else gt[idx]).reshape(rg.shape), rg)
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
    print(fam, " ".join("%s=%.2e" % (k[-40:], v) for k, v in worst))
