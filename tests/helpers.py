"""Shared test plumbing: run the oracle (CPU) and the HIP cell on the same parameters and inputs."""
import ctypes as C

import numpy as np
import torch

from oracle import mac_oracle as mo


def make_case(config_name, B, S, N, d, p, seed=1234, **over):
    cfg = mo.flag_file_config(config_name, netLength=p, memDim=d, ctrlDim=d, attDim=d, **over)
    vq, words, lengths, kb = mo.synthetic_inputs(B, S, N, d, seed=seed)
    return cfg, vq, words, lengths, kb


def oracle_run(cfg, ref_params, vq, words, lengths, kb, train=False, seed=0, b0=0, dtype=torch.float64, need_grad=False,
               d_memory=None, d_control=None):
    """Oracle forward (+ backward).  ref_params: {reference variable name: tensor}."""
    params = {k: v.detach().cpu().to(dtype).clone().requires_grad_(need_grad) for k, v in ref_params.items()}
    vs = mo.VarStore(params=params, dtype=dtype)
    vq_, words_, kb_ = [t.detach().cpu().to(dtype).clone().requires_grad_(need_grad) for t in (vq, words, kb)]
    keeps = (cfg.memoryDropout, cfg.readDropout, cfg.writeDropout) if train else (1.0, 1.0, 1.0)
    mask_fn = mo.hash_mask_fn(seed, keeps, b0=b0) if train else None
    c, m, cell = mo.mac_network(cfg, vs, vq_, words_, words_, lengths.cpu(), kb_, train=train, mask_fn=mask_fn, keeps=keeps)
    out = dict(control=c, memory=m, cell=cell, params=params, inputs=(vq_, words_, kb_))
    if need_grad:
        loss = 0
        if d_memory is not None:
            loss = loss + (m * d_memory.detach().cpu().to(dtype)).sum()
        if d_control is not None:
            loss = loss + (c * d_control.detach().cpu().to(dtype)).sum()
        loss.backward()
    return out


def rel_err(a, b, floor=1e-6):
    """max |a-b| relative to the largest reference entry (floored: gradients that are analytically
    zero, e.g. d/d(logit bias) of a softmax, are compared absolutely)."""
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), floor))


def max_abs(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


def default_gemm_mode():
    """Kernel family the session runs on (MACX_GEMM=native|split, default split): what mode-switching tests restore."""
    import os
    return 0 if os.environ.get("MACX_GEMM") == "native" else 1
