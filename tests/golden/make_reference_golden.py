#!/usr/bin/env python3
"""Generates tests/golden/reference/*.npz by EXECUTING THE REFERENCE'S OWN CODE (/root/reference/{config,ops,mac_cell,
model}.py, unmodified, over the eager TF-1.x stand-in of tests/tf1_shim -- see tests/ref_exec.py).  Run in the build
container, where /root/reference exists; the vectors travel to the GPU box, the reference does not.

    python tests/golden/make_reference_golden.py

Two families:
  oracle_<case>.npz   small fp64 cases (d = 8): flag values as the reference's parser produced them, the reference-created
                      variables, inputs, the uniform draws, and every output (states, histories, attentions, logits, loss,
                      predictions, gradients).  tests/test_reference_golden.py replays them through oracle/mac_oracle.py.
  hip_<flagfile>.npz  d = 128 cases for the HIP path (tests/test_gpu_reference_golden.py): parameters come from
                      helpers.hashed_reference_params (not stored), dropout masks from the product's stateless stream
                      (oracle/dropout_hash.py) converted to uniform draws and injected into the reference run.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import ref_exec as rx                      # noqa: E402
import helpers                             # noqa: E402
from oracle import dropout_hash as dh      # noqa: E402
from oracle import mac_oracle as mo        # noqa: E402
import test_reference_exec as T            # noqa: E402  (the option tables)

OUT = os.path.join(HERE, "reference")


def np64(t):
    return torch.as_tensor(t).detach().double().numpy()


def pack_run(ref, cfg, keeps, output_keep, train, answers, grads=True):
    rc = ref["cell"]
    z = {"flags": np.frombuffer(json.dumps(rx.snapshot(cfg)).encode(), dtype=np.uint8),
         "meta": np.frombuffer(json.dumps(dict(keeps=list(keeps), output_keep=output_keep, train=bool(train))).encode(), dtype=np.uint8),
         "var_names": np.frombuffer(json.dumps(list(ref["variables"].keys())).encode(), dtype=np.uint8),
         "answers": answers.numpy(), "control": np64(ref["control"]), "memory": np64(ref["memory"]),
         "controls": np64(rc.controls), "memories": np64(rc.memories), "infos": np64(rc.infos),
         "logits": np64(ref["logits"]), "loss": np64(ref["loss"]), "preds": np64(ref["preds"]).astype(np.int64)}
    for key in ("kb", "question", "self", "gate"):
        for i, a in enumerate(rc.attentions[key]):
            z["att_%s_%d" % (key, i)] = np64(a)
    for i, u in enumerate(ref["draws"]):
        z["draw_%d" % i] = np64(u)
    z["n_draws"] = np.int64(len(ref["draws"]))
    return z


def small_case(tag, flag_file, extra, train):
    cfg = rx.parse_flags(flag_file, *(list(extra) + rx.dims_flags(8, 3, 6)))
    B, S, N, D, ANS = 3, 5, 6, 8, 5
    vq, words, lengths, kb = mo.synthetic_inputs(B, S, N, D, seed=21, dtype=torch.float64)
    g = torch.Generator().manual_seed(22)
    raw = torch.rand((B, S, D), generator=g, dtype=torch.float64) * 2 - 1
    answers = torch.randint(0, ANS, (B,), generator=g)
    keeps = (cfg.memoryDropout, cfg.readDropout, 0.9) if train else (1.0, 1.0, 1.0)
    if train:
        cfg.writeDropout = 0.9
    output_keep = cfg.outputDropout if train else 1.0
    ref = rx.run_reference(cfg, vq, raw, words, lengths, kb, train=train, keeps=keeps, output_keep=output_keep, seed=7,
                           need_grad=True, answers=answers, answerWordsNum=ANS)
    dc = torch.randn((B, D), generator=g, dtype=torch.float64)
    (ref["loss"] + (ref["control"] * dc).sum()).backward()
    z = pack_run(ref, cfg, keeps, output_keep, train, answers)
    z.update(vecQ=vq.numpy(), questionWords=raw.numpy(), questionCntxWords=words.numpy(), lengths=lengths.numpy(),
             kb=kb.numpy(), d_control=dc.numpy(), answerWordsNum=np.int64(ANS))
    for k, v in ref["variables"].items():
        z["var/" + k] = np64(v)
        if v.grad is not None:
            z["grad/" + k] = np64(v.grad)
    for k, t in ref["inputs"].items():
        if t.grad is not None:
            z["gin/" + k] = np64(t.grad)
    np.savez_compressed(os.path.join(OUT, "oracle_%s.npz" % tag), **z)


def u_from_mask(mask, keep):
    """a uniform draw u with floor(keep + u) == mask"""
    m = torch.as_tensor(mask, dtype=torch.float64)
    return m * (1.0 - keep / 2.0) + (1.0 - m) * ((1.0 - keep) / 2.0)


def hip_case(name, p, B=2, S=6, N=20, D=128, HID=64, ANS=28, seed=31, compact=False, tag=None):
    """compact: a full-width case (d = 512, N = 196 -- the chain kernels' 64-row-tile geometry) kept small: the inputs are NOT
    stored (the test regenerates them from the seed: configs.synthetic_inputs is oracle.synthetic_inputs; two checksums are),
    and every large gradient travels as its sums along each axis."""
    cfg = rx.parse_flags(name + ".txt", *rx.dims_flags(D, p, HID))
    vq, words, lengths, kb = mo.synthetic_inputs(B, S, N, D, seed=seed, dtype=torch.float32)
    g = torch.Generator().manual_seed(seed + 1)
    answers = torch.randint(0, ANS, (B,), generator=g)
    keeps = (cfg.memoryDropout, cfg.readDropout, cfg.writeDropout)
    output_keep = cfg.outputDropout
    # variable table from a throw-away reference run, then hashed values for it
    probe = rx.run_reference(cfg, vq, words, words, lengths, kb, answerWordsNum=ANS)
    shapes = {k: tuple(v.shape) for k, v in probe["variables"].items()}
    preset = helpers.hashed_reference_params(shapes, seed)
    # the product's mask stream, in the reference's draw order: record what the oracle asks for
    ocfg = mo.default_config(answerWordsNum=ANS, **rx.snapshot(cfg))
    base = mo.hash_mask_fn(seed, keeps)
    order = []

    def rec(site, step, shape):
        m = base(site, step, shape)
        order.append((m, {dh.SITE_MEM_VAR: keeps[0], dh.SITE_MEM: keeps[0], dh.SITE_WRITE_INFO: keeps[2]}.get(site, keeps[1])))
        return m

    vs = mo.VarStore(params={k: v.double() for k, v in preset.items()}, dtype=torch.float64)
    mo.mac_network(ocfg, vs, vq.double(), words.double(), words.double(), lengths, kb.double(), train=True, mask_fn=rec, keeps=keeps)
    draws = [u_from_mask(m, k) for m, k in order]
    cls_masks = [dh.mask_for(seed, 7, 0, output_keep, (B, 2 * D)), dh.mask_for(seed, 8, 0, output_keep, (B, HID))]
    draws += [u_from_mask(m, output_keep) for m in cls_masks]
    ref = rx.run_reference(cfg, vq, words, words, lengths, kb, train=True, keeps=keeps, output_keep=output_keep,
                           preset=preset, need_grad=True, answers=answers, answerWordsNum=ANS, draws=draws)
    ref["loss"].backward()
    z = pack_run(ref, cfg, keeps, output_keep, True, answers)
    for k in [k for k in z if k.startswith("draw_")]:
        del z[k]
    if compact:
        z.update(lengths=lengths.numpy(), input_checksums=np.array([float(kb.double().sum()), float(kb.double().abs().sum()),
                                                                    float(words.double().sum()), float(vq.double().sum())]))
    else:
        z.update(vecQ=vq.numpy(), words=words.numpy(), lengths=lengths.numpy(), kb=kb.numpy())
    z.update(shape=np.array([B, S, N, D, p, HID, ANS, seed], dtype=np.int64),
             var_shapes=np.frombuffer(json.dumps({k: list(s) for k, s in shapes.items()}).encode(), dtype=np.uint8))
    for k, v in ref["variables"].items():
        gk = v.grad
        if gk is None:
            continue
        if gk.dim() == 2 and gk.numel() > 4096 and (compact or ("memKbProj_2" not in k and "newMemory" not in k)):
            z["gsum0/" + k] = np64(gk.sum(0)).astype(np.float32)      # column sums and row sums of the big matrices
            z["gsum1/" + k] = np64(gk.sum(1)).astype(np.float32)
        else:
            z["grad/" + k] = np64(gk).astype(np.float32)
    for k, t in ref["inputs"].items():
        if t.grad is not None and k != "questionWords":
            if compact and t.grad.dim() == 3:
                z["ginsum1/" + k] = np64(t.grad.sum(1)).astype(np.float32)
                z["ginsum2/" + k] = np64(t.grad.sum(2)).astype(np.float32)
            else:
                z["gin/" + k] = np64(t.grad).astype(np.float32)
    for k in ("controls", "memories", "infos"):
        z[k] = z[k].astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "hip_%s_p%d.npz" % (tag or name, p)), **z)


def metric_depth_case():
    """The metric's depth and geometry (p = 12, d = 512, N = 196, S = 50; 16 questions = 49 tiles of 64 rows) in training mode: the
    reference executed over twelve steps of the chain kernels' products, compact form."""
    hip_case("args", 12, B=16, S=50, N=196, D=512, HID=64, seed=41, compact=True, tag="args_d512")


def main():
    assert rx.available(), "needs /root/reference"
    os.makedirs(OUT, exist_ok=True)
    if "--metric-depth-only" in sys.argv:
        metric_depth_case()
        return
    for f in os.listdir(OUT):
        if f.endswith(".npz") and (f.startswith("oracle_") or f.startswith("hip_")):      # (training_steps.npz: make_training_golden.py)
            os.remove(os.path.join(OUT, f))
    for name in T.FLAG_FILES:
        for train in (False, True):
            small_case("%s_%s" % (name, "train" if train else "eval"), name + ".txt", [], train)
    for variant, extra in sorted(T.VARIANTS.items()):
        small_case("opt_%s_train" % variant, None, extra, True)
    for name in T.FLAG_FILES:
        hip_case(name, 4)
    hip_case("args", 12)
    # the chain kernels' d = 512 geometry (64-row tiles over 8 x 196 rows = 24.5 tiles) against the executed reference directly
    hip_case("args", 4, B=8, S=9, N=196, D=512, HID=64, seed=37, compact=True, tag="args_d512")
    metric_depth_case()
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("wrote %d files, %.1f KB" % (len(os.listdir(OUT)), total / 1024))


if __name__ == "__main__":
    main()
