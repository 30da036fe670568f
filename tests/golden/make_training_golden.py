#!/usr/bin/env python3
"""Generates tests/golden/reference/training_steps.npz by EXECUTING the reference's own training op -- MACnet.addOptimizerOp,
computeGradients and addTrainingOp (/root/reference/model.py:615-669, unmodified) behind MACnetwork / outputOp / classifier /
addAnswerLossOp -- for four steps on the eager TF-1.x stand-in (tests/ref_exec.run_reference_training; the stand-in's
tf.train.AdamOptimizer / ExponentialMovingAverage / clip_by_global_norm restate TF's).  Stored: the variables' initial values,
and per step the gradients the reference computed, the global norm, and the variables / Adam moments / EMA shadows afterwards.
oracle/optim_oracle.py (CPU) and macx_adam_ema_step (GPU) are fed the same gradients and must land on the same numbers.

    python tests/golden/make_training_golden.py        # build container only: /root/reference must exist
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
from oracle import mac_oracle as mo        # noqa: E402

STEPS, LR = 5, 4e-2
FLAGS = ["--gradMaxNorm", 1]               # (an int flag, config.py) -- the inputs are scaled so that the norms straddle 1


def run():
    cfg = rx.parse_flags("args.txt", *(FLAGS + rx.dims_flags(8, 2, 6)))
    vq, words, lengths, kb = mo.synthetic_inputs(4, 5, 6, 8, seed=11)
    answers = torch.tensor([1, 4, 2, 0])
    r = rx.run_reference_training(cfg, vq, words, words, lengths, kb, answers, steps=STEPS, lr=LR, seed=5)
    return cfg, r


def main():
    assert rx.available(), "needs /root/reference"
    cfg, r = run()
    names = list(r["steps"][0]["variables"])
    z = {"names": np.frombuffer(json.dumps(names).encode(), dtype=np.uint8),
         "hyper": np.frombuffer(json.dumps(dict(lr=LR, beta1=0.9, beta2=0.999, eps=1e-8, clip=float(cfg.gradMaxNorm),
                                                decay=float(cfg.emaDecayRate), steps=STEPS)).encode(), dtype=np.uint8)}
    # initial values: step 1's variables are  initial - update ; the stand-in keeps them (state.initial)
    tf = rx.load()["tf"]
    for i, n in enumerate(names):
        z["init_%d" % i] = tf.state.initial[n].double().numpy()
    for t, s in enumerate(r["steps"]):
        z["norm_%d" % t] = np.float64(s["norm"])
        z["loss_%d" % t] = np.float64(s["loss"])
        for i, n in enumerate(names):
            z["g_%d_%d" % (t, i)] = s["grads"][n].double().numpy()
            z["p_%d_%d" % (t, i)] = s["variables"][n].double().numpy()
            z["m_%d_%d" % (t, i)] = s["m"][n].double().numpy()
            z["v_%d_%d" % (t, i)] = s["v"][n].double().numpy()
            z["e_%d_%d" % (t, i)] = s["ema"][n].double().numpy()
    out = os.path.join(HERE, "reference", "training_steps.npz")
    np.savez_compressed(out, **z)
    print("wrote %s (%.1f KB), norms %s" % (out, os.path.getsize(out) / 1024, [round(s["norm"], 3) for s in r["steps"]]))


if __name__ == "__main__":
    main()
