"""CPU, runs anywhere: oracle/mac_oracle.py against vectors PRODUCED BY THE REFERENCE'S OWN CODE.

tests/golden/reference/oracle_*.npz were written by tests/golden/make_reference_golden.py, which executes
/root/reference/{config,ops,mac_cell,model}.py unmodified (tests/ref_exec.py).  Each file holds the flag values the
reference's parser produced, the variables its graph created, the inputs, its uniform draws and all of its outputs and
gradients in fp64; the oracle must reproduce them to 1e-12.  (The live comparison, which needs /root/reference, is
tests/test_reference_exec.py.)"""
import glob
import json
import os

import numpy as np
import pytest
import torch

import ref_exec as rx
from oracle import mac_oracle as mo

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "reference", "oracle_*.npz")))


def _json(z, key):
    return json.loads(bytes(z[key]).decode())


def test_fixtures_present():
    assert len(FILES) >= 10 + 30, "reference-generated vectors missing: run tests/golden/make_reference_golden.py"


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[7:-4] for f in FILES])
def test_oracle_reproduces_reference_vectors(path):
    z = np.load(path)
    flags, meta = _json(z, "flags"), _json(z, "meta")
    ANS = int(z["answerWordsNum"])
    cfg = mo.default_config(answerWordsNum=ANS, **flags)
    keeps, output_keep, train = tuple(meta["keeps"]), meta["output_keep"], meta["train"]
    names = _json(z, "var_names")
    params = {k: torch.from_numpy(z["var/" + k]).clone().requires_grad_(True) for k in names}
    vs = mo.VarStore(params=params, dtype=torch.float64)
    draws = [torch.from_numpy(z["draw_%d" % i]) for i in range(int(z["n_draws"]))]
    if cfg.memoryVariationalDropout and keeps[0] == 1.0:
        draws = draws[1:]                      # generateVarDpMask draws even at keep == 1 (all-ones mask)
    mask_fn = rx.replay_mask_fn(draws, keeps)
    ins = {k: torch.from_numpy(z[k]).clone().requires_grad_(True) for k in ("vecQ", "questionWords", "questionCntxWords", "kb")}
    lengths = torch.from_numpy(z["lengths"])
    c, m, cell = mo.mac_network(cfg, vs, ins["vecQ"], ins["questionWords"], ins["questionCntxWords"], lengths, ins["kb"],
                                train=train, mask_fn=mask_fn, keeps=keeps)
    masks = mask_fn.rest(output_keep) if output_keep != 1.0 else None
    logits = mo.output_classifier(cfg, vs, m, ins["vecQ"], output_keep=output_keep, masks=masks)
    assert mask_fn.left() == 0
    answers = torch.from_numpy(z["answers"])
    loss, preds = mo.answer_loss_and_pred(logits, answers)

    def close(a, key, tol=1e-12):
        b = torch.from_numpy(np.asarray(z[key]))
        assert tuple(a.shape) == tuple(b.shape), key
        assert float((a.detach().double() - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), key

    assert list(vs.params.keys()) == names                      # same variables, none extra
    close(c, "control"); close(m, "memory"); close(cell.controls, "controls"); close(cell.memories, "memories")
    close(cell.infos, "infos"); close(logits, "logits"); close(loss, "loss")
    assert torch.equal(preds.long(), torch.from_numpy(z["preds"]).long())
    for key in ("kb", "question", "self", "gate"):
        n = len([k for k in z.files if k.startswith("att_%s_" % key)])
        assert len(cell.attentions[key]) == n
        for i in range(n):
            close(cell.attentions[key][i], "att_%s_%d" % (key, i))
    (loss + (c * torch.from_numpy(z["d_control"])).sum()).backward()
    for k in names:
        if "grad/" + k in z.files:
            close(params[k].grad, "grad/" + k)
        else:
            assert params[k].grad is None or float(params[k].grad.abs().max()) == 0.0, k
    for k, t in ins.items():
        if "gin/" + k in z.files:
            close(t.grad, "gin/" + k)


def test_optimizer_oracle_reproduces_the_reference_training_op():
    """oracle/optim_oracle.adam_ema_step, fed the gradients MACnet.computeGradients produced, lands on the variables, Adam moments,
    EMA shadows and global norm MACnet.addTrainingOp (model.py:639-669) left behind -- five steps, clip active on three."""
    from helpers import load_training_fixture, flat64
    from oracle import optim_oracle as oo
    hyper, names, init, steps = load_training_fixture()
    p = flat64(init, names)
    m, v, e = p * 0, p * 0, p.copy()
    clipped = 0
    for t, s in enumerate(steps, start=1):
        p, m, v, e, norm = oo.adam_ema_step(p, flat64(s["g"], names), m, v, e, hyper["lr"], t, beta1=hyper["beta1"], beta2=hyper["beta2"],
                                            eps=hyper["eps"], clip=hyper["clip"], decay=hyper["decay"])
        clipped += norm > hyper["clip"]
        assert abs(norm - s["norm"]) < 1e-12 * s["norm"]
        for got, key in ((p, "p"), (m, "m"), (v, "v"), (e, "e")):
            assert np.abs(got - flat64(s[key], names)).max() < 1e-12, (t, key)
    assert 0 < clipped < len(steps)
