"""-m gpu: the HIP path against vectors PRODUCED BY THE REFERENCE'S OWN CODE, with no oracle in between.

tests/golden/reference/hip_<flagfile>_p<p>.npz were written by tests/golden/make_reference_golden.py: the reference's
mac_cell.py / ops.py / model.py executed unmodified (fp64) on these inputs, on parameters from
helpers.hashed_reference_params, in TRAINING mode with the dropout masks of the product's stateless stream injected as
its random draws.  The bar is the task's: classifier logits within 1e-4, identical answer argmax; states, attentions and
every gradient are compared as well."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from helpers import hashed_reference_params, rel_err, max_abs

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FILES = sorted(glob.glob(os.path.join(HERE, "golden", "reference", "hip_*.npz")))


def test_fixtures_present():
    assert len(FILES) >= 7
    assert any(f.endswith("hip_args_d512_p4.npz") for f in FILES)       # full width: the chain kernels' 64-row tiles


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[4:-4] for f in FILES])
def test_hip_path_reproduces_reference_vectors(macx, dev, path):
    z = np.load(path)
    B, S, N, D, p, HID, ANS, seed = [int(v) for v in z["shape"]]
    flags = json.loads(bytes(z["flags"]).decode())
    cfg = macx.configs.default_config(answerWordsNum=ANS, **flags)
    shapes = {k: tuple(v) for k, v in json.loads(bytes(z["var_shapes"]).decode()).items()}
    ref_params = hashed_reference_params(shapes, seed)
    params = macx.MACCellParams(cfg, p).to(dev)
    params.load_reference_dict(ref_params)
    out = macx.OutputClassifier(cfg).to(dev)
    with torch.no_grad():
        for f, name in macx.output.REF_NAMES.items():
            getattr(out, f).copy_(ref_params[name].to(dev))
    if "kb" in z.files:
        vq, words, kb = [torch.from_numpy(z[k]).to(dev).requires_grad_(True) for k in ("vecQ", "words", "kb")]
    else:
        # a compact full-width fixture: inputs regenerated from the seed (the generator of the fixture used the same function)
        vq_c, words_c, len_c, kb_c = macx.configs.synthetic_inputs(B, S, N, D, seed=seed)
        assert np.array_equal(len_c.numpy(), z["lengths"])
        cs = z["input_checksums"]
        got_cs = [float(kb_c.double().sum()), float(kb_c.double().abs().sum()), float(words_c.double().sum()), float(vq_c.double().sum())]
        assert np.allclose(got_cs, cs, rtol=1e-9, atol=1e-6), (got_cs, cs)
        vq, words, kb = [t.to(dev).requires_grad_(True) for t in (vq_c, words_c, kb_c)]
    lengths = torch.from_numpy(z["lengths"]).to(dev)
    cell = macx.MACCell(vecQuestions=vq, questionWords=words, questionCntxWords=words, questionLengths=lengths,
                        knowledgeBase=kb, memoryDropout=cfg.memoryDropout, readDropout=cfg.readDropout,
                        writeDropout=cfg.writeDropout, batchSize=B, train=True, config=cfg, params=params, seed=seed)
    state = cell.run()
    logits = out(state.memory, vq, train=True, seed=seed)
    answers = torch.from_numpy(z["answers"]).to(dev)
    loss, preds = macx.output.answer_loss_and_pred(logits, answers)
    loss.backward()
    torch.cuda.synchronize()

    ref = lambda k: torch.from_numpy(np.asarray(z[k]))
    assert max_abs(logits, ref("logits")) < 1e-4                         # north-star bar
    assert torch.equal(logits.argmax(-1).cpu(), ref("logits").argmax(-1))
    assert torch.equal(preds.cpu().long(), ref("preds").long())
    assert abs(float(loss) - float(ref("loss"))) < 1e-5
    assert rel_err(state.memory, ref("memory")) < 2e-5 and rel_err(state.control, ref("control")) < 2e-5
    assert rel_err(cell.controls, ref("controls")) < 2e-5
    assert rel_err(cell.memories, ref("memories")) < 2e-5
    assert rel_err(cell.infos, ref("infos")) < 2e-5
    for key in ("kb", "question", "self", "gate"):
        n = len([k for k in z.files if k.startswith("att_%s_" % key)])
        assert len(cell.attentions[key]) == n, key
        for i in range(n):
            assert max_abs(cell.attentions[key][i], ref("att_%s_%d" % (key, i))) < 2e-6, (key, i)
    bad = {}
    for k, t in (("gin/vecQ", vq), ("gin/questionCntxWords", words), ("gin/kb", kb)):
        if k in z.files:
            e = rel_err(t.grad, ref(k))
        else:
            e = max(rel_err(t.grad.sum(1), ref("ginsum1/" + k[4:])), rel_err(t.grad.sum(2), ref("ginsum2/" + k[4:])))
        if not e < 2e-4:
            bad[k] = e
    got = {}
    names = macx.params.reference_names(cfg, p)
    for f in params.fields:
        for refname, idx in names[f]:
            g = getattr(params, f).grad
            got[refname] = g if idx is None else g[idx]
    for f, name in macx.output.REF_NAMES.items():
        got[name] = getattr(out, f).grad
    for name, g in got.items():
        floor = 5e-2 if name.endswith("linearLayerlogits/biases/bias") else 1e-6     # analytically zero: compare absolutely
        if "grad/" + name in z.files:
            r = ref("grad/" + name)
            e = rel_err(g.reshape(r.shape), r, floor=floor)
        else:
            e = max(rel_err(g.sum(0), ref("gsum0/" + name)), rel_err(g.sum(1), ref("gsum1/" + name)))
        if not e < 2e-4:
            bad[name] = e
    assert not bad, bad
