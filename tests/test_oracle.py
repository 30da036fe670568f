"""CPU: the oracle against itself -- two independent restatements, the scalar golden vectors,
finite differences, and the invariants of SURVEY.md 8c.  (The reference ships no vectors: every fixture HERE is
generated in this repo and labelled so.  The pin to the reference itself is tests/test_reference_exec.py -- the reference's
own code executed on tests/tf1_shim -- and its committed outputs, tests/test_reference_golden.py.)"""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import dropout_hash as dh
from oracle import mac_numpy as mn
from oracle import mac_oracle as mo

HERE = os.path.dirname(os.path.abspath(__file__))


def run_both(name, B=3, S=7, N=10, d=8, p=3, train=False, seed=3, **over):
    cfg = mo.flag_file_config(name, netLength=p, memDim=d, ctrlDim=d, attDim=d, **over)
    vq, words, lengths, kb = mo.synthetic_inputs(B, S, N, d, seed=11, dtype=torch.float64)
    vs = mo.VarStore(generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    keeps = (cfg.memoryDropout, cfg.readDropout, 0.9) if train else (1.0, 1.0, 1.0)
    if train:
        cfg.writeDropout = 0.9
    mask_fn = mo.hash_mask_fn(seed, keeps) if train else None
    c, m, cell = mo.mac_network(cfg, vs, vq, words, words, lengths, kb, train=train, mask_fn=mask_fn, keeps=keeps)
    prm = {k: v.detach().numpy() for k, v in vs.params.items()}
    ref = mn.forward(cfg, prm, vq.numpy(), words.numpy(), lengths.numpy(), kb.numpy(), keeps=keeps, masks=mask_fn)
    return cfg, cell, c, m, ref


@pytest.mark.parametrize("name", ["args", "args1", "args2", "args3", "args4"])
@pytest.mark.parametrize("train", [False, True])
def test_torch_and_numpy_restatements_agree(name, train):
    cfg, cell, c, m, ref = run_both(name, train=train)
    assert np.abs(m.numpy() - ref["memory"]).max() < 1e-12
    assert np.abs(c.numpy() - ref["control"]).max() < 1e-12
    assert np.abs(cell.controls.transpose(0, 1).numpy() - ref["controls"]).max() < 1e-12
    assert np.abs(cell.memories.transpose(0, 1).numpy() - ref["memories"]).max() < 1e-12
    for i in range(cfg.netLength):
        assert np.abs(cell.attentions["kb"][i].numpy() - ref["att_kb"][i]).max() < 1e-13
        assert np.abs(cell.attentions["question"][i].numpy() - ref["att_q"][i]).max() < 1e-13
    if cfg.writeSelfAtt:
        for i in range(cfg.netLength):
            assert cell.attentions["self"][i].shape == (3, i + 1)       # histories hold init + steps 0..i-1
            assert np.abs(cell.attentions["self"][i].numpy() - ref["att_self"][i]).max() < 1e-13
    if cfg.writeGate:
        assert np.abs(cell.attentions["gate"][0].numpy() - ref["gates"][0]).max() < 1e-13


def test_fp32_oracle_tracks_fp64():
    cfg = mo.flag_file_config("args", netLength=4, memDim=64, ctrlDim=64, attDim=64)
    vq, words, lengths, kb = mo.synthetic_inputs(4, 9, 30, 64, seed=2, dtype=torch.float64)
    vs64 = mo.VarStore(generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    _, m64, _ = mo.mac_network(cfg, vs64, vq, words, words, lengths, kb)
    vs32 = mo.VarStore(params={k: v.float() for k, v in vs64.params.items()}, dtype=torch.float32)
    _, m32, _ = mo.mac_network(cfg, vs32, vq.float(), words.float(), words.float(), lengths, kb.float())
    assert (m32.double() - m64).abs().max() < 2e-5


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(HERE, "golden", "tiny_args_*.json"))))
def test_golden_scalar_vectors(path):
    """tests/golden/*.json come from the loop-and-math restatement in make_golden.py (ours, not the reference's)."""
    case = json.load(open(path))
    assert "NOT reference output" in case["provenance"]
    d, S, N, p = case["d"], case["S"], case["N"], case["p"]
    P = case["params"]
    cfg = mo.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
    Pn = "MACnetwork/MACCell/"
    t = lambda x: torch.tensor(x, dtype=torch.float64)
    prm = {"MACnetwork/initMem": t(P["initMem"]),
           Pn + "linearLayerqInput/weights/weight": t(P["qInput_W"]), Pn + "linearLayerqInput/biases/bias": t(P["qInput_b"]),
           Pn + "control/inter2logits/linearLayerlogits/weights/weight": t(P["ctrlLogits_w"]),
           Pn + "control/inter2logits/linearLayerlogits/biases/bias": t(P["ctrlLogits_b"]),
           Pn + "read/mulmemInter/linearLayerprojX/weights/weight": t(P["projX_W"]),
           Pn + "read/mulmemInter/linearLayerprojX/biases/bias": t(P["projX_b"]),
           Pn + "read/mulmemInter/linearLayerprojY/weights/weight": t(P["projY_W"]),
           Pn + "read/mulmemInter/linearLayerprojY/biases/bias": t(P["projY_b"]),
           Pn + "read/linearLayermemKbProj/weights/weight": t(P["memKbProj_W"]),
           Pn + "read/linearLayermemKbProj/biases/bias": t(P["memKbProj_b"]),
           Pn + "read/linearLayermemKbProj/linearLayermemKbProj_2/weights/weight": t(P["memKbProj2_W"]),
           Pn + "read/linearLayermemKbProj/linearLayermemKbProj_2/biases/bias": t(P["memKbProj2_b"]),
           Pn + "read/inter2att/inter2logits/linearLayerlogits/weights/weight": t(P["kbLogits_w"]),
           Pn + "read/inter2att/inter2logits/linearLayerlogits/biases/bias": t(P["kbLogits_b"]),
           Pn + "write/linearLayernewMemory/weights/weight": t(P["newMemory_W"]),
           Pn + "write/linearLayernewMemory/biases/bias": t(P["newMemory_b"])}
    for i in range(p):
        prm[Pn + "linearLayerqInput%d/weights/weight" % i] = t(P["qInputU_W"][i])
        prm[Pn + "linearLayerqInput%d/biases/bias" % i] = t(P["qInputU_b"][i])
    vq, words, kb = t([case["vecQ"]]), t([case["words"]]), t([case["kb"]])
    lengths = torch.tensor([case["length"]], dtype=torch.int32)
    vs = mo.VarStore(params=prm, dtype=torch.float64)
    c, m, cell = mo.mac_network(cfg, vs, vq, words, words, lengths, kb)
    exp = case["expected"]
    assert np.abs(cell.memories[0].numpy() - np.array(exp["memories"])).max() < 1e-12
    assert np.abs(cell.controls[0].numpy() - np.array(exp["controls"])).max() < 1e-12
    for i in range(p):
        assert np.abs(cell.attentions["kb"][i][0].numpy() - np.array(exp["att_kb"][i])).max() < 1e-13
        assert np.abs(cell.attentions["question"][i][0].numpy() - np.array(exp["att_q"][i])).max() < 1e-13
    ref = mn.forward(cfg, {k: v.numpy() for k, v in prm.items()}, vq.numpy(), words.numpy(), lengths.numpy(), kb.numpy())
    assert np.abs(ref["memories"][:, 0] - np.array(exp["memories"])).max() < 1e-12


@pytest.mark.parametrize("name", ["args", "args1", "args3", "args4"])
def test_autograd_matches_finite_differences(name):
    """torch-autograd of the op-for-op oracle vs fp64 central differences of the NUMPY restatement."""
    B, S, N, d, p = 2, 4, 5, 4, 2
    cfg = mo.flag_file_config(name, netLength=p, memDim=d, ctrlDim=d, attDim=d)
    vq, words, lengths, kb = mo.synthetic_inputs(B, S, N, d, seed=4, dtype=torch.float64)
    vs = mo.VarStore(generator=torch.Generator().manual_seed(8), dtype=torch.float64, requires_grad=True)
    kb_t = kb.clone().requires_grad_(True)
    c, m, cell = mo.mac_network(cfg, vs, vq, words, words, lengths, kb_t)
    gw = torch.randn(B, d, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    (m * gw).sum().backward()

    def loss_np(prm, kbv):
        out = mn.forward(cfg, prm, vq.numpy(), words.numpy(), lengths.numpy(), kbv)
        return float((out["memory"] * gw.numpy()).sum())

    base = {k: v.detach().numpy().copy() for k, v in vs.params.items()}
    eps = 1e-6
    rng = np.random.RandomState(0)
    for key in base:
        arr = base[key]
        flat_idx = rng.randint(0, arr.size) if arr.size else 0
        pert = {k: v.copy() for k, v in base.items()}
        idx = np.unravel_index(flat_idx, arr.shape) if arr.shape else ()
        pert[key][idx] += eps
        up = loss_np(pert, kb.numpy())
        pert[key][idx] -= 2 * eps
        dn = loss_np(pert, kb.numpy())
        fd = (up - dn) / (2 * eps)
        an = float(vs.params[key].grad[idx])
        assert abs(fd - an) < 1e-6 * max(1.0, abs(an)), (key, fd, an)
    kbn = kb.numpy().copy()
    kbn[1, 2, 3] += eps
    up = loss_np(base, kbn)
    kbn[1, 2, 3] -= 2 * eps
    dn = loss_np(base, kbn)
    assert abs((up - dn) / (2 * eps) - float(kb_t.grad[1, 2, 3])) < 1e-6


def test_invariants():
    cfg, cell, c, m, ref = run_both("args", B=4, S=9, N=12, d=8, p=3)
    lengths = cell.questionLengths
    for i in range(3):
        ak, aq = cell.attentions["kb"][i], cell.attentions["question"][i]
        assert float(ak.min()) >= 0 and float(aq.min()) >= 0
        assert (ak.sum(-1) - 1).abs().max() < 1e-12 and (aq.sum(-1) - 1).abs().max() < 1e-12
        for b in range(4):
            assert float(aq[b, int(lengths[b]):].abs().sum()) == 0.0      # padded words: exactly zero
    assert cell.controls.shape == (4, 4, 8) and cell.memories.shape == (4, 4, 8) and cell.infos.shape == (4, 4, 8)
    # infos is seeded with the initial MEMORY (mac_cell.py:551)
    assert torch.equal(cell.infos[:, 0], cell.memories[:, 0])


def test_eval_mode_dropout_is_identity():
    """keep = 1.0 (model.py:118-125): x / 1.0 * floor(1.0 + U) == x exactly, variational mask included."""
    cfg = mo.flag_file_config("args", netLength=2, memDim=8, ctrlDim=8, attDim=8)
    vq, words, lengths, kb = mo.synthetic_inputs(2, 5, 6, 8, seed=1, dtype=torch.float64)
    vs = mo.VarStore(generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    _, m1, _ = mo.mac_network(cfg, vs, vq, words, words, lengths, kb, train=False)
    ones = lambda site, step, shape: np.ones(shape, dtype=np.float32)
    _, m2, _ = mo.mac_network(cfg, vs, vq, words, words, lengths, kb, train=True, mask_fn=ones, keeps=(1.0, 1.0, 1.0))
    assert torch.equal(m1, m2)


def test_dropout_stream_statistics_and_shard_consistency():
    m = dh.mask_for(1234, dh.SITE_READ_KB, 3, 0.85, (8, 49, 64))
    assert abs(m.mean() - 0.85) < 0.01
    # a data-parallel shard (questions 4..7) sees the masks of the full batch
    shard = dh.mask_for(1234, dh.SITE_READ_KB, 3, 0.85, (4, 49, 64), b0=4)
    assert np.array_equal(shard, m[4:])
    assert not np.array_equal(dh.mask_for(1234, dh.SITE_READ_KB, 4, 0.85, (8, 49, 64)), m)
    assert dh.mask_for(1, dh.SITE_MEM_VAR, 0, 1.0, (3, 5)).min() == 1.0
    # the run's mask word (include/macx.h, macx_dropout.mask_word): XORed into the site key; 0 = the plain stream
    assert np.array_equal(dh.mask_for(1234, dh.SITE_READ_KB, 3, 0.85, (8, 49, 64), word=0), m)
    w = dh.mask_for(1234, dh.SITE_READ_KB, 3, 0.85, (8, 49, 64), word=0xDEADBEEF)
    assert not np.array_equal(w, m) and abs(w.mean() - 0.85) < 0.01
    assert dh.site_key(1234, dh.SITE_READ_KB, 3) ^ 0xDEADBEEF == dh.site_key(1234, dh.SITE_READ_KB, 3) ^ 0xDEADBEEF


@pytest.mark.parametrize("over,exc", [
    (dict(readMemAttType="DIAG"), UnboundLocalError),        # ops.py:704-707
    (dict(initKBwithQ="CNCT"), TypeError),                   # mac_cell.py:564
    (dict(addNullWord=True), UnboundLocalError),             # mac_cell.py:519,573-574
    (dict(relu="LKY"), AttributeError),                      # ops.py:175, config.py:221
    (dict(relu="SELU"), UnboundLocalError),                  # ops.py:171-179
    (dict(readProjInputs=False), UnboundLocalError),         # ops.py:691,716 (readMemConcatProj without proj)
    (dict(writeGate=True, writeGateShared=True), ValueError),
])
def test_options_that_raise_in_the_reference_raise_here(over, exc):
    cfg = mo.flag_file_config("args", netLength=1, memDim=8, ctrlDim=8, attDim=8, **over)
    vq, words, lengths, kb = mo.synthetic_inputs(2, 4, 5, 8, dtype=torch.float64)
    vs = mo.VarStore(generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    with pytest.raises(exc):
        mo.mac_network(cfg, vs, vq, words, words, lengths, kb)


def test_classifier_logits_and_argmax():
    cfg = mo.flag_file_config("args", netLength=1, memDim=8, ctrlDim=8, attDim=8, outClassifierDims=[6], answerWordsNum=5)
    vs = mo.VarStore(generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    mem = torch.randn(3, 8, dtype=torch.float64)
    vq = torch.randn(3, 8, dtype=torch.float64)
    logits = mo.output_classifier(cfg, vs, mem, vq)
    assert logits.shape == (3, 5)
    assert "outputUnit/linearLayeroutQuestion/weights/weight" in vs.params
    assert vs.params["classifier/linearLayerfc_0/weights/weight"].shape == (16, 6)
    loss, pred = mo.answer_loss_and_pred(logits, torch.tensor([0, 1, 2]))
    assert pred.dtype == torch.int32 and loss.ndim == 0


def test_question_encoder_restatement_matches_torch_lstm():
    """The biLSTM restatement (tf.nn.bidirectional_dynamic_rnn + BasicLSTMCell, gate order i, j, f, o, forget bias 1)
    against torch.nn.LSTM over packed ragged sequences -- an independent implementation of the same recurrence."""
    E, h, V = 5, 4, 6
    cfg = mo.flag_file_config("args", ctrlDim=2 * h, memDim=2 * h, attDim=2 * h, encDim=2 * h, wrdEmbDim=E)
    vs = mo.VarStore(generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    q = torch.tensor([[1, 2, 3, 0, 0], [3, 6, 2, 1, 4], [5, 0, 0, 0, 0]])
    L = torch.tensor([3, 5, 1], dtype=torch.int32)
    w, v = mo.question_encoder(cfg, vs, q, L, V)
    P = vs.params
    for d in ("fw", "bw"):
        P["encoder/birnnLayer/bidirectional_rnn/%s/basic_lstm_cell/bias" % d] = torch.rand(4 * h, dtype=torch.float64) - 0.5
    w, v = mo.question_encoder(cfg, vs, q, L, V)
    lstm = torch.nn.LSTM(E, h, batch_first=True, bidirectional=True).double()

    def load(sfx, K, b):
        def reord(M):      # TF i, j, f, o -> torch i, f, g, o
            i, j, f, o = M.split(h, 0)
            return torch.cat([i, f, j, o], 0)
        bb = b.clone()
        bb[2 * h:3 * h] += 1.0
        getattr(lstm, "weight_ih_l0" + sfx).data = reord(K[:E].T.contiguous())
        getattr(lstm, "weight_hh_l0" + sfx).data = reord(K[E:].T.contiguous())
        getattr(lstm, "bias_ih_l0" + sfx).data = reord(bb)
        getattr(lstm, "bias_hh_l0" + sfx).data.zero_()
    for d, sfx in (("fw", ""), ("bw", "_reverse")):
        load(sfx, P["encoder/birnnLayer/bidirectional_rnn/%s/basic_lstm_cell/kernel" % d],
             P["encoder/birnnLayer/bidirectional_rnn/%s/basic_lstm_cell/bias" % d])
    table = torch.cat([torch.zeros(1, E, dtype=torch.float64), P["qEmbeddings/emb"]])
    pk = torch.nn.utils.rnn.pack_padded_sequence(table[q], L.long(), batch_first=True, enforce_sorted=False)
    out, (hn, _) = lstm(pk)
    out, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=5)
    assert (out - w).abs().max() < 1e-12 and (torch.cat([hn[0], hn[1]], 1) - v).abs().max() < 1e-12
    assert float(w[0, 3:].abs().max()) == 0.0 and float(w[2, 1:].abs().max()) == 0.0


def test_prelu_boundary_modes_take_the_two_one_sided_derivatives():
    """relu_boundary(mode): PReLU = relu(x) - alpha relu(-x) takes derivative alpha (mode 0) or 1 (mode 1) at the jump and the
    usual one-sided derivatives away from it; with no boundary mode it is torch's (derivative of relu at 0 = 0 on both terms)."""
    from helpers import relu_boundary
    cfg = mo.default_config(relu="PRM")
    x0 = torch.tensor([[-2.0, 0.0, 1e-9, 3.0]], dtype=torch.float64)
    for mode, at_jump in ((0, 0.25), (1, 1.0)):
        vs = mo.VarStore(generator=torch.Generator().manual_seed(0), dtype=torch.float64)
        ops = mo.Ops(cfg, vs)
        x = x0.clone().requires_grad_(True)
        with relu_boundary(mode, 1e-6):
            y = ops.relu(x)
        y.sum().backward()
        assert torch.allclose(x.grad, torch.tensor([[0.25, at_jump, at_jump, 1.0]], dtype=torch.float64))
        assert torch.allclose(y.detach(), torch.tensor([[-0.5, 0.0, 1e-9, 3.0]], dtype=torch.float64))
