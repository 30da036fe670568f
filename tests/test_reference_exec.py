"""CPU: the parity anchor.  The reference's own mac_cell.py / ops.py / model.py / config.py are executed
UNMODIFIED (tests/ref_exec.py over the eager TF-1.x stand-in of tests/tf1_shim) and oracle/mac_oracle.py --
the checker every GPU parity test uses -- must reproduce them: same parsed flag values, same variable names and
shapes, same states / attentions / logits / loss / predictions to 1e-12 in fp64, and the same gradients, for all
five published flag files (configs/args*.txt) in evaluation and in training mode (identical dropout draws), plus
the legal option values outside the flag files and the option values that raise.

/root/reference only exists in the build container; on the GPU box these tests skip and
tests/test_reference_golden.py checks the oracle against the vectors this machinery generated
(tests/golden/reference_*.npz, tests/golden/make_reference_golden.py).
"""
import numpy as np
import os

import pytest
import sys
import torch

import ref_exec as rx
from oracle import mac_oracle as mo

pytestmark = pytest.mark.skipif(not rx.available(), reason="/root/reference is not present on this machine")

FLAG_FILES = ["args", "args1", "args2", "args3", "args4"]
B, S, N, D, P, HID, ANS = 3, 6, 10, 16, 3, 8, 7


def oracle_config_from(cfg):
    """An oracle config carrying exactly what the reference's parser produced."""
    snap = rx.snapshot(cfg)
    return mo.default_config(answerWordsNum=ANS, **snap)


def inputs(seed=11):
    vq, words, lengths, kb = mo.synthetic_inputs(B, S, N, D, seed=seed, dtype=torch.float64)
    g = torch.Generator().manual_seed(seed + 1)
    raw = (torch.rand((B, S, D), generator=g, dtype=torch.float64) * 2 - 1)     # questionWords != questionCntxWords
    answers = torch.randint(0, ANS, (B,), generator=g)
    return vq, raw, words, lengths, kb, answers


def run_pair(cfg, train, need_grad=False, seed=5, keeps=None, output_keep=None):
    """reference run, then the oracle on the reference's variables and draws"""
    vq, raw, words, lengths, kb, answers = inputs()
    if keeps is None:
        keeps = (cfg.memoryDropout, cfg.readDropout, 0.9 if train else cfg.writeDropout) if train else (1.0, 1.0, 1.0)
    if train:
        cfg.writeDropout = keeps[2]        # the write dropout op exists only if config.writeDropout < 1 (mac_cell.py:461)
    if output_keep is None:
        output_keep = cfg.outputDropout if train else 1.0
    ref = rx.run_reference(cfg, vq, raw, words, lengths, kb, train=train, keeps=keeps, output_keep=output_keep,
                           seed=seed, need_grad=need_grad, answers=answers, answerWordsNum=ANS)
    ocfg = oracle_config_from(cfg)
    params = {k: v.detach().clone().requires_grad_(need_grad) for k, v in ref["variables"].items()}
    vs = mo.VarStore(params=params, dtype=torch.float64)
    draws = ref["draws"]
    if cfg.memoryVariationalDropout and keeps[0] == 1.0:
        # ops.generateVarDpMask draws even when keep == 1 (ops.py:1054-1059): floor(1 + U) is all ones, the oracle skips it
        assert bool((torch.floor(1.0 + draws[0]) == 1).all())
        draws = draws[1:]
    mask_fn = rx.replay_mask_fn(draws, keeps)
    ins = [t.clone().requires_grad_(need_grad) for t in (vq, raw, words, kb)]
    c, m, cell = mo.mac_network(ocfg, vs, ins[0], ins[1], ins[2], lengths, ins[3], train=train, mask_fn=mask_fn, keeps=keeps)
    masks = mask_fn.rest(output_keep) if output_keep != 1.0 else None
    logits = mo.output_classifier(ocfg, vs, m, ins[0], output_keep=output_keep, masks=masks)
    assert mask_fn.left() == 0, "the reference drew more random tensors than the oracle consumed"
    loss, preds = mo.answer_loss_and_pred(logits, answers)
    orc = dict(control=c, memory=m, cell=cell, logits=logits, loss=loss, preds=preds, params=params, inputs=ins, store=vs)
    return ref, orc, answers


def assert_same_forward(ref, orc, tol=1e-12):
    def close(a, b, what):
        a, b = torch.as_tensor(a).detach().double(), torch.as_tensor(b).detach().double()
        assert a.shape == b.shape, "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
        err = float((a - b).abs().max()) if a.numel() else 0.0
        assert err <= tol, "%s differs from the reference by %.3e" % (what, err)

    rc, oc = ref["cell"], orc["cell"]
    close(orc["control"], ref["control"], "final control")
    close(orc["memory"], ref["memory"], "final memory")
    close(oc.controls, rc.controls, "controls history")
    close(oc.memories, rc.memories, "memories history")
    close(oc.infos, rc.infos, "infos history")
    for key in ("kb", "question", "self", "gate"):
        assert len(oc.attentions[key]) == len(rc.attentions[key]), "attentions[%s] length" % key
        for i, (a, b) in enumerate(zip(oc.attentions[key], rc.attentions[key])):
            close(a, b, "attentions[%s][%d]" % (key, i))
    close(orc["logits"], ref["logits"], "logits")
    close(orc["loss"], ref["loss"], "loss")
    # (rows whose two largest logits tie to round-off -- e.g. a memory that batch norm turned into zeros -- have no defined argmax)
    top2 = torch.topk(torch.as_tensor(ref["logits"]).detach().double(), 2, dim=-1).values
    decided = (top2[:, 0] - top2[:, 1]) > 1e-9
    assert torch.equal(orc["preds"].long()[decided], torch.as_tensor(ref["preds"]).long()[decided]), "predicted answers"


# ---------------------------------------------------------------------------------------------------------------
# the flag tables
# ---------------------------------------------------------------------------------------------------------------
def test_default_flag_values_match_the_reference_parser():
    """oracle.default_config(), macx.options.DEFAULTS and macx.configs mirror config.py's defaults."""
    cfg = rx.parse_flags(None)
    snap = rx.snapshot(cfg)
    oc = mo.default_config()
    for k, v in snap.items():
        assert getattr(oc, k) == v, "oracle default of --%s is %r, the reference parses %r" % (k, getattr(oc, k), v)
    import macx
    for k, v in macx.options.DEFAULTS.items():
        assert getattr(cfg, k) == v, "options.DEFAULTS[%s] = %r, the reference parses %r" % (k, v, getattr(cfg, k))


@pytest.mark.parametrize("name", FLAG_FILES)
def test_flag_files_parse_to_the_oracle_tables(name):
    cfg = rx.parse_flags(name + ".txt")
    snap = rx.snapshot(cfg)
    oc = mo.flag_file_config(name)
    for k, v in snap.items():
        assert getattr(oc, k) == v, "%s: oracle has --%s = %r, the reference parses %r" % (name, k, getattr(oc, k), v)
    import macx
    pc = macx.configs.flag_file_config(name)
    for k, v in snap.items():
        if hasattr(pc, k):
            assert getattr(pc, k) == v, "%s: macx.configs has --%s = %r, the reference parses %r" % (name, k, getattr(pc, k), v)


# ---------------------------------------------------------------------------------------------------------------
# the published configurations
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", FLAG_FILES)
@pytest.mark.parametrize("train", [False, True])
def test_oracle_reproduces_the_reference(name, train):
    cfg = rx.parse_flags(name + ".txt", *rx.dims_flags(D, P, HID))
    ref, orc, _ = run_pair(cfg, train)
    assert_same_forward(ref, orc)


@pytest.mark.parametrize("name", FLAG_FILES)
def test_oracle_creates_the_reference_variables(name):
    """Same variable names, shapes and creation order when the oracle builds its own parameters."""
    cfg = rx.parse_flags(name + ".txt", *rx.dims_flags(D, P, HID))
    vq, raw, words, lengths, kb, answers = inputs()
    ref = rx.run_reference(cfg, vq, raw, words, lengths, kb, answerWordsNum=ANS)
    vs = mo.VarStore(generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    ocfg = oracle_config_from(cfg)
    _, m, _ = mo.mac_network(ocfg, vs, vq, raw, words, lengths, kb)
    mo.output_classifier(ocfg, vs, m, vq)
    assert list(vs.params.keys()) == list(ref["variables"].keys())
    for k, v in ref["variables"].items():
        assert tuple(vs.params[k].shape) == tuple(v.shape), k


@pytest.mark.parametrize("name", FLAG_FILES)
@pytest.mark.parametrize("train", [False, True])
def test_oracle_gradients_match_the_reference_graph(name, train):
    """d loss / d (every variable, vecQuestions, words, knowledge base) through the reference's op graph."""
    cfg = rx.parse_flags(name + ".txt", *rx.dims_flags(D, P, HID))
    ref, orc, answers = run_pair(cfg, train, need_grad=True)
    g = torch.Generator().manual_seed(3)
    dc = torch.randn((B, D), generator=g, dtype=torch.float64)
    (ref["loss"] + (ref["control"] * dc).sum()).backward()
    (orc["loss"] + (orc["control"] * dc).sum()).backward()
    for k, v in ref["variables"].items():
        a, b = orc["params"][k].grad, v.grad
        assert (a is None) == (b is None), k
        if b is not None:
            assert float((a - b).abs().max()) <= 1e-12 * max(1.0, float(b.abs().max())), k
    names = ["vecQ", "questionWords", "questionCntxWords", "kb"]
    for nme, t in zip(names, orc["inputs"]):
        b = ref["inputs"][nme].grad
        a = t.grad
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0, nme
        else:
            assert float((a - torch.as_tensor(b)).abs().max()) <= 1e-12 * max(1.0, float(b.abs().max())), nme


def test_fp32_reference_execution_tracks_fp64():
    """The same reference code in fp32 (what TF computes in) stays within fp32 round-off of the fp64 run."""
    cfg = rx.parse_flags("args.txt", *rx.dims_flags(D, 4, HID))
    vq, raw, words, lengths, kb, answers = inputs()
    r64 = rx.run_reference(cfg, vq, raw, words, lengths, kb, answerWordsNum=ANS)
    r32 = rx.run_reference(cfg, vq, raw, words, lengths, kb, answerWordsNum=ANS, dtype=torch.float32,
                           preset={k: v.detach() for k, v in r64["variables"].items()})
    assert float((r32["logits"].double() - r64["logits"]).abs().max()) < 2e-5


# ---------------------------------------------------------------------------------------------------------------
# legal option values outside the published flag files
# ---------------------------------------------------------------------------------------------------------------
BASE = ["--relu", "ELU", "--initCtrl", "Q"]
FULLREAD = ["--readProjInputs", "--readMemConcatKB", "--readMemConcatProj", "--readMemProj", "--readCtrl",
            "--writeMemProj", "--memoryVariationalDropout", "--controlContextual", "--controlInputUnshared"]
VARIANTS = {
    "defaults": [],
    "defaults_std_relu": ["--relu", "STD"],
    "read_noproj_concat": ["--readMemConcatKB", "--readMemProj", "--readCtrl"] + BASE,
    "read_proj_shared": FULLREAD + ["--readProjShared"] + BASE,
    "read_bilinear": FULLREAD + ["--readMemAttType", "BL", "--readCtrlAttType", "BL"] + BASE,
    "read_additive": FULLREAD + ["--readMemAttType", "ADD", "--readCtrlAttType", "ADD"] + BASE,
    "read_ctrl_concat_proj": FULLREAD + ["--readCtrlConcatKB", "--readCtrlConcatProj"] + BASE,
    "read_ctrl_concat_kb": FULLREAD + ["--readCtrlConcatKB"] + BASE,
    "read_smry_proj": FULLREAD + ["--readSmryKBProj"] + BASE,
    "read_acts": FULLREAD + ["--readMemAct", "TANH", "--readCtrlAct", "TANH", "--relu", "STD"],
    "read_memact_non": FULLREAD + ["--readMemAct", "NON", "--readCtrlAct", "NON"] + BASE,
    "mul_bias": FULLREAD + ["--mulBias", "0.5"] + BASE,
    "write_mem": FULLREAD + ["--writeInputs", "MEM"] + BASE,
    "write_info": FULLREAD + ["--writeInputs", "INFO", "--writeInfoProj", "--writeInfoAct", "TANH"] + BASE,
    "write_sum": FULLREAD + ["--writeInputs", "SUM", "--writeGate"] + BASE,
    "write_concat_mul": FULLREAD + ["--writeConcatMul", "--writeMemAct", "RELU"] + BASE,
    "write_merge_ctrl": FULLREAD + ["--writeMergeCtrl", "--writeSelfAtt", "--writeSelfAttMod", "NON"] + BASE,
    "write_gate_bias": FULLREAD + ["--writeGate", "--writeGateBias", "-0.5"] + BASE,
    "write_noproj": ["--readProjInputs", "--readMemProj", "--readCtrl", "--writeInputs", "SUM"] + BASE,
    "control_proj": FULLREAD + ["--controlProj", "--controlProjAct", "TANH", "--controlConcatWords"] + BASE,
    "control_concat_words": FULLREAD + ["--controlConcatWords"] + BASE,
    "control_words_proj": FULLREAD + ["--controlInWordsProj"] + BASE,
    "control_words_proj_out": FULLREAD + ["--controlOutWordsProj"] + BASE,
    "control_continuous": FULLREAD + ["--controlContinuous", "--controlFeedPrev", "--controlContAct", "RELU"] + BASE,
    "control_whole_q": FULLREAD + ["--controlWholeQ"] + BASE,
    "control_feed_cont": FULLREAD + ["--controlFeedPrev", "--controlFeedInputs", "--initCtrl", "ZERO", "--relu", "ELU"],
    "control_raw_words": [f for f in FULLREAD if f != "--controlContextual"] + BASE,
    "control_shared_input": [f for f in FULLREAD if f != "--controlInputUnshared"] + BASE,
    "unshared_cells": FULLREAD + ["--unsharedCells", "1"] + BASE,
    "relu_prm": FULLREAD + ["--relu", "PRM", "--initCtrl", "Q", "--controlContAct", "RELU", "--controlFeedPrev"],
    "relu_prm_unshared": FULLREAD + ["--relu", "PRM", "--initCtrl", "Q", "--unsharedCells", "1", "--writeMemAct", "RELU"],
    "relu_prm_two_in_one_scope": FULLREAD + ["--relu", "PRM", "--writeInfoAct", "RELU", "--writeMemAct", "RELU"],
    "init_zero_mem": FULLREAD + ["--initMem", "ZERO", "--initCtrl", "PRM", "--relu", "ELU"],
    "init_q_mem": FULLREAD + ["--initMem", "Q"] + BASE,
    "no_var_dropout": [f for f in FULLREAD if f != "--memoryVariationalDropout"] + BASE,
    "memory_bn": FULLREAD + ["--memoryBN"] + BASE,
    "memory_bn_affine": FULLREAD + ["--memoryBN", "--bnCenter", "--bnScale"] + BASE,
    "out_question_mul": FULLREAD + ["--outQuestion", "--outQuestionMul"] + BASE,
    "out_no_question": FULLREAD + BASE,
    "out_deep_classifier": FULLREAD + ["--outQuestion", "--outQuestionMul", "--outClassifierDims", "12", "10", "9", "--relu", "PRM"],
    "out_no_hidden_layer": FULLREAD + ["--outQuestion", "--outClassifierDims"] + BASE,
}


def variant_flags(variant):
    flags = VARIANTS[variant]
    return flags + rx.dims_flags(D, P, None if "--outClassifierDims" in flags else HID)


@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("train", [False, True])
def test_oracle_reproduces_the_reference_on_other_legal_options(variant, train):
    cfg = rx.parse_flags(None, *variant_flags(variant))
    ref, orc, _ = run_pair(cfg, train)
    assert_same_forward(ref, orc)
    assert list(orc["store"].params.keys()) == list(ref["variables"].keys())


@pytest.mark.parametrize("variant", ["defaults", "read_bilinear", "read_additive", "write_concat_mul", "write_sum",
                                     "control_proj", "unshared_cells", "relu_prm", "memory_bn", "read_ctrl_concat_proj",
                                     "write_merge_ctrl", "control_continuous", "mul_bias", "relu_prm_two_in_one_scope",
                                     "out_deep_classifier"])
def test_oracle_gradients_on_other_legal_options(variant):
    cfg = rx.parse_flags(None, *variant_flags(variant))
    ref, orc, _ = run_pair(cfg, True, need_grad=True)
    ref["loss"].backward()
    orc["loss"].backward()
    for k, v in ref["variables"].items():
        a, b = orc["params"][k].grad, v.grad
        assert (a is None) == (b is None), k
        if b is not None:
            assert float((a - b).abs().max()) <= 1e-12 * max(1.0, float(b.abs().max())), k


# ---------------------------------------------------------------------------------------------------------------
# option values that raise in the reference: the oracle and the product raise the same class
# ---------------------------------------------------------------------------------------------------------------
RAISING = {
    "diag_mem": FULLREAD + ["--readMemAttType", "DIAG"] + BASE,
    "diag_ctrl": FULLREAD + ["--readCtrlAttType", "DIAG"] + BASE,
    "concat_proj_without_proj": ["--readMemConcatKB", "--readMemConcatProj", "--readMemProj"] + BASE,
    "init_kb_with_q": FULLREAD + ["--initKBwithQ", "CNCT"] + BASE,
    "init_kb_with_q_mul": FULLREAD + ["--initKBwithQ", "MUL"] + BASE,
    "add_null_word": FULLREAD + ["--addNullWord"] + BASE,
    "relu_lky": FULLREAD + ["--relu", "LKY", "--initCtrl", "Q"],
    "relu_selu": FULLREAD + ["--relu", "SELU", "--initCtrl", "Q"],
    "gate_shared": FULLREAD + ["--writeGate", "--writeGateShared"] + BASE,
    "att_dim_mismatch": FULLREAD + BASE + ["--attDim", "8"],
    # mac_cell.py:248-250 keeps `dim` when ops.mul concatenates its x operand: inter2att then gets a [.., 2 dim] tensor
    "read_ctrl_concat_inter": FULLREAD + ["--readCtrlConcatInter"] + BASE,
    "read_ctrl_concat_inter_kb": FULLREAD + ["--readCtrlConcatInter", "--readCtrlConcatKB"] + BASE,
    # found by the random option sets below: tf.concat / tensor * None reject None with ValueError (ops.convert_to_tensor) ...
    "ctrl_concat_proj_without_proj": ["--readMemProj", "--readCtrl", "--readCtrlConcatKB", "--readCtrlConcatProj"] + BASE,
    "smry_proj_without_proj": ["--readMemProj", "--readCtrl", "--readSmryKBProj"] + BASE,
    "concat_proj_without_proj_no_x": ["--readMemConcatProj", "--readMemProj"] + BASE,
    # ... and PReLU on the control input lives in the cell's own scope (reuse=None): step 1 creates its slope again
    "prelu_on_control_input": FULLREAD + ["--relu", "PRM", "--controlInputAct", "RELU", "--initCtrl", "Q"],
}


@pytest.mark.parametrize("variant", sorted(RAISING))
def test_rejected_option_values_raise_the_same_exception(variant):
    extra = RAISING[variant]
    dims = rx.dims_flags(D, P, HID)
    cfg = rx.parse_flags(None, *(dims + extra))         # `extra` last: it may override a dimension
    vq, raw, words, lengths, kb, answers = inputs()
    with pytest.raises(Exception) as ref_exc:
        rx.run_reference(cfg, vq, raw, words, lengths, kb, answerWordsNum=ANS)
    ocfg = oracle_config_from(cfg)
    vs = mo.VarStore(generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    with pytest.raises(Exception) as orc_exc:
        mo.mac_network(ocfg, vs, vq, raw, words, lengths, kb)
    assert orc_exc.type is ref_exc.type, "reference raises %r, oracle raises %r" % (ref_exc.value, orc_exc.value)
    import macx
    with pytest.raises(Exception) as prod_exc:
        macx.options.reject_like_reference(ocfg)
        macx.options.resolve_activations(ocfg)
    assert prod_exc.type is ref_exc.type, "reference raises %r, macx.options raises %r" % (ref_exc.value, prod_exc.value)


# ---------------------------------------------------------------------------------------------------------------
# random option sets: whatever the reference does with a combination -- build and run, or raise -- the oracle does too
# ---------------------------------------------------------------------------------------------------------------
FUZZ_BOOL = ["readProjInputs", "readProjShared", "readMemConcatKB", "readMemConcatProj", "readMemProj", "readCtrl", "readCtrlConcatKB",
             "readCtrlConcatProj", "readSmryKBProj", "writeConcatMul", "writeInfoProj", "writeSelfAtt", "writeMergeCtrl", "writeMemProj",
             "writeGate", "memoryVariationalDropout", "controlContextual", "controlInWordsProj", "controlOutWordsProj",
             "controlInputUnshared", "controlFeedPrev", "controlFeedPrevAtt", "controlFeedInputs", "controlConcatWords", "controlProj",
             "controlContinuous", "controlWholeQ", "memoryBN", "bnCenter", "bnScale", "outQuestion", "outQuestionMul"]
FUZZ_CHOICE = {"initCtrl": ["PRM", "ZERO", "Q"], "initMem": ["PRM", "ZERO", "Q"], "controlInputAct": ["NON", "RELU", "TANH"],
               "controlContAct": ["NON", "RELU", "TANH"], "controlProjAct": ["NON", "RELU", "TANH"],
               "readMemAttType": ["MUL", "BL", "ADD"], "readCtrlAttType": ["MUL", "BL", "ADD"], "readMemAct": ["NON", "RELU", "TANH"],
               "readCtrlAct": ["NON", "RELU", "TANH"], "writeInputs": ["MEM", "INFO", "SUM", "BOTH"],
               "writeInfoAct": ["NON", "RELU", "TANH"], "writeSelfAttMod": ["NON", "CONT"], "writeMemAct": ["NON", "RELU", "TANH"],
               "relu": ["STD", "PRM", "ELU"], "unsharedCells": [None, "1"], "mulBias": ["0", "0.5"], "writeGateBias": ["0", "1.0"],
               "outClassifierDims": [["8"], ["8", "6"], []], "attDim": [None, None, "32"]}          # attDim != memDim = 16
FUZZ_RARE = {"readMemAttType": "DIAG", "readCtrlAttType": "DIAG", "relu": "LKY", "initKBwithQ": "CNCT"}      # values that raise


def random_flags(rnd):
    flags = []
    on = 0.85 if rnd.random() < 0.6 else 0.4             # mostly the projecting read structure, so that most graphs build
    for b in FUZZ_BOOL:
        pr = on if b in ("readProjInputs", "readMemProj", "readCtrl") else 0.4
        if rnd.random() < pr:
            flags.append("--" + b)
    for k, vals in FUZZ_CHOICE.items():
        if rnd.random() < 0.5:
            v = rnd.choice(vals)
            if v is not None:
                flags += ["--" + k] + (v if isinstance(v, list) else [v])
    for k, v in FUZZ_RARE.items():
        if rnd.random() < 0.04:
            flags += ["--" + k, v]
    for b in ("readCtrlConcatInter", "writeGateShared", "addNullWord"):
        if rnd.random() < 0.04:
            flags.append("--" + b)
    return flags


@pytest.mark.parametrize("seed", range(8))
def test_random_option_sets_match_the_reference(seed):
    import random
    rnd = random.Random(1000 + seed)
    built = 0
    for case in range(int(os.environ.get("MACX_FUZZ_CASES", "14"))):
        flags = random_flags(rnd)
        train = rnd.random() < 0.5
        cfg = rx.parse_flags(None, *(rx.dims_flags(D, P, None if "--outClassifierDims" in flags else HID) + flags))
        vq, raw, words, lengths, kb, answers = inputs()
        ref_exc = orc_exc = None
        try:
            rx.run_reference(cfg, vq, raw, words, lengths, kb, answerWordsNum=ANS)
        except Exception as e:          # noqa: BLE001 -- whatever the reference raises is the specification
            ref_exc = e
        if ref_exc is None:
            ref, orc, _ = run_pair(cfg, train)
            assert_same_forward(ref, orc, tol=1e-10)      # (batch norm over 3 questions amplifies fp64 round-off past 1e-12)
            assert list(orc["store"].params.keys()) == list(ref["variables"].keys()), flags
            built += 1
            continue
        ocfg = oracle_config_from(cfg)
        vs = mo.VarStore(generator=torch.Generator().manual_seed(1), dtype=torch.float64)
        try:
            c, m, cell = mo.mac_network(ocfg, vs, vq, raw, words, lengths, kb)
            mo.output_classifier(ocfg, vs, m, vq)
        except Exception as e:          # noqa: BLE001
            orc_exc = e
        assert type(orc_exc) is type(ref_exc), "%s: reference raises %r, oracle %r" % (" ".join(flags), ref_exc, orc_exc)
    assert built >= 1


# ---------------------------------------------------------------------------------------------------------------
# the stem: the reference's own MACnet.stem / ops.CNNLayer / ops.cnn against the oracle's stem_cnn
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("relu", ["ELU", "STD"])
def test_oracle_stem_reproduces_the_reference(train, relu):
    Bs, H, W, C, stem_dim, out_dim = 2, 4, 3, 10, 12, D
    cfg = rx.parse_flags(None, *(rx.dims_flags(D, P, HID) + ["--stemDim", str(stem_dim), "--relu", relu]))
    g = torch.Generator().manual_seed(5)
    images = torch.randn(Bs, H, W, C, generator=g, dtype=torch.float64)
    keep = cfg.stemDropout if train else 1.0
    assert 0.0 < cfg.stemDropout < 1.0
    ref = rx.run_reference_stem(cfg, images, keep=keep, need_grad=True)
    assert list(ref["variables"]) == ["stem/cnnLayercnn_0/kernels/kernel", "stem/cnnLayercnn_0/biases/bias",
                                      "stem/cnnLayercnn_1/kernels/kernel", "stem/cnnLayercnn_1/biases/bias"]
    assert tuple(ref["variables"]["stem/cnnLayercnn_0/kernels/kernel"].shape) == (3, 3, C, stem_dim)          # HWIO
    assert tuple(ref["kb"].shape) == (Bs, H * W, out_dim)
    ocfg = mo.default_config(memDim=out_dim, relu=relu)
    ocfg.stemDim = stem_dim
    params = {k: v.detach().clone().requires_grad_(True) for k, v in ref["variables"].items()}
    vs = mo.VarStore(params=params, dtype=torch.float64)
    masks = None
    if train:       # tf.nn.dropout draws once per layer input, in layer order
        assert [tuple(u.shape) for u in ref["draws"]] == [(Bs, H, W, C), (Bs, H, W, stem_dim)]
        masks = [torch.floor(keep + u).reshape(Bs, H * W, -1) for u in ref["draws"]]
    img = images.clone().requires_grad_(True)
    kb = mo.stem_cnn(ocfg, vs, img.reshape(Bs, H * W, C), H, W, keep=keep, masks=masks)
    assert float((kb - ref["kb"]).abs().max()) <= 1e-12
    w = torch.randn(kb.shape, generator=g, dtype=torch.float64)
    (kb * w).sum().backward()
    (ref["kb"] * w).sum().backward()
    assert float((img.grad - ref["images"].grad).abs().max()) <= 1e-12
    for k, v in ref["variables"].items():
        assert float((params[k].grad - v.grad).abs().max()) <= 1e-12 * max(1.0, float(v.grad.abs().max())), k


# ---------------------------------------------------------------------------------------------------------------
# the question encoder: the reference's qEmbeddingsOp + encoder / ops.RNNLayer / ops.biRNNLayer -- scopes,
# variable names, the input dropout, which final state is kept, the concat order -- against the oracle's question_encoder
# (the cell and the dynamic-rnn loop are the shim's restatement of TF's; the oracle's LSTM is also pinned to torch.nn.LSTM)
# ---------------------------------------------------------------------------------------------------------------
ENC_VARIANTS = {
    "bi": ["--encBi"],                                                   # the published flag files
    "uni": [],                                                           # the parser's default: one forward LSTM(encDim)
    "bi_proj": ["--encBi", "--encProj", "--encProjQAct", "TANH"],        # projCW / projQ (+ its second layer)
    "uni_other_width": ["--ctrlDim", "12"],                              # encDim != ctrlDim projects as well (model.py:785)
}


def test_two_encoder_layers_raise_in_the_reference_and_the_oracle():
    cfg = rx.parse_flags(None, *(rx.dims_flags(D, P, HID) + ["--encDim", "8", "--wrdEmbDim", "5", "--encBi", "--encNumLayers", "2"]))
    q = torch.tensor([[1, 2, 0], [3, 0, 0]])
    lengths = torch.tensor([2, 1], dtype=torch.int32)
    emb = torch.randn(4, 5, dtype=torch.float64)
    with pytest.raises(ValueError, match="already exists"):
        rx.run_reference_encoder(cfg, q, lengths, emb)
    ocfg = mo.default_config(encDim=8, wrdEmbDim=5, encBi=True, encNumLayers=2)
    with pytest.raises(ValueError, match="already exists"):
        mo.question_encoder(ocfg, mo.VarStore(generator=torch.Generator().manual_seed(0), dtype=torch.float64), q, lengths, 4)


@pytest.mark.parametrize("variant", sorted(ENC_VARIANTS))
@pytest.mark.parametrize("train", [False, True])
def test_oracle_encoder_reproduces_the_reference(train, variant):
    Bq, Sq, vocab, E, enc = 4, 6, 9, 5, 8
    cfg = rx.parse_flags(None, *(rx.dims_flags(D, P, HID) + ["--encDim", str(enc), "--wrdEmbDim", str(E)] + ENC_VARIANTS[variant]))
    g = torch.Generator().manual_seed(3)
    lengths = torch.tensor([6, 1, 4, 3], dtype=torch.int32)
    questions = torch.randint(1, vocab + 1, (Bq, Sq), generator=g)
    questions = questions * (torch.arange(Sq).unsqueeze(0) < lengths.unsqueeze(1))          # 0 = pad
    emb = torch.randn(vocab, E, generator=g, dtype=torch.float64)
    ki, kq = (cfg.encInputDropout, cfg.qDropout) if train else (1.0, 1.0)
    ref = rx.run_reference_encoder(cfg, questions, lengths, emb, keep_input=ki, keep_question=kq, need_grad=True)
    bi = "--encBi" in ENC_VARIANTS[variant]
    cell_vars = (["encoder/birnnLayer/bidirectional_rnn/%s/basic_lstm_cell/%s" % (d_, v_) for d_ in ("fw", "bw") for v_ in ("kernel", "bias")]
                 if bi else ["encoder/rnnLayer/rnn/basic_lstm_cell/kernel", "encoder/rnnLayer/rnn/basic_lstm_cell/bias"])
    assert list(ref["variables"])[:1 + len(cell_vars)] == ["qEmbeddings/emb"] + cell_vars
    ocfg = mo.default_config(encDim=enc, wrdEmbDim=E, encBi=bi, encProj=cfg.encProj, encProjQAct=cfg.encProjQAct, ctrlDim=cfg.ctrlDim)
    params = {k: v.detach().clone().requires_grad_(True) for k, v in ref["variables"].items()}
    vs = mo.VarStore(params=params, dtype=torch.float64)
    masks = None
    if train:
        assert [tuple(u.shape) for u in ref["draws"]] == [(Bq, Sq, E), (Bq, enc)]
        masks = [torch.floor(ki + ref["draws"][0]), torch.floor(kq + ref["draws"][1])]
    words, vecQ = mo.question_encoder(ocfg, vs, questions, lengths, vocab, keep_input=ki, keep_question=kq, masks=masks)
    assert float((words - ref["words"]).abs().max()) <= 1e-12 and float((vecQ - ref["vecQ"]).abs().max()) <= 1e-12
    w1 = torch.randn(words.shape, generator=g, dtype=torch.float64)
    w2 = torch.randn(vecQ.shape, generator=g, dtype=torch.float64)
    ((words * w1).sum() + (vecQ * w2).sum()).backward()
    ((ref["words"] * w1).sum() + (ref["vecQ"] * w2).sum()).backward()
    for k, v in ref["variables"].items():
        assert float((params[k].grad - v.grad).abs().max()) <= 1e-12 * max(1.0, float(v.grad.abs().max())), k


@pytest.mark.parametrize("seed", range(4))
def test_stem_and_encoder_on_random_shapes_match_the_reference(seed):
    """Random small shapes (one image, 1 x 1 grids, one-word questions, one-dimensional embeddings, either direction mode):
    the oracle's stem and encoder against the reference's on the stand-in, values and every gradient."""
    import random
    rnd = random.Random(300 + seed)
    for case in range(5):
        # ---- stem
        Bs, H, W, C, sd, od = rnd.randint(1, 3), rnd.randint(1, 5), rnd.randint(1, 5), rnd.randint(1, 9), rnd.randint(1, 9), rnd.randint(1, 9)
        train = rnd.random() < 0.5
        cfg = rx.parse_flags(None, *(rx.dims_flags(od, P, HID) + ["--stemDim", str(sd)]))
        g = torch.Generator().manual_seed(seed * 10 + case)
        images = torch.randn(Bs, H, W, C, generator=g, dtype=torch.float64)
        keep = cfg.stemDropout if train else 1.0
        ref = rx.run_reference_stem(cfg, images, keep=keep, need_grad=True)
        ocfg = mo.default_config(memDim=od)
        ocfg.stemDim = sd
        params = {k: v.detach().clone().requires_grad_(True) for k, v in ref["variables"].items()}
        masks = [torch.floor(keep + u).reshape(Bs, H * W, -1) for u in ref["draws"]] if train else None
        kb = mo.stem_cnn(ocfg, mo.VarStore(params=params, dtype=torch.float64), images.reshape(Bs, H * W, C), H, W, keep=keep, masks=masks)
        assert float((kb - ref["kb"]).abs().max()) <= 1e-11, ("stem", Bs, H, W, C, sd, od, train)
        w = torch.randn(kb.shape, generator=g, dtype=torch.float64)
        (kb * w).sum().backward()
        (ref["kb"] * w).sum().backward()
        for k, v in ref["variables"].items():
            assert float((params[k].grad - v.grad).abs().max()) <= 1e-11 * max(1.0, float(v.grad.abs().max())), ("stem", k)
        # ---- encoder
        Bq, Sq, vocab, E, bi = rnd.randint(1, 4), rnd.randint(1, 6), rnd.randint(1, 7), rnd.randint(1, 6), rnd.random() < 0.5
        enc = 2 * rnd.randint(1, 4)
        ctrl = enc if rnd.random() < 0.6 else enc + 2
        train = rnd.random() < 0.5
        flags = ["--encDim", str(enc), "--wrdEmbDim", str(E), "--ctrlDim", str(ctrl)] + (["--encBi"] if bi else [])
        cfg = rx.parse_flags(None, *(rx.dims_flags(D, P, HID) + flags))
        lengths = torch.tensor([rnd.randint(1, Sq) for _ in range(Bq)], dtype=torch.int32)
        qs = torch.tensor([[rnd.randint(1, vocab) if t < int(lengths[b]) else 0 for t in range(Sq)] for b in range(Bq)])
        emb = torch.randn(vocab, E, generator=g, dtype=torch.float64)
        ki, kq = (cfg.encInputDropout, cfg.qDropout) if train else (1.0, 1.0)
        ref = rx.run_reference_encoder(cfg, qs, lengths, emb, keep_input=ki, keep_question=kq, need_grad=True)
        ocfg = mo.default_config(encDim=enc, wrdEmbDim=E, encBi=bi, ctrlDim=ctrl)
        params = {k: v.detach().clone().requires_grad_(True) for k, v in ref["variables"].items()}
        masks = [torch.floor(ki + ref["draws"][0]), torch.floor(kq + ref["draws"][1])] if train else None
        vs = mo.VarStore(params=params, dtype=torch.float64)
        words, vecQ = mo.question_encoder(ocfg, vs, qs, lengths, vocab, keep_input=ki, keep_question=kq, masks=masks)
        what = ("encoder", Bq, Sq, vocab, E, enc, ctrl, bi, train)
        assert list(vs.params) == list(ref["variables"]), what
        assert float((words - ref["words"]).abs().max()) <= 1e-11 and float((vecQ - ref["vecQ"]).abs().max()) <= 1e-11, what
        w1 = torch.randn(words.shape, generator=g, dtype=torch.float64)
        w2 = torch.randn(vecQ.shape, generator=g, dtype=torch.float64)
        ((words * w1).sum() + (vecQ * w2).sum()).backward()
        ((ref["words"] * w1).sum() + (ref["vecQ"] * w2).sum()).backward()
        for k, v in ref["variables"].items():
            if v.grad is not None:
                assert float((params[k].grad - v.grad).abs().max()) <= 1e-11 * max(1.0, float(v.grad.abs().max())), (what, k)


def test_training_op_matches_the_optimizer_oracle_live():
    """MACnet.addOptimizerOp / computeGradients / addTrainingOp (model.py:615-669) executed unmodified for five steps; the
    committed fixture is exactly this run, and the optimizer oracle follows it step by step."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_training_golden as G
    from helpers import load_training_fixture, flat64
    from oracle import optim_oracle as oo
    cfg, r = G.run()
    hyper, names, init, steps = load_training_fixture()
    assert list(r["steps"][0]["variables"]) == names and r["global_step"] == hyper["steps"]
    assert r["ema_names"] == sorted(n + "/ExponentialMovingAverage" for n in names)       # emaDict (model.py:665)
    tf = rx.load()["tf"]
    p = flat64({n: tf.state.initial[n].numpy() for n in names}, names)
    assert np.abs(p - flat64(init, names)).max() == 0
    m, v, e = p * 0, p * 0, p.copy()
    for t, (live, fx) in enumerate(zip(r["steps"], steps), start=1):
        g = flat64({n: live["grads"][n].numpy() for n in names}, names)
        assert np.abs(g - flat64(fx["g"], names)).max() < 1e-13
        p, m, v, e, norm = oo.adam_ema_step(p, g, m, v, e, hyper["lr"], t, clip=hyper["clip"], decay=hyper["decay"])
        assert abs(norm - live["norm"]) < 1e-12
        for got, key in ((p, "variables"), (m, "m"), (v, "v"), (e, "ema")):
            want = flat64({n: live[key][n].numpy() for n in names}, names)
            assert np.abs(got - want).max() < 1e-12, (t, key)
