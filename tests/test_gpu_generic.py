"""-m gpu: the generic option path (mac-network_amd/generic.py: one HIP kernel per reference op) against the fp64 oracle,
for legal option sets the fused cell kernels do not cover -- the reference's default configuration first.
The oracle itself is pinned to the reference's code for every one of these option sets by tests/test_reference_exec.py
(VARIANTS there; same names here)."""
import pytest
import torch

from oracle import mac_oracle as mo
from helpers import oracle_run, rel_err, max_abs

pytestmark = pytest.mark.gpu

FWD_TOL = 2e-5
GRAD_TOL = 2e-4

# name -> (base config, overrides); "default" = config.py's defaults, "args" = configs/args.txt
VARIANTS = {
    "defaults": ("default", {}),
    "defaults_std_relu": ("default", dict(relu="STD")),
    "read_noproj_concat": ("default", dict(readMemConcatKB=True, readMemProj=True, readCtrl=True, relu="ELU")),
    "read_proj_shared": ("args", dict(readProjShared=True)),
    "read_bilinear": ("args", dict(readMemAttType="BL", readCtrlAttType="BL")),
    "read_additive": ("args", dict(readMemAttType="ADD", readCtrlAttType="ADD")),
    "read_ctrl_concat_kb": ("args", dict(readCtrlConcatKB=True)),
    "read_ctrl_concat_proj": ("args", dict(readCtrlConcatKB=True, readCtrlConcatProj=True)),
    "read_smry_proj": ("args", dict(readSmryKBProj=True)),
    "read_acts": ("args", dict(readMemAct="TANH", readCtrlAct="TANH", relu="STD")),
    "read_memact_non": ("args", dict(readMemAct="NON", readCtrlAct="NON")),
    "mul_bias": ("args", dict(mulBias=0.5)),
    "write_mem": ("args", dict(writeInputs="MEM")),
    "write_info": ("args", dict(writeInputs="INFO", writeInfoProj=True, writeInfoAct="TANH")),
    "write_sum": ("args", dict(writeInputs="SUM", writeGate=True)),
    "write_concat_mul": ("args", dict(writeConcatMul=True, writeMemAct="RELU")),
    "write_merge_ctrl": ("args", dict(writeMergeCtrl=True, writeSelfAtt=True, writeSelfAttMod="NON")),
    "control_proj": ("args", dict(controlProj=True, controlProjAct="TANH", controlConcatWords=True)),
    "control_concat_words": ("args", dict(controlConcatWords=True)),
    "control_words_proj": ("args", dict(controlInWordsProj=True)),
    "control_words_proj_out": ("args", dict(controlOutWordsProj=True)),
    "control_continuous": ("args", dict(controlContinuous=True, controlFeedPrev=True, controlContAct="RELU")),
    "control_whole_q": ("args", dict(controlWholeQ=True)),
    "unshared_cells": ("args", dict(unsharedCells=True)),
    "relu_prm": ("args", dict(relu="PRM", initCtrl="Q", controlContAct="RELU", controlFeedPrev=True)),
    "relu_prm_unshared": ("args", dict(relu="PRM", initCtrl="Q", unsharedCells=True, writeMemAct="RELU")),
    "relu_prm_two_in_one_scope": ("args", dict(relu="PRM", writeInfoAct="RELU", writeMemAct="RELU")),     # write/prelu, write/prelu_1
    "no_var_dropout": ("args", dict(memoryVariationalDropout=False)),
    "memory_bn": ("args", dict(memoryBN=True)),
    "memory_bn_affine": ("args", dict(memoryBN=True, bnCenter=True, bnScale=True)),
}


def bn_slack(variant):
    """Batch norm over a batch of 3 divides by a small standard deviation: fp32 round-off is amplified ~5x."""
    return 5.0 if variant.startswith("memory_bn") else 1.0


def make_cfg(variant, d, p):
    base, over = VARIANTS[variant]
    kw = dict(netLength=p, memDim=d, ctrlDim=d, attDim=d)
    kw.update(over)
    return mo.default_config(**kw) if base == "default" else mo.flag_file_config("args", **kw)


def oracle_params(cfg, vq, words, lengths, kb, seed=5):
    """The variables the oracle creates for this option set (reference names), biases / PReLU slopes perturbed."""
    vs = mo.VarStore(generator=torch.Generator().manual_seed(seed))
    mo.mac_network(cfg, vs, vq, words, words, lengths, kb)
    g = torch.Generator().manual_seed(seed + 1)
    for k, v in vs.params.items():
        if "/biases/" in k or k.endswith("alpha") or k.endswith("BatchNorm/beta") or k.endswith("BatchNorm/gamma"):
            v.add_((torch.rand(v.shape, generator=g) - 0.5) * 0.2)
    return {k: v.clone() for k, v in vs.params.items()}


def assert_grad(got, want, name, tol=GRAD_TOL):
    if float(want.abs().max()) < 1e-9:          # analytically zero (the bias in front of a softmax): absolute
        assert float(got.abs().max()) < 2e-5 * (tol / GRAD_TOL), name
    else:
        assert rel_err(got.reshape(want.shape), want) < tol, name


def run_generic(macx, dev, cfg, params, vq, words, lengths, kb, train, seed, b0, grad):
    gp = macx.GenericParams(device=dev).load_reference_dict(params)
    vqd, wd, kbd = [t.to(dev).requires_grad_(grad) for t in (vq, words, kb)]
    # (built directly: a few of the option sets below also have fused kernels, and macx.MACCell would pick those)
    cell = macx.GenericMACCell(vecQuestions=vqd, questionWords=wd, questionCntxWords=wd, questionLengths=lengths.to(dev),
                               knowledgeBase=kbd, memoryDropout=cfg.memoryDropout, readDropout=cfg.readDropout,
                               writeDropout=cfg.writeDropout, batchSize=vq.shape[0], train=train, config=cfg, params=gp,
                               seed=seed, b0=b0)
    return cell, gp, (vqd, wd, kbd)


@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("train", [False, True])
def test_generic_path_matches_oracle(macx, dev, variant, train):
    B, S, N, d, p = 3, 7, 20, 128, 3
    cfg = make_cfg(variant, d, p)
    vq, words, lengths, kb = mo.synthetic_inputs(B, S, N, d, seed=11)
    params = oracle_params(cfg, vq, words, lengths, kb)
    g = torch.Generator().manual_seed(3)
    dM, dC = torch.randn(B, d, generator=g), torch.randn(B, d, generator=g)
    ref = oracle_run(cfg, params, vq, words, lengths, kb, train=train, seed=91, b0=1, need_grad=True, d_memory=dM, d_control=dC)
    cell, gp, (vqd, wd, kbd) = run_generic(macx, dev, cfg, params, vq, words, lengths, kb, train, 91, 1, True)
    state = cell.zero_state(B)
    for i in range(p):
        cell.iteration = i
        _, state = cell(cell.none, state)
    loss = (state.memory * dM.to(dev)).sum() + (state.control * dC.to(dev)).sum()
    loss.backward()
    torch.cuda.synchronize()
    rc = ref["cell"]
    slack = bn_slack(variant)
    assert set(gp.names) == set(params), "variable names differ from the reference's"
    assert rel_err(state.memory, ref["memory"]) < FWD_TOL * slack
    assert rel_err(state.control, ref["control"]) < FWD_TOL * slack
    assert rel_err(cell.memories, rc.memories) < FWD_TOL * slack and rel_err(cell.infos, rc.infos) < FWD_TOL * slack
    for kind in ("kb", "question", "self", "gate"):
        assert len(cell.attentions[kind]) == len(rc.attentions[kind])
        for a, b in zip(cell.attentions[kind], rc.attentions[kind]):
            assert max_abs(a, b) < 2e-6 * slack
    grads = gp.grads_by_name()
    for k, v in ref["params"].items():
        if v.grad is None:
            assert grads[k] is None or float(grads[k].abs().max()) == 0.0, k
            continue
        assert grads[k] is not None, k
        assert_grad(grads[k], v.grad, k, GRAD_TOL * slack)
    for name, got, want in zip(("vecQuestions", "words", "knowledgeBase"), (vqd, wd, kbd), ref["inputs"]):
        if want.grad is None:
            continue
        assert rel_err(got.grad, want.grad) < GRAD_TOL * slack, name


@pytest.mark.parametrize("variant,d,train", [("defaults", 144, True), ("read_bilinear", 72, True), ("write_sum", 200, False),
                                             ("control_proj", 100, True)])
def test_generic_path_at_widths_off_the_128_granule(macx, dev, variant, d, train):
    """config.py:294-296 takes any width: on the one-kernel-per-op path the products zero-pad their operands to the kernels'
    128-column granule (generic.k_matmul / k_wgrad) and every other op works at the logical width -- states, attentions and all
    gradients against the fp64 oracle; macx.MACCell dispatches such a configuration here."""
    B, S, N, p = 3, 7, 20, 2
    cfg = make_cfg(variant, d, p)
    vq, words, lengths, kb = mo.synthetic_inputs(B, S, N, d, seed=11)
    params = oracle_params(cfg, vq, words, lengths, kb)
    dM = torch.randn(B, d, generator=torch.Generator().manual_seed(3))
    ref = oracle_run(cfg, params, vq, words, lengths, kb, train=train, seed=91, b0=1, need_grad=True, d_memory=dM)
    gp = macx.GenericParams(device=dev).load_reference_dict(params)
    vqd, wd, kbd = [t.to(dev).requires_grad_(True) for t in (vq, words, kb)]
    cell = macx.MACCell(vqd, wd, wd, lengths.to(dev), kbd, cfg.memoryDropout, cfg.readDropout, cfg.writeDropout, B, train,
                        config=cfg, params=gp, seed=91, b0=1)
    assert getattr(cell, "generic", False), "a width off the granule with a non-fused option set belongs on the generic path"
    state = cell.zero_state(B)
    for i in range(p):
        cell.iteration = i
        _, state = cell(cell.none, state)
    (state.memory * dM.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert state.memory.shape == (B, d)
    assert rel_err(state.memory, ref["memory"]) < FWD_TOL and rel_err(state.control, ref["control"]) < FWD_TOL
    for a, b in zip(cell.attentions["kb"], ref["cell"].attentions["kb"]):
        assert max_abs(a, b) < 2e-6
    grads = gp.grads_by_name()
    for k, v in ref["params"].items():
        if v.grad is not None:
            assert grads[k] is not None, k
            assert_grad(grads[k], v.grad, k)
    for name, got, want in zip(("vecQuestions", "words", "knowledgeBase"), (vqd, wd, kbd), ref["inputs"]):
        if want.grad is not None:
            assert rel_err(got.grad, want.grad) < GRAD_TOL, name


def test_generic_path_under_a_mask_word(macx, dev):
    """macx_dropout.mask_word on the one-kernel-per-op path (macx_op_dropout_w): the masks of (seed, word), forward and backward"""
    B, S, N, d, p = 3, 7, 20, 128, 2
    cfg = make_cfg("no_var_dropout", d, p)          # plain memory dropout: a site of its own per step
    vq, words, lengths, kb = mo.synthetic_inputs(B, S, N, d, seed=11)
    params = oracle_params(cfg, vq, words, lengths, kb)
    dM = torch.randn(B, d, generator=torch.Generator().manual_seed(3))
    word = 0x1234ABCD
    ref = oracle_run(cfg, params, vq, words, lengths, kb, train=True, seed=91, b0=1, need_grad=True, d_memory=dM, word=word)
    gp = macx.GenericParams(device=dev).load_reference_dict(params)
    vqd, wd, kbd = [t.to(dev).requires_grad_(True) for t in (vq, words, kb)]
    cell = macx.GenericMACCell(vqd, wd, wd, lengths.to(dev), kbd, cfg.memoryDropout, cfg.readDropout, cfg.writeDropout, B, True,
                               config=cfg, params=gp, seed=91, b0=1, mask_word=torch.tensor([word], dtype=torch.int32, device=dev))
    state = cell.zero_state(B)
    for i in range(p):
        cell.iteration = i
        _, state = cell(cell.none, state)
    (state.memory * dM.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert rel_err(state.memory, ref["memory"]) < FWD_TOL
    assert rel_err(kbd.grad, ref["inputs"][2].grad) < GRAD_TOL
    plain = oracle_run(cfg, params, vq, words, lengths, kb, train=True, seed=91, b0=1)
    assert rel_err(state.memory, plain["memory"]) > 1e-3          # not the plain seed's masks


def test_dispatch_between_the_fused_and_the_generic_path(macx, dev):
    vq, words, lengths, kb = [t.to(dev) for t in mo.synthetic_inputs(2, 5, 14, 128)]
    mk = lambda cfg: macx.MACCell(vq, words, words, lengths, kb, 0.85, 0.85, 1.0, 2, False, config=cfg)
    assert type(mk(mo.flag_file_config("args", netLength=2, memDim=128, ctrlDim=128, attDim=128))) is macx.MACCell
    for variant in ("defaults", "write_sum", "read_bilinear", "relu_prm", "unshared_cells", "control_proj", "write_concat_mul"):
        assert type(mk(make_cfg(variant, 128, 2))) is macx.GenericMACCell, variant


def test_generic_path_shards_like_the_full_batch(macx, dev):
    """Data-parallel contract on the generic path: a shard run with b0 sees the masks of the full batch."""
    B, S, N, d, p = 4, 6, 14, 128, 2
    cfg = make_cfg("defaults", d, p)
    vq, words, lengths, kb = mo.synthetic_inputs(B, S, N, d, seed=2)
    params = oracle_params(cfg, vq, words, lengths, kb)
    full, _, _ = run_generic(macx, dev, cfg, params, vq, words, lengths, kb, True, 7, 0, False)
    with torch.no_grad():
        mf = full.run().memory
        part, _, _ = run_generic(macx, dev, cfg, params, vq[2:], words[2:], lengths[2:], kb[2:], True, 7, 2, False)
        mp = part.run().memory
    assert torch.equal(mf[2:], mp)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("MACX_FUZZ_SEEDS", "6"))))
def test_random_option_sets_on_the_gpu(macx, dev, seed):
    """48 random option combinations (the generator of tests/test_generic_host.py): where the oracle builds, the HIP kernels
    of the generic path give its state (<= 2e-5) and every gradient (<= 2e-4); where it raises, the product raises the same class."""
    from test_generic_host import run_random_sets
    run_random_sets(macx, seed, dev=dev)
    torch.cuda.synchronize()
