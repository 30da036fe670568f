"""CPU: the HOST LOGIC of the generic option path (mac-network_amd/generic.py) -- scope stacking and variable names, the
chaining of ops per option, and the backward formulas of its autograd nodes -- with the nine kernel-call functions
(generic.k_*) swapped for torch restatements.  The kernels themselves are checked on the GPU (tests/test_gpu_generic.py);
nothing here runs product arithmetic, and the product never takes this route (it refuses CPU tensors)."""
import numpy as np
import pytest
import torch

from oracle import dropout_hash as dh
from oracle import mac_oracle as mo
from helpers import oracle_run, rel_err, max_abs
from test_gpu_generic import VARIANTS, make_cfg, oracle_params, assert_grad, bn_slack


def _torch_kernels(G):
    def binary(op, bmode, a, b, mid, inner, scale=1.0):
        if bmode == G.B_MID:
            aa, bb = a.reshape(-1, mid, inner), b.reshape(-1, 1, inner)
        elif bmode == G.B_CHANNEL:
            aa, bb = a.reshape(-1, inner), b.reshape(1, inner)
        elif bmode == G.B_ROW:
            aa, bb = a.reshape(-1, inner), b.reshape(-1, 1)
        else:
            aa, bb = a, b.reshape(a.shape)
        return (scale * (aa * bb if op == G.OP_MUL else aa + bb)).reshape(a.shape).contiguous()

    def reduce(mode, x, outer, mid, inner):
        if mode == G.R_MID:
            return x.reshape(outer, mid, inner).sum(1)
        return x.reshape(outer, inner).sum(1 if mode == G.R_LAST else 0)

    def act(a, x, alpha):
        if a == G.ACT_PRELU:
            return torch.where(x > 0, x, alpha * x)
        if a == G.ACT_RSQRT_EPS:
            return 1.0 / torch.sqrt(x + alpha[0])
        return {0: lambda v: v, 1: torch.tanh, 2: torch.sigmoid, 3: torch.nn.functional.elu, 4: torch.relu}[a](x)

    def act_bwd(a, x, alpha, g):
        if a == G.ACT_PRELU:
            return g * torch.where(x > 0, torch.ones_like(x), alpha.expand_as(x)), torch.where(x > 0, torch.zeros_like(x), g * x)
        xx = x.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            act(a, xx, alpha).backward(g)
        return xx.grad, None

    def softmax(x, lengths):
        if lengths is not None:
            n = x.shape[-1]
            m = torch.arange(n).unsqueeze(0) < lengths.reshape(-1, 1)
            x = torch.where(m.reshape(x.shape), x, torch.full_like(x, float("-inf")))
        return torch.softmax(x, dim=-1)

    def softmax_bwd(a, g):
        return a * (g - (a * g).sum(-1, keepdim=True))

    def dropout(x, seed, site, step, keep, first, mask_word=None):
        word = int(mask_word.item()) & 0xFFFFFFFF if mask_word is not None else 0
        m = torch.as_tensor(dh.keep_mask(seed, site, step, keep, first, x.numel(), word=word)).reshape(x.shape).to(x.dtype)
        return x * np.float32(1.0 / keep) * m

    def matmul(x, W, b, big):
        return x @ W + (b if b is not None else 0)

    def wgrad(x2, g2):
        return x2.t() @ g2

    def embed(ids, emb, E, ld, keep, seed, first_row):
        table = torch.cat([torch.zeros(1, E), emb], dim=0)
        x = table[ids.long()]
        if keep < 1.0:
            m = torch.as_tensor(dh.keep_mask(seed, 11, 0, keep, first_row * E, x.numel())).reshape(x.shape).to(x.dtype)
            x = x * np.float32(1.0 / keep) * m
        return torch.cat([x, torch.zeros(x.shape[0], ld - E)], dim=1)

    def embed_bwd(ids, dx, E, V, keep, seed, first_row):
        g = dx[:, :E]
        if keep < 1.0:
            m = torch.as_tensor(dh.keep_mask(seed, 11, 0, keep, first_row * E, g.numel())).reshape(g.shape).to(g.dtype)
            g = g * np.float32(1.0 / keep) * m
        out = torch.zeros(V + 1, E)
        out.index_add_(0, ids.long(), g)
        return out[1:]

    return dict(k_binary=binary, k_reduce=reduce, k_act=act, k_act_bwd=act_bwd, k_softmax=softmax, k_softmax_bwd=softmax_bwd,
                k_dropout=dropout, k_matmul=matmul, k_wgrad=wgrad, k_embed=embed, k_embed_bwd=embed_bwd)


@pytest.fixture
def host_generic(macx, monkeypatch):
    G = macx.generic
    for k, f in _torch_kernels(G).items():
        monkeypatch.setattr(G, k, f)
    monkeypatch.setattr(G, "_require_device", lambda t, name: None)
    return G


@pytest.mark.parametrize("variant", sorted(VARIANTS))
@pytest.mark.parametrize("train", [False, True])
def test_generic_host_logic_matches_oracle(macx, host_generic, variant, train):
    B, S, N, d, p = 3, 7, 10, 128, 3
    cfg = make_cfg(variant, d, p)
    vq, words, lengths, kb = mo.synthetic_inputs(B, S, N, d, seed=11)
    params = oracle_params(cfg, vq, words, lengths, kb)
    g = torch.Generator().manual_seed(3)
    dM, dC = torch.randn(B, d, generator=g), torch.randn(B, d, generator=g)
    ref = oracle_run(cfg, params, vq, words, lengths, kb, train=train, seed=91, b0=1, need_grad=True, d_memory=dM, d_control=dC)
    gp = macx.GenericParams().load_reference_dict(params)
    vqd, wd, kbd = [t.clone().requires_grad_(True) for t in (vq, words, kb)]
    cell = macx.GenericMACCell(vecQuestions=vqd, questionWords=wd, questionCntxWords=wd, questionLengths=lengths, knowledgeBase=kbd,
                               memoryDropout=cfg.memoryDropout, readDropout=cfg.readDropout, writeDropout=cfg.writeDropout,
                               batchSize=B, train=train, config=cfg, params=gp, seed=91, b0=1)
    state = cell.run()
    ((state.memory * dM).sum() + (state.control * dC).sum()).backward()
    rc = ref["cell"]
    k = bn_slack(variant)
    assert list(gp.names) == list(params), "variables are created in the reference's order under the reference's names"
    assert rel_err(state.memory, ref["memory"]) < 2e-5 * k and rel_err(state.control, ref["control"]) < 2e-5 * k
    assert rel_err(cell.memories, rc.memories) < 2e-5 * k and rel_err(cell.controls, rc.controls) < 2e-5 * k
    for kind in ("kb", "question", "self", "gate"):
        assert len(cell.attentions[kind]) == len(rc.attentions[kind])
        for a, b in zip(cell.attentions[kind], rc.attentions[kind]):
            assert max_abs(a, b) < 2e-6 * k
    grads = gp.grads_by_name()
    for k, v in ref["params"].items():
        if v.grad is None:
            assert grads[k] is None or float(grads[k].abs().max()) == 0.0, k
        else:
            assert_grad(grads[k], v.grad, k, 2e-4 * bn_slack(variant))
    for got, want in zip((vqd, wd, kbd), ref["inputs"]):
        if want.grad is not None:
            assert rel_err(got.grad, want.grad) < 2e-4 * bn_slack(variant)


def test_generic_path_refuses_cpu_tensors(macx):
    cfg = mo.default_config(netLength=1, memDim=128, ctrlDim=128, attDim=128)
    vq, words, lengths, kb = mo.synthetic_inputs(2, 4, 5, 128)
    with pytest.raises(RuntimeError, match="no CPU path"):
        macx.MACCell(vq, words, words, lengths, kb, 0.85, 0.85, 1.0, 2, True, config=cfg)


def test_macx_maccell_dispatches_unfused_option_sets_to_the_generic_path(macx, host_generic):
    vq, words, lengths, kb = mo.synthetic_inputs(2, 4, 5, 128)
    for variant in ("defaults", "write_sum", "read_bilinear", "relu_prm", "unshared_cells", "control_proj", "write_concat_mul"):
        cell = macx.MACCell(vq, words, words, lengths, kb, 0.85, 0.85, 1.0, 2, False, config=make_cfg(variant, 128, 2))
        assert type(cell) is macx.GenericMACCell, variant


def test_generic_path_rejections(macx, host_generic):
    vq, words, lengths, kb = mo.synthetic_inputs(2, 4, 5, 128)
    mk = lambda **over: macx.GenericMACCell(vq, words, words, lengths, kb, 1.0, 1.0, 1.0, 2, False,
                                     config=mo.flag_file_config("args", netLength=1, memDim=128, ctrlDim=128, attDim=128, **over))
    with pytest.raises(UnboundLocalError):
        mk(readMemAttType="DIAG")
    with pytest.raises(UnboundLocalError):
        mk(relu="SELU", writeMemAct="RELU").run()
    # widths off the kernels' 128-column granule are accepted (zero-padded inside the products, round 4) ...
    macx.GenericMACCell(vq[:, :64], words[:, :, :64], words[:, :, :64], lengths, kb[:, :, :64], 1.0, 1.0, 1.0, 2, False,
                        config=mo.default_config(netLength=1, memDim=64, ctrlDim=64, attDim=64))
    # ... rows that are not whole 16-byte units are not
    with pytest.raises(macx.UnsupportedOptions):
        macx.GenericMACCell(vq[:, :66], words[:, :, :66], words[:, :, :66], lengths, kb[:, :, :66], 1.0, 1.0, 1.0, 2, False,
                            config=mo.default_config(netLength=1, memDim=66, ctrlDim=66, attDim=66))


# ---------------------------------------------------------------------------------------------------------------------
# random option sets: the product's host logic builds and differentiates what the oracle builds, and raises where it raises
# ---------------------------------------------------------------------------------------------------------------------
FUZZ_BOOL = ["readProjInputs", "readProjShared", "readMemConcatKB", "readMemConcatProj", "readMemProj", "readCtrl", "readCtrlConcatKB",
             "readCtrlConcatProj", "readSmryKBProj", "writeConcatMul", "writeInfoProj", "writeSelfAtt", "writeMergeCtrl", "writeMemProj",
             "writeGate", "memoryVariationalDropout", "controlContextual", "controlInWordsProj", "controlOutWordsProj",
             "controlInputUnshared", "controlFeedPrev", "controlFeedPrevAtt", "controlFeedInputs", "controlConcatWords", "controlProj",
             "controlContinuous", "controlWholeQ", "memoryBN", "bnCenter", "bnScale", "unsharedCells"]
FUZZ_CHOICE = {"initCtrl": ["PRM", "ZERO", "Q"], "initMem": ["PRM", "ZERO", "Q"], "controlInputAct": ["NON", "RELU", "TANH"],
               "controlContAct": ["NON", "RELU", "TANH"], "controlProjAct": ["NON", "RELU", "TANH"],
               "readMemAttType": ["MUL", "BL", "ADD"], "readCtrlAttType": ["MUL", "BL", "ADD"], "readMemAct": ["NON", "RELU", "TANH"],
               "readCtrlAct": ["NON", "RELU", "TANH"], "writeInputs": ["MEM", "INFO", "SUM", "BOTH"],
               "writeInfoAct": ["NON", "RELU", "TANH"], "writeSelfAttMod": ["NON", "CONT"], "writeMemAct": ["NON", "RELU", "TANH"],
               "relu": ["STD", "PRM", "ELU"], "mulBias": [0.0, 0.5], "writeGateBias": [0.0, 1.0]}


@pytest.mark.parametrize("seed", range(6))
def test_random_option_sets_host_logic(macx, host_generic, seed):
    run_random_sets(macx, seed)


def run_random_sets(macx, seed, dev=None):
    """dev=None: CPU tensors through the torch kernel stand-ins (host logic); a device: the HIP kernels (tests/test_gpu_generic.py)."""
    import random
    to = (lambda t: t.to(dev)) if dev is not None else (lambda t: t.clone())
    rnd = random.Random(500 + seed)
    B, S, N, d, p = 3, 6, 8, 128, 3
    built = 0
    for case in range(8):
        over = {}
        on = 0.85 if rnd.random() < 0.7 else 0.4
        for b in FUZZ_BOOL:
            if rnd.random() < (on if b in ("readProjInputs", "readMemProj", "readCtrl") else 0.35):
                over[b] = True
        for k, vals in FUZZ_CHOICE.items():
            if rnd.random() < 0.5:
                over[k] = rnd.choice(vals)
        train = rnd.random() < 0.5
        cfg = mo.default_config(netLength=p, memDim=d, ctrlDim=d, attDim=d, **over)
        vq, words, lengths, kb = mo.synthetic_inputs(B, S, N, d, seed=11)
        g = torch.Generator().manual_seed(3)
        dM, dC = torch.randn(B, d, generator=g), torch.randn(B, d, generator=g)
        oexc = pexc = params = None
        try:
            params = oracle_params(cfg, vq, words, lengths, kb)
            ref = oracle_run(cfg, params, vq, words, lengths, kb, train=train, seed=91, b0=1, need_grad=True, d_memory=dM, d_control=dC)
        except Exception as e:          # noqa: BLE001 -- the oracle raises what the reference raises (tests/test_reference_exec.py)
            oexc = e
        try:
            gp = macx.GenericParams(device=dev)
            if params is not None:
                gp.load_reference_dict(params)
            vqd, wd, kbd = [to(t).requires_grad_(True) for t in (vq, words, kb)]
            cell = macx.GenericMACCell(vecQuestions=vqd, questionWords=wd, questionCntxWords=wd, questionLengths=to(lengths),
                                       knowledgeBase=kbd, memoryDropout=cfg.memoryDropout, readDropout=cfg.readDropout,
                                       writeDropout=cfg.writeDropout, batchSize=B, train=train, config=cfg, params=gp, seed=91, b0=1)
            state = cell.run()
            ((state.memory * to(dM)).sum() + (state.control * to(dC)).sum()).backward()
        except Exception as e:          # noqa: BLE001
            pexc = e
        if oexc is not None or pexc is not None:
            assert type(oexc) is type(pexc), "%s: oracle raises %r, product %r" % (over, oexc, pexc)
            continue
        built += 1
        k = 5.0 if cfg.memoryBN else 1.0
        # ReLU / PReLU have a derivative jump at 0: a pre-activation within round-off of zero lands on the other side in fp32
        # than in the fp64 oracle and moves one (row, column) of the read unit's gradients by its whole contribution (seen on
        # the GPU at 1e-3 of the largest entry, identical in two kernel families, absent with ELU).  For such option sets the
        # oracle also runs with derivative 0 and with derivative 1 (PReLU: alpha and 1) wherever |pre-activation| <= 4e-6 of the
        # tensor's largest: the distance between those two gradients is what the undecidable elements can contribute to a
        # tensor, and it is added to that tensor's tolerance (as tests/test_gpu_fuzz.py does for the fused cell); where no
        # pre-activation is that close to the jump the two coincide and the tolerance is the usual one
        lohi = []
        if cfg.relu in ("STD", "PRM"):
            from helpers import relu_boundary
            for mode in (0, 1):
                with relu_boundary(mode, 4e-6):
                    lohi.append(oracle_run(cfg, params, vq, words, lengths, kb, train=train, seed=91, b0=1, need_grad=True, d_memory=dM,
                                           d_control=dC))

        def undecidable(pick):
            if not lohi:
                return 0.0
            a, b = pick(lohi[0]), pick(lohi[1])
            return 0.0 if a is None or b is None else rel_err(a, b)
        # (absolute floor: writeInputs=MEM under batch norm normalises identical rows -- the memory is round-off around zero)
        assert list(gp.names) == list(params), over
        assert rel_err(state.memory, ref["memory"], floor=1e-3) < 2e-5 * k and rel_err(state.control, ref["control"], floor=1e-3) < 2e-5 * k, over
        grads = gp.grads_by_name()
        for name, v in ref["params"].items():
            if v.grad is None:
                assert grads[name] is None or float(grads[name].abs().max()) == 0.0, (over, name)
            elif float(v.grad.abs().max()) > 1e-6:
                assert_grad(grads[name], v.grad, name, 2e-4 * k + undecidable(lambda r, name=name: r["params"][name].grad))
        for j, (got, want) in enumerate(zip((vqd, wd, kbd), ref["inputs"])):
            if want.grad is not None and float(want.grad.abs().max()) > 1e-6:
                assert rel_err(got.grad, want.grad) < 2e-4 * k + undecidable(lambda r, j=j: r["inputs"][j].grad), over
    assert built >= 2


# ---------------------------------------------------------------------------------------------------------------------
# the units on their own (plan.compile_unit behind GenericMACCell.control / read / write), host logic
# ---------------------------------------------------------------------------------------------------------------------
def _unit_cells(macx, cfg, B, S, N, d, train):
    vq, words, lengths, kb = mo.synthetic_inputs(B, S, N, d, seed=13)
    params = oracle_params(cfg, vq, words, lengths, kb)
    keeps = (cfg.memoryDropout, cfg.readDropout, cfg.writeDropout) if train else (1.0, 1.0, 1.0)
    op = {k: v.double().clone().requires_grad_(True) for k, v in params.items()}
    vs = mo.VarStore(params=op, dtype=torch.float64)
    ocell = mo.MACCellOracle(cfg, vs, vq.double(), words.double(), words.double(), lengths, kb.double(), keeps[0], keeps[1], keeps[2], B, train,
                             mask_fn=mo.hash_mask_fn(5, keeps, b0=0) if train else None)
    with vs.scope("MACnetwork"):
        ocell.zero_state(B)
    gp = macx.GenericParams().load_reference_dict(params)
    hcell = macx.GenericMACCell(vq, words, words, lengths, kb, cfg.memoryDropout, cfg.readDropout, cfg.writeDropout, B, train,
                                config=cfg, params=gp, seed=5, b0=0)
    hcell.zero_state(B)
    return ocell, vs, op, hcell, gp, kb, lengths


@pytest.mark.parametrize("train", [False, True])
def test_units_on_their_own_host_logic(macx, host_generic, train):
    B, S, N, d = 3, 6, 12, 128
    rand = lambda seed: torch.randn((B, d), generator=torch.Generator().manual_seed(seed))
    cfg = mo.flag_file_config("args3", netLength=2, memDim=d, ctrlDim=d, attDim=d, controlProj=True, writeGate=True)
    ocell, vs, op, hcell, gp, kb, lengths = _unit_cells(macx, cfg, B, S, N, d, train)
    mem, ctl, cin, info, w = rand(1), rand(2), rand(3), rand(4), rand(5)
    with vs.scope("MACnetwork"), vs.scope("MACCell"):
        ref_c, ref_cc = ocell.control(cin.double(), ocell.inWords, ocell.outWords, lengths, ctl.double(), ctl.double())
        ref_r = ocell.read(kb.double(), mem.double(), ctl.double())
        ref_w = ocell.write(mem.double(), info.double(), ctl.double(), ctl.double())
    ((ref_c + ref_cc + ref_r + ref_w) * w.double()).sum().backward()
    got_c, got_cc = hcell.control(cin, hcell.inWords, hcell.outWords, hcell.questionLengths, ctl, ctl)
    got_r = hcell.read(kb, mem, ctl)
    got_w = hcell.write(mem, info, ctl, ctl)
    ((got_c + got_cc + got_r + got_w) * w).sum().backward()
    for a, b in ((got_c, ref_c), (got_cc, ref_cc), (got_r, ref_r), (got_w, ref_w)):
        assert rel_err(a, b) < 1e-5
    grads = gp.grads_by_name()
    seen = 0
    for k, v in op.items():
        if v.grad is not None:
            assert_grad(grads[k], v.grad, k)
            seen += 1
    assert seen > 10
