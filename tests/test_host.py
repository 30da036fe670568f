"""CPU: host logic -- option freezing, parameter naming, the C-ABI library loads and exports every
symbol include/macx.h declares (no compute calls: there is no GPU here)."""
import ctypes as C
import os
import re

import pytest
import torch

from oracle import mac_oracle as mo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_the_header(macx):
    L = macx._lib.lib()
    assert L.macx_abi_version() == 5
    header = open(os.path.join(ROOT, "include", "macx.h")).read()
    declared = set(re.findall(r"\b(macx_[a-z_0-9]+)\s*\(", header))
    declared -= {"macx_opts", "macx_shapes"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(L, name), "libmacx.so does not export %s" % name
    assert set(macx._lib.EXPORTS) == declared


def test_struct_layouts_match_header(macx):
    assert C.sizeof(macx._lib.MacxOpts) == (20 + 16) * 4          # + the per-call tuning table (ABI 5)
    assert C.sizeof(macx._lib.MacxShapes) == 7 * 4
    assert C.sizeof(macx._lib.MacxDropout) == 4 * 4 + 8         # + mask_word (device pointer)
    assert C.sizeof(macx._lib.MacxParams) == 30 * 8 == C.sizeof(macx._lib.MacxParamGrads)
    header = open(os.path.join(ROOT, "include", "macx.h")).read()
    body = header[header.index("typedef struct macx_params"): header.index("} macx_params;")]
    fields = re.findall(r"const float\*\s+(\w+);", body)
    assert tuple(fields) == macx._lib.PARAM_FIELDS


def test_check_and_sizing_without_gpu(macx):
    L = macx._lib.lib()
    o = macx.freeze(mo.flag_file_config("args"))
    s = macx._lib.MacxShapes(B=64, S=50, N=196, d=512, p=12, b0=0)
    assert L.macx_check(C.byref(o), C.byref(s)) == 0
    # the kernel family is a field of the options (per call, not process state): sizing under "split" without touching the default
    o1 = macx.freeze(mo.flag_file_config("args"), gemm="split")
    assert L.macx_gemm_mode(-1) == 2
    keep = L.macx_saved_floats(C.byref(o1), C.byref(s), 1)
    nokeep = L.macx_saved_floats(C.byref(o1), C.byref(s), 0)
    # X, H1, I2, dropped KB (fp32) + the two 1-bit dropout masks, kept for 11 more steps (no per-question exponent arrays any
    # more: the consumers take the minima from the tensors' exponent bytes, round 4)
    dd_ = 512 * 512
    packs_ = 4 * (dd_ * 3 // 2) + dd_ + 2 * dd_ + dd_ + 12 * dd_ + 2 * dd_ + 3 * dd_
    assert keep - nokeep == 11 * 64 * 196 * 512 * 4 + 11 * 2 * (64 * 196 * 512 // 32) + packs_
    assert L.macx_gemm_mode(-1) == 2
    # the default family keeps the same tensors as H2: 4 bytes per element as well (+ exponents and 64 pad rows each)
    keep = L.macx_saved_floats(C.byref(o), C.byref(s), 1)
    nokeep = L.macx_saved_floats(C.byref(o), C.byref(s), 0)
    h2 = L.macx_h2_floats(64 * 196, 512)
    assert 64 * 196 * 512 < h2 < 1.01 * 64 * 196 * 512
    # (+ round 5: the backward pass's transposed weight packs, written by the forward pack launch of a run that keeps its activations)
    dd = 512 * 512
    packs = 4 * (dd * 3 // 2) + dd + 2 * dd + dd + 12 * dd + 2 * dd + 3 * dd
    assert 11 * 4 * h2 + packs < keep - nokeep < 11 * 4 * h2 + packs + 11 * 2 * (64 * 196 + 64) * 64 // 4 + 64
    off, cnt = C.c_size_t(), C.c_size_t()
    assert L.macx_saved_segment(C.byref(o), C.byref(s), 1, macx._lib.SEG["att_kb"], C.byref(off), C.byref(cnt)) == 0
    assert cnt.value == 12 * 64 * 196
    bad = macx._lib.MacxShapes(B=64, S=50, N=196, d=500, p=12, b0=0)
    assert L.macx_check(C.byref(o), C.byref(bad)) == macx._lib.MACX_EINVAL
    assert L.macx_saved_floats(C.byref(o), C.byref(bad), 1) == 0
    assert b"invalid" in L.macx_strerror(-1)


def test_args1_freezes_recurrent_control(macx):
    o = macx.freeze(mo.flag_file_config("args1"))
    assert o.control_feed_prev == 1 and o.control_feed_prev_att == 1 and o.control_feed_inputs == 1
    assert o.control_cont_act == macx._lib.ACT["TANH"] and o.init_ctrl == macx._lib.INIT["PRM"]


@pytest.mark.parametrize("name", ["args", "args2", "args3", "args4"])
def test_supported_flag_files_freeze(macx, name):
    o = macx.freeze(mo.flag_file_config(name))
    assert o.control_input_unshared == 1 and o.init_ctrl == macx._lib.INIT["Q"] and o.init_mem == macx._lib.INIT["PRM"]
    assert o.read_mem_act == macx._lib.ACT["ELU"] and o.memory_variational_dropout == 1


@pytest.mark.parametrize("over,exc", [
    (dict(readMemAttType="DIAG"), UnboundLocalError), (dict(initKBwithQ="MUL"), TypeError),
    (dict(addNullWord=True), UnboundLocalError), (dict(relu="LKY"), AttributeError), (dict(relu="SELU"), UnboundLocalError),
    (dict(readProjInputs=False), UnboundLocalError), (dict(writeGate=True, writeGateShared=True), ValueError),
])
def test_freeze_rejects_like_the_reference(macx, over, exc):
    with pytest.raises(exc):
        macx.freeze(mo.flag_file_config("args", **over))


@pytest.mark.parametrize("over", [dict(unsharedCells=True), dict(controlProj=True), dict(writeInputs="SUM"),
                                  dict(relu="PRM"), dict(readMemAct="NON"), dict(mulBias=0.5), dict(memoryBN=True)])
def test_unsupported_combinations_fail_loudly(macx, over):
    with pytest.raises(macx.UnsupportedOptions):
        macx.freeze(mo.flag_file_config("args", **over))


def test_parameter_names_are_the_reference_variable_names(macx):
    cfg = mo.flag_file_config("args", netLength=3, memDim=128, ctrlDim=128, attDim=128)
    prm = macx.MACCellParams(cfg, 3, generator=torch.Generator().manual_seed(0))
    ref = prm.to_reference_dict()
    vs = mo.VarStore(generator=torch.Generator().manual_seed(0))
    vq, words, lengths, kb = mo.synthetic_inputs(2, 4, 5, 128)
    mo.mac_network(cfg, vs, vq, words, words, lengths, kb)
    assert set(ref) == set(vs.params)
    for k in ref:
        assert tuple(ref[k].shape) == tuple(vs.params[k].shape), k
    prm2 = macx.MACCellParams(cfg, 3).load_reference_dict(vs.params)
    assert torch.equal(prm2.qInputU_W[2], vs.params["MACnetwork/MACCell/linearLayerqInput2/weights/weight"])
    # xavier limits (ops.py:20): sqrt(6/(in+out)); 1-D: sqrt(3/n)
    assert float(prm.memKbProj_W.abs().max()) <= (6.0 / (256 + 128)) ** 0.5
    assert float(prm.kbLogits_w.abs().max()) <= (3.0 / 128) ** 0.5


def test_cell_refuses_cpu_tensors(macx):
    cfg = mo.flag_file_config("args", netLength=1, memDim=128, ctrlDim=128, attDim=128)
    vq, words, lengths, kb = mo.synthetic_inputs(2, 4, 5, 128)
    with pytest.raises(RuntimeError, match="no CPU path"):
        macx.MACCell(vq, words, words, lengths, kb, 0.85, 0.85, 1.0, 2, True, config=cfg,
                     params=macx.MACCellParams(cfg, 1))


def test_checkpoint_roundtrip_reference_names(tmp_path):
    """Weights leave and enter under the TF variable names (macModel/...:0), incl. EMA shadows."""
    import macx
    from oracle import mac_oracle as mo
    cfg = mo.flag_file_config("args", netLength=2, memDim=128, ctrlDim=128, attDim=128, encDim=256, wrdEmbDim=12, outClassifierDims=[16])
    cfg.ctrlDim = cfg.memDim = cfg.attDim = cfg.encDim = 256
    cfg.stemDim = 128
    net = macx.MACNet(cfg, vocab=7, H=3, W=2, imageInDim=128, answerWordsNum=5, generator=torch.Generator().manual_seed(0))
    sd = macx.checkpoint.reference_state_dict(net)
    assert "macModel/qEmbeddings/emb:0" in sd and "macModel/stem/cnnLayercnn_0/kernels/kernel:0" in sd
    assert "macModel/MACnetwork/MACCell/read/linearLayermemKbProj/linearLayermemKbProj_2/weights/weight:0" in sd
    assert "macModel/encoder/birnnLayer/bidirectional_rnn/bw/basic_lstm_cell/kernel:0" in sd
    assert sum(v.numel() for v in sd.values()) == sum(t.numel() for t in net.tensors())
    ema = [t.detach() * 0.5 for t in net.tensors()]
    path = str(tmp_path / "w.npz")
    macx.checkpoint.save_npz(path, net, ema_tensors=ema)
    before = [t.detach().clone() for t in net.tensors()]
    with torch.no_grad():
        for t in net.tensors():
            t.add_(1.0)
    macx.checkpoint.load_npz(path, net)
    assert all(torch.equal(a, b) for a, b in zip(before, net.tensors()))
    macx.checkpoint.load_npz(path, net, use_ema=True)
    assert all(torch.equal(a * 0.5, b) for a, b in zip(before, net.tensors()))
    # bare names (no macModel/ prefix, no :0), missing and mis-shaped variables
    bare = {k[len("macModel/"):-2]: v for k, v in sd.items()}
    macx.checkpoint.load_reference(net, bare)
    short = dict(bare)
    short.pop("qEmbeddings/emb")
    with pytest.raises(KeyError):
        macx.checkpoint.load_reference(net, short)
    assert macx.checkpoint.load_reference(net, short, strict=False) == ["qEmbeddings/emb"]
    bad = dict(bare)
    bad["qEmbeddings/emb"] = torch.zeros(3, 3)
    with pytest.raises(ValueError):
        macx.checkpoint.load_reference(net, bad)


def test_checkpoint_reaches_a_lazily_built_cell(tmp_path):
    """The reference's DEFAULT configuration runs on the generic path, whose cell variables only appear on first use: a
    checkpoint loaded before the first forward pass must still land in the cell (it used to be skipped silently, and the first
    forward then drew random weights), and a strict load from a source without cell variables must fail."""
    import macx
    from oracle import mac_oracle as mo
    dcfg = mo.default_config(netLength=2, memDim=128, ctrlDim=128, attDim=128)
    vq, words, lengths, kb = mo.synthetic_inputs(2, 5, 6, 128, seed=0)
    vs = mo.VarStore(generator=torch.Generator().manual_seed(3))
    mo.mac_network(dcfg, vs, vq, words, words, lengths, kb)                 # creates the cell's variables under their reference names
    src = {"macModel/" + k + ":0": v for k, v in vs.params.items()}
    assert any(k.startswith("macModel/MACnetwork/") for k in src)
    net = macx.MACNetCore(dcfg, H=3, W=2, imageInDim=128, answerWordsNum=5, generator=torch.Generator().manual_seed(0))
    # the plan is compiled at construction: the cell holds exactly the variables the reference's graph creates, in its order
    assert isinstance(net.cell, macx.GenericParams)
    assert list(net.cell.to_reference_dict()) == list(vs.params)
    full = dict(macx.checkpoint.reference_state_dict(net))                  # stem + classifier of this net ...
    full.update(src)                                                        # ... + the cell of the "checkpoint"
    full["macModel/MACnetwork/initMem/Adam:0"] = torch.zeros(128)           # optimizer slots are not model variables
    macx.checkpoint.load_reference(net, full)
    got = net.cell.to_reference_dict()
    assert set(got) == set(vs.params) and all(torch.equal(got[k], vs.params[k].to(torch.float32)) for k in got)
    # the variables it now holds round-trip through a file
    path = str(tmp_path / "lazy.npz")
    macx.checkpoint.save_npz(path, net)
    net2 = macx.MACNetCore(dcfg, H=3, W=2, imageInDim=128, answerWordsNum=5, generator=torch.Generator().manual_seed(1))
    assert macx.checkpoint.load_npz(path, net2) == []
    assert all(torch.equal(a, b) for a, b in zip(net.cell.tensors(), net2.cell.tensors()))
    # strict: a source that lacks ONE variable the plan needs is an error (it used to be drawn at random on the first forward) ...
    net3 = macx.MACNetCore(dcfg, H=3, W=2, imageInDim=128, answerWordsNum=5, generator=torch.Generator().manual_seed(2))
    one_short = dict(full)
    dropped = "macModel/" + list(vs.params)[-1] + ":0"
    del one_short[dropped]
    with pytest.raises(KeyError, match="missing variables"):
        macx.checkpoint.load_reference(net3, one_short)
    assert macx.checkpoint.load_reference(net3, one_short, strict=False) == [dropped[len("macModel/"):-2]]
    # ... and a variable of a DIFFERENT option set stored under the cell's scope is ignored: it does not become a Parameter
    extra = dict(full)
    extra["macModel/MACnetwork/MACCell/write/linearLayergate/weights/weight:0"] = torch.zeros(128, 128)
    net4 = macx.MACNetCore(dcfg, H=3, W=2, imageInDim=128, answerWordsNum=5, generator=torch.Generator().manual_seed(2))
    macx.checkpoint.load_reference(net4, extra)
    assert list(net4.cell.to_reference_dict()) == list(vs.params)


def test_flat_optimizer_shares_the_gradient_buffer_layout():
    """FlatAdamEMA packs its parameters like MACCellParams.grad_buffer() / dp.GradBucket.flat (segments padded to 4 floats;
    the scalar logits biases make tight packing differ), and rejects a flat gradient of any other size."""
    import macx
    cfg = macx.configs.flag_file_config("args", netLength=2, memDim=128, ctrlDim=128, attDim=128)
    prm = macx.MACCellParams(cfg, 2)
    sizes = [t.numel() for t in prm.tensors()]
    assert any(n % 4 for n in sizes)
    padded = sum((n + 3) & ~3 for n in sizes)
    assert prm.grad_buffer().numel() == padded and padded != sum(sizes)
    bucket = macx.dp.GradBucket(prm.tensors(), params=prm)
    assert bucket.flat.data_ptr() == prm.grad_buffer().data_ptr()
    offs, off = [], 0
    for n in sizes:
        offs.append(off)
        off += (n + 3) & ~3
    assert bucket.offsets == offs
    # one backward pass per step may take the persistent buffer; a second one before the release gets its own
    assert prm.claim_grad_buffer() is not None and prm.claim_grad_buffer() is None
    prm.release_grad_buffer()
    assert prm.claim_grad_buffer() is not None
    # without a registered consumer nobody is handed the persistent buffer
    assert macx.MACCellParams(cfg, 2).claim_grad_buffer() is None


def test_product_configs_match_the_oracle_copies():
    """macx.configs (what bench.py / smoke() build workloads from) and the oracle's own copies describe the same flag files
    and draw the same synthetic inputs -- the measured path never imports oracle/."""
    import macx
    from oracle import mac_oracle as mo
    for name in ("args", "args1", "args2", "args3", "args4"):
        a = vars(macx.configs.flag_file_config(name, netLength=5))
        b = vars(mo.flag_file_config(name, netLength=5))
        assert a == b, name
    for x, y in zip(macx.configs.synthetic_inputs(3, 7, 5, 8, seed=4), mo.synthetic_inputs(3, 7, 5, 8, seed=4)):
        assert torch.equal(x, y)
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    body = src[src.index("def main():"):]
    assert "oracle" not in body.replace("cpu_baseline", "")     # only the cpu_baseline leg touches the oracle


def test_mix32_is_the_dropout_streams_finaliser(macx):
    """graph.mix32 (iteration number -> mask word of a captured training step) is hash_mix of macx_common.hip.h / oracle/dropout_hash.py"""
    from oracle import dropout_hash as dh
    from macx.graph import mix32
    for x in (0, 1, 2, 12345, 0xFFFFFFFF, 0x80000000):
        assert mix32(x) == int(dh.hash_mix((x + 0x7F4A7C15) & 0xFFFFFFFF))
    assert len({mix32(i) for i in range(1000)}) == 1000


def test_bench_settle_and_best_block_logic(monkeypatch):
    """bench.settle stops when two replay blocks agree within 2 % and the eager block is within 8 % of them, or after max_s;
    bench.best_block reports the fastest block (pure host logic: time and the device are faked)"""
    import bench
    clock = {"t": 0.0}
    monkeypatch.setattr(bench.time, "perf_counter", lambda: clock["t"])
    monkeypatch.setattr(bench.torch.cuda, "synchronize", lambda *a, **k: None)
    costs = {"g": [4.0e-3] * 100, "e": [6.0e-3, 5.0e-3, 4.1e-3] + [4.1e-3] * 100}
    calls = {"g": 0, "e": 0}

    def mk(kind):
        def step(i):
            clock["t"] += costs[kind][calls[kind] // 20]
            calls[kind] += 1
        return step

    out = bench.settle(mk("g"), mk("e"), 30.0)
    assert out["quiet"] and out["n_blocks"] == 3 and out["blocks_replay_eager_ms"][-1] == (4.0, 4.1)
    # a box that never gets quiet: gives up after max_s
    calls.update(g=0, e=0)
    costs["e"] = [9.0e-3] * 1000
    out = bench.settle(mk("g"), mk("e"), 1.0)
    assert not out["quiet"] and out["seconds"] >= 1.0
    # best_block: the fastest of three blocks
    seq = iter([5e-3] * 4 + [3e-3] * 4 + [4e-3] * 4)
    def one(i):
        clock["t"] += next(seq)
    assert abs(bench.best_block(one, 4) - 3e-3) < 1e-12


def test_bench_metric_blocks_report_the_median_and_the_spread(monkeypatch):
    """bench.time_blocks / block_summary: the metric is METRIC_BLOCKS blocks of exactly K steps; value is the MEDIAN block, with
    min / max / spread beside it; step indices run on across the blocks (fresh dropout masks per step)"""
    import bench
    clock = {"t": 0.0}
    monkeypatch.setattr(bench.time, "perf_counter", lambda: clock["t"])
    monkeypatch.setattr(bench.torch.cuda, "synchronize", lambda *a, **k: None)
    per_block = [5e-3, 4e-3, 4.2e-3, 9e-3, 4.1e-3]
    seen = []

    def step(i):
        seen.append(i)
        n_timed = len(seen) - 6                      # 2 priming + 3 warmup steps come first
        clock["t"] += per_block[max(n_timed, 0) // 10]

    dts = bench.time_blocks(step, 10, 3, 2, lambda: None, 1, None, None, blocks=5)
    assert seen == list(range(55))
    assert [round(d / 10, 6) for d in dts] == per_block
    med, info = bench.block_summary(dts, 10, 64)
    assert abs(med / 10 - 4.2e-3) < 1e-12
    assert info["ms_per_step_min"] == 4.0 and info["ms_per_step_max"] == 9.0 and info["blocks"] == 5
    assert abs(info["spread"] - (9.0 - 4.0) / 4.2) < 1e-3


def test_product_code_never_touches_the_debug_knobs():
    """The A/B hooks travel per call in macx_opts.tune (ABI 5; the process-global macx_debug_set of ABI <= 4 is gone from the library),
    and what is left of process-wide state -- macx_gemm_mode's default kernel family, options.SESSION_TUNE on the Python side -- is
    never written by a module of the product package: only bench.py, tests/ and tools/ do, from MACX_* environment variables."""
    import glob
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "mac-network_amd")
    offenders = []
    for path in sorted(glob.glob(os.path.join(pkg, "*.py"))):
        src = open(path).read()
        for m in re.finditer(r"\.(macx_debug_set|macx_gemm_mode)\s*\((?!\s*-1\s*\))", src):      # (macx_gemm_mode(-1) only reads the mode)
            offenders.append("%s:%d %s" % (os.path.basename(path), src[:m.start()].count("\n") + 1, m.group(1)))
        for m in re.finditer(r"SESSION_TUNE\s*(\[[^\]]*\]\s*=|\.(update|setdefault|pop|clear)\()", src):
            offenders.append("%s:%d writes SESSION_TUNE" % (os.path.basename(path), src[:m.start()].count("\n") + 1))
    assert not offenders, offenders
    hdr = open(os.path.join(root, "include", "macx.h")).read()
    assert "macx_debug_set(" not in hdr and "int32_t tune[16]" in hdr
    # no `static int` knob is left in the library's sources: every accessor reads the call's table
    for path in sorted(glob.glob(os.path.join(pkg, "csrc", "*"))):
        src = open(path).read()
        assert not re.search(r"inline int&\s+\w+\(\)\s*\{\s*static int", src.replace("gemm_default_mode", "")), path




def test_tune_keys_match_the_header():
    """macx._lib.TUNE (names -> indices of macx_opts.tune) is the MACX_TUNE_* enumeration of include/macx.h, key for key."""
    import macx
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "macx.h")).read()
    enum = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"MACX_TUNE_([A-Z0-9_]+)\s*=\s*(\d+)", hdr)}
    assert enum == dict(macx._lib.TUNE)
    assert len(set(enum.values())) == len(enum) and max(enum.values()) < 16
