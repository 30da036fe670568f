"""Freezes the reference's option surface (config.py:194-223, 292-387) into the POD `macx_opts`
the C ABI takes.

`freeze(config)` accepts ANY object carrying the reference's flag names (its own `config`
singleton, an argparse namespace, oracle.default_config()).  Three outcomes, as in SURVEY.md 8a:
  * option values that raise in the reference raise the same exception class here;
  * legal combinations the HIP path does not cover yet raise UnsupportedOptions (never a CPU fallback);
  * everything else becomes a MacxOpts struct.
"""
from . import _lib

# every cell flag and its config.py default; missing attributes take the default
DEFAULTS = dict(
    memDim=512, ctrlDim=512, attDim=512, unsharedCells=False, initCtrl="PRM", initMem="PRM", initKBwithQ="NON",
    addNullWord=False, controlWholeQ=False, controlContinuous=False, controlContextual=False,
    controlInWordsProj=False, controlOutWordsProj=False, controlInputUnshared=False, controlInputAct="TANH",
    controlFeedPrev=False, controlFeedPrevAtt=False, controlFeedInputs=False, controlContAct="NON",
    controlConcatWords=False, controlProj=False, controlProjAct="NON", readProjInputs=False, readProjShared=False,
    readMemAttType="MUL", readMemConcatKB=False, readMemConcatProj=False, readMemProj=False, readMemAct="RELU",
    readCtrl=False, readCtrlAttType="MUL", readCtrlConcatKB=False, readCtrlConcatProj=False,
    readCtrlConcatInter=False, readCtrlAct="RELU", readSmryKBProj=False, writeInputs="BOTH", writeConcatMul=False,
    writeInfoProj=False, writeInfoAct="NON", writeSelfAtt=False, writeSelfAttMod="NON", writeMergeCtrl=False,
    writeMemProj=False, writeMemAct="NON", writeGate=False, writeGateShared=False, writeGateBias=1.0,
    memoryVariationalDropout=False, memoryDropout=0.85, readDropout=0.85, writeDropout=1.0, relu="STD", mulBias=0.0,
    memoryBN=False, bnDecay=0.999, bnCenter=False, bnScale=False, netLength=16,
)


_seed_counter = [0x5EED]


def fresh_seed(seed, train):
    """Dropout stream seed of one graph run.  The reference draws fresh masks on every session.run; a caller that does
    not pass `seed` therefore gets a new one per training call (a process-wide counter), never a repeated sub-network.
    Data-parallel ranks must pass the same explicit seed so that their shards see the masks of the full batch."""
    if seed is not None:
        return int(seed)
    if not train:
        return 0
    _seed_counter[0] += 1
    return _seed_counter[0]


class UnsupportedOptions(NotImplementedError):
    """A legal reference option combination that has no HIP path yet (MACX_EUNSUPPORTED)."""


def get(config, name):
    return getattr(config, name, DEFAULTS[name])


def _resolve_act(config, name):
    """ops.activations (ops.py:181-187); "RELU" resolves through config.relu (ops.py:161-179)."""
    if name != "RELU":
        return _lib.ACT[name]
    r = get(config, "relu")
    if r == "STD":
        return _lib.ACT["RELU"]
    if r == "ELU":
        return _lib.ACT["ELU"]
    if r == "LKY":   # config.reluAlpha's flag is commented out (config.py:221)
        raise AttributeError("'Config' object has no attribute 'reluAlpha'")
    if r == "SELU":  # accepted by argparse (config.py:220), no branch in ops.relu (ops.py:171-179)
        raise UnboundLocalError("local variable 'output' referenced before assignment")
    if r == "PRM":
        raise UnsupportedOptions("relu=PRM (PReLU with a per-call-site alpha) has no HIP path yet")
    raise ValueError("relu=%r" % r)


def resolve_activations(config):
    """Resolve every activation flag the graph would actually apply (the same call sites as the reference:
    mac_cell.py:149-150, 164, 237, 262, 313, 355, 445), raising what ops.relu raises for LKY / SELU."""
    g = lambda n: get(config, n)
    acts = {"controlInputAct": _resolve_act(config, g("controlInputAct")), "writeInfoAct": _resolve_act(config, g("writeInfoAct")),
            "writeMemAct": _resolve_act(config, g("writeMemAct"))}
    if g("controlFeedPrev"):
        acts["controlContAct"] = _resolve_act(config, g("controlContAct"))
    if g("controlProj"):
        acts["controlProjAct"] = _resolve_act(config, g("controlProjAct"))
    if g("readMemProj"):
        acts["readMemAct"] = _resolve_act(config, g("readMemAct"))
    if g("readCtrl"):
        acts["readCtrlAct"] = _resolve_act(config, g("readCtrlAct"))
    return acts


def _relu_check(config, act):
    """What ops.activations[act] raises when it is applied (ops.py:161-187)."""
    if act != "RELU":
        return
    r = get(config, "relu")
    if r == "LKY":   # config.reluAlpha's flag is commented out (config.py:221)
        raise AttributeError("'Config' object has no attribute 'reluAlpha'")
    if r == "SELU":  # accepted by argparse (config.py:220), no branch in ops.relu (ops.py:171-179)
        raise UnboundLocalError("local variable 'output' referenced before assignment")


def reject_like_reference(config):
    """Raise what the reference raises while it builds the graph for these option values (SURVEY appendix B and what running
    the reference on random combinations added) -- as a DRY RUN of mac_cell.py in the reference's own order, so that a
    combination with several defects raises the one the reference meets first (tests/test_reference_exec.py)."""
    g = lambda n: get(config, n)
    none = lambda: ValueError("None values not supported.")          # ops.convert_to_tensor(None): tf.concat / tensor * None
    # ---- zero_state (mac_cell.py:539-592)
    if g("initKBwithQ") != "NON":          # mac_cell.py:564 passes expandY=, ops.concat takes extendY (ops.py:65)
        raise TypeError("concat() got an unexpected keyword argument 'expandY'")
    if g("addNullWord"):                   # mac_cell.py:519,573-574
        raise UnboundLocalError("local variable 'questionLengths' referenced before assignment")
    # ---- step 0: control input and control unit (mac_cell.py:442-451, 133-187)
    _relu_check(config, g("controlInputAct"))
    if g("controlFeedPrev"):
        _relu_check(config, g("controlContAct"))
    if g("controlProj"):
        _relu_check(config, g("controlProjAct"))
    # ---- read unit (mac_cell.py:209-277)
    mem, att, ctrl = g("memDim"), g("attDim"), g("ctrlDim")
    proj = bool(g("readProjInputs"))
    dim = att if proj else mem
    inter = dim
    if g("readMemConcatProj") and not proj:          # ops.py:716 (evaluated whether or not concat["x"] is set)
        raise UnboundLocalError("local variable 'projVals' referenced before assignment")
    if g("readMemAttType") == "DIAG":                # ops.py:704-707: the DIAG branch assigns `activations`, not `output`
        raise UnboundLocalError("local variable 'output' referenced before assignment")
    if g("readMemConcatKB"):
        inter += att if g("readMemConcatProj") else mem
    if g("readMemProj"):
        _relu_check(config, g("readMemAct"))
    else:
        dim = inter
    width = dim
    if g("readCtrl"):
        if ctrl != dim:                              # mac_cell.py:245-246 references an undefined `ctrlDim`
            raise NameError("name 'ctrlDim' is not defined")
        if g("readCtrlAttType") == "DIAG":
            raise UnboundLocalError("local variable 'output' referenced before assignment")
        if g("readCtrlConcatInter"):
            width = 2 * dim                          # ops.mul returns the wider tensor, mac_cell.py:248-250 keeps `dim`
        if g("readCtrlConcatKB"):
            if g("readCtrlConcatProj") and not proj:
                raise none()                         # mac_cell.py:252-258: tf.concat([interactions, projectedKB = None])
            added = att if g("readCtrlConcatProj") else mem
            width, dim = width + added, dim + added
        _relu_check(config, g("readCtrlAct"))
    if width != dim:                                 # inter2att builds a [dim] weight for a wider tensor
        raise ValueError("Dimensions must be equal")
    if g("readSmryKBProj") and not proj:             # mac_cell.py:271-275: att2Smry(attention, None) -> tensor * None
        raise none()
    # ---- write unit (mac_cell.py:305-375); widths as TF's shape inference sees them
    dims = lambda: ValueError("Dimensions must be equal")
    iw = att if (g("readSmryKBProj") and proj) else mem          # the summary of the PROJECTED knowledge base is attDim wide
    if g("writeInfoProj"):
        if iw != mem:
            raise dims()
    _relu_check(config, g("writeInfoAct"))
    nw = dim = mem
    if g("writeInputs") == "INFO":
        nw = iw
    elif g("writeInputs") == "SUM":
        if iw != mem:
            raise dims()
    elif g("writeInputs") == "BOTH":
        if g("writeConcatMul") and iw != mem:
            raise dims()
        nw, dim = (2 * mem + iw, 3 * mem) if g("writeConcatMul") else (mem + iw, 2 * mem)
    if g("writeSelfAtt"):
        nw, dim = nw + mem, dim + mem
    if g("writeMergeCtrl"):
        nw, dim = nw + ctrl, dim + mem
    if g("writeMemProj") or dim != mem:
        if nw != dim:
            raise dims()
        nw = mem
    _relu_check(config, g("writeMemAct"))
    if g("writeGate"):
        if g("writeGateShared"):      # [B,d] * [B] does not broadcast (ops.py:317, mac_cell.py:367)
            raise dims()
        if nw != mem:
            raise dims()
    if nw != mem and int(g("netLength")) > 1:            # the next step multiplies / projects a memory of the wrong width
        raise dims()
    # ---- step 1
    if g("relu") == "PRM" and g("controlInputAct") == "RELU" and int(g("netLength")) > 1:
        # mac_cell.py:445 applies the activation in the cell's own scope (reuse=None, mac_cell.py:422): step 1 creates
        # MACCell/prelu/alpha a second time
        raise ValueError("Variable MACnetwork/MACCell/prelu/alpha already exists, disallowed. "
                         "Did you mean to set reuse=True in VarScope?")


GEMM_FAMILIES = {None: 0, "default": 0, "native": 1, "split": 2, "h2": 3}

# Tuning values every MacxOpts frozen from now on starts with ({key name or number: value}; _lib.TUNE): the switchboard of the A/B
# tools and of the test session's MACX_* environment variables (tests/conftest.py, bench.py, tools/kv_sweep.py).  Empty in
# production -- no module of this package writes it -- and it is HOST state of the Python layer: what reaches the library is the
# per-call table macx_opts.tune.
SESSION_TUNE = {}


def freeze(config, gemm=None, tune=None):
    """config -> MacxOpts.  gemm: kernel family of the knowledge-base GEMMs for every call made with these options
    ("native" | "split" | "h2"; None = the process default, macx_gemm_mode) -- macx_opts.gemm_family.  tune: {key: value} for
    macx_opts.tune (A/B hooks, _lib.TUNE), on top of SESSION_TUNE."""
    g = lambda n: get(config, n)
    if gemm not in GEMM_FAMILIES:
        raise ValueError("gemm=%r: one of %s" % (gemm, sorted(k for k in GEMM_FAMILIES if k)))
    reject_like_reference(config)
    unsupported = []
    d = g("memDim")
    if not (g("ctrlDim") == d and g("attDim") == d):
        unsupported.append("memDim == ctrlDim == attDim required")
    for flag in ("unsharedCells", "controlWholeQ", "controlContinuous", "controlInWordsProj", "controlOutWordsProj",
                 "controlConcatWords", "controlProj", "readProjShared", "readCtrlConcatKB", "readCtrlConcatProj",
                 "readCtrlConcatInter", "readSmryKBProj", "writeConcatMul", "writeInfoProj", "writeMergeCtrl", "memoryBN"):
        if g(flag):
            unsupported.append("--%s" % flag)
    for flag in ("readProjInputs", "readMemConcatKB", "readMemConcatProj", "readMemProj", "readCtrl"):
        if not g(flag):
            unsupported.append("without --%s" % flag)
    if g("readMemAttType") != "MUL" or g("readCtrlAttType") != "MUL":
        unsupported.append("read*AttType != MUL")
    if g("writeInputs") != "BOTH":
        unsupported.append("writeInputs=%s" % g("writeInputs"))
    if g("writeInfoAct") != "NON":
        unsupported.append("writeInfoAct=%s" % g("writeInfoAct"))
    if g("mulBias") != 0.0:
        unsupported.append("mulBias != 0")
    o = _lib.MacxOpts()
    o.abi_version = _lib.ABI_VERSION
    o.init_ctrl = _lib.INIT[g("initCtrl")]
    o.init_mem = _lib.INIT[g("initMem")]
    o.control_input_unshared = int(bool(g("controlInputUnshared")))
    o.control_input_act = _resolve_act(config, g("controlInputAct"))
    o.control_feed_prev = int(bool(g("controlFeedPrev")))
    o.control_feed_prev_att = int(bool(g("controlFeedPrevAtt")))
    o.control_feed_inputs = int(bool(g("controlFeedInputs")))
    o.control_cont_act = _resolve_act(config, g("controlContAct"))
    o.read_mem_act = _resolve_act(config, g("readMemAct"))
    o.read_ctrl_act = _resolve_act(config, g("readCtrlAct"))
    o.write_inputs = _lib.WRITE_INPUTS[g("writeInputs")]
    o.write_self_att = int(bool(g("writeSelfAtt")))
    o.write_self_att_cont = int(g("writeSelfAttMod") == "CONT")
    o.write_mem_act = _resolve_act(config, g("writeMemAct"))
    o.write_gate = int(bool(g("writeGate")))
    o.write_gate_shared = int(bool(g("writeGateShared")))
    o.write_gate_bias = float(g("writeGateBias"))
    o.memory_variational_dropout = int(bool(g("memoryVariationalDropout")))
    o.gemm_family = GEMM_FAMILIES[gemm]
    for k, v in list(SESSION_TUNE.items()) + list((tune or {}).items()):
        _lib.set_tune(o, k, v)
    if o.read_mem_act == _lib.ACT["NON"]:
        unsupported.append("readMemAct=NON (no memKbProj_2 layer, ops.py:325)")
    if unsupported:
        raise UnsupportedOptions("no HIP path yet for: " + ", ".join(unsupported))
    return o
