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


def reject_like_reference(config):
    """Raise what the reference raises at graph-build time for broken option values."""
    g = lambda n: get(config, n)
    if g("initKBwithQ") != "NON":          # mac_cell.py:564 passes expandY=, ops.concat takes extendY (ops.py:65)
        raise TypeError("concat() got an unexpected keyword argument 'expandY'")
    if g("addNullWord"):                   # mac_cell.py:519,573-574
        raise UnboundLocalError("local variable 'questionLengths' referenced before assignment")
    if g("readMemAttType") == "DIAG" or (g("readCtrl") and g("readCtrlAttType") == "DIAG"):   # ops.py:704-707
        raise UnboundLocalError("local variable 'output' referenced before assignment")
    if g("readMemConcatProj") and not g("readProjInputs"):          # ops.py:691,716 (evaluated whether or not concat["x"] is set)
        raise UnboundLocalError("local variable 'projVals' referenced before assignment")
    if g("readCtrl") and g("readProjInputs") and g("attDim") != g("ctrlDim"):                  # mac_cell.py:245-246
        raise NameError("name 'ctrlDim' is not defined")
    if g("writeGate") and g("writeGateShared"):   # [B,d] * [B] does not broadcast (ops.py:317, mac_cell.py:367)
        raise ValueError("Dimensions must be equal")
    if g("readCtrl") and g("readCtrlConcatKB") and g("readCtrlConcatProj") and not g("readProjInputs") and not g("readCtrlConcatInter"):
        # mac_cell.py:252-258: tf.concat([interactions, projectedKB]) with projectedKB = None (ops.convert_to_tensor(None))
        raise ValueError("None values not supported.")
    if g("readCtrl") and g("readCtrlConcatInter"):
        # mac_cell.py:248-250 drops the width ops.mul returns: inter2att builds a [dim] weight for a [.., 2 dim] tensor
        raise ValueError("Dimensions must be equal")
    if g("readSmryKBProj") and not g("readProjInputs"):
        # mac_cell.py:271-272: att2Smry(attention, None) -> tensor * None (ops.py:150)
        raise ValueError("None values not supported.")
    if g("relu") == "PRM" and g("controlInputAct") == "RELU" and int(g("netLength")) > 1:
        # mac_cell.py:445 applies the activation in the cell's own scope (reuse=None, mac_cell.py:422): step 1 creates
        # MACCell/prelu/alpha a second time
        raise ValueError("Variable MACnetwork/MACCell/prelu/alpha already exists, disallowed. "
                         "Did you mean to set reuse=True in VarScope?")


GEMM_FAMILIES = {None: 0, "default": 0, "native": 1, "split": 2, "h2": 3}


def freeze(config, gemm=None):
    """config -> MacxOpts.  gemm: kernel family of the knowledge-base GEMMs for every call made with these options
    ("native" | "split" | "h2"; None = the process default, macx_gemm_mode) -- macx_opts.gemm_family."""
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
    if o.read_mem_act == _lib.ACT["NON"]:
        unsupported.append("readMemAct=NON (no memKbProj_2 layer, ops.py:325)")
    if unsupported:
        raise UnsupportedOptions("no HIP path yet for: " + ", ".join(unsupported))
    return o
