"""TEST INFRASTRUCTURE: executes the reference's OWN code -- /root/reference/{config,ops,mac_cell,model}.py,
unmodified -- on the eager TF-1.x stand-in of tests/tf1_shim, and adapts its inputs/outputs so that
oracle/mac_oracle.py can be compared with it on identical parameters, inputs and dropout draws.

What comes from the reference here: the flag parser (config.parseArgs on the published flag files), every
line of MACCell (mac_cell.py:59-592), the ops.py primitives it calls, and MACnet.MACnetwork / outputOp /
classifier / addAnswerLossOp / addPredOp / stem (model.py:165-204, 428-489, 512-612), called as plain functions
on a stand-in `self` that carries only the attributes those methods read.
What does not: TensorFlow (replaced by the shim) -- and therefore TF's RNG stream; draws are injected.

/root/reference exists only in the build container: callers must check `available()` first.
"""
import importlib
import os
import sys
from types import SimpleNamespace

import torch

REF = os.environ.get("MACX_REFERENCE_DIR", "/root/reference")
SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tf1_shim")

_mods = {}


def available():
    return os.path.isfile(os.path.join(REF, "mac_cell.py"))


def load():
    """Import the reference modules against the shim (once)."""
    if _mods:
        return _mods
    for p in (REF, SHIM):
        if p not in sys.path:
            sys.path.insert(0, p)
    tf = importlib.import_module("tensorflow")
    if not hasattr(tf, "shim_reset"):
        raise RuntimeError("a real tensorflow shadowed the test shim")
    for name in ("config", "ops", "mac_cell", "model"):
        m = importlib.import_module(name)
        if not os.path.abspath(m.__file__).startswith(os.path.abspath(REF)):
            raise RuntimeError("module %s resolved to %s, not the reference" % (name, m.__file__))
        _mods[name] = m
    _mods["tf"] = tf
    return _mods


def parse_flags(flag_file, *extra):
    """The reference's own parser (config.py:95-428) over one of its published flag files
    (configs/args*.txt) plus extra command-line flags; returns the `config` singleton."""
    M = load()
    # argparse only fills in defaults for attributes the namespace does not have yet, and the reference parses into its
    # module-level singleton (config.py:428): start every parse from a fresh process's state
    M["config"].config.__dict__.clear()
    argv = sys.argv
    try:
        sys.argv = ["main.py"] + (["@" + os.path.join(REF, "configs", flag_file)] if flag_file else []) + [str(a) for a in extra]
        M["config"].parseArgs()
    finally:
        sys.argv = argv
    cfg = M["config"].config
    # the reference's parser declares --unsharedCells with type=bool (config.py:297): any string is True; leave as parsed
    return cfg


def dims_flags(d, p, hidden=None):
    f = ["--netLength", p, "--memDim", d, "--ctrlDim", d, "--attDim", d]
    if hidden is not None:
        f += ["--outClassifierDims", hidden]
    return f


CELL_FLAGS = [  # every flag the cell reads
    "netLength", "memDim", "ctrlDim", "attDim", "unsharedCells", "initCtrl", "initMem", "initKBwithQ", "addNullWord",
    "controlWholeQ", "controlContinuous", "controlContextual", "controlInWordsProj", "controlOutWordsProj",
    "controlInputUnshared", "controlInputAct", "controlFeedPrev", "controlFeedPrevAtt", "controlFeedInputs",
    "controlContAct", "controlConcatWords", "controlProj", "controlProjAct", "readProjInputs", "readProjShared",
    "readMemAttType", "readMemConcatKB", "readMemConcatProj", "readMemProj", "readMemAct", "readCtrl",
    "readCtrlAttType", "readCtrlConcatKB", "readCtrlConcatProj", "readCtrlConcatInter", "readCtrlAct", "readSmryKBProj",
    "writeInputs", "writeConcatMul", "writeInfoProj", "writeInfoAct", "writeSelfAtt", "writeSelfAttMod",
    "writeMergeCtrl", "writeMemProj", "writeMemAct", "writeGate", "writeGateShared", "writeGateBias",
    "memoryVariationalDropout", "memoryDropout", "readDropout", "writeDropout", "relu", "mulBias", "memoryBN",
    "bnDecay", "bnCenter", "bnScale", "outQuestion", "outQuestionMul", "outClassifierDims", "outputDropout",
]


def snapshot(cfg):
    """The parsed values of the cell's flags as a plain dict (what the oracle's config must equal)."""
    return {k: getattr(cfg, k) for k in CELL_FLAGS}


def run_reference(cfg, vecQ, questionWords, questionCntxWords, lengths, kb, train=False, keeps=(1.0, 1.0, 1.0),
                  output_keep=1.0, preset=None, seed=0, dtype=torch.float64, need_grad=False, answers=None,
                  with_output=True, answerWordsNum=7, draws=None):
    """MACnet.MACnetwork (+ output unit) exactly as model.py builds it.  `draws`: optional list of uniform tensors
    replayed in call order (otherwise the shim's generator draws and logs them).
    Returns dict(control, memory, cell, variables, draws, logits, loss, preds, inputs)."""
    M = load()
    tf, model = M["tf"], M["model"]
    tf.shim_reset(dtype=dtype, seed=seed, preset=preset, require_grad=need_grad)
    if draws is not None:
        q = list(draws)
        tf.state.uniform_hook = lambda shape: _pop(q, shape)
    cfg.answerWordsNum = answerWordsNum       # set at run time by preprocess.py:685-686
    ins = [tf.wrap(t.detach().to(dtype).clone()).requires_grad_(need_grad) for t in (vecQ, questionWords, questionCntxWords, kb)]
    vq_, qw_, cw_, kb_ = ins
    B = vecQ.shape[0]
    fake = SimpleNamespace(dropouts={"memory": keeps[0], "read": keeps[1], "write": keeps[2], "output": output_keep},
                           batchSize=B, train=train, batchNorm=None, answerLossList=[], correctNumList=[], answerAccList=[])
    control, memory = model.MACnet.MACnetwork(fake, kb_, vq_, qw_, cw_, tf.wrap(lengths.clone()))
    out = dict(control=control, memory=memory, cell=fake.macCell, inputs=dict(vecQ=vq_, questionWords=qw_,
                                                                               questionCntxWords=cw_, kb=kb_))
    if with_output:
        features, dim = model.MACnet.outputOp(fake, memory, vq_, None, None)
        logits = model.MACnet.classifier(fake, features, dim)
        out["logits"] = logits
        if answers is not None:
            loss, losses = model.MACnet.addAnswerLossOp(fake, logits, answers)
            preds, corrects, correctNum = model.MACnet.addPredOp(fake, logits, answers)
            out.update(loss=loss, losses=losses, preds=preds, correctNum=correctNum)
    out["variables"] = dict(tf.state.variables)
    out["draws"] = [u for _, u in tf.state.draws]
    return out


def _pop(q, shape):
    if not q:
        raise AssertionError("the reference drew more random tensors than were recorded")
    u = q.pop(0)
    if tuple(u.shape) != tuple(shape):
        raise AssertionError("draw shape %s, recorded %s" % (tuple(shape), tuple(u.shape)))
    return u


def replay_mask_fn(draws, keeps, output_keep=1.0):
    """mask_fn for oracle.mac_oracle: hands the oracle floor(keep + U) for the reference's draws, FIFO.
    The oracle asks for masks in graph order; so does the reference (both are op-for-op)."""
    from oracle import dropout_hash as dh
    q = list(draws)
    site_keep = {dh.SITE_MEM_VAR: keeps[0], dh.SITE_MEM: keeps[0], dh.SITE_READ_KB: keeps[1], dh.SITE_READ_MEM: keeps[1],
                 dh.SITE_READ_ATT: keeps[1], dh.SITE_WRITE_INFO: keeps[2]}

    def fn(site, step, shape):
        u = _pop(q, shape)
        return torch.floor(site_keep[site] + u)

    def rest(keep):
        """remaining draws as masks (classifier layers), in order"""
        out = [torch.floor(keep + u) for u in q]
        del q[:]
        return out

    fn.rest = rest
    fn.left = lambda: len(q)
    return fn


def run_reference_stem(cfg, images_nhwc, keep=1.0, seed=0, dtype=torch.float64, need_grad=False, draws=None, out_dim=None):
    """MACnet.stem (model.py:165-204) exactly as the reference builds it: ops.CNNLayer -> ops.cnn (ops.py:380-438).
    images_nhwc [B,H,W,C] (the graph's layout after model.py:67-68).  Returns dict(kb [B,H*W,outDim], variables, draws, images)."""
    M = load()
    tf, model = M["tf"], M["model"]
    tf.shim_reset(dtype=dtype, seed=seed, require_grad=need_grad)
    if draws is not None:
        q = list(draws)
        tf.state.uniform_hook = lambda shape: _pop(q, shape)
    B, H, W, C = images_nhwc.shape
    img = tf.wrap(images_nhwc.detach().to(dtype).clone()).requires_grad_(need_grad)
    fake = SimpleNamespace(dropouts={"stem": keep}, batchSize=B, H=H, W=W, batchNorm=None)
    kb = model.MACnet.stem(fake, img, C, out_dim if out_dim is not None else cfg.memDim)
    return dict(kb=kb, variables=dict(tf.state.variables), draws=[u for _, u in tf.state.draws], images=img)


def run_reference_encoder(cfg, questions, lengths, emb, keep_input=1.0, keep_question=1.0, seed=0, dtype=torch.float64,
                          need_grad=False, draws=None):
    """qEmbeddingsOp (model.py:207-219) + encoder (model.py:279-307 -> ops.RNNLayer / biRNNLayer, ops.py:859-950) as the
    reference builds them; the recurrent cell and tf.nn.bidirectional_dynamic_rnn are the shim's restatement of TF's.
    Returns dict(words [B,S,encDim], vecQ [B,encDim], variables, draws)."""
    M = load()
    tf, model = M["tf"], M["model"]
    tf.shim_reset(dtype=dtype, seed=seed, require_grad=need_grad)
    if draws is not None:
        q = list(draws)
        tf.state.uniform_hook = lambda shape: _pop(q, shape)
    fake = SimpleNamespace(dropouts={"encInput": keep_input, "question": keep_question, "stateInput": 1.0}, batchSize=questions.shape[0])
    qs, _ = model.MACnet.qEmbeddingsOp(fake, tf.wrap(questions.clone()), emb.detach().to(dtype).clone())
    proj = (cfg.encDim != cfg.ctrlDim) or cfg.encProj                                    # model.py:785-787, as MACnet.build calls it
    words, vecQ = model.MACnet.encoder(fake, qs, tf.wrap(lengths.clone()), proj, proj, cfg.ctrlDim)
    return dict(words=words, vecQ=vecQ, variables=dict(tf.state.variables), draws=[u for _, u in tf.state.draws])


def run_reference_training(cfg, vecQ, questionWords, questionCntxWords, lengths, kb, answers, steps, lr, preset=None, seed=0,
                           dtype=torch.float64, answerWordsNum=7):
    """`steps` training steps exactly as model.py strings them: MACnetwork -> outputOp -> classifier -> addAnswerLossOp
    (evaluation-mode dropout: the optimizer is what is under test), then MACnet.addOptimizerOp (once) and, per step,
    MACnet.computeGradients + MACnet.addTrainingOp (model.py:615-669) -- clip_by_global_norm, AdamOptimizer.apply_gradients
    and ExponentialMovingAverage.apply as that code calls them, on the shim's restatement of those three TF classes.
    Returns per step: the gradients the reference computed, the global norm, and the variables / Adam slots / EMA shadows
    afterwards (name -> tensor, names without ':0')."""
    M = load()
    tf, model = M["tf"], M["model"]
    tf.shim_reset(dtype=dtype, seed=seed, preset=preset, require_grad=True)
    cfg.answerWordsNum = answerWordsNum
    B = vecQ.shape[0]
    fake = SimpleNamespace(dropouts={"memory": 1.0, "read": 1.0, "write": 1.0, "output": 1.0}, batchSize=B, train=True,
                           batchNorm=None, answerLossList=[], correctNumList=[], answerAccList=[], lr=lr)
    optimizer = model.MACnet.addOptimizerOp(fake)                       # model.py:616-620
    ins = [tf.wrap(t.detach().to(dtype).clone()) for t in (vecQ, questionWords, questionCntxWords, kb)]
    vq_, qw_, cw_, kb_ = ins
    out = []
    for step in range(steps):
        with tf.variable_scope(tf.get_variable_scope(), reuse=(True if step > 0 else None)):
            control, memory = model.MACnet.MACnetwork(fake, kb_, vq_, qw_, cw_, tf.wrap(lengths.clone()))
            features, dim = model.MACnet.outputOp(fake, memory, vq_, None, None)
            logits = model.MACnet.classifier(fake, features, dim)
            loss, _ = model.MACnet.addAnswerLossOp(fake, logits, answers)
        gv = model.MACnet.computeGradients(fake, optimizer, loss, None)                      # model.py:626-637
        grads = {v.name[:-2]: (None if g is None else g.detach().clone()) for g, v in gv}
        _, norm = model.MACnet.addTrainingOp(fake, optimizer, gv)                            # model.py:643-669
        names = list(tf.state.variables)
        out.append(dict(loss=float(loss), norm=float(norm), grads=grads,
                        variables={k: tf.state.variables[k].detach().clone() for k in names},
                        m={k: optimizer.m[k + ":0"].clone() for k in names if k + ":0" in optimizer.m},
                        v={k: optimizer.v[k + ":0"].clone() for k in names if k + ":0" in optimizer.v},
                        ema={k: tf.state.ema[k].clone() for k in names if k in tf.state.ema}))
    return dict(steps=out, global_step=fake.globalStep.value, ema_names=sorted(fake.emaDict))
