"""TEST INFRASTRUCTURE -- not part of the product path.  PARITY PINNED to the reference's own code (see below).

Op-for-op CPU restatement (PyTorch, fp32 or fp64) of the reference's MAC cell:
    /root/reference/mac_cell.py   MACCell.control :133-187, read :209-277, write :305-375,
                                  __call__ :420-480, initState :496-505, zero_state :539-592
    /root/reference/ops.py        getWeight :18-24, getBias :38-42, multiply :50-59, concat :65-78,
                                  inter2logits :114-120, inter2att :140-144, att2Smry :149-150,
                                  relu :161-179, activations :181-187, expMask :243-247,
                                  linear :298-333, FCLayer :349-359, mul :668-725,
                                  generateVarDpMask/applyVarDpMask :1054-1067
    /root/reference/model.py      MACnetwork :428-489, outputOp :512-528, classifier :547-576,
                                  addAnswerLossOp :593-599, addPredOp :603-612

Every TF graph node of those functions is one torch op here, including the tensors the reference
materialises and a fused implementation would not (the `zeros_like(x) + y` broadcast of ops.py:697,
the [B,N,2d] concat of ops.py:718, the `inp * W` product of ops.py:317, `att * KB` of ops.py:150).
That makes this file both the checker for the HIP path and the "reference CPU path" timed by
bench.py (`cpu_baseline.kind = "port"`).

PARITY PINNED (since round 2).  The reference ships no tests, golden vectors or seeds and TensorFlow
1.x cannot be imported here, so the pin is the reference's own code EXECUTED: /root/reference/{config,ops,mac_cell,model}.py
are imported unmodified on an eager stand-in for the ~60 `tf.*` calls they make (tests/tf1_shim/tensorflow, torch fp64)
and this file must reproduce them to 1e-12 -- states, attentions, logits, loss, predictions, every gradient, variable names
and creation order, the random draws in order -- for the five flag files x eval/train, 40 further option sets, ~100 random
option combinations (raise where the reference raises, with the same class), the stem and the question encoder
(tests/test_reference_exec.py, tests/ref_exec.py; runs wherever /root/reference exists).  The same reference runs are
committed as fixtures (tests/golden/reference/*.npz + the script that wrote them) so that the check travels to machines
without the reference (tests/test_reference_golden.py on CPU, tests/test_gpu_reference_golden.py for the HIP path).
What the stand-in restates rather than executes: the tf.* primitives themselves (matmul, softmax, dropout's
x / keep * floor(keep + U), variable scoping incl. default-name uniquification, batch_norm, conv2d, BasicLSTMCell and the
dynamic-rnn loop) -- each a few lines, checked against torch's own implementations where one exists.  Further
self-consistency: oracle/mac_numpy.py (independent fp64 closed form), finite differences of the autograd gradients.

Dropout: TF's `x / keep * floor(keep + U)`; the uniform draw is replaced by explicit 0/1 masks handed
in through `mask_fn(site, step, shape)` (sites in oracle/dropout_hash.py).
"""
import contextlib
import math
from contextlib import contextmanager
from types import SimpleNamespace

import torch

from . import dropout_hash as dh

inf = 1e30  # ops.py:10


# -------------------------------------------------------------------------------------------------
# config: defaults of every flag the cell reads (config.py:194-223, 292-387) + the flag files
# -------------------------------------------------------------------------------------------------
def _tf_shape_errors(fn):
    """TensorFlow infers static shapes while the graph is built and rejects incompatible ones with ValueError
    ("Dimensions must be equal ..."); torch meets the same defect at run time as RuntimeError.  Graph-building entry
    points of the oracle translate, so that a broken width raises what the reference raises."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **kw):
        try:
            return fn(*a, **kw)
        except RuntimeError as e:
            msg = str(e)
            if any(t in msg for t in ("must match the size", "cannot be multiplied", "Sizes of tensors must match", "is invalid for input of size",
                                      "must match except in dimension")):
                raise ValueError("Dimensions must be equal (shape inference): " + msg) from None
            raise
    return wrapped


def default_config(**over):
    c = SimpleNamespace(
        # dims / init (config.py:292-303)
        netLength=16, memDim=512, ctrlDim=512, attDim=512, unsharedCells=False,
        initCtrl="PRM", initMem="PRM", initKBwithQ="NON", addNullWord=False,
        # control (config.py:307-327)
        controlWholeQ=False, controlContinuous=False, controlContextual=False,
        controlInWordsProj=False, controlOutWordsProj=False, controlInputUnshared=False,
        controlInputAct="TANH", controlFeedPrev=False, controlFeedPrevAtt=False,
        controlFeedInputs=False, controlContAct="NON", controlConcatWords=False,
        controlProj=False, controlProjAct="NON",
        # read (config.py:344-362)
        readProjInputs=False, readProjShared=False, readMemAttType="MUL", readMemConcatKB=False,
        readMemConcatProj=False, readMemProj=False, readMemAct="RELU", readCtrl=False,
        readCtrlAttType="MUL", readCtrlConcatKB=False, readCtrlConcatProj=False,
        readCtrlConcatInter=False, readCtrlAct="RELU", readSmryKBProj=False,
        # write (config.py:369-387)
        writeInputs="BOTH", writeConcatMul=False, writeInfoProj=False, writeInfoAct="NON",
        writeSelfAtt=False, writeSelfAttMod="NON", writeMergeCtrl=False, writeMemProj=False,
        writeMemAct="NON", writeGate=False, writeGateShared=False, writeGateBias=1.0,
        # misc (config.py:194-223)
        memoryVariationalDropout=False, memoryDropout=0.85, readDropout=0.85, writeDropout=1.0,
        relu="STD", mulBias=0.0, memoryBN=False, bnDecay=0.999, bnCenter=False, bnScale=False,
        # output unit / classifier (config.py, model.py:512-576)
        outQuestion=False, outQuestionMul=False, outClassifierDims=[512], outputDropout=0.85,
        answerWordsNum=28,
        # question encoder (config.py:178-206, 262-270)
        wrdEmbDim=300, encDim=512, encType="LSTM", encBi=False, encNumLayers=1, encVariationalDropout=False,
        encProj=False, encProjQAct="NON", encInputDropout=0.85, qDropout=0.92, wrdEmbFixed=False,
    )
    for k, v in over.items():
        if not hasattr(c, k):
            raise AttributeError("unknown config flag %r" % k)
        setattr(c, k, v)
    return c


_COMMON = dict(memoryVariationalDropout=True, relu="ELU", outQuestion=True, controlContextual=True, encBi=True,
               readProjInputs=True, readMemConcatKB=True, readMemConcatProj=True, readMemProj=True,
               readCtrl=True, writeMemProj=True)
FLAG_FILES = {
    # configs/args.txt:1-22
    "args": dict(_COMMON, initCtrl="Q", controlInputUnshared=True),
    # configs/args1.txt:1-25
    "args1": dict(_COMMON, initCtrl="PRM", controlFeedPrev=True, controlFeedPrevAtt=True,
                  controlFeedInputs=True, controlContAct="TANH"),
    # configs/args2.txt (cell identical to args.txt; differs in qDropout/stemDropout/bucketing)
    "args2": dict(_COMMON, initCtrl="Q", controlInputUnshared=True),
    # configs/args3.txt:23-24
    "args3": dict(_COMMON, initCtrl="Q", controlInputUnshared=True, writeSelfAtt=True, writeSelfAttMod="CONT"),
    # configs/args4.txt:23
    "args4": dict(_COMMON, initCtrl="Q", controlInputUnshared=True, writeGate=True),
}


def flag_file_config(name, **over):
    kw = dict(FLAG_FILES[name])
    kw.update(over)
    return default_config(**kw)


# -------------------------------------------------------------------------------------------------
# variables: tf.variable_scope / tf.get_variable emulation with the reference's names
# -------------------------------------------------------------------------------------------------
class _ReluAtBoundary(torch.autograd.Function):
    """max(x, 0) whose derivative is `mode` (0 or 1) where |x| <= eps * max|x|: plain ReLU jumps at 0, and a pre-activation
    within round-off of zero falls on either side in an fp32 implementation.  Tests run the oracle under both modes and accept a
    gradient between the two (tests/helpers.py: relu_boundary, hull_err); the forward value is torch.relu's."""

    @staticmethod
    def forward(ctx, x, mode, eps):
        ctx.save_for_backward(x)
        ctx.mode, ctx.eps = mode, eps
        return torch.relu(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        near = x.abs() <= ctx.eps * x.abs().max()
        d = torch.where(near, torch.full_like(x, float(ctx.mode)), (x > 0).to(x.dtype))
        return g * d, None, None


RELU_BOUNDARY = None      # None: torch.relu; (mode, eps): _ReluAtBoundary -- set by tests only


def relu_std(x, flip=False):
    """flip: the derivative at the jump is taken from the OTHER side (PReLU = relu(x) - alpha relu(-x): its two one-sided
    derivatives are 1 and alpha, so the second term runs with the opposite mode)"""
    if RELU_BOUNDARY is None:
        return torch.relu(x)
    return _ReluAtBoundary.apply(x, RELU_BOUNDARY[0] ^ int(flip), RELU_BOUNDARY[1])


class VarStore:
    """name -> tensor.  Missing variables are created with the reference's initialisers
    (xavier-uniform for weights ops.py:20, zeros for biases ops.py:40, N(0,1) for state
    variables mac_cell.py:498-499)."""

    def __init__(self, params=None, generator=None, dtype=torch.float32, requires_grad=False):
        self.params = params if params is not None else {}
        self.gen = generator
        self.dtype = dtype
        self.requires_grad = requires_grad
        self._stack = []
        self._counts = {}          # full scope name -> times opened (tf.variable_scope(None, default_name=...) uniquification)

    @contextmanager
    def scope(self, name, default=False):
        """tf.variable_scope(name) -- or, with default=True, tf.variable_scope(None, default_name=name): the first opening
        inside the current scope is `name`, later ones `name_1`, `name_2`, ... (ops.py:163 "prelu", batch_norm's "BatchNorm");
        the counts of a scope's children are forgotten when it closes, as in TF, so a re-entered cell scope starts over."""
        if default:
            cur = "/".join(self._stack + [name])
            if self._counts.get(cur, 0) > 0:
                idx = 1
                while self._counts.get("%s_%d" % (cur, idx), 0) > 0:
                    idx += 1
                name = "%s_%d" % (name, idx)
        self._stack.append(name)
        full = "/".join(self._stack)
        self._counts[full] = self._counts.get(full, 0) + 1
        try:
            yield
        finally:
            self._stack.pop()
            pre = full + "/"
            for k in list(self._counts):
                if k.startswith(pre):
                    self._counts[k] = 0

    def full(self, name):
        return "/".join(self._stack + [name])

    def get(self, name, shape, init):
        key = self.full(name)
        if key not in self.params:
            if self.gen is None:
                raise KeyError("variable %s missing and no generator to create it" % key)
            shape = tuple(shape)
            if init == "xavier":
                # tf.contrib.layers.xavier_initializer (uniform): limit sqrt(6 / (fan_in + fan_out));
                # 1-D shape (n,): fan_in = fan_out = n -> sqrt(3/n)
                if len(shape) == 1:
                    fan_in = fan_out = shape[0]
                else:
                    fan_in, fan_out = shape[-2], shape[-1]
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                t = (torch.rand(shape, generator=self.gen, dtype=torch.float64) * 2 - 1) * lim
            elif init == "zeros":
                t = torch.zeros(shape, dtype=torch.float64)
            elif init == "ones":
                t = torch.ones(shape, dtype=torch.float64)
            elif init == "normal":
                t = torch.randn(shape, generator=self.gen, dtype=torch.float64)
            elif isinstance(init, float):
                t = torch.full(shape, init, dtype=torch.float64)
            else:
                raise ValueError(init)
            t = t.to(self.dtype)
            if self.requires_grad:
                t.requires_grad_(True)
            self.params[key] = t
        v = self.params[key]
        if tuple(v.shape) != tuple(shape):
            raise ValueError("variable %s has shape %s, expected %s" % (key, tuple(v.shape), tuple(shape)))
        return v


class Ops:
    """The ops.py primitives the cell calls, bound to one config + variable store."""

    def __init__(self, config, store):
        self.config = config
        self.vs = store

    # ---- variables (ops.py:18-42)
    def getWeight(self, shape, name=""):
        with self.vs.scope("weights"):
            return self.vs.get("weight" + name, shape, "xavier")

    def getBias(self, shape, name=""):
        with self.vs.scope("biases"):
            return self.vs.get("bias" + name, shape, "zeros")

    # ---- basics (ops.py:50-78)
    @staticmethod
    def multiply(inp, W):
        inDim, outDim = W.shape
        newDims = tuple(inp.shape[:-1]) + (outDim,)
        out = torch.matmul(inp.reshape(-1, inDim), W)
        return out.reshape(newDims)

    @staticmethod
    def concat(x, y, dim, mul=False, extendY=False):
        if extendY:
            y = y.unsqueeze(-2)
            y = torch.zeros_like(x) + y
        if mul:
            out = torch.cat([x, y, x * y], dim=-1)
            dim *= 3
        else:
            out = torch.cat([x, y], dim=-1)
            dim *= 2
        return out, dim

    # ---- dropout: tf.nn.dropout(x, keep) = x / keep * floor(keep + U)
    @staticmethod
    def dropout(x, keep, mask):
        if mask is None:
            if keep != 1.0:
                raise ValueError("dropout with keep %r needs an explicit mask" % keep)
            return x
        return (x / keep) * mask

    # ---- attention (ops.py:114-150)
    def inter2logits(self, interactions, dim, sumMod="LIN", dropout=1.0, mask=None, name=""):
        with self.vs.scope("inter2logits" + name):
            if sumMod == "SUM":
                return interactions.sum(dim=-1)
            return self.linear(interactions, dim, 1, dropout=dropout, mask=mask, name="logits")

    def inter2att(self, interactions, dim, dropout=1.0, mask=None, name=""):
        with self.vs.scope("inter2att" + name):
            logits = self.inter2logits(interactions, dim, dropout=dropout, mask=mask)
            return torch.softmax(logits, dim=-1)

    @staticmethod
    def att2Smry(attention, features):
        if features is None:
            # tensor * None: TF converts the operand with ops.convert_to_tensor, which rejects None (ops.py:150 with
            # readSmryKBProj but no projected knowledge base, mac_cell.py:271-272)
            raise ValueError("None values not supported.")
        return (attention.unsqueeze(-1) * features).sum(dim=-2)

    # ---- activations (ops.py:161-187)
    def relu(self, inp):
        r = self.config.relu
        if r == "PRM":
            with self.vs.scope("prelu", default=True):
                alpha = self.vs.get("alpha", (inp.shape[-1],), 0.25)
            return relu_std(inp) - alpha * relu_std(-inp, flip=True)
        if r == "ELU":
            return torch.nn.functional.elu(inp)
        if r == "LKY":
            # config.reluAlpha's flag is commented out (config.py:221) -> AttributeError in the reference
            return torch.maximum(inp, self.config.reluAlpha * inp)
        if r == "STD":
            return relu_std(inp)
        # SELU: accepted by argparse (config.py:220) but no branch (ops.py:171-179): `output` unbound
        raise UnboundLocalError("local variable 'output' referenced before assignment")

    def act(self, name, x):
        if name == "NON":
            return x
        if name == "TANH":
            return torch.tanh(x)
        if name == "SIGMOID":
            return torch.sigmoid(x)
        if name == "RELU":
            return self.relu(x)
        if name == "ELU":
            return torch.nn.functional.elu(x)
        raise KeyError(name)

    # ---- sequence mask (ops.py:243-247)
    @staticmethod
    def expMask(seq, seqLength):
        maxLength = seq.shape[-1]
        m = (torch.arange(maxLength).unsqueeze(0) < seqLength.unsqueeze(1)).to(seq.dtype)
        return seq + (1 - m) * (-inf)

    # ---- linear (ops.py:298-333)
    def linear(self, inp, inDim, outDim, dropout=1.0, mask=None, addBias=True, bias=0.0, act="NON",
               actLayer=True, actDropout=1.0, name=""):
        with self.vs.scope("linearLayer" + name):
            W = self.getWeight((inDim, outDim) if outDim > 1 else (inDim,))
            b = self.getBias((outDim,) if outDim > 1 else ()) + bias
            inp = self.dropout(inp, dropout, mask)
            if outDim > 1:
                output = self.multiply(inp, W)
            else:
                output = (inp * W).sum(dim=-1)
            if addBias:
                output = output + b
            output = self.act(act, output)
            if act != "NON" and actLayer:
                output = self.linear(output, outDim, outDim, dropout=actDropout, addBias=addBias, act="NON",
                                     actLayer=False, name=name + "_2")
        return output

    def FCLayer(self, features, dims, dropout=1.0, masks=None, act="RELU"):
        layersNum = len(dims) - 1
        for i in range(layersNum):
            features = self.linear(features, dims[i], dims[i + 1], name="fc_%d" % i, dropout=dropout,
                                   mask=None if masks is None else masks[i])
            if i < layersNum - 1:
                features = self.act(act, features)
        return features

    # ---- multiplicative interaction (ops.py:668-725)
    def mul(self, x, y, dim, dropout=1.0, proj=None, interMod="MUL", concat=None, mulBias=None, extendY=True,
            name="", masks=None):
        masks = masks or {}
        with self.vs.scope("mul" + name):
            origVals = {"x": x, "y": y, "dim": dim}
            x = self.dropout(x, dropout, None)
            y = self.dropout(y, dropout, None)
            projVals = None
            if proj is not None:
                x = self.dropout(x, proj.get("dropout", 1.0), masks.get("x"))
                y = self.dropout(y, proj.get("dropout", 1.0), masks.get("y"))
                if proj["shared"]:
                    xName = yName = "proj"
                else:
                    xName, yName = "projX", "projY"
                x = self.linear(x, dim, proj["dim"], name=xName)
                y = self.linear(y, dim, proj["dim"], name=yName)
                dim = proj["dim"]
                projVals = {"x": x, "y": y, "dim": dim}
                proj["x"], proj["y"] = x, y
            if extendY:
                y = y.unsqueeze(-2)
                y = torch.zeros_like(x) + y
            if interMod == "MUL":
                if mulBias is None:
                    mulBias = self.config.mulBias
                output = (x + mulBias) * (y + mulBias)
            elif interMod == "DIAG":
                # ops.py:704-707 assigns to `activations`, leaving `output` unbound
                raise UnboundLocalError("local variable 'output' referenced before assignment")
            elif interMod == "BL":
                W = self.getWeight((dim, dim))
                b = self.getBias((dim,))
                output = self.multiply(x, W) * y + b
            else:  # "ADD"
                output = torch.tanh(x + y)
            if concat is not None:
                if concat.get("proj", False):
                    if projVals is None:
                        # ops.py:691,716: projVals is only bound inside `if proj is not None`
                        raise UnboundLocalError("local variable 'projVals' referenced before assignment")
                    concatVals = projVals
                else:
                    concatVals = origVals
                if concat.get("x", False):
                    output = torch.cat([output, concatVals["x"]], dim=-1)
                    dim += concatVals["dim"]
                if concat.get("y", False):
                    # ops.py:721-723 calls `ops.concat` from inside ops.py
                    raise NameError("name 'ops' is not defined")
        return output, dim

    # ---- variational dropout (ops.py:1054-1067)
    @staticmethod
    def applyVarDpMask(inp, mask, keepProb):
        return (inp / keepProb) * mask

    # ---- tf.contrib.layers.batch_norm(updates_collections=None) on the last axis (mac_cell.py:370-373)
    def batch_norm(self, inputs, decay, center, scale, is_training, epsilon=0.001):
        with self.vs.scope("BatchNorm", default=True):
            c = inputs.shape[-1]
            beta = self.vs.get("beta", (c,), "zeros") if center else None
            gamma = self.vs.get("gamma", (c,), "ones") if scale else None
            mm = self.vs.get("moving_mean", (c,), "zeros")
            mv = self.vs.get("moving_variance", (c,), "ones")
            if is_training:
                axes = list(range(inputs.dim() - 1))
                mean = inputs.mean(dim=axes)
                var = ((inputs - mean) ** 2).mean(dim=axes)
                with torch.no_grad():      # moving averages are updated in place, outside the gradient
                    mm.sub_((1 - decay) * (mm - mean))
                    mv.sub_((1 - decay) * (mv - var))
            else:
                mean, var = mm.detach(), mv.detach()
            out = (inputs - mean) / torch.sqrt(var + epsilon)
            if gamma is not None:
                out = out * gamma
            if beta is not None:
                out = out + beta
        return out


# -------------------------------------------------------------------------------------------------
# the cell
# -------------------------------------------------------------------------------------------------
class MACCellOracle:
    """mac_cell.py:30-592, same constructor / zero_state / __call__ / attribute surface."""

    def __init__(self, config, store, vecQuestions, questionWords, questionCntxWords, questionLengths, knowledgeBase,
                 memoryDropout, readDropout, writeDropout, batchSize, train, mask_fn=None):
        self.config = config
        self.vs = store
        self.ops = Ops(config, store)
        self.vecQuestions = vecQuestions
        self.questionWords = questionWords
        self.questionCntxWords = questionCntxWords
        self.questionLengths = questionLengths
        self.knowledgeBase = knowledgeBase
        self.dropouts = {"memory": memoryDropout, "read": readDropout, "write": writeDropout}
        self.batchSize = batchSize
        self.train = train
        self.mask_fn = mask_fn or (lambda site, step, shape: None)
        self.none = torch.zeros((batchSize, 1), dtype=vecQuestions.dtype)
        self.iteration = 0

    def _mask(self, site, shape, keep):
        if keep == 1.0:
            return None
        m = self.mask_fn(site, self.iteration, tuple(shape))
        if m is None:
            raise ValueError("keep < 1 at site %d without a mask" % site)
        return torch.as_tensor(m, dtype=self.vecQuestions.dtype)

    # ---- control (mac_cell.py:133-187)
    def control(self, controlInput, inWords, outWords, questionLengths, control, contControl=None, name=""):
        config, ops = self.config, self.ops
        with self.vs.scope("control" + name):
            dim = config.ctrlDim
            newContControl = controlInput
            if config.controlFeedPrev:
                newContControl = control if config.controlFeedPrevAtt else contControl
                if config.controlFeedInputs:
                    newContControl = torch.cat([newContControl, controlInput], dim=-1)
                    dim += config.ctrlDim
                newContControl = ops.linear(newContControl, dim, config.ctrlDim, act=config.controlContAct,
                                            name="contControl")
                dim = config.ctrlDim
            interactions = newContControl.unsqueeze(1) * inWords
            if config.controlConcatWords:
                interactions = torch.cat([interactions, inWords], dim=-1)
                dim += config.ctrlDim
            if config.controlProj:
                interactions = ops.linear(interactions, dim, config.ctrlDim, act=config.controlProjAct)
                dim = config.ctrlDim
            logits = ops.inter2logits(interactions, dim)
            attention = torch.softmax(ops.expMask(logits, questionLengths), dim=-1)
            self.attentions["question"].append(attention)
            newControl = ops.att2Smry(attention, outWords)
            if config.controlContinuous:
                newControl = newContControl
        return newControl, newContControl

    # ---- read (mac_cell.py:209-277)
    def read(self, knowledgeBase, memory, control, name=""):
        config, ops = self.config, self.ops
        with self.vs.scope("read" + name):
            dim = config.memDim
            if config.memoryVariationalDropout:
                memory = ops.applyVarDpMask(memory, self.memDpMask, self.dropouts["memory"]) \
                    if self.memDpMask is not None else memory
            else:
                memory = ops.dropout(memory, self.dropouts["memory"],
                                     self._mask(dh.SITE_MEM, memory.shape, self.dropouts["memory"]))
            proj = None
            if config.readProjInputs:
                proj = {"dim": config.attDim, "shared": config.readProjShared, "dropout": self.dropouts["read"]}
                dim = config.attDim
            concat = {"x": config.readMemConcatKB, "proj": config.readMemConcatProj}
            masks = {"x": self._mask(dh.SITE_READ_KB, knowledgeBase.shape, self.dropouts["read"]),
                     "y": self._mask(dh.SITE_READ_MEM, memory.shape, self.dropouts["read"])} if proj else None
            interactions, interDim = ops.mul(x=knowledgeBase, y=memory, dim=config.memDim, proj=proj, concat=concat,
                                             interMod=config.readMemAttType, name="memInter", masks=masks)
            projectedKB = proj.get("x") if proj else None
            if config.readMemProj:
                interactions = ops.linear(interactions, interDim, dim, act=config.readMemAct, name="memKbProj")
            else:
                dim = interDim
            if config.readCtrl:
                if config.ctrlDim != dim:
                    # mac_cell.py:246 references an undefined `ctrlDim`
                    raise NameError("name 'ctrlDim' is not defined")
                interactions, interDim = ops.mul(interactions, control, dim, interMod=config.readCtrlAttType,
                                                 concat={"x": config.readCtrlConcatInter}, name="ctrlInter")
                if config.readCtrlConcatKB:
                    if config.readCtrlConcatProj:
                        addedInp, addedDim = projectedKB, config.attDim
                    else:
                        addedInp, addedDim = knowledgeBase, config.memDim
                    if addedInp is None:
                        # tf.concat([interactions, None]): readCtrlConcatProj without readProjInputs (mac_cell.py:252-258)
                        raise ValueError("None values not supported.")
                    interactions = torch.cat([interactions, addedInp], dim=-1)
                    dim += addedDim
                interactions = ops.act(config.readCtrlAct, interactions)
            if interactions.shape[-1] != dim:
                # mac_cell.py:248-250 drops the width ops.mul returns under readCtrlConcatInter: the [dim] logits weight
                # meets a wider tensor and TF's shape inference raises ValueError while building the graph
                raise ValueError("Dimensions must be equal, but are %d and %d" % (interactions.shape[-1], dim))
            attention = ops.inter2att(interactions, dim, dropout=self.dropouts["read"],
                                      mask=self._mask(dh.SITE_READ_ATT, interactions.shape, self.dropouts["read"]))
            self.attentions["kb"].append(attention)
            if config.readSmryKBProj:
                knowledgeBase = projectedKB
            information = ops.att2Smry(attention, knowledgeBase)
        return information

    # ---- write (mac_cell.py:305-375)
    def write(self, memory, info, control, contControl=None, name=""):
        config, ops = self.config, self.ops
        with self.vs.scope("write" + name):
            if config.writeInfoProj:
                info = ops.linear(info, config.memDim, config.memDim, name="info")
            info = ops.act(config.writeInfoAct, info)
            if config.writeSelfAtt:
                selfControl = control
                if config.writeSelfAttMod == "CONT":
                    selfControl = contControl
                selfControl = ops.linear(selfControl, config.ctrlDim, config.ctrlDim, name="ctrlProj")
                interactions = self.controls * selfControl.unsqueeze(1)
                attention = ops.inter2att(interactions, config.ctrlDim, name="selfAttention")
                self.attentions["self"].append(attention)
                selfSmry = ops.att2Smry(attention, self.memories)
            newMemory, dim = memory, config.memDim
            if config.writeInputs == "INFO":
                newMemory = info
            elif config.writeInputs == "SUM":
                newMemory = newMemory + info
            elif config.writeInputs == "BOTH":
                newMemory, dim = ops.concat(newMemory, info, dim, mul=config.writeConcatMul)
            if config.writeSelfAtt:
                newMemory = torch.cat([newMemory, selfSmry], dim=-1)
                dim += config.memDim
            if config.writeMergeCtrl:
                newMemory = torch.cat([newMemory, control], dim=-1)
                dim += config.memDim
            if config.writeMemProj or (dim != config.memDim):
                newMemory = ops.linear(newMemory, dim, config.memDim, name="newMemory")
            newMemory = ops.act(config.writeMemAct, newMemory)
            if config.writeGate:
                gateDim = config.memDim
                if config.writeGateShared:
                    gateDim = 1
                z = torch.sigmoid(ops.linear(control, config.ctrlDim, gateDim, name="gate", bias=config.writeGateBias))
                if gateDim == 1 and z.shape[0] != newMemory.shape[-1]:
                    # ops.linear with outDim == 1 returns [B] (ops.py:317); [B,d] * [B] does not broadcast
                    raise ValueError("Dimensions must be equal, but are %d and %d" % (newMemory.shape[-1], z.shape[0]))
                self.attentions["gate"].append(z)
                newMemory = newMemory * z + memory * (1 - z)
            if config.memoryBN:
                newMemory = ops.batch_norm(newMemory, config.bnDecay, config.bnCenter, config.bnScale, self.train)
        return newMemory

    # ---- one step (mac_cell.py:420-480)
    @_tf_shape_errors
    def __call__(self, inputs, state, scope=None):
        config, ops = self.config, self.ops
        scope = scope or "MACCell"
        with self.vs.scope(scope):
            control, memory = state
            inputName = "qInput"
            inputNameU = "qInputU"
            if config.controlInputUnshared:
                inputNameU = "qInput%d" % self.iteration
            cellName = ""
            if config.unsharedCells:
                cellName = str(self.iteration)
            controlInput = ops.linear(self.vecQuestions, config.ctrlDim, config.ctrlDim, name=inputName)
            if config.controlInputAct == "RELU" and config.relu == "PRM" and self.iteration > 0:
                # mac_cell.py:445 applies the activation directly in the cell's scope, which is opened with reuse=None
                # (mac_cell.py:422): the second step asks tf.get_variable for MACCell/prelu/alpha again
                raise ValueError("Variable %s already exists, disallowed. Did you mean to set reuse=True in VarScope?"
                                 % self.vs.full("prelu/alpha"))
            controlInput = ops.act(config.controlInputAct, controlInput)
            controlInput = ops.linear(controlInput, config.ctrlDim, config.ctrlDim, name=inputNameU)
            newControl, self.contControl = self.control(controlInput, self.inWords, self.outWords,
                                                        self.questionLengths, control, self.contControl, name=cellName)
            if config.controlWholeQ:
                newControl = self.vecQuestions
            info = self.read(self.knowledgeBase, memory, newControl, name=cellName)
            if config.writeDropout < 1.0:
                info = ops.dropout(info, self.dropouts["write"],
                                   self._mask(dh.SITE_WRITE_INFO, info.shape, self.dropouts["write"]))
            newMemory = self.write(memory, info, newControl, self.contControl, name=cellName)
            self.controls = torch.cat([self.controls, newControl.unsqueeze(1)], dim=1)
            self.memories = torch.cat([self.memories, newMemory.unsqueeze(1)], dim=1)
            self.infos = torch.cat([self.infos, info.unsqueeze(1)], dim=1)
        return self.none, (newControl, newMemory)

    # ---- state init (mac_cell.py:496-505)
    def initState(self, name, dim, initType, batchSize):
        if initType == "PRM":
            prm = self.vs.get(name, (dim,), "normal")
            return prm.unsqueeze(0).repeat(batchSize, 1)
        if initType == "ZERO":
            return torch.zeros((batchSize, dim), dtype=self.vecQuestions.dtype)
        return self.vecQuestions

    # ---- zero_state (mac_cell.py:539-592)
    def zero_state(self, batchSize):
        config, ops = self.config, self.ops
        self.attentions = {"kb": [], "question": [], "self": [], "gate": []}
        initialControl = self.initState("initCtrl", config.ctrlDim, config.initCtrl, batchSize)
        initialMemory = self.initState("initMem", config.memDim, config.initMem, batchSize)
        self.controls = initialControl.unsqueeze(1)
        self.memories = initialMemory.unsqueeze(1)
        self.infos = initialMemory.unsqueeze(1)
        self.contControl = initialControl
        if config.initKBwithQ != "NON":
            # mac_cell.py:564 passes `expandY=`, ops.concat's parameter is `extendY` (ops.py:65)
            raise TypeError("concat() got an unexpected keyword argument 'expandY'")
        words = self.questionCntxWords if config.controlContextual else self.questionWords
        if config.addNullWord:
            # mac_cell.py:519,574: addNullWord lacks `self`, questionLengths is unbound
            raise UnboundLocalError("local variable 'questionLengths' referenced before assignment")
        self.inWords = self.outWords = words
        if config.controlInWordsProj or config.controlOutWordsProj:
            pWords = ops.linear(words, config.ctrlDim, config.ctrlDim, name="wordsProj")
            self.inWords = pWords if config.controlInWordsProj else words
            self.outWords = pWords if config.controlOutWordsProj else words
        self.memDpMask = None
        if config.memoryVariationalDropout and self.dropouts["memory"] != 1.0:
            self.iteration = 0
            self.memDpMask = self._mask(dh.SITE_MEM_VAR, (batchSize, config.memDim), self.dropouts["memory"])
        return (initialControl, initialMemory)


def hash_mask_fn(seed, keeps, b0=0, word=0):
    """mask_fn backed by the stateless stream (oracle/dropout_hash.py), keyed like the HIP path (word: the run's mask word,
    include/macx.h macx_dropout.mask_word)."""
    site_keep = {dh.SITE_MEM_VAR: keeps[0], dh.SITE_MEM: keeps[0], dh.SITE_READ_KB: keeps[1],
                 dh.SITE_READ_MEM: keeps[1], dh.SITE_READ_ATT: keeps[1], dh.SITE_WRITE_INFO: keeps[2]}

    def fn(site, step, shape):
        st = 0 if site == dh.SITE_MEM_VAR else step
        return dh.mask_for(seed, site, st, site_keep[site], shape, b0=b0, word=word)

    return fn


def mac_network(config, store, vecQuestions, questionWords, questionCntxWords, questionLengths, knowledgeBase,
                train=False, mask_fn=None, keeps=None):
    """MACnet.MACnetwork (model.py:428-489): build the cell, zero_state, netLength calls.
    Returns (finalControl, finalMemory, cell)."""
    B = vecQuestions.shape[0]
    if keeps is None:
        keeps = (config.memoryDropout, config.readDropout, config.writeDropout) if train else (1.0, 1.0, 1.0)
    with store.scope("MACnetwork"):
        cell = MACCellOracle(config, store, vecQuestions, questionWords, questionCntxWords, questionLengths,
                             knowledgeBase, keeps[0], keeps[1], keeps[2], B, train, mask_fn=mask_fn)
        state = cell.zero_state(B)
        none = torch.zeros((B, 1), dtype=vecQuestions.dtype)
        for i in range(config.netLength):
            cell.iteration = i
            _, state = cell(none, state)
    return state[0], state[1], cell


@_tf_shape_errors
def output_classifier(config, store, memory, vecQuestions, output_keep=1.0, masks=None):
    """outputOp (model.py:512-528) + classifier (model.py:547-576): logits [B, answerWordsNum]."""
    ops = Ops(config, store)
    with store.scope("outputUnit"):
        features, dim = memory, config.memDim
        if config.outQuestion:
            eVecQuestions = ops.linear(vecQuestions, config.ctrlDim, config.memDim, name="outQuestion")
            features, dim = ops.concat(features, eVecQuestions, config.memDim, mul=config.outQuestionMul)
    with store.scope("classifier"):
        dims = [dim] + list(config.outClassifierDims) + [config.answerWordsNum]
        logits = ops.FCLayer(features, dims, dropout=output_keep, masks=masks)
    return logits


def stem_cnn(config, store, images, H, W, keep=1.0, masks=None):
    """MACnet.stem (model.py:165-204) -> ops.CNNLayer / ops.cnn (ops.py:380-438), default 2 x conv3x3 SAME.
    images [B, H*W, C] (NHWC flattened); masks: [layer-0 input mask, layer-1 input mask] when keep < 1."""
    ops = Ops(config, store)
    B, N, inDim = images.shape
    dims = [inDim] + [getattr(config, "stemDim", 512)] * 1 + [config.memDim]
    feats = images.reshape(B, H, W, inDim)
    with store.scope("stem"):
        for i in range(2):
            with store.scope("cnnLayercnn_%d" % i):
                with store.scope("kernels"):
                    kshape = (3, 3, dims[i], dims[i + 1])
                    key = store.full("kernel")
                    if key not in store.params:
                        lim = math.sqrt(6.0 / (9 * dims[i] + 9 * dims[i + 1]))
                        t = (torch.rand(kshape, generator=store.gen, dtype=torch.float64) * 2 - 1) * lim
                        store.params[key] = t.to(store.dtype).requires_grad_(store.requires_grad)
                    kernel = store.params[key]
                b = ops.getBias((dims[i + 1],))
                m = None if masks is None else masks[i].reshape(B, H, W, dims[i])
                inp = ops.dropout(feats, keep, m)
                out = torch.nn.functional.conv2d(inp.permute(0, 3, 1, 2), kernel.permute(3, 2, 0, 1).contiguous(), padding=1)   # SAME, stride 1
                out = out.permute(0, 2, 3, 1) + b
                feats = ops.act("RELU", out)
    return feats.reshape(B, N, dims[-1])


@_tf_shape_errors
def question_encoder(config, store, questions, lengths, vocab, keep_input=1.0, keep_question=1.0, masks=None):
    """qEmbeddingsOp (model.py:207-219) + encoder (model.py:279-307) -> ops.RNNLayer -> biRNNLayer / fwRNNLayer (ops.py:798-950)
    with encType = LSTM and no variational dropout; the output projections of model.py:785-787 (projWords = projQuestion =
    encDim != ctrlDim or encProj) included.
    questions [B,S] int (0 = pad -> the zero row prepended at model.py:216; id i >= 1 -> emb[i-1]); lengths [B].
    --encBi: tf.nn.bidirectional_dynamic_rnn restated -- BasicLSTMCell(encDim / 2) per direction under
    encoder/birnnLayer/bidirectional_rnn/{fw,bw}; the backward cell runs over the length-reversed question
    (array_ops.reverse_sequence) and its outputs are reversed back.  Without it: tf.nn.dynamic_rnn of one BasicLSTMCell(encDim)
    under encoder/rnnLayer/rnn.  Past a question's end dynamic_rnn emits zeros and copies the state through.
    BasicLSTMCell (TF1 rnn_cell_impl): gates i, j, f, o = split([x, h] @ kernel + bias); c' = c * sigmoid(f + 1) +
    sigmoid(i) * tanh(j); h' = tanh(c') * sigmoid(o).
    masks: [input mask [B,S,E], question mask [B,encDim]] when the keeps are < 1.
    Returns questionCntxWords [B,S,encDim or ctrlDim], vecQuestions [B,encDim or ctrlDim]."""
    ops = Ops(config, store)
    B, S = questions.shape
    bi = bool(config.encBi)
    if config.encType != "LSTM" or config.encVariationalDropout:
        raise NotImplementedError("oracle: encType LSTM without variational dropout only")
    E, h = config.wrdEmbDim, (config.encDim // 2 if bi else config.encDim)
    dt = store.dtype
    with store.scope("qEmbeddings"):
        emb = store.get("emb", (vocab, E), "normal")     # embInit: random rows (preprocess.py initEmbRandom), any init works here
    table = torch.cat([torch.zeros((1, E), dtype=dt), emb], dim=0)
    x = table[questions.long()]
    scopes = ("birnnLayer", "bidirectional_rnn") if bi else ("rnnLayer", "rnn")
    if config.encNumLayers > 1:
        # model.py:295-298 builds every layer from `questions` under the same scope (RNNLayer's "rnnLayer<name>" scope closes
        # before the layer is built, ops.py:938-950): the second layer asks for the first one's variables again
        raise ValueError("Variable encoder/%s/%s/%sbasic_lstm_cell/kernel already exists, disallowed. Did you mean to set "
                         "reuse=True in VarScope?" % (scopes[0], scopes[1], "fw/" if bi else ""))
    x = ops.dropout(x, keep_input, None if masks is None else masks[0])
    outs, finals = [], []
    with store.scope("encoder"), store.scope(scopes[0]), store.scope(scopes[1]):
        for direction in (("fw", "bw") if bi else ("",)):
            with contextlib.ExitStack() as es:
                if direction:
                    es.enter_context(store.scope(direction))
                es.enter_context(store.scope("basic_lstm_cell"))
                kernel = store.get("kernel", (E + h, 4 * h), "xavier")     # glorot_uniform: TF's default for get_variable
                bias = store.get("bias", (4 * h,), "zeros")
            hs = torch.zeros((B, h), dtype=dt)
            cs = torch.zeros((B, h), dtype=dt)
            out = [[None] * S for _ in range(B)]
            for tau in range(S):
                active = (tau < lengths).to(dt).unsqueeze(1)                                  # [B,1]
                pos = torch.full((B,), tau, dtype=torch.long) if direction != "bw" else (lengths.long() - 1 - tau).clamp(min=0)
                xt = x[torch.arange(B), pos]                                                  # [B,E]
                z = torch.cat([xt, hs], dim=1) @ kernel + bias
                i, j, f, o = z.split(h, dim=1)
                cn = cs * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
                hn = torch.tanh(cn) * torch.sigmoid(o)
                hs = active * hn + (1 - active) * hs
                cs = active * cn + (1 - active) * cs
                for b in range(B):
                    if tau < int(lengths[b]):
                        out[b][int(pos[b])] = hn[b]
            zero = torch.zeros(h, dtype=dt)
            outs.append(torch.stack([torch.stack([o_ if o_ is not None else zero for o_ in row]) for row in out]))
            finals.append(hs)
    words = torch.cat(outs, dim=-1)
    vecQ = torch.cat(finals, dim=-1)
    vecQ = ops.dropout(vecQ, keep_question, None if masks is None else masks[1])
    if config.encDim != config.ctrlDim or config.encProj:                  # model.py:785-787, 301-306
        with store.scope("encoder"):
            words = ops.linear(words, config.encDim, config.ctrlDim, name="projCW")
            vecQ = ops.linear(vecQ, config.encDim, config.ctrlDim, act=config.encProjQAct, name="projQ")
    return words, vecQ


def answer_loss_and_pred(logits, answers):
    """addAnswerLossOp (model.py:593-599) + addPredOp (model.py:603-612)."""
    losses = torch.nn.functional.cross_entropy(logits, answers, reduction="none")
    return losses.mean(), logits.argmax(dim=-1).to(torch.int32)


# -------------------------------------------------------------------------------------------------
# synthetic inputs of the CLEVR shape
# -------------------------------------------------------------------------------------------------
def synthetic_inputs(B, S, N, d, seed=1234, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    vecQ = (torch.rand((B, d), generator=g, dtype=torch.float64) * 2 - 1)
    words = (torch.rand((B, S, d), generator=g, dtype=torch.float64) * 2 - 1)
    lengths = torch.randint(3, S + 1, (B,), generator=g, dtype=torch.int32)
    lengths[0] = S
    m = (torch.arange(S).unsqueeze(0) < lengths.unsqueeze(1)).to(torch.float64)
    words = words * m.unsqueeze(-1)
    kb = torch.nn.functional.elu(torch.randn((B, N, d), generator=g, dtype=torch.float64))
    return vecQ.to(dtype), words.to(dtype), lengths, kb.to(dtype)
