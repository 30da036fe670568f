"""The generic option path of the MAC cell as a COMPILED PLAN.

`compile_cell(config, netLength)` turns an option set into data, once: an ordered variable table (reference names, shapes,
initialisers -- the order TensorFlow would create them in) and, for `zero_state` and for each of the p steps, a flat list of
primitive operations over numbered value slots -- dense layers, broadcast products, activations, softmax, attention-weighted
sums, dropout sites, concatenations.  Nothing of the configuration is consulted after that: `Segment.run` executes a list,
and differentiates it by walking the same list backwards with one backward rule per primitive.  A whole `zero_state` or cell
step is therefore ONE node of PyTorch's autograd graph (the interface the reference's loop needs: state in, state out), not
one node per reference op, and adjacent primitives are visible to a later fusion pass as data.

Every primitive's arithmetic is a kernel of libmacx.so (generic.k_*: macx_linear / macx_h2_gemm / macx_wgrad / macx_op_*);
PyTorch owns memory, streams, concat / stack copies.  What the option values mean is the reference's (mac_cell.py:133-375,
420-480, 539-592; ops.py:161-187, 298-333, 668-725) and is checked against it by executing the reference
(tests/test_reference_exec.py -> oracle -> tests/test_generic_host.py, tests/test_gpu_generic.py); how it is organised --
an option compiler emitting a dataflow plan, an interpreter with its own reverse sweep -- is not.
"""
import collections
from contextlib import ExitStack, contextmanager

import torch

from . import _lib
from . import generic as G
from .options import UnsupportedOptions, get

Node = collections.namedtuple("Node", "op out ins attr")
VarSpec = collections.namedtuple("VarSpec", "shape init")
SITE = dict(mem_var=1, mem=2, read_kb=3, read_mem=4, read_att=5, write_info=6)
NOGRAD_OPS = ("fill", "zeros", "bn_update")


# ---------------------------------------------------------------------------------------------------------------------
# naming: TF variable scopes, names only (default-name scopes count up within one opening of their parent)
# ---------------------------------------------------------------------------------------------------------------------
class ScopeNames:
    def __init__(self):
        self.stack, self.opened = [], {}

    @contextmanager
    def __call__(self, name, default=False):
        if default:
            base = "/".join(self.stack + [name])
            if self.opened.get(base, 0):
                k = 1
                while self.opened.get("%s_%d" % (base, k), 0):
                    k += 1
                name = "%s_%d" % (name, k)
        self.stack.append(name)
        full = "/".join(self.stack)
        self.opened[full] = self.opened.get(full, 0) + 1
        try:
            yield
        finally:
            self.stack.pop()
            for key in [k for k in self.opened if k.startswith(full + "/")]:
                self.opened[key] = 0

    def full(self, leaf):
        return "/".join(self.stack + [leaf])


# ---------------------------------------------------------------------------------------------------------------------
# one segment of the plan: feeds -> nodes -> results
# ---------------------------------------------------------------------------------------------------------------------
class Segment:
    def __init__(self, label):
        self.label = label
        self.nodes, self.feeds, self.vars, self.results = [], {}, {}, collections.OrderedDict()
        self.n = 0

    def slot(self):
        self.n += 1
        return self.n - 1

    def feed(self, name):
        if name not in self.feeds:
            self.feeds[name] = self.slot()
        return self.feeds[name]

    def var(self, full_name):
        if full_name not in self.vars:
            self.vars[full_name] = self.slot()
        return self.vars[full_name]

    def emit(self, kind_, *ins, **attr):
        out = self.slot()
        self.nodes.append(Node(kind_, out, tuple(ins), attr))
        return out


class CellPlan:
    """variables (ordered: reference creation order) + the segments `init`, `step[0..p-1]`"""

    def __init__(self):
        self.variables = collections.OrderedDict()
        self.init = None
        self.steps = []

    def describe(self):
        """the plan as text (one line per primitive): what a fixture of the plan, or a reader, looks at"""
        lines = ["var %-90s %s %s" % (k, tuple(v.shape), v.init) for k, v in self.variables.items()]
        for seg in [self.init] + self.steps:
            lines.append("segment %s: feeds %s -> results %s" % (seg.label, sorted(seg.feeds), list(seg.results)))
            for nd in seg.nodes:
                lines.append("  %%%d = %s(%s) %s" % (nd.out, nd.op, ", ".join("%%%d" % i for i in nd.ins),
                                                   " ".join("%s=%s" % kv for kv in sorted(nd.attr.items()))))
        return "\n".join(lines)


# ---------------------------------------------------------------------------------------------------------------------
# the option compiler
# ---------------------------------------------------------------------------------------------------------------------
class _Emitter:
    """layer-level vocabulary on top of Segment.emit; keeps the variable table, the scope names and every slot's last-axis width"""

    FEED_WIDTH = {"knowledgeBase": "memDim", "memory": "memDim", "mem_mask": "memDim", "memories": "memDim", "info": "memDim",
                  "control": "ctrlDim", "control_input": "ctrlDim",
                  "cont_control": "ctrlDim", "controls": "ctrlDim", "vecQuestions": "ctrlDim", "words": "ctrlDim", "in_words": "ctrlDim",
                  "out_words": "ctrlDim"}

    def __init__(self, config, plan):
        self.cfg, self.plan, self.names, self.seg, self.w = config, plan, ScopeNames(), None, {}

    def opt(self, key):
        return get(self.cfg, key)

    def begin(self, seg):
        self.seg, self.w = seg, {}
        return seg

    def feed(self, name):
        s = self.seg.feed(name)
        if name in self.FEED_WIDTH:
            self.w[s] = int(self.opt(self.FEED_WIDTH[name]))
        return s

    def emit(self, width_, kind_, *ins, **attr):
        s = self.seg.emit(kind_, *ins, **attr)
        self.w[s] = width_
        return s

    # -- variables
    def variable(self, leaf, shape, init):
        name = self.names.full(leaf)
        shape = tuple(int(v) for v in shape)
        known = self.plan.variables.get(name)
        if known is None:
            self.plan.variables[name] = VarSpec(shape, init)
        elif known.shape != shape:
            raise ValueError("variable %s has shape %s, expected %s" % (name, known.shape, shape))
        return self.seg.var(name)

    # -- activations: the names of config.py -> kernel codes; "RELU" is whatever --relu says (ops.py:161-187)
    def activation(self, kind, x):
        if kind == "NON":
            return x
        if kind == "RELU":
            flavour = self.opt("relu")
            if flavour == "PRM":
                with self.names("prelu", default=True):
                    slope = self.variable("alpha", (self.w[x],), 0.25)
                return self.emit(self.w[x], "act", x, slope, code=G.ACT_PRELU)
            if flavour == "LKY":
                raise AttributeError("'Config' object has no attribute 'reluAlpha'")
            if flavour == "SELU":
                raise UnboundLocalError("local variable 'output' referenced before assignment")
            kind = "ELU" if flavour == "ELU" else "RELU"
        return self.emit(self.w[x], "act", x, code=_lib.ACT[kind])

    # -- a dense layer family member "linearLayer<tag>" (ops.py:298-333): K -> n (n = 1: a dot product per row), optional
    #    input dropout, activation, and -- when there is an activation -- the stacked "<tag>_2" layer
    def dense(self, x, k, n, tag="", act="NON", drop=None, const_bias=0.0, stacked=True):
        with self.names("linearLayer" + tag):
            with self.names("weights"):
                w = self.variable("weight", (k, n) if n > 1 else (k,), "xavier")
            with self.names("biases"):
                b = self.variable("bias", (n,) if n > 1 else (), "zeros")
            if drop is not None:
                x = self.dropout(x, drop)
            y = self.emit(n if n > 1 else None, "linear" if n > 1 else "rowdot", x, w, b, const=float(const_bias))
            y = self.activation(act, y)
            if act != "NON" and stacked:
                y = self.dense(y, n, n, tag=tag + "_2", stacked=False)
        return y

    def dropout(self, x, site):
        """site = (stream, name of the keep probability, step): a node even when the run's keep turns out to be 1 (it then passes
        its input through) -- the plan does not depend on train / eval"""
        stream, keep, step = site
        return self.emit(self.w.get(x), "drop", x, site=SITE[stream], step=step, keep=keep)

    def pair(self, op, a, b, spread="same", scale=1.0):
        """scale * (a op b); spread: how b is broadcast over a -- same | mid ([B,N,d] x [B,d]) | channel ([..,c] x [c]) | row"""
        return self.emit(self.w.get(a), "bin", a, b, op=op, spread=spread, scale=float(scale))

    def join(self, parts):
        return parts[0] if len(parts) == 1 else self.emit(sum(self.w[q] for q in parts), "cat", *parts)

    # -- attention: d -> 1 scores ("inter2logits/linearLayerlogits", inside "inter2att<tag>" when the unit wraps it), softmax
    #    (behind the length mask when `lengths` is fed), then the weights' sum over `values`
    def attend(self, x, width, values, tag="", drop=None, lengths=None, wrap=True):
        path = (["inter2att" + tag] if wrap else []) + ["inter2logits"]
        with ExitStack() as scopes:
            for part in path:
                scopes.enter_context(self.names(part))
            scores = self.dense(x, width, 1, tag="logits", drop=drop)
        weights = self.emit(None, "softmax", scores, *(() if lengths is None else (lengths,)))
        return weights, self.emit(self.w.get(values), "wsum", weights, values)

    # -- the pairwise interaction of ops.mul (ops.py:668-725): optional projections of both sides (with their dropout sites),
    #    MUL / bilinear / additive combination, optional concat of an operand
    def interact(self, x, y, width, tag, mode="MUL", project=None, keep_x=False, keep_projected=False, sites=None):
        out = {}
        with self.names("mul" + tag):
            raw_x, raw_w = x, width
            if project is not None:
                if sites:
                    x, y = self.dropout(x, sites[0]), self.dropout(y, sites[1])
                tx, ty = ("proj", "proj") if project["shared"] else ("projX", "projY")
                x = self.dense(x, width, project["width"], tag=tx)
                y = self.dense(y, width, project["width"], tag=ty)
                width = project["width"]
                out["projected_x"] = x
            projected = x
            offset = self.opt("mulBias")
            if mode == "MUL":
                if offset != 0.0:
                    x = self.emit(self.w[x], "shift", x, by=float(offset))
                    y = self.emit(self.w[y], "shift", y, by=float(offset))
                z = self.pair("mul", x, y, "mid")
            elif mode == "DIAG":
                raise UnboundLocalError("local variable 'output' referenced before assignment")
            elif mode == "BL":
                with self.names("weights"):
                    w = self.variable("weight", (width, width), "xavier")
                with self.names("biases"):
                    b = self.variable("bias", (width,), "zeros")
                z = self.pair("add", self.pair("mul", self.emit(width, "linear", x, w, -1, const=0.0), y, "mid"), b, "channel")
            else:
                z = self.emit(self.w[x], "act", self.pair("add", x, y, "mid"), code=_lib.ACT["TANH"])
            if keep_projected and project is None:
                raise UnboundLocalError("local variable 'projVals' referenced before assignment")
            if keep_x:
                z = self.join([z, projected if keep_projected else raw_x])
                width += (project["width"] if keep_projected else raw_w)
        out["value"], out["width"] = z, width
        return out


def compile_cell(config, netLength):
    """option set -> CellPlan.  Raises what the reference raises for option values it cannot build."""
    plan = CellPlan()
    e = _Emitter(config, plan)
    o = e.opt
    dc, dm, da = int(o("ctrlDim")), int(o("memDim")), int(o("attDim"))

    # ---------------- zero_state (mac_cell.py:539-592): runs in MACnetwork's scope, outside the cell's
    seg = plan.init = e.begin(Segment("init"))
    with e.names("MACnetwork"):
        def initial(leaf, width, how):
            if how == "PRM":
                return e.emit(width, "rows", e.variable(leaf, (width,), "normal"))
            return e.emit(width, "zeros", width=width) if how == "ZERO" else e.feed("vecQuestions")
        seg.results["control"] = initial("initCtrl", dc, o("initCtrl"))
        seg.results["memory"] = initial("initMem", dm, o("initMem"))
        words = e.feed("words")
        seg.results["in_words"] = seg.results["out_words"] = words
        if o("controlInWordsProj") or o("controlOutWordsProj"):
            projected = e.dense(words, dc, dc, tag="wordsProj")
            if o("controlInWordsProj"):
                seg.results["in_words"] = projected
            if o("controlOutWordsProj"):
                seg.results["out_words"] = projected
        if o("memoryVariationalDropout"):
            # ops.generateVarDpMask (ops.py:1054-1059): one mask per run, already divided by keep
            seg.results["mem_mask"] = e.dropout(e.emit(dm, "fill", width=dm, value=1.0), ("mem_var", "memory", 0))

    # ---------------- the steps (mac_cell.py:420-480)
    for i in range(netLength):
        seg = e.begin(Segment("step%d" % i))
        plan.steps.append(seg)
        vq = e.feed("vecQuestions")
        with e.names("MACnetwork"), e.names("MACCell"):
            # question -> this step's control input (mac_cell.py:442-448)
            cin = e.activation(o("controlInputAct"), e.dense(vq, dc, dc, tag="qInput"))
            cin = e.dense(cin, dc, dc, tag=("qInput%d" % i) if o("controlInputUnshared") else "qInputU")
            control, query = _control_unit(e, i, cin)
            if o("controlWholeQ"):
                control = vq
            info = _read_unit(e, i, e.feed("knowledgeBase"), e.feed("memory"), control)
            info = e.dropout(info, ("write_info", "write", i))
            seg.results["info"] = info
            seg.results["control"], seg.results["memory"] = control, _write_unit(e, i, e.feed("memory"), info, control, query)
    return plan


def compile_unit(config, unit, step=0):
    """ONE unit of step `step` as a segment of its own (per-unit contract), named like the reference's methods:
    "control" (controlInput, in_words, out_words, lengths, control, cont_control), "read" (knowledgeBase, memory, control),
    "write" (memory, info, control, cont_control).  Variables are named as inside the full cell."""
    plan = CellPlan()
    e = _Emitter(config, plan)
    seg = plan.init = e.begin(Segment(unit))
    with e.names("MACnetwork"), e.names("MACCell"):
        if unit == "control":
            ctl, query = _control_unit(e, step, e.feed("control_input"))
            seg.results["control"] = ctl
            seg.results.move_to_end("control")
        elif unit == "read":
            seg.results["info"] = _read_unit(e, step, e.feed("knowledgeBase"), e.feed("memory"), e.feed("control"))
        elif unit == "write":
            seg.results["memory"] = _write_unit(e, step, e.feed("memory"), e.feed("info"), e.feed("control"), e.feed("cont_control"))
        else:
            raise KeyError(unit)
    return plan


def _control_unit(e, i, cin):
    """mac_cell.py:133-187 -> (control, continuous control)"""
    o, seg = e.opt, e.seg
    dc = int(o("ctrlDim"))
    with e.names("control" + (str(i) if o("unsharedCells") else "")):
        query, width = cin, dc
        if o("controlFeedPrev"):
            parts = [e.feed("control") if o("controlFeedPrevAtt") else e.feed("cont_control")]
            if o("controlFeedInputs"):
                parts.append(cin)
            query = e.dense(e.join(parts), dc * len(parts), dc, tag="contControl", act=o("controlContAct"))
        scored = e.pair("mul", e.feed("in_words"), query, "mid")
        if o("controlConcatWords"):
            scored, width = e.join([scored, e.feed("in_words")]), width + dc
        if o("controlProj"):
            scored, width = e.dense(scored, width, dc, act=o("controlProjAct")), dc
        att_q, control = e.attend(scored, width, e.feed("out_words"), lengths=seg.feed("lengths"), wrap=False)
        if o("controlContinuous"):
            control = query
    seg.results["att_question"], seg.results["cont_control"] = att_q, query
    return control, query


def _read_unit(e, i, kb, memory, control):
    """mac_cell.py:209-277 -> information"""
    o, seg = e.opt, e.seg
    dc, dm, da = int(o("ctrlDim")), int(o("memDim")), int(o("attDim"))
    with e.names("read" + (str(i) if o("unsharedCells") else "")):
        if o("memoryVariationalDropout"):
            remembered = e.pair("mul", memory, e.feed("mem_mask"))
        else:
            remembered = e.dropout(memory, ("mem", "memory", i))
        project = {"width": da, "shared": o("readProjShared")} if o("readProjInputs") else None
        width = da if project else dm
        first = e.interact(kb, remembered, dm, "memInter", mode=o("readMemAttType"), project=project,
                           keep_x=o("readMemConcatKB"), keep_projected=o("readMemConcatProj"),
                           sites=(("read_kb", "read", i), ("read_mem", "read", i)))
        found = first["value"]
        if o("readMemProj"):
            found = e.dense(found, first["width"], width, tag="memKbProj", act=o("readMemAct"))
        else:
            width = first["width"]
        if o("readCtrl"):
            if dc != width:
                raise NameError("name 'ctrlDim' is not defined")                       # mac_cell.py:246
            found = e.interact(found, control, width, "ctrlInter", mode=o("readCtrlAttType"), keep_x=o("readCtrlConcatInter"))["value"]
            if o("readCtrlConcatKB"):
                extra, extra_w = (first.get("projected_x"), da) if o("readCtrlConcatProj") else (kb, dm)
                if extra is None:
                    raise ValueError("None values not supported.")                    # tf.concat([..., None])
                found, width = e.join([found, extra]), width + extra_w
            found = e.activation(o("readCtrlAct"), found)
        if e.w[found] != width:            # the logits layer is built for `width` inputs: TF's shape check
            raise ValueError("Dimensions must be equal, but are %d and %d" % (e.w[found], width))
        source = kb
        if o("readSmryKBProj"):
            source = first.get("projected_x")
            if source is None:
                raise ValueError("None values not supported.")                        # attention * None
        att_kb, info = e.attend(found, width, source, drop=("read_att", "read", i))
    seg.results["att_kb"] = att_kb
    return info


def _write_unit(e, i, memory, info, control, query):
    """mac_cell.py:305-375 -> new memory (`query`: this step's continuous control)"""
    o, seg = e.opt, e.seg
    dc, dm = int(o("ctrlDim")), int(o("memDim"))
    with e.names("write" + (str(i) if o("unsharedCells") else "")):
        if o("writeInfoProj"):
            info = e.dense(info, dm, dm, tag="info")
        info = e.activation(o("writeInfoAct"), info)
        recalled = None
        if o("writeSelfAtt"):
            probe = e.dense(query if o("writeSelfAttMod") == "CONT" else control, dc, dc, tag="ctrlProj")
            att_s, recalled = e.attend(e.pair("mul", e.feed("controls"), probe, "mid"), dc, e.feed("memories"), tag="selfAttention")
            seg.results["att_self"] = att_s
        how = o("writeInputs")
        if how == "INFO":
            parts = [info]
        elif how == "SUM":
            parts = [e.pair("add", memory, info)]
        elif how == "BOTH":
            parts = [memory, info] + ([e.pair("mul", memory, info)] if o("writeConcatMul") else [])
        else:
            parts = [memory]
        if recalled is not None:
            parts.append(recalled)
        if o("writeMergeCtrl"):
            parts.append(control)
        new, width = e.join(parts), dm * len(parts)
        if o("writeMemProj") or width != dm:
            new = e.dense(new, width, dm, tag="newMemory")
        new = e.activation(o("writeMemAct"), new)
        if o("writeGate"):
            if o("writeGateShared"):
                raise ValueError("Dimensions must be equal")                          # [B,d] * [B] (mac_cell.py:367)
            gate = e.emit(dm, "act", e.dense(control, dc, dm, tag="gate", const_bias=o("writeGateBias")), code=_lib.ACT["SIGMOID"])
            seg.results["att_gate"] = gate
            new = e.emit(dm, "blend", new, memory, gate)                              # new * z + memory * (1 - z)
        if o("memoryBN"):
            new = _batch_norm(e, new, dm, o("bnDecay"), o("bnCenter"), o("bnScale"))
    return new


def _batch_norm(e, x, c, decay, center, scale, eps=0.001):
    """tf.contrib.layers.batch_norm(updates_collections=None) on [B, c] (mac_cell.py:370-373): batch statistics (biased
    variance) + moving-average update in training, moving averages in evaluation"""
    seg = e.seg
    with e.names("BatchNorm", default=True):
        beta = e.variable("beta", (c,), "zeros") if center else None
        gamma = e.variable("gamma", (c,), 1.0) if scale else None
        mov_mean = e.variable("moving_mean", (c,), "zeros")
        mov_var = e.variable("moving_variance", (c,), 1.0)
    return e.emit(c, "bn", x, mov_mean, mov_var, *(v for v in (gamma, beta) if v is not None), decay=float(decay),
                  has_gamma=gamma is not None, has_beta=beta is not None, eps=eps)


# ---------------------------------------------------------------------------------------------------------------------
# execution: forward = the list; backward = the list reversed, one rule per primitive
# ---------------------------------------------------------------------------------------------------------------------
_SPREAD = {"same": G.B_SAME, "mid": G.B_MID, "channel": G.B_CHANNEL, "row": G.B_ROW}
_OPC = {"add": G.OP_ADD, "mul": G.OP_MUL}


def _ones_like(t):
    return torch.ones_like(t)


class _Exec:
    """the per-run context of a segment execution: dropout stream, mode, batch"""

    def __init__(self, seed, b0, keeps, train, batch, device, mask_word=None):
        self.seed, self.b0, self.keeps, self.train, self.batch, self.device = seed, b0, keeps, train, batch, device
        self.mask_word = mask_word                 # macx_dropout.mask_word of the run (device tensor) or None

    # ---- forward rules: (node, inputs) -> output (+ what backward wants, stashed per node)
    def fwd(self, nd, x, stash):
        op, a = nd.op, nd.attr
        if op == "linear":
            inp, W, b = x[0].contiguous(), x[1], (x[2] if len(x) > 2 else None)
            K, n = W.shape
            big = (inp.shape[0], inp.shape[1]) if (inp.dim() == 3 and inp.shape[1] <= 1024) else None
            if b is not None and a["const"] != 0.0:
                b = G.k_binary(G.OP_ADD, G.B_SAME, b.contiguous(), torch.full_like(b, a["const"]), 1, b.shape[-1])
            stash[nd.out] = big
            return G.k_matmul(inp.reshape(-1, K), W, b, big).reshape(inp.shape[:-1] + (n,))
        if op == "rowdot":
            inp, w = x[0].contiguous(), x[1]
            c = inp.shape[-1]
            y = G.k_reduce(G.R_LAST, G.k_binary(G.OP_MUL, G.B_CHANNEL, inp, w, 1, c), inp.numel() // c, 1, c)
            b = x[2] if len(x) > 2 else None
            if b is not None:
                bb = b.reshape(1) + a["const"] if a["const"] else b.reshape(1)
                y = G.k_binary(G.OP_ADD, G.B_CHANNEL, y.reshape(-1, 1), bb.contiguous(), 1, 1).reshape(-1)
            return y.reshape(inp.shape[:-1])
        if op == "act":
            return G.k_act(a["code"], x[0].contiguous(), x[1] if len(x) > 1 else None)
        if op == "bin":
            p, q = x[0].contiguous(), x[1].contiguous()
            inner = p.shape[-1]
            mid = p.shape[-2] if (a["spread"] == "mid" and p.dim() >= 2) else 1
            return G.k_binary(_OPC[a["op"]], _SPREAD[a["spread"]], p, q, mid, inner, a["scale"])
        if op == "shift":
            p = x[0].contiguous()
            return G.k_binary(G.OP_ADD, G.B_CHANNEL, p, torch.full((p.shape[-1],), a["by"], device=p.device), 1, p.shape[-1])
        if op == "softmax":
            return G.k_softmax(x[0].contiguous(), x[1] if len(x) > 1 else None)
        if op == "wsum":
            att, vals = x[0].contiguous(), x[1].contiguous()
            B, N, d = vals.shape
            return G.k_reduce(G.R_MID, G.k_binary(G.OP_MUL, G.B_ROW, vals, att.reshape(-1), 1, d), B, N, d)
        if op == "drop":
            if self.keeps[a["keep"]] == 1.0:
                return x[0]
            p = x[0].contiguous()
            return G.k_dropout(p, self.seed, a["site"], a["step"], self.keeps[a["keep"]], self.b0 * (p.numel() // p.shape[0]), self.mask_word)
        if op == "cat":
            return torch.cat(list(x), dim=-1)
        if op == "rows":                      # a [c] variable as the B rows of a [B, c] state
            return G.k_binary(G.OP_ADD, G.B_CHANNEL, torch.zeros((self.batch, x[0].shape[0]), device=x[0].device), x[0], 1, x[0].shape[0])
        if op == "zeros":
            return torch.zeros((self.batch, a["width"]), dtype=torch.float32, device=self.device)
        if op == "fill":
            return torch.full((self.batch, a["width"]), a["value"], dtype=torch.float32, device=self.device)
        if op == "blend":
            new, old, z = [t.contiguous() for t in x]
            inner = z.shape[-1]
            ones = _ones_like(z)
            rest = G.k_binary(G.OP_ADD, G.B_SAME, G.k_binary(G.OP_MUL, G.B_SAME, z, ones, 1, inner, -1.0), ones, 1, inner)
            stash[nd.out] = rest
            return G.k_binary(G.OP_ADD, G.B_SAME, G.k_binary(G.OP_MUL, G.B_SAME, new, z, 1, inner), G.k_binary(G.OP_MUL, G.B_SAME, old, rest, 1, inner), 1, inner)
        if op == "bn":
            return self._bn_fwd(nd, x, stash)
        raise KeyError(op)

    # ---- backward rules: (node, upstream gradient, inputs, output, which inputs want one) -> gradients per input
    def bwd(self, nd, g, x, y, want, stash):
        op, a = nd.op, nd.attr
        g = g.contiguous()
        if op == "linear":
            inp, W = x[0].contiguous(), x[1]
            K, n = W.shape
            g2, big = g.reshape(-1, n), stash.get(nd.out)
            dx = G.k_matmul(g2, W.t().contiguous(), None, big).reshape(inp.shape) if want[0] else None
            dW = G.k_wgrad(inp.reshape(-1, K), g2) if want[1] else None
            db = G.k_reduce(G.R_ROWS, g2, g2.shape[0], 1, n) if (len(x) > 2 and want[2]) else None
            return [dx, dW, db][:len(x)]
        if op == "rowdot":
            inp, w = x[0].contiguous(), x[1]
            c = inp.shape[-1]
            rows = inp.numel() // c
            spread = G.k_binary(G.OP_MUL, G.B_ROW, torch.ones((rows, c), device=g.device), g.reshape(-1), 1, c)     # g per row over the columns
            dx = G.k_binary(G.OP_MUL, G.B_CHANNEL, spread, w, 1, c).reshape(inp.shape) if want[0] else None
            dw = G.k_reduce(G.R_ROWS, G.k_binary(G.OP_MUL, G.B_SAME, spread, inp.reshape(rows, c), 1, c), rows, 1, c) if want[1] else None
            out = [dx, dw]
            if len(x) > 2:
                out.append(G.k_reduce(G.R_ROWS, g.reshape(-1, 1), rows, 1, 1).reshape(x[2].shape) if want[2] else None)
            return out
        if op == "act":
            inp = x[0].contiguous()
            slope = x[1] if len(x) > 1 else None
            dx, de = G.k_act_bwd(a["code"], inp, slope, g)
            out = [dx]
            if slope is not None:
                c = inp.shape[-1]
                out.append(G.k_reduce(G.R_ROWS, de, inp.numel() // c, 1, c) if (de is not None and want[1]) else None)
            return out
        if op == "bin":
            p, q = x[0].contiguous(), x[1].contiguous()
            inner, n = p.shape[-1], g.numel()
            mid = p.shape[-2] if (a["spread"] == "mid" and p.dim() >= 2) else 1
            mode, scale = _SPREAD[a["spread"]], a["scale"]
            if a["op"] == "add":
                gs = g if scale == 1.0 else G.k_binary(G.OP_ADD, G.B_SAME, g, torch.zeros_like(g), 1, inner, scale)
                dp, full = (gs if want[0] else None), gs
            else:
                dp = G.k_binary(G.OP_MUL, mode, g, q, mid, inner, scale) if want[0] else None
                full = G.k_binary(G.OP_MUL, G.B_SAME, g, p, 1, inner, scale) if want[1] else None
            dq = None
            if want[1]:
                if mode == G.B_SAME:
                    dq = full
                elif mode == G.B_MID:
                    dq = G.k_reduce(G.R_MID, full, n // (mid * inner), mid, inner)
                elif mode == G.B_CHANNEL:
                    dq = G.k_reduce(G.R_ROWS, full, n // inner, 1, inner)
                else:
                    dq = G.k_reduce(G.R_LAST, full, n // inner, 1, inner)
                dq = dq.reshape(q.shape)
            return [dp, dq]
        if op == "shift":
            return [g]
        if op == "softmax":
            return [G.k_softmax_bwd(y, g)] + [None] * (len(x) - 1)
        if op == "wsum":
            att, vals = x[0].contiguous(), x[1].contiguous()
            B, N, d = vals.shape
            gb = G.k_binary(G.OP_MUL, G.B_MID, torch.ones((B, N, d), device=g.device), g, N, d)        # g over the N cells
            d_att = G.k_reduce(G.R_LAST, G.k_binary(G.OP_MUL, G.B_SAME, gb, vals, 1, d), B * N, 1, d).reshape(att.shape) if want[0] else None
            d_val = G.k_binary(G.OP_MUL, G.B_ROW, gb, att.reshape(-1), 1, d) if want[1] else None
            return [d_att, d_val]
        if op == "drop":
            if self.keeps[a["keep"]] == 1.0:
                return [g]
            return [G.k_dropout(g, self.seed, a["site"], a["step"], self.keeps[a["keep"]], self.b0 * (g.numel() // g.shape[0]), self.mask_word)]
        if op == "cat":
            out, at = [], 0
            for t, w in zip(x, want):
                out.append(g.narrow(-1, at, t.shape[-1]).contiguous() if w else None)
                at += t.shape[-1]
            return out
        if op == "rows":
            return [G.k_reduce(G.R_ROWS, g, g.shape[0], 1, g.shape[1])]
        if op == "blend":
            new, old, z = [t.contiguous() for t in x]
            inner, rest = z.shape[-1], stash[nd.out]
            d_new = G.k_binary(G.OP_MUL, G.B_SAME, g, z, 1, inner) if want[0] else None
            d_old = G.k_binary(G.OP_MUL, G.B_SAME, g, rest, 1, inner) if want[1] else None
            d_z = None
            if want[2]:
                diff = G.k_binary(G.OP_ADD, G.B_SAME, new, G.k_binary(G.OP_MUL, G.B_SAME, old, _ones_like(old), 1, inner, -1.0), 1, inner)
                d_z = G.k_binary(G.OP_MUL, G.B_SAME, g, diff, 1, inner)
            return [d_new, d_old, d_z]
        if op == "bn":
            return self._bn_bwd(nd, g, x, want, stash)
        raise KeyError(op)

    # ---- batch norm over the rows of [B, c]
    def _bn_fwd(self, nd, x, stash):
        a = nd.attr
        inp, mov_mean, mov_var = x[0].contiguous(), x[1], x[2]
        rest = list(x[3:])
        gamma = rest.pop(0) if a["has_gamma"] else None
        beta = rest.pop(0) if a["has_beta"] else None
        B, c = inp.shape
        ones_c = torch.ones(c, dtype=torch.float32, device=inp.device)
        neg = lambda v: G.k_binary(G.OP_MUL, G.B_SAME, v.contiguous(), ones_c, 1, c, -1.0)
        if self.train:
            mean = G.k_binary(G.OP_MUL, G.B_SAME, G.k_reduce(G.R_ROWS, inp, B, 1, c), ones_c, 1, c, 1.0 / B)
            cen = G.k_binary(G.OP_ADD, G.B_CHANNEL, inp, neg(mean), 1, c)
            var = G.k_binary(G.OP_MUL, G.B_SAME, G.k_reduce(G.R_ROWS, G.k_binary(G.OP_MUL, G.B_SAME, cen, cen, 1, c), B, 1, c), ones_c, 1, c, 1.0 / B)
            with torch.no_grad():          # assign_moving_average: m -= (1 - decay) (m - stat)
                for mov, stat in ((mov_mean, mean), (mov_var, var)):
                    kept = G.k_binary(G.OP_MUL, G.B_SAME, mov.detach().contiguous(), ones_c, 1, c, a["decay"])
                    mov.copy_(G.k_binary(G.OP_ADD, G.B_SAME, kept, G.k_binary(G.OP_MUL, G.B_SAME, stat.contiguous(), ones_c, 1, c, 1.0 - a["decay"]), 1, c))
        else:
            cen = G.k_binary(G.OP_ADD, G.B_CHANNEL, inp, neg(mov_mean.detach()), 1, c)
            var = mov_var.detach().contiguous()
        eps = torch.full((1,), a["eps"], dtype=torch.float32, device=inp.device)
        inv = G.k_act(G.ACT_RSQRT_EPS, var, eps)
        xhat = G.k_binary(G.OP_MUL, G.B_CHANNEL, cen, inv, 1, c)
        stash[nd.out] = (cen, var, inv, xhat, eps)
        out = xhat
        if gamma is not None:
            out = G.k_binary(G.OP_MUL, G.B_CHANNEL, out, gamma, 1, c)
        if beta is not None:
            out = G.k_binary(G.OP_ADD, G.B_CHANNEL, out, beta, 1, c)
        return out

    def _bn_bwd(self, nd, g, x, want, stash):
        a = nd.attr
        cen, var, inv, xhat, eps = stash[nd.out]
        B, c = g.shape
        ones_c = torch.ones(c, dtype=torch.float32, device=g.device)
        rest = list(x[3:])
        gamma = rest.pop(0) if a["has_gamma"] else None
        d_gamma = d_beta = None
        g_hat = g
        if gamma is not None:
            d_gamma = G.k_reduce(G.R_ROWS, G.k_binary(G.OP_MUL, G.B_SAME, g, xhat, 1, c), B, 1, c)
            g_hat = G.k_binary(G.OP_MUL, G.B_CHANNEL, g, gamma, 1, c)
        if a["has_beta"]:
            d_beta = G.k_reduce(G.R_ROWS, g, B, 1, c)
        # xhat = cen * inv(var)
        d_cen = G.k_binary(G.OP_MUL, G.B_CHANNEL, g_hat, inv, 1, c)
        dx = d_cen
        if self.train:
            d_inv = G.k_reduce(G.R_ROWS, G.k_binary(G.OP_MUL, G.B_SAME, g_hat, cen, 1, c), B, 1, c)
            d_var, _ = G.k_act_bwd(G.ACT_RSQRT_EPS, var, eps, d_inv)
            # var = mean(cen^2): d cen += 2 cen d_var / B ; cen = x - mean(x): dx = d_cen - mean(d_cen)
            d_cen = G.k_binary(G.OP_ADD, G.B_SAME, d_cen, G.k_binary(G.OP_MUL, G.B_CHANNEL, cen, d_var, 1, c, 2.0 / B), 1, c)
            col = G.k_binary(G.OP_MUL, G.B_SAME, G.k_reduce(G.R_ROWS, d_cen, B, 1, c), ones_c, 1, c, -1.0 / B)
            dx = G.k_binary(G.OP_ADD, G.B_CHANNEL, d_cen, col, 1, c)
        return [dx, None, None] + [v for v, have in ((d_gamma, a["has_gamma"]), (d_beta, a["has_beta"])) if have]


class _SegmentFn(torch.autograd.Function):
    """one plan segment = one autograd node: forward runs the list, backward runs it in reverse"""

    @staticmethod
    def forward(ctx, seg, ex, order, *tensors):
        vals, stash = [None] * seg.n, {}
        for s, t in zip(order, tensors):
            vals[s] = t
        for nd in seg.nodes:
            vals[nd.out] = ex.fwd(nd, [vals[i] for i in nd.ins if i >= 0], stash)
        ctx.seg, ctx.ex, ctx.order, ctx.vals, ctx.stash = seg, ex, order, vals, stash
        ctx.needs = [t is not None and isinstance(t, torch.Tensor) and t.requires_grad for t in tensors]
        # a result that IS an input (a state passed through), or the same value under two names, must not alias for autograd
        outs, seen = [], [id(t) for t in tensors if isinstance(t, torch.Tensor)]
        for sl in seg.results.values():
            v = vals[sl]
            outs.append(v.view_as(v) if id(v) in seen else v)
            seen.append(id(v))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        seg, ex, vals, stash = ctx.seg, ctx.ex, ctx.vals, ctx.stash
        req = [False] * seg.n
        for s, need in zip(ctx.order, ctx.needs):
            req[s] = need
        for nd in seg.nodes:
            req[nd.out] = nd.op not in NOGRAD_OPS and any(req[i] for i in nd.ins if i >= 0)
        grads = [None] * seg.n

        def add(slot, g):
            if g is None:
                return
            if grads[slot] is None:
                grads[slot] = g
            else:
                inner = g.shape[-1] if g.dim() else 1
                grads[slot] = G.k_binary(G.OP_ADD, G.B_SAME, grads[slot].contiguous(), g.contiguous().reshape(grads[slot].shape), 1, inner)

        for s, g in zip(seg.results.values(), gouts):
            if g is not None and req[s]:
                add(s, g)
        for nd in reversed(seg.nodes):
            g = grads[nd.out]
            if g is None or not req[nd.out]:
                continue
            ins = [i for i in nd.ins if i >= 0]
            gi = ex.bwd(nd, g, [vals[i] for i in ins], vals[nd.out], [req[i] for i in ins], stash)
            for i, gg in zip(ins, gi):
                if req[i]:
                    add(i, gg)
            grads[nd.out] = None
        out = [None, None, None]
        for s, need in zip(ctx.order, ctx.needs):
            out.append(grads[s] if need else None)
        return tuple(out)


def run_segment(seg, ex, feeds, params):
    """feeds: name -> tensor; params: object with ensure(name, shape, init).  Returns name -> tensor for seg.results."""
    order, tensors = [], []
    for name, s in seg.feeds.items():
        order.append(s)
        tensors.append(feeds[name])
    for name, s in seg.vars.items():
        order.append(s)
        tensors.append(params.table[params.names[name]])
    outs = _SegmentFn.apply(seg, ex, order, *tensors)
    return dict(zip(seg.results.keys(), outs))
