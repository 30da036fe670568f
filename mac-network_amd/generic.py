"""The GENERIC option path of the MAC cell: every legal option combination of mac_cell.py that the fused cell kernels
(cell.py -> macx_cell_*) answer with UnsupportedOptions -- the reference's DEFAULT configuration, writeInputs != BOTH,
readMemAttType BL / ADD, relu = PRM, writeConcatMul, controlProj, controlConcatWords, unsharedCells, mulBias, ... -- runs
as one HIP kernel per primitive of a PLAN that plan.compile_cell derives from the option set once (plan.py: variable table in
the reference's creation order + a flat operation list per zero_state / step, executed and differentiated by plan.run_segment).
This module holds what the plan stands on: the kernel calls (k_*), the variable store under the reference's names, the cell
class that feeds the plan's segments, and the few eager layer helpers the generic question encoder / output unit use.

Every arithmetic step is a kernel of libmacx.so behind the C ABI (include/macx.h: macx_linear / macx_h2_gemm / macx_wgrad and
the macx_op_* primitives); PyTorch owns the device memory, the stream, the concat / slice copies and the autograd tape that
orders the backward kernels -- plumbing.  There is no CPU path: tensors must live on the HIP device.

Variables are created on first use under the reference's TF names (the same scope stacking as ops.py / mac_cell.py), so a
reference checkpoint loads by name (GenericParams.load_reference_dict) whatever the option set.

Not covered (UnsupportedOptions, never a silent fallback): dimensions that are not multiples of 128.
"""
import collections
import ctypes as C
import math
from contextlib import contextmanager
from types import SimpleNamespace

import torch

from . import _lib
from .options import UnsupportedOptions, fresh_seed, get, reject_like_reference, DEFAULTS

MACCellTuple = collections.namedtuple("MACCellTuple", ("control", "memory"))

OP_ADD, OP_MUL = 0, 1
B_SAME, B_MID, B_CHANNEL, B_ROW = 0, 1, 2, 3
R_MID, R_LAST, R_ROWS = 0, 1, 2
ACT_PRELU = 16
ACT_RSQRT_EPS = 17
SITE_MEM_VAR, SITE_MEM, SITE_READ_KB, SITE_READ_MEM, SITE_READ_ATT, SITE_WRITE_INFO = 1, 2, 3, 4, 5, 6


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _st(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _require_device(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the HIP device: the MAC cell has no CPU path" % name)


def _dev(t, name="tensor"):
    _require_device(t, name)
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32" % name)
    return t.contiguous()


def _L():
    return _lib.lib()


# -------------------------------------------------------------------------------------------------------------------
# the kernel calls (everything below this block is host logic; tests/test_generic_host.py swaps exactly these functions for
# torch restatements to check the chaining, the variable names and the backward formulas on a machine without a GPU)
# -------------------------------------------------------------------------------------------------------------------
def k_binary(op, bmode, a, b, mid, inner, scale=1.0):
    out = torch.empty_like(a)
    _lib.check(_L().macx_op_binary(op, bmode, _p(a), _p(b), a.numel(), mid, inner, scale, _p(out), _st(a)), "macx_op_binary")
    return out


def k_reduce(mode, x, outer, mid, inner):
    shape = {R_MID: (outer, inner), R_LAST: (outer,), R_ROWS: (inner,)}[mode]
    out = torch.empty(shape, dtype=torch.float32, device=x.device)
    ws = torch.empty(64 * inner, dtype=torch.float32, device=x.device) if mode == R_ROWS else None
    _lib.check(_L().macx_op_reduce(mode, _p(x), outer, mid, inner, _p(out), _p(ws), _st(x)), "macx_op_reduce")
    return out


def k_act(act, x, alpha):
    out = torch.empty_like(x)
    _lib.check(_L().macx_op_act(act, _p(x), _p(alpha), x.numel(), x.shape[-1], _p(out), _st(x)), "macx_op_act")
    return out


def k_act_bwd(act, x, alpha, g):
    dx = torch.empty_like(x)
    de = torch.empty_like(x) if act == ACT_PRELU else None          # (ACT_RSQRT_EPS: alpha holds eps, no gradient)
    _lib.check(_L().macx_op_act_bwd(act, _p(x), _p(alpha), _p(g), x.numel(), x.shape[-1], _p(dx), _p(de), _st(x)), "macx_op_act_bwd")
    return dx, de


def k_softmax(x, lengths):
    n = x.shape[-1]
    rows = x.numel() // n
    out = torch.empty_like(x)
    rpl = rows // lengths.numel() if lengths is not None else 1
    _lib.check(_L().macx_op_softmax(_p(x), _p(lengths), rpl, rows, n, _p(out), _st(x)), "macx_op_softmax")
    return out


def k_softmax_bwd(a, g):
    n = a.shape[-1]
    dx = torch.empty_like(a)
    _lib.check(_L().macx_op_softmax_bwd(_p(a), _p(g), a.numel() // n, n, _p(dx), _st(a)), "macx_op_softmax_bwd")
    return dx


def k_dropout(x, seed, site, step, keep, first, mask_word=None):
    """mask_word: macx_dropout.mask_word (1-element int32 device tensor XORed into the site key when the kernel runs) or None"""
    out = torch.empty_like(x)
    _lib.check(_L().macx_op_dropout_w(_p(x), x.numel(), seed, site, step, keep, first, _p(mask_word) if mask_word is not None else None,
                                      _p(out), _st(x)), "macx_op_dropout")
    return out


def _pad128(v):
    return (v + 127) // 128 * 128


def k_matmul(x, W, b, big):
    """x [rows, K] @ W [K, n] (+ b): the knowledge-base GEMM for [B, N, .] operands (big = (B, N)), macx_linear otherwise.
    Layer widths the kernels' 128-column granule does not divide (config.py:294-296 and every *Dim option take any integer) run
    zero-padded: zero rows of W meet zero columns of x, the padded output columns are dropped."""
    L = _L()
    rows, K = x.shape
    n = W.shape[1]
    Kp, np_ = _pad128(K), _pad128(n)
    if Kp != K or np_ != n:
        pad = torch.nn.functional.pad
        xp = pad(x, (0, Kp - K)).contiguous() if Kp != K else x
        Wp = pad(W, (0, np_ - n, 0, Kp - K)).contiguous()
        bp = pad(b, (0, np_ - n)).contiguous() if b is not None else None
        return k_matmul(xp, Wp, bp, big)[:, :n].contiguous()
    out = torch.empty((rows, n), dtype=torch.float32, device=x.device)
    bias = b if b is not None else torch.zeros(n, dtype=torch.float32, device=x.device)
    if big is not None:
        B, N = big
        nws = L.macx_h2_floats(rows, K) + L.macx_h2_floats(rows, n) + K * n + 64
        ws = torch.empty(nws, dtype=torch.float32, device=x.device)
        _lib.check(L.macx_h2_gemm(_p(x), B, N, K, _p(W), n, _p(bias), 0, _p(out), _p(ws), nws, _st(x)), "macx_h2_gemm")
    else:
        wp = torch.empty(K * n, dtype=torch.float32, device=x.device)
        _lib.check(L.macx_pack_weight(_p(W), K, n, _lib.PACK_F32MFMA, _p(wp), _st(x)), "macx_pack_weight")
        _lib.check(L.macx_linear(_p(x), K, None, 0, rows, _p(wp), _p(bias), 0.0, n, 0, _p(out), _st(x)), "macx_linear")
    return out


def k_embed(ids, emb, E, ld, keep, seed, first_row):
    """rows of the zero-padded embedding table for `ids` [rows] int32, through the input dropout: [rows, ld] (columns >= E zero)."""
    out = torch.empty((ids.numel(), ld), dtype=torch.float32, device=emb.device)
    _lib.check(_L().macx_embed_lookup(_p(ids), _p(emb), ids.numel(), E, ld, keep, seed & 0xFFFFFFFF, first_row, _p(out), _st(emb)),
               "macx_embed_lookup")
    return out


def k_embed_bwd(ids, dx, E, V, keep, seed, first_row):
    d_emb = torch.empty((V, E), dtype=torch.float32, device=dx.device)
    _lib.check(_L().macx_embed_lookup_bwd(_p(ids), _p(dx), ids.numel(), E, dx.shape[1], V, keep, seed & 0xFFFFFFFF, first_row,
                                          _p(d_emb), _st(dx)), "macx_embed_lookup_bwd")
    return d_emb


def k_wgrad(x2, g2):
    """dW [K, n] = x2^T g2 (fixed-order split reduction)."""
    L = _L()
    rows, K = x2.shape
    n = g2.shape[1]
    Kp, np_ = _pad128(K), _pad128(n)
    if Kp != K or np_ != n:                       # widths off the 128-column granule: zero-padded operands, the padding dropped
        pad = torch.nn.functional.pad
        return k_wgrad(pad(x2, (0, Kp - K)).contiguous(), pad(g2, (0, np_ - n)).contiguous())[:K, :n].contiguous()
    dW = torch.empty((K, n), dtype=torch.float32, device=g2.device)
    ws = torch.empty(L.macx_wgrad_splits(rows, K, n) * K * n, dtype=torch.float32, device=g2.device)
    _lib.check(L.macx_wgrad(_p(x2), K, _p(g2), n, rows, K, n, _p(dW), _p(ws), _st(g2)), "macx_wgrad")
    return dW


# -------------------------------------------------------------------------------------------------------------------
# autograd nodes: forward and backward are the kernels above
# -------------------------------------------------------------------------------------------------------------------
class _Binary(torch.autograd.Function):
    """out = scale * (a op b); b broadcast by `bmode` (macx_op_binary).  a: [..., mid, inner] contiguous."""

    @staticmethod
    def forward(ctx, a, b, op, bmode, scale):
        a, b = _dev(a), _dev(b)
        inner = a.shape[-1]
        mid = a.shape[-2] if (bmode == B_MID and a.dim() >= 2) else 1
        ctx.meta = (op, bmode, scale, mid, inner, tuple(b.shape))
        ctx.save_for_backward(a, b)
        return k_binary(op, bmode, a, b, mid, inner, scale)

    @staticmethod
    def backward(ctx, g):
        op, bmode, scale, mid, inner, bshape = ctx.meta
        a, b = ctx.saved_tensors
        g = g.contiguous()
        n = g.numel()
        need_a, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        da = db = None
        if op == OP_ADD:
            gs = g if scale == 1.0 else k_binary(OP_ADD, B_SAME, g, torch.zeros_like(g), 1, inner, scale)
            da, gb_full = (gs if need_a else None), gs
        else:
            if need_a:
                da = k_binary(OP_MUL, bmode, g, b, mid, inner, scale)
            gb_full = k_binary(OP_MUL, B_SAME, g, a, 1, inner, scale) if need_b else None
        if need_b:
            if bmode == B_SAME:
                db = gb_full
            elif bmode == B_MID:
                db = k_reduce(R_MID, gb_full, n // (mid * inner), mid, inner)
            elif bmode == B_CHANNEL:
                db = k_reduce(R_ROWS, gb_full, n // inner, 1, inner)
            else:
                db = k_reduce(R_LAST, gb_full, n // inner, 1, inner)
            db = db.reshape(bshape)
        return da, db, None, None, None


class _Reduce(torch.autograd.Function):
    """Sum over the middle axis of [B,N,d] (R_MID) or the last axis of [...,d] (R_LAST) (macx_op_reduce)."""

    @staticmethod
    def forward(ctx, x, mode):
        x = _dev(x)
        ctx.mode, ctx.shape = mode, tuple(x.shape)
        if mode == R_MID:
            B, N, d = x.shape
            return k_reduce(R_MID, x, B, N, d)
        d = x.shape[-1]
        return k_reduce(R_LAST, x, x.numel() // d, 1, d).reshape(x.shape[:-1])

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        ones = torch.ones(ctx.shape, dtype=torch.float32, device=g.device)
        if ctx.mode == R_MID:
            return k_binary(OP_MUL, B_MID, ones, g, ctx.shape[1], ctx.shape[2]), None
        return k_binary(OP_MUL, B_ROW, ones, g.reshape(-1), 1, ctx.shape[-1]), None


class _RowSum(torch.autograd.Function):
    """Sum over the rows of [rows, c] (macx_op_reduce ROWS): batch statistics."""

    @staticmethod
    def forward(ctx, x):
        x = _dev(x)
        ctx.shape = tuple(x.shape)
        return k_reduce(R_ROWS, x, x.shape[0], 1, x.shape[1])

    @staticmethod
    def backward(ctx, g):
        ones = torch.ones(ctx.shape, dtype=torch.float32, device=g.device)
        return k_binary(OP_MUL, B_CHANNEL, ones, g.contiguous(), 1, ctx.shape[1])


class _Act(torch.autograd.Function):
    """ops.activations / ops.relu (ops.py:161-187) on macx_op_act; PRELU carries a per-channel alpha."""

    @staticmethod
    def forward(ctx, x, act, alpha):
        x = _dev(x)
        al = _dev(alpha) if alpha is not None else None
        ctx.act = act
        ctx.save_for_backward(x, al)
        return k_act(act, x, al)

    @staticmethod
    def backward(ctx, g):
        x, al = ctx.saved_tensors
        inner = x.shape[-1]
        dx, de = k_act_bwd(ctx.act, x, al, g.contiguous())
        dal = k_reduce(R_ROWS, de, x.numel() // inner, 1, inner) if de is not None else None
        return dx, None, dal


class _Softmax(torch.autograd.Function):
    """softmax over the last axis, optionally behind ops.expMask (ops.py:243-247) (macx_op_softmax)."""

    @staticmethod
    def forward(ctx, x, lengths):
        out = k_softmax(_dev(x), lengths)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        return k_softmax_bwd(a, g.contiguous()), None


class _Dropout(torch.autograd.Function):
    """tf.nn.dropout on the stateless stream: x / keep * mask(seed, site, step, first + i) (macx_op_dropout)."""

    @staticmethod
    def forward(ctx, x, seed, site, step, keep, first):
        ctx.meta = (seed, site, step, keep, first)
        return k_dropout(_dev(x), seed, site, step, keep, first)

    @staticmethod
    def backward(ctx, g):
        return (k_dropout(g.contiguous(), *ctx.meta),) + (None,) * 5


class _Linear(torch.autograd.Function):
    """y = x W + b over the last axis (ops.multiply + bias, ops.py:50-58, 320-323): macx_h2_gemm / macx_linear forward and
    backward-data, macx_wgrad for dW, a fixed-order column sum for db."""

    @staticmethod
    def forward(ctx, x, W, b):
        x, W = _dev(x), _dev(W)
        K, n = W.shape
        x2 = x.reshape(-1, K)
        big = (x.shape[0], x.shape[1]) if (x.dim() == 3 and x.shape[1] <= 1024) else None
        ctx.big, ctx.xshape, ctx.has_b = big, tuple(x.shape), b is not None
        ctx.save_for_backward(x2, W)
        return k_matmul(x2, W, _dev(b) if b is not None else None, big).reshape(x.shape[:-1] + (n,))

    @staticmethod
    def backward(ctx, g):
        x2, W = ctx.saved_tensors
        n = W.shape[1]
        g2 = g.contiguous().reshape(-1, n)
        dx = k_matmul(g2, W.t().contiguous(), None, ctx.big).reshape(ctx.xshape) if ctx.needs_input_grad[0] else None
        dW = k_wgrad(x2, g2) if ctx.needs_input_grad[1] else None
        db = k_reduce(R_ROWS, g2, g2.shape[0], 1, n) if ctx.has_b and ctx.needs_input_grad[2] else None
        return dx, dW, db


class _Embed(torch.autograd.Function):
    """tf.nn.embedding_lookup(concat([zeros(1, E), emb]), ids) + tf.nn.dropout (model.py:207-219, ops.py:812/880):
    macx_embed_lookup forward, macx_embed_lookup_bwd (one workgroup per vocabulary row, fixed order) backward."""

    @staticmethod
    def forward(ctx, ids, emb, ld, keep, seed, first_row):
        emb = _dev(emb, "emb")
        _require_device(ids, "questions")
        ids = ids.reshape(-1).to(torch.int32).contiguous()
        ctx.save_for_backward(ids)
        ctx.meta = (emb.shape[0], emb.shape[1], keep, seed, first_row)
        return k_embed(ids, emb, emb.shape[1], ld, keep, seed, first_row)

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        V, E, keep, seed, first_row = ctx.meta
        return None, k_embed_bwd(ids, g.contiguous(), E, V, keep, seed, first_row), None, None, None, None


# -------------------------------------------------------------------------------------------------------------------
# variables under the reference's names
# -------------------------------------------------------------------------------------------------------------------
class GenericParams(torch.nn.Module):
    """name -> Parameter, created on first use with the reference's initialisers (xavier-uniform weights ops.py:20, zero
    biases ops.py:40, N(0,1) state variables mac_cell.py:498-499, 0.25 PReLU slopes ops.py:172)."""

    def __init__(self, device=None, generator=None):
        super().__init__()
        self.device = device
        self.gen = generator
        self.table = torch.nn.ParameterDict()
        self.names = {}               # reference name -> ParameterDict key
        self._stack = []
        self._counts = {}             # full scope name -> times opened (default-name scopes: prelu, prelu_1, ...)

    @contextmanager
    def scope(self, name, default=False):
        """tf.variable_scope(name); default=True: tf.variable_scope(None, default_name=name) -- `name`, then `name_1`, ... within
        one opening of the enclosing scope (ops.py:163; the counts of a scope's children reset when it closes, as in TF)."""
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

    def get(self, name, shape, init):
        key = "/".join(self._stack + [name])
        shape = tuple(shape)
        if key not in self.names:
            if init == "xavier":
                fan_in, fan_out = (shape[0], shape[0]) if len(shape) == 1 else (shape[-2], shape[-1])
                lim = math.sqrt(6.0 / (fan_in + fan_out))
                t = (torch.rand(shape, generator=self.gen, dtype=torch.float64) * 2 - 1) * lim
            elif init == "zeros":
                t = torch.zeros(shape, dtype=torch.float64)
            elif init == "normal":
                t = torch.randn(shape, generator=self.gen, dtype=torch.float64)
            else:
                t = torch.full(shape, float(init), dtype=torch.float64)
            self._add(key, t.to(torch.float32))
        v = self.table[self.names[key]]
        if tuple(v.shape) != shape:
            raise ValueError("variable %s has shape %s, expected %s" % (key, tuple(v.shape), shape))
        return v

    def ensure(self, full_name, shape, init):
        """the variable `full_name` (a complete scope path), created with `init` unless it exists (a loaded checkpoint)"""
        saved, self._stack = self._stack, []
        try:
            return self.get(full_name, shape, init)
        finally:
            self._stack = saved

    def _apply(self, fn, *a, **kw):
        """.to(device) / .cuda() on this module or on any module that holds it: variables created later follow."""
        out = super()._apply(fn, *a, **kw)
        held = self.tensors()
        if held:
            self.device = held[0].device
        else:
            try:
                probe = torch.empty(0, device=self.device) if self.device is not None else torch.empty(0)
                self.device = fn(probe).device
            except Exception:          # noqa: BLE001 -- keep the device
                pass
        return out

    def _add(self, key, t):
        mangled = "v%d" % len(self.names)
        self.names[key] = mangled
        self.table[mangled] = torch.nn.Parameter(t.to(self.device) if self.device is not None else t)

    def tensors(self):
        return [self.table[m] for m in self.names.values()]

    def to_reference_dict(self):
        return {k: self.table[m].detach().clone() for k, m in self.names.items()}

    def grads_by_name(self):
        return {k: self.table[m].grad for k, m in self.names.items()}

    @torch.no_grad()
    def load_reference_dict(self, ref):
        """Adopt {TF variable name: array} (names with or without a leading 'macModel/' and a trailing ':0')."""
        for k, v in ref.items():
            k = k[len("macModel/"):] if k.startswith("macModel/") else k
            k = k[:-2] if k.endswith(":0") else k
            t = torch.as_tensor(v).detach().to(torch.float32)
            if k in self.names:
                self.table[self.names[k]].copy_(t.reshape(self.table[self.names[k]].shape))
            else:
                self._add(k, t.clone())
        return self


# -------------------------------------------------------------------------------------------------------------------
# eager layer helpers for the generic question encoder and output unit (encoder.py, output.py): a dense layer of the
# "linearLayer<name>" family and the activation table, on the autograd nodes above
# -------------------------------------------------------------------------------------------------------------------
class _Ops:
    def __init__(self, config, store):
        self.config, self.vs = config, store

    def g(self, name):
        return get(self.config, name)

    def getWeight(self, shape):
        with self.vs.scope("weights"):
            return self.vs.get("weight", shape, "xavier")

    def getBias(self, shape):
        with self.vs.scope("biases"):
            return self.vs.get("bias", shape, "zeros")

    def act(self, name, x):
        """config.py's activation names on macx_op_act; "RELU" is whatever --relu selects (ops.py:161-187)"""
        if name == "NON":
            return x
        flavour = self.g("relu") if name == "RELU" else None
        if flavour == "PRM":
            with self.vs.scope("prelu", default=True):
                return _Act.apply(x, ACT_PRELU, self.vs.get("alpha", (x.shape[-1],), 0.25))
        if flavour == "LKY":
            raise AttributeError("'Config' object has no attribute 'reluAlpha'")
        if flavour == "SELU":
            raise UnboundLocalError("local variable 'output' referenced before assignment")
        code = {None: name, "ELU": "ELU", "STD": "RELU"}[flavour]
        return _Act.apply(x, _lib.ACT[code], None)

    def linear(self, inp, inDim, outDim, dropout=None, act="NON", name=""):
        """[.., inDim] -> [.., outDim] under "linearLayer<name>" (ops.py:298-333): input dropout, x W + b, activation and -- with an
        activation -- the stacked "<name>_2" layer.  (Widths of 1 and constant bias offsets only occur inside the cell: plan.py.)"""
        with self.vs.scope("linearLayer" + name):
            W, b = self.getWeight((inDim, outDim)), self.getBias((outDim,))
            y = self.act(act, _Linear.apply(dropout(inp) if dropout is not None else inp, W, b))
            return y if act == "NON" else self.linear(y, outDim, outDim, name=name + "_2")


# -------------------------------------------------------------------------------------------------------------------
# the cell (the interface of mac_cell.py:30-592 over a compiled plan)
# -------------------------------------------------------------------------------------------------------------------
class GenericMACCell:
    """Same constructor / zero_state / __call__ / attribute surface as mac_cell.MACCell and macx.MACCell; built by
    macx.MACCell(...) when the option set has no fused kernels.  zero_state and every step execute one segment of the plan
    compiled from the option set (plan.compile_cell) -- one autograd node each."""

    generic = True

    def __init__(self, vecQuestions, questionWords, questionCntxWords, questionLengths, knowledgeBase,
                 memoryDropout, readDropout, writeDropout, batchSize, train, reuse=None, *, config=None, params=None,
                 netLength=None, seed=None, b0=0, mask_word=None, tune=None):
        from .cell import _mask_word
        del tune            # (the A/B hooks of macx_opts.tune select among FUSED kernels; this path has one kernel per op)
        self.mask_word = _mask_word(mask_word, knowledgeBase)
        self.config = config if config is not None else SimpleNamespace()
        reject_like_reference(self.config)
        bad = [k for k in ("memDim", "ctrlDim", "attDim") if self.g(k) % 4]
        if bad:
            raise UnsupportedOptions("the generic path needs %s %% 4 == 0 (16-byte rows; widths off the 128-column granule of the "
                                     "product kernels run zero-padded inside them)" % bad[0])
        _L()                                           # fails loudly without libmacx.so
        _require_device(questionLengths, "questionLengths")
        lengths = questionLengths.to(torch.int32).contiguous()
        if lengths.shape != (vecQuestions.shape[0],):
            raise ValueError("questionLengths must be [batchSize]")
        self.train, self.batchSize, self.b0, self.reuse = bool(train), int(batchSize), int(b0), reuse
        # the reference's attribute names (mac_cell.py:59-79), device-checked fp32 tensors
        self.__dict__.update(vecQuestions=_dev(vecQuestions, "vecQuestions"), questionWords=questionWords,
                             questionCntxWords=questionCntxWords, questionLengths=lengths,
                             knowledgeBase=_dev(knowledgeBase, "knowledgeBase"))
        rates = dict(memory=memoryDropout, read=readDropout, write=writeDropout)
        self.dropouts = {k: (float(v) if self.train else 1.0) for k, v in rates.items()}      # evaluation feeds keep = 1 (model.py:118-125)
        self.netLength = int(netLength if netLength is not None else self.g("netLength"))
        self.seed = fresh_seed(seed, self.train) & 0xFFFFFFFF
        dev = self.knowledgeBase.device
        self.params = params if params is not None else GenericParams(device=dev)
        self.none = torch.zeros((self.batchSize, 1), dtype=torch.float32, device=dev)
        self.iteration = 0
        self._plan = self._exec = None

    def g(self, name):
        return get(self.config, name)

    @property
    def state_size(self):
        return MACCellTuple(self.g("ctrlDim"), self.g("memDim"))

    @property
    def output_size(self):
        return 1

    def plan(self):
        """the compiled plan of this option set (first call: compiles it and creates the variables it names, in its order --
        the reference's creation order -- unless a checkpoint already put them into `params`)"""
        if self._plan is None:
            from . import plan as _plan
            self._plan = _plan.compile_cell(self.config, self.netLength)
            for name, spec in self._plan.variables.items():
                self.params.ensure(name, spec.shape, spec.init)
        return self._plan

    def _segment(self, seg, feeds):
        from . import plan as _plan
        if self._exec is None:
            self._exec = _plan._Exec(self.seed, self.b0, self.dropouts, self.train, self.batchSize, self.knowledgeBase.device,
                                     mask_word=self.mask_word)
        return _plan.run_segment(seg, self._exec, feeds, self.params)

    # histories as the reference exposes them: [B, steps + 1, d] (mac_cell.py:472-474, 549-551)
    controls = property(lambda self: torch.stack(self._hist["control"], dim=1))
    memories = property(lambda self: torch.stack(self._hist["memory"], dim=1))
    infos = property(lambda self: torch.stack(self._hist["info"], dim=1))

    # ---- zero_state (mac_cell.py:539-592)
    def zero_state(self, batchSize=None, dtype=torch.float32):
        if batchSize is not None and int(batchSize) != self.batchSize:
            raise ValueError("zero_state(%d) on a cell built for batchSize %d" % (batchSize, self.batchSize))
        words = _dev(self.questionCntxWords if self.g("controlContextual") else self.questionWords, "question words")
        out = self._segment(self.plan().init, {"vecQuestions": self.vecQuestions, "words": words})
        self._carry = {k: out[k] for k in ("in_words", "out_words", "mem_mask") if k in out}
        self._carry["cont_control"] = out["control"]
        self._hist = {"control": [out["control"]], "memory": [out["memory"]], "info": [out["memory"]]}
        self.attentions = {"kb": [], "question": [], "self": [], "gate": []}
        self.contControl = out["control"]
        self.iteration = 0
        return MACCellTuple(out["control"], out["memory"])

    # ---- one step (mac_cell.py:420-480)
    def __call__(self, inputs, state, scope=None):
        if scope not in (None, "MACCell"):
            raise UnsupportedOptions("the plan names the cell's variables under MACnetwork/MACCell (model.py:453-458 passes no scope)")
        seg = self.plan().steps[int(self.iteration)]
        have = {"vecQuestions": self.vecQuestions, "knowledgeBase": self.knowledgeBase, "lengths": self.questionLengths,
                "control": state.control, "memory": state.memory, **self._carry}
        if "controls" in seg.feeds:
            have["controls"], have["memories"] = self.controls, self.memories
        out = self._segment(seg, {k: have[k] for k in seg.feeds})
        self._carry["cont_control"] = self.contControl = out["cont_control"]
        for key, name in (("question", "att_question"), ("kb", "att_kb"), ("self", "att_self"), ("gate", "att_gate")):
            if name in out:
                self.attentions[key].append(out[name])
        for key in ("control", "memory", "info"):
            self._hist[key].append(out[key])
        return self.none, MACCellTuple(out["control"], out["memory"])

    # ---- the units on their own, with the reference's method signatures (mac_cell.py:133, 209, 305): each call compiles (once)
    #      and runs the unit's segment of the plan for the current `iteration`; attentions are appended like in a step
    def _unit(self, which, feeds):
        from . import plan as _plan
        key = (which, int(self.iteration))
        cache = self.__dict__.setdefault("_unit_plans", {})
        if key not in cache:
            cache[key] = _plan.compile_unit(self.config, which, int(self.iteration))
            for name, spec in cache[key].variables.items():
                self.params.ensure(name, spec.shape, spec.init)
        seg = cache[key].init
        have = dict(feeds, lengths=self.questionLengths, **getattr(self, "_carry", {}))
        if "controls" in seg.feeds:
            have["controls"], have["memories"] = self.controls, self.memories
        out = self._segment(seg, {k: have[k] for k in seg.feeds})
        for k, name in (("question", "att_question"), ("kb", "att_kb"), ("self", "att_self"), ("gate", "att_gate")):
            if name in out:
                self.attentions[k].append(out[name])
        return out

    def control(self, controlInput, inWords, outWords, questionLengths, control, contControl=None, name=""):
        out = self._unit("control", {"control_input": controlInput, "in_words": inWords, "out_words": outWords, "control": control,
                                     "cont_control": contControl})
        return out["control"], out["cont_control"]

    def read(self, knowledgeBase, memory, control, name=""):
        return self._unit("read", {"knowledgeBase": knowledgeBase, "memory": memory, "control": control})["info"]

    def write(self, memory, info, control, contControl=None, name=""):
        return self._unit("write", {"memory": memory, "info": info, "control": control, "cont_control": contControl})["memory"]

    inWords = property(lambda self: self._carry["in_words"])
    outWords = property(lambda self: self._carry["out_words"])

    # ---- the loop of model.py:453-458
    def run(self):
        state = self.zero_state(self.batchSize)
        for i in range(self.netLength):
            self.iteration = i
            _, state = self(self.none, state)
        return state
