"""The GENERIC option path of the MAC cell: every legal option combination of mac_cell.py that the fused cell kernels
(cell.py -> macx_cell_*) answer with UnsupportedOptions -- the reference's DEFAULT configuration, writeInputs != BOTH,
readMemAttType BL / ADD, relu = PRM, writeConcatMul, controlProj, controlConcatWords, unsharedCells, mulBias, ... -- runs
here as ONE HIP KERNEL PER REFERENCE OP, chained exactly as mac_cell.py:133-375, 420-480 chains ops.py.

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


def k_dropout(x, seed, site, step, keep, first):
    out = torch.empty_like(x)
    _lib.check(_L().macx_op_dropout(_p(x), x.numel(), seed, site, step, keep, first, _p(out), _st(x)), "macx_op_dropout")
    return out


def k_matmul(x, W, b, big):
    """x [rows, K] @ W [K, n] (+ b): the knowledge-base GEMM for [B, N, .] operands (big = (B, N)), macx_linear otherwise."""
    L = _L()
    rows, K = x.shape
    n = W.shape[1]
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
        if K % 128 or n % 128:
            raise UnsupportedOptions("the generic path needs layer widths that are multiples of 128 (got %d -> %d)" % (K, n))
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
# ops.py on the kernels
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

    # ops.activations (ops.py:181-187) / ops.relu (ops.py:161-179)
    def act(self, name, x):
        if name == "NON":
            return x
        if name == "RELU":
            r = self.g("relu")
            if r == "PRM":
                with self.vs.scope("prelu", default=True):
                    alpha = self.vs.get("alpha", (x.shape[-1],), 0.25)
                return _Act.apply(x, ACT_PRELU, alpha)
            if r == "LKY":
                raise AttributeError("'Config' object has no attribute 'reluAlpha'")
            if r == "SELU":
                raise UnboundLocalError("local variable 'output' referenced before assignment")
            name = "ELU" if r == "ELU" else "RELU"
        return _Act.apply(x, _lib.ACT[name], None)

    # tf.contrib.layers.batch_norm(updates_collections=None) on the last axis of a [B, c] tensor (mac_cell.py:370-373):
    # batch statistics (biased variance) in training, moving averages in evaluation; eps = 0.001
    def batch_norm(self, x, decay, center, scale, is_training, epsilon=0.001):
        with self.vs.scope("BatchNorm", default=True):
            c = x.shape[-1]
            beta = self.vs.get("beta", (c,), "zeros") if center else None
            gamma = self.vs.get("gamma", (c,), 1.0) if scale else None
            mm = self.vs.get("moving_mean", (c,), "zeros")
            mv = self.vs.get("moving_variance", (c,), 1.0)
            B = x.shape[0]
            ones_c = torch.ones(c, dtype=torch.float32, device=x.device)
            eps = torch.full((1,), epsilon, dtype=torch.float32, device=x.device)
            x = x.contiguous()
            if is_training:
                mean = _Binary.apply(_RowSum.apply(x), ones_c, OP_MUL, B_SAME, 1.0 / B)
                cen = _Binary.apply(x, _Binary.apply(mean, ones_c, OP_MUL, B_SAME, -1.0), OP_ADD, B_CHANNEL, 1.0)
                var = _Binary.apply(_RowSum.apply(_Binary.apply(cen, cen, OP_MUL, B_SAME, 1.0)), ones_c, OP_MUL, B_SAME, 1.0 / B)
                with torch.no_grad():          # assign_moving_average: m -= (1 - decay) (m - stat), outside the gradient
                    for mov, stat in ((mm, mean), (mv, var)):
                        kept = k_binary(OP_MUL, B_SAME, mov.detach().contiguous(), ones_c, 1, c, decay)
                        mov.copy_(k_binary(OP_ADD, B_SAME, kept, k_binary(OP_MUL, B_SAME, stat.detach().contiguous(), ones_c, 1, c, 1.0 - decay), 1, c))
            else:
                cen = _Binary.apply(x, _Binary.apply(mm.detach(), ones_c, OP_MUL, B_SAME, -1.0), OP_ADD, B_CHANNEL, 1.0)
                var = mv.detach()
            out = _Binary.apply(cen, _Act.apply(var, ACT_RSQRT_EPS, eps), OP_MUL, B_CHANNEL, 1.0)
            if gamma is not None:
                out = _Binary.apply(out, gamma, OP_MUL, B_CHANNEL, 1.0)
            if beta is not None:
                out = _Binary.apply(out, beta, OP_ADD, B_CHANNEL, 1.0)
        return out

    # ops.linear (ops.py:298-333)
    def linear(self, inp, inDim, outDim, dropout=None, addBias=True, bias=0.0, act="NON", actLayer=True, name=""):
        with self.vs.scope("linearLayer" + name):
            W = self.getWeight((inDim, outDim) if outDim > 1 else (inDim,))
            b = self.getBias((outDim,) if outDim > 1 else ())
            if dropout is not None:
                inp = dropout(inp)
            if outDim > 1:
                bb = None
                if addBias:
                    bb = b if bias == 0.0 else _Binary.apply(b, torch.full_like(b, bias), OP_ADD, B_SAME, 1.0)
                output = _Linear.apply(inp, W, bb)
            else:
                # ops.py:317: reduce_sum(inp * W, axis=-1)
                output = _Reduce.apply(_Binary.apply(inp, W, OP_MUL, B_CHANNEL, 1.0), R_LAST)
                if addBias:
                    flat = output.reshape(-1, 1)
                    output = _Binary.apply(flat, (b.reshape(1) + bias) if bias else b.reshape(1), OP_ADD, B_CHANNEL, 1.0).reshape(output.shape)
            output = self.act(act, output)
            if act != "NON" and actLayer:
                output = self.linear(output, outDim, outDim, addBias=addBias, act="NON", actLayer=False, name=name + "_2")
        return output

    # ops.inter2logits / inter2att (ops.py:114-146)
    def inter2logits(self, interactions, dim, dropout=None, name=""):
        with self.vs.scope("inter2logits" + name):
            return self.linear(interactions, dim, 1, dropout=dropout, name="logits")

    def inter2att(self, interactions, dim, dropout=None, name=""):
        with self.vs.scope("inter2att" + name):
            return _Softmax.apply(self.inter2logits(interactions, dim, dropout=dropout), None)

    # ops.att2Smry (ops.py:150)
    @staticmethod
    def att2Smry(attention, features):
        B, N, d = features.shape
        w = _Binary.apply(features.contiguous(), attention.reshape(-1), OP_MUL, B_ROW, 1.0)
        return _Reduce.apply(w, R_MID)

    @staticmethod
    def bcast_mul(x, y):
        """x [B,N,d] * y [B,d] (the extendY broadcast of ops.mul / ops.concat, ops.py:65-67, 693-695)."""
        return _Binary.apply(x.contiguous(), y, OP_MUL, B_MID, 1.0)

    # ops.mul (ops.py:668-725)
    def mul(self, x, y, dim, proj=None, interMod="MUL", concat=None, extendY=True, name="", drops=None):
        drops = drops or {}
        with self.vs.scope("mul" + name):
            origVals = {"x": x, "y": y, "dim": dim}
            projVals = None
            if proj is not None:
                if drops.get("x") is not None:
                    x = drops["x"](x)
                if drops.get("y") is not None:
                    y = drops["y"](y)
                xName, yName = ("proj", "proj") if proj["shared"] else ("projX", "projY")
                x = self.linear(x, dim, proj["dim"], name=xName)
                y = self.linear(y, dim, proj["dim"], name=yName)
                dim = proj["dim"]
                projVals = {"x": x, "y": y, "dim": dim}
                proj["x"], proj["y"] = x, y
            mulBias = self.g("mulBias")
            if interMod == "MUL":
                if mulBias != 0.0:
                    x = _Binary.apply(x.contiguous(), torch.full((x.shape[-1],), mulBias, device=x.device), OP_ADD, B_CHANNEL, 1.0)
                    y = _Binary.apply(y.contiguous(), torch.full((y.shape[-1],), mulBias, device=y.device), OP_ADD, B_CHANNEL, 1.0)
                output = self.bcast_mul(x, y) if extendY else _Binary.apply(x.contiguous(), y, OP_MUL, B_SAME, 1.0)
            elif interMod == "DIAG":
                raise UnboundLocalError("local variable 'output' referenced before assignment")
            elif interMod == "BL":
                W = self.getWeight((dim, dim))
                b = self.getBias((dim,))
                output = self.bcast_mul(_Linear.apply(x, W, None), y)
                output = _Binary.apply(output, b, OP_ADD, B_CHANNEL, 1.0)
            else:   # "ADD": tanh(x + y)
                output = _Act.apply(_Binary.apply(x.contiguous(), y, OP_ADD, B_MID if extendY else B_SAME, 1.0), _lib.ACT["TANH"], None)
            if concat is not None:
                if concat.get("proj", False):
                    if projVals is None:
                        raise UnboundLocalError("local variable 'projVals' referenced before assignment")
                    concatVals = projVals
                else:
                    concatVals = origVals
                if concat.get("x", False):
                    output = torch.cat([output, concatVals["x"]], dim=-1)
                    dim += concatVals["dim"]
        return output, dim


# -------------------------------------------------------------------------------------------------------------------
# the cell (mac_cell.py:30-592)
# -------------------------------------------------------------------------------------------------------------------
class GenericMACCell:
    """Same constructor / zero_state / __call__ / attribute surface as mac_cell.MACCell and macx.MACCell; built by
    macx.MACCell(...) when the option set has no fused kernels."""

    generic = True

    def __init__(self, vecQuestions, questionWords, questionCntxWords, questionLengths, knowledgeBase,
                 memoryDropout, readDropout, writeDropout, batchSize, train, reuse=None, *, config=None, params=None,
                 netLength=None, seed=None, b0=0):
        self.config = config if config is not None else SimpleNamespace()
        reject_like_reference(self.config)
        g = self.g
        for dname in ("memDim", "ctrlDim", "attDim"):
            if g(dname) % 128:
                raise UnsupportedOptions("the generic path needs %s %% 128 == 0" % dname)
        _L()                                           # fails loudly without libmacx.so
        self.netLength = int(netLength if netLength is not None else g("netLength"))
        self.vecQuestions = _dev(vecQuestions, "vecQuestions")
        self.questionWords, self.questionCntxWords = questionWords, questionCntxWords
        _require_device(questionLengths, "questionLengths")
        self.questionLengths = questionLengths.to(torch.int32).contiguous()
        if self.questionLengths.shape != (self.vecQuestions.shape[0],):
            raise ValueError("questionLengths must be [batchSize]")
        self.knowledgeBase = _dev(knowledgeBase, "knowledgeBase")
        self.train = bool(train)
        self.dropouts = {"memory": float(memoryDropout) if train else 1.0, "read": float(readDropout) if train else 1.0,
                         "write": float(writeDropout) if train else 1.0}
        self.batchSize = int(batchSize)
        self.reuse = reuse
        self.seed = fresh_seed(seed, self.train) & 0xFFFFFFFF
        self.b0 = int(b0)
        dev = self.knowledgeBase.device
        self.params = params if params is not None else GenericParams(device=dev)
        self.vs = self.params
        self.ops = _Ops(self.config, self.params)
        self.none = torch.zeros((self.batchSize, 1), dtype=torch.float32, device=dev)
        self.iteration = 0

    def g(self, name):
        return get(self.config, name)

    @property
    def state_size(self):
        return MACCellTuple(self.g("ctrlDim"), self.g("memDim"))

    @property
    def output_size(self):
        return 1

    def _drop(self, site, keep, step=None):
        """x -> tf.nn.dropout(x, keep) on the stateless stream; None when keep == 1 (the reference still builds the op)."""
        if keep == 1.0:
            return None
        st = 0 if site == SITE_MEM_VAR else (self.iteration if step is None else step)

        def f(x):
            per_q = x.numel() // x.shape[0]
            return _Dropout.apply(x, self.seed, site, st, keep, self.b0 * per_q)
        return f

    # ---- control (mac_cell.py:133-187)
    def control(self, controlInput, inWords, outWords, questionLengths, control, contControl=None, name=""):
        g, ops = self.g, self.ops
        with self.vs.scope("control" + name):
            dim = g("ctrlDim")
            newContControl = controlInput
            if g("controlFeedPrev"):
                newContControl = control if g("controlFeedPrevAtt") else contControl
                if g("controlFeedInputs"):
                    newContControl = torch.cat([newContControl, controlInput], dim=-1)
                    dim += g("ctrlDim")
                newContControl = ops.linear(newContControl, dim, g("ctrlDim"), act=g("controlContAct"), name="contControl")
                dim = g("ctrlDim")
            interactions = ops.bcast_mul(inWords, newContControl)
            if g("controlConcatWords"):
                interactions = torch.cat([interactions, inWords], dim=-1)
                dim += g("ctrlDim")
            if g("controlProj"):
                interactions = ops.linear(interactions, dim, g("ctrlDim"), act=g("controlProjAct"))
                dim = g("ctrlDim")
            logits = ops.inter2logits(interactions, dim)
            attention = _Softmax.apply(logits, questionLengths)      # softmax(expMask(logits, lengths)), mac_cell.py:176-177
            self.attentions["question"].append(attention)
            newControl = ops.att2Smry(attention, outWords)
            if g("controlContinuous"):
                newControl = newContControl
        return newControl, newContControl

    # ---- read (mac_cell.py:209-277)
    def read(self, knowledgeBase, memory, control, name=""):
        g, ops = self.g, self.ops
        with self.vs.scope("read" + name):
            dim = g("memDim")
            if g("memoryVariationalDropout"):
                if self.memDpMask is not None:
                    memory = _Binary.apply(memory.contiguous(), self.memDpMask, OP_MUL, B_SAME, 1.0)
            else:
                dm = self._drop(SITE_MEM, self.dropouts["memory"])
                memory = dm(memory) if dm else memory
            proj = None
            if g("readProjInputs"):
                proj = {"dim": g("attDim"), "shared": g("readProjShared")}
                dim = g("attDim")
            concat = {"x": g("readMemConcatKB"), "proj": g("readMemConcatProj")}
            drops = {"x": self._drop(SITE_READ_KB, self.dropouts["read"]),
                     "y": self._drop(SITE_READ_MEM, self.dropouts["read"])} if proj else None
            interactions, interDim = ops.mul(x=knowledgeBase, y=memory, dim=g("memDim"), proj=proj, concat=concat,
                                             interMod=g("readMemAttType"), name="memInter", drops=drops)
            projectedKB = proj.get("x") if proj else None
            if g("readMemProj"):
                interactions = ops.linear(interactions, interDim, dim, act=g("readMemAct"), name="memKbProj")
            else:
                dim = interDim
            if g("readCtrl"):
                if g("ctrlDim") != dim:
                    raise NameError("name 'ctrlDim' is not defined")          # mac_cell.py:246
                interactions, interDim = ops.mul(interactions, control, dim, interMod=g("readCtrlAttType"),
                                                 concat={"x": g("readCtrlConcatInter")}, name="ctrlInter")
                if g("readCtrlConcatKB"):
                    if g("readCtrlConcatProj"):
                        addedInp, addedDim = projectedKB, g("attDim")
                    else:
                        addedInp, addedDim = knowledgeBase, g("memDim")
                    interactions = torch.cat([interactions, addedInp], dim=-1)
                    dim += addedDim
                interactions = ops.act(g("readCtrlAct"), interactions)
            if interactions.shape[-1] != dim:
                raise ValueError("Dimensions must be equal, but are %d and %d" % (interactions.shape[-1], dim))
            attention = ops.inter2att(interactions, dim, dropout=self._drop(SITE_READ_ATT, self.dropouts["read"]))
            self.attentions["kb"].append(attention)
            if g("readSmryKBProj"):
                knowledgeBase = projectedKB
            information = ops.att2Smry(attention, knowledgeBase)
        return information

    # ---- write (mac_cell.py:305-375)
    def write(self, memory, info, control, contControl=None, name=""):
        g, ops = self.g, self.ops
        with self.vs.scope("write" + name):
            if g("writeInfoProj"):
                info = ops.linear(info, g("memDim"), g("memDim"), name="info")
            info = ops.act(g("writeInfoAct"), info)
            if g("writeSelfAtt"):
                selfControl = contControl if g("writeSelfAttMod") == "CONT" else control
                selfControl = ops.linear(selfControl, g("ctrlDim"), g("ctrlDim"), name="ctrlProj")
                interactions = ops.bcast_mul(self.controls, selfControl)
                attention = ops.inter2att(interactions, g("ctrlDim"), name="selfAttention")
                self.attentions["self"].append(attention)
                selfSmry = ops.att2Smry(attention, self.memories)
            newMemory, dim = memory, g("memDim")
            if g("writeInputs") == "INFO":
                newMemory = info
            elif g("writeInputs") == "SUM":
                newMemory = _Binary.apply(newMemory.contiguous(), info, OP_ADD, B_SAME, 1.0)
            elif g("writeInputs") == "BOTH":
                parts = [newMemory, info]
                if g("writeConcatMul"):
                    parts.append(_Binary.apply(newMemory.contiguous(), info, OP_MUL, B_SAME, 1.0))
                newMemory, dim = torch.cat(parts, dim=-1), dim * len(parts)
            if g("writeSelfAtt"):
                newMemory = torch.cat([newMemory, selfSmry], dim=-1)
                dim += g("memDim")
            if g("writeMergeCtrl"):
                newMemory = torch.cat([newMemory, control], dim=-1)
                dim += g("memDim")
            if g("writeMemProj") or (dim != g("memDim")):
                newMemory = ops.linear(newMemory, dim, g("memDim"), name="newMemory")
            newMemory = ops.act(g("writeMemAct"), newMemory)
            if g("writeGate"):
                gateDim = 1 if g("writeGateShared") else g("memDim")
                if gateDim == 1:
                    raise ValueError("Dimensions must be equal")              # [B,d] * [B] (ops.py:317, mac_cell.py:367)
                z = ops.act("SIGMOID", ops.linear(control, g("ctrlDim"), gateDim, name="gate", bias=g("writeGateBias")))
                self.attentions["gate"].append(z)
                # newMemory * z + memory * (1 - z)
                one_minus = _Binary.apply(_Binary.apply(z, torch.ones_like(z), OP_MUL, B_SAME, -1.0), torch.ones_like(z), OP_ADD, B_SAME, 1.0)
                newMemory = _Binary.apply(_Binary.apply(newMemory.contiguous(), z, OP_MUL, B_SAME, 1.0),
                                          _Binary.apply(memory.contiguous(), one_minus, OP_MUL, B_SAME, 1.0), OP_ADD, B_SAME, 1.0)
            if g("memoryBN"):
                newMemory = ops.batch_norm(newMemory, g("bnDecay"), g("bnCenter"), g("bnScale"), self.train)
        return newMemory

    @contextmanager
    def _net_scope(self):
        """model.py:441 wraps the cell in variable_scope("MACnetwork"); entered here unless the caller already did."""
        if "MACnetwork" in self.vs._stack:
            yield
        else:
            with self.vs.scope("MACnetwork"):
                yield

    # ---- one step (mac_cell.py:420-480)
    def __call__(self, inputs, state, scope=None):
        with self._net_scope():
            return self._step(state, scope)

    def _step(self, state, scope):
        g, ops = self.g, self.ops
        with self.vs.scope(scope or "MACCell"):
            control, memory = state
            inputNameU = "qInput%d" % self.iteration if g("controlInputUnshared") else "qInputU"
            cellName = str(self.iteration) if g("unsharedCells") else ""
            controlInput = ops.linear(self.vecQuestions, g("ctrlDim"), g("ctrlDim"), name="qInput")
            controlInput = ops.act(g("controlInputAct"), controlInput)
            controlInput = ops.linear(controlInput, g("ctrlDim"), g("ctrlDim"), name=inputNameU)
            newControl, self.contControl = self.control(controlInput, self.inWords, self.outWords, self.questionLengths,
                                                        control, self.contControl, name=cellName)
            if g("controlWholeQ"):
                newControl = self.vecQuestions
            info = self.read(self.knowledgeBase, memory, newControl, name=cellName)
            dw = self._drop(SITE_WRITE_INFO, self.dropouts["write"])
            if dw:
                info = dw(info)
            newMemory = self.write(memory, info, newControl, self.contControl, name=cellName)
            self.controls = torch.cat([self.controls, newControl.unsqueeze(1)], dim=1)
            self.memories = torch.cat([self.memories, newMemory.unsqueeze(1)], dim=1)
            self.infos = torch.cat([self.infos, info.unsqueeze(1)], dim=1)
        return self.none, MACCellTuple(newControl, newMemory)

    def initState(self, name, dim, initType, batchSize):
        if initType == "PRM":
            prm = self.vs.get(name, (dim,), "normal")
            return _Binary.apply(torch.zeros((batchSize, dim), dtype=torch.float32, device=prm.device), prm, OP_ADD, B_CHANNEL, 1.0)
        if initType == "ZERO":
            return torch.zeros((batchSize, dim), dtype=torch.float32, device=self.knowledgeBase.device)
        return self.vecQuestions

    # ---- zero_state (mac_cell.py:539-592)
    def zero_state(self, batchSize=None, dtype=torch.float32):
        with self._net_scope():
            return self._zero_state(batchSize)

    def _zero_state(self, batchSize):
        g, ops = self.g, self.ops
        batchSize = self.batchSize if batchSize is None else batchSize
        self.attentions = {"kb": [], "question": [], "self": [], "gate": []}
        initialControl = self.initState("initCtrl", g("ctrlDim"), g("initCtrl"), batchSize)
        initialMemory = self.initState("initMem", g("memDim"), g("initMem"), batchSize)
        self.controls = initialControl.unsqueeze(1)
        self.memories = initialMemory.unsqueeze(1)
        self.infos = initialMemory.unsqueeze(1)
        self.contControl = initialControl
        words = self.questionCntxWords if g("controlContextual") else self.questionWords
        words = _dev(words, "question words")
        self.inWords = self.outWords = words
        if g("controlInWordsProj") or g("controlOutWordsProj"):
            pWords = ops.linear(words, g("ctrlDim"), g("ctrlDim"), name="wordsProj")
            self.inWords = pWords if g("controlInWordsProj") else words
            self.outWords = pWords if g("controlOutWordsProj") else words
        self.memDpMask = None
        keep = self.dropouts["memory"]
        if g("memoryVariationalDropout") and keep != 1.0:
            # ops.generateVarDpMask / applyVarDpMask (ops.py:1054-1067): one mask per batch, x / keep * mask
            ones = torch.ones((batchSize, g("memDim")), dtype=torch.float32, device=self.knowledgeBase.device)
            self.memDpMask = _Dropout.apply(ones, self.seed, SITE_MEM_VAR, 0, keep, self.b0 * g("memDim"))
        self.iteration = 0
        return MACCellTuple(initialControl, initialMemory)

    # ---- the loop of model.py:453-458
    def run(self):
        state = self.zero_state(self.batchSize)
        for i in range(self.netLength):
            self.iteration = i
            _, state = self(self.none, state)
        return state
