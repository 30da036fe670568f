"""Output unit + classifier (model.py:512-576, ops.py:349-359) over libmacx.so: the consumer of the
cell's final memory, and the tensor the parity bar is stated on (classifier logits within 1e-4,
identical answer argmax).  SURVEY.md 8f row 2.

    out = OutputClassifier(config, answerWordsNum=28).to(device)
    logits = out(memory, vecQuestions, train=True, seed=step)       # [B, answers], differentiable
"""
import ctypes as C
import math

import torch

from . import _lib
from .options import UnsupportedOptions, _resolve_act, fresh_seed

def _ptr(t):
    return C.c_void_p(t.data_ptr())


REF_NAMES = {
    "outQuestion_W": "outputUnit/linearLayeroutQuestion/weights/weight",
    "outQuestion_b": "outputUnit/linearLayeroutQuestion/biases/bias",
    "fc0_W": "classifier/linearLayerfc_0/weights/weight", "fc0_b": "classifier/linearLayerfc_0/biases/bias",
    "fc1_W": "classifier/linearLayerfc_1/weights/weight", "fc1_b": "classifier/linearLayerfc_1/biases/bias",
}


class _OutFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, keep, seed, b0, memory, vecQ, *params):
        L = _lib.lib()
        B, d = memory.shape
        sh = _lib.MacxOutShapes(B=B, d=d, hidden=mod.hidden, answers=mod.answers, b0=b0)
        ps = _lib.MacxOutParams(*[p.data_ptr() for p in params])
        n_saved = L.macx_output_saved_floats(C.byref(sh))
        if n_saved == 0:
            raise ValueError("output unit: d and hidden must be multiples of 16")
        saved = torch.empty(n_saved, dtype=torch.float32, device=memory.device)
        logits = torch.empty(B, mod.answers, dtype=torch.float32, device=memory.device)
        memory, vecQ = memory.contiguous(), vecQ.contiguous()
        st = C.c_void_p(torch.cuda.current_stream(memory.device).cuda_stream)
        _lib.check(L.macx_output_forward(C.byref(sh), mod.act, keep, seed & 0xFFFFFFFF, C.byref(ps), memory.data_ptr(), vecQ.data_ptr(),
                                         logits.data_ptr(), saved.data_ptr(), n_saved, st), "macx_output_forward")
        ctx.stuff = (mod, keep, seed, sh, saved, n_saved, memory, vecQ, params)
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        mod, keep, seed, sh, saved, n_saved, memory, vecQ, params = ctx.stuff
        L = _lib.lib()
        n_ws = L.macx_output_ws_floats(C.byref(sh))
        ws = torch.empty(n_ws, dtype=torch.float32, device=memory.device)
        grads = [torch.empty_like(p) for p in params]
        gs = _lib.MacxOutGrads(*[g.data_ptr() for g in grads])
        ps = _lib.MacxOutParams(*[p.data_ptr() for p in params])
        dmem, dvq = torch.empty_like(memory), torch.empty_like(vecQ)
        d_logits = d_logits.contiguous()
        st = C.c_void_p(torch.cuda.current_stream(memory.device).cuda_stream)
        _lib.check(L.macx_output_backward(C.byref(sh), mod.act, keep, seed & 0xFFFFFFFF, C.byref(ps), memory.data_ptr(), vecQ.data_ptr(),
                                          saved.data_ptr(), n_saved, ws.data_ptr(), n_ws, d_logits.data_ptr(), C.byref(gs),
                                          dmem.data_ptr(), dvq.data_ptr(), st), "macx_output_backward")
        return (None, None, None, None, dmem, dvq) + tuple(grads)


class OutputClassifier(torch.nn.Module):
    """Fused output unit (--outQuestion, one hidden layer).  Constructing it for any other legal option set returns a
    GenericOutputClassifier: the same interface on one HIP kernel per reference op."""

    def __new__(cls, config=None, *args, **kw):
        if cls is OutputClassifier and config is not None:
            try:
                _check_fused(config)
            except UnsupportedOptions:
                return GenericOutputClassifier(config, *args, **kw)
        return super().__new__(cls)

    def __init__(self, config, answerWordsNum=None, generator=None):
        super().__init__()
        g = lambda n, dflt: getattr(config, n, dflt)
        _check_fused(config)
        dims = list(g("outClassifierDims", [512]))
        d = g("memDim", 512)
        self.d, self.hidden = d, dims[0]
        self.answers = int(answerWordsNum if answerWordsNum is not None else g("answerWordsNum", 28))
        self.act = _resolve_act(config, "RELU")          # FCLayer's default act (ops.py:349)
        self.keep = float(g("outputDropout", 0.85))
        shapes = {"outQuestion_W": (d, d), "outQuestion_b": (d,), "fc0_W": (2 * d, self.hidden), "fc0_b": (self.hidden,),
                  "fc1_W": (self.hidden, self.answers), "fc1_b": (self.answers,)}
        for f in _lib.OUT_FIELDS:
            sh = shapes[f]
            if f.endswith("_b"):
                t = torch.zeros(sh, dtype=torch.float64)
            else:
                lim = math.sqrt(6.0 / (sh[0] + sh[1]))       # xavier-uniform, ops.py:20
                t = (torch.rand(sh, generator=generator, dtype=torch.float64) * 2 - 1) * lim
            self.register_parameter(f, torch.nn.Parameter(t.float()))

    def tensors(self):
        return [getattr(self, f) for f in _lib.OUT_FIELDS]

    def to_reference_dict(self):
        return {REF_NAMES[f]: getattr(self, f).detach().clone() for f in _lib.OUT_FIELDS}

    def forward(self, memory, vecQuestions, train=False, seed=None, b0=0):
        if not memory.is_cuda:
            raise RuntimeError("the output unit has no CPU path")
        keep = self.keep if train else 1.0          # model.py:118-125
        return _OutFunction.apply(self, keep, fresh_seed(seed, train), int(b0), memory, vecQuestions, *self.tensors())


def _check_fused(config):
    g = lambda n, dflt: getattr(config, n, dflt)
    if not g("outQuestion", False) or g("outQuestionMul", False) or g("outImage", False) or g("answerMod", "NON") != "NON":
        raise UnsupportedOptions("output unit: only --outQuestion (no outQuestionMul/outImage/answerMod) has a fused HIP path")
    if len(list(g("outClassifierDims", [512]))) != 1:
        raise UnsupportedOptions("classifier: exactly one hidden layer (outClassifierDims=[h]) has a fused HIP path")
    if g("outputBN", False):
        raise UnsupportedOptions("--outputBN has no fused HIP path")
    if g("relu", "ELU") == "PRM":
        raise UnsupportedOptions("--relu PRM has no fused HIP path in the classifier")


def fc_site(layer):
    """Dropout site of the input of classifier layer `layer` on the stateless stream (macx_common.hip.h: 7, 8; 13, 14, ... for
    deeper classifiers)."""
    return 7 + layer if layer < 2 else 11 + layer


class GenericOutputClassifier(torch.nn.Module):
    """outputOp (model.py:512-528) + classifier (model.py:547-576) for the legal option sets the fused kernels refuse --
    no --outQuestion, --outQuestionMul, any number of classifier layers, --relu PRM/STD -- as ONE HIP KERNEL PER REFERENCE
    OP (generic._Ops on macx_linear / macx_wgrad / macx_op_*), variables under the reference's names in its creation
    order.  --outImage and --answerMod != NON raise what the reference raises (its own calls do not match its own
    signatures); --outputBN is refused.  Layer widths must be multiples of 128 (the last layer's answerWordsNum is padded
    with zero columns inside the call)."""

    def __init__(self, config, answerWordsNum=None, generator=None, device=None):
        super().__init__()
        from .generic import GenericParams, _Ops
        dflt = dict(outImage=False, answerMod="NON", outputBN=False, outputDropout=0.85, memDim=512, ctrlDim=512,
                    outClassifierDims=[512], outQuestion=False, outQuestionMul=False, relu="STD")        # config.py:196, 281-288
        g = lambda n: getattr(config, n, dflt[n])
        if g("outImage"):
            # model.py:521-522 calls ops.linearizeFeatures(..., outputDim=...); its parameter is `outDim` (ops.py:595)
            raise TypeError("linearizeFeatures() got an unexpected keyword argument 'outputDim'")
        if g("answerMod") == "DIAG":
            raise UnboundLocalError("local variable 'output' referenced before assignment")      # ops.py:704-707 via model.py:561
        if g("answerMod") != "NON":
            raise NameError("name 'outputDim' is not defined")          # model.py:563
        if g("outputBN"):
            raise UnsupportedOptions("--outputBN: batch normalisation inside ops.linear has no HIP path")
        self.config = config
        self.answers = int(answerWordsNum if answerWordsNum is not None else getattr(config, "answerWordsNum", 28))
        self.keep = float(g("outputDropout"))
        self.d, self.ctrl = int(g("memDim")), int(g("ctrlDim"))
        self.dims = [int(x) for x in g("outClassifierDims")]
        self.params = GenericParams(device=device, generator=generator)
        self.ops = _Ops(config, self.params)
        self._declare()

    # the variables, in the order the reference's graph creates them (the forward pass below finds them by name)
    def _declare(self):
        vs, ops, cfg = self.params, self.ops, self.config
        dim = self.d
        with vs.scope("outputUnit"):
            if getattr(cfg, "outQuestion", False):
                with vs.scope("linearLayeroutQuestion"):
                    ops.getWeight((self.ctrl, self.d))
                    ops.getBias((self.d,))
                dim = 3 * self.d if getattr(cfg, "outQuestionMul", False) else 2 * self.d
        self.in_dim = dim
        dims = [dim] + self.dims + [self.answers]
        with vs.scope("classifier"):
            for i in range(len(dims) - 1):
                with vs.scope("linearLayerfc_%d" % i):
                    ops.getWeight((dims[i], dims[i + 1]))
                    ops.getBias((dims[i + 1],))
                if i < len(dims) - 2 and getattr(cfg, "relu", "ELU") == "PRM":
                    with vs.scope("prelu", default=True):
                        vs.get("alpha", (dims[i + 1],), 0.25)

    def tensors(self):
        return self.params.tensors()

    def to_reference_dict(self):
        return self.params.to_reference_dict()

    def load_reference_dict(self, ref):
        own = set(self.params.names)
        self.params.load_reference_dict({k: v for k, v in ref.items() if k in own})
        return self

    def to(self, *a, **kw):
        out = super().to(*a, **kw)
        t = self.params.tensors()
        self.params.device = t[0].device if t else self.params.device
        return out

    def forward(self, memory, vecQuestions, train=False, seed=None, b0=0):
        from . import generic as G
        from .generic import B_SAME, OP_MUL, _Binary, _Dropout, _Linear
        G._require_device(memory, "memory")                              # no CPU path
        vs, ops, cfg = self.params, self.ops, self.config
        keep = self.keep if train else 1.0                               # model.py:118-125
        seed = fresh_seed(seed, train)
        features, dim = memory, self.d
        with vs.scope("outputUnit"):
            if getattr(cfg, "outQuestion", False):
                eq = ops.linear(vecQuestions, self.ctrl, self.d, name="outQuestion")
                parts = [features, eq]
                if getattr(cfg, "outQuestionMul", False):
                    parts.append(_Binary.apply(features.contiguous(), eq, OP_MUL, B_SAME, 1.0))
                features = torch.cat(parts, dim=-1)                       # ops.concat (ops.py:65-78)
                dim = self.d * len(parts)
        dims = [dim] + self.dims + [self.answers]
        with vs.scope("classifier"):                                     # ops.FCLayer (ops.py:349-359), act = RELU -> config.relu
            for i in range(len(dims) - 1):
                drop = None
                if keep < 1.0:
                    per_q = dims[i]
                    drop = (lambda x, i=i, per_q=per_q: _Dropout.apply(x, seed, fc_site(i), 0, keep, int(b0) * per_q))
                last = i == len(dims) - 2
                if last and dims[i + 1] % 128:
                    # answerWordsNum is not a multiple of the kernels' 128-column granule: zero columns are appended for the
                    # call and cut off again (memory ops only; their gradient is exactly zero)
                    with vs.scope("linearLayerfc_%d" % i):
                        W, b = ops.getWeight((dims[i], dims[i + 1])), ops.getBias((dims[i + 1],))
                    pad = (-dims[i + 1]) % 128
                    x = drop(features) if drop is not None else features
                    Wp = torch.cat([W, W.new_zeros(dims[i], pad)], dim=1)
                    bp = torch.cat([b, b.new_zeros(pad)])
                    features = _Linear.apply(x, Wp, bp)[:, : dims[i + 1]]
                else:
                    features = ops.linear(features, dims[i], dims[i + 1], dropout=drop, name="fc_%d" % i)
                if not last:
                    features = ops.act("RELU", features)
        return features


class _AnswerLoss(torch.autograd.Function):
    """macx_answer_loss: per-question cross-entropy + argmax in one kernel; the backward pass returns the dlogits the same
    kernel wrote (gradient of the MEAN loss), scaled by the incoming gradient."""

    @staticmethod
    def forward(ctx, logits, answers):
        if not logits.is_cuda:
            raise RuntimeError("logits must live on the HIP device: the loss has no CPU path")
        L = _lib.lib()
        lg = logits.detach().contiguous()
        B, A = lg.shape
        ans = answers.to(device=lg.device, dtype=torch.int32).contiguous()      # out-of-range labels: NaN loss, as TF on a GPU
        rows = torch.empty(B, dtype=torch.float32, device=lg.device)
        pred = torch.empty(B, dtype=torch.int32, device=lg.device)
        dl = torch.empty_like(lg)
        st = C.c_void_p(torch.cuda.current_stream(lg.device).cuda_stream)
        _lib.check(L.macx_answer_loss(_ptr(lg), _ptr(ans), B, A, _ptr(rows), _ptr(pred), _ptr(dl), 1.0 / B, st), "macx_answer_loss")
        ctx.save_for_backward(dl)
        ctx.mark_non_differentiable(pred)
        return rows, pred

    @staticmethod
    def backward(ctx, g_rows, _g_pred):
        (dl,) = ctx.saved_tensors
        # rows feed a mean: g_rows is constant 1/B * upstream; dl already carries 1/B, so scale by B * g_rows per row
        return dl * (g_rows * dl.shape[0]).unsqueeze(1), None


def answer_loss_and_pred(logits, answers):
    """addAnswerLossOp (model.py:593-599) + addPredOp (model.py:603-612): mean sparse CE, int32 argmax -- one HIP kernel
    (macx_answer_loss) for both and for the gradient."""
    rows, pred = _AnswerLoss.apply(logits, answers)
    return rows.mean(), pred
