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
    def __init__(self, config, answerWordsNum=None, generator=None):
        super().__init__()
        g = lambda n, dflt: getattr(config, n, dflt)
        if not g("outQuestion", False) or g("outQuestionMul", False) or g("outImage", False) or g("answerMod", "NON") != "NON":
            raise UnsupportedOptions("output unit: only --outQuestion (no outQuestionMul/outImage/answerMod) has a HIP path")
        dims = list(g("outClassifierDims", [512]))
        if len(dims) != 1:
            raise UnsupportedOptions("classifier: exactly one hidden layer (outClassifierDims=[h]) has a HIP path")
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
