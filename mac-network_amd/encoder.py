"""Question input unit (model.py:207-219 qEmbeddingsOp, model.py:279-307 encoder, ops.py:859-911 biRNNLayer):
embedding lookup -> dropout(encInputDropout) -> bidirectional BasicLSTMCell(encDim/2) with per-question lengths ->
questionCntxWords [B,S,encDim], vecQuestions = dropout(concat(final h_fw, h_bw), qDropout).  Producer of the cell's
`vecQuestions` / `questionCntxWords` inputs; SURVEY.md 8f row 4.  All compute behind libmacx.so
(macx_encoder_forward / macx_encoder_backward); no CPU path.

    enc = QuestionEncoder(config, vocab=90).to(device)
    questionCntxWords, vecQuestions = enc(questions, lengths, train=True, seed=step)
"""
import ctypes as C
import math

import torch

from . import _lib
from .options import UnsupportedOptions, fresh_seed

REF_NAMES = {"emb": "qEmbeddings/emb",
             "fw_kernel": "encoder/birnnLayer/bidirectional_rnn/fw/basic_lstm_cell/kernel",
             "fw_bias": "encoder/birnnLayer/bidirectional_rnn/fw/basic_lstm_cell/bias",
             "bw_kernel": "encoder/birnnLayer/bidirectional_rnn/bw/basic_lstm_cell/kernel",
             "bw_bias": "encoder/birnnLayer/bidirectional_rnn/bw/basic_lstm_cell/bias"}


class _EncoderFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, keep_in, keep_q, seed, b0, questions, lengths, *params):
        L = _lib.lib()
        B, S = questions.shape
        sh = _lib.MacxEncShapes(B=B, S=S, V=mod.vocab, E=mod.E, h=mod.h, b0=b0)
        n_saved = L.macx_encoder_saved_floats(C.byref(sh))
        if n_saved == 0:
            raise ValueError("encoder: encDim/2 must be a multiple of 128")
        dev = questions.device
        saved = torch.empty(n_saved, dtype=torch.float32, device=dev)
        words = torch.empty(B, S, 2 * mod.h, dtype=torch.float32, device=dev)
        vecQ = torch.empty(B, 2 * mod.h, dtype=torch.float32, device=dev)
        ps = _lib.MacxEncParams(*[p.data_ptr() for p in params])
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(L.macx_encoder_forward(C.byref(sh), keep_in, keep_q, seed & 0xFFFFFFFF, C.byref(ps), questions.data_ptr(),
                                          lengths.data_ptr(), words.data_ptr(), vecQ.data_ptr(), saved.data_ptr(), n_saved, st),
                   "macx_encoder_forward")
        ctx.stuff = (mod, keep_in, keep_q, seed, sh, saved, n_saved, questions, lengths, params)
        return words, vecQ

    @staticmethod
    def backward(ctx, d_words, d_vecQ):
        mod, keep_in, keep_q, seed, sh, saved, n_saved, questions, lengths, params = ctx.stuff
        L = _lib.lib()
        dev = saved.device
        n_ws = L.macx_encoder_ws_floats(C.byref(sh))
        ws = torch.empty(n_ws, dtype=torch.float32, device=dev)
        grads = [torch.empty_like(p) for p in params]
        gs = _lib.MacxEncGrads(*[g.data_ptr() for g in grads])
        ps = _lib.MacxEncParams(*[p.data_ptr() for p in params])
        d_words = saved.new_zeros((sh.B, sh.S, 2 * sh.h)) if d_words is None else d_words.contiguous()
        d_vecQ = saved.new_zeros((sh.B, 2 * sh.h)) if d_vecQ is None else d_vecQ.contiguous()
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(L.macx_encoder_backward(C.byref(sh), keep_in, keep_q, seed & 0xFFFFFFFF, C.byref(ps), questions.data_ptr(),
                                           lengths.data_ptr(), saved.data_ptr(), n_saved, ws.data_ptr(), n_ws, d_words.data_ptr(),
                                           d_vecQ.data_ptr(), C.byref(gs), st), "macx_encoder_backward")
        if not mod.emb.requires_grad:
            grads[0] = None
        return (None,) * 7 + tuple(grads)


class QuestionEncoder(torch.nn.Module):
    """Mirrors `MACnet.embeddingsOp` + `MACnet.encoder` for the configuration every flag file uses
    (encType LSTM, --encBi, encNumLayers 1, encDim == ctrlDim so no projections)."""

    def __init__(self, config, vocab, embInit=None, generator=None):
        super().__init__()
        g = lambda n, dflt: getattr(config, n, dflt)
        if g("encType", "LSTM") != "LSTM" or not g("encBi", True) or g("encNumLayers", 1) != 1:
            raise UnsupportedOptions("encoder: only the bidirectional single-layer LSTM has a HIP path")
        if g("encVariationalDropout", False):
            raise UnsupportedOptions("encoder: encVariationalDropout has no HIP path")
        if g("encProj", False) or g("encDim", 512) != g("ctrlDim", 512):
            raise UnsupportedOptions("encoder: output projections (encProj / encDim != ctrlDim) have no HIP path")
        if g("ansEmbMod", "NON") == "SHARED":
            raise UnsupportedOptions("encoder: shared question/answer embeddings have no HIP path")
        self.vocab, self.E, self.h = int(vocab), int(g("wrdEmbDim", 300)), int(g("encDim", 512)) // 2
        self.keep_in, self.keep_q = float(g("encInputDropout", 0.85)), float(g("qDropout", 0.92))
        E, h = self.E, self.h
        if embInit is None:
            emb = torch.randn((self.vocab, E), generator=generator, dtype=torch.float64)
        else:
            emb = torch.as_tensor(embInit, dtype=torch.float64)
            if tuple(emb.shape) != (self.vocab, E):
                raise ValueError("embInit must be [%d, %d]" % (self.vocab, E))
        self.emb = torch.nn.Parameter(emb.float(), requires_grad=not g("wrdEmbFixed", False))
        lim = math.sqrt(6.0 / (E + h + 4 * h))        # glorot_uniform over [E+h, 4h]
        for d in ("fw", "bw"):
            k = (torch.rand((E + h, 4 * h), generator=generator, dtype=torch.float64) * 2 - 1) * lim
            self.register_parameter(d + "_kernel", torch.nn.Parameter(k.float()))
            self.register_parameter(d + "_bias", torch.nn.Parameter(torch.zeros(4 * h)))

    def tensors(self):
        return [getattr(self, f) for f in _lib.ENC_FIELDS]

    def to_reference_dict(self):
        return {REF_NAMES[f]: getattr(self, f).detach().clone() for f in _lib.ENC_FIELDS}

    def load_reference_dict(self, d):
        with torch.no_grad():
            for f in _lib.ENC_FIELDS:
                getattr(self, f).copy_(torch.as_tensor(d[REF_NAMES[f]]))

    def forward(self, questions, lengths, train=False, seed=None, b0=0, check_ids=True):
        """questions [B,S] int32 (0 = pad), lengths [B] int32 -> (questionCntxWords [B,S,2h], vecQuestions [B,2h])."""
        if not questions.is_cuda:
            raise RuntimeError("the question encoder has no CPU path")
        questions = questions.to(torch.int32).contiguous()
        lengths = lengths.to(device=questions.device, dtype=torch.int32).contiguous()
        if lengths.shape != (questions.shape[0],):
            raise ValueError("questionLengths must be [batchSize]")
        if check_ids:                                                                         # host sync; off in the bench loop
            if int(questions.max()) > self.vocab or int(questions.min()) < 0:
                raise IndexError("question word id outside [0, %d]" % self.vocab)
            # tf.reverse_sequence / dynamic_rnn reject lengths beyond the padded length (InvalidArgumentError)
            if int(lengths.max()) > questions.shape[1] or int(lengths.min()) < 0:
                raise ValueError("question length outside [0, %d]" % questions.shape[1])
        keep_in = self.keep_in if train else 1.0
        keep_q = self.keep_q if train else 1.0
        return _EncoderFunction.apply(self, keep_in, keep_q, fresh_seed(seed, train), int(b0), questions, lengths, *self.tensors())
