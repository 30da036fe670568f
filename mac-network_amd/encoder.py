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


def _check_fused(config):
    g = lambda n, dflt: getattr(config, n, dflt)
    if g("encType", "LSTM") != "LSTM" or not g("encBi", False) or g("encNumLayers", 1) != 1:
        raise UnsupportedOptions("encoder: only the bidirectional single-layer LSTM has a fused HIP path")
    if g("encVariationalDropout", False):
        raise UnsupportedOptions("encoder: encVariationalDropout has no HIP path")
    if g("encProj", False) or g("encDim", 512) != g("ctrlDim", 512):
        raise UnsupportedOptions("encoder: output projections (encProj / encDim != ctrlDim) have no fused HIP path")
    if g("ansEmbMod", "NON") == "SHARED":
        raise UnsupportedOptions("encoder: shared question/answer embeddings have no HIP path")
    if (int(g("encDim", 512)) // 2) % 128 or int(g("encDim", 512)) % 2:
        raise UnsupportedOptions("encoder: the fused kernels need 128 k units per direction (the generic path takes any multiple of 4)")


class QuestionEncoder(torch.nn.Module):
    """Mirrors `MACnet.embeddingsOp` + `MACnet.encoder` for the configuration every flag file uses
    (encType LSTM, --encBi, encNumLayers 1, encDim == ctrlDim so no projections) on the fused kernels.  Constructing it for
    another LSTM configuration -- no --encBi (the parser's default), --encProj, encDim != ctrlDim -- returns a
    GenericQuestionEncoder: the same interface on one HIP kernel per reference op."""

    def __new__(cls, config=None, *args, **kw):
        if cls is QuestionEncoder and config is not None:
            try:
                _check_fused(config)
            except UnsupportedOptions:
                return GenericQuestionEncoder(config, *args, **kw)
        return super().__new__(cls)

    def __init__(self, config, vocab, embInit=None, generator=None):
        super().__init__()
        g = lambda n, dflt: getattr(config, n, dflt)
        _check_fused(config)
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

    def embed(self, questions):
        """embeddingsOp's `questionWords` (model.py:207-219, 783): the embedded question words [B,S,wrdEmbDim] WITHOUT the encoder's
        input dropout -- what the cell attends over when --controlContextual is off (mac_cell.py:570)."""
        from .generic import _Embed
        B, S = questions.shape
        return _Embed.apply(questions, self.emb, self.E, 1.0, 0, 0).reshape(B, S, self.E)

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


class GenericQuestionEncoder(torch.nn.Module):
    """qEmbeddingsOp + encoder (model.py:207-219, 279-307 -> ops.RNNLayer -> biRNNLayer / fwRNNLayer, ops.py:798-950) for the
    LSTM configurations the fused kernels refuse -- a forward-only LSTM(encDim) (no --encBi: the parser's default), the output
    projections projCW / projQ (--encProj or encDim != ctrlDim, model.py:785-787, with --encProjQAct) -- as ONE HIP KERNEL PER
    REFERENCE OP: macx_embed_lookup(+_bwd), macx_linear / macx_wgrad for [x, h] @ kernel + bias, macx_op_act / macx_op_binary
    for the gates and the sequence-length masking, macx_op_dropout for the question vector.  torch owns memory, the
    per-step slices / stack / concat, the reverse-sequence gather and the autograd tape.  Variables under the reference's
    names (encoder/rnnLayer/rnn/basic_lstm_cell/kernel, ... / birnnLayer/bidirectional_rnn/{fw,bw}/..., linearLayerprojCW, ...).
    --encNumLayers > 1 raises what the reference raises (its layers collide on one variable scope); other cell types,
    variational dropout and shared answer embeddings are refused.  Hidden width per direction: any multiple of 4 (the products
    zero-pad their operands to the kernels' 128-column granule, generic.k_matmul)."""

    def __init__(self, config, vocab, embInit=None, generator=None, device=None):
        super().__init__()
        from .generic import GenericParams, _Ops
        dflt = dict(encType="LSTM", encBi=False, encNumLayers=1, encVariationalDropout=False, encProj=False, encProjQAct="NON",
                    encDim=512, ctrlDim=512, wrdEmbDim=300, encInputDropout=0.85, qDropout=0.92, wrdEmbFixed=False, ansEmbMod="NON")
        g = lambda n: getattr(config, n, dflt[n])
        if g("encType") != "LSTM":
            raise UnsupportedOptions("encoder: encType=%s has no HIP path" % g("encType"))
        if g("encVariationalDropout"):
            raise UnsupportedOptions("encoder: encVariationalDropout has no HIP path")
        if g("ansEmbMod") == "SHARED":
            raise UnsupportedOptions("encoder: shared question/answer embeddings have no HIP path")
        self.bi = bool(g("encBi"))
        self.scopes = ("birnnLayer", "bidirectional_rnn") if self.bi else ("rnnLayer", "rnn")
        if g("encNumLayers") > 1:
            # model.py:295-298 builds every layer from `questions` under one scope (ops.py:938-950 closes "rnnLayer<name>" before
            # the layer is built): the second layer asks tf.get_variable for the first one's kernel
            raise ValueError("Variable encoder/%s/%s/%sbasic_lstm_cell/kernel already exists, disallowed. Did you mean to set "
                             "reuse=True in VarScope?" % (self.scopes[0], self.scopes[1], "fw/" if self.bi else ""))
        self.config = config
        self.vocab, self.E = int(vocab), int(g("wrdEmbDim"))
        self.enc, self.ctrl = int(g("encDim")), int(g("ctrlDim"))
        self.hh = self.enc // 2 if self.bi else self.enc
        if self.hh % 4:
            raise UnsupportedOptions("encoder: the LSTM width per direction must be a multiple of 4 (got %d): 16-byte rows; widths "
                                     "off the product kernels' 128-column granule run zero-padded inside them" % self.hh)
        self.proj = g("encProj") or self.enc != self.ctrl
        self.proj_act = g("encProjQAct")
        self.keep_in, self.keep_q = float(g("encInputDropout")), float(g("qDropout"))
        self.fixed = bool(g("wrdEmbFixed"))
        self.params = GenericParams(device=device, generator=generator)
        self.ops = _Ops(config, self.params)
        self._declare(embInit)

    def _cells(self):
        return ("fw", "bw") if self.bi else ("",)

    def _declare(self, embInit):
        vs, ops = self.params, self.ops
        with vs.scope("qEmbeddings"):
            emb = vs.get("emb", (self.vocab, self.E), "normal")
        if embInit is not None:
            init = torch.as_tensor(embInit, dtype=torch.float32)
            if tuple(init.shape) != (self.vocab, self.E):
                raise ValueError("embInit must be [%d, %d]" % (self.vocab, self.E))
            with torch.no_grad():
                emb.copy_(init)
        emb.requires_grad_(not self.fixed)
        for _ in self._cell_vars():
            pass
        if self.proj:
            with vs.scope("encoder"):
                with vs.scope("linearLayerprojCW"):
                    ops.getWeight((self.enc, self.ctrl)), ops.getBias((self.ctrl,))
                with vs.scope("linearLayerprojQ"):
                    ops.getWeight((self.enc, self.ctrl)), ops.getBias((self.ctrl,))
                    if self.proj_act != "NON":
                        if self.proj_act == "RELU" and getattr(self.config, "relu", "STD") == "PRM":
                            with vs.scope("prelu", default=True):
                                vs.get("alpha", (self.ctrl,), 0.25)
                        with vs.scope("linearLayerprojQ_2"):
                            ops.getWeight((self.ctrl, self.ctrl)), ops.getBias((self.ctrl,))

    def _cell_vars(self):
        """(direction, kernel [E + hh, 4 hh], bias [4 hh]) per cell, created on first use under the reference's names."""
        vs = self.params
        with vs.scope("encoder"), vs.scope(self.scopes[0]), vs.scope(self.scopes[1]):
            for d in self._cells():
                if d:
                    with vs.scope(d), vs.scope("basic_lstm_cell"):
                        yield d, vs.get("kernel", (self.E + self.hh, 4 * self.hh), "xavier"), vs.get("bias", (4 * self.hh,), "zeros")
                else:
                    with vs.scope("basic_lstm_cell"):
                        yield d, vs.get("kernel", (self.E + self.hh, 4 * self.hh), "xavier"), vs.get("bias", (4 * self.hh,), "zeros")

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

    def embed(self, questions):
        """embeddingsOp's `questionWords` (model.py:207-219, 783): the embedded question words [B,S,wrdEmbDim] without the input
        dropout -- what the cell attends over when --controlContextual is off (mac_cell.py:570)."""
        from .generic import _Embed
        B, S = questions.shape
        with self.params.scope("qEmbeddings"):
            emb = self.params.get("emb", (self.vocab, self.E), "normal")
        return _Embed.apply(questions, emb, self.E, 1.0, 0, 0).reshape(B, S, self.E)

    def forward(self, questions, lengths, train=False, seed=None, b0=0, check_ids=True):
        """questions [B,S] int (0 = pad), lengths [B] -> (questionCntxWords [B,S,w], vecQuestions [B,w]), w = ctrlDim when projected."""
        from . import generic as G
        from .generic import B_CHANNEL, B_ROW, B_SAME, OP_ADD, OP_MUL, _Act, _Binary, _Dropout, _Embed, _Linear
        G._require_device(questions, "questions")                       # no CPU path
        dev = questions.device
        B, S = questions.shape
        lengths = lengths.to(device=dev)
        if lengths.shape != (B,):
            raise ValueError("questionLengths must be [batchSize]")
        if check_ids:
            if int(questions.max()) > self.vocab or int(questions.min()) < 0:
                raise IndexError("question word id outside [0, %d]" % self.vocab)
            if int(lengths.max()) > S or int(lengths.min()) < 0:
                raise ValueError("question length outside [0, %d]" % S)
        keep_in, keep_q = (self.keep_in, self.keep_q) if train else (1.0, 1.0)
        seed = fresh_seed(seed, train)
        E, hh = self.E, self.hh
        Ep = (E + 127) // 128 * 128
        vs = self.params
        with vs.scope("qEmbeddings"):
            emb = vs.get("emb", (self.vocab, E), "normal")
        x = _Embed.apply(questions, emb, Ep, keep_in, seed, int(b0) * S).reshape(B, S, Ep)          # zero columns [E, Ep)
        t_idx = torch.arange(S, device=dev).unsqueeze(0)                                            # host-side index bookkeeping
        L_ = lengths.long().unsqueeze(1)
        live = (t_idx < L_).to(torch.float32)                                                       # [B,S]
        rev = torch.where(t_idx < L_, L_ - 1 - t_idx, t_idx)                                        # array_ops.reverse_sequence
        ones = torch.ones(hh, dtype=torch.float32, device=dev)                                      # BasicLSTMCell forget_bias
        sig, tanh = _lib.ACT["SIGMOID"], _lib.ACT["TANH"]
        mul = lambda a, b: _Binary.apply(a.contiguous(), b.contiguous(), OP_MUL, B_SAME, 1.0)
        rowmul = lambda a, r: _Binary.apply(a.contiguous(), r.contiguous(), OP_MUL, B_ROW, 1.0)
        outs, finals = [], []
        for d, kernel, bias in self._cell_vars():
            # the kernel with zero rows under the padded embedding columns (memory op; the gradient of the padding is dropped)
            Kp = torch.cat([kernel[:E], kernel.new_zeros(Ep - E, 4 * hh), kernel[E:]], dim=0) if Ep != E else kernel
            xin = torch.gather(x, 1, rev.unsqueeze(-1).expand(B, S, Ep)) if d == "bw" else x
            h = torch.zeros(B, hh, dtype=torch.float32, device=dev)
            c = torch.zeros(B, hh, dtype=torch.float32, device=dev)
            seq = []
            for t in range(S):
                z = _Linear.apply(torch.cat([xin[:, t], h], dim=1), Kp, bias)                       # [x, h] @ kernel + bias
                i, j, f, o = [z[:, k * hh:(k + 1) * hh].contiguous() for k in range(4)]
                f1 = _Binary.apply(f, ones, OP_ADD, B_CHANNEL, 1.0)
                cn = _Binary.apply(mul(c, _Act.apply(f1, sig, None)), mul(_Act.apply(i, sig, None), _Act.apply(j, tanh, None)),
                                   OP_ADD, B_SAME, 1.0)
                hn = mul(_Act.apply(cn, tanh, None), _Act.apply(o, sig, None))
                lv, dead = live[:, t], 1.0 - live[:, t]                                             # dynamic_rnn: copy the state through
                h = _Binary.apply(rowmul(hn, lv), rowmul(h, dead), OP_ADD, B_SAME, 1.0)
                c = _Binary.apply(rowmul(cn, lv), rowmul(c, dead), OP_ADD, B_SAME, 1.0)
                seq.append(rowmul(hn, lv))                                                          # ... and emit zeros
            out = torch.stack(seq, dim=1)
            if d == "bw":
                out = torch.gather(out, 1, rev.unsqueeze(-1).expand(B, S, hh))
            outs.append(out)
            finals.append(h)
        words = torch.cat(outs, dim=-1) if len(outs) > 1 else outs[0]
        vecQ = torch.cat(finals, dim=-1) if len(finals) > 1 else finals[0]
        if keep_q < 1.0:
            vecQ = _Dropout.apply(vecQ.contiguous(), seed, 12, 0, keep_q, int(b0) * vecQ.shape[1])   # SITE_QUESTION
        if self.proj:
            with vs.scope("encoder"):
                words = self.ops.linear(words.contiguous(), self.enc, self.ctrl, name="projCW")
                vecQ = self.ops.linear(vecQ, self.enc, self.ctrl, act=self.proj_act, name="projQ")
        return words, vecQ
