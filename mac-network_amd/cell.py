"""MACCell -- the host-side mirror of the reference's recurrent cell (mac_cell.py:30-592) over
libmacx.so.

Same constructor arguments, `zero_state`, per-step `__call__`, `.iteration`, `.attentions`,
`.controls/.memories/.infos` as the reference class, so the loop of model.py:453-458 runs
unchanged:

    cell = MACCell(vecQuestions=..., questionWords=..., questionCntxWords=..., questionLengths=...,
                   knowledgeBase=..., memoryDropout=..., readDropout=..., writeDropout=...,
                   batchSize=B, train=True, config=config, params=params)
    state = cell.zero_state(B)
    for i in range(config.netLength):
        cell.iteration = i
        _, state = cell(none, state)

All arithmetic runs in hand-written HIP kernels behind the C ABI of include/macx.h; PyTorch only
owns the device memory and the stream.  There is no CPU or eager fallback: without a loadable
libmacx.so or without a HIP device every entry point raises.
"""
import collections
import ctypes as C

import torch

from . import _lib
from .options import freeze, fresh_seed, get
from .params import MACCellParams

MACCellTuple = collections.namedtuple("MACCellTuple", ("control", "memory"))   # mac_cell.py:8


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _f32c(t, name):
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must live on the HIP device: the MAC cell has no CPU path" % name)
    return t.contiguous()


def _mask_word(t, like):
    """macx_dropout.mask_word: one 32-bit word in device memory (an int32 / uint32 tensor with one element), or None."""
    if t is None:
        return None
    if not torch.is_tensor(t) or t.numel() != 1 or t.dtype not in (torch.int32, torch.uint32) or not t.is_cuda:
        raise TypeError("mask_word must be a 1-element int32 tensor on the HIP device (the kernels read it when they run)")
    if t.device != like.device:
        raise ValueError("mask_word lives on %s, the cell on %s" % (t.device, like.device))
    return t


class _Run:
    """One cell run: shapes, frozen options, buffers, and the ctypes structs that describe them."""

    def __init__(self, cell, keep_activations):
        self.cell = cell
        self.keep = int(bool(keep_activations))
        self.L = _lib.lib()
        self.opts = cell.opts
        B, S, d = cell.words.shape
        N = cell.knowledgeBase.shape[1]
        self.shapes = _lib.MacxShapes(B=B, S=S, N=N, d=d, p=cell.netLength, b0=cell.b0, d_logical=int(getattr(cell, "d_logical", 0)))
        _lib.check(self.L.macx_check(C.byref(self.opts), C.byref(self.shapes)), "macx_check")
        self.drop = _lib.MacxDropout(keep_memory=cell.dropouts["memory"], keep_read=cell.dropouts["read"],
                                     keep_write=cell.dropouts["write"], seed=cell.seed & 0xFFFFFFFF,
                                     mask_word=cell.mask_word.data_ptr() if cell.mask_word is not None else None)
        dev = cell.knowledgeBase.device
        self.saved_floats = self.L.macx_saved_floats(C.byref(self.opts), C.byref(self.shapes), self.keep)
        self.saved = torch.empty(self.saved_floats, dtype=torch.float32, device=dev)
        self.ws_fwd = torch.empty(max(4, self.L.macx_ws_floats(C.byref(self.opts), C.byref(self.shapes), 0)),
                                  dtype=torch.float32, device=dev)
        self.params = cell.params
        self.pstruct = _lib.MacxParams()
        self._ptensors = {}
        for f in _lib.PARAM_FIELDS:
            t = getattr(self.params, f, None) if f in self.params.fields else None
            if t is not None:
                t = _f32c(t.detach(), f)
                self._ptensors[f] = t
            setattr(self.pstruct, f, t.data_ptr() if t is not None else None)
        self.inputs = _lib.MacxInputs(vecQuestions=cell.vecQuestions.data_ptr(), words=cell.words.data_ptr(),
                                      questionLengths=cell.questionLengths.data_ptr(),
                                      knowledgeBase=cell.knowledgeBase.data_ptr())
        self.stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def _common(self):
        return (C.byref(self.opts), C.byref(self.shapes), C.byref(self.drop), C.byref(self.pstruct), C.byref(self.inputs),
                _ptr(self.saved), C.c_size_t(self.saved_floats), _ptr(self.ws_fwd), C.c_size_t(self.ws_fwd.numel()))

    def begin(self):
        _lib.check(self.L.macx_cell_begin(*self._common(), self.keep, self.stream), "macx_cell_begin")

    def step(self, i):
        _lib.check(self.L.macx_cell_step(*self._common(), self.keep, int(i), self.stream), "macx_cell_step")

    def forward(self):
        _lib.check(self.L.macx_cell_forward(*self._common(), self.keep, self.stream), "macx_cell_forward")

    def segment(self, name, shape):
        off, cnt = C.c_size_t(0), C.c_size_t(0)
        _lib.check(self.L.macx_saved_segment(C.byref(self.opts), C.byref(self.shapes), self.keep, _lib.SEG[name],
                                             C.byref(off), C.byref(cnt)), "macx_saved_segment")
        n = 1
        for x in shape:
            n *= x
        if n > cnt.value:
            raise ValueError("segment %s holds %d floats, view wants %d" % (name, cnt.value, n))
        return self.saved[off.value: off.value + n].view(*shape)

    def backward_begin(self, d_control, d_memory):
        """Buffers and argument block of this run's backward pass: (args of macx_cell_backward[_phase] without the trailing phase /
        stream, {field: gradient view}, (d vecQuestions, d words, d knowledgeBase), the flat gradient buffer, keep-alive list)."""
        if not self.keep:
            raise RuntimeError("this run did not keep its activations (train=False / no_grad)")
        dev = self.saved.device
        ws_floats = self.L.macx_ws_floats(C.byref(self.opts), C.byref(self.shapes), 1)
        ws = torch.empty(ws_floats, dtype=torch.float32, device=dev)
        gstruct = _lib.MacxParamGrads()
        grads = {}
        present = [f for f in self.params.fields if f in self._ptensors]
        sizes = [(self._ptensors[f].numel() + 3) & ~3 for f in present]          # 16-byte aligned views
        # the parameters' persistent flat gradient buffer (one fill instead of one per parameter; the views below become the
        # parameters' .grad, so a flat all-reduce / optimizer needs no gather) -- only for a registered flat consumer and only
        # once per step: a second cell run inside the same backward pass (micro-batches, two towers on one device), or a
        # caller that keeps torch.autograd.grad results across calls, gets a buffer of its own.
        flat = None
        if hasattr(self.params, "claim_grad_buffer") and present == list(self.params.fields):
            if all(getattr(self.params, f).grad is None for f in present):
                buf = self.params.claim_grad_buffer()          # None: no flat consumer, or already handed out this step
                if buf is not None and buf.numel() == sum(sizes):
                    flat = buf
        if flat is None:
            flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        off = 0
        for f, n in zip(present, sizes):
            t = self._ptensors[f]
            grads[f] = flat[off: off + t.numel()].view(t.shape)
            off += n
        for f in _lib.PARAM_FIELDS:
            setattr(gstruct, f, grads[f].data_ptr() if f in grads else None)
        cell = self.cell
        gi_vq = torch.empty_like(cell.vecQuestions)
        gi_words = torch.empty_like(cell.words)
        gi_kb = torch.empty_like(cell.knowledgeBase)
        gistruct = _lib.MacxInputGrads(vecQuestions=gi_vq.data_ptr(), words=gi_words.data_ptr(),
                                       knowledgeBase=gi_kb.data_ptr())
        dm = _f32c(d_memory, "d_memory") if d_memory is not None else None
        dc = _f32c(d_control, "d_control") if d_control is not None else None
        args = (C.byref(self.opts), C.byref(self.shapes), C.byref(self.drop), C.byref(self.pstruct), C.byref(self.inputs),
                _ptr(self.saved), C.c_size_t(self.saved_floats), _ptr(ws), C.c_size_t(ws_floats), _ptr(dm), _ptr(dc),
                C.byref(gstruct), C.byref(gistruct))
        return args, grads, (gi_vq, gi_words, gi_kb), flat, (ws, gstruct, gistruct, dm, dc)

    def backward_phase(self, args, phase):
        """macx_cell_backward_phase on torch's CURRENT stream (1: everything but the read unit's deferred contractions, 2: those)"""
        stream = C.c_void_p(torch.cuda.current_stream(self.saved.device).cuda_stream)
        _lib.check(self.L.macx_cell_backward_phase(*args, int(phase), stream), "macx_cell_backward_phase(%d)" % phase)

    def backward(self, d_control, d_memory):
        args, grads, (gi_vq, gi_words, gi_kb), flat, _keep = self.backward_begin(d_control, d_memory)
        stream = C.c_void_p(torch.cuda.current_stream(self.saved.device).cuda_stream)
        hook = getattr(self.params, "after_backward_phase1", None)
        if hook is not None and flat is getattr(self.params, "_grad_flat", None):
            # every gradient except the read unit's big contractions is final: let the data-parallel layer start on it
            self.backward_phase(args, 1)
            hook(flat)
            self.backward_phase(args, 2)
        else:
            _lib.check(self.L.macx_cell_backward(*args, stream), "macx_cell_backward")
        return grads, gi_vq, gi_words, gi_kb


class _CellFunction(torch.autograd.Function):
    """Autograd node for one whole p-step run.  mode 'run' executes the forward here; mode 'adopt'
    wraps a forward that the per-step API already executed."""

    @staticmethod
    def forward(ctx, run, mode, vecQ, words, kb, *params):
        if mode == "run":
            run.begin_and_forward()
        ctx.run = run
        # an output nobody differentiates (the final control, when only the memory feeds the classifier) arrives as None instead of a
        # freshly filled zeros tensor: one fill and one copy launch less per step -- macx_cell_backward takes NULL for it
        ctx.set_materialize_grads(False)
        p = run.shapes.p
        controls = run.segment("controls", (p + 1, run.shapes.B, run.shapes.d))
        memories = run.segment("memories", (p + 1, run.shapes.B, run.shapes.d))
        # views of the run's own `saved` buffer (nothing writes it after the forward pass): no copy launches
        return controls[p], memories[p]

    @staticmethod
    def backward(ctx, d_control, d_memory):
        run = ctx.run
        grads, gvq, gwords, gkb = run.backward(d_control, d_memory)
        pgrads = tuple(grads[f] for f in run.params.fields)
        return (None, None, gvq, gwords, gkb) + pgrads


def _begin_and_forward(self):
    self.forward()


_Run.begin_and_forward = _begin_and_forward


class MACCell:
    """Drop-in for mac_cell.MACCell.  Extra keyword arguments (all optional):
    config   object with the reference's flag names (default: the reference defaults)
    params   MACCellParams (default: freshly initialised like tf.get_variable would)
    seed     dropout stream seed;  b0  global index of this shard's first question (data parallel)
    mask_word  None, or a 1-element int32 tensor on the device: every dropout site XORs it into its key WHEN THE KERNELS RUN
             (macx.h, macx_dropout.mask_word).  The seed is baked into a captured HIP graph, this word is not: write a new
             value between replays and one capture draws fresh masks per step (graph.CapturedTrainStep).  None == word 0.
    gemm     kernel family of the knowledge-base GEMMs of THIS cell: "h2" | "split" | "native" (None: the process default)
    tune     {key: value} for macx_opts.tune, the per-call A/B hooks (_lib.TUNE; measurement only, never needed for results)
    """

    def __new__(cls, *args, config=None, gemm=None, **kw):
        """Option sets without fused kernels (the reference's default configuration among them) are built on the generic
        one-kernel-per-op path (generic.GenericMACCell, same interface); option values the reference itself rejects raise
        what the reference raises."""
        from types import SimpleNamespace
        from .options import UnsupportedOptions
        if cls is MACCell and config is not None and _needs_padding(config):
            # a width the kernels' 128-column granule does not divide (config.py:294-296 takes any): the same cell, zero-padded --
            # as a whole for the option sets with fused kernels, inside each product on the generic path
            try:
                freeze(config)
            except UnsupportedOptions:
                if gemm is not None:
                    raise
                from .generic import GenericMACCell
                return GenericMACCell(*args, config=config, **kw)
            return PaddedMACCell(*args, config=config, gemm=gemm, **kw)
        try:
            freeze(config if config is not None else SimpleNamespace())
        except UnsupportedOptions:
            if gemm is not None:
                raise UnsupportedOptions("gemm=%r selects the kernel family of the FUSED read unit; this option set runs on the "
                                         "generic one-kernel-per-op path, which has no such choice" % (gemm,))
            from .generic import GenericMACCell
            return GenericMACCell(*args, config=config, **kw)
        return super().__new__(cls)

    def __init__(self, vecQuestions, questionWords, questionCntxWords, questionLengths, knowledgeBase,
                 memoryDropout, readDropout, writeDropout, batchSize, train, reuse=None, *, config=None, params=None,
                 netLength=None, seed=None, b0=0, gemm=None, d_logical=0, mask_word=None, tune=None):
        from types import SimpleNamespace
        self.mask_word = _mask_word(mask_word, knowledgeBase)
        self.d_logical = int(d_logical)        # > 0: this is the zero-padded image of a d_logical-wide cell (PaddedMACCell)
        self.config = config if config is not None else SimpleNamespace()
        self.opts = freeze(self.config, gemm, tune)     # raises for rejected / unsupported option sets
        self.netLength = int(netLength if netLength is not None else get(self.config, "netLength"))
        self.vecQuestions = _f32c(vecQuestions, "vecQuestions")
        self.questionWords = questionWords
        self.questionCntxWords = questionCntxWords
        # word source select, mac_cell.py:570
        words = questionCntxWords if get(self.config, "controlContextual") else questionWords
        self._words_src = words
        self.words = _f32c(words, "question words")
        if not questionLengths.is_cuda:
            raise RuntimeError("questionLengths must live on the HIP device: the MAC cell has no CPU path")
        if questionLengths.dtype != torch.int32:
            questionLengths = questionLengths.to(torch.int32)
        if questionLengths.shape != (self.words.shape[0],):
            raise ValueError("questionLengths must be [batchSize]")
        self.questionLengths = questionLengths.contiguous()
        self._kb_src = knowledgeBase
        self._vq_src = vecQuestions
        self.knowledgeBase = _f32c(knowledgeBase, "knowledgeBase")
        train = bool(train)
        self.train = train
        # evaluation feeds keep = 1.0 to every dropout placeholder (model.py:118-125)
        self.dropouts = {"memory": float(memoryDropout) if train else 1.0,
                         "read": float(readDropout) if train else 1.0,
                         "write": float(writeDropout) if train else 1.0}
        self.batchSize = int(batchSize)
        self.reuse = reuse
        self.seed = fresh_seed(seed, bool(train))
        self.b0 = int(b0)
        self.params = params if params is not None else MACCellParams(self.config, self.netLength, device=self.knowledgeBase.device)
        self._none = None           # the dummy cell input / output (mac_cell.py:75): allocated when first asked for (run() never does)
        self.iteration = 0
        self._run = None

    # mac_cell.py:84-93
    @property
    def state_size(self):
        d = get(self.config, "memDim")
        return MACCellTuple(d, d)

    @property
    def output_size(self):
        return 1

    def _needs_grad(self):
        if not torch.is_grad_enabled():
            return False
        srcs = [self._vq_src, self._words_src, self._kb_src] + self.params.tensors()
        return any(t.requires_grad for t in srcs)

    def _views(self):
        run, s = self._run, self._run.shapes
        self._controls_all = run.segment("controls", (s.p + 1, s.B, s.d))
        self._memories_all = run.segment("memories", (s.p + 1, s.B, s.d))
        self._infos_all = run.segment("infos", (s.p, s.B, s.d))
        self._att_q = run.segment("att_question", (s.p, s.B, s.S))
        self._att_kb = run.segment("att_kb", (s.p, s.B, s.N))
        self._att_self = run.segment("att_self", (s.p, s.B, s.p)) if self.opts.write_self_att else None
        self._att_gate = run.segment("att_gate", (s.p, s.B, s.d)) if self.opts.write_gate else None

    # ---- zero_state (mac_cell.py:539-592)
    @property
    def none(self):
        """tf.zeros((batchSize, 1)): the cell's dummy input and output (mac_cell.py:75, :480)"""
        if self._none is None:
            self._none = torch.zeros((self.batchSize, 1), dtype=torch.float32, device=self.knowledgeBase.device)
        return self._none

    def zero_state(self, batchSize=None, dtype=torch.float32):
        self._run = _Run(self, keep_activations=self._needs_grad())
        self._run.begin()
        self._views()
        self._steps_done = 0
        self.attentions = {"kb": [], "question": [], "self": [], "gate": []}
        return MACCellTuple(self._controls_all[0], self._memories_all[0])

    def _publish_step(self, i):
        self.attentions["question"].append(self._att_q[i])
        self.attentions["kb"].append(self._att_kb[i])
        if self._att_self is not None:      # [B, i+1]: initial state + steps 0..i-1 (mac_cell.py:324-329)
            self.attentions["self"].append(self._att_self[i, :, : i + 1])
        if self._att_gate is not None:
            self.attentions["gate"].append(self._att_gate[i])

    # histories as the reference exposes them: [B, steps+1, d] (mac_cell.py:472-474)
    @property
    def controls(self):
        return self._controls_all[: self._steps_done + 1].transpose(0, 1)

    @property
    def memories(self):
        return self._memories_all[: self._steps_done + 1].transpose(0, 1)

    @property
    def infos(self):
        # the reference seeds `infos` with the initial MEMORY (mac_cell.py:551)
        return torch.cat([self._memories_all[:1], self._infos_all[: self._steps_done]], dim=0).transpose(0, 1)

    # ---- one step (mac_cell.py:420-480)
    def __call__(self, inputs, state, scope=None):
        if self._run is None:
            raise RuntimeError("call zero_state() first (model.py:447)")
        i = int(self.iteration)
        if i != self._steps_done:
            raise RuntimeError("steps must run in order: expected iteration %d, got %d" % (self._steps_done, i))
        self._run.step(i)
        self._steps_done = i + 1
        self._publish_step(i)
        control, memory = self._controls_all[i + 1], self._memories_all[i + 1]
        if i == self.netLength - 1 and self._run.keep:
            # the final state carries the autograd edge of the whole run
            control, memory = _CellFunction.apply(self._run, "adopt", self._vq_src, self._words_src, self._kb_src,
                                                  *self.params.tensors())
        return self.none, MACCellTuple(control, memory)

    # ---- the whole loop of model.py:453-458 in one ABI call
    def run(self):
        self._run = _Run(self, keep_activations=self._needs_grad())
        self.attentions = {"kb": [], "question": [], "self": [], "gate": []}
        if self._run.keep:
            control, memory = _CellFunction.apply(self._run, "run", self._vq_src, self._words_src, self._kb_src,
                                                  *self.params.tensors())
        else:
            self._run.forward()
            control = memory = None
        self._views()
        self._steps_done = self.netLength
        for i in range(self.netLength):
            self._publish_step(i)
        if control is None:
            control, memory = self._controls_all[self.netLength], self._memories_all[self.netLength]
        return MACCellTuple(control, memory)


# ---------------------------------------------------------------------------------------------------------------------------
# widths that are not multiples of 128 (config.py:294-296 takes any int): the same cell, zero-padded
# ---------------------------------------------------------------------------------------------------------------------------
def _needs_padding(config):
    d = int(get(config, "memDim"))
    return d % 128 != 0 and d % 8 == 0 and int(get(config, "ctrlDim")) == d and int(get(config, "attDim")) == d


def _pad_last(t, n):
    return torch.nn.functional.pad(t, (0, n - t.shape[-1]))


def _pad_blocks(t, d, dp):
    """[k * d, d] (k row blocks, one per concatenated input: ops.concat, ops.py:65-78) -> [k * dp, dp], each block padded on its own"""
    k = t.shape[0] // d
    return torch.nn.functional.pad(t.reshape(k, d, d), (0, dp - d, 0, dp - d)).reshape(k * dp, dp)


class _PaddedParams:
    """The zero-padded image of a MACCellParams: same fields, every width d -> dp.  The padded tensors are autograd results of
    the logical parameters (torch.nn.functional.pad), so gradients arrive at the logical tensors, sliced, on their own."""

    def __init__(self, params, d, dp):
        self.fields = list(params.fields)
        self.p = params.p
        for f in self.fields:
            t = getattr(params, f)
            if t.dim() == 1:
                v = t if t.shape[0] == 1 else _pad_last(t, dp)                       # scalar logit biases stay
            elif f in ("memKbProj_W", "newMemory_W", "contControl_W"):
                v = _pad_blocks(t, d, dp)
            elif t.dim() == 2 and t.shape[0] != d:                                   # [nU, d] stacked biases
                v = _pad_last(t, dp)
            elif t.dim() == 2:
                v = torch.nn.functional.pad(t, (0, dp - d, 0, dp - d))
            else:                                                                    # [nU, d, d]
                v = torch.nn.functional.pad(t, (0, dp - d, 0, dp - d))
            setattr(self, f, v)

    def tensors(self):
        return [getattr(self, f) for f in self.fields]


class PaddedMACCell:
    """MACCell for memDim == ctrlDim == attDim == d with d % 128 != 0 (d % 8 == 0): the cell runs dp = ceil(d / 128) * 128 wide
    on zero-padded weights, biases and inputs (macx_shapes.d_logical = d).  Padded columns stay exact zeros through every
    linear layer (zero weights, zero bias), every unit activation config.py offers (NON / RELU in its --relu variants / TANH:
    act(0) = 0; the gate's sigmoid(0) = 0.5 mixes two zeros), attention logit (zero terms of a dot product) and gradient, and
    the kernels take the dropout element index at the LOGICAL width, so states, attentions and gradients are those of the
    unpadded cell with the unpadded cell's masks.  A unit activation of SIGMOID (ops.activations has it, config.py's choices do
    not) would put 0.5 into the padded columns of H1 / the memory -- harmless after slicing, but no longer 'exact zeros' and a
    different H2 row exponent than the unpadded cell's: refused here (UnsupportedOptions) rather than run untested.  Same
    interface as MACCell; states and histories come back d wide."""

    def __init__(self, vecQuestions, questionWords, questionCntxWords, questionLengths, knowledgeBase,
                 memoryDropout, readDropout, writeDropout, batchSize, train, reuse=None, *, config=None, params=None,
                 netLength=None, seed=None, b0=0, gemm=None, mask_word=None, tune=None):
        import copy
        from .options import UnsupportedOptions
        # the dropout index at the logical width (macx_shapes.d_logical) is implemented by the H2 kernel family: the padded cell runs
        # on it whatever the process default is; asking for another family is refused, not silently overridden
        if gemm not in (None, "h2"):
            raise UnsupportedOptions("gemm=%r: a cell whose width is not a multiple of 128 runs zero-padded on the H2 kernel family only" % (gemm,))
        gemm = "h2"
        for o in ("controlInputAct", "controlContAct", "readMemAct", "readCtrlAct", "writeMemAct"):
            if str(get(config, o)).upper() == "SIGMOID":
                raise UnsupportedOptions("%s=SIGMOID on a zero-padded width: sigmoid(0) = 0.5 would fill the padded columns" % o)
        self.config = config
        d = self.d = int(get(config, "memDim"))
        dp = self.dp = (d + 127) // 128 * 128
        self.netLength = int(netLength if netLength is not None else get(config, "netLength"))
        self.params = params if params is not None else MACCellParams(config, self.netLength, device=knowledgeBase.device)
        wide = copy.copy(config)
        wide.memDim = wide.ctrlDim = wide.attDim = dp
        pad = lambda t: _pad_last(t, dp)
        words_p = pad(questionWords)
        cntx_p = words_p if questionCntxWords is questionWords else pad(questionCntxWords)
        self.inner = MACCell(pad(vecQuestions), words_p, cntx_p, questionLengths, pad(knowledgeBase), memoryDropout, readDropout,
                             writeDropout, batchSize, train, reuse, config=wide, params=_PaddedParams(self.params, d, dp),
                             netLength=self.netLength, seed=seed, b0=b0, gemm=gemm, d_logical=d, mask_word=mask_word, tune=tune)
        self.batchSize, self.train, self.seed, self.b0 = self.inner.batchSize, self.inner.train, self.inner.seed, self.inner.b0

    none = property(lambda self: self.inner.none)
    iteration = property(lambda self: self.inner.iteration, lambda self, v: setattr(self.inner, "iteration", v))
    attentions = property(lambda self: self.inner.attentions)

    @property
    def state_size(self):
        return MACCellTuple(self.d, self.d)

    @property
    def output_size(self):
        return 1

    def _cut(self, state):
        return MACCellTuple(state.control[..., :self.d], state.memory[..., :self.d])

    def zero_state(self, batchSize=None, dtype=torch.float32):
        return self._cut(self.inner.zero_state(batchSize, dtype))

    def __call__(self, inputs, state, scope=None):
        out, st = self.inner(inputs, state, scope)
        return out, self._cut(st)

    def run(self):
        return self._cut(self.inner.run())

    controls = property(lambda self: self.inner.controls[..., :self.d])
    memories = property(lambda self: self.inner.memories[..., :self.d])
    infos = property(lambda self: self.inner.infos[..., :self.d])
