"""Inference forward of the MAC cell replayed from ONE captured HIP graph.

Forward only at p = 4 (BASELINE.json configs[1]) is about 70 kernel launches of 5 - 80 us each: issued one ctypes call at a
time the host, not the GPU, sets the pace (3.2 ms per batch of 64 against 0.7 ms of kernel time).  The loop of
model.py:453-458 has no data-dependent control flow, every buffer is caller-owned and every launch goes to the stream the
caller passes, so the whole run is capturable: `CapturedForward` captures `MACCell(...).run()` once on static input / output
tensors and replays it per batch.

    fwd = macx.CapturedForward(cfg, params, B=64, S=50, N=196)
    memory = fwd(vecQuestions, questionCntxWords, questionLengths, knowledgeBase)       # [B, d], valid until the next call
    att_kb = fwd.attentions["kb"]                                                       # p x [B, N] views, refreshed by replay

Evaluation only (no dropout, nothing kept for a backward pass); parameters are read at replay time, so an optimizer step or
a checkpoint load between calls is seen.  Changing a parameter's storage (`.to()`, `.data = `) needs a new capture.

A capture checks itself before it is used (`verify=True`): three replays on random inputs must reproduce the eager run bit
for bit; a process whose replays do not falls back to eager launches (`captured` is False, a warning says so): slower where
the host is slow, never wrong.  The check exists because of a bug it would have caught.  Until the end of round 3 the
per-matrix maximum behind every packed weight's exponent was a 16-byte hipMemsetAsync followed by integer atomicMax, and in
about one process in ten (one in three for the smallest shapes) the replayed graph ran the two out of order from its second
replay on -- maximum 0, exponent 0, weights split at the wrong scale, results finite and 1e-2 off, deterministically for
that process, while eager launches of the same kernels stayed bit-identical.  Round 3 removed that one pair; round 4 found
the general rule behind it (tools/graph_train_probe_verify.py, profiles/r04_graph_train_verify_8_processes.txt): under
ROCm 7.2 a captured hipMemsetAsync / hipMemcpyAsync becomes a memset / memcpy NODE, and replay does not keep such nodes in
stream order with the kernel nodes around them.  The library therefore issues no memset or memcpy at all any more -- every
fill and copy on the step's path is a kernel (`dev_zero` / `dev_copy` / `dev_copy2d` in macx_api.hip) -- and reductions that
used to start from a memset write per-workgroup partials that a later kernel combines.
(What torch adds around the library inside a capture is not under its control: the two `.clone()`s of the final state are
device-to-device copy nodes.  They deliver `memory`, which the self-check compares on every verification replay, and they have
never been seen out of order -- but they are the reason the check stays on by default.)
"""
import warnings

import torch

from .cell import MACCell
from .dp import TwoPhaseStep
from .options import get


def mix32(x):
    """the 32-bit finaliser the dropout stream hashes with (macx_common.hip.h, hash_mix): iteration number -> mask word"""
    h = (x + 0x7F4A7C15) & 0xFFFFFFFF
    h = (h * 0x9E3779B1) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 0x85EBCA77) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE3D) & 0xFFFFFFFF
    h ^= h >> 16
    return h


class CapturedForward:
    def __init__(self, config, params, B, S, N, device=None, netLength=None, warmup=2, verify=True):
        dev = torch.device(device) if device is not None else params.tensors()[0].device
        if dev.type != "cuda":
            raise RuntimeError("CapturedForward needs the HIP device: the MAC cell has no CPU path")
        d = int(get(config, "memDim"))
        self.config, self.params = config, params
        self.netLength = int(netLength if netLength is not None else get(config, "netLength"))
        self.vecQuestions = torch.zeros(B, d, device=dev)
        self.words = torch.zeros(B, S, d, device=dev)
        self.lengths = torch.full((B,), S, dtype=torch.int32, device=dev)
        self.knowledgeBase = torch.zeros(B, N, d, device=dev)
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):          # code objects, LDS attributes and the allocator settle outside the capture
                self._cell().run()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.cell = self._cell()
            state = self.cell.run()
            self.memory, self.control = state.memory, state.control
        self.attentions = self.cell.attentions
        self.captured = True
        if verify and not self._replays_match_eager():
            self.captured = False
            warnings.warn("CapturedForward: replays of the captured run do not reproduce the eager run in this process; "
                          "falling back to eager launches", RuntimeWarning)

    def _replays_match_eager(self, replays=3):
        g = torch.Generator().manual_seed(20240519)
        dev = self.knowledgeBase.device
        self.vecQuestions.copy_(torch.randn(self.vecQuestions.shape, generator=g).to(dev))
        self.words.copy_(torch.randn(self.words.shape, generator=g).to(dev))
        self.knowledgeBase.copy_(torch.randn(self.knowledgeBase.shape, generator=g).to(dev))
        with torch.no_grad():
            want = self._cell().run().memory.clone()
        ok = True
        for _ in range(replays):
            self.graph.replay()
            ok = ok and bool(torch.equal(self.memory, want))
        torch.cuda.synchronize(dev)
        return ok

    def _cell(self):
        return MACCell(vecQuestions=self.vecQuestions, questionWords=self.words, questionCntxWords=self.words,
                       questionLengths=self.lengths, knowledgeBase=self.knowledgeBase, memoryDropout=1.0, readDropout=1.0,
                       writeDropout=1.0, batchSize=self.vecQuestions.shape[0], train=False, config=self.config,
                       params=self.params, netLength=self.netLength)

    def load(self, vecQuestions, words, lengths, knowledgeBase):
        """Copy a batch into the captured run's input tensors (or write into fwd.knowledgeBase etc. directly and skip this)."""
        self.vecQuestions.copy_(vecQuestions)
        self.words.copy_(words)
        self.lengths.copy_(lengths)
        self.knowledgeBase.copy_(knowledgeBase)

    def replay(self):
        if self.captured:
            self.graph.replay()
        else:                                    # (see the module docstring)
            with torch.no_grad():
                self.cell = self._cell()
                state = self.cell.run()
            self.memory, self.control = state.memory, state.control
            self.attentions = self.cell.attentions
        return self.memory

    def __call__(self, vecQuestions, words, lengths, knowledgeBase):
        self.load(vecQuestions, words, lengths, knowledgeBase)
        return self.replay()


class CapturedTrainStep:
    """Forward + backward of the cell (train-mode dropout, every gradient) replayed from ONE captured HIP graph.

        step = macx.CapturedTrainStep(cfg, params, B=64, S=50, N=196, seed=1234)
        step.load(vecQuestions, words, lengths, knowledgeBase, d_memory)     # or write into step.knowledgeBase etc.
        step.replay()                        # params' .grad, step.knowledgeBase.grad, step.words.grad, step.vecQuestions.grad
        memory = step.memory                 # [B, d] final memory of the run

    About 140 launches per step (p = 12) plus what autograd adds around them; a host that cannot issue them faster than the
    GPU retires them (gpurun boxes differ by 7x in host speed) sets the pace of the eager step, a replay does not depend on it.
    The library issues no memset / memcpy node (module docstring; the minimum-exponent arrays behind the deferred contractions
    were the last memset-then-atomicMin pair), which is what makes the backward pass capturable.

    Fresh masks per replay -- the dropout masks are a function of (seed, site, step, element) and the seed travels BY VALUE in
    the kernel arguments, so a capture bakes it.  The run therefore also carries one 32-bit word in DEVICE memory
    (`macx_dropout.mask_word`, `step.mask_word`) that every dropout site XORs into its key when the kernel RUNS:

        for it in range(steps):
            step.load(...)
            step.replay(iteration=it)        # word = mix32(it): the masks of (seed, word); same word => same masks

    `replay()` without an argument keeps the word it has (0 after construction: the masks of the plain seed), which is what
    measurement wants.  The eager step behind `_eager()` reads the same word, so verification compares like with like.

    `verify=True` replays three times against the eager step on random inputs and falls back to eager launches when a replay
    differs in any gradient (`captured` False, a warning says so)."""

    def __init__(self, config, params, B, S, N, seed=0, device=None, netLength=None, b0=0, warmup=2, verify=True):
        dev = torch.device(device) if device is not None else params.tensors()[0].device
        if dev.type != "cuda":
            raise RuntimeError("CapturedTrainStep needs the HIP device: the MAC cell has no CPU path")
        d = int(get(config, "memDim"))
        self.config, self.params, self.seed, self.b0 = config, params, int(seed), int(b0)
        self.netLength = int(netLength if netLength is not None else get(config, "netLength"))
        self.vecQuestions = torch.zeros(B, d, device=dev, requires_grad=True)
        self.words = torch.zeros(B, S, d, device=dev, requires_grad=True)
        self.lengths = torch.full((B,), S, dtype=torch.int32, device=dev)
        self.knowledgeBase = torch.zeros(B, N, d, device=dev, requires_grad=True)
        self.d_memory = torch.zeros(B, d, device=dev)
        self.mask_word = torch.zeros(1, dtype=torch.int32, device=dev)        # macx_dropout.mask_word of every run of this step
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._clear_grads()
        with torch.cuda.graph(self.graph):
            self.memory = self._eager()
        self.captured = True
        if verify and not self._replays_match_eager():
            self.captured = False
            warnings.warn("CapturedTrainStep: replays of the captured step do not reproduce the eager step in this process; "
                          "falling back to eager launches", RuntimeWarning)

    def _leaves(self):
        return [self.vecQuestions, self.words, self.knowledgeBase] + list(self.params.tensors())

    def _clear_grads(self):
        for t in self._leaves():
            t.grad = None

    def _eager(self):
        self._clear_grads()
        cell = MACCell(vecQuestions=self.vecQuestions, questionWords=self.words, questionCntxWords=self.words,
                       questionLengths=self.lengths, knowledgeBase=self.knowledgeBase,
                       memoryDropout=float(get(self.config, "memoryDropout")), readDropout=float(get(self.config, "readDropout")),
                       writeDropout=float(get(self.config, "writeDropout")), batchSize=self.vecQuestions.shape[0], train=True,
                       config=self.config, params=self.params, netLength=self.netLength, seed=self.seed, b0=self.b0,
                       mask_word=self.mask_word)
        state = cell.run()
        torch.autograd.backward([state.memory], [self.d_memory])
        return state.memory.detach()

    def set_mask_word(self, word):
        """the raw 32-bit word the next replays XOR into every dropout key (0: the masks of the plain seed)"""
        word &= 0xFFFFFFFF
        self.mask_word.fill_(word - (1 << 32) if word >= (1 << 31) else word)

    def _replays_match_eager(self, replays=3):
        g = torch.Generator().manual_seed(20240520)
        dev = self.knowledgeBase.device
        self.set_mask_word(0x5bd1e995)          # a non-trivial word: the check covers the device-read path as well
        with torch.no_grad():
            for t in (self.vecQuestions, self.words, self.knowledgeBase, self.d_memory):
                t.copy_(torch.randn(t.shape, generator=g).to(dev))
        captured_grads = [t.grad for t in self._leaves()]        # the tensors the graph writes
        mem = self._eager().clone()
        want = [t.grad.clone() for t in self._leaves()]
        for t, gcap in zip(self._leaves(), captured_grads):
            t.grad = gcap
        ok = True
        self.verify_report = []              # (replay, index into [memory] + leaves) of every tensor that differed
        for r in range(replays):
            self.graph.replay()
            if not torch.equal(self.memory, mem):
                self.verify_report.append((r, 0))
            for i, (t, w) in enumerate(zip(self._leaves(), want)):
                if not torch.equal(t.grad, w):
                    self.verify_report.append((r, i + 1))
        torch.cuda.synchronize(dev)
        self.set_mask_word(0)
        ok = not self.verify_report
        return ok

    def load(self, vecQuestions, words, lengths, knowledgeBase, d_memory):
        with torch.no_grad():
            self.vecQuestions.copy_(vecQuestions)
            self.words.copy_(words)
            self.lengths.copy_(lengths)
            self.knowledgeBase.copy_(knowledgeBase)
            self.d_memory.copy_(d_memory)

    def replay(self, iteration=None):
        """iteration: None keeps the current mask word; an int draws the masks of word mix32(iteration)"""
        if iteration is not None:
            self.set_mask_word(mix32(int(iteration)))
        if self.captured:
            self.graph.replay()
        else:
            self.memory = self._eager()
        return self.memory


class CapturedDPTrainStep(TwoPhaseStep):
    """One DATA-PARALLEL training step of the cell as TWO graph replays with the gradient exchange between and behind them.

    The eager data-parallel step is ~125 host-issued launches per rank; at 8 questions per GPU that is more host work than GPU work
    and inherits the host's launch rate (DESIGN: boxes differ 7x).  The exchange itself cannot sit inside a graph (RCCL calls are
    issued by torch.distributed), but the backward pass has exactly one seam where it is needed: after phase 1
    (macx_cell_backward_phase) every gradient except the read unit's deferred contractions is final.  So:

        graph A = forward + backward phase 1            replay
        early bucket: scale + all-reduce                side stream, in flight under graph B     (bucket._phase1)
        graph B = backward phase 2                      replay
        late bucket: scale + all-reduce, join           (bucket.allreduce_)

    -- 2 replays + 2 collectives per step instead of ~125 launches.  `bucket` is a macx.dp.OverlappedBuckets (two buckets; RCCL) or a
    macx.dp.GradBucket built over params (one all-reduce behind graph B); the step drives it exactly as the cell's autograd node
    drives it in the eager step, so both produce the same bits (tests/test_gpu_dp.py, two ranks sharing the GPU).  No autograd is
    involved: the two phases are the C-ABI calls themselves, captured on static buffers; parameters' .grad are views of the flat
    gradient buffer.  Fresh dropout masks per step through the mask word, as CapturedTrainStep.

        bucket = macx.dp.OverlappedBuckets(params)
        step = macx.CapturedDPTrainStep(cfg, params, bucket, B=shard, S=50, N=196, global_batch=64, seed=1234, b0=lo)
        step.load(vecQ[lo:hi], words[lo:hi], lengths[lo:hi], kb[lo:hi], d_memory[lo:hi])
        step.step(iteration=it)              # params' .grad = the all-reduced full-batch gradient; step.memory: [shard, d]
    """

    def __init__(self, config, params, bucket, B, S, N, global_batch, seed=0, b0=0, device=None, netLength=None, warmup=2, capture=True):
        dev = torch.device(device) if device is not None else params.tensors()[0].device
        if dev.type != "cuda":
            raise RuntimeError("CapturedDPTrainStep needs the HIP device: the MAC cell has no CPU path")
        if bucket.flat.data_ptr() != params.grad_buffer().data_ptr():
            raise ValueError("the bucket must be built over params' own flat gradient buffer (OverlappedBuckets(params) / "
                             "GradBucket(params.tensors(), params=params))")
        d = int(get(config, "memDim"))
        super().__init__(params, bucket, B, global_batch)
        self.config = config
        self.seed, self.b0 = int(seed), int(b0)
        self.netLength = int(netLength if netLength is not None else get(config, "netLength"))
        self.vecQuestions = torch.zeros(B, d, device=dev)
        self.words = torch.zeros(B, S, d, device=dev)
        self.lengths = torch.full((B,), S, dtype=torch.int32, device=dev)
        self.knowledgeBase = torch.zeros(B, N, d, device=dev)
        self.d_memory = torch.zeros(B, d, device=dev)
        self.mask_word = torch.zeros(1, dtype=torch.int32, device=dev)
        self.graph_a, self.graph_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        self.captured = False
        for t in params.tensors():
            t.grad = None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                      # (warm-up outside capture: code objects, LDS attributes, allocator)
            for _ in range(max(1, warmup)):
                self._phase_a()
                self._phase_b()
                params.release_grad_buffer()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if capture:
            with torch.cuda.graph(self.graph_a):
                self._phase_a()
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool()):
                self._phase_b()
            self.captured = True
        params.release_grad_buffer()

    def _phase_a(self):
        """forward + backward phase 1 on the current stream; leaves self._run / self._args for phase 2"""
        from .cell import _Run
        with torch.no_grad():
            cell = MACCell(vecQuestions=self.vecQuestions, questionWords=self.words, questionCntxWords=self.words,
                           questionLengths=self.lengths, knowledgeBase=self.knowledgeBase,
                           memoryDropout=float(get(self.config, "memoryDropout")), readDropout=float(get(self.config, "readDropout")),
                           writeDropout=float(get(self.config, "writeDropout")), batchSize=self.shard, train=True,
                           config=self.config, params=self.params, netLength=self.netLength, seed=self.seed, b0=self.b0,
                           mask_word=self.mask_word)
            run = _Run(cell, True)
            run.forward()
            args, grads, gi, flat, keep = run.backward_begin(None, self.d_memory)
            if flat.data_ptr() != self.bucket.flat.data_ptr():
                raise RuntimeError("the backward pass did not receive the parameters' flat gradient buffer (is a gradient still attached?)")
            run.backward_phase(args, 1)
        p = run.shapes.p
        self.memory = run.segment("memories", (p + 1, run.shapes.B, run.shapes.d))[p]
        self.d_vecQuestions, self.d_words, self.d_knowledgeBase = gi
        self._grads = grads
        self._run, self._args, self._keep = run, args, (keep, cell)

    def _phase_b(self):
        with torch.no_grad():
            self._run.backward_phase(self._args, 2)

    def set_mask_word(self, word):
        word &= 0xFFFFFFFF
        self.mask_word.fill_(word - (1 << 32) if word >= (1 << 31) else word)

    def load(self, vecQuestions, words, lengths, knowledgeBase, d_memory):
        with torch.no_grad():
            self.vecQuestions.copy_(vecQuestions)
            self.words.copy_(words)
            self.lengths.copy_(lengths)
            self.knowledgeBase.copy_(knowledgeBase)
            self.d_memory.copy_(d_memory)

    def run_part_a(self):
        if self.captured:
            self.graph_a.replay()
        else:
            for t in self.params.tensors():
                t.grad = None
            self._phase_a()
        return self._grads

    def run_part_b(self):
        if self.captured:
            self.graph_b.replay()
        else:
            self._phase_b()

    def step(self, iteration=None):
        """one data-parallel step; afterwards every parameter's .grad is its view of the all-reduced flat buffer"""
        if iteration is not None:
            self.set_mask_word(mix32(int(iteration)))
        self.exchange_step()
        return self.memory
