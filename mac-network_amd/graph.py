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
the host is slow, never wrong.  The check exists because of a bug it would have caught: until the end of round 3 the
per-matrix maximum behind every packed weight's exponent was a 16-byte memset followed by integer atomicMax, and in about
one process in ten (one in three for the smallest shapes) the replayed graph ran the two out of order from its second
replay on -- maximum 0, exponent 0, weights split at the wrong scale, results finite and 1e-2 off, deterministically for
that process, while eager launches of the same kernels stayed bit-identical.  `absmax4` now writes per-workgroup partials
and a second kernel combines them (no memset, no atomics); tools/graph_replay_probe.py is the probe that found it.
"""
import warnings

import torch

from .cell import MACCell
from .options import get


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
