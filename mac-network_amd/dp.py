"""Batch data parallelism for the MAC cell: one process per GPU, `torch.distributed` over RCCL/xGMI.

The reference builds one tower per GPU and slices the batch with `initTowerBatch`
(model.py:139-149) but never exchanges gradients -- `averageAcrossTowers` keeps tower 0
(model.py:671-679, "TODO (add back support for multi-gpu..)").  Here every question is independent
in the forward pass, so the only exchange is ONE all-reduce per step of a flat fp32
gradient buffer (cell only: 2.1 M + p * 0.26 M parameters = 21 MB at p = 12).  Each rank computes
the MEAN loss over its shard (model.py:596); shard gradients are combined weighted by shard size so
that the result equals the full-batch gradient.
"""
import torch
import torch.distributed as dist


def tower_slice(batch_size, rank, world):
    """model.py:139-149: floor(B/R) questions per tower, the last tower takes the remainder."""
    per = batch_size // world
    start = rank * per
    end = (rank + 1) * per if rank < world - 1 else batch_size
    return start, end


class GradBucket:
    """Flat gradient buffer reused across steps; one all-reduce (sum) per step.

    `flat`: a persistent buffer the backward pass already writes into (MACCellParams.grad_buffer(): macx_cell_backward
    receives pointers into it, and autograd hands the views through as `.grad`).  When every gradient is found to be a
    view of that buffer at its expected offset the all-reduce runs on it in place -- no gather copy; anything else (a
    gradient produced elsewhere, accumulated, or absent) falls back to one gather kernel into the buffer."""

    def __init__(self, tensors, flat=None, params=None):
        """params: the MACCellParams whose grad_buffer() `flat` is -- registers this bucket as its flat consumer (the backward
        pass then writes into the buffer, once per step) and releases the buffer again after every all-reduce."""
        self.owner = params
        if params is not None:
            if flat is None:
                flat = params.grad_buffer()
            params.register_grad_buffer_user()
        self.tensors = list(tensors)
        self.sizes = [t.numel() for t in self.tensors]
        self.offsets, off = [], 0
        for n in self.sizes:
            self.offsets.append(off)
            off += (n + 3) & ~3                      # 16-byte aligned segments (what the kernels write through)
        if flat is None:
            flat = torch.zeros(off, dtype=torch.float32, device=self.tensors[0].device)
        if flat.numel() < off:
            raise ValueError("flat buffer holds %d floats, the gradients need %d" % (flat.numel(), off))
        self.flat = flat
        self.zero_copy_steps = 0

    def _in_place(self):
        base = self.flat.data_ptr()
        for t, o in zip(self.tensors, self.offsets):
            g = t.grad
            if g is None or not g.is_contiguous() or g.data_ptr() != base + 4 * o:
                return False
        return True

    def allreduce_(self, shard_size, global_size, group=None):
        """grads <- sum_r (shard_r / global) * grads_r  == gradient of the full-batch mean loss."""
        w = float(shard_size) / float(global_size)
        if self._in_place():
            self.zero_copy_steps += 1
        else:
            for t, o, n in zip(self.tensors, self.offsets, self.sizes):
                dst = self.flat[o:o + n]
                if t.grad is None:
                    dst.zero_()
                else:
                    dst.copy_(t.grad.reshape(-1))
        if w != 1.0:
            self.flat.mul_(w)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        for t, o, n in zip(self.tensors, self.offsets, self.sizes):            # gradients are views of the reduced buffer
            t.grad = self.flat[o:o + n].view_as(t)
        self._release()
        return self.flat

    def _release(self):
        """The step's gradients are final: the next backward pass may take the persistent buffer again (the caller drops the
        .grad views -- zero_grad / an optimizer step -- before it runs, as with any accumulate-free training loop)."""
        if self.owner is not None:
            self.owner.release_grad_buffer()


class OverlappedBuckets(GradBucket):
    """Two buckets over MACCellParams.grad_buffer(), the first in flight while the backward pass is still running.

    The cell's backward pass produces its gradients in two parts (macx_cell_backward_phase): everything except the read
    unit's [B,N,d]-contraction weights is final after phase 1 -- 80 % of the bytes at p = 12 -- and sits in ONE contiguous
    range at the front of the flat buffer (MACCellParams.fields puts the late tensors last).  The cell calls
    `after_backward_phase1` between the phases; this class answers by starting the all-reduce of that range on a side stream
    (RCCL over xGMI), so it overlaps phase 2, the last ~10 % of the backward pass.  `allreduce_()` after backward() reduces the
    rest and joins the side stream.  Falls back to the single all-reduce of GradBucket whenever the gradients are not the
    zero-copy views of the flat buffer.

        bucket = OverlappedBuckets(params)
        bucket.begin_step(shard_size, global_size)     # before backward(): the hook needs the shard weight
        loss.backward()
        bucket.allreduce_(shard_size, global_size)
    """

    def __init__(self, params, group=None):
        super().__init__(params.tensors(), flat=params.grad_buffer(), params=params)
        self.early = params.early_floats()
        self.group = group
        self.weight = 1.0
        self._early_work = None
        self._early_started = False
        self.side = torch.cuda.Stream(device=self.flat.device) if self.flat.is_cuda else None
        self.overlapped_steps = 0
        params.after_backward_phase1 = self._phase1

    def _active(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def begin_step(self, shard_size, global_size):
        self.weight = float(shard_size) / float(global_size)
        self._early_started = False
        self._early_work = None

    def _phase1(self, flat):
        if not self._active() or flat.data_ptr() != self.flat.data_ptr() or self.early == 0:
            return
        part = self.flat[: self.early]
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.side):
                if self.weight != 1.0:
                    part.mul_(self.weight)
                self._early_work = dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            if self.weight != 1.0:
                part.mul_(self.weight)
            self._early_work = dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._early_started = True

    def allreduce_(self, shard_size, global_size, group=None):
        w = float(shard_size) / float(global_size)
        if not (self._early_started and w == self.weight and self._in_place()):
            if self._early_started:
                raise RuntimeError("the early bucket is already in flight but the gradients are not views of the flat buffer")
            return super().allreduce_(shard_size, global_size, group if group is not None else self.group)
        late = self.flat[self.early:]
        if w != 1.0:
            late.mul_(w)
        if late.numel():
            dist.all_reduce(late, op=dist.ReduceOp.SUM, group=self.group)
        if self._early_work is not None:
            self._early_work.wait()
        if self.side is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.side)
        self._early_started = False
        self.overlapped_steps += 1
        self.zero_copy_steps += 1
        for t, o, n in zip(self.tensors, self.offsets, self.sizes):
            t.grad = self.flat[o:o + n].view_as(t)
        self._release()
        return self.flat


class TwoPhaseStep:
    """The host side of a data-parallel step whose backward pass is cut at the phase-1 seam (macx_cell_backward_phase): run part A
    (forward + backward phase 1), hand the finished front of the flat gradient buffer to the bucket -- the early all-reduce starts on the
    side stream -- run part B (backward phase 2) under it, then the late bucket and the join.  What the parts ARE is the subclass's
    business: graph.CapturedDPTrainStep replays two captured HIP graphs (2 replays + 2 collectives per step instead of ~125 launches);
    tests/test_dp_gloo.py drives this class with stand-in parts on CPU.  The bucket is driven exactly as the cell's autograd node
    drives it in the eager step (begin_step -> phase-1 hook -> allreduce_), so both give the same bits."""

    def __init__(self, params, bucket, shard, global_batch):
        self.params, self.bucket = params, bucket
        self.shard, self.global_batch = int(shard), int(global_batch)

    def run_part_a(self):          # forward + backward phase 1; returns {field: gradient view of the flat buffer}
        raise NotImplementedError

    def run_part_b(self):          # backward phase 2
        raise NotImplementedError

    def exchange_step(self):
        b = self.bucket
        if hasattr(b, "begin_step"):
            b.begin_step(self.shard, self.global_batch)
        grads = self.run_part_a()
        for f, g in grads.items():                       # (views of the flat buffer at the bucket's offsets: its zero-copy path)
            getattr(self.params, f).grad = g
        hook = getattr(self.params, "after_backward_phase1", None)
        if hook is not None:
            hook(b.flat)                                 # early bucket: scale + all-reduce (side stream on the GPU)
        self.run_part_b()
        return b.allreduce_(self.shard, self.global_batch)      # late bucket (or the one bucket), join; releases the flat buffer


class TowerBuckets:
    """Data parallelism for the WHOLE tower the reference builds per GPU (model.py:775-826: embeddings, question encoder, stem,
    MAC cell, output unit + classifier -- 57 MB of fp32 gradients at p = 12), two buckets over ONE flat buffer:

        [ output unit + classifier | cell, early fields | cell, late fields | stem | question encoder ]
          `--------- early bucket ---------'             `------------ late bucket -------------'

    Backward order decides the split: the classifier's gradients exist before the cell's backward pass starts and the cell's
    early fields are final after its phase 1 (macx_cell_backward_phase), so that range is all-reduced on a side stream from the
    cell's phase-1 hook while phase 2 -- and then the stem's and the encoder's backward passes, which need the cell's input
    gradients -- still run; the rest follows after backward().  The cell's gradient buffer IS its range of the flat buffer
    (MACCellParams.grad_buffer() is pointed at it), the other modules' gradients are gathered into theirs.  Clipping comes
    after the exchange, as in model.py:645-650: hand `flat` to optim.FlatAdamEMA(bucket.tensors()).step(flat_grad=bucket.flat).

        bucket = TowerBuckets(net); opt = FlatAdamEMA(bucket.tensors(), ...)
        bucket.begin_step(shard, global_batch); loss.backward(); bucket.allreduce_(shard, global_batch); opt.step(bucket.flat)
    """

    def __init__(self, net, group=None):
        self.net, self.group = net, group
        cell = net.cell
        if not hasattr(cell, "early_floats"):
            raise TypeError("TowerBuckets needs the fused cell's parameters (MACCellParams); a generic-path cell is exchanged with "
                            "GradBucket(net.tensors())")
        enc = list(net.enc.tensors()) if hasattr(net, "enc") else []
        groups = [list(net.out.tensors()), list(cell.tensors()), list(net.stem.tensors()), enc]
        self._tensors = [t for g in groups for t in g]
        self.sizes = [t.numel() for t in self._tensors]
        self.offsets, off = [], 0
        for n in self.sizes:
            self.offsets.append(off)
            off += (n + 3) & ~3
        dev = self._tensors[0].device
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        n_out = len(groups[0])
        self.cell_lo = self.offsets[n_out]
        cell_floats = sum((t.numel() + 3) & ~3 for t in groups[1])
        self.cell_hi = self.cell_lo + cell_floats
        # the cell's backward pass writes straight into its range
        object.__setattr__(cell, "_grad_flat", self.flat[self.cell_lo:self.cell_hi])
        cell.register_grad_buffer_user()
        cell.after_backward_phase1 = self._phase1
        self.early = self.cell_lo + cell.early_floats()
        self.n_out = n_out
        self.n_cell = len(groups[1])
        self.weight = 1.0
        self._early_work, self._early_started = None, False
        self.side = torch.cuda.Stream(device=dev) if self.flat.is_cuda else None
        self.overlapped_steps = 0
        self.allreduce_ms = None

    def tensors(self):
        """the parameters in the flat buffer's order (what optim.FlatAdamEMA must be built over to take `flat` as it is)"""
        return list(self._tensors)

    def _active(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def begin_step(self, shard_size, global_size):
        self.weight = float(shard_size) / float(global_size)
        self._early_started, self._early_work = False, None

    def _gather(self, lo, hi):
        for t, o, n in zip(self._tensors[lo:hi], self.offsets[lo:hi], self.sizes[lo:hi]):
            dst = self.flat[o:o + n]
            if t.grad is None:
                dst.zero_()
            elif t.grad.data_ptr() != dst.data_ptr():
                dst.copy_(t.grad.reshape(-1))

    def _phase1(self, flat):
        if flat.data_ptr() != self.flat.data_ptr() + 4 * self.cell_lo:
            return                                    # this backward pass got a buffer of its own: everything goes late
        if any(t.grad is None for t in self._tensors[:self.n_out]):
            return                                    # the classifier's gradients are not there yet (unusual graph): all late
        if not self._active():
            return                                    # single process: allreduce_ gathers and scales the WHOLE buffer (like OverlappedBuckets)
        self._gather(0, self.n_out)
        part = self.flat[:self.early]
        cur = torch.cuda.current_stream(self.flat.device) if self.side is not None else None
        if self.side is not None:
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                if self.weight != 1.0:
                    part.mul_(self.weight)
                self._early_work = dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            if self.weight != 1.0:
                part.mul_(self.weight)
            self._early_work = dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._early_started = True

    def allreduce_(self, shard_size, global_size):
        w = float(shard_size) / float(global_size)
        cell_ok = all(t.grad is not None and t.grad.data_ptr() == self.flat.data_ptr() + 4 * o
                      for t, o in zip(self._tensors[self.n_out:self.n_out + self.n_cell], self.offsets[self.n_out:self.n_out + self.n_cell]))
        early_done = self._early_started and w == self.weight and cell_ok
        if self._early_started and not early_done:
            raise RuntimeError("the early bucket is already in flight but the cell's gradients are not views of the flat buffer")
        lo = self.early if early_done else 0
        if not early_done:
            self._gather(0, self.n_out + self.n_cell)
        self._gather(self.n_out + self.n_cell, len(self._tensors))
        late = self.flat[lo:]
        if w != 1.0:
            late.mul_(w)
        if self._active():
            dist.all_reduce(late, op=dist.ReduceOp.SUM, group=self.group)
            if self._early_work is not None:
                self._early_work.wait()
        if self.side is not None and early_done:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.side)
        if early_done:
            self.overlapped_steps += 1
        self._early_started = False
        for t, o, n in zip(self._tensors, self.offsets, self.sizes):
            t.grad = self.flat[o:o + n].view_as(t)
        self.net.cell.release_grad_buffer()
        return self.flat
