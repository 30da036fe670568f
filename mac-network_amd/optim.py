"""Training step of the reference (model.py:615-669): tf.clip_by_global_norm(8) -> tf.train.AdamOptimizer
-> tf.train.ExponentialMovingAverage(0.999), as ONE multi-tensor HIP pass over a flat fp32 buffer.

The parameters are re-pointed at slices of one flat tensor; gradients are gathered into (or, after
macx.dp.GradBucket.allreduce_, already live in) one flat tensor; Adam moments and the EMA shadow are flat
buffers too.  SURVEY.md 8f row 3."""
import ctypes as C

import torch

from . import _lib


class FlatAdamEMA:
    def __init__(self, params, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, clip_norm=8.0, ema_decay=0.999, grad_owner=None):
        """grad_owner: the MACCellParams whose grad_buffer() will be handed to step(flat_grad=...) by a cell-only training loop
        with no data-parallel bucket in between -- this optimizer then is the buffer's flat consumer: it registers (the backward
        pass writes into the persistent buffer only for a registered consumer) and releases the buffer at the end of every
        step.  With a dp.GradBucket / OverlappedBuckets / TowerBuckets the bucket is the consumer: leave this None."""
        self.grad_owner = grad_owner
        if grad_owner is not None:
            grad_owner.register_grad_buffer_user()
        self.params = [p for p in params]
        if not self.params or not self.params[0].is_cuda:
            raise RuntimeError("FlatAdamEMA needs parameters on the HIP device")
        self.sizes = [p.numel() for p in self.params]
        # 16-byte aligned segments: the layout of MACCellParams.grad_buffer() and dp.GradBucket.flat, so that either can be
        # handed to step(flat_grad=...) as it is (pad floats stay zero: gradient, moments and update)
        self.offsets, n = [], 0
        for k in self.sizes:
            self.offsets.append(n)
            n += (k + 3) & ~3
        dev = self.params[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, k, off in zip(self.params, self.sizes, self.offsets):          # parameters become views of the flat buffer
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p.data)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.ema = self.flat.clone() if ema_decay is not None and ema_decay >= 0 else None   # shadow starts at the initial value
        self.ws = torch.empty(1024, dtype=torch.float32, device=dev)
        self.norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.clip_norm = clip_norm if clip_norm else 0.0
        self.ema_decay = ema_decay if self.ema is not None else -1.0
        self.t = 0

    def step(self, flat_grad=None):
        """flat_grad: an already flat gradient in THIS layout -- the parameters in order, each segment padded to a multiple of
        4 floats: dp.GradBucket.flat / TowerBuckets.flat after the all-reduce, or MACCellParams.grad_buffer() for a cell-only
        optimizer built with grad_owner= (without a registered consumer the backward pass does not write into that buffer: a
        stale or zero gradient would be stepped on -- refused below; a dp bucket over the parameters is such a consumer).  A buffer of any other size is rejected.  Without
        flat_grad the .grad of every parameter is gathered."""
        if flat_grad is not None and self.grad_owner is None and getattr(flat_grad, "_macx_cell_grad_buffer", False):
            # the buffer is only ever written for a REGISTERED consumer: a dp bucket built over it (GradBucket(params=...),
            # OverlappedBuckets) is one, and hands its `flat` -- this very tensor -- over after the exchange
            owner = getattr(flat_grad, "_macx_cell_grad_owner", None)
            owner = owner() if owner is not None else None
            if owner is None or not getattr(owner, "_grad_flat_registered", False):
                raise ValueError("this is a MACCellParams.grad_buffer() nobody is registered for: build the optimizer with "
                                 "grad_owner=params (or a dp bucket over the parameters) so that the backward pass writes into it")
        if flat_grad is None:
            for p, k, off in zip(self.params, self.sizes, self.offsets):
                if p.grad is None:
                    self.grad[off:off + k].zero_()
                else:
                    self.grad[off:off + k].copy_(p.grad.reshape(-1))
            flat_grad = self.grad
        elif flat_grad.numel() != self.flat.numel() or flat_grad.dtype != torch.float32:
            raise ValueError("flat_grad holds %d floats, this optimizer's layout %d (segments padded to 4 floats, parameters in "
                             "the order given at construction)" % (flat_grad.numel(), self.flat.numel()))
        self.t += 1
        L = _lib.lib()
        dev = self.flat.device
        p_ = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        _lib.check(L.macx_adam_ema_step(self.flat.numel(), p_(self.flat), p_(flat_grad), p_(self.m), p_(self.v), p_(self.ema), self.lr,
                                        self.beta1, self.beta2, self.eps, self.t, self.clip_norm, self.ema_decay, p_(self.ws),
                                        p_(self.norm), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "macx_adam_ema_step")
        if self.grad_owner is not None:
            self.grad_owner.release_grad_buffer()          # the step's gradients are consumed: the next backward may claim the buffer
        return self.norm

    def ema_state(self):
        """{index: tensor} views of the EMA shadow, shaped like the parameters (what emaSaver restores, main.py:711-729)."""
        return [self.ema[off:off + k].view_as(p) for p, k, off in zip(self.params, self.sizes, self.offsets)]
