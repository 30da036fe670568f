"""CPU, world_size 2 over gloo: the data-parallel path (tower split rule of model.py:139-149 + one
flat gradient all-reduce) reproduces the full-batch gradient.  Shard gradients come from the oracle
(the HIP cell cannot run here); what is under test is macx.dp and the global-question-index
convention of the dropout stream (shard b0)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import macx
    from oracle import mac_oracle as mo
    Bg, S, N, d, p = 5, 6, 7, 8, 2            # odd global batch: the last tower takes the remainder
    cfg = mo.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
    vq, words, lengths, kb = mo.synthetic_inputs(Bg, S, N, d, seed=3, dtype=torch.float64)
    answers_w = torch.randn(Bg, d, generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    keeps = (cfg.memoryDropout, cfg.readDropout, cfg.writeDropout)

    def grads(lo, hi):
        vs = mo.VarStore(generator=torch.Generator().manual_seed(7), dtype=torch.float64, requires_grad=True)
        c, m, _ = mo.mac_network(cfg, vs, vq[lo:hi], words[lo:hi], words[lo:hi], lengths[lo:hi], kb[lo:hi], train=True,
                                 mask_fn=mo.hash_mask_fn(42, keeps, b0=lo), keeps=keeps)
        loss = (m * answers_w[lo:hi]).sum(dim=1).mean()         # mean over the shard (model.py:596)
        loss.backward()
        return vs

    lo, hi = macx.dp.tower_slice(Bg, rank, world)
    vs = grads(lo, hi)
    names = sorted(vs.params)
    tensors = [vs.params[k].float() for k in names]
    for t, k in zip(tensors, names):
        t.grad = vs.params[k].grad.float()
    bucket = macx.dp.GradBucket(tensors)
    bucket.allreduce_(hi - lo, Bg)
    if rank == 0:
        full = grads(0, Bg)
        worst = 0.0
        for t, k in zip(tensors, names):
            ref = full.params[k].grad
            worst = max(worst, float((t.grad.double() - ref).abs().max() / max(float(ref.abs().max()), 1e-6)))
        ret["worst"] = worst
        ret["slices"] = [macx.dp.tower_slice(Bg, r, world) for r in range(world)]
    dist.destroy_process_group()


def test_two_rank_gradient_equals_full_batch():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["slices"] == [(0, 2), (2, 5)]
    assert ret["worst"] < 1e-5, ret["worst"]


def test_tower_slice_rule():
    import macx
    assert [macx.dp.tower_slice(1024, r, 8) for r in (0, 7)] == [(0, 128), (896, 1024)]
    assert [macx.dp.tower_slice(10, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 10)]


def _overlap_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import macx
    from oracle import mac_oracle as mo
    cfg = mo.flag_file_config("args", netLength=3, memDim=128, ctrlDim=128, attDim=128)
    params = macx.MACCellParams(cfg, 3)
    bucket = macx.dp.OverlappedBuckets(params)
    flat = params.grad_buffer()
    assert 0 < bucket.early < flat.numel() and bucket.early == params.early_floats()
    late_names = [f for f in params.fields if f in macx.params.LATE_FIELDS]
    assert params.fields[-len(late_names):] == late_names          # the phase-2 gradients close the buffer
    shard, glob = (2, 5) if rank == 0 else (3, 5)
    bucket.begin_step(shard, glob)
    # what the cell's backward does: gradients are views of the flat buffer; phase 1 fills the front, calls the hook, phase 2 the rest
    off = 0
    for t in params.tensors():
        t.grad = flat[off: off + t.numel()].view_as(t)
        off += (t.numel() + 3) & ~3
    flat.zero_()
    flat[: bucket.early] = float(rank + 1)
    params.after_backward_phase1(flat)
    flat[bucket.early:] = float(10 * (rank + 1))
    bucket.allreduce_(shard, glob)
    want_early = 1.0 * 2 / 5 + 2.0 * 3 / 5
    ok = bool(torch.allclose(flat[: bucket.early], torch.full((bucket.early,), want_early)))
    ok = ok and bool(torch.allclose(flat[bucket.early:], torch.full((flat.numel() - bucket.early,), 10 * want_early)))
    ok = ok and all(t.grad.data_ptr() >= flat.data_ptr() for t in params.tensors())
    if rank == 0:
        ret["ok"], ret["overlapped"] = ok, bucket.overlapped_steps
    dist.destroy_process_group()


def test_overlapped_buckets_two_ranks():
    """macx.dp.OverlappedBuckets: the early bucket starts from the backward pass's phase-1 hook, the late one after it; both
    weighted by shard size; gradients stay views of the flat buffer."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_overlap_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["ok"] and ret["overlapped"] == 1


def _two_phase_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import macx
    from oracle import mac_oracle as mo
    cfg = mo.flag_file_config("args", netLength=3, memDim=128, ctrlDim=128, attDim=128)
    params = macx.MACCellParams(cfg, 3)
    bucket = macx.dp.OverlappedBuckets(params)
    flat = params.grad_buffer()
    shard, glob = (2, 5) if rank == 0 else (3, 5)
    g = torch.Generator().manual_seed(100 + rank)
    values = torch.randn(flat.numel(), generator=g)               # this rank's "gradients"

    def views():
        out, off = {}, 0
        for f, t in zip(params.fields, params.tensors()):
            out[f] = flat[off: off + t.numel()].view_as(t)
            off += (t.numel() + 3) & ~3
        return out

    # the eager data-parallel step, as the cell's autograd node drives the bucket: phase 1 fills the front, hook, phase 2 the rest
    bucket.begin_step(shard, glob)
    for f, v in views().items():
        getattr(params, f).grad = v
    flat.zero_()
    flat[: bucket.early] = values[: bucket.early]
    params.after_backward_phase1(flat)
    flat[bucket.early:] = values[bucket.early:]
    want = bucket.allreduce_(shard, glob).clone()

    class Parts(macx.dp.TwoPhaseStep):          # the two "graph replays" as stand-ins: they write what the phases would
        def run_part_a(self):
            flat.zero_()
            flat[: bucket.early] = values[: bucket.early]
            return views()

        def run_part_b(self):
            flat[bucket.early:] = values[bucket.early:]

    for t in params.tensors():
        t.grad = None
    step = Parts(params, bucket, shard, glob)
    same = []
    for _ in range(3):
        got = step.exchange_step()
        same.append(bool(torch.equal(got, want)) and all(t.grad is not None and t.grad.data_ptr() >= flat.data_ptr() for t in params.tensors()))
    if rank == 0:
        ret["same"], ret["overlapped"] = same, bucket.overlapped_steps
    dist.destroy_process_group()


def test_two_phase_step_equals_the_eager_dp_step_bit_for_bit():
    """macx.dp.TwoPhaseStep (the host side of graph.CapturedDPTrainStep: part A, early bucket from the phase-1 hook, part B, late
    bucket) leaves the flat buffer bit for bit as the eager sequence does, step after step, with the early bucket in flight each time."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 37500 + (os.getpid() % 2000)
    mp.spawn(_two_phase_worker, args=(2, port, ret), nprocs=2, join=True)
    assert list(ret["same"]) == [True, True, True]
    assert ret["overlapped"] == 4
