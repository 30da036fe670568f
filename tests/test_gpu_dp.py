"""-m gpu: two ranks (gloo, sharing the one GPU of the test box) drive the HIP cell on their tower slices with b0 and combine
the gradients through macx.dp.OverlappedBuckets -- early bucket launched from the backward pass's phase-1 hook on a side
stream, late bucket after it.  The result must equal the full-batch gradient of one process."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(macx, dev, cfg, params, data, lo, hi, Bg, bucket=None):
    vq, words, lengths, kb, wmem = data
    for t in params.tensors():
        t.grad = None
    cell = macx.MACCell(vq[lo:hi].to(dev), words[lo:hi].to(dev), words[lo:hi].to(dev), lengths[lo:hi].to(dev), kb[lo:hi].to(dev),
                        cfg.memoryDropout, cfg.readDropout, cfg.writeDropout, hi - lo, True, config=cfg, params=params, seed=11, b0=lo)
    state = cell.run()
    loss = (state.memory * wmem[lo:hi].to(dev)).sum(dim=1).mean()          # mean over the shard (model.py:596)
    if bucket is not None:
        bucket.begin_step(hi - lo, Bg)
    loss.backward()
    if bucket is not None:
        bucket.allreduce_(hi - lo, Bg)
    torch.cuda.synchronize()
    return [t.grad.detach().cpu().clone() for t in params.tensors()]


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import macx
    dev = torch.device("cuda:0")
    Bg, S, N, d, p = 5, 8, 49, 128, 3
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
    vq, words, lengths, kb = macx.configs.synthetic_inputs(Bg, S, N, d, seed=3)
    wmem = torch.randn(Bg, d, generator=torch.Generator().manual_seed(0))
    data = (vq, words, lengths, kb, wmem)
    params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(7)).to(dev)
    params.requires_grad_(True)
    bucket = macx.dp.OverlappedBuckets(params)
    lo, hi = macx.dp.tower_slice(Bg, rank, world)
    got = _run(macx, dev, cfg, params, data, lo, hi, Bg, bucket)
    if rank == 0:
        overlapped = bucket.overlapped_steps
        params.after_backward_phase1 = None
        full = _run(macx, dev, cfg, params, data, 0, Bg, Bg)
        # (the biases in front of a softmax have an analytically zero gradient: rounding noise, compared on a 1e-2 floor)
        errs = {f: float((a - b).abs().max() / max(float(b.abs().max()), 1e-2)) for f, a, b in zip(params.fields, got, full)}
        # the two-phase backward alone (a hook that does nothing) against the one-call backward, same process
        params.after_backward_phase1 = lambda flat: None
        split = _run(macx, dev, cfg, params, data, 0, Bg, Bg)
        ret["phase_split"] = max(float((a - b).abs().max()) for a, b in zip(split, full))
        ret["errs"] = {f: e for f, e in errs.items() if e > 2e-5}
        ret["worst"], ret["overlapped"] = max(errs.values()), overlapped
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_drive_the_hip_cell_and_match_the_full_batch(dev):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["phase_split"] == 0.0, "phase 1 + phase 2 of the backward pass differ from the single call"
    assert ret["overlapped"] == 1, "the early bucket did not start from the phase-1 hook"
    assert ret["worst"] < 2e-5, dict(ret["errs"])


# ---- the whole tower (model.py:775-826): encoder + stem + cell + classifier through macx.dp.TowerBuckets
def _tower(macx, dev):
    import helpers as _h  # noqa: F401
    B, H, W, Cin, d, p, S, A, V, E = 5, 4, 3, 128, 256, 2, 6, 7, 12, 20
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d, encDim=d, wrdEmbDim=E, outClassifierDims=[32],
                                        answerWordsNum=A)
    cfg.stemDim = 128
    net = macx.MACNet(cfg, vocab=V, H=H, W=W, imageInDim=Cin, answerWordsNum=A, generator=torch.Generator().manual_seed(4)).to(dev)
    g = torch.Generator().manual_seed(6)
    img = torch.relu(torch.randn(B, H * W, Cin, generator=g))
    lengths = torch.randint(2, S + 1, (B,), generator=g, dtype=torch.int32)
    lengths[0] = S
    q = torch.randint(1, V + 1, (B, S), generator=g, dtype=torch.int32)
    q = q * (torch.arange(S).unsqueeze(0) < lengths.unsqueeze(1)).to(torch.int32)
    ans = torch.randint(0, A, (B,), generator=g)
    return net, (img, q, lengths, ans), B


def _tower_step(net, data, dev, lo, hi, Bg, bucket=None):
    img, q, lengths, ans = data
    for t in net.tensors():
        t.grad = None
    logits = net(img[lo:hi].to(dev), q[lo:hi].to(dev), lengths[lo:hi].to(dev), train=True, seed=21, b0=lo)
    loss, _ = net.loss_and_pred(logits, ans[lo:hi].to(dev))          # mean over the shard (model.py:596)
    if bucket is not None:
        bucket.begin_step(hi - lo, Bg)
    loss.backward()
    if bucket is not None:
        bucket.allreduce_(hi - lo, Bg)
    torch.cuda.synchronize()
    return [t.grad.detach().cpu().clone() for t in net.tensors()]


def _tower_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import macx
    dev = torch.device("cuda:0")
    net, data, Bg = _tower(macx, dev)
    bucket = macx.dp.TowerBuckets(net)
    lo, hi = macx.dp.tower_slice(Bg, rank, world)
    got = _tower_step(net, data, dev, lo, hi, Bg, bucket)
    if rank == 0:
        overlapped = bucket.overlapped_steps
        flat_ok = all(t.grad.data_ptr() == bucket.flat.data_ptr() + 4 * o for t, o in zip(bucket.tensors(), bucket.offsets))
        net.cell.after_backward_phase1 = None
        full = _tower_step(net, data, dev, 0, Bg, Bg)
        errs = [float((a - b).abs().max() / max(float(b.abs().max()), 1e-2)) for a, b in zip(got, full)]
        ret["worst"], ret["overlapped"], ret["flat_ok"] = max(errs), overlapped, flat_ok
        ret["n"] = len(errs)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_whole_tower_matches_the_full_batch(dev):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_tower_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["overlapped"] == 1, "the early bucket (classifier + the cell's early fields) did not start from the phase-1 hook"
    assert ret["flat_ok"], "after the exchange every gradient must be a view of the tower's flat buffer"
    assert ret["worst"] < 3e-5, dict(ret)


def test_flat_optimizer_on_the_cells_gradient_buffer(macx, dev):
    """optim.FlatAdamEMA(..., grad_owner=params).step(flat_grad=params.grad_buffer()) == the gather path over two steps; without
    grad_owner the buffer is refused (the backward pass would not have written into it)."""
    B, S, N, d, p = 4, 6, 33, 128, 2
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
    vq, words, lengths, kb = [t.to(dev) for t in macx.configs.synthetic_inputs(B, S, N, d, seed=3)]

    def train(flat):
        params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(7)).to(dev)
        opt = macx.optim.FlatAdamEMA(params.tensors(), lr=1e-2, grad_owner=params if flat else None)
        for i in range(2):
            for t in params.tensors():
                t.grad = None
            cell = macx.MACCell(vq, words, words, lengths, kb, cfg.memoryDropout, cfg.readDropout, cfg.writeDropout, B, True,
                                config=cfg, params=params, seed=5 + i)
            cell.run().memory.square().sum().backward()
            opt.step(flat_grad=params.grad_buffer() if flat else None)
        torch.cuda.synchronize()
        return opt.flat.clone(), params

    a, _ = train(False)
    b, prm = train(True)
    assert torch.equal(a, b)
    # a grad_buffer() NOBODY registered for is refused (the backward pass would not have written into it)
    orphan = macx.MACCellParams(cfg, p).to(dev)
    other = macx.optim.FlatAdamEMA(orphan.tensors())
    with pytest.raises(ValueError):
        other.step(flat_grad=orphan.grad_buffer())


@pytest.mark.parametrize("kind", ["bucket", "overlapped"])
def test_flat_optimizer_takes_a_cell_buckets_flat_buffer(macx, dev, kind):
    """The documented cell-only data-parallel flow: bucket = GradBucket(params=prm) / OverlappedBuckets(prm) is the registered
    consumer of prm.grad_buffer(), bucket.flat IS that buffer, and FlatAdamEMA(prm.tensors()).step(flat_grad=bucket.flat) steps on
    it (round 4's guard refused the tagged tensor).  Two steps == the gather path."""
    B, S, N, d, p = 4, 6, 33, 128, 2
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
    vq, words, lengths, kb = [t.to(dev) for t in macx.configs.synthetic_inputs(B, S, N, d, seed=3)]

    def train(flat):
        params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(7)).to(dev)
        bucket = None
        if flat:
            bucket = macx.dp.GradBucket(params.tensors(), params=params) if kind == "bucket" else macx.dp.OverlappedBuckets(params)
            assert bucket.flat.data_ptr() == params.grad_buffer().data_ptr()
        opt = macx.optim.FlatAdamEMA(params.tensors(), lr=1e-2)
        for i in range(2):
            for t in params.tensors():
                t.grad = None
            cell = macx.MACCell(vq, words, words, lengths, kb, cfg.memoryDropout, cfg.readDropout, cfg.writeDropout, B, True,
                                config=cfg, params=params, seed=5 + i)
            if bucket is not None and kind == "overlapped":
                bucket.begin_step(B, B)
            cell.run().memory.square().sum().backward()
            if bucket is not None:
                bucket.allreduce_(B, B)
                assert bucket.zero_copy_steps == i + 1
            opt.step(flat_grad=bucket.flat if flat else None)
        torch.cuda.synchronize()
        return opt.flat.clone()

    assert torch.equal(train(False), train(True))


def test_tower_buckets_single_process_shard_weight(macx, dev):
    """One process, shard != global (shard emulation): TowerBuckets must scale EVERY gradient by shard / global -- the classifier's
    and the cell's early fields too (round 4 marked them 'already exchanged' in the phase-1 hook and left them unscaled)."""
    net, data, Bg = _tower(macx, dev)
    lo, hi = 1, 4
    plain = _tower_step(net, data, dev, lo, hi, Bg)
    bucket = macx.dp.TowerBuckets(net)
    got = _tower_step(net, data, dev, lo, hi, Bg, bucket)
    w = float(hi - lo) / float(Bg)
    for a, b in zip(got, plain):
        assert torch.allclose(a, b * w, rtol=1e-6, atol=1e-9)
    assert all(t.grad.data_ptr() == bucket.flat.data_ptr() + 4 * o for t, o in zip(bucket.tensors(), bucket.offsets))


# ---- the captured data-parallel step: two graph replays with the exchange between and behind them (graph.CapturedDPTrainStep)
def _captured_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import macx
    dev = torch.device("cuda:0")
    Bg, S, N, d, p = 6, 8, 49, 128, 3
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
    vq, words, lengths, kb = macx.configs.synthetic_inputs(Bg, S, N, d, seed=3)
    lo, hi = macx.dp.tower_slice(Bg, rank, world)
    dm = (torch.randn(Bg, d, generator=torch.Generator().manual_seed(0)) / (hi - lo))[lo:hi].to(dev)      # d(mean over the shard)/d memory
    params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(7)).to(dev)
    params.requires_grad_(True)
    bucket = macx.dp.OverlappedBuckets(params)
    x = [t[lo:hi].to(dev) for t in (vq, words, lengths, kb)]

    def eager():
        for t in params.tensors():
            t.grad = None
        cell = macx.MACCell(x[0], x[1], x[1], x[2], x[3], cfg.memoryDropout, cfg.readDropout, cfg.writeDropout, hi - lo, True, config=cfg,
                            params=params, seed=11, b0=lo)
        state = cell.run()
        bucket.begin_step(hi - lo, Bg)
        (state.memory * dm).sum().backward()
        bucket.allreduce_(hi - lo, Bg)
        torch.cuda.synchronize()
        return bucket.flat.detach().cpu().clone(), state.memory.detach().cpu().clone()

    want, want_mem = eager()
    for t in params.tensors():
        t.grad = None
    res = {}
    for capture in (True, False):
        step = macx.CapturedDPTrainStep(cfg, params, bucket, B=hi - lo, S=S, N=N, global_batch=Bg, seed=11, b0=lo, capture=capture)
        step.load(x[0], x[1], x[2], x[3], dm)
        for it in range(3):                               # (replays of one capture: the same step three times, mask word 0)
            mem = step.step()
            torch.cuda.synchronize()
            ok = torch.equal(bucket.flat.detach().cpu(), want) and torch.equal(mem.detach().cpu(), want_mem)
            ok = ok and all(t.grad is not None and t.grad.data_ptr() >= bucket.flat.data_ptr() for t in params.tensors())
            res[(capture, it)] = bool(ok)
        res[("captured", capture)] = bool(step.captured)
        for t in params.tensors():
            t.grad = None
        del step
    # a new mask word draws other masks, identically on the eager and the captured path of a later step (smoke: finite, different)
    if rank == 0:
        ret["res"] = {str(k): v for k, v in res.items()}
        ret["overlapped"] = bucket.overlapped_steps
    dist.barrier()
    dist.destroy_process_group()


def test_captured_dp_step_equals_the_eager_dp_step_bit_for_bit(dev):
    """Two ranks on one GPU (gloo): forward + backward phase 1 as one graph replay, the early bucket's exchange on the side stream,
    backward phase 2 as a second replay, the late bucket -- the all-reduced flat gradient buffer and the final memory are bit for bit
    those of the eager data-parallel step (autograd node + phase-1 hook), replay after replay; the same class without capture too."""
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_captured_worker, args=(2, port, ret), nprocs=2, join=True)
    res = dict(ret["res"])
    assert res["('captured', True)"] and not res["('captured', False)"]
    bad = [k for k, v in res.items() if not v and not k.startswith("('captured'")]
    assert not bad, bad
    assert ret["overlapped"] >= 7            # one eager step + 2 x 3 steps, every one with the early bucket in flight
