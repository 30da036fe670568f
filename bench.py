#!/usr/bin/env python3
"""bench.py -- questions/sec (fwd+bwd) of the MAC cell on MI355X, BASELINE.json's metric.

A "step" is one pass of the hot path over one synthetic CLEVR-shaped batch: p = 12 applications of
the MAC cell (control -> read -> write) forward in training mode (dropout keep .85/.85/1.0) plus the
full backward (parameter, knowledge-base, question-word and question-vector gradients), through the
C ABI of libmacx.so.  For N > 1 the metric's batch of 64 questions is split over the ranks by the
reference's tower rule (strong scaling: the configuration BASELINE.json's metric names) and the flat
gradient buffer is all-reduced over RCCL inside the timed step; the same line also carries the
weak-scaling number (64 questions per GPU) and BASELINE configs[3] (128 per GPU, p = 16).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (metric, value, roofline, cpu_baseline ...).  Inputs are resident in HBM
before the timed region; nothing under /root/reference is read.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B, S, N, D, P = 64, 50, 196, 512, 12
PEAK_FP32_MFMA = 157.3e12       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA = 2500e12        # MI355X_MICROARCH.md: fp16/bf16 dense MFMA peak (~2.5 PF).  Measured on this instruction with nothing else in the
                                # loop (profiles/r04_mfma_probe.txt): 2440 TF on all-zero operands, 1880 TF on random fp16 operands (DVFS)
METRIC_BLOCKS = 5               # the metric is timed as this many blocks of --steps steps; the MEDIAN block is `value` / `ms_per_step`


def flops_per_question_step(n=N, s=S, d=D):
    """F = 2 d^2 (4N + 5) + 6 N d + 5 S d  (reference op count, forward, args.txt)."""
    return 2 * d * d * (4 * n + 5) + 6 * n * d + 5 * s * d


def physical_cores():
    """distinct (physical id, core id) pairs of /proc/cpuinfo; None if it cannot be read"""
    try:
        cores, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        return len(cores) or None
    except Exception:
        return None


def cpu_baseline(seed, iters, sample_b):
    """The op-for-op torch-CPU restatement of the reference graph (oracle/, kind = "port") timed on this host's cores on the metric's
    own configuration (B = 64, p = 12, train mode, fwd + bwd): a thread-count sweep (8 / 32 / the physical cores; one
    warm + one timed pass each), then `iters` timed passes at the best count, median.  About 30 s of CPU work."""
    from oracle import mac_oracle as mo
    cfg = mo.flag_file_config("args", netLength=P, memDim=D, ctrlDim=D, attDim=D)
    vq, words, lengths, kb = mo.synthetic_inputs(sample_b, S, N, D, seed=seed)
    vs = mo.VarStore(generator=torch.Generator().manual_seed(seed), requires_grad=True)
    keeps = (cfg.memoryDropout, cfg.readDropout, cfg.writeDropout)
    mask_fn = mo.hash_mask_fn(seed, keeps)
    gm = torch.randn(sample_b, D, generator=torch.Generator().manual_seed(1)) / sample_b

    def one():
        kbr = kb.clone().requires_grad_(True)
        t0 = time.perf_counter()
        c, m, _ = mo.mac_network(cfg, vs, vq, words, words, lengths, kbr, train=True, mask_fn=mask_fn, keeps=keeps)
        (m * gm).sum().backward()
        dt = time.perf_counter() - t0
        for v in vs.params.values():
            v.grad = None
        return dt

    ncpu = os.cpu_count() or 1
    sweep = sorted({min(8, ncpu), min(32, ncpu), min(physical_cores() or ncpu, ncpu)})
    per_threads = {}
    for t in sweep:
        torch.set_num_threads(t)
        one()                                # (the first pass creates the variables / warms the allocator and the thread pool)
        per_threads[t] = one()
    best = min(per_threads, key=per_threads.get)
    torch.set_num_threads(best)
    times = sorted(one() for _ in range(max(iters, 5)))
    med = times[len(times) // 2]
    return {"value": round(sample_b / med, 3), "unit": "questions/s", "cores": best,
            "thread_sweep_questions_per_s": {str(t): round(sample_b / dt, 2) for t, dt in per_threads.items()},
            "host_logical_cpus": ncpu, "host_physical_cores": physical_cores(), "kind": "port",
            "sample": "%d x (B=%d, S=%d, N=%d, d=%d, p=%d) fwd+bwd, torch-CPU fp32 op-for-op restatement of the TF1 graph "
                      "(oracle/mac_oracle.py), median; thread count = the best of the sweep" % (len(times), sample_b, S, N, D, P)}


def model_level(macx, dev, seed, steps=6):
    """Secondary number that honours "KB = 14x14x1024": the whole tower body of MACnet.build -- question encoder
    (embedding 300 + biLSTM 2x256) and stem CNN (1024->512->512) -> MAC cell x p -> output unit + classifier -> mean CE,
    backward, fused clip + Adam + EMA step; batch 64, train-mode dropouts.  Inputs are the reference's feed dict:
    question word ids + lengths, 14x14x1024 image features, answer ids."""
    cfg = macx.configs.flag_file_config("args", netLength=P, memDim=D, ctrlDim=D, attDim=D)
    VOCAB = 90                                                    # CLEVR question vocabulary size (preprocess.py)
    net = macx.MACNet(cfg, vocab=VOCAB, generator=torch.Generator().manual_seed(seed)).to(dev)
    # the tower's flat gradient buffer (what the data-parallel step exchanges): the cell's backward pass writes straight into its
    # range and the optimizer reads the buffer as it is -- no per-tensor gather of the cell's 26 gradients in front of the step
    bucket = macx.dp.TowerBuckets(net)
    opt = macx.optim.FlatAdamEMA(bucket.tensors(), lr=1e-4, clip_norm=8.0, ema_decay=0.999)
    g = torch.Generator().manual_seed(seed)
    _, _, lengths, _ = macx.configs.synthetic_inputs(B, S, 1, 8, seed=seed)
    img = torch.relu(torch.randn(B, N, 1024, generator=g)).to(dev)
    qs = torch.randint(1, VOCAB + 1, (B, S), generator=g, dtype=torch.int32)
    qs = (qs * (torch.arange(S).unsqueeze(0) < lengths.unsqueeze(1)).to(torch.int32)).to(dev)
    lengths = lengths.to(dev)
    ans = torch.randint(0, 28, (B,), generator=g).to(dev)

    def one(i):
        for t in net.tensors():
            t.grad = None
        logits = net(img, qs, lengths, train=True, seed=seed + i, check_ids=False)
        loss, _ = net.loss_and_pred(logits, ans)
        bucket.begin_step(B, B)
        loss.backward()
        bucket.allreduce_(B, B)              # one process: gathers the other modules' gradients into the buffer, exchanges nothing
        opt.step(flat_grad=bucket.flat)
        return loss

    for i in range(3):
        loss = one(i)
    dt = best_block(one, steps, first=3)
    loss = one(3 + 3 * steps)
    stem_flops = 2.0 * 9 * N * (1024 * 512 + 512 * 512)          # forward, per question
    return {"value": round(B / dt, 2), "unit": "questions/s", "ms_per_step": round(dt * 1e3, 3),
            "includes": "question encoder (emb + biLSTM) + stem CNN + MAC cell (p=%d) + output unit/classifier + CE loss, "
                        "fwd+bwd, clip+Adam+EMA step; B=%d, S=%d" % (P, B, S),
            "final_loss": round(float(loss), 4),
            "flops_per_question_fwd_bwd": 3 * (P * flops_per_question_step() + stem_flops)}


def model_level_dp(macx, dev, dist, world, rank, global_batch, seed, steps=5):
    """The whole tower data-parallel (model.py:775-826): every rank runs encoder + stem + cell + classifier on its tower slice of
    `global_batch` questions, the 57 MB flat gradient is exchanged in two buckets (macx.dp.TowerBuckets: the classifier's and the
    cell's early gradients from the cell's phase-1 hook, the rest after backward), then clip + Adam + EMA on the reduced buffer
    (clip after the exchange, model.py:645-650)."""
    lo, hi = macx.dp.tower_slice(global_batch, rank, world)
    bl = hi - lo
    cfg = macx.configs.flag_file_config("args", netLength=P, memDim=D, ctrlDim=D, attDim=D)
    VOCAB = 90
    net = macx.MACNet(cfg, vocab=VOCAB, generator=torch.Generator().manual_seed(seed)).to(dev)
    bucket = macx.dp.TowerBuckets(net)
    opt = macx.optim.FlatAdamEMA(bucket.tensors(), lr=1e-4, clip_norm=8.0, ema_decay=0.999)
    g = torch.Generator().manual_seed(seed + rank)
    _, _, lengths, _ = macx.configs.synthetic_inputs(bl, S, 1, 8, seed=seed + rank)
    img = torch.relu(torch.randn(bl, N, 1024, generator=g)).to(dev)
    qs = torch.randint(1, VOCAB + 1, (bl, S), generator=g, dtype=torch.int32)
    qs = (qs * (torch.arange(S).unsqueeze(0) < lengths.unsqueeze(1)).to(torch.int32)).to(dev)
    lengths = lengths.to(dev)
    ans = torch.randint(0, 28, (bl,), generator=g).to(dev)

    def one(i):
        for t in net.tensors():
            t.grad = None
        logits = net(img, qs, lengths, train=True, seed=seed + i, b0=lo, check_ids=False)
        loss, _ = net.loss_and_pred(logits, ans)
        bucket.begin_step(bl, global_batch)
        loss.backward()
        bucket.allreduce_(bl, global_batch)
        opt.step(flat_grad=bucket.flat)

    for i in range(3):
        one(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        one(3 + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    # the exchange on its own: the whole flat buffer, nothing to overlap with
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ar_ms = None
    if world > 1:
        dist.all_reduce(bucket.flat)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            dist.all_reduce(bucket.flat)
        e1.record()
        torch.cuda.synchronize()
        ar_ms = e0.elapsed_time(e1) / 3
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return {"value": round(global_batch * steps / dt, 2), "unit": "questions/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "global_batch": global_batch, "shard_on_rank0": bl, "gradient_bytes": 4 * int(bucket.flat.numel()),
            "early_bucket_bytes": 4 * int(bucket.early), "overlapped_steps": int(bucket.overlapped_steps),
            "allreduce_alone_ms": None if ar_ms is None else round(ar_ms, 3),
            "includes": "encoder + stem + cell (p=%d) + classifier fwd+bwd, two-bucket all-reduce, clip + Adam + EMA" % P}


def run_bytes(macx, cfg, b, p):
    """bytes of the two caller-owned buffers of one training run (forward `saved`, backward `ws`) at batch b, p steps"""
    L = macx._lib.lib()
    sh = macx._lib.MacxShapes(B=b, S=S, N=N, d=D, p=p, b0=0)
    opts = macx.options.freeze(cfg)
    return {"saved_bytes": 4 * int(L.macx_saved_floats(C.byref(opts), C.byref(sh), 1)),
            "ws_bytes": 4 * int(L.macx_ws_floats(C.byref(opts), C.byref(sh), 1))}


def fwd_only_p4(macx, dev, seed, p=4, steps=30):
    """BASELINE configs[1]: synthetic CLEVR-shape (B=64, S=50, KB=14x14, d=512, p=4) FORWARD ONLY on one MI355X: evaluation
    mode (no dropout), nothing kept for a backward pass."""
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=D, ctrlDim=D, attDim=D)
    vq, words, lengths, kb = macx.configs.synthetic_inputs(B, S, N, D, seed=seed)
    params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(seed)).to(dev)
    vqd, wd, kbd, ld = vq.to(dev), words.to(dev), kb.to(dev), lengths.to(dev)

    def fwd():
        with torch.no_grad():
            cell = macx.MACCell(vecQuestions=vqd, questionWords=wd, questionCntxWords=wd, questionLengths=ld, knowledgeBase=kbd,
                                memoryDropout=1.0, readDropout=1.0, writeDropout=1.0, batchSize=B, train=False, config=cfg,
                                params=params)
            return cell.run().memory

    for _ in range(6):
        fwd()
    dt_eager = best_block(lambda i: fwd(), steps)
    # the same run replayed from one captured HIP graph (macx.CapturedForward): ~70 launches of 5-80 us each are host-bound when
    # issued one ctypes call at a time; the batch is copied into the captured run's input tensors inside the timed region
    cap = macx.CapturedForward(cfg, params, B, S, N)
    ref = fwd()
    got = cap(vqd, wd, ld, kbd)
    torch.cuda.synchronize()
    if not torch.equal(ref, got):
        raise SystemExit("captured forward differs from the eager forward")
    for _ in range(6):
        cap(vqd, wd, ld, kbd)
    dt = best_block(lambda i: cap(vqd, wd, ld, kbd), steps)
    units = 4 + 3 * (p - 1)            # knowledge-base products executed: step 0 all four, later steps reuse X
    executed = 3.0 * units * 2.0 * B * N * D * D
    return {"value": round(B / dt, 1), "unit": "questions/s", "ms_per_batch": round(dt * 1e3, 3), "steps": steps, "p": p, "batch": B,
            "launch": ("one captured HIP graph per batch (inputs copied in)" if cap.captured else
                       "EAGER launches: this process's graph replays failed CapturedForward's self-check (mac-network_amd/graph.py)")
                      + "; eager ctypes launches: %.3f ms per batch = %.0f questions/s" % (dt_eager * 1e3, B / dt_eager),
            "graph_replay": bool(cap.captured),
            "hoist": "evaluation has no read dropout, so the projected knowledge base X = KB Wx + bx is step-invariant: step 0 "
                     "computes it, steps 1..p-1 read it back (the reference's graph recomputes it per step, ops.py:688)",
            "reference_flops_per_question": p * flops_per_question_step(),
            "roofline": {"bound": "mfma", "executed_tflops": round(executed / dt / 1e12, 1), "peak": PEAK_BF16_MFMA / 1e12,
                         "frac": round(executed / dt / PEAK_BF16_MFMA, 4),
                         "note": "whole forward pass (launch gaps and the [B,d] kernels included) over the fp16-pipe FLOPs it executes"}}


def gqa_shape_p4(macx, dev, seed, flag_file, steps=8):
    """BASELINE configs[4]: GQA-shape knowledge base 7 x 7 x 2048 -> stem (2048 -> 512 -> 512) -> MAC cell with N = 49, p = 4 and
    the write unit of configs/args3.txt (self-attention) / args4.txt (self-attention + memory gate), fwd + bwd, B = 64."""
    p, Ng, H, W, Cin = 4, 49, 7, 7, 2048
    cfg = macx.configs.flag_file_config(flag_file, netLength=p, memDim=D, ctrlDim=D, attDim=D)
    stem = macx.Stem(cfg, H=H, W=W, inDim=Cin, generator=torch.Generator().manual_seed(seed)).to(dev)
    params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(seed)).to(dev)
    vq, words, lengths, _ = macx.configs.synthetic_inputs(B, S, 1, D, seed=seed)
    vqd, wd = [t.to(dev).requires_grad_(True) for t in (vq, words)]
    ld = lengths.to(dev)
    img = torch.relu(torch.randn(B, Ng, Cin, generator=torch.Generator().manual_seed(seed))).to(dev)
    gmem = (torch.randn(B, D, generator=torch.Generator().manual_seed(1)) / B).to(dev)
    leaves = [vqd, wd] + stem.tensors() + params.tensors()

    def one(i):
        for t in leaves:
            t.grad = None
        kb = stem(img, train=True, seed=seed + i)
        cell = macx.MACCell(vecQuestions=vqd, questionWords=wd, questionCntxWords=wd, questionLengths=ld, knowledgeBase=kb,
                            memoryDropout=cfg.memoryDropout, readDropout=cfg.readDropout, writeDropout=cfg.writeDropout,
                            batchSize=B, train=True, config=cfg, params=params, seed=seed + i)
        torch.autograd.backward([cell.run().memory], [gmem])

    for i in range(3):
        one(i)
    dt = best_block(one, steps, first=3)
    cell_flops = 3.0 * p * flops_per_question_step(n=Ng)
    stem_flops = 3.0 * 2.0 * 9 * Ng * (Cin * 512 + 512 * 512)
    return {"value": round(B / dt, 1), "unit": "questions/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps, "p": p, "batch": B,
            "includes": "stem 7x7x%d -> 512 -> 512 + MAC cell (configs/%s.txt, N=%d, p=%d), fwd+bwd, train-mode dropout" % (Cin, flag_file, Ng, p),
            "reference_flops_per_question": cell_flops + stem_flops,
            "fp32_equiv_tflops": round(B / dt * (cell_flops + stem_flops) / 1e12, 1)}


def train_step_graph(macx, dev, seed, steps=20):
    """The metric's step (B=64, p=12, fwd+bwd, train-mode dropout) replayed from ONE captured HIP graph (macx.CapturedTrainStep):
    what the step costs when the host is out of the loop.  Every replay draws fresh dropout masks, as a training loop does: the
    seed is baked into the capture, the run's mask word (macx_dropout.mask_word, device memory) is rewritten per replay."""
    cfg = macx.configs.flag_file_config("args", netLength=P, memDim=D, ctrlDim=D, attDim=D)
    vq, words, lengths, kb = macx.configs.synthetic_inputs(B, S, N, D, seed=seed)
    params = macx.MACCellParams(cfg, P, generator=torch.Generator().manual_seed(seed)).to(dev)
    cap = macx.CapturedTrainStep(cfg, params, B, S, N, seed=seed)
    gm = (torch.randn(B, D, generator=torch.Generator().manual_seed(1)) / B).to(dev)
    cap.load(vq.to(dev), words.to(dev), lengths.to(dev), kb.to(dev), gm)
    for it in range(4):
        cap.replay(iteration=it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        cap.replay(iteration=4 + it)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(B / dt, 1), "unit": "questions/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
            "graph_replay": bool(cap.captured), "masks": "fresh per replay (mask word rewritten in device memory, one 4-byte fill per step)",
            "launch": ("one captured HIP graph per step (forward + full backward; verified bit for bit against the eager step, all "
                       "gradients, three replays)" if cap.captured else
                       "EAGER launches: this process's replays failed CapturedTrainStep's self-check (mac-network_amd/graph.py)")}


def latest_roofline_inputs():
    """profiles/latest_roofline_inputs.json if present, else the newest profiles/rNN_roofline_inputs.json (tools/profile_post.py
    writes both): per-launch HBM bytes (PMC) and in-step kernel averages of the dominant kernel"""
    import glob
    d = os.path.join(ROOT, "profiles")
    cands = [os.path.join(d, "latest_roofline_inputs.json")] + sorted(glob.glob(os.path.join(d, "r[0-9][0-9]_roofline_inputs.json")), reverse=True)
    for c in cands:
        try:
            out = json.load(open(c))
            out.setdefault("file", os.path.relpath(c, ROOT))
            return out
        except Exception:
            continue
    return {}


def train_b128_p12(macx, dev, dist, seed, steps=8):
    """BASELINE configs[2]: netLength 12, batch 128, fwd + bwd + global-norm clip + Adam + EMA (model.py:615-669) on one MI355X."""
    b, p = 128, 12
    step, params, _, _ = make_step(macx, dev, dist, 1, 0, b, p, seed)
    opt = macx.optim.FlatAdamEMA(params.tensors(), lr=1e-4, clip_norm=8.0, ema_decay=0.999)

    def one(i):
        step(i)
        opt.step()

    for i in range(4):
        one(i)
    dt = best_block(one, steps, first=4)
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=D, ctrlDim=D, attDim=D)
    F = flops_per_question_step()
    out = {"value": round(b / dt, 1), "unit": "questions/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps, "p": p, "batch": b,
           "includes": "cell fwd + bwd (train-mode dropout) + clip + Adam + EMA over the cell's %d parameters"
                       % sum(t.numel() for t in params.tensors()),
           "vs_f32_mfma_roof": round(b / dt * 3 * p * F / PEAK_FP32_MFMA, 4),
           "executed_fp16_frac": round(b / dt * 3 * (3 * p * F) / PEAK_BF16_MFMA, 4)}
    out.update(run_bytes(macx, cfg, b, p))
    return out


def best_block(one, steps, first=0, blocks=3):
    """seconds per step: the fastest of `blocks` timed blocks of `steps` steps (side legs only -- boxes of the pool have bursts of
    host-side interference that double or triple an eager leg for seconds at a time; the metric's own timing is K steps, once)"""
    best, i = None, first
    for _ in range(blocks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one(i)
            i += 1
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        best = dt if best is None else min(best, dt)
    return best


def mode_is_h2(L):
    return L.macx_gemm_mode(-1) == 2


def settle(gstep, estep, max_s):
    """untimed: wait for a quiet box (see the call site); returns what it saw"""
    def block(fn, n=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            fn(i)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    t_start, prev, log = time.perf_counter(), None, []
    while True:
        g, e = block(gstep), block(estep)
        log.append((round(g, 3), round(e, 3)))
        quiet = prev is not None and abs(g - prev) <= 0.02 * prev and e <= 1.08 * g
        if quiet or time.perf_counter() - t_start > max_s:
            return {"blocks_replay_eager_ms": log[-6:], "n_blocks": len(log), "seconds": round(time.perf_counter() - t_start, 1), "quiet": bool(quiet)}
        prev = g


def time_steps(step, steps, warmup, prime, barrier, world, dev, dist):
    for i in range(prime + warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    trace = os.environ.get("MACX_BENCH_TRACE")
    for i in range(steps):
        if trace:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
        step(prime + warmup + i)
        if trace:
            torch.cuda.synchronize()
            print("[trace] step %d: %.1f ms" % (i, (time.perf_counter() - t1) * 1e3), file=sys.stderr, flush=True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


def time_blocks(step, steps, warmup, prime, barrier, world, dev, dist, blocks=METRIC_BLOCKS):
    """The metric's timing: `blocks` blocks of EXACTLY `steps` steps, each bracketed by barrier + synchronize on both sides and
    reduced with MAX over the ranks; returns the blocks' seconds in the order they ran.  One block is what time_steps measures;
    several of them let the line say how far two runs of the same build can differ (boxes of the pool: DESIGN 9.8)."""
    out, first = [], 0
    for b in range(blocks):
        out.append(time_steps(lambda i: step(first + i), steps, warmup if b == 0 else 0, prime if b == 0 else 0, barrier, world, dev, dist))
        first += steps + (warmup + prime if b == 0 else 0)
    return out


def block_summary(dts, steps, global_batch):
    """median block -> (seconds of the median block, fields for the JSON line)"""
    order = sorted(dts)
    med = order[len(order) // 2]
    ms = [round(d / steps * 1e3, 3) for d in dts]
    return med, {"blocks": len(dts), "ms_per_step_blocks": ms, "ms_per_step_min": min(ms), "ms_per_step_max": max(ms),
                 "value_min": round(global_batch * steps / max(dts), 2), "value_max": round(global_batch * steps / min(dts), 2),
                 "spread": round((max(dts) - min(dts)) / med, 4),
                 "what": "value / ms_per_step = the MEDIAN of %d timed blocks of %d steps (each block: barrier + synchronize on both "
                         "sides, MAX over ranks); min / max / spread = (max - min) / median of the same blocks" % (len(dts), steps)}


def make_step(macx, dev, dist, world, rank, global_batch, p, seed):
    """One data-parallel step of the cell on `global_batch` questions split by the tower rule (model.py:139-149):
    forward + backward on this rank's shard, then ONE all-reduce of the flat gradient buffer."""
    lo, hi = macx.dp.tower_slice(global_batch, rank, world)
    bl = hi - lo
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=D, ctrlDim=D, attDim=D)
    vq, words, lengths, kb = macx.configs.synthetic_inputs(bl, S, N, D, seed=seed + rank)
    params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(seed)).to(dev)
    vqd, wd, kbd = [t.to(dev).requires_grad_(True) for t in (vq, words, kb)]
    ld = lengths.to(dev)
    gmem = (torch.randn(bl, D, generator=torch.Generator().manual_seed(1)) / global_batch).to(dev)
    # two buckets over the flat gradient buffer: everything but the read unit's deferred contractions is all-reduced on a
    # side stream while the last phase of the backward pass still runs (macx.dp.OverlappedBuckets)
    # RCCL: two buckets, the first in flight on a side stream while the backward pass finishes.  gloo (only ever used to exercise
    # this path with two ranks on ONE GPU) stages CUDA tensors through the host from a worker thread; with two collectives in
    # flight it was seen to stall for seconds, so it gets the single all-reduce.  MACX_BENCH_BUCKET=single|overlap overrides.
    kind = os.environ.get("MACX_BENCH_BUCKET") or ("overlap" if dist.get_backend() == "nccl" else "single") if world > 1 else None
    if kind == "single":
        bucket = macx.dp.GradBucket(params.tensors(), params=params)
        bucket.begin_step = lambda shard, glob: None
    else:
        bucket = macx.dp.OverlappedBuckets(params) if world > 1 else None

    def step(i):
        cell = macx.MACCell(vecQuestions=vqd, questionWords=wd, questionCntxWords=wd, questionLengths=ld,
                            knowledgeBase=kbd, memoryDropout=cfg.memoryDropout, readDropout=cfg.readDropout,
                            writeDropout=cfg.writeDropout, batchSize=bl, train=True, config=cfg, params=params,
                            seed=seed + i, b0=lo)
        state = cell.run()
        for t in (vqd, wd, kbd):
            t.grad = None
        for t in params.tensors():
            t.grad = None
        if world > 1:
            bucket.begin_step(bl, global_batch)
        torch.autograd.backward([state.memory], [gmem])
        if world > 1:
            bucket.allreduce_(bl, global_batch)

    step.bucket, step.slice = bucket, (lo, hi)
    return step, params, kbd, bl


def make_dp_graph_step(macx, dev, dist, params, bucket, kbd, lo, hi, global_batch, p, seed):
    """The data-parallel step as TWO graph replays (forward + backward phase 1 | backward phase 2) with the early bucket's all-reduce
    between them on the side stream and the late bucket behind them (macx.CapturedDPTrainStep): 2 replays + 2 collectives per rank
    and step instead of ~125 host-issued launches.  Checked here, on every rank, against one eager data-parallel step on the same
    inputs, seed and mask word: the all-reduced flat gradient buffer must be bit-identical.  Returns (step, ok_on_all_ranks)."""
    bl = hi - lo
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=D, ctrlDim=D, attDim=D)
    vq, words, lengths, _ = macx.configs.synthetic_inputs(bl, S, N, D, seed=seed + dist.get_rank())
    gm = (torch.randn(bl, D, generator=torch.Generator().manual_seed(1)) / global_batch).to(dev)
    vqd, wd, ld = vq.to(dev), words.to(dev), lengths.to(dev)
    # the eager reference step (the cell's autograd node drives the bucket), mask word 0
    for t in params.tensors():
        t.grad = None
    cell = macx.MACCell(vecQuestions=vqd, questionWords=wd, questionCntxWords=wd, questionLengths=ld, knowledgeBase=kbd.detach(),
                        memoryDropout=cfg.memoryDropout, readDropout=cfg.readDropout, writeDropout=cfg.writeDropout, batchSize=bl,
                        train=True, config=cfg, params=params, seed=seed, b0=lo)
    state = cell.run()
    bucket.begin_step(bl, global_batch)
    torch.autograd.backward([state.memory], [gm])
    bucket.allreduce_(bl, global_batch)
    torch.cuda.synchronize()
    want = bucket.flat.detach().clone()
    for t in params.tensors():
        t.grad = None
    del cell, state
    cap = macx.CapturedDPTrainStep(cfg, params, bucket, B=bl, S=S, N=N, global_batch=global_batch, seed=seed, b0=lo)
    cap.load(vqd, wd, ld, kbd.detach(), gm)
    ok = True
    for _ in range(2):
        cap.step()
        torch.cuda.synchronize()
        ok = ok and bool(torch.equal(bucket.flat, want))
    flag = torch.tensor([1.0 if (ok and cap.captured) else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)

    def step(i):
        cap.step(iteration=i)

    return step, bool(flag.item() > 0.5), cap


def make_graph_step(macx, dev, params, kbd, bl, p, seed):
    """The same step on one GPU as ONE captured HIP graph (macx.CapturedTrainStep: forward + every gradient, verified bit for bit
    against the eager step when it is built): step(i) rewrites the run's mask word in device memory -- fresh dropout masks per
    step, as the eager step's seed + i gives -- and replays.  Returns (step, captured)."""
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=D, ctrlDim=D, attDim=D)
    vq, words, lengths, _ = macx.configs.synthetic_inputs(bl, S, N, D, seed=seed)
    cap = macx.CapturedTrainStep(cfg, params, bl, S, N, seed=seed)
    gm = (torch.randn(bl, D, generator=torch.Generator().manual_seed(1)) / bl).to(dev)
    cap.load(vq.to(dev), words.to(dev), lengths.to(dev), kbd.detach(), gm)

    def step(i):
        cap.replay(iteration=i)

    return step, cap


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-model-level", action="store_true")
    ap.add_argument("--no-native", action="store_true", help="skip the comparison legs on the other two kernel families")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip BASELINE configs[1] (forward only, p=4) and configs[2] (B=128, Adam+EMA)")
    ap.add_argument("--no-probe", action="store_true", help="profiling runs (tools/profile_round.sh): only the timed steps, no roofline "
                    "probe launches of the dominant kernel behind them -- its in-step average in a kernel trace then counts the step's launches only")
    ap.add_argument("--no-extra-dp", action="store_true", help="N > 1: only the metric's (strong-scaling) configuration")
    ap.add_argument("--eager", action="store_true", default=bool(os.environ.get("MACX_BENCH_EAGER")),
                    help="N = 1: time the eager step instead of the captured one")
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--cpu-batch", type=int, default=B)
    ap.add_argument("--p", type=int, default=P, help=argparse.SUPPRESS)
    ap.add_argument("--per-gpu-batch", type=int, default=0, help=argparse.SUPPRESS)   # exploration only; the metric is global B=64
    args = ap.parse_args()

    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the MAC cell has no CPU path")
    ndev = torch.cuda.device_count()
    shared = os.environ.get("MACX_BENCH_BACKEND", "nccl") != "nccl"       # gloo: ranks may share one GPU (exercises the path only)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` on its own: become the launcher of N ranks, one process per GPU over RCCL -- the same command
        # line the driver uses (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)
        if ndev < args.gpus and not shared:
            raise SystemExit("bench.py --gpus %d: this node has %d GPU(s) visible (RCCL needs one device per rank; "
                             "MACX_BENCH_BACKEND=gloo lets ranks share a GPU to exercise the path)" % (args.gpus, ndev))
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, args.gpus):
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d: launch one rank per GPU "
                         "(python -m torch.distributed.run --nproc-per-node %d ... bench.py --gpus %d)"
                         % (args.gpus, world, args.gpus, args.gpus))
    if world > ndev and not shared:
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible" % (world, ndev))
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  MACX_BENCH_BACKEND=gloo lets two ranks share ONE GPU to exercise this path
        # on a single-GPU box (RCCL refuses duplicate devices); it is never used for reported numbers.
        backend = os.environ.get("MACX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import macx
    L = macx._lib.lib()
    if os.environ.get("MACX_GEMM"):         # native | split | h2 (default): kernel family of the read unit
        L.macx_gemm_mode({"native": 0, "split": 1, "h2": 2}[os.environ["MACX_GEMM"]])
    # A/B only: the per-call tuning table (macx_opts.tune) of every cell this run freezes
    for env, key in (("MACX_DBG", "phase_mask"), ("MACX_CHAIN", "chain"), ("MACX_CHAIN_KV", "chain_kv"), ("MACX_SB_DEFER", "sb_defer"),
                     ("MACX_FORCE_RT", "row_tiles")):
        if os.environ.get(env):
            macx.options.SESSION_TUNE[key] = int(os.environ[env])
    p = args.p
    seed = 1234
    global_batch = args.per_gpu_batch * world if args.per_gpu_batch else B      # the metric: 64 questions in all

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed priming before the W requested warmup steps: the caching allocator settles on the saved/ws block sizes in the
    # first two, the rest (~0.1 s of work) lets the clocks of an idle GPU ramp before anything is timed
    PRIME = 16
    step, params, kbd, bl = make_step(macx, dev, dist, world, rank, global_batch, p, seed)
    settle_dp = None
    if world > 1:
        # untimed, all ranks in lockstep: blocks of 10 steps until two consecutive blocks (MAX over ranks) agree within 3 %, 40 blocks
        # at most -- the same wait for a quiet box as on one GPU (settle() below), without a captured step to compare with
        prev, n_blk = None, 0
        for n_blk in range(1, 41):
            barrier()
            t0 = time.perf_counter()
            for i in range(10):
                step(i)
            torch.cuda.synchronize()
            tt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            cur = float(tt.item())
            if prev is not None and abs(cur - prev) <= 0.03 * prev:
                break
            prev = cur
        settle_dp = {"n_blocks": n_blk, "last_block_ms_per_step": round(cur / 10 * 1e3, 3)}
    dts = time_blocks(step, args.steps, args.warmup, PRIME, barrier, world, dev, dist, blocks=METRIC_BLOCKS if world > 1 else 1)
    # One GPU: the timed step is the product's captured training step (one HIP graph per step: the ~135 launches of a step
    # cost what the GPU needs, not what the host can issue -- boxes of the pool differ by 7x in host speed); the eager step
    # timed above stays in the line as `eager_step`.  N > 1 (and MACX_BENCH_EAGER=1) time the eager step: the exchange is not
    # captured.
    launch_mode, eager_leg, settle_log = "eager launches (one C-ABI call per unit)", None, None
    if world == 1 and not args.eager and mode_is_h2(L):
        gstep, cap = make_graph_step(macx, dev, params, kbd, bl, p, seed)
        if cap.captured:
            # Boxes of the pool were seen busy with something else for the first 20 - 30 s of a command (eager launches 1.7x slower,
            # a replay 15 % slower, both back to normal later in the same process: tools/eager_over_time.py).  Untimed
            # priming until the box is quiet, with the replay as the yardstick: blocks of 20 replays and 20 eager steps until two
            # consecutive replay blocks agree within 2 % and the eager block is within 8 % of them, 30 s at most.
            settle_log = settle(gstep, step, 30.0)
            edts = time_blocks(step, args.steps, args.warmup, 2, barrier, world, dev, dist, blocks=3)
            emed, _ = block_summary(edts, args.steps, global_batch)
            eager_leg = {"ms_per_step": round(emed / args.steps * 1e3, 3), "value": round(global_batch * args.steps / emed, 2), "unit": "questions/s",
                         "ms_per_step_blocks": [round(d / args.steps * 1e3, 3) for d in edts],
                         "what": "the same step as eager launches, same process: median of 3 blocks of --steps steps.  This is the 1-GPU "
                                 "figure an N > 1 line (eager data-parallel steps) compares with"}
            dts = time_blocks(gstep, args.steps, args.warmup, 4, barrier, world, dev, dist)
            launch_mode = ("one captured HIP graph per step (macx.CapturedTrainStep: forward + full backward, self-checked bit for bit against "
                           "the eager step incl. every gradient; the run's mask word is rewritten before each replay: fresh dropout masks per step)")
        del gstep, cap
    dp_graph_note = None
    if world > 1 and not args.eager and mode_is_h2(L):
        # N > 1: the captured data-parallel step -- two graph replays with the exchange between and behind them -- is what a rank runs
        # when its replays reproduce the eager data-parallel step bit for bit on EVERY rank; the eager step timed above stays in the
        # line as `eager_step`.  Anything else (a failed self-check on some rank, an exception while capturing) keeps the eager timing.
        try:
            lo, hi = step.slice
            cstep, ok, cap_dp = make_dp_graph_step(macx, dev, dist, params, step.bucket, kbd, lo, hi, global_batch, p, seed)
        except Exception as e:          # (every rank runs the same code on the same shapes: an exception here is an exception everywhere)
            ok, dp_graph_note = False, "capture failed: %s: %s" % (type(e).__name__, str(e)[:200])
        if ok:
            emed, _ = block_summary(dts, args.steps, global_batch)
            eager_leg = {"ms_per_step": round(emed / args.steps * 1e3, 3), "value": round(global_batch * args.steps / emed, 2), "unit": "questions/s",
                         "ms_per_step_blocks": [round(d / args.steps * 1e3, 3) for d in dts],
                         "what": "the same data-parallel step as eager launches (autograd node + phase-1 hook), same processes"}
            cdts = time_blocks(cstep, args.steps, args.warmup, 4, barrier, world, dev, dist, blocks=METRIC_BLOCKS)
            cmed, _ = block_summary(cdts, args.steps, global_batch)
            if cmed > emed:
                # (seen with two gloo ranks SHARING one GPU, where two processes' graph launches contend: the eager step stays the metric)
                dp_graph_note = ("the captured data-parallel step verified bit for bit but measured slower here (%.3f vs %.3f ms per step): "
                                 "the eager step is the metric" % (cmed / args.steps * 1e3, emed / args.steps * 1e3))
                eager_leg = None
                ok = False
        if ok:
            dts = cdts
            launch_mode = ("two captured HIP graphs per rank and step (macx.CapturedDPTrainStep: forward + backward phase 1 | backward phase 2), "
                           "the early bucket's all-reduce between them on a side stream, the late bucket behind them; self-checked bit for "
                           "bit against the eager data-parallel step on every rank; fresh dropout masks per step through the mask word")
        elif dp_graph_note is None:
            dp_graph_note = "the replayed step did not reproduce the eager data-parallel step on every rank: eager launches timed"
    if world == 1 and len(dts) == 1:          # the eager step IS the metric here (--eager, another kernel family, no capture): time the remaining blocks
        dts = dts + time_blocks(step, args.steps, 0, 0, barrier, world, dev, dist, blocks=METRIC_BLOCKS - 1)
    dt, timing = block_summary(dts, args.steps, global_batch)
    ms_per_step = dt / args.steps * 1e3
    qps = global_batch * args.steps / dt
    F = flops_per_question_step()

    extra = {}
    if world > 1 and not args.no_extra_dp:
        # the same line carries the weak-scaling reading (64 questions per GPU) and BASELINE configs[3] (128 per GPU, p = 16)
        for key, gb, pp in (("weak_scaling_b64_per_gpu", B * world, p), ("config3_dp_b128_per_gpu_p16", 128 * world, 16)):
            st2, _, _, _ = make_step(macx, dev, dist, world, rank, gb, pp, seed)
            d2 = time_steps(st2, max(5, args.steps // 2), 2, 3, barrier, world, dev, dist)
            n2 = max(5, args.steps // 2)
            extra[key] = {"value": round(gb * n2 / d2, 2), "unit": "questions/s", "global_batch": gb, "p": pp,
                          "ms_per_step": round(d2 / n2 * 1e3, 3), "steps": n2, "scaling": "weak"}
            del st2
            torch.cuda.empty_cache()
        extra["model_level_dp"] = model_level_dp(macx, dev, dist, world, rank, global_batch, seed)
        torch.cuda.empty_cache()

    if rank == 0 and args.no_probe:
        print(json.dumps({"metric": "questions/sec fwd+bwd (B=64,d=512,p=12,KB=14x14x1024) at 1/2/4/8 MI355X", "value": round(qps, 2),
                          "unit": "questions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(ms_per_step, 3), "timing": timing, "roofline": None, "launch": launch_mode, "eager_step": eager_leg,
                          "note": "--no-probe: profiling run"}), flush=True)
    elif rank == 0:
        ptr = lambda t: C.c_void_p(t.data_ptr())
        mode = L.macx_gemm_mode(-1)
        st = None
        # ---- roofline of the dominant kernel: the knowledge-base GEMM X = KBd Wx + bx (6 of the 9 GEMM-class launches per
        # cell step share its main loop).  Algorithmic FLOPs per launch = 2 (B N) d d; duration from HIP events on the stream
        # the kernel is launched on (torch's current stream is what the C ABI receives), averaged over 32 launches that each
        # read a different input and write a different output (8 rotating pairs, 2 x 8 x 25.7 MB > the 256 MB Infinity
        # Cache share that matters), as the in-step launches do.
        Bp = bl
        NBUF = 8
        bx = params.projX_b.detach()
        k_flops = 2.0 * Bp * N * D * D
        nrep = 32
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        chain_ms = None
        if mode == 2:
            # the read unit's forward chain kernel (KB -> X -> H1 -> I2 -> logits, one launch per cell step): the library re-launches
            # it between two HIP events on this stream (macx_read_chain_time), rotating the output buffers over the run's p steps
            cfg_r = macx.configs.flag_file_config("args", netLength=p, memDim=D, ctrlDim=D, attDim=D)
            vq_r, w_r, len_r, _ = macx.configs.synthetic_inputs(Bp, S, N, D, seed=seed)
            vqd_r, wd_r, ld_r = vq_r.to(dev), w_r.to(dev), len_r.to(dev)
            cell_r = macx.MACCell(vecQuestions=vqd_r.detach(), questionWords=wd_r.detach(), questionCntxWords=wd_r.detach(),
                                  questionLengths=ld_r, knowledgeBase=kbd.detach(), memoryDropout=cfg_r.memoryDropout,
                                  readDropout=cfg_r.readDropout, writeDropout=cfg_r.writeDropout, batchSize=Bp, train=True,
                                  config=cfg_r, params=params, seed=seed)
            run = macx.cell._Run(cell_r, True)
            # (1) IN A RUNNING FORWARD PASS: the library launches each of a pass's p chain kernels with a start and a stop HIP event on
            # this stream (macx_cell_forward_chain_time: the kernel's own timestamps) -- the step's [B,d] linear in front of a launch,
            # the attention kernel behind it, as in the timed step.  Median of five passes = roofline.kernel_ms.
            ms = C.c_float(0.0)
            in_step = []
            for _ in range(6):
                rc = L.macx_cell_forward_chain_time(*run._common(), C.byref(ms), run.stream)
                if rc != 0:
                    break
                in_step.append(float(ms.value))
            chain_in_step_ms = sorted(in_step[1:])[len(in_step[1:]) // 2] if len(in_step) > 1 else None
            # (2) back to back: 48 launches between ONE event pair (macx_read_chain_time); the next launch starts while the previous
            # one's 100 MB of stores still drain -- a side figure
            run.begin()
            run.step(0)
            rc = L.macx_read_chain_time(*run._common()[:7], 0, 48, C.byref(ms), run.stream)
            chain_b2b_ms = float(ms.value) if rc == 0 else None
            chain_ms = chain_in_step_ms if chain_in_step_ms is not None else chain_b2b_ms
            del run, cell_r
        if chain_ms is not None:
            k_ms = chain_ms
            # reference op count of what the launch computes: projX 2 N d^2 + memKbProj 4 N d^2 (K = 2d) + memKbProj_2
            # 2 N d^2 per question = 4 x 2 (B N) d^2
            k_flops = 4 * 2.0 * Bp * N * D * D
            c_ms = None
            kname = ("chain_fwd_kernel<512> (read unit forward: dropout(KB) -> X -> H1 -> I2 -> attention logits in one launch; H2 tiles "
                     "stay in LDS between the products; 3 x v_mfma_f32_16x16x32_f16 per product, fp32 accumulate)")
            terms, pipe = 3, "fp16 (v_mfma_f32_16x16x32_f16)"
        elif mode == 2:
            # the library launches the kernel KREP times per call (C-side loop, read once from the environment) so that the
            # python / ctypes cost of a call -- tens of microseconds on some hosts, more than the kernel -- drops out of the
            # HIP-event interval
            KREP = 8
            os.environ["MACX_H2_DEBUG_REPS"] = str(KREP)
            hf = L.macx_h2_floats(Bp * N, D)
            wh = torch.empty(D * D + 64, device=dev)
            macx._lib.check(L.macx_h2_pack_weight(ptr(params.projX_W.detach()), D, D, 0, ptr(wh), st), "pack")
            hin = [torch.empty(hf, device=dev) for _ in range(NBUF)]
            hout = [torch.empty(hf, device=dev) for _ in range(NBUF)]
            src = [(kbd.detach() * (1.0 + 0.01 * i)).contiguous() for i in range(NBUF)]
            for i in range(NBUF):
                macx._lib.check(L.macx_h2_from_f32(ptr(src[i]), Bp, N, D, ptr(hin[i]), st), "from")
                macx._lib.check(L.macx_h2_gemm_planes(ptr(hin[i]), Bp, N, D, ptr(wh), D, ptr(bx), 0, ptr(hout[i]), st), "gemm")
            e0.record()
            for i in range(nrep):
                L.macx_h2_gemm_planes(ptr(hin[i % NBUF]), Bp, N, D, ptr(wh), D, ptr(bx), 0, ptr(hout[i % NBUF]), st)
            e1.record()
            torch.cuda.synchronize()
            k_ms = e0.elapsed_time(e1) / (nrep * KREP)
            # an HBM-bound kernel of the same step, timed the same way: fp32 knowledge base -> H2 (read 4 B, write 4 B per element)
            e0.record()
            for i in range(nrep):
                L.macx_h2_from_f32(ptr(src[i % NBUF]), Bp, N, D, ptr(hin[i % NBUF]), st)
            e1.record()
            torch.cuda.synchronize()
            c_ms = e0.elapsed_time(e1) / nrep
            del hin, hout, src
            kname = ("kb_gemm_h2_kernel<13,B_PLAIN,E_BIAS_ACT,false> (X = KBd Wx + bx; operands as two fp16 planes with per-(row,128-col) "
                     "exponents, 3 x v_mfma_f32_16x16x32_f16 per product, fp32 accumulate)")
            terms, pipe = 3, "fp16 (v_mfma_f32_16x16x32_f16)"
        else:
            sh = macx._lib.MacxShapes(B=Bp, S=S, N=N, d=D, p=p, b0=0)
            dp = macx._lib.MacxDropout(keep_memory=1.0, keep_read=1.0, keep_write=1.0, seed=seed)
            wp = torch.empty(2 * D * D, device=dev)
            macx._lib.check(L.macx_pack_weight(ptr(params.projX_W.detach()), D, D, macx._lib.kb_pack_flags(), ptr(wp), st), "pack")
            kbs = [kbd.detach().clone() for _ in range(NBUF)]
            xos = [torch.empty(Bp, N, D, device=dev) for _ in range(NBUF)]
            for i in range(NBUF):
                L.macx_kb_project(C.byref(sh), C.byref(dp), 0, ptr(kbs[i]), ptr(wp), ptr(bx), ptr(xos[i]), None, st)
            e0.record()
            for i in range(nrep):
                L.macx_kb_project(C.byref(sh), C.byref(dp), 0, ptr(kbs[i % NBUF]), ptr(wp), ptr(bx), ptr(xos[i % NBUF]), None, st)
            e1.record()
            torch.cuda.synchronize()
            k_ms = e0.elapsed_time(e1) / nrep
            c_ms = None
            del kbs, xos
            kname = ("kb_gemm6_kernel<13,A_PLAIN,B_PLAIN,E_BIAS_ACT,false> (3 x bf16 split, 6 MFMA terms)" if mode == 1 else
                     "kb_gemm_kernel<13,8,A_PLAIN,B_PLAIN,E_BIAS_ACT,false> (v_mfma_f32_16x16x4_f32)")
            terms, pipe = (6, "bf16 (v_mfma_f32_16x16x32_bf16)") if mode == 1 else (1, "f32 (v_mfma_f32_16x16x4_f32)")
        alg = k_flops / (k_ms * 1e-3)
        peak = PEAK_BF16_MFMA if mode else PEAK_FP32_MFMA
        # per-launch HBM bytes (PMC) and in-step kernel averages (rocprofv3 --kernel-trace), written by tools/profile_round.sh
        prof = latest_roofline_inputs()
        # SURVEY 8d's convention: `achieved` = ALGORITHMIC flops of the launch (one fp32 product counted once, the reference's op count of
        # what the launch computes) / its in-step duration; `frac` = achieved / the dense peak of the pipe the products execute on.
        # What the pipe EXECUTES for it (3 fp16 terms per product) is the side field executed_*.
        roofline = {"bound": "mfma", "kernel": kname,
                    "achieved": round(alg / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s", "frac": round(alg / peak, 4),
                    "kernel_ms": round(k_ms, 4),
                    "kernel_ms_how": ("start / stop HIP events of each chain launch of a running forward pass on the launch stream "
                                      "(hipExtLaunchKernelGGL: the kernel's own timestamps; macx_cell_forward_chain_time), median of 5 "
                                      "passes x %d launches.  Since round 6 a launch also carries the filler workgroups on the CUs its "
                                      "tile grid leaves idle (the previous step's write linear, this step's y, the next step's stage 0: "
                                      "DESIGN 3.0); the FLOPs counted are the tiles' four products only" % p) if chain_ms is not None
                                     else "HIP events around %d back-to-back launches" % nrep,
                    "algorithmic_flops_per_launch": k_flops,
                    "traffic": prof.get("hbm_bytes_per_launch"), "traffic_source": prof.get("source"), "profile_file": prof.get("file"),
                    "pipe": pipe, "mfma_terms_per_product": terms,
                    "executed_tflops": round(terms * alg / 1e12, 2), "executed_frac": round(terms * alg / peak, 4),
                    "back_to_back_kernel_ms": None if chain_ms is None or chain_b2b_ms is None else round(chain_b2b_ms, 4),
                    "profile_in_step_kernel_ms": prof.get("in_step_kernel_ms"),
                    # what the pipe sustains on THIS instruction with random fp16 operands, all 256 CUs, nothing else in the loop
                    # (tools/probes/mfma_probe.hip, profiles/r04_mfma_probe.txt): the matrix pipe's rate is data-dependent -- the chip
                    # clocks to its power budget -- 2440 TF on zero operands, 1880 TF on random ones; `peak` stays the guide's dense figure
                    "pipe_rate_random_operands_tflops": 1880.0 if mode == 2 else None,
                    # chain kernel: the fp32 knowledge base in; dropout(KB), X, H1, I2 out as H2 (4 B per element) + keep bits; 4 weights
                    "algorithmic_bytes_per_launch": ((1 + 4) * Bp * N * D * 4 + 2 * Bp * N * D // 8 + 4 * D * D * 4) if chain_ms is not None
                                                    else 2 * Bp * N * D * 4 + D * D * 4,
                    # the whole step priced by the REFERENCE's op count (3 p F per question) against the f32-input MFMA
                    # peak (157.3 TF), the arithmetic the metric is stated in.  A RATIO, not a roofline fraction: the step does not run
                    # under that roof (its products execute on the fp16 pipe), so values above 1 are expected
                    "vs_f32_mfma_roof": round(qps / world * 3 * p * F / PEAK_FP32_MFMA, 4),
                    # ... and the whole step's executed fp16-pipe flops (3 terms x 22/24 of the reference count: the y-fold) / 2.5 PF
                    "whole_step_executed_frac": round(qps / world * 3 * p * F * (22.0 / 24.0) * 3 / PEAK_BF16_MFMA, 4)}
        hbm = []
        if c_ms:
            byt = 2.0 * Bp * N * D * 4
            hbm.append({"kernel": "h2_from_f32_kernel (fp32 knowledge base -> H2 through the read-dropout site)", "bound": "hbm",
                        "bytes_per_launch": byt, "kernel_ms": round(c_ms, 4), "achieved": round(byt / (c_ms * 1e-3) / 1e9, 1),
                        "peak": 8000.0, "unit": "GB/s", "frac": round(byt / (c_ms * 1e-3) / 8e12, 4), "timed": "live, HIP events"})
        for row in prof.get("hbm_kernels", []):
            hbm.append(dict(row, timed="rocprofv3 in-step average (profiles/)"))
        if hbm:
            roofline["hbm_kernels"] = hbm
        out = {"metric": "questions/sec fwd+bwd (B=64,d=512,p=12,KB=14x14x1024) at 1/2/4/8 MI355X",
               "value": round(qps, 2), "unit": "questions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 3), "timing": timing, "higher_is_better": True,
               "scaling": "weak" if args.per_gpu_batch else "strong", "vs_baseline": None,
               "dtype": "f32 (3xfp16-split emulation)" if mode == 2 else ("f32 (6xbf16-split emulation)" if mode == 1 else "f32"), "data": "synthetic",
               "dtype_note": ("fp32 in, fp32 accumulate, fp32-class results; the large contractions multiply on the fp16 matrix pipe: every "
                              "operand is stored once as x 2^e = hi + lo (two fp16, |error| <= 2^-24) with one exponent per (row, 128 "
                              "columns), a product is the 3 leading terms (dropped: lo*lo <= 2^-24 |ab|); measured error vs fp64 <= the "
                              "native f32-MFMA kernel's (tests/test_gpu_h2.py); other_families = the same step on the 6-term bf16 split "
                              "and on v_mfma_f32_16x16x4_f32") if mode == 2 else "kernel family %d" % mode,
               "config": {"workload": "MAC cell fwd+bwd, configs/args.txt options, train-mode dropout .85/.85/1.0, global batch %d "
                                      "split over %d rank(s) by the tower rule (%d questions on rank 0), S=%d, KB=[B,%d,%d] (stem output of "
                                      "14x14x1024 features), d=%d, p=%d; cell only (stem/encoder/classifier: model_level)"
                                      % (global_batch, world, bl, S, N, D, D, p),
                          "global_batch": global_batch, "parallelism": "dp%d" % world,
                          "collective": None if world == 1 else "%s all-reduce of the flat gradient buffer (RCCL: in two buckets, the first overlapped with the last phase of the backward pass on a side stream), world size %d" % (
                              "RCCL (backend nccl)" if backend == "nccl" else backend, dist.get_world_size()),
                          "flops_per_question_fwd_bwd": 3 * p * F,
                          **run_bytes(macx, macx.configs.flag_file_config("args", netLength=p, memDim=D, ctrlDim=D, attDim=D), bl, p)},
               "roofline": roofline}
        out.update(extra)
        out["launch"] = launch_mode
        if dp_graph_note:
            out["launch_note"] = dp_graph_note
        out["side_legs_timing"] = ("every leg below the metric (other_families, fwd_only_p4, train_b128_p12_adam_ema, gqa_shape_*, model_level) reports "
                                   "the fastest of 3 timed blocks of its `steps` steps; the metric itself is the MEDIAN of %d blocks of K steps (`timing`)" % METRIC_BLOCKS)
        if eager_leg is not None:
            out["eager_step"] = eager_leg
            out["settle"] = settle_log
        if settle_dp is not None:
            out["settle"] = settle_dp
        if world == 1 and mode == 2 and not args.no_native:
            # the same step on the other two kernel families of the library
            fam = {}
            for name, m in (("split_bf16_6term", 1), ("native_f32_mfma", 0)):
                L.macx_gemm_mode(m)
                st3, _, _, _ = make_step(macx, dev, dist, 1, 0, global_batch, p, seed)
                for i in range(3):
                    st3(i)
                d3 = best_block(st3, 5, first=3) * 5
                fam[name] = {"value": round(global_batch * 5 / d3, 2), "unit": "questions/s", "ms_per_step": round(d3 / 5 * 1e3, 3), "steps": 5}
                del st3
            L.macx_gemm_mode(2)
            out["other_families"] = fam
        if world == 1 and not args.no_extra_legs:
            # BASELINE.json configs[1] and configs[2]
            out["fwd_only_p4"] = fwd_only_p4(macx, dev, seed)
            out["train_b128_p12_adam_ema"] = train_b128_p12(macx, dev, dist, seed)
            if eager_leg is None:         # the headline was timed on eager launches: the captured step as a side leg
                out["train_step_graph"] = train_step_graph(macx, dev, seed)
            out["gqa_shape_p4_args3"] = gqa_shape_p4(macx, dev, seed, "args3")
            out["gqa_shape_p4_args4"] = gqa_shape_p4(macx, dev, seed, "args4")
        if world == 1 and not args.no_model_level:
            out["model_level"] = model_level(macx, dev, seed)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(seed, args.cpu_iters, args.cpu_batch)
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
