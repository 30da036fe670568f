"""The captured-graph inference path (macx.CapturedForward) must reproduce the eager run bit for bit, follow parameter
updates, and accept new inputs -- whether the process replays the graph or (self-check failed, see graph.py) falls back to
eager launches behind the same interface."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_captured_forward_equals_eager(macx, dev):
    B, S, N, d, p = 6, 7, 40, 128, 3
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
    params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(0)).to(dev)
    cap = macx.CapturedForward(cfg, params, B, S, N)

    def eager(vq, words, lengths, kb):
        with torch.no_grad():
            cell = macx.MACCell(vq, words, words, lengths, kb, 1.0, 1.0, 1.0, B, False, config=cfg, params=params)
            st = cell.run()
            return st.memory.clone(), [a.clone() for a in cell.attentions["kb"]]

    for seed in (1, 2):
        vq, words, lengths, kb = [t.to(dev) for t in macx.configs.synthetic_inputs(B, S, N, d, seed=seed)]
        ref, ref_att = eager(vq, words, lengths, kb)
        got = cap(vq, words, lengths, kb)
        torch.cuda.synchronize()
        assert torch.equal(ref, got)
        assert all(torch.equal(a, b) for a, b in zip(ref_att, cap.attentions["kb"]))
    # parameters are read at replay time
    with torch.no_grad():
        params.projX_W.mul_(1.5)
    ref, _ = eager(vq, words, lengths, kb)
    assert torch.equal(ref, cap.replay())


def test_eager_fallback_behind_the_same_interface(macx, dev):
    B, S, N, d, p = 4, 6, 33, 128, 2
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
    params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(0)).to(dev)
    cap = macx.CapturedForward(cfg, params, B, S, N, verify=False)
    cap.captured = False                       # what a failed self-check sets
    vq, words, lengths, kb = [t.to(dev) for t in macx.configs.synthetic_inputs(B, S, N, d, seed=3)]
    with torch.no_grad():
        cell = macx.MACCell(vq, words, words, lengths, kb, 1.0, 1.0, 1.0, B, False, config=cfg, params=params)
        ref = cell.run().memory.clone()
    got = cap(vq, words, lengths, kb)
    torch.cuda.synchronize()
    assert torch.equal(ref, got)
    assert all(torch.equal(a, b) for a, b in zip(cell.attentions["kb"], cap.attentions["kb"]))


def test_captured_train_step_equals_eager(macx, dev):
    """forward + backward from one captured HIP graph: final memory and EVERY gradient bit for bit the eager step's, over
    several replays and after new inputs (nothing in the step orders itself against a memset any more)"""
    B, S, N, d, p = 6, 7, 40, 128, 3
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
    params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(0)).to(dev)
    step = macx.CapturedTrainStep(cfg, params, B, S, N, seed=77)
    assert step.captured, "the capture's self-check failed in this process"
    for seed in (1, 2):
        vq, words, lengths, kb = [t.to(dev) for t in macx.configs.synthetic_inputs(B, S, N, d, seed=seed)]
        gm = torch.randn(B, d, generator=torch.Generator().manual_seed(seed)).to(dev)
        step.load(vq, words, lengths, kb, gm)
        for _ in range(2):
            mem = step.replay().clone()
            got = [t.grad.clone() for t in step._leaves()]
            # the eager step on the same static tensors (replays write into the captured .grad tensors: keep and restore them)
            keep = [t.grad for t in step._leaves()]
            ref_mem = step._eager().clone()
            ref = [t.grad.clone() for t in step._leaves()]
            for t, g in zip(step._leaves(), keep):
                t.grad = g
            torch.cuda.synchronize()
            assert torch.equal(mem, ref_mem)
            for a, b in zip(got, ref):
                assert torch.equal(a, b)


def test_captured_train_step_draws_fresh_masks_per_replay(macx, dev):
    """one capture, a new mask word per replay (macx_dropout.mask_word): replays of different iterations differ, each equals the
    eager step under the same word bit for bit, and word 0 is the plain seed's step (a cell built without a word)"""
    B, S, N, d, p = 5, 7, 40, 128, 3
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
    params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(0)).to(dev)
    step = macx.CapturedTrainStep(cfg, params, B, S, N, seed=77)
    assert step.captured
    vq, words, lengths, kb = [t.to(dev) for t in macx.configs.synthetic_inputs(B, S, N, d, seed=1)]
    gm = torch.randn(B, d, generator=torch.Generator().manual_seed(1)).to(dev)
    step.load(vq, words, lengths, kb, gm)
    seen = []
    for it in (1, 2, 1):
        mem = step.replay(iteration=it).clone()
        got = [t.grad.clone() for t in step._leaves()]
        keep = [t.grad for t in step._leaves()]
        ref_mem = step._eager().clone()                      # eager, same device word
        ref = [t.grad.clone() for t in step._leaves()]
        for t, g in zip(step._leaves(), keep):
            t.grad = g
        torch.cuda.synchronize()
        assert torch.equal(mem, ref_mem) and all(torch.equal(a, b) for a, b in zip(got, ref))
        seen.append((mem, got))
    assert not torch.equal(seen[0][0], seen[1][0])                              # iteration 1 vs 2: other masks
    assert torch.equal(seen[0][0], seen[2][0]) and all(torch.equal(a, b) for a, b in zip(seen[0][1], seen[2][1]))   # 1 again
    step.set_mask_word(0)
    mem0 = step.replay().clone()
    kb0 = step.knowledgeBase.grad.clone()
    vq2, w2, kb2 = [t.detach().clone().requires_grad_(True) for t in (step.vecQuestions, step.words, step.knowledgeBase)]
    cell = macx.MACCell(vq2, w2, w2, step.lengths, kb2, cfg.memoryDropout, cfg.readDropout, cfg.writeDropout, B, True, config=cfg,
                        params=params, seed=77)
    for t in params.tensors():
        t.grad = None
    st = cell.run()
    torch.autograd.backward([st.memory], [step.d_memory])
    torch.cuda.synchronize()
    assert torch.equal(mem0, st.memory) and torch.equal(kb0, kb2.grad)


def test_captured_train_step_metric_shape(macx, dev):
    """... at the metric's shape (B = 64, N = 196, d = 512, p = 12): the chain kernels' 64-row tiles, the deferred contractions"""
    B, S, N, d, p = 64, 50, 196, 512, 12
    cfg = macx.configs.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d)
    params = macx.MACCellParams(cfg, p, generator=torch.Generator().manual_seed(0)).to(dev)
    step = macx.CapturedTrainStep(cfg, params, B, S, N, seed=5)         # verify=True: three replays against the eager step
    assert step.captured
