"""-m gpu: no kernel writes outside the buffers the boundary sizes for it.  Every 1-D fp32 device buffer the host side allocates
while a module runs (`saved`, `ws`: sized by macx_*_saved_floats / macx_*_ws_floats) gets a sentinel-filled guard band on
both sides; after forward + backward on random shapes the bands must be untouched.  (The encoder's slab workspace was found
too small for h > pad128(E) by the shape fuzz -- this is the direct check.)"""
import contextlib
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

GUARD = 2048                 # floats per side (multiple of 4: the inner view stays 16-byte aligned)
SENTINEL = -7.25e11


@contextlib.contextmanager
def guarded_allocations():
    real_empty = torch.empty
    seen = []

    def empty(*size, **kw):
        one_d = len(size) == 1 and isinstance(size[0], int)
        if one_d and kw.get("dtype") is torch.float32 and str(kw.get("device", "cpu")).startswith("cuda") and size[0] >= 16:
            n = (size[0] + 3) & ~3
            full = real_empty(n + 2 * GUARD, **kw)
            full.fill_(SENTINEL)
            seen.append((full, n))
            return full[GUARD: GUARD + size[0]]
        return real_empty(*size, **kw)

    torch.empty = empty
    try:
        yield seen
    finally:
        torch.empty = real_empty


def assert_guards_intact(seen, what):
    assert seen, what
    for full, n in seen:
        lo, hi = full[:GUARD], full[GUARD + n:]
        assert bool((lo == SENTINEL).all()) and bool((hi == SENTINEL).all()), \
            (what, "buffer of %d floats: %d guard floats overwritten below, %d above"
             % (n, int((lo != SENTINEL).sum()), int((hi != SENTINEL).sum())))


@pytest.mark.parametrize("seed", range(3))
def test_no_out_of_bounds_writes_on_random_shapes(macx, dev, seed):
    from helpers import make_case
    from test_gpu_cell import build_cell
    from test_gpu_encoder import make_questions
    from oracle import mac_oracle as mo
    rnd = random.Random(4000 + seed)
    for case in range(4):
        # ---- fused cell
        name = rnd.choice(["args", "args1", "args3", "args4"])
        B, S, N, d, p = rnd.choice([1, 2, 5, 9, 64]), rnd.randint(3, 20), rnd.choice([1, 7, 49, 196, 209, 420]), rnd.choice([128, 256, 512]), rnd.randint(1, 6)
        train = rnd.random() < 0.7
        cfg, vq, words, lengths, kb = make_case(name, B, S, N, d, p, writeDropout=rnd.choice([1.0, 0.9]))
        with guarded_allocations() as seen:
            cell, params, (vqd, wd, kbd) = build_cell(macx, dev, cfg, vq, words, lengths, kb, train, seed=3, requires_grad=True)
            st = cell.run()
            (st.memory.sum() + st.control.sum()).backward()
            torch.cuda.synchronize()
        assert_guards_intact(seen, ("cell", name, B, S, N, d, p, train))
        # ---- question encoder
        B, S, V, E, h = rnd.choice([1, 3, 7, 16]), rnd.randint(1, 30), rnd.randint(1, 40), rnd.choice([4, 7, 50, 130, 300]), rnd.choice([128, 256])
        cfg = mo.flag_file_config("args", ctrlDim=2 * h, memDim=2 * h, attDim=2 * h, encDim=2 * h, wrdEmbDim=E)
        enc = macx.QuestionEncoder(cfg, vocab=V, generator=torch.Generator().manual_seed(3)).to(dev)
        q, lengths = make_questions(B, S, V, seed=11 + case)
        with guarded_allocations() as seen:
            w, v = enc(q.to(dev), lengths.to(dev), train=True, seed=9, b0=1)
            (w.sum() + v.sum()).backward()
            torch.cuda.synchronize()
        assert_guards_intact(seen, ("encoder", B, S, V, E, h))
        # ---- stem
        B, H, W = rnd.choice([1, 2, 5]), rnd.randint(1, 14), rnd.randint(2, 14)
        Cin, Cmid, Cout = [rnd.choice([128, 256, 384]) for _ in range(3)]
        cfg = mo.flag_file_config("args", memDim=Cout, ctrlDim=Cout, attDim=Cout)
        cfg.stemDim = Cmid
        stem = macx.Stem(cfg, H=H, W=W, inDim=Cin, generator=torch.Generator().manual_seed(1)).to(dev)
        img = torch.relu(torch.randn(B, H * W, Cin)).to(dev)
        with guarded_allocations() as seen:
            out = stem(img, train=True, seed=5, b0=0)
            out.sum().backward()
            torch.cuda.synchronize()
        assert_guards_intact(seen, ("stem", B, H, W, Cin, Cmid, Cout))
        # ---- output unit + classifier
        B, d, Hd, A = rnd.choice([1, 3, 64]), rnd.choice([64, 128, 512]), rnd.choice([16, 48, 512]), rnd.choice([2, 28, 33])
        cfg = mo.flag_file_config("args", memDim=d, ctrlDim=d, attDim=d, outClassifierDims=[Hd], answerWordsNum=A)
        clf = macx.OutputClassifier(cfg, generator=torch.Generator().manual_seed(1)).to(dev)
        mem, vq2 = torch.randn(B, d, device=dev, requires_grad=True), torch.randn(B, d, device=dev, requires_grad=True)
        with guarded_allocations() as seen:
            lg = clf(mem, vq2, train=True, seed=9, b0=2)
            lg.sum().backward()
            torch.cuda.synchronize()
        assert_guards_intact(seen, ("classifier", B, d, Hd, A))
