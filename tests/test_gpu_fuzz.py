"""-m gpu: randomised shapes -- the H2 (fp16-plane), the split-bf16 and the native f32-MFMA kernel families must agree on the whole cell
(final state and every gradient) to fp32 round-off for odd batch sizes, N from 1 cell to beyond one 208-row tile, both
dropout modes and all flag files (tools/mode_fuzz.py holds the generator)."""
import os
import random
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_split_and_native_kernels_agree_on_random_shapes(macx, dev):
    import mode_fuzz
    from helpers import default_gemm_mode
    rnd = random.Random(20260925)
    try:
        for case in range(10):
            name = rnd.choice(["args", "args1", "args3", "args4"])
            B = rnd.choice([1, 2, 3, 5, 8, 17]); S = rnd.randint(3, 12)
            N = rnd.choice([1, 7, 16, 30, 49, 100, 113, 196, 209, 250]); d = rnd.choice([128, 256])
            p = rnd.randint(1, 4); train = rnd.random() < 0.7
            b = mode_fuzz.run(0, name, B, S, N, d, p, train, case)
            for mode in (2, 1):
                a = mode_fuzz.run(mode, name, B, S, N, d, p, train, case)
                for k in a:
                    den = float(b[k].abs().max()) + 1e-20
                    if den > 1e-6:
                        assert float((a[k] - b[k]).abs().max()) / den < 2e-4, (mode, case, name, B, S, N, d, p, train, k)
    finally:
        macx._lib.lib().macx_gemm_mode(default_gemm_mode())


@pytest.mark.parametrize("seed", range(int(os.environ.get("MACX_FUZZ_SEEDS", "6"))))      # more seeds for a bug hunt
def test_fused_cell_matches_the_oracle_on_random_shapes_and_options(macx, dev, seed):
    """The fused kernels against the fp64 oracle -- final state, every parameter gradient, dKB / dwords / dvecQ -- on random
    draws from the option surface they cover (flag file, initial states, activations, gate / self-attention / recurrent
    control switches, variational or plain memory dropout, any keep values) and random shapes (odd batches, N from one
    cell to beyond a row tile, d in {128, 256}, 1-5 steps)."""
    import torch
    from helpers import make_case, oracle_run, rel_err, relu_boundary
    from test_gpu_cell import build_cell, FWD_TOL, GRAD_TOL
    rnd = random.Random(7000 + seed)
    ran = 0
    for case in range(6):
        name = rnd.choice(["args", "args1", "args3", "args4"])
        over = {}
        if rnd.random() < 0.5:
            over["initCtrl"] = rnd.choice(["PRM", "ZERO", "Q"])
        if rnd.random() < 0.5:
            over["initMem"] = rnd.choice(["PRM", "ZERO", "Q"])
        if rnd.random() < 0.4:
            over["controlInputUnshared"] = rnd.random() < 0.5
        if rnd.random() < 0.4:
            over["controlInputAct"] = rnd.choice(["NON", "RELU", "TANH"])
        if rnd.random() < 0.5:
            over["relu"] = rnd.choice(["STD", "ELU"])
        if rnd.random() < 0.4:
            over["readMemAct"] = rnd.choice(["RELU", "TANH"])
        if rnd.random() < 0.4:
            over["readCtrlAct"] = rnd.choice(["NON", "RELU", "TANH"])
        if rnd.random() < 0.4:
            over["writeMemAct"] = rnd.choice(["NON", "RELU", "TANH"])
        if rnd.random() < 0.3:
            over["writeGate"] = True
            over["writeGateBias"] = rnd.choice([0.0, 1.0, -0.5])
        if rnd.random() < 0.3:
            over["writeSelfAtt"] = True
            over["writeSelfAttMod"] = rnd.choice(["NON", "CONT"])
        if rnd.random() < 0.3:
            over["memoryVariationalDropout"] = rnd.random() < 0.5
        over["memoryDropout"] = rnd.choice([1.0, 0.85, 0.6])
        over["readDropout"] = rnd.choice([1.0, 0.85, 0.5])
        over["writeDropout"] = rnd.choice([1.0, 1.0, 0.9])
        B = rnd.choice([1, 2, 3, 5, 9]); S = rnd.randint(3, 14)
        N = rnd.choice([1, 7, 30, 49, 100, 196, 209]); d = rnd.choice([128, 128, 256]); p = rnd.randint(1, 5)
        train = rnd.random() < 0.7
        cfg, vq, words, lengths, kb = make_case(name, B, S, N, d, p, **over)
        try:
            macx.options.freeze(cfg)
        except macx.UnsupportedOptions:
            continue                                      # (lands on the generic path: tests/test_gpu_generic.py)
        cell, params, (vqd, wd, kbd) = build_cell(macx, dev, cfg, vq, words, lengths, kb, train, seed=5 + case, requires_grad=True)
        assert type(cell) is macx.MACCell
        g = torch.Generator().manual_seed(9)
        dmem, dctl = torch.randn(B, d, generator=g) / B, torch.randn(B, d, generator=g) / B
        state = cell.run()
        ((state.memory * dmem.to(dev)).sum() + (state.control * dctl.to(dev)).sum()).backward()
        torch.cuda.synchronize()
        what = (seed, case, name, over, B, S, N, d, p, train)
        # plain ReLU has a derivative jump at 0: a pre-activation within fp32 round-off of zero lands on the other side in fp32 than
        # in the fp64 oracle and moves ONE (row, column) of dI1 by its whole contribution (~5e-4 of the largest entry of dW1 /
        # db1 / dWx / dKB for one element; tests/case_probe.py) -- and a [B,N,d] pre-activation tensor has dozens of entries
        # within 1e-6 of its largest magnitude around zero.  Under --relu STD the oracle therefore also runs with derivative 0
        # and with derivative 1 at |pre-activation| <= 1e-6 max|pre-activation| (oracle._ReluAtBoundary): the distance between
        # those two gradients is what the undecidable elements can contribute, and it is added to the tolerance of that tensor.
        # Where no pre-activation is that close to the jump the two coincide and the tolerance is the usual one.
        ref = oracle_run(cfg, params.to_reference_dict(), vq, words, lengths, kb, train=train, seed=5 + case, need_grad=True,
                         d_memory=dmem, d_control=dctl)
        lohi = []
        if cfg.relu == "STD":
            for mode in (0, 1):
                with relu_boundary(mode, 1e-6):
                    lohi.append(oracle_run(cfg, params.to_reference_dict(), vq, words, lengths, kb, train=train, seed=5 + case,
                                           need_grad=True, d_memory=dmem, d_control=dctl))

        def undecidable(pick):
            """relative size of what the masked elements contribute to the gradient `pick` selects from a run"""
            if not lohi:
                return 0.0
            a, b = pick(lohi[0]), pick(lohi[1])
            return 0.0 if a is None else rel_err(a, b)
        gtol = GRAD_TOL
        assert rel_err(state.memory, ref["memory"]) < FWD_TOL and rel_err(state.control, ref["control"]) < FWD_TOL, what
        for k, (got, nm) in enumerate(((vqd, "vecQ"), (wd, "words"), (kbd, "kb"))):
            want = ref["inputs"][k]
            if want.grad is not None and float(want.grad.abs().max()) > 1e-9:
                assert rel_err(got.grad, want.grad) < gtol + undecidable(lambda r: r["inputs"][k].grad), (what, nm)
        names = macx.params.reference_names(cfg, p)
        for f in params.fields:
            gt = getattr(params, f).grad
            for refname, idx in names[f]:
                rg = ref["params"][refname].grad
                got = gt if idx is None else gt[idx]
                if rg is None:
                    assert float(got.abs().max()) == 0.0, (what, refname)
                    continue
                if float(rg.abs().max()) < 1e-9:
                    # analytically zero (a bias that shifts every logit of a softmax alike, e.g. memKbProj_2's under
                    # readCtrlAct = NON): the fp64 oracle leaves 1e-18, fp32 kernels 1e-9 -- compare absolutely
                    assert float(got.abs().max()) < 1e-6, (what, refname)
                    continue
                floor = 5e-2 if refname.endswith("linearLayerlogits/biases/bias") else 1e-6
                assert rel_err(got.reshape(rg.shape), rg, floor=floor) < gtol + undecidable(lambda r: r["params"][refname].grad), (what, refname)
        ran += 1
    assert ran >= 3


@pytest.mark.parametrize("seed", range(int(os.environ.get("MACX_FUZZ_SEEDS", "3"))))
def test_stem_encoder_and_classifier_on_random_shapes(macx, dev, seed):
    """The three modules around the cell against their oracles on random shapes: odd image grids down to 1 x 2, one image,
    any multiple-of-128 widths; one-word questions, one question, embedding widths that are not multiples of anything,
    vocabularies of one word; classifier widths down to 16 and answer counts that are not multiples of anything."""
    import torch
    from helpers import rel_err, max_abs
    import test_gpu_stem as TS
    import test_gpu_encoder as TE
    import test_gpu_output as TO
    from oracle import mac_oracle as mo
    rnd = random.Random(9000 + seed)
    for case in range(4):
        # ---- stem
        B, H, W = rnd.choice([1, 2, 3, 5]), rnd.randint(1, 9), rnd.randint(2, 9)
        Cin, Cmid, Cout = [rnd.choice([128, 256, 384]) for _ in range(3)]
        train = rnd.random() < 0.6
        stem, kb, ref, prm = TS.run_case(macx, dev, B, H, W, Cin, Cmid, Cout, train, b0=rnd.randint(0, 3))
        what = ("stem", seed, case, B, H, W, Cin, Cmid, Cout, train)
        assert rel_err(kb, ref) < 1e-5, what
        for f, name in macx.stem.REF_NAMES.items():
            assert rel_err(getattr(stem, f).grad, prm[name].grad, floor=1e-7) < 2e-4, (what, f)
        # ---- question encoder
        B, S, V, E, h = rnd.choice([1, 2, 4, 7]), rnd.randint(1, 12), rnd.randint(1, 40), rnd.choice([4, 7, 50, 300]), rnd.choice([128, 256])
        train = rnd.random() < 0.6
        enc, words, vecQ, rw, rq, prm, lengths = TE.run_case(macx, dev, B, S, V, E, h, train, b0=rnd.randint(0, 3))
        what = ("encoder", seed, case, B, S, V, E, h, train)
        assert rel_err(words, rw) < 1e-5 and rel_err(vecQ, rq) < 1e-5, what
        for f, name in macx.encoder.REF_NAMES.items():
            assert rel_err(getattr(enc, f).grad, prm[name].grad, floor=1e-7) < 2e-4, (what, f)
        # ---- output unit + classifier (fused)
        B, d, Hd, A = rnd.choice([1, 3, 8, 64]), rnd.choice([64, 128, 256]), rnd.choice([16, 48, 128, 512]), rnd.choice([2, 7, 28, 33])
        train = rnd.random() < 0.6
        cfg = mo.flag_file_config("args", memDim=d, ctrlDim=d, attDim=d, outClassifierDims=[Hd], answerWordsNum=A)
        out = macx.OutputClassifier(cfg, generator=torch.Generator().manual_seed(1)).to(dev)
        assert type(out) is macx.OutputClassifier
        g = torch.Generator().manual_seed(2 + case)
        mem, vq, dl = torch.randn(B, d, generator=g), torch.rand(B, d, generator=g) * 2 - 1, torch.randn(B, A, generator=g)
        memd, vqd = mem.to(dev).requires_grad_(True), vq.to(dev).requires_grad_(True)
        b0 = rnd.randint(0, 5)
        logits = out(memd, vqd, train=train, seed=9, b0=b0)
        (logits * dl.to(dev)).sum().backward()
        torch.cuda.synchronize()
        keep = cfg.outputDropout if train else 1.0
        mr, vr = mem.double().requires_grad_(True), vq.double().requires_grad_(True)
        refl, prm = TO.oracle_logits(cfg, out.to_reference_dict(), mr, vr, keep, 9, b0=b0, need_grad=True)
        (refl * dl.double()).sum().backward()
        what = ("classifier", seed, case, B, d, Hd, A, train)
        assert max_abs(logits, refl) < 2e-5, what
        assert rel_err(memd.grad, mr.grad) < 1e-4 and rel_err(vqd.grad, vr.grad) < 1e-4, what
        for f, name in macx.output.REF_NAMES.items():
            assert rel_err(getattr(out, f).grad, prm[name].grad, floor=1e-7) < 1e-4, (what, f)
