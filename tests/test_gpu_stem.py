"""-m gpu: stem CNN (model.py:165-204) -- implicit-GEMM conv forward / backward-data / kernel gradient
against the torch conv2d restatement in the oracle, identical dropout masks."""
import pytest
import torch

from oracle import dropout_hash as dh
from oracle import mac_oracle as mo
from helpers import rel_err, max_abs

pytestmark = pytest.mark.gpu


def run_case(macx, dev, B, H, W, Cin, Cmid, Cout, train, dtype=torch.float64, b0=0):
    cfg = mo.flag_file_config("args", memDim=Cout, ctrlDim=Cout, attDim=Cout)
    cfg.stemDim = Cmid
    stem = macx.Stem(cfg, H=H, W=W, inDim=Cin, generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        stem.bias0.copy_(torch.rand_like(stem.bias0) - 0.5)
        stem.bias1.copy_(torch.rand_like(stem.bias1) - 0.5)
    g = torch.Generator().manual_seed(2)
    img = torch.relu(torch.randn(B, H * W, Cin, generator=g))
    kb = stem(img.to(dev), train=train, seed=5, b0=b0)
    dkb = torch.randn(B, H * W, Cout, generator=g) / B
    (kb * dkb.to(dev)).sum().backward()
    torch.cuda.synchronize()
    keep = stem.keep if train else 1.0
    prm = {k: v.cpu().to(dtype).requires_grad_(True) for k, v in stem.to_reference_dict().items()}
    vs = mo.VarStore(params=prm, dtype=dtype)
    masks = None
    if train:
        masks = [torch.from_numpy(dh.mask_for(5, 9, 0, keep, (B, H * W, Cin), b0=b0)).to(dtype),
                 torch.from_numpy(dh.mask_for(5, 10, 0, keep, (B, H * W, Cmid), b0=b0)).to(dtype)]
    ref = mo.stem_cnn(cfg, vs, img.to(dtype), H, W, keep=keep, masks=masks)
    (ref * dkb.to(dtype)).sum().backward()
    return stem, kb, ref, prm


@pytest.mark.parametrize("B,H,W,Cin,Cmid,Cout,train", [
    (2, 14, 14, 256, 128, 128, False), (3, 14, 14, 128, 256, 128, True), (2, 7, 7, 256, 128, 128, True), (1, 5, 3, 128, 128, 128, False)])
def test_stem_matches_conv2d_oracle(macx, dev, B, H, W, Cin, Cmid, Cout, train):
    stem, kb, ref, prm = run_case(macx, dev, B, H, W, Cin, Cmid, Cout, train, b0=1)
    assert rel_err(kb, ref) < 1e-5
    for f, name in macx.stem.REF_NAMES.items():
        assert rel_err(getattr(stem, f).grad, prm[name].grad) < 2e-4, f


def test_stem_clevr_shape(macx, dev):
    """config.imageDims = 14 x 14 x 1024 -> 512 -> 512 (north_star's shape; both convolutions on kb_conv_chain_kernel, forward and
    backward-data) at B = 8 against the fp64 conv2d restatement: forward 2e-5, every gradient 2e-4 -- the bounds of the GQA-shape
    case (round 4 compared with an fp32 oracle at 1e-3)."""
    stem, kb, ref, prm = run_case(macx, dev, 8, 14, 14, 1024, 512, 512, True, dtype=torch.float64)
    assert rel_err(kb, ref) < 2e-5
    for f, name in macx.stem.REF_NAMES.items():
        assert rel_err(getattr(stem, f).grad, prm[name].grad) < 2e-4, f


@pytest.mark.parametrize("H,W,Cin", [(14, 14, 1024), (7, 7, 512)])
def test_stem_on_wide_dynamic_range(macx, dev, H, W, Cin):
    """The stem's products carry ONE exponent per operand TENSOR (macx_gemm3h.hip.h: image, hidden layer, both kernels).  Entries
    spanning 2^+-12 inside each of those tensors (independently per element): the output against fp64 per unit of
    conv(|a|, |w|) + |b| of the LAST layer, at most 1.5x what the native f32-MFMA kernels (macx_gemm_mode 0) leave on the same data.
    What the format does NOT promise is stated in DESIGN 11: a pixel whose channels all sit 2^-k below the tensor's largest entry
    keeps 22 - k bits -- the per-element spread of this test is not that case."""
    L = macx._lib.lib()
    B, Cmid, Cout = 4, 512, 512
    cfg = mo.flag_file_config("args", memDim=Cout, ctrlDim=Cout, attDim=Cout)
    cfg.stemDim = Cmid
    g = torch.Generator().manual_seed(8)
    wide = lambda shape, span: torch.randn(shape, generator=g) * torch.exp2((torch.rand(shape, generator=g) * 2 - 1) * span)
    stem = macx.Stem(cfg, H=H, W=W, inDim=Cin, generator=torch.Generator().manual_seed(1)).to(dev)
    with torch.no_grad():
        stem.kernel0.copy_(wide(stem.kernel0.shape, 12.0) * (2.0 / (9 * Cin)) ** 0.5 / 64)
        stem.kernel1.copy_(wide(stem.kernel1.shape, 12.0) * (2.0 / (9 * Cmid)) ** 0.5 / 64)
        stem.bias0.copy_(torch.rand(Cmid, generator=g) - 0.5)
        stem.bias1.copy_(torch.rand(Cout, generator=g) - 0.5)
    img = torch.relu(wide((B, H * W, Cin), 12.0))
    imgd = img.to(dev)

    def fwd():
        with torch.no_grad():
            out = stem(imgd, train=False)
        torch.cuda.synchronize()
        return out.cpu().double()

    got = fwd()
    try:
        L.macx_gemm_mode(0)
        native = fwd()
    finally:
        from helpers import default_gemm_mode
        L.macx_gemm_mode(default_gemm_mode())
    # fp64 reference and the last layer's sum |a| |w| + |b|
    k0, k1 = stem.kernel0.detach().cpu().double(), stem.kernel1.detach().cpu().double()
    b0_, b1_ = stem.bias0.detach().cpu().double(), stem.bias1.detach().cpu().double()
    conv = lambda x, k: torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), k.permute(3, 2, 0, 1).contiguous(), padding=1).permute(0, 2, 3, 1)
    x0 = img.double().reshape(B, H, W, Cin)
    h = torch.nn.functional.elu(conv(x0, k0) + b0_)
    ref = torch.nn.functional.elu(conv(h, k1) + b1_).reshape(B, H * W, Cout)
    scale = (conv(h.abs(), k1.abs()) + b1_.abs()).reshape(B, H * W, Cout) + 1e-300
    e, e0 = (got - ref).abs() / scale, (native - ref).abs() / scale
    assert float(e.max()) < 2e-6, float(e.max())
    assert float(e.max()) <= 1.5 * float(e0.max()) + 1e-9 and float(e.mean()) <= 1.5 * float(e0.mean()) + 1e-10, \
        (float(e.max()), float(e0.max()), float(e.mean()), float(e0.mean()))


def test_stem_gqa_shape(macx, dev):
    """BASELINE configs[4]: GQA-shape features 7 x 7 x 2048 -> 512 -> 512 (model.py:791, config.py:433), training mode, B = 8:
    the K loop of the first convolution walks 9 taps x 2048 channels = 18432."""
    stem, kb, ref, prm = run_case(macx, dev, 8, 7, 7, 2048, 512, 512, True, dtype=torch.float64)
    assert rel_err(kb, ref) < 2e-5
    for f, name in macx.stem.REF_NAMES.items():
        assert rel_err(getattr(stem, f).grad, prm[name].grad) < 2e-4, f


def test_core_graph_stem_cell_classifier_gradients(macx, dev):
    """images -> stem -> MAC cell x p -> classifier -> CE: logits and every parameter gradient against the
    oracle chain (stem_cnn -> mac_network -> output_classifier), identical dropout masks."""
    from oracle import mac_oracle as mo
    B, H, W, Cin, d, p, S, A = 3, 5, 4, 128, 128, 2, 6, 7
    cfg = mo.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d, outClassifierDims=[32], answerWordsNum=A)
    cfg.stemDim = 128
    net = macx.MACNetCore(cfg, H=H, W=W, imageInDim=Cin, answerWordsNum=A, generator=torch.Generator().manual_seed(4)).to(dev)
    g = torch.Generator().manual_seed(6)
    img = torch.relu(torch.randn(B, H * W, Cin, generator=g))
    vq, words, lengths, _ = mo.synthetic_inputs(B, S, 1, d, seed=8)
    ans = torch.tensor([1, 5, 2])
    logits = net(img.to(dev), vq.to(dev), words.to(dev), lengths.to(dev), train=True, seed=21)
    loss, pred = net.loss_and_pred(logits, ans.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    # oracle chain in fp64
    dt = torch.float64
    prm = {}
    for src in (net.stem.to_reference_dict(), net.cell.to_reference_dict(), net.out.to_reference_dict()):
        prm.update({k: v.cpu().to(dt).requires_grad_(True) for k, v in src.items()})
    vs = mo.VarStore(params=prm, dtype=dt)
    keeps = (cfg.memoryDropout, cfg.readDropout, cfg.writeDropout)
    sk, ok = net.stem.keep, net.out.keep
    smasks = [torch.from_numpy(dh.mask_for(21, 9, 0, sk, (B, H * W, Cin))).to(dt), torch.from_numpy(dh.mask_for(21, 10, 0, sk, (B, H * W, 128))).to(dt)]
    kb = mo.stem_cnn(cfg, vs, img.to(dt), H, W, keep=sk, masks=smasks)
    c, m, _ = mo.mac_network(cfg, vs, vq.to(dt), words.to(dt), words.to(dt), lengths, kb, train=True, mask_fn=mo.hash_mask_fn(21, keeps), keeps=keeps)
    omasks = [torch.from_numpy(dh.mask_for(21, 7, 0, ok, (B, 2 * d))).to(dt), torch.from_numpy(dh.mask_for(21, 8, 0, ok, (B, 32))).to(dt)]
    rl = mo.output_classifier(cfg, vs, m, vq.to(dt), output_keep=ok, masks=omasks)
    rloss, rpred = mo.answer_loss_and_pred(rl, ans)
    rloss.backward()
    assert max_abs(logits, rl) < 5e-5 and torch.equal(pred.cpu(), rpred)
    assert abs(float(loss) - float(rloss)) < 1e-5
    names = {}
    names.update({f: [(n, None)] for f, n in macx.stem.REF_NAMES.items()})
    bad = {}
    for mod, refs in ((net.stem, {f: [(n, None)] for f, n in macx.stem.REF_NAMES.items()}),
                      (net.cell, macx.params.reference_names(cfg, p)),
                      (net.out, {f: [(n, None)] for f, n in macx.output.REF_NAMES.items()})):
        for f, lst in refs.items():
            if not hasattr(mod, f):
                continue
            for refname, idx in lst:
                rg = prm[refname].grad
                got = getattr(mod, f).grad
                got = got if idx is None else got[idx]
                floor = 5e-2 if refname.endswith("linearLayerlogits/biases/bias") else 1e-7
                e = rel_err(got.reshape(rg.shape), rg, floor=floor)
                if not e < 3e-4:
                    bad[refname] = e
    assert not bad, bad


def test_stem_accepts_feed_dict_layout(macx, dev):
    """h5 features are [B, C, H, W]; model.py:68 transposes them to NHWC.  macx_images_to_nhwc does that on the device:
    bit-identical to feeding the NHWC tensor."""
    cfg = mo.flag_file_config("args", memDim=128, ctrlDim=128, attDim=128)
    cfg.stemDim = 128
    stem = macx.Stem(cfg, H=14, W=14, inDim=256, generator=torch.Generator().manual_seed(1)).to(dev)
    nchw = torch.relu(torch.randn(3, 256, 14, 14, generator=torch.Generator().manual_seed(2))).to(dev)
    nhwc = nchw.permute(0, 2, 3, 1).contiguous()
    a = stem(nchw, train=True, seed=3)
    b = stem(nhwc, train=True, seed=3)
    c = stem(nhwc.reshape(3, 196, 256), train=True, seed=3)
    assert torch.equal(a, b) and torch.equal(a, c)
