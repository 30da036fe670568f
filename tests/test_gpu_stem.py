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


def test_stem_clevr_shape_fp32(macx, dev):
    """config.imageDims = 14 x 14 x 1024 -> 512 -> 512 at B = 8 against the fp32 conv2d restatement."""
    stem, kb, ref, prm = run_case(macx, dev, 8, 14, 14, 1024, 512, 512, True, dtype=torch.float32)
    assert rel_err(kb, ref) < 1e-4
    for f, name in macx.stem.REF_NAMES.items():
        assert rel_err(getattr(stem, f).grad, prm[name].grad) < 1e-3, f
