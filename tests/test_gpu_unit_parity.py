"""-m gpu: per-UNIT parity: the control, read and write units called on their own -- caller-chosen control /
memory / info, forward and backward -- on the HIP op kernels (macx.GenericMACCell exposes the reference's unit methods
`control`, `read`, `write` with the reference's signatures, mac_cell.py:133, 209, 305) against the oracle's units on the
same parameters and dropout masks.  Forward <= 1e-5 relative, every gradient <= 2e-4."""
import pytest
import torch

from oracle import mac_oracle as mo
from helpers import rel_err
from test_gpu_generic import oracle_params, assert_grad

pytestmark = pytest.mark.gpu


def build(macx, dev, cfg, B, S, N, d, train, seed=5):
    vq, words, lengths, kb = mo.synthetic_inputs(B, S, N, d, seed=13)
    params = oracle_params(cfg, vq, words, lengths, kb)
    keeps = (cfg.memoryDropout, cfg.readDropout, cfg.writeDropout) if train else (1.0, 1.0, 1.0)
    # oracle cell (fp64) with gradients
    op = {k: v.double().clone().requires_grad_(True) for k, v in params.items()}
    vs = mo.VarStore(params=op, dtype=torch.float64)
    ocell = mo.MACCellOracle(cfg, vs, vq.double(), words.double(), words.double(), lengths, kb.double(), keeps[0], keeps[1], keeps[2], B, train,
                             mask_fn=mo.hash_mask_fn(seed, keeps, b0=0) if train else None)
    with vs.scope("MACnetwork"):
        ocell.zero_state(B)
    gp = macx.GenericParams(device=dev).load_reference_dict(params)
    hcell = macx.GenericMACCell(vq.to(dev), words.to(dev), words.to(dev), lengths.to(dev), kb.to(dev), cfg.memoryDropout,
                                cfg.readDropout, cfg.writeDropout, B, train, config=cfg, params=gp, seed=seed, b0=0)
    hcell.zero_state(B)
    return ocell, vs, op, hcell, gp, (vq, words, lengths, kb)


def rand(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def check_grads(gp, op, prefix):
    grads = gp.grads_by_name()
    seen = 0
    for k, v in op.items():
        if v.grad is not None:
            assert grads[k] is not None, k
            assert_grad(grads[k], v.grad, k)
            seen += 1
    assert seen > 0, prefix


@pytest.mark.parametrize("name,over", [("args", {}), ("default", {}), ("args", dict(readMemAttType="BL", readCtrlAttType="ADD"))])
@pytest.mark.parametrize("train", [False, True])
def test_read_unit(macx, dev, name, over, train):
    B, S, N, d = 3, 6, 30, 128
    kw = dict(netLength=2, memDim=d, ctrlDim=d, attDim=d, **over)
    cfg = mo.flag_file_config("args", **kw) if name == "args" else mo.default_config(**kw)
    ocell, vs, op, hcell, gp, (vq, words, lengths, kb) = build(macx, dev, cfg, B, S, N, d, train)
    mem, ctl, w = rand((B, d), 1), rand((B, d), 2), rand((B, d), 3)
    mo_, co_, kbo = mem.double().requires_grad_(True), ctl.double().requires_grad_(True), kb.double().requires_grad_(True)
    with vs.scope("MACnetwork"), vs.scope("MACCell"):
        info_ref = ocell.read(kbo, mo_, co_)
    (info_ref * w.double()).sum().backward()
    mh, ch, kbh = [t.to(dev).requires_grad_(True) for t in (mem, ctl, kb)]
    info = hcell.read(kbh, mh, ch)
    (info * w.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert rel_err(info, info_ref) < 1e-5
    assert float((hcell.attentions["kb"][-1].cpu().double() - ocell.attentions["kb"][-1]).abs().max()) < 2e-6
    for got, want, nm in ((mh, mo_, "memory"), (ch, co_, "control"), (kbh, kbo, "knowledgeBase")):
        if want.grad is not None:
            assert rel_err(got.grad, want.grad) < 2e-4, nm
    check_grads(gp, op, "read")


@pytest.mark.parametrize("over", [{}, dict(writeSelfAtt=True, writeGate=True), dict(writeInputs="SUM"), dict(writeConcatMul=True, writeMemAct="RELU")])
def test_write_unit(macx, dev, over):
    B, S, N, d = 3, 6, 10, 128
    cfg = mo.flag_file_config("args", netLength=2, memDim=d, ctrlDim=d, attDim=d, **over)
    ocell, vs, op, hcell, gp, _ = build(macx, dev, cfg, B, S, N, d, False)
    mem, info, ctl, w = rand((B, d), 1), rand((B, d), 2), rand((B, d), 3), rand((B, d), 4)
    a = [t.double().requires_grad_(True) for t in (mem, info, ctl)]
    with vs.scope("MACnetwork"), vs.scope("MACCell"):
        ref = ocell.write(a[0], a[1], a[2], a[2])
    (ref * w.double()).sum().backward()
    h = [t.to(dev).requires_grad_(True) for t in (mem, info, ctl)]
    out = hcell.write(h[0], h[1], h[2], h[2])
    (out * w.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert rel_err(out, ref) < 1e-5
    for got, want in zip(h, a):
        if want.grad is not None:
            assert rel_err(got.grad, want.grad) < 2e-4
    check_grads(gp, op, "write")


@pytest.mark.parametrize("name,over", [("args", {}), ("args1", {}), ("args", dict(controlProj=True, controlConcatWords=True))])
def test_control_unit(macx, dev, name, over):
    B, S, N, d = 3, 7, 10, 128
    cfg = mo.flag_file_config(name, netLength=2, memDim=d, ctrlDim=d, attDim=d, **over)
    ocell, vs, op, hcell, gp, (vq, words, lengths, kb) = build(macx, dev, cfg, B, S, N, d, False)
    cin, ctl, cc, w = rand((B, d), 1), rand((B, d), 2), rand((B, d), 3), rand((B, d), 4)
    a = [t.double().requires_grad_(True) for t in (cin, ctl, cc)]
    with vs.scope("MACnetwork"), vs.scope("MACCell"):
        ref, ref_cc = ocell.control(a[0], ocell.inWords, ocell.outWords, lengths, a[1], a[2])
    ((ref + ref_cc) * w.double()).sum().backward()
    h = [t.to(dev).requires_grad_(True) for t in (cin, ctl, cc)]
    out, out_cc = hcell.control(h[0], hcell.inWords, hcell.outWords, hcell.questionLengths, h[1], h[2])
    ((out + out_cc) * w.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert rel_err(out, ref) < 1e-5 and rel_err(out_cc, ref_cc) < 1e-5
    assert float((hcell.attentions["question"][-1].cpu().double() - ocell.attentions["question"][-1]).abs().max()) < 2e-6
    for got, want in zip(h, a):
        if want.grad is not None:
            assert rel_err(got.grad, want.grad) < 2e-4
    check_grads(gp, op, "control")
