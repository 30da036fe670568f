"""CPU: host logic of macx.GenericOutputClassifier (output unit + classifier for the option sets the fused kernels refuse)
with the kernel-call functions swapped for torch restatements (see tests/test_generic_host.py): variable names and order,
op chaining, dropout sites, the padded last layer, the reference's exceptions.  The kernels run in tests/test_gpu_output.py."""
import pytest
import torch

from oracle import dropout_hash as dh
from oracle import mac_oracle as mo
from helpers import rel_err, max_abs
from test_generic_host import host_generic  # noqa: F401  (fixture)

OUT_VARIANTS = {
    "no_question": dict(outQuestion=False),
    "question_mul": dict(outQuestion=True, outQuestionMul=True),
    "deep": dict(outQuestion=True, outClassifierDims=[256, 128]),
    "no_hidden": dict(outQuestion=True, outClassifierDims=[]),
    "prelu": dict(outQuestion=True, relu="PRM"),
    "std_relu_mul_deep": dict(outQuestion=True, outQuestionMul=True, relu="STD", outClassifierDims=[128, 128, 128]),
}


def out_cfg(variant, d=128, A=7):
    kw = dict(memDim=d, ctrlDim=d, attDim=d, answerWordsNum=A, outClassifierDims=[128])
    kw.update(OUT_VARIANTS[variant])
    return mo.flag_file_config("args", **kw)


def oracle_out(cfg, ref_params, memory, vq, keep, seed, b0, site_of):
    prm = {k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in ref_params.items()}
    vs = mo.VarStore(params=prm, dtype=torch.float64)
    B, d = memory.shape
    dim = d * ((3 if cfg.outQuestionMul else 2) if cfg.outQuestion else 1)
    dims = [dim] + list(cfg.outClassifierDims)
    masks = None
    if keep < 1.0:
        masks = [torch.from_numpy(dh.mask_for(seed, site_of(i), 0, keep, (B, w), b0=b0)).double() for i, w in enumerate(dims)]
    return mo.output_classifier(cfg, vs, memory, vq, output_keep=keep, masks=masks), prm


def reference_named_params(cfg, B, d, seed=4):
    """The variables the oracle creates for this output unit, biases / slopes perturbed."""
    vs = mo.VarStore(generator=torch.Generator().manual_seed(seed))
    mo.output_classifier(cfg, vs, torch.zeros(B, d), torch.zeros(B, d))
    g = torch.Generator().manual_seed(seed + 1)
    for k, v in vs.params.items():
        if "/biases/" in k or k.endswith("alpha"):
            v.add_((torch.rand(v.shape, generator=g) - 0.5) * 0.2)
    return {k: v.clone() for k, v in vs.params.items()}


def run_pair(macx, cfg, train, dev=None):
    B, d, A = 5, cfg.memDim, cfg.answerWordsNum
    params = reference_named_params(cfg, B, d)
    out = macx.OutputClassifier(cfg, generator=torch.Generator().manual_seed(1))
    assert type(out) is macx.GenericOutputClassifier
    assert list(out.params.names) == list(params), "variables under the reference's names in the reference's order"
    out.load_reference_dict(params)
    if dev is not None:
        out = out.to(dev)
    g = torch.Generator().manual_seed(2)
    mem, vq, dl = torch.randn(B, d, generator=g), torch.rand(B, d, generator=g) * 2 - 1, torch.randn(B, A, generator=g)
    to = (lambda t: t.to(dev)) if dev is not None else (lambda t: t.clone())
    memd, vqd = to(mem).requires_grad_(True), to(vq).requires_grad_(True)
    logits = out(memd, vqd, train=train, seed=9, b0=3)
    (logits * to(dl)).sum().backward()
    keep = cfg.outputDropout if train else 1.0
    mr, vr = mem.double().requires_grad_(True), vq.double().requires_grad_(True)
    ref, prm = oracle_out(cfg, params, mr, vr, keep, 9, 3, macx.output.fc_site)
    (ref * dl.double()).sum().backward()
    return out, logits, ref, (memd, mr), (vqd, vr), prm


def check(out, logits, ref, mem, vq, prm, gtol):
    assert logits.shape == ref.shape and max_abs(logits, ref) < 2e-5
    assert rel_err(mem[0].grad, mem[1].grad) < gtol
    if vq[1].grad is not None:
        assert rel_err(vq[0].grad, vq[1].grad) < gtol
    else:
        assert vq[0].grad is None or float(vq[0].grad.abs().max()) == 0.0
    grads = out.params.grads_by_name()
    for k, v in prm.items():
        assert grads[k] is not None and rel_err(grads[k], v.grad) < gtol, k


@pytest.mark.parametrize("variant", sorted(OUT_VARIANTS))
@pytest.mark.parametrize("train", [False, True])
def test_generic_output_host_logic_matches_oracle(macx, host_generic, variant, train):
    check(*run_pair(macx, out_cfg(variant), train), gtol=1e-5)


def test_generic_output_dispatch_and_rejections(macx, host_generic):
    assert type(macx.OutputClassifier(out_cfg("question_mul"))) is macx.GenericOutputClassifier
    fused = mo.flag_file_config("args", memDim=128, ctrlDim=128, attDim=128, answerWordsNum=7, outClassifierDims=[64])
    assert type(macx.OutputClassifier(fused)) is macx.OutputClassifier
    def with_flag(**kw):
        cfg = mo.flag_file_config("args")
        for k, v in kw.items():
            setattr(cfg, k, v)
        return cfg
    with pytest.raises(TypeError, match="outputDim"):                 # model.py:521 vs ops.py:595
        macx.OutputClassifier(with_flag(outImage=True))
    with pytest.raises(NameError, match="outputDim"):                 # model.py:561
        macx.OutputClassifier(with_flag(answerMod="MUL"))
    with pytest.raises(UnboundLocalError):                            # ops.mul's DIAG branch never assigns `output`
        macx.OutputClassifier(with_flag(answerMod="DIAG"))
    with pytest.raises(macx.UnsupportedOptions):
        macx.OutputClassifier(with_flag(outputBN=True))
    # hidden widths off the kernels' 128-column granule are accepted on the generic path (zero-padded inside the products, round 4)
    out = macx.OutputClassifier(mo.flag_file_config("args", memDim=128, ctrlDim=128, outClassifierDims=[64, 64], answerWordsNum=7))
    assert type(out) is macx.GenericOutputClassifier
    assert out(torch.zeros(2, 128), torch.zeros(2, 128)).shape == (2, 7)


def test_generic_output_refuses_cpu_tensors(macx):
    out = macx.OutputClassifier(out_cfg("no_question"))
    with pytest.raises(RuntimeError, match="no CPU path"):
        out(torch.zeros(2, 128), torch.zeros(2, 128))
