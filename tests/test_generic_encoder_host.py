"""CPU: host logic of macx.GenericQuestionEncoder (the LSTM encoder configurations the fused kernels refuse: no --encBi, output
projections) with the kernel-call functions swapped for torch restatements (tests/test_generic_host.py): variable names and
order, the per-step op chain, sequence-length masking, the reverse-sequence gather, dropout sites.  The kernels run in
tests/test_gpu_encoder.py."""
import pytest
import torch

from oracle import dropout_hash as dh
from oracle import mac_oracle as mo
from helpers import rel_err
from test_generic_host import host_generic  # noqa: F401  (fixture)

ENC_VARIANTS = {
    "uni": dict(encBi=False, encDim=128, ctrlDim=128),
    "uni_projected": dict(encBi=False, encDim=128, ctrlDim=256),
    "bi_proj_tanh": dict(encBi=True, encDim=256, ctrlDim=256, encProj=True, encProjQAct="TANH"),
    "uni_proj_prelu": dict(encBi=False, encDim=128, ctrlDim=128, encProj=True, encProjQAct="RELU", relu="PRM"),
    "bi_other_width": dict(encBi=True, encDim=256, ctrlDim=128),
    "bi_off_granule": dict(encBi=True, encDim=144, ctrlDim=144),        # 72 units per direction: off the kernels' 128-column granule
    "uni_off_granule": dict(encBi=False, encDim=100, ctrlDim=200),
}


def enc_cfg(variant, E=7):
    kw = dict(wrdEmbDim=E, memDim=128, attDim=128)
    kw.update(ENC_VARIANTS[variant])
    return mo.default_config(**kw)


def questions(B, S, V, seed):
    g = torch.Generator().manual_seed(seed)
    lengths = torch.randint(1, S + 1, (B,), generator=g, dtype=torch.int32)
    lengths[0] = S
    if B > 1:
        lengths[1] = 1
    q = torch.randint(1, V + 1, (B, S), generator=g, dtype=torch.int32)
    return q * (torch.arange(S).unsqueeze(0) < lengths.unsqueeze(1)).to(torch.int32), lengths


def run_pair(macx, cfg, train, dev=None, B=4, S=6, V=9):
    q, lengths = questions(B, S, V, 5)
    vs0 = mo.VarStore(generator=torch.Generator().manual_seed(4))
    mo.question_encoder(cfg, vs0, q, lengths, V)                                   # the variables the oracle creates
    g = torch.Generator().manual_seed(6)
    for k, v in vs0.params.items():
        if k.endswith("/bias") or "/biases/" in k or k.endswith("alpha"):
            v.add_((torch.rand(v.shape, generator=g) - 0.5) * 0.2)
    params = {k: v.clone() for k, v in vs0.params.items()}
    enc = macx.QuestionEncoder(cfg, vocab=V, generator=torch.Generator().manual_seed(1))
    assert type(enc) is macx.GenericQuestionEncoder
    assert list(enc.params.names) == list(params), "variables under the reference's names in the reference's order"
    enc.load_reference_dict(params)
    if dev is not None:
        enc = enc.to(dev)
    to = (lambda t: t.to(dev)) if dev is not None else (lambda t: t)
    words, vecQ = enc(to(q), to(lengths), train=train, seed=9, b0=2)
    dW, dQ = torch.randn(words.shape, generator=g), torch.randn(vecQ.shape, generator=g)
    ((words * to(dW)).sum() + (vecQ * to(dQ)).sum()).backward()
    ki, kq = (cfg.encInputDropout, cfg.qDropout) if train else (1.0, 1.0)
    prm = {k: v.double().clone().requires_grad_(True) for k, v in params.items()}
    vs = mo.VarStore(params=prm, dtype=torch.float64)
    masks = None
    if train:
        masks = [torch.from_numpy(dh.mask_for(9, 11, 0, ki, (B, S, cfg.wrdEmbDim), b0=2)).double(),
                 torch.from_numpy(dh.mask_for(9, 12, 0, kq, (B, cfg.encDim), b0=2)).double()]
    rw, rq = mo.question_encoder(cfg, vs, q, lengths, V, keep_input=ki, keep_question=kq, masks=masks)
    ((rw * dW.double()).sum() + (rq * dQ.double()).sum()).backward()
    return enc, (words, rw), (vecQ, rq), prm, lengths


def check(enc, words, vecQ, prm, lengths, tol, gtol):
    assert words[0].shape == words[1].shape and rel_err(words[0], words[1]) < tol and rel_err(vecQ[0], vecQ[1]) < tol
    for b in range(words[0].shape[0]):          # dynamic_rnn emits zeros past the end (before any projection adds its bias)
        n = int(lengths[b])
        if n < words[0].shape[1] and not enc.proj:
            assert float(words[0][b, n:].detach().abs().max()) == 0.0
    grads = enc.params.grads_by_name()
    for k, v in prm.items():
        assert grads[k] is not None and rel_err(grads[k], v.grad, floor=1e-7) < gtol, k


@pytest.mark.parametrize("variant", sorted(ENC_VARIANTS))
@pytest.mark.parametrize("train", [False, True])
def test_generic_encoder_host_logic_matches_oracle(macx, host_generic, variant, train):
    enc, words, vecQ, prm, lengths = run_pair(macx, enc_cfg(variant), train)
    check(enc, words, vecQ, prm, lengths, tol=2e-6, gtol=2e-5)


def test_generic_encoder_dispatch_and_rejections(macx, host_generic):
    fused = mo.flag_file_config("args", encDim=256, ctrlDim=256, memDim=256, attDim=256, wrdEmbDim=8)
    assert type(macx.QuestionEncoder(fused, vocab=5)) is macx.QuestionEncoder
    assert type(macx.QuestionEncoder(mo.default_config(encDim=128, ctrlDim=128, wrdEmbDim=8), vocab=5)) is macx.GenericQuestionEncoder
    with pytest.raises(ValueError, match="already exists"):             # model.py:295-298: the layers collide on one scope
        macx.QuestionEncoder(mo.default_config(encDim=128, ctrlDim=128, encNumLayers=2), vocab=5)
    with pytest.raises(macx.UnsupportedOptions):
        macx.QuestionEncoder(mo.default_config(encType="GRU"), vocab=5)
    # widths off the kernels' 128-column granule: accepted on the generic path (round 4), also for the flag files' configuration
    assert type(macx.QuestionEncoder(mo.default_config(encDim=64, ctrlDim=64), vocab=5)) is macx.GenericQuestionEncoder
    assert type(macx.QuestionEncoder(mo.flag_file_config("args", encDim=144, ctrlDim=144, memDim=144, attDim=144, wrdEmbDim=8), vocab=5)) \
        is macx.GenericQuestionEncoder
    with pytest.raises(macx.UnsupportedOptions, match="multiple of 4"):
        macx.QuestionEncoder(mo.default_config(encDim=66, ctrlDim=66), vocab=5)
    fixed = macx.QuestionEncoder(mo.default_config(encDim=128, ctrlDim=128, wrdEmbDim=8, wrdEmbFixed=True), vocab=5)
    assert not fixed.params.table[fixed.params.names["qEmbeddings/emb"]].requires_grad


def test_generic_encoder_refuses_cpu_tensors(macx):
    enc = macx.QuestionEncoder(mo.default_config(encDim=128, ctrlDim=128, wrdEmbDim=8), vocab=5)
    with pytest.raises(RuntimeError, match="no CPU path"):
        enc(torch.ones(2, 3, dtype=torch.int32), torch.tensor([3, 1], dtype=torch.int32))
