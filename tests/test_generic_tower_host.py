"""CPU: macx.MACNet on an option set of the GENERIC path (the reference's default configuration: forward-only encoder, default
read / write structure, no --outQuestion) -- the module picks the generic encoder, cell and classifier, their variables
follow `.to(device)`, and ids -> logits -> loss matches the oracle chain.  Kernel calls are the torch stand-ins of
tests/test_generic_host.py and the (fused-only) stem is replaced by the oracle's conv restatement: the subject is the tower
plumbing, the kernels are checked on the GPU."""
import pytest
import torch

from oracle import mac_oracle as mo
from helpers import rel_err, max_abs
from test_generic_host import host_generic  # noqa: F401  (fixture)
from test_generic_encoder_host import questions


class _StemStandIn(torch.nn.Module):
    def __init__(self, cfg, params, H, W):
        super().__init__()
        self.cfg, self.H, self.W = cfg, H, W
        self.p = torch.nn.ParameterDict({"k%d" % i: torch.nn.Parameter(v.clone()) for i, v in enumerate(params.values())})
        self.names = list(params)
        self.keep = 1.0

    def tensors(self):
        return list(self.p.values())

    def to_reference_dict(self):
        return {n: v.detach().clone() for n, v in zip(self.names, self.p.values())}

    def forward(self, images, train=False, seed=None, b0=0):
        vs = mo.VarStore(params=dict(zip(self.names, self.p.values())), dtype=torch.float32)
        return mo.stem_cnn(self.cfg, vs, images, self.H, self.W)


@pytest.mark.parametrize("contextual", [True, False])
def test_tower_on_the_reference_default_configuration(macx, host_generic, contextual):
    run_tower(macx, contextual)


def run_tower(macx, contextual, dev=None):
    """dev=None: CPU stand-ins (host logic, stem replaced); a device: everything on the HIP kernels, the real stem included."""
    B, H, W, d, p, S, A, V = 3, 3, 2, 128, 2, 5, 6, 9
    Cin = 16 if dev is None else 128
    E = 7 if contextual else d      # without --controlContextual the cell attends over the raw embeddings: wrdEmbDim == ctrlDim needed
    cfg = mo.default_config(netLength=p, memDim=d, ctrlDim=d, attDim=d, encDim=d, wrdEmbDim=E, outClassifierDims=[128], answerWordsNum=A,
                            controlContextual=contextual, relu="ELU" if dev is not None else "STD")
    cfg.stemDim = 16 if dev is None else 128
    q, lengths = questions(B, S, V, 3)
    g = torch.Generator().manual_seed(2)
    img = torch.relu(torch.randn(B, H * W, Cin, generator=g))
    # the oracle chain creates the variables (reference names, reference order)
    vs0 = mo.VarStore(generator=torch.Generator().manual_seed(7))
    words, vq = mo.question_encoder(cfg, vs0, q, lengths, V)
    kb = mo.stem_cnn(cfg, vs0, img, H, W)
    raw_of = lambda store: torch.cat([torch.zeros(1, E, dtype=store.dtype), store.params["qEmbeddings/emb"]], dim=0)[q.long()]
    c, m, _ = mo.mac_network(cfg, vs0, vq, raw_of(vs0), words, lengths, kb)
    mo.output_classifier(cfg, vs0, m, vq)
    params = {k: v.clone() for k, v in vs0.params.items()}

    if dev is None:
        stem = _StemStandIn(cfg, {k: v for k, v in params.items() if k.startswith("stem/")}, H, W)
        real_stem = macx.model.Stem
        macx.model.Stem = lambda *a, **k: stem                     # the constructor's own choices, without the HIP stem
        try:
            built = macx.MACNet(cfg, vocab=V, H=H, W=W, imageInDim=Cin, answerWordsNum=A)
        finally:
            macx.model.Stem = real_stem
    else:
        built = macx.MACNet(cfg, vocab=V, H=H, W=W, imageInDim=Cin, answerWordsNum=A)
        macx.checkpoint.load_reference(built.stem, {k: v for k, v in params.items() if k.startswith("stem/")})
    assert type(built.enc) is macx.GenericQuestionEncoder and type(built.out) is macx.GenericOutputClassifier
    # the cell's plan is compiled at construction: its variables exist, under the reference's names and in its creation order
    assert type(built.cell) is macx.GenericParams
    assert list(built.cell.names) == [k for k in params if k.startswith("MACnetwork/")]
    built.enc.load_reference_dict(params)
    built.out.load_reference_dict(params)
    built.cell.load_reference_dict({k: v for k, v in params.items() if k.startswith("MACnetwork/")})
    target = torch.device("cpu") if dev is None else dev
    built = built.to(target)
    assert built.cell.device == target
    to = (lambda t: t.to(dev)) if dev is not None else (lambda t: t)
    logits = built(to(img), to(q), to(lengths), train=False)
    prm = {k: v.double().clone().requires_grad_(True) for k, v in params.items()}
    vs = mo.VarStore(params=prm, dtype=torch.float64)
    w2, v2 = mo.question_encoder(cfg, vs, q, lengths, V)
    kb2 = mo.stem_cnn(cfg, vs, img.double(), H, W)
    c2, m2, _ = mo.mac_network(cfg, vs, v2, raw_of(vs), w2, lengths, kb2)
    rl = mo.output_classifier(cfg, vs, m2, v2)
    assert max_abs(logits, rl) < 5e-5
    dl = torch.randn(logits.shape, generator=g)
    (logits * to(dl)).sum().backward()
    (rl * dl.double()).sum().backward()
    seen = 0
    for mod in (built.enc.params, built.cell, built.out.params):
        for k, gr in mod.grads_by_name().items():
            if prm[k].grad is not None and float(prm[k].grad.abs().max()) > 1e-9:
                assert gr is not None and rel_err(gr, prm[k].grad, floor=1e-7) < 5e-3, k        # (plain ReLU: derivative jumps)
                seen += 1
    assert seen >= 6
    # every variable the reference would have created exists, under its name
    have = set(built.enc.to_reference_dict()) | set(built.cell.to_reference_dict()) | set(built.out.to_reference_dict()) | set(built.stem.to_reference_dict())
    assert have == set(params)
