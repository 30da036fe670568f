"""-m gpu: question input unit (model.py:207-219, 279-307; ops.biRNNLayer ops.py:859-911) -- embedding + biLSTM on the
HIP path against the op-for-op restatement of tf.nn.bidirectional_dynamic_rnn / BasicLSTMCell in the oracle, with
identical dropout masks, ragged question lengths and pad ids."""
import pytest
import torch

from oracle import dropout_hash as dh
from oracle import mac_oracle as mo
from helpers import rel_err, max_abs

pytestmark = pytest.mark.gpu


def make_questions(B, S, V, seed, min_len=1):
    g = torch.Generator().manual_seed(seed)
    lengths = torch.randint(min_len, S + 1, (B,), generator=g, dtype=torch.int32)
    lengths[0] = S
    if B > 1:
        lengths[1] = min_len
    q = torch.randint(1, V + 1, (B, S), generator=g, dtype=torch.int32)
    q = q * (torch.arange(S).unsqueeze(0) < lengths.unsqueeze(1)).to(torch.int32)      # 0 = pad (preprocess.py vectorize)
    return q, lengths


def run_case(macx, dev, B, S, V, E, h, train, dtype=torch.float64, b0=0, fixed=False):
    cfg = mo.flag_file_config("args", ctrlDim=2 * h, memDim=2 * h, attDim=2 * h, encDim=2 * h, wrdEmbDim=E, wrdEmbFixed=fixed)
    enc = macx.QuestionEncoder(cfg, vocab=V, generator=torch.Generator().manual_seed(3)).to(dev)
    gb = torch.Generator().manual_seed(7)
    with torch.no_grad():
        enc.fw_bias.copy_(torch.rand(4 * h, generator=gb) - 0.5)
        enc.bw_bias.copy_(torch.rand(4 * h, generator=gb) - 0.5)
    q, lengths = make_questions(B, S, V, seed=11)
    words, vecQ = enc(q.to(dev), lengths.to(dev), train=train, seed=9, b0=b0)
    g = torch.Generator().manual_seed(5)
    dW = torch.randn(B, S, 2 * h, generator=g) / B
    dQ = torch.randn(B, 2 * h, generator=g) / B
    ((words * dW.to(dev)).sum() + (vecQ * dQ.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    prm = {k: v.cpu().to(dtype).requires_grad_(True) for k, v in enc.to_reference_dict().items()}
    vs = mo.VarStore(params=prm, dtype=dtype)
    ki, kq = (enc.keep_in, enc.keep_q) if train else (1.0, 1.0)
    masks = None
    if train:
        masks = [torch.from_numpy(dh.mask_for(9, 11, 0, ki, (B, S, E), b0=b0)).to(dtype),
                 torch.from_numpy(dh.mask_for(9, 12, 0, kq, (B, 2 * h), b0=b0)).to(dtype)]
    rw, rq = mo.question_encoder(cfg, vs, q, lengths, V, keep_input=ki, keep_question=kq, masks=masks)
    ((rw * dW.to(dtype)).sum() + (rq * dQ.to(dtype)).sum()).backward()
    return enc, words, vecQ, rw, rq, prm, lengths


@pytest.mark.parametrize("B,S,V,E,h,train", [(3, 5, 11, 20, 128, False), (4, 7, 13, 300, 128, True), (2, 9, 30, 64, 256, True),
                                             (1, 1, 4, 16, 128, False),
                                             # hidden width above the padded embedding width with more than 64 positions: the
                                             # recurrent block's split-reduction slabs are the larger ones (found by the fuzz)
                                             (7, 11, 33, 4, 256, False)])
def test_encoder_matches_bilstm_oracle(macx, dev, B, S, V, E, h, train):
    enc, words, vecQ, rw, rq, prm, lengths = run_case(macx, dev, B, S, V, E, h, train, b0=2)
    assert rel_err(words, rw) < 1e-5 and rel_err(vecQ, rq) < 1e-5
    for b in range(B):      # dynamic_rnn: zero output past the question's end
        assert int(lengths[b]) == S or float(words[b, int(lengths[b]):].detach().abs().max()) == 0.0
    for f, name in macx.encoder.REF_NAMES.items():
        assert rel_err(getattr(enc, f).grad, prm[name].grad, floor=1e-7) < 2e-4, f


def test_encoder_clevr_shape_fp32(macx, dev):
    """CLEVR shape: vocabulary ~90, wrdEmbDim 300, encDim 512, questions up to 43 words, against the fp32 restatement."""
    enc, words, vecQ, rw, rq, prm, _ = run_case(macx, dev, 16, 43, 90, 300, 256, True, dtype=torch.float32)
    assert max_abs(words, rw) < 2e-5 and max_abs(vecQ, rq) < 2e-5
    for f, name in macx.encoder.REF_NAMES.items():
        assert rel_err(getattr(enc, f).grad, prm[name].grad, floor=1e-6) < 2e-3, f


def test_encoder_fixed_embeddings_and_determinism(macx, dev):
    enc, words, vecQ, rw, rq, prm, _ = run_case(macx, dev, 3, 6, 9, 32, 128, True, fixed=True)
    assert enc.emb.grad is None and enc.fw_kernel.grad is not None
    enc2, words2, vecQ2, *_ = run_case(macx, dev, 3, 6, 9, 32, 128, True, fixed=True)
    assert torch.equal(words, words2) and torch.equal(vecQ, vecQ2) and torch.equal(enc.fw_kernel.grad, enc2.fw_kernel.grad)


def test_encoder_shard_draws_full_batch_masks(macx, dev):
    """b0: a data-parallel shard computes exactly the rows the full batch would (dropout keyed on the global question)."""
    cfg = mo.flag_file_config("args", ctrlDim=256, memDim=256, attDim=256, encDim=256, wrdEmbDim=24)
    enc = macx.QuestionEncoder(cfg, vocab=10, generator=torch.Generator().manual_seed(3)).to(dev)
    q, lengths = make_questions(6, 5, 10, seed=2)
    w_full, v_full = enc(q.to(dev), lengths.to(dev), train=True, seed=4)
    w_sh, v_sh = enc(q[4:].to(dev), lengths[4:].to(dev), train=True, seed=4, b0=4)
    assert torch.equal(w_full[4:], w_sh) and torch.equal(v_full[4:], v_sh)


def test_encoder_rejects_unsupported(macx):
    for over in (dict(encType="GRU"), dict(encVariationalDropout=True)):
        with pytest.raises(macx.UnsupportedOptions):
            macx.QuestionEncoder(mo.flag_file_config("args", **over), vocab=10)
    with pytest.raises(ValueError, match="already exists"):             # what the reference raises for a second layer
        macx.QuestionEncoder(mo.flag_file_config("args", encNumLayers=2), vocab=10)
    # LSTM configurations without a fused kernel come back on the generic path
    for over in (dict(encBi=False), dict(encProj=True), dict(encDim=256)):
        assert type(macx.QuestionEncoder(mo.flag_file_config("args", **over), vocab=10)) is macx.GenericQuestionEncoder


def test_full_tower_ids_to_logits_gradients(macx, dev):
    """The reference's feed dict (question ids, lengths, image features, answers) -> MACNet (encoder -> stem -> cell x p ->
    classifier -> CE): logits, loss and every parameter gradient against the oracle chain with identical masks."""
    B, H, W, Cin, d, p, S, A, V, E = 3, 4, 3, 128, 256, 2, 6, 7, 12, 20
    cfg = mo.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d, encDim=d, wrdEmbDim=E, outClassifierDims=[32],
                              answerWordsNum=A)
    cfg.stemDim = 128
    net = macx.MACNet(cfg, vocab=V, H=H, W=W, imageInDim=Cin, answerWordsNum=A, generator=torch.Generator().manual_seed(4)).to(dev)
    g = torch.Generator().manual_seed(6)
    img = torch.relu(torch.randn(B, H * W, Cin, generator=g))
    q, lengths = make_questions(B, S, V, seed=3, min_len=2)
    ans = torch.tensor([1, 5, 2])
    logits = net(img.to(dev), q.to(dev), lengths.to(dev), train=True, seed=21)
    loss, pred = net.loss_and_pred(logits, ans.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    dt = torch.float64
    prm = {}
    for src in (net.enc.to_reference_dict(), net.stem.to_reference_dict(), net.cell.to_reference_dict(), net.out.to_reference_dict()):
        prm.update({k: v.cpu().to(dt).requires_grad_(True) for k, v in src.items()})
    vs = mo.VarStore(params=prm, dtype=dt)
    keeps = (cfg.memoryDropout, cfg.readDropout, cfg.writeDropout)
    sk, ok, ki, kq = net.stem.keep, net.out.keep, net.enc.keep_in, net.enc.keep_q
    mk = lambda site, keep, shape: torch.from_numpy(dh.mask_for(21, site, 0, keep, shape)).to(dt)
    words, vq = mo.question_encoder(cfg, vs, q, lengths, V, keep_input=ki, keep_question=kq, masks=[mk(11, ki, (B, S, E)), mk(12, kq, (B, d))])
    kb = mo.stem_cnn(cfg, vs, img.to(dt), H, W, keep=sk, masks=[mk(9, sk, (B, H * W, Cin)), mk(10, sk, (B, H * W, 128))])
    c, m, _ = mo.mac_network(cfg, vs, vq, words, words, lengths, kb, train=True, mask_fn=mo.hash_mask_fn(21, keeps), keeps=keeps)
    rl = mo.output_classifier(cfg, vs, m, vq, output_keep=ok, masks=[mk(7, ok, (B, 2 * d)), mk(8, ok, (B, 32))])
    rloss, rpred = mo.answer_loss_and_pred(rl, ans)
    rloss.backward()
    assert max_abs(logits, rl) < 5e-5 and torch.equal(pred.cpu(), rpred)
    assert abs(float(loss) - float(rloss)) < 1e-5
    bad = {}
    one = lambda names: {f: [(n, None)] for f, n in names.items()}
    for mod, refs in ((net.enc, one(macx.encoder.REF_NAMES)), (net.stem, one(macx.stem.REF_NAMES)),
                      (net.cell, macx.params.reference_names(cfg, p)), (net.out, one(macx.output.REF_NAMES))):
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


@pytest.mark.parametrize("variant", ["uni", "uni_projected", "bi_proj_tanh", "uni_proj_prelu", "bi_other_width", "bi_off_granule",
                                     "uni_off_granule"])
@pytest.mark.parametrize("train", [False, True])
def test_generic_encoder_matches_oracle(macx, dev, variant, train):
    """macx.QuestionEncoder on the LSTM configurations the fused kernels refuse (no --encBi = the parser's default; projCW /
    projQ output projections): one HIP kernel per reference op (macx_embed_lookup, macx_linear, macx_op_*, macx_wgrad) against
    the fp64 oracle with identical dropout masks -- outputs <= 1e-5, every gradient <= 2e-4."""
    from test_generic_encoder_host import enc_cfg, run_pair, check
    enc, words, vecQ, prm, lengths = run_pair(macx, enc_cfg(variant), train, dev=dev)
    torch.cuda.synchronize()
    check(enc, words, vecQ, prm, lengths, tol=1e-5, gtol=2e-4)


@pytest.mark.parametrize("contextual", [True, False])
def test_generic_tower_on_the_gpu(macx, dev, contextual):
    """macx.MACNet on the reference's default option structure (generic encoder + generic cell + generic classifier around the
    fused stem), with and without --controlContextual: ids -> logits and the gradients against the oracle chain."""
    from test_generic_tower_host import run_tower
    run_tower(macx, contextual, dev=dev)
    torch.cuda.synchronize()
