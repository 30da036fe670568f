"""TEST INFRASTRUCTURE -- not part of the product path.  (Pinned through oracle/mac_oracle.py, which is checked against the
reference's own code executed on tests/tf1_shim; this file is that oracle's independent cross-check.)

Second, independent restatement of the reference's MAC cell: closed-form numpy fp64 forward for the
five published flag files (configs/args.txt, args1-4.txt), written from the equations rather than
from the op sequence -- the fused algebra of SURVEY.md 8a "Forward spec":

    cI   = act_in(vecQ Wq + bq) Wq_i + bq_i                        mac_cell.py:442-448
    cc   = cI                      | args1: tanh([c_{i-1}, cI] Wc + bc) Wc2 + bc2     :141-151
    a_c  = softmax_s(mask((cc * words) . w_c + b_c)) ; c_i = sum_s a_c words          :155-181
    m~   = m_{i-1} / keep_m * maskM                                                  :214-215
    X    = drop(KB) Wx + bx ; y = drop(m~) Wy + by                                   ops.py:678-689
    I1   = [X*y, X] W1 + b1 ; H1 = elu(I1) ; I2 = H1 W2 + b2                         ops.py:703,718,326
    a_k  = softmax_n(drop(elu(I2 * c_i)) . w_k + b_k) ; r_i = sum_n a_k KB           mac_cell.py:248-275
    m_i  = [m_{i-1}, r_i (, s^)] Wm + bm   (args3: self attention; args4: gate)      :305-375

It shares no code with mac_oracle.py (different library, different op order, concat-free), so the two
agreeing to fp32/fp64 round-off is the oracle's self-consistency check (tests/test_oracle.py).
"""
import numpy as np

P = "MACnetwork/MACCell/"


def _W(prm, scope):
    return np.asarray(prm[P + scope + "/weights/weight"], dtype=np.float64)


def _b(prm, scope):
    return np.asarray(prm[P + scope + "/biases/bias"], dtype=np.float64)


def _elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))


def _softmax(x):
    x = x - x.max(axis=-1, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=-1, keepdims=True)


def _act(cfg, name, x):
    if name == "NON":
        return x
    if name == "TANH":
        return np.tanh(x)
    if name == "RELU":
        return {"ELU": _elu, "STD": lambda t: np.maximum(t, 0)}[cfg.relu](x)
    raise KeyError(name)


def forward(cfg, prm, vecQ, words, lengths, kb, keeps=(1.0, 1.0, 1.0), masks=None):
    """Returns dict(control, memory, controls[p+1], memories[p+1], infos[p], att_q[p], att_kb[p], ...).
    masks: callable (site, step, shape) -> 0/1 array (oracle/dropout_hash.py), needed when a keep < 1."""
    from . import dropout_hash as dh
    vecQ, words, kb = [np.asarray(t, dtype=np.float64) for t in (vecQ, words, kb)]
    lengths = np.asarray(lengths)
    B, S, d = words.shape
    p = cfg.netLength
    km, kr, kw = keeps

    def mask(site, step, shape, keep):
        if keep == 1.0:
            return 1.0
        return np.asarray(masks(site, step, shape), dtype=np.float64) / keep

    c = {"PRM": lambda: np.tile(np.asarray(prm.get("MACnetwork/initCtrl", np.zeros(d)), dtype=np.float64), (B, 1)),
         "ZERO": lambda: np.zeros((B, d)), "Q": lambda: vecQ}[cfg.initCtrl]()
    m = {"PRM": lambda: np.tile(np.asarray(prm.get("MACnetwork/initMem", np.zeros(d)), dtype=np.float64), (B, 1)),
         "ZERO": lambda: np.zeros((B, d)), "Q": lambda: vecQ}[cfg.initMem]()
    controls, memories, infos, att_q, att_kb, att_self, gates = [c], [m], [], [], [], [], []
    cont = c
    word_mask = np.arange(S)[None, :] < lengths[:, None]
    mem_var = mask(dh.SITE_MEM_VAR, 0, (B, d), km) if cfg.memoryVariationalDropout else None
    t = _act(cfg, cfg.controlInputAct, vecQ @ _W(prm, "linearLayerqInput") + _b(prm, "linearLayerqInput"))
    W1 = _W(prm, "read/linearLayermemKbProj")
    for i in range(p):
        u = "linearLayerqInput%d" % i if cfg.controlInputUnshared else "linearLayerqInputU"
        cI = t @ _W(prm, u) + _b(prm, u)
        cc = cI
        if cfg.controlFeedPrev:
            prev = c if cfg.controlFeedPrevAtt else cont
            x = np.concatenate([prev, cI], axis=-1) if cfg.controlFeedInputs else prev
            cc = x @ _W(prm, "control/linearLayercontControl") + _b(prm, "control/linearLayercontControl")
            if cfg.controlContAct != "NON":
                cc = _act(cfg, cfg.controlContAct, cc)
                s2 = "control/linearLayercontControl/linearLayercontControl_2"
                cc = cc @ _W(prm, s2) + _b(prm, s2)
        cont = cc
        lw = _W(prm, "control/inter2logits/linearLayerlogits")
        logits = np.einsum("bd,bsd,d->bs", cc, words, lw) + _b(prm, "control/inter2logits/linearLayerlogits")
        logits = logits + (1.0 - word_mask) * (-1e30)
        a_c = _softmax(logits)
        c_new = np.einsum("bs,bsd->bd", a_c, words)
        # read
        if cfg.memoryVariationalDropout:
            mt = m * mem_var
        else:
            mt = m * mask(dh.SITE_MEM, i, (B, d), km)
        X = (kb * mask(dh.SITE_READ_KB, i, kb.shape, kr)) @ _W(prm, "read/mulmemInter/linearLayerprojX") \
            + _b(prm, "read/mulmemInter/linearLayerprojX")
        y = (mt * mask(dh.SITE_READ_MEM, i, (B, d), kr)) @ _W(prm, "read/mulmemInter/linearLayerprojY") \
            + _b(prm, "read/mulmemInter/linearLayerprojY")
        I1 = np.einsum("bnk,bk,kj->bnj", X, y, W1[:d]) + X @ W1[d:] + _b(prm, "read/linearLayermemKbProj")
        H1 = _act(cfg, cfg.readMemAct, I1)
        s2 = "read/linearLayermemKbProj/linearLayermemKbProj_2"
        I2 = H1 @ _W(prm, s2) + _b(prm, s2)
        G = _act(cfg, cfg.readCtrlAct, I2 * c_new[:, None, :])
        G = G * mask(dh.SITE_READ_ATT, i, G.shape, kr)
        lk = "read/inter2att/inter2logits/linearLayerlogits"
        a_k = _softmax(G @ _W(prm, lk) + _b(prm, lk))
        r = np.einsum("bn,bnd->bd", a_k, kb)
        if kw < 1.0:
            r = r * mask(dh.SITE_WRITE_INFO, i, (B, d), kw)
        # write
        Wm = _W(prm, "write/linearLayernewMemory")
        parts = [m, r]
        if cfg.writeSelfAtt:
            sc_in = cont if cfg.writeSelfAttMod == "CONT" else c_new
            sc = sc_in @ _W(prm, "write/linearLayerctrlProj") + _b(prm, "write/linearLayerctrlProj")
            ls = "write/inter2attselfAttention/inter2logits/linearLayerlogits"
            Chist = np.stack(controls, axis=1)
            Mhist = np.stack(memories, axis=1)
            a_s = _softmax(np.einsum("bjd,bd,d->bj", Chist, sc, _W(prm, ls)) + _b(prm, ls))
            att_self.append(a_s)
            parts.append(np.einsum("bj,bjd->bd", a_s, Mhist))
        m_new = np.concatenate(parts, axis=-1) @ Wm + _b(prm, "write/linearLayernewMemory")
        m_new = _act(cfg, cfg.writeMemAct, m_new)
        if cfg.writeGate:
            z = 1.0 / (1.0 + np.exp(-(c_new @ _W(prm, "write/linearLayergate") + _b(prm, "write/linearLayergate")
                                      + cfg.writeGateBias)))
            gates.append(z)
            m_new = m_new * z + m * (1 - z)
        c, m = c_new, m_new
        controls.append(c)
        memories.append(m)
        infos.append(r)
        att_q.append(a_c)
        att_kb.append(a_k)
    return dict(control=c, memory=m, controls=np.stack(controls), memories=np.stack(memories), infos=np.stack(infos),
                att_q=np.stack(att_q), att_kb=np.stack(att_kb), att_self=att_self, gates=gates)
