"""-m gpu: the FUSED read and write units through their own C-ABI entry points (macx_read_{fwd,bwd},
macx_write_{fwd,bwd}, macx_workspace_bytes) -- caller-chosen memory / control / info and incoming gradient -- against the
oracle's units (mac_cell.py:209-277, :305-375) on the same parameters and dropout masks.
Forward <= 1e-5 relative (attention <= 2e-6 absolute), every gradient <= 2e-4."""
import ctypes as C

import pytest
import torch

from oracle import mac_oracle as mo
from oracle import dropout_hash as dh
from helpers import rel_err
from test_gpu_generic import oracle_params, assert_grad

pytestmark = pytest.mark.gpu

READ_FIELDS = ("projX_W", "projX_b", "projY_W", "projY_b", "memKbProj_W", "memKbProj_b", "memKbProj2_W", "memKbProj2_b",
               "kbLogits_w", "kbLogits_b")
WRITE_FIELDS = ("newMemory_W", "newMemory_b", "gate_W", "gate_b")


def P(t):
    return C.c_void_p(t.data_ptr())


class Unit:
    """ctypes plumbing of one unit call: frozen options, shapes with p = 1, parameter / gradient structs, buffers."""

    def __init__(self, macx, dev, cfg, B, N, d, train, seed):
        lib = macx._lib
        self.macx, self.cfg = macx, cfg
        self.lib, self.L = lib, lib.lib()
        self.opts = macx.options.freeze(cfg)
        self.shapes = lib.MacxShapes(B=B, S=4, N=N, d=d, p=1, b0=0)
        keeps = (cfg.memoryDropout, cfg.readDropout, cfg.writeDropout) if train else (1.0, 1.0, 1.0)
        self.keeps = keeps
        self.drop = lib.MacxDropout(keep_memory=keeps[0], keep_read=keeps[1], keep_write=keeps[2], seed=seed)
        vq, words, lengths, kb = mo.synthetic_inputs(B, 4, N, d, seed=13)
        self.kb = kb
        self.ref_params = oracle_params(cfg, vq, words, lengths, kb)
        self.params = macx.MACCellParams(cfg, 1).load_reference_dict(self.ref_params).to(dev)
        self.pstruct, self.gstruct, self.grads = lib.MacxParams(), lib.MacxParamGrads(), {}
        for f in lib.PARAM_FIELDS:
            t = getattr(self.params, f, None) if f in self.params.fields else None
            setattr(self.pstruct, f, t.data_ptr() if t is not None else None)
            if t is not None:
                self.grads[f] = torch.full_like(t, float("nan"))
            setattr(self.gstruct, f, self.grads[f].data_ptr() if t is not None else None)
        o, s = C.byref(self.opts), C.byref(self.shapes)
        self.saved_floats = self.L.macx_saved_floats(o, s, 1)
        self.saved = torch.empty(self.saved_floats, dtype=torch.float32, device=dev)
        wb = self.L.macx_workspace_bytes(o, s, 1)
        assert wb == 4 * self.L.macx_ws_floats(o, s, 1) and wb > 0
        self.ws = torch.empty(wb // 4, dtype=torch.float32, device=dev)
        self.stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        # the oracle cell (fp64) on the same variables
        self.op = {k: v.double().clone().requires_grad_(True) for k, v in self.ref_params.items()}
        self.vs = mo.VarStore(params=self.op, dtype=torch.float64)
        self.ocell = mo.MACCellOracle(cfg, self.vs, vq.double(), words.double(), words.double(), lengths, kb.double(), keeps[0],
                                      keeps[1], keeps[2], B, train, mask_fn=mo.hash_mask_fn(seed, keeps, b0=0) if train else None)
        with self.vs.scope("MACnetwork"):
            self.ocell.zero_state(B)

    def head(self):
        return (C.byref(self.opts), C.byref(self.shapes), C.byref(self.drop), C.byref(self.pstruct))

    def check_param_grads(self, fields):
        """The unit's parameter gradients against the oracle's (reference names -> macx_params fields through a second
        MACCellParams loaded with the gradients); the other units' gradient buffers must be untouched (still NaN)."""
        holder = self.macx.MACCellParams(self.cfg, 1, dtype=torch.float64)
        holder.load_reference_dict({k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in self.op.items()})
        seen = 0
        for f in fields:
            if f not in self.grads:
                continue
            if all(self.op[name].grad is None for name, _ in self.params._names[f]):
                continue
            assert torch.isfinite(self.grads[f]).all(), f
            assert_grad(self.grads[f].cpu().double(), getattr(holder, f).detach(), f, tol=2e-4)
            seen += 1
        assert seen >= 2
        for f, g in self.grads.items():
            if f not in fields:
                assert torch.isnan(g).all(), f


def rand(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize("name,over", [("args", {}), ("args4", {}), ("args", dict(readMemAct="RELU", readCtrlAct="TANH"))])
@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("B,N,d", [(3, 30, 128), (2, 196, 256)])
def test_read_unit_exports(macx, dev, name, over, train, B, N, d):
    cfg = mo.flag_file_config(name, netLength=1, memDim=d, ctrlDim=d, attDim=d, **over)
    u = Unit(macx, dev, cfg, B, N, d, train, seed=11)
    mem, ctl, w = rand((B, d), 1), rand((B, d), 2), rand((B, d), 3)
    mo_, co_, kbo = mem.double().requires_grad_(True), ctl.double().requires_grad_(True), u.kb.double().requires_grad_(True)
    with u.vs.scope("MACnetwork"), u.vs.scope("MACCell"):
        info_ref = u.ocell.read(kbo, mo_, co_)
    (info_ref * w.double()).sum().backward()

    kbd, memd, ctld, wd = [t.to(dev).contiguous() for t in (u.kb, mem, ctl, w)]
    info, att = torch.empty(B, d, device=dev), torch.empty(B, N, device=dev)
    u.lib.check(u.L.macx_read_fwd(*u.head(), P(kbd), P(memd), P(ctld), P(u.saved), u.saved_floats, P(info), P(att), u.stream), "read_fwd")
    dkb, dmem, dctl = torch.empty_like(kbd), torch.empty_like(memd), torch.empty_like(ctld)
    u.lib.check(u.L.macx_read_bwd(*u.head(), P(kbd), P(u.saved), u.saved_floats, P(u.ws), u.ws.numel(), P(wd), C.byref(u.gstruct),
                                  P(dkb), P(dmem), P(dctl), u.stream), "read_bwd")
    torch.cuda.synchronize()
    assert rel_err(info, info_ref) < 1e-5
    assert float((att.cpu().double() - u.ocell.attentions["kb"][-1].detach()).abs().max()) < 2e-6
    assert rel_err(dkb, kbo.grad) < 2e-4
    assert rel_err(dmem, mo_.grad) < 2e-4
    assert rel_err(dctl, co_.grad) < 2e-4
    u.check_param_grads(READ_FIELDS)


@pytest.mark.parametrize("name,over,train", [("args", {}, False), ("args", {}, True), ("args4", {}, False), ("args4", {}, True),
                                             ("args", dict(writeMemAct="ELU"), True)])
def test_write_unit_exports(macx, dev, name, over, train):
    B, N, d = 5, 10, 128
    cfg = mo.flag_file_config(name, netLength=1, memDim=d, ctrlDim=d, attDim=d, writeDropout=0.8, **over)
    u = Unit(macx, dev, cfg, B, N, d, train, seed=17)
    mem, info, ctl, w = rand((B, d), 1), rand((B, d), 2), rand((B, d), 3), rand((B, d), 4)
    a = [t.double().requires_grad_(True) for t in (mem, info, ctl)]
    info_in = a[1]
    if train:       # the write dropout sits between read and write in the cell's step (mac_cell.py:461-463)
        mask = torch.as_tensor(dh.mask_for(17, dh.SITE_WRITE_INFO, 0, 0.8, (B, d)), dtype=torch.float64)
        info_in = mo.Ops.dropout(a[1], 0.8, mask)
    with u.vs.scope("MACnetwork"), u.vs.scope("MACCell"):
        ref = u.ocell.write(a[0], info_in, a[2], a[2])
    (ref * w.double()).sum().backward()

    memd, infod, ctld, wd = [t.to(dev).contiguous() for t in (mem, info, ctl, w)]
    out = torch.empty(B, d, device=dev)
    u.lib.check(u.L.macx_write_fwd(*u.head(), P(memd), P(infod), P(ctld), P(u.saved), u.saved_floats, P(out), u.stream), "write_fwd")
    dmem, dinfo, dctl = torch.empty_like(memd), torch.empty_like(memd), torch.empty_like(memd)
    u.lib.check(u.L.macx_write_bwd(*u.head(), P(u.saved), u.saved_floats, P(u.ws), u.ws.numel(), P(wd), C.byref(u.gstruct),
                                   P(dmem), P(dinfo), P(dctl), u.stream), "write_bwd")
    torch.cuda.synchronize()
    assert rel_err(out, ref) < 1e-5
    assert rel_err(dmem, a[0].grad) < 2e-4
    assert rel_err(dinfo, a[1].grad) < 2e-4
    if a[2].grad is not None:
        assert rel_err(dctl, a[2].grad) < 2e-4
    else:
        assert float(dctl.abs().max()) == 0.0
    u.check_param_grads(WRITE_FIELDS)


def test_unit_exports_reject_what_they_cannot_do(macx, dev):
    d = 128
    cfg = mo.flag_file_config("args3", netLength=1, memDim=d, ctrlDim=d, attDim=d)          # writeSelfAtt
    u = Unit(macx, dev, cfg, 2, 10, d, False, seed=1)
    x = torch.zeros(2, d, device=dev)
    rc = u.L.macx_write_fwd(*u.head(), P(x), P(x), P(x), P(u.saved), u.saved_floats, P(x), u.stream)
    assert rc == -2                                                                               # MACX_EUNSUPPORTED
    u.shapes.p = 2                                                                                # units are single steps
    att = torch.zeros(2, 10, device=dev)
    kb = u.kb.to(dev)
    assert u.L.macx_read_fwd(*u.head(), P(kb), P(x), P(x), P(u.saved), u.saved_floats, P(x), P(att), u.stream) == -1
    u.shapes.p = 1
    assert u.L.macx_read_fwd(*u.head(), P(kb), P(x), P(x), P(u.saved), 16, P(x), P(att), u.stream) == -4      # MACX_ESMALL


@pytest.mark.parametrize("unshared,act,B,d,p", [(True, "TANH", 5, 128, 3), (False, "TANH", 3, 256, 2), (True, "NON", 64, 512, 12),
                                                (True, "RELU", 4, 128, 2)])
def test_ctrl_inputs_exports(macx, dev, unshared, act, B, d, p):
    """SURVEY 8b's `ctrl_inputs` unit (mac_cell.py:442-448) through macx_ctrl_inputs_fwd / _bwd against the closed form in fp64:
    t = act(vecQ Wq + bq), cI_i = t WqU_i + bqU_i; d_vecQ and the four parameter gradients from a random d_cI."""
    lib = macx._lib
    L = lib.lib()
    cfg = mo.flag_file_config("args", netLength=p, memDim=d, ctrlDim=d, attDim=d, controlInputUnshared=unshared, controlInputAct=act)
    opts = macx.options.freeze(cfg)
    shapes = lib.MacxShapes(B=B, S=4, N=16, d=d, p=p, b0=0)
    nU = p if unshared else 1
    g = torch.Generator().manual_seed(3)
    vq = torch.randn(B, d, generator=g)
    Wq, bq = torch.randn(d, d, generator=g) / d ** 0.5, torch.randn(d, generator=g) * 0.1
    WqU, bqU = torch.randn(nU, d, d, generator=g) / d ** 0.5, torch.randn(nU, d, generator=g) * 0.1
    dcI = torch.randn(p, B, d, generator=g)
    dv = [t.to(dev).contiguous() for t in (vq, Wq, bq, WqU, bqU, dcI)]
    vqd, Wqd, bqd, WqUd, bqUd, dcId = dv
    ps, gs = lib.MacxParams(), lib.MacxParamGrads()
    ps.qInput_W, ps.qInput_b, ps.qInputU_W, ps.qInputU_b = Wqd.data_ptr(), bqd.data_ptr(), WqUd.data_ptr(), bqUd.data_ptr()
    grads = {k: torch.full_like(t, float("nan")) for k, t in (("qInput_W", Wqd), ("qInput_b", bqd), ("qInputU_W", WqUd), ("qInputU_b", bqUd))}
    for k, t in grads.items():
        setattr(gs, k, t.data_ptr())
    n = L.macx_ctrl_inputs_ws_floats(C.byref(opts), C.byref(shapes))
    assert n > 0
    ws = torch.empty(n, device=dev)
    t_out, cI = torch.empty(B, d, device=dev), torch.empty(p, B, d, device=dev)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    lib.check(L.macx_ctrl_inputs_fwd(C.byref(opts), C.byref(shapes), C.byref(ps), P(vqd), P(t_out), P(cI), P(ws), n, st), "ctrl_inputs_fwd")
    dvq = torch.empty(B, d, device=dev)
    lib.check(L.macx_ctrl_inputs_bwd(C.byref(opts), C.byref(shapes), C.byref(ps), P(vqd), P(t_out), P(dcId), C.byref(gs), P(dvq), P(ws), n, st),
              "ctrl_inputs_bwd")
    torch.cuda.synchronize()
    # fp64 closed form through autograd
    vq64, Wq64, bq64, WqU64, bqU64 = [t.double().requires_grad_(True) for t in (vq, Wq, bq, WqU, bqU)]
    assert cfg.relu == "ELU"                  # configs/args.txt: "RELU" resolves to ELU (ops.py:161-179)
    pre = vq64 @ Wq64 + bq64
    t64 = {"TANH": torch.tanh, "NON": lambda x: x, "RELU": torch.nn.functional.elu}[act](pre)
    c64 = torch.stack([t64 @ WqU64[i if unshared else 0] + bqU64[i if unshared else 0] for i in range(p)])
    (c64 * dcI.double()).sum().backward()
    assert rel_err(t_out, t64.detach()) < 1e-5
    assert rel_err(cI, c64.detach()) < 1e-5
    for name, got, ref in (("d_vecQ", dvq, vq64.grad), ("qInput_W", grads["qInput_W"], Wq64.grad), ("qInput_b", grads["qInput_b"], bq64.grad),
                           ("qInputU_W", grads["qInputU_W"], WqU64.grad), ("qInputU_b", grads["qInputU_b"], bqU64.grad)):
        assert torch.isfinite(got).all(), name
        assert rel_err(got, ref) < 2e-4, (name, rel_err(got, ref))
    # too small a workspace, a missing field
    assert L.macx_ctrl_inputs_fwd(C.byref(opts), C.byref(shapes), C.byref(ps), P(vqd), P(t_out), P(cI), P(ws), n - 1, st) != 0
    ps.qInputU_W = None
    assert L.macx_ctrl_inputs_fwd(C.byref(opts), C.byref(shapes), C.byref(ps), P(vqd), P(t_out), P(cI), P(ws), n, st) != 0
