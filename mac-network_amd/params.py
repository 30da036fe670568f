"""Parameters of the MAC cell, stored in the layouts the kernels read and exposed under the
reference's TF variable names for checkpoint import/export."""
import math
import weakref

import torch

from . import _lib
from .options import get

SCOPE = "MACnetwork/MACCell/"
# gradients written by phase 2 of the backward pass (the deferred contractions over all p*B*N rows and their bias sums)
LATE_FIELDS = ("projX_W", "projX_b", "memKbProj_W", "memKbProj_b", "memKbProj2_W", "memKbProj2_b", "kbLogits_w", "kbLogits_b")
_LIN = "linearLayer%s/weights/weight"
_BIAS = "linearLayer%s/biases/bias"


def reference_names(config, p):
    """internal field -> list of (reference variable name, index into the stacked tensor or None)."""
    names = {
        "qInput_W": [(SCOPE + _LIN % "qInput", None)],
        "qInput_b": [(SCOPE + _BIAS % "qInput", None)],
        "ctrlLogits_w": [(SCOPE + "control/inter2logits/" + _LIN % "logits", None)],
        "ctrlLogits_b": [(SCOPE + "control/inter2logits/" + _BIAS % "logits", None)],
        "projX_W": [(SCOPE + "read/mulmemInter/" + _LIN % "projX", None)],
        "projX_b": [(SCOPE + "read/mulmemInter/" + _BIAS % "projX", None)],
        "projY_W": [(SCOPE + "read/mulmemInter/" + _LIN % "projY", None)],
        "projY_b": [(SCOPE + "read/mulmemInter/" + _BIAS % "projY", None)],
        "memKbProj_W": [(SCOPE + "read/" + _LIN % "memKbProj", None)],
        "memKbProj_b": [(SCOPE + "read/" + _BIAS % "memKbProj", None)],
        "memKbProj2_W": [(SCOPE + "read/linearLayermemKbProj/" + _LIN % "memKbProj_2", None)],
        "memKbProj2_b": [(SCOPE + "read/linearLayermemKbProj/" + _BIAS % "memKbProj_2", None)],
        "kbLogits_w": [(SCOPE + "read/inter2att/inter2logits/" + _LIN % "logits", None)],
        "kbLogits_b": [(SCOPE + "read/inter2att/inter2logits/" + _BIAS % "logits", None)],
        "newMemory_W": [(SCOPE + "write/" + _LIN % "newMemory", None)],
        "newMemory_b": [(SCOPE + "write/" + _BIAS % "newMemory", None)],
    }
    if get(config, "controlInputUnshared"):
        names["qInputU_W"] = [(SCOPE + _LIN % ("qInput%d" % i), i) for i in range(p)]
        names["qInputU_b"] = [(SCOPE + _BIAS % ("qInput%d" % i), i) for i in range(p)]
    else:
        names["qInputU_W"] = [(SCOPE + _LIN % "qInputU", 0)]
        names["qInputU_b"] = [(SCOPE + _BIAS % "qInputU", 0)]
    # the state variables exist only for parametric initialisation (mac_cell.py:496-505)
    if get(config, "initMem") == "PRM":
        names["initMem"] = [("MACnetwork/initMem", None)]
    if get(config, "initCtrl") == "PRM":
        names["initCtrl"] = [("MACnetwork/initCtrl", None)]
    if get(config, "writeSelfAtt"):
        names["selfCtrl_W"] = [(SCOPE + "write/" + _LIN % "ctrlProj", None)]
        names["selfCtrl_b"] = [(SCOPE + "write/" + _BIAS % "ctrlProj", None)]
        names["selfLogits_w"] = [(SCOPE + "write/inter2attselfAttention/inter2logits/" + _LIN % "logits", None)]
        names["selfLogits_b"] = [(SCOPE + "write/inter2attselfAttention/inter2logits/" + _BIAS % "logits", None)]
    if get(config, "writeGate"):
        names["gate_W"] = [(SCOPE + "write/" + _LIN % "gate", None)]
        names["gate_b"] = [(SCOPE + "write/" + _BIAS % "gate", None)]
    if get(config, "controlFeedPrev"):
        names["contControl_W"] = [(SCOPE + "control/" + _LIN % "contControl", None)]
        names["contControl_b"] = [(SCOPE + "control/" + _BIAS % "contControl", None)]
        if get(config, "controlContAct") != "NON":
            names["contControl2_W"] = [(SCOPE + "control/linearLayercontControl/" + _LIN % "contControl_2", None)]
            names["contControl2_b"] = [(SCOPE + "control/linearLayercontControl/" + _BIAS % "contControl_2", None)]
    return names


def shapes(config, p):
    d = get(config, "memDim")
    nU = p if get(config, "controlInputUnshared") else 1
    win = 2 * d + (d if get(config, "writeSelfAtt") else 0)
    cin = d + (d if get(config, "controlFeedInputs") else 0)
    sh = {
        "initMem": (d,), "initCtrl": (d,), "qInput_W": (d, d), "qInput_b": (d,), "qInputU_W": (nU, d, d),
        "qInputU_b": (nU, d), "ctrlLogits_w": (d,), "ctrlLogits_b": (1,), "projX_W": (d, d), "projX_b": (d,),
        "projY_W": (d, d), "projY_b": (d,), "memKbProj_W": (2 * d, d), "memKbProj_b": (d,), "memKbProj2_W": (d, d),
        "memKbProj2_b": (d,), "kbLogits_w": (d,), "kbLogits_b": (1,), "newMemory_W": (win, d), "newMemory_b": (d,),
        "selfCtrl_W": (d, d), "selfCtrl_b": (d,), "selfLogits_w": (d,), "selfLogits_b": (1,), "gate_W": (d, d),
        "gate_b": (d,), "contControl_W": (cin, d), "contControl_b": (d,), "contControl2_W": (d, d),
        "contControl2_b": (d,),
    }
    return sh


class MACCellParams(torch.nn.Module):
    """Trainable parameters of the cell.  One tensor per macx_params field."""

    def __init__(self, config, netLength=None, device=None, generator=None, dtype=torch.float32):
        super().__init__()
        self.p = int(netLength if netLength is not None else get(config, "netLength"))
        self._names = reference_names(config, self.p)
        sh = shapes(config, self.p)
        created = [f for f in _lib.PARAM_FIELDS if f in self._names]
        # order of tensors() / of the flat gradient buffer: the read unit's [B,N,d]-contraction weights LAST.  Their gradients
        # are the last thing the backward pass produces (macx_cell_backward_phase, phase 2), so everything in front of them
        # is one contiguous range that a data-parallel all-reduce can take while phase 2 still runs (dp.OverlappedBuckets)
        self.fields = [f for f in created if f not in LATE_FIELDS] + [f for f in created if f in LATE_FIELDS]
        gen = generator
        for f in created:
            shape = sh[f]
            if f in ("initMem", "initCtrl"):
                t = torch.randn(shape, generator=gen, dtype=torch.float64)          # mac_cell.py:498-499
            elif f.endswith("_b"):
                t = torch.zeros(shape, dtype=torch.float64)                          # ops.py:40
            else:
                # xavier-uniform (ops.py:20): limit sqrt(6/(fan_in+fan_out)); 1-D weight: sqrt(3/n)
                if f.endswith("_w"):
                    lim = math.sqrt(3.0 / shape[0])
                else:
                    lim = math.sqrt(6.0 / (shape[-2] + shape[-1]))
                t = (torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * lim
            self.register_parameter(f, torch.nn.Parameter(t.to(dtype).to(device) if device else t.to(dtype)))

    def tensors(self):
        return [getattr(self, f) for f in self.fields]

    def early_floats(self):
        """Floats of grad_buffer() in front of the first LATE_FIELDS segment: complete after phase 1 of the backward pass."""
        return sum((getattr(self, f).numel() + 3) & ~3 for f in self.fields if f not in LATE_FIELDS)

    def grad_buffer(self):
        """Persistent flat fp32 buffer the backward pass writes the parameter gradients into (16-byte aligned segments in
        `fields` order): the gradients autograd hands out are views of it, so a data-parallel all-reduce (macx.dp.GradBucket)
        or a flat optimizer over exactly these tensors (optim.FlatAdamEMA(..., grad_owner=params) uses the same padded layout)
        can run on it without a gather copy.  Allocated on first use; the backward pass only writes into it for a registered
        consumer (register_grad_buffer_user), once per step, until the consumer releases it (release_grad_buffer)."""
        dev = self.tensors()[0].device
        n = sum((t.numel() + 3) & ~3 for t in self.tensors())
        buf = getattr(self, "_grad_flat", None)
        if buf is None or buf.numel() != n or buf.device != dev:
            buf = torch.zeros(n, dtype=torch.float32, device=dev)
            buf._macx_cell_grad_buffer = True          # optim.FlatAdamEMA.step refuses it while NOBODY is registered as its consumer
            buf._macx_cell_grad_owner = weakref.ref(self)
            object.__setattr__(self, "_grad_flat", buf)
        return buf

    def register_grad_buffer_user(self):
        """A flat consumer (dp.GradBucket / OverlappedBuckets, a flat optimizer) announces itself: from now on the backward pass
        writes into grad_buffer() -- once per step; the consumer calls release_grad_buffer() when it is done with the step's
        gradients.  Without a registered consumer every backward pass gets a buffer of its own (two cell runs in one backward,
        torch.autograd.grad results that must survive the next call)."""
        object.__setattr__(self, "_grad_flat_registered", True)
        object.__setattr__(self, "_grad_flat_busy", False)

    def claim_grad_buffer(self):
        """The persistent buffer, zeroed, if a consumer registered and this step's buffer has not been handed out yet; else None."""
        if not getattr(self, "_grad_flat_registered", False) or getattr(self, "_grad_flat_busy", False):
            return None
        object.__setattr__(self, "_grad_flat_busy", True)
        return self.grad_buffer().zero_()

    def release_grad_buffer(self):
        object.__setattr__(self, "_grad_flat_busy", False)

    def to_reference_dict(self):
        """{TF variable name: tensor} with the reference's shapes (scalar biases are 0-d)."""
        out = {}
        for f in self.fields:
            t = getattr(self, f).detach()
            for name, idx in self._names[f]:
                v = t if idx is None else t[idx]
                if f in ("ctrlLogits_b", "kbLogits_b", "selfLogits_b"):
                    v = v.reshape(())
                out[name] = v.clone()
        return out

    @torch.no_grad()
    def load_reference_dict(self, ref):
        for f in self.fields:
            t = getattr(self, f)
            for name, idx in self._names[f]:
                src = torch.as_tensor(ref[name]).to(t.dtype).to(t.device)
                dst = t if idx is None else t[idx]
                dst.copy_(src.reshape(dst.shape))
        return self
