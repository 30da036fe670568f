"""Image input unit ("stem", model.py:165-204): 2 x (dropout -> conv3x3 SAME -> +b -> act) over the
pre-extracted 14 x 14 x 1024 features, producing the cell's knowledge base [B, H*W, memDim].
Implicit-GEMM convolutions on the fp32-MFMA knowledge-base GEMM kernel behind libmacx.so.
SURVEY.md 8f row 1.

    stem = Stem(config, H=14, W=14, inDim=1024).to(device)
    kb = stem(images, train=True, seed=step)          # images [B, H*W, inDim] (NHWC), kb [B, H*W, memDim]
"""
import ctypes as C
import math

import torch

from . import _lib
from .options import UnsupportedOptions, _resolve_act, fresh_seed

REF_NAMES = {"kernel0": "stem/cnnLayercnn_0/kernels/kernel", "bias0": "stem/cnnLayercnn_0/biases/bias",
             "kernel1": "stem/cnnLayercnn_1/kernels/kernel", "bias1": "stem/cnnLayercnn_1/biases/bias"}


class _StemFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, keep, seed, b0, images, *params):
        L = _lib.lib()
        B = images.shape[0]
        sh = _lib.MacxStemShapes(B=B, H=mod.H, W=mod.W, Cin=mod.inDim, Cmid=mod.midDim, Cout=mod.outDim, b0=b0)
        n_saved = L.macx_stem_saved_floats(C.byref(sh))
        if n_saved == 0:
            raise ValueError("stem: channel counts must be multiples of 128")
        images = images.contiguous()
        saved = torch.empty(n_saved, dtype=torch.float32, device=images.device)
        kb = torch.empty(B, mod.H * mod.W, mod.outDim, dtype=torch.float32, device=images.device)
        ps = _lib.MacxStemParams(*[p.data_ptr() for p in params])
        st = C.c_void_p(torch.cuda.current_stream(images.device).cuda_stream)
        _lib.check(L.macx_stem_forward(C.byref(sh), mod.act, keep, seed & 0xFFFFFFFF, C.byref(ps), images.data_ptr(), kb.data_ptr(),
                                       saved.data_ptr(), n_saved, st), "macx_stem_forward")
        ctx.stuff = (mod, keep, seed, sh, saved, n_saved, kb, params)
        return kb

    @staticmethod
    def backward(ctx, d_kb):
        mod, keep, seed, sh, saved, n_saved, kb, params = ctx.stuff
        L = _lib.lib()
        n_ws = L.macx_stem_ws_floats(C.byref(sh))
        ws = torch.empty(n_ws, dtype=torch.float32, device=kb.device)
        grads = [torch.empty_like(p) for p in params]
        gs = _lib.MacxStemGrads(*[g.data_ptr() for g in grads])
        ps = _lib.MacxStemParams(*[p.data_ptr() for p in params])
        d_kb = d_kb.contiguous()
        st = C.c_void_p(torch.cuda.current_stream(kb.device).cuda_stream)
        _lib.check(L.macx_stem_backward(C.byref(sh), mod.act, keep, seed & 0xFFFFFFFF, C.byref(ps), kb.data_ptr(), saved.data_ptr(),
                                        n_saved, ws.data_ptr(), n_ws, d_kb.data_ptr(), C.byref(gs), st), "macx_stem_backward")
        return (None, None, None, None, None) + tuple(grads)     # image features are inputs, not trained (extract_features.py)


class Stem(torch.nn.Module):
    def __init__(self, config, H=14, W=14, inDim=1024, generator=None):
        super().__init__()
        g = lambda n, dflt: getattr(config, n, dflt)
        if g("stemLinear", False) or g("stemBN", False) or g("stemGridRnn", False) or g("locationAware", False):
            raise UnsupportedOptions("stem: only the default 2-layer 3x3 CNN has a HIP path")
        if g("stemNumLayers", 2) != 2 or g("stemKernelSize", 3) != 3 or g("stemKernelSizes", None) or g("stemStrideSizes", None):
            raise UnsupportedOptions("stem: stemNumLayers=2, stemKernelSize=3, stride 1 only")
        self.H, self.W, self.inDim = H, W, inDim
        self.midDim, self.outDim = g("stemDim", 512), g("memDim", 512)
        self.act = _resolve_act(config, "RELU")               # CNNLayer's default act (ops.py:423)
        self.keep = float(g("stemDropout", 0.82))
        shapes = {"kernel0": (3, 3, inDim, self.midDim), "bias0": (self.midDim,), "kernel1": (3, 3, self.midDim, self.outDim),
                  "bias1": (self.outDim,)}
        for f in _lib.STEM_FIELDS:
            sh = shapes[f]
            if f.startswith("bias"):
                t = torch.zeros(sh, dtype=torch.float64)
            else:   # xavier-uniform over conv fans (ops.py:30): fan_in = 9*in, fan_out = 9*out
                lim = math.sqrt(6.0 / (9 * sh[2] + 9 * sh[3]))
                t = (torch.rand(sh, generator=generator, dtype=torch.float64) * 2 - 1) * lim
            self.register_parameter(f, torch.nn.Parameter(t.float()))

    def tensors(self):
        return [getattr(self, f) for f in _lib.STEM_FIELDS]

    def to_reference_dict(self):
        return {REF_NAMES[f]: getattr(self, f).detach().clone() for f in _lib.STEM_FIELDS}

    def forward(self, images, train=False, seed=None, b0=0):
        """images: [B, H*W, inDim] / [B, H, W, inDim] (NHWC, what the graph sees after model.py:68) or the feed-dict layout
        [B, inDim, H, W] (h5 features, extract_features.py), which is transposed on the device first."""
        if not images.is_cuda:
            raise RuntimeError("the stem has no CPU path")
        if images.dim() == 4 and images.shape[1] == self.inDim and tuple(images.shape[2:]) == (self.H, self.W):
            src = images.contiguous()
            images = torch.empty(src.shape[0], self.H * self.W, self.inDim, dtype=torch.float32, device=src.device)
            st = C.c_void_p(torch.cuda.current_stream(src.device).cuda_stream)
            _lib.check(_lib.lib().macx_images_to_nhwc(src.data_ptr(), src.shape[0], self.inDim, self.H * self.W, images.data_ptr(), st),
                       "macx_images_to_nhwc")
        elif images.dim() == 4:
            images = images.reshape(images.shape[0], self.H * self.W, self.inDim)
        keep = self.keep if train else 1.0
        return _StemFunction.apply(self, keep, fresh_seed(seed, train), int(b0), images, *self.tensors())
