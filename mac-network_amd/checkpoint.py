"""Weights in and out under the reference's TF variable names (SURVEY.md 8b "parameters", 8f row 4 "ckpt formats").

Two carriers: the reference's own V2 checkpoint files (`load_tf_checkpoint` / `save_tf_checkpoint`, read and written by
tf_bundle.py without TensorFlow), and an .npz of the same name -> array mapping, which a maintainer can also dump in a
TF environment

    np.savez("macx_weights.npz", **{v.name: sess.run(v) for v in tf.global_variables()})      # main.py:185-193 restores them

Names are `macModel/<scope>/...:0` (model.py:774); EMA shadows (model.py:659-663, main.py:172-174 `emaSaver`) are
`<name>/ExponentialMovingAverage`.  `load_reference` accepts names with or without the `macModel/` prefix and the `:0`
suffix; `reference_state_dict` / `save_npz` emit the canonical form.  Optimizer slots (Adam m, v) are not carried
over: the reference's `saver` stores them under `trainAddOptimizer/...` names that depend on graph construction order.
"""
import numpy as np
import torch

PREFIX = "macModel/"
EMA_SUFFIX = "/ExponentialMovingAverage"


def _modules(net):
    mods = []
    for attr in ("enc", "stem", "cell", "out"):
        m = getattr(net, attr, None)
        if m is not None:
            mods.append(m)
    if not mods:          # a single component (MACCellParams, Stem, ...)
        mods = [net]
    return mods


def reference_state_dict(net):
    """{TF variable name: tensor (CPU)} for a MACNet / MACNetCore / single component."""
    out = {}
    for m in _modules(net):
        for k, v in m.to_reference_dict().items():
            out[PREFIX + k + ":0"] = v.detach().cpu()
    return out


def _normalise(d):
    out = {}
    for k, v in d.items():
        k = k[:-2] if k.endswith(":0") else k
        k = k[len(PREFIX):] if k.startswith(PREFIX) else k
        out[k] = v
    return out


@torch.no_grad()
def load_reference(net, d, use_ema=False, strict=True):
    """Copy weights named as in the reference into `net`.  use_ema: take the `/ExponentialMovingAverage` shadow of each
    variable when present (what the reference evaluates with, main.py:172-174).  strict: every variable of `net` must be
    present with the reference's shape; extra keys (optimizer slots, baseline variables) are ignored."""
    src = _normalise(d)
    if use_ema:
        for k in list(src):
            if k.endswith(EMA_SUFFIX):
                src[k[:-len(EMA_SUFFIX)]] = src[k]
    missing, bad = [], []
    lazy = [m for m in _modules(net) if getattr(m, "lazy_scopes", None)]
    for m in lazy:
        # a module whose variables only appear on first use (generic.GenericParams before the first forward pass) has nothing to
        # ask for yet: hand it every model variable stored under its scope, so that the first forward finds them instead of
        # drawing fresh ones
        scoped = {k: v for k, v in src.items() if k.startswith(tuple(m.lazy_scopes)) and not k.endswith(EMA_SUFFIX)
                  and "/Adam" not in k and "/ExponentialMovingAverage" not in k}
        have_now = set(m.to_reference_dict())
        new = {k: v for k, v in scoped.items() if k not in have_now}
        if new:
            m.load_reference_dict(new)
        if strict and not m.tensors():
            raise KeyError("no variable under %s in the source: the lazily built module would start from random weights"
                           % (m.lazy_scopes,))
    for m in _modules(net):
        want = m.to_reference_dict()
        for k, w in want.items():
            if k not in src:
                missing.append(k)
            elif tuple(np.shape(src[k])) != tuple(w.shape):
                bad.append((k, tuple(np.shape(src[k])), tuple(w.shape)))
    if bad:
        raise ValueError("shape mismatch (name, given, expected): %s" % bad[:5])
    if missing and strict:
        raise KeyError("missing variables: %s%s" % (missing[:5], " ..." if len(missing) > 5 else ""))
    for m in _modules(net):
        want = m.to_reference_dict()
        have = {k: (src[k] if k in src else want[k]) for k in want}
        if hasattr(m, "load_reference_dict"):
            m.load_reference_dict(have)
        else:                                   # Stem / OutputClassifier: field -> name tables
            names = m.REF_NAMES if hasattr(m, "REF_NAMES") else __import__(m.__module__, fromlist=["REF_NAMES"]).REF_NAMES
            for f, n in names.items():
                getattr(m, f).copy_(torch.as_tensor(have[n]).to(getattr(m, f).dtype).reshape(getattr(m, f).shape))
    return missing


def save_npz(path, net, ema_tensors=None):
    """Write the reference-named weights (and, when given, the EMA shadows in `net.tensors()` order) to an .npz."""
    arrs = {k: v.numpy() for k, v in reference_state_dict(net).items()}
    if ema_tensors is not None:
        live = [p.detach().clone() for p in net.tensors()]
        try:
            with torch.no_grad():
                for p, e in zip(net.tensors(), ema_tensors):
                    p.copy_(e.reshape(p.shape))
            for k, v in reference_state_dict(net).items():
                arrs[k[:-2] + EMA_SUFFIX + ":0"] = v.numpy()
        finally:
            with torch.no_grad():
                for p, l in zip(net.tensors(), live):
                    p.copy_(l)
    np.savez(path, **arrs)
    return sorted(arrs)


def load_npz(path, net, use_ema=False, strict=True):
    with np.load(path) as z:
        return load_reference(net, {k: z[k] for k in z.files}, use_ema=use_ema, strict=strict)


def load_tf_checkpoint(prefix, net, use_ema=False, strict=True, verify=True):
    """Restore `net` from a TensorFlow V2 checkpoint written by the reference's saver (main.py:236-262), e.g.
    prefix = "weights/clevrExperiment/weights25.ckpt".  Returns the list of missing variables (empty under strict)."""
    from . import tf_bundle
    return load_reference(net, tf_bundle.read_checkpoint(prefix, verify=verify), use_ema=use_ema, strict=strict)


def save_tf_checkpoint(prefix, net):
    """Write the reference-named weights as a V2 checkpoint the reference's `saver.restore` (main.py:185-193) can read
    for every variable it finds (variable names carry no ":0" in a checkpoint)."""
    from . import tf_bundle
    return tf_bundle.write_checkpoint(prefix, {k[:-2]: v.numpy() for k, v in reference_state_dict(net).items()})
