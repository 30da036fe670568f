"""TEST INFRASTRUCTURE.  fp64 restatement of the reference's training op (model.py:639-669):
tf.clip_by_global_norm -> tf.train.AdamOptimizer (TF1 formulation with lr_t and epsilon-hat) ->
tf.train.ExponentialMovingAverage (applied after the Adam update, the order an eager run of addTrainingOp gives).

PINNED: MACnet.addOptimizerOp / computeGradients / addTrainingOp run unmodified on the eager TF-1.x stand-in
(tests/ref_exec.run_reference_training) for five steps that straddle the clip threshold; this restatement, fed the gradients the
reference computed, reproduces its variables, both Adam moments, the EMA shadows and the global norm to 1e-12
(tests/test_reference_exec.py live, tests/test_reference_golden.py from tests/golden/reference/training_steps.npz).  The
stand-in's AdamOptimizer / ExponentialMovingAverage are themselves restatements of TF's (training/adam.py, ApplyAdam,
moving_averages.py) -- TensorFlow cannot be installed here.

ASSUMPTION, not reference behaviour: the EMA reads the POST-update weights.  In the reference `maintainAveragesOp =
expMovingAverage.apply(...)` is built OUTSIDE `tf.control_dependencies([train])` (model.py:657-663: the `with` block only wraps
the `tf.group` that returns both), so in a real TF1 graph nothing orders the shadow update after the Adam update: within one
session.run either may run first, and the shadow may lag the weights by one step (decay 0.999: a 1e-3-relative difference in the
shadow, none in the weights).  The eager stand-in executes ops in program order -- apply_gradients, then apply -- which is the
order pinned here and the order macx_adam_ema_step implements; `ema_pre_update=True` gives the other legal order."""
import numpy as np


def adam_ema_step(p, g, m, v, ema, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, clip=8.0, decay=0.999, ema_pre_update=False):
    p, g, m, v = [np.asarray(x, dtype=np.float64) for x in (p, g, m, v)]
    p_before = p
    norm = np.sqrt((g * g).sum())
    if clip and clip > 0:
        g = g * clip / max(norm, clip)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    lr_t = lr * np.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    p = p - lr_t * m / (np.sqrt(v) + eps)
    if ema is not None:
        ema = np.asarray(ema, dtype=np.float64)
        ema = ema - (1 - decay) * (ema - (p_before if ema_pre_update else p))
    return p, m, v, ema, norm
