"""-m gpu: randomised shapes -- the H2 (fp16-plane), the split-bf16 and the native f32-MFMA kernel families must agree on the whole cell
(final state and every gradient) to fp32 round-off for odd batch sizes, N from 1 cell to beyond one 208-row tile, both
dropout modes and all flag files (tools/mode_fuzz.py holds the generator)."""
import os
import random
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_split_and_native_kernels_agree_on_random_shapes(macx, dev):
    import mode_fuzz
    from helpers import default_gemm_mode
    rnd = random.Random(20260925)
    try:
        for case in range(10):
            name = rnd.choice(["args", "args1", "args3", "args4"])
            B = rnd.choice([1, 2, 3, 5, 8, 17]); S = rnd.randint(3, 12)
            N = rnd.choice([1, 7, 16, 30, 49, 100, 113, 196, 209, 250]); d = rnd.choice([128, 256])
            p = rnd.randint(1, 4); train = rnd.random() < 0.7
            b = mode_fuzz.run(0, name, B, S, N, d, p, train, case)
            for mode in (2, 1):
                a = mode_fuzz.run(mode, name, B, S, N, d, p, train, case)
                for k in a:
                    den = float(b[k].abs().max()) + 1e-20
                    if den > 1e-6:
                        assert float((a[k] - b[k]).abs().max()) / den < 2e-4, (mode, case, name, B, S, N, d, p, train, k)
    finally:
        macx._lib.lib().macx_gemm_mode(default_gemm_mode())
