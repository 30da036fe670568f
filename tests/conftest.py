import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


GEMM_MODES = {"native": 0, "split": 1, "h2": 2}


def default_gemm_mode():
    return GEMM_MODES.get(os.environ.get("MACX_GEMM", "h2"), 2)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def macx():
    import macx as m
    if os.environ.get("MACX_GEMM"):         # run the whole suite on one kernel family: native | split | h2 (default)
        m._lib.lib().macx_gemm_mode(GEMM_MODES[os.environ["MACX_GEMM"]])
    # A/B routes for a whole session: the per-call tuning table (macx_opts.tune) of every cell frozen from here on
    for env, key in (("MACX_DBG", "phase_mask"),        # debugging aid: timing bits (results are wrong under a non-zero mask)
                     ("MACX_CHAIN", "chain"),           # 0: the read unit's products as separate launches instead of the chain kernels
                     ("MACX_SB_DEFER", "sb_defer"),     # 0: the S_b contraction once per step
                     ("MACX_CHAIN_KV", "chain_kv"),     # 0: the chain kernels' K loop with fragment requests in front of a slice
                     ("MACX_SB_WIDE", "sb_wide")):      # 0: the deferred S_b contraction on the 128 x 128 kernel
        if os.environ.get(env):
            m.options.SESSION_TUNE[key] = int(os.environ[env])
    return m


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


@pytest.fixture
def native_gemm(macx):
    """Run one test on the native f32-MFMA knowledge-base GEMM (the default is the H2 fp16-plane family); restored afterwards."""
    L = macx._lib.lib()
    L.macx_gemm_mode(0)
    yield
    L.macx_gemm_mode(default_gemm_mode())


@pytest.fixture
def split_gemm(macx):
    """Run one test on the split-bf16 (3 x bf16, six MFMA terms) family; restored afterwards."""
    L = macx._lib.lib()
    L.macx_gemm_mode(1)
    yield
    L.macx_gemm_mode(default_gemm_mode())


def pytest_sessionfinish(session, exitstatus):
    """observed parity errors of this session -> gpurun_out/parity_margins.json (see helpers.check_margin)"""
    try:
        import helpers
        out = os.path.join(ROOT, "gpurun_out")
        if helpers._MARGINS:
            os.makedirs(out, exist_ok=True)
            helpers.dump_margins(os.path.join(out, "parity_margins.json"))
    except Exception:
        pass
