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
    if os.environ.get("MACX_DBG"):          # debugging aid: kernel-selection / timing bits of macx_debug_set(1, .)
        m._lib.lib().macx_debug_set(1, int(os.environ["MACX_DBG"]))
    if os.environ.get("MACX_CHAIN"):        # 0: the read unit's forward products as four launches instead of the fused chain kernel
        m._lib.lib().macx_debug_set(4, int(os.environ["MACX_CHAIN"]))
    if os.environ.get("MACX_SB_DEFER"):     # 0: the S_b contraction once per step
        m._lib.lib().macx_debug_set(5, int(os.environ["MACX_SB_DEFER"]))
    if os.environ.get("MACX_STEM_CHAIN"):   # 0: the stem's convolutions on kb_gemm3h_kernel instead of kb_conv_chain_kernel
        m._lib.lib().macx_debug_set(9, int(os.environ["MACX_STEM_CHAIN"]))
    if os.environ.get("MACX_CHAIN_KV"):     # K-loop variant of the chain kernels (macx_chain_h2.hip.h: ChainCtx::kloop)
        m._lib.lib().macx_debug_set(7, int(os.environ["MACX_CHAIN_KV"]))
    if os.environ.get("MACX_SB_WIDE"):      # 0: the deferred S_b contraction on the 128 x 128 kernel
        m._lib.lib().macx_debug_set(8, int(os.environ["MACX_SB_WIDE"]))
    if os.environ.get("MACX_OVERLAP"):      # 0: no side queue in the backward pass
        m._lib.lib().macx_debug_set(6, int(os.environ["MACX_OVERLAP"]))
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
