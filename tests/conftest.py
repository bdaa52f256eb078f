import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "mm-vqa-healthcare_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size CPU oracle runs (tens of seconds)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _fused_cross_attention_at_test_batch_sizes(request):
    """The product takes the fused cross-attention TRAINING path from a per-call batch of 96 on (ops.XATTN_TRAIN_MIN_BATCH:
    below it the composition is faster).  The GPU tests run batches of 2-5, so they lower the threshold to keep that path
    under test; tests/test_gpu_xattn.py::test_fused_training_path_batch_rule checks the default rule itself."""
    if "gpu" not in request.keywords:
        yield
        return
    from m3ae_amd import ops
    old = ops.XATTN_TRAIN_MIN_BATCH
    ops.XATTN_TRAIN_MIN_BATCH = 0
    try:
        yield
    finally:
        ops.XATTN_TRAIN_MIN_BATCH = old
