import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")))


@pytest.fixture(scope="session")
def gpu():
    """Session-wide check that the HIP library is loaded and a device is present; GPU tests fail loudly otherwise."""
    try:
        import torch   # noqa: F401 -- torch's HIP runtime has to be loaded before libkkdirect's, or torch later sees no device
    except ImportError:
        pass
    import kkdirect
    kkdirect.load_library()
    if kkdirect.device_count() < 1:
        pytest.fail("GPU test on a machine without a HIP device (libkkdirect has no CPU fallback)")
    return kkdirect


def pytest_sessionfinish(session, exitstatus):
    """measured same-stream relMSE values of the GPU image tests -> gpurun_out/parity_same_stream.json (the bounds of those tests are
    set from these; profiles/ holds the committed copy)"""
    import json
    import helpers
    if helpers.MEASURED:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "parity_same_stream.json"), "w") as f:
            json.dump(helpers.MEASURED, f, indent=1, sort_keys=True)
