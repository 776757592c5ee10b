import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def paper():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "paper_example.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def text_figure():
    """Figure 1 of the paper (text GCATCATA$: BWT, SA, LCP, LF), transcribed by tests/golden/make_text_example.py."""
    import json
    with open(os.path.join(ROOT, "tests", "golden", "text_example.json")) as f:
        return json.load(f)


def _gpu_unavailable_reason():
    """None when a GPU and the built HIP library are usable, else why not.  With `-m gpu` selected explicitly (the
    GPU box) nothing is skipped: a missing library or device must fail loudly there, not pass as skipped."""
    try:
        from gcsa2_amd import binding
        if not os.path.exists(binding.LIB_PATH):
            return f"{binding.LIB_PATH} is not built"
        count = binding.load_library().gcsa2_device_count()
        return None if count > 0 else "no HIP device visible"
    except Exception as e:                                   # library not loadable on this host
        return f"HIP library not usable: {e}"


def pytest_collection_modifyitems(config, items):
    if "gpu" in (config.getoption("-m") or ""):             # -m gpu / -m "not gpu": the caller has chosen
        return
    reason = None
    for item in items:
        if item.get_closest_marker("gpu") is not None:
            if reason is None:
                reason = _gpu_unavailable_reason() or ""
            if reason:
                item.add_marker(pytest.mark.skip(reason=f"gpu test: {reason}"))
