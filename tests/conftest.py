import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def lo():
    """The product package (linearoperators.jl_amd/), loaded through __graft_entry__."""
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def dev():
    import torch
    return torch.device("cuda", 0)


@pytest.fixture(scope="session")
def kat():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "kat_reference_tests.json")) as f:
        return json.load(f)["cases"]


def pytest_sessionfinish(session, exitstatus):
    out = os.environ.get("MXLO_ENVELOPE_OUT")
    if not out:
        return
    try:
        import json
        import tolerances
        with open(out, "w") as f:
            json.dump(tolerances.envelope(), f, indent=1)
    except Exception as e:          # pragma: no cover
        print(f"[mxlo] could not write {out}: {e!r}")
