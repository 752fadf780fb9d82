"""pytest configuration: the ``gpu`` marker and repo-root imports."""

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(GOLDEN_DIR, "reference_known_answers.json")) as f:
        return json.load(f)


def pytest_sessionfinish(session, exitstatus):
    """What the table-lookup parity tests excluded (tests/exclusions.py), kept with the GPU logs."""
    import json
    try:
        import exclusions
    except ImportError:
        return
    if exclusions.LOG:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_exclusions.json"), "w") as handle:
            json.dump(exclusions.LOG, handle, indent=1)
