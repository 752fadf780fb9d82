"""Run the REFERENCE'S OWN TEST SUITE on the reference's code behind ``numpy_tf`` (build container
only): a check of the stand-in, not of the product.

``/root/reference/safe_learning/tests/*.py`` are collected from where they lie (nothing is written
there: no bytecode, no pytest cache; pytest sees symbolic links in a scratch directory under
``build/``) with the reference's modules loaded by ``numpy_tf.load_reference``.  Tests that need
gpflow or TensorFlow features outside the hot path (optimisers, ``tf.gradients``, the Xavier
initialiser, GP models) cannot pass and are reported with the stand-in that stopped them.  The
outcome per test is written to ``tests/golden/reference_test_results.json``
(``tests/test_oracle_reference_functions.py`` checks that file).  NumPy-1 aliases the tests use:
``np.float``, ``np.math``.

    python tests/golden/run_reference_tests.py          (needs /root/reference)
"""

import json
import math
import os
import re
import shutil
import sys
import tempfile

sys.dont_write_bytecode = True

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy_tf                                         # noqa: E402

OUT = os.path.join(HERE, "reference_test_results.json")


class _Recorder(object):
    def __init__(self):
        self.outcomes = {}

    def pytest_runtest_logreport(self, report):
        if report.when == "call" or (report.when == "setup" and report.outcome != "passed"):
            reason = ""
            if report.outcome != "passed":
                reason = str(report.longrepr).strip().splitlines()[-1][:200]
                reason = re.sub(r"/\S*reference_tests_\w+/", "", reason)
            nodeid = report.nodeid[report.nodeid.index("test_"):]     # without the scratch directory
            self.outcomes[nodeid] = [report.outcome, reason]


def main():
    ref = numpy_tf.load_reference(examples=False)
    tf = sys.modules["tensorflow"]
    numpy_tf.install_test_extras(tf)
    package = sys.modules["safe_learning"]
    for module in (ref.functions, ref.lyapunov, ref.reinforcement_learning,
                   sys.modules["safe_learning.utilities"]):
        for name in getattr(module, "__all__", []):
            setattr(package, name, getattr(module, name))
    np.float = float                                     # test_rl.py:151 (NumPy < 1.24)
    np.math = math                                       # test_functions.py:555 (NumPy < 2)
    recorder = _Recorder()
    source = os.path.join(numpy_tf.REFERENCE_ROOT, "safe_learning", "tests")
    # symbolic links in a scratch directory: collected from /root/reference directly, pytest would
    # import the real safe_learning/__init__.py as the tests' parent package
    scratch = tempfile.mkdtemp(prefix="reference_tests_", dir=os.path.join(ROOT, "build"))
    try:
        for name in sorted(os.listdir(source)):
            if name.startswith("test_") and name.endswith(".py"):
                os.symlink(os.path.join(source, name), os.path.join(scratch, name))
        pytest.main([scratch, "-q", "-p", "no:cacheprovider", "--rootdir=" + scratch,
                     "-c", "/dev/null", "--import-mode=importlib",
                     "--tb=short" if "-v" in sys.argv else "--tb=no"], plugins=[recorder])
    finally:
        shutil.rmtree(scratch)
    counts = {}
    for outcome, _ in recorder.outcomes.values():
        counts[outcome] = counts.get(outcome, 0) + 1
    with open(OUT, "w") as handle:
        json.dump({"counts": counts, "tests": recorder.outcomes}, handle, indent=1, sort_keys=True)
    print(counts)


if __name__ == "__main__":
    main()
