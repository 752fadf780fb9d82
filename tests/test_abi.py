"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol
that include/sl_hip.h declares, and fails loudly (no fallback) when there is no GPU."""

import ctypes as C
import os
import re

import pytest

from conftest import ROOT, _have_gpu
from safe_learning_amd import _hip


def _declared_symbols():
    with open(os.path.join(ROOT, "include", "sl_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sl_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_match_binding():
    assert _declared_symbols() == sorted(_hip.EXPORTS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_hip.LIB_PATH):
        from safe_learning_amd._build import build
        build()
    lib = C.CDLL(_hip.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), "libslhip.so does not export %s" % name
    lib.sl_version.restype = C.c_int
    assert lib.sl_version() >= 100


def test_library_exports_nothing_else():
    """-fvisibility=hidden + SL_API: the dynamic symbol table defines the C ABI and nothing of the
    C++ internals (launchers, device stubs, sl_fail)."""
    import subprocess
    if not os.path.exists(_hip.LIB_PATH):
        from safe_learning_amd._build import build
        build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", _hip.LIB_PATH], text=True)
    defined = sorted(line.split()[-1] for line in out.splitlines() if " T " in line or " t " in line)
    assert defined == _declared_symbols(), sorted(set(defined) ^ set(_declared_symbols()))


def test_struct_layout_matches_header():
    """ctypes mirrors must have the C sizes (compiled check with the host compiler)."""
    import subprocess
    import tempfile
    src = r'''
    #include <stdio.h>
    #include "sl_hip.h"
    int main(void) {
        printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(sl_grid_desc), sizeof(sl_policy_desc),
               sizeof(sl_dynamics_desc), sizeof(sl_value_desc), sizeof(sl_lipschitz_desc),
               sizeof(sl_model_desc), sizeof(sl_key), sizeof(sl_sweep_result));
        return 0;
    }'''
    with tempfile.TemporaryDirectory() as tmp:
        c = os.path.join(tmp, "sizes.c")
        with open(c, "w") as f:
            f.write(src)
        exe = os.path.join(tmp, "sizes")
        subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    mirrors = [_hip.GridDesc, _hip.PolicyDesc, _hip.DynamicsDesc, _hip.ValueDesc,
               _hip.LipschitzDesc, _hip.ModelDesc, _hip.Key]
    assert sizes[:7] == [C.sizeof(m) for m in mirrors]
    assert sizes[7] == 8 * _hip.RESULT_WORDS


@pytest.mark.skipif(_have_gpu(), reason="only meaningful on a GPU-less host")
def test_no_cpu_fallback():
    import safe_learning_amd as sl
    import numpy as np
    grid = sl.GridWorld([[-1, 1]], 5)
    with pytest.raises(sl.HipEngineError):
        sl.Lyapunov(grid, sl.QuadraticFunction([[1.0]]), sl.LinearSystem((np.array([[1., 1.]]),)),
                    0.4, 0.3, 0.1, sl.LinearSystem((np.array([[-0.1]]),)))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "safe_learning_amd")
    for dirpath, _, files in os.walk(pkg):
        for name in files:
            if name.endswith((".py", ".hip", ".h", ".cpp")):
                with open(os.path.join(dirpath, name)) as f:
                    text = f.read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M), name
                assert "hostsim" not in text or name == "sl_model.h", name
