"""Development builds of sl_bellman4.hip next to the shipped library (never loaded by default):

    python tools/build_dev.py timing -DSL_B4S_TIMING       # phase stamps of k_bellman4s, printed by the kernel
    python tools/build_dev.py skip3 -DSL_B4S_SKIP=3         # phases left out (results wrong): what does each cost
    SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_timing.so python bench.py --config C5 --steps 2 ...

The flags change the generated code, so the audits of the shipped listing do not apply; the other
translation units are taken from the shipped build (safe_learning_amd/build/).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
name, flags = sys.argv[1], sys.argv[2:]

from safe_learning_amd import _build                                 # noqa: E402

_build.build()                                                       # the shipped objects exist (plain flags)
os.environ["SL_EXTRA_FLAGS"] = " ".join([os.environ.get("SL_EXTRA_FLAGS", "")] + flags).strip()
print(_build.build(force=True, run_audits=False, only=["sl_bellman4"],
                   lib=os.path.join(ROOT, "safe_learning_amd", "libslhip_%s.so" % name)))
