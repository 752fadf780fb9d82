"""Development build with phase stamps in k_bellman4s (-DSL_B4S_TIMING): safe_learning_amd/libslhip_timing.so.

The stamps change the generated code (the audits of the shipped listing do not apply), so the
library is written next to the shipped one and only used through SL_LIB_PATH:

    python tools/build_timing.py
    SL_LIB_PATH=$PWD/safe_learning_amd/libslhip_timing.so python bench.py --config C5 --steps 2 ...
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SL_EXTRA_FLAGS"] = (os.environ.get("SL_EXTRA_FLAGS", "") + " -DSL_B4S_TIMING").strip()

from safe_learning_amd import _build                                 # noqa: E402

print(_build.build(force=True, run_audits=False,
                   lib=os.path.join(ROOT, "safe_learning_amd", "libslhip_timing.so")))
