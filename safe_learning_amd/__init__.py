"""safe_learning_amd - MI355X-native engine for safe_learning's Lyapunov sweep.

Same names as ``safe_learning`` for the hot path (``GridWorld``, ``Lyapunov``,
``PolicyIteration`` and the function classes they take); the arithmetic runs in hand-written
HIP kernels for gfx950 behind the C ABI of ``include/sl_hip.h``.  There is no CPU fallback.
"""

from .configuration import config
from .functions import *          # noqa: F401,F403
from .lyapunov import *           # noqa: F401,F403
from .reinforcement_learning import *   # noqa: F401,F403
from . import utilities, distributed, kernels
from ._hip import HipEngineError
