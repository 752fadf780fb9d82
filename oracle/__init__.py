"""CPU oracle: a NumPy float64 restatement of befelix/safe_learning's Lyapunov sweep.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  ``safe_learning_amd`` (the product) never imports
it and has no CPU fallback: without the HIP extension the product raises.

What it restates (reference file:line, relative to the upstream checkout):

* ``oracle.np_utilities``  - ``safe_learning/utilities.py:224-249`` (batchify),
  ``:327-357`` (dlqr).
* ``oracle.np_grid``    - ``safe_learning/functions.py:579-817`` (GridWorld).
* ``oracle.np_functions``  - ``safe_learning/functions.py:254-354`` (FunctionStack,
  Saturation), ``:357-546`` (GPRCached / GaussianProcess), ``:981-1370``
  (_Triangulation), ``:1513-1583`` (QuadraticFunction, LinearSystem) and
  ``examples/utilities.py:48-104, 144-437`` (LyapunovNetwork, InvertedPendulum,
  CartPole).
* ``oracle.np_lyapunov`` - ``safe_learning/lyapunov.py:22-56, 142-606`` and ``:609-797``
  (``perturb_actions``, ``get_safe_sample``).
* ``oracle.np_rl`` - ``safe_learning/reinforcement_learning.py:26-140,
  213-279``.

* ``oracle/cpu_sweep.cpp`` + ``oracle.cpu_sweep`` (round 6) - the batch loop of
  ``safe_learning/lyapunov.py:517-529`` for the shared-kernel RBF configurations in C++ / OpenMP: the
  CPU BASELINE of ``bench.py`` at host strength.  Measurement only: it checks nothing,
  the NumPy code below checks it (``tests/test_cpu_sweep.py``).

Third-party arithmetic that is not in the upstream checkout: the RBF kernel and
``GPR.build_predict`` of gpflow==0.4.0 (pinned in the reference's
``requirements.txt:3``).  The published algorithm is restated in
``oracle.np_functions.RBF`` and pinned by the reference's own known-answer test
``safe_learning/tests/test_functions.py:237-261``.  The kernels the reference's notebooks use
instead (``Matern32``, ``Linear``, ``Add``, ``Prod`` with ``active_dims``;
``examples/inverted_pendulum.ipynb:152-158``) are restated from the same published sources in
``oracle.np_functions`` as well; no test of the reference holds a number for them: their FORMULAS
are **parity unpinned** (three independent restatements - this one, the gpflow stand-in of the
fixtures, the engine - are compared with each other), the reference's arithmetic around them is
pinned by running its ``GPRCached`` / ``FunctionStack`` / ``Lyapunov`` / ``PolicyIteration`` on such
models (``tests/golden/reference_gp_kernels.npz`` and the ``notebook_kernels*`` scenarios).

Pinning, five layers:

1. the literal known-answer values of the reference's own tests, transcribed in
   ``tests/golden/reference_known_answers.json`` (``tests/test_oracle_golden.py``);
2. arrays computed by the reference's own ``GridWorld`` / ``_Triangulation`` (pure
   NumPy/SciPy) run in the build container (``tests/golden/make_reference_fixtures.py``);
3. the reference's hot path run END TO END in the build container: its ``lyapunov.py``,
   ``reinforcement_learning.py``, function classes and Euler models executed unmodified
   behind ``tests/golden/numpy_tf.py``, a deferred-NumPy stand-in for the TensorFlow ops
   they request (TensorFlow 1.x is absent).  The committed fixtures
   (``reference_safe_sets.npz``, ``reference_policy_iteration.npz``,
   ``reference_functions.npz``) are reproduced by this oracle bit for bit
   (``tests/test_oracle_reference_*.py``): safe sets, ``c_max``, refinement arrays,
   samples, value and policy tables, per-class outputs;
4. the GP posterior by the reference's own ``GPRCached`` / ``GaussianProcess`` /
   ``FunctionStack`` (``functions.py:254-307, 357-546``) at n = 3 ... 1024 training points,
   executed behind ``tests/golden/numpy_gpflow.py`` (the restatement of the gpflow 0.4.0
   pieces underneath, itself checked by the reference's four GP tests):
   ``reference_gp_posterior.npz``, reproduced within 8 eps cond(K)
   (``tests/test_oracle_reference_gp.py``);
5. ``get_lyapunov_region`` by the reference's own function (``lyapunov.py:59-139``, Python-2 /
   NumPy-1 code run behind three era shims): ``reference_regions.npz``, reproduced bit for bit
   (``tests/test_oracle_reference_regions.py``).

What stays "parity unpinned": the order of equal-valued cells in ``update_safe_set`` and in the
flood of ``get_lyapunov_region`` (the reference uses NumPy's default argsort / the push order of a
heap; the oracle fixes ascending (value, flat index)), and the reference's behaviour for a flood
started on a lower grid boundary (it wraps around the grid).

Canonical arithmetic: small linear-algebra forms (policy, linear dynamics,
quadratic V, thresholds) are evaluated left-to-right, one IEEE-754 rounding per
multiply and per add (no FMA, no BLAS), see ``oracle.np_functions.ordered_matmul``.
TensorFlow/Eigen's summation order in the reference is not specified, so this is
a definition, and it is what the HIP kernels reproduce bit-for-bit
(compiled with ``-ffp-contract=off``).
"""

from .np_utilities import batchify, dlqr
from .np_grid import GridWorld, DimensionError
from .np_functions import (ordered_matmul, QuadraticFunction, LinearSystem, Saturation,
                        RBF, GPRCached, GaussianProcess, FunctionStack, Triangulation,
                        InvertedPendulum, CartPole, LyapunovNetwork, NeuralNetwork, AbsFunction,
                        Norm1Function, NegatedFunction, ConstantPolicy,
                        TriangulationGradient)
from .np_lyapunov import (Lyapunov, smallest_boundary_value, config, get_safe_sample,
                          perturb_actions, unique_rows, get_lyapunov_region)
from .np_rl import PolicyIteration
