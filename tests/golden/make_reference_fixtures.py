"""Fixtures computed BY THE REFERENCE'S OWN CODE (build container only).

``/root/reference/safe_learning/functions.py`` cannot be imported as a package here (TensorFlow
1.x, gpflow 0.4 and ``future`` are missing, and ``lyapunov.py`` is Python-2 era), but its
``GridWorld`` (``functions.py:579-817``) and ``_Triangulation`` (``:981-1326``) are pure
NumPy/SciPy.  This script loads ``configuration.py``, ``utilities.py`` and ``functions.py`` from
the reference checkout as plain modules behind stand-ins for the missing imports, runs those two
classes on seeded inputs and stores inputs + outputs in ``reference_grid_triangulation.npz``.
``tests/test_oracle_golden.py`` then requires the oracle (``oracle/np_grid.py``,
``oracle/np_functions.py::Triangulation``) to reproduce every array bit for bit.  Nothing of the
reference (source or bytecode) is copied into the repository: the fixture is data.

The stand-ins: ``tensorflow`` / ``gpflow`` are objects whose attributes resolve to placeholders;
calling a placeholder is allowed while the modules are being imported (decorators, default
arguments, base classes of code that is never run) and RAISES once the import is finished, so no
number below can come from a stand-in.  Three TensorFlow calls are answered with inert objects
because ``Function.__init__`` (``functions.py:38-50``, the base class of ``_Triangulation``)
makes them for graph book-keeping: ``tf.get_default_graph()`` (a holder for the feed dict),
``tf.variable_scope(name)`` (a context manager) and ``tf.make_template(...)`` (never called
afterwards).  ``future.builtins`` maps to the Python-3 builtins, ``np.int`` to ``int``
(``functions.py:597, 909``).

Caveat: the unit-cell simplices come from SciPy's Qhull (``functions.py:1019-1023``); the fixture
records the SciPy version it was made with.  ``find_simplex`` starts each search at the previous
query's result, so on points that lie on a shared face the answer depends on the query history:
the test replays the same queries in the same order on fresh objects.

    python tests/golden/make_reference_fixtures.py          (needs /root/reference)
"""

import collections
import importlib.util
import os
import sys
import types

import numpy as np
import scipy
import scipy.interpolate    # noqa: F401  (imported before np.int is patched in)
import scipy.linalg         # noqa: F401
import scipy.sparse         # noqa: F401
import scipy.spatial        # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/safe_learning"
OUT = os.path.join(HERE, "reference_grid_triangulation.npz")


class StandInCalled(RuntimeError):
    """A tensorflow / gpflow stand-in was called after import: the result would not be the
    reference's."""


_armed = [False]


class _Placeholder(object):
    def __init__(self, name):
        self._name = name

    def __getattr__(self, key):
        if key.startswith("__") and key.endswith("__"):
            raise AttributeError(key)
        return _Placeholder(self._name + "." + key)

    def __call__(self, *args, **kwargs):
        if _armed[0]:
            raise StandInCalled(self._name)
        return _Placeholder(self._name + "()")


class _StandInModule(types.ModuleType):
    def __getattr__(self, key):
        if key.startswith("__") and key.endswith("__"):
            raise AttributeError(key)
        return _Placeholder(self.__name__ + "." + key)


class _Scope(object):
    original_name_scope = "scope/"

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def load_reference(gpflow_module=None):
    """-> the reference's ``functions`` module (GridWorld, _Triangulation usable).
    ``gpflow_module``: what ``import gpflow`` resolves to instead of the refusing stand-in
    (``numpy_gpflow.module()``, so that ``GPRCached`` inherits from a working ``GPR``)."""
    np.int = int
    tf = _StandInModule("tensorflow")
    tf.float64 = types.SimpleNamespace(as_numpy_dtype=np.float64)
    graph = types.SimpleNamespace()
    tf.get_default_graph = lambda: graph                       # functions.py:40
    tf.variable_scope = lambda name: _Scope()                  # functions.py:44
    tf.make_template = lambda *args, **kwargs: None            # functions.py:49 (never called)
    if gpflow_module is not None:
        gpflow = gpflow_module
    else:
        gpflow = _StandInModule("gpflow")
        gpflow.gpr = _StandInModule("gpflow.gpr")
        gpflow.gpr.GPR = type("GPR", (object,), {})            # base class of GPRCached (unused)
    future = types.ModuleType("future")
    future_builtins = types.ModuleType("future.builtins")
    future_builtins.zip, future_builtins.range, future_builtins.object = zip, range, object
    future_backports = types.ModuleType("future.backports")
    future_backports.OrderedDict = collections.OrderedDict
    sys.modules.update({"tensorflow": tf, "gpflow": gpflow, "future": future,
                        "future.builtins": future_builtins, "future.backports": future_backports})
    package = types.ModuleType("safe_learning")
    package.__path__ = [REF]
    sys.modules["safe_learning"] = package

    def load(name):
        spec = importlib.util.spec_from_file_location("safe_learning." + name,
                                                      os.path.join(REF, name + ".py"))
        module = importlib.util.module_from_spec(spec)
        sys.modules["safe_learning." + name] = module
        spec.loader.exec_module(module)
        setattr(package, name, module)
        return module

    package.config = load("configuration").Configuration()
    load("utilities")
    functions = load("functions")
    _armed[0] = True
    return functions


# (name, limits, num_points): 1-D .. 4-D, including the 64^4 spacing of BASELINE config C5
GRIDS = [
    ("1d", [[-1.0, 1.0]], [11]),
    ("2d", [[-1.0, 1.0], [-2.0, 3.0]], [9, 6]),
    ("2d_test", [[0.0, 1.0], [0.0, 1.0]], [3, 3]),              # the grid of test_functions.py:501
    ("3d", [[-1.0, 1.0], [0.0, 2.0], [-0.5, 0.5]], [5, 4, 6]),
    ("4d_64", [[-1.0, 1.0]] * 4, [64] * 4),
    ("4d_aniso", [[-1.0, 1.0], [-2.0, 2.0], [0.0, 1.0], [-3.0, -1.0]], [4, 5, 3, 6]),
]
NPOINTS = 1500


def big_table(nindex):
    """Vertex values of the large grids (not stored in the fixture)."""
    k = np.arange(nindex, dtype=np.int64)
    return (((k * 2654435761) % 1000003).astype(np.float64) / 1000003.0 - 0.5)[:, None]


def query_points(rng, limits, num_points):
    """Seeded queries: inside, outside the limits (projection / clipping), exactly on vertices and
    on grid lines (the `%` wrap-around of functions.py:1121), near the upper boundary."""
    limits = np.asarray(limits, dtype=np.float64)
    d = len(limits)
    span = limits[:, 1] - limits[:, 0]
    inside = limits[:, 0] + rng.random((NPOINTS // 2, d)) * span
    outside = limits[:, 0] - 0.3 * span + rng.random((NPOINTS // 6, d)) * 1.6 * span
    axes = [np.linspace(lo, hi, n) for (lo, hi), n in zip(limits, num_points)]
    pick = np.stack([rng.integers(0, n, NPOINTS // 6) for n in num_points], axis=1)
    vertices = np.stack([axes[k][pick[:, k]] for k in range(d)], axis=1)
    lines = limits[:, 0] + rng.random((NPOINTS // 6, d)) * span
    col = rng.integers(0, d, len(lines))
    lines[np.arange(len(lines)), col] = vertices[:len(lines)][np.arange(len(lines)), col]
    return np.ascontiguousarray(np.concatenate((inside, outside, vertices, lines)))


def main():
    functions = load_reference()
    rng = np.random.default_rng(20260927)
    out = {"_scipy_version": np.array(scipy.__version__), "_names": np.array([g[0] for g in GRIDS])}
    for name, limits, num_points in GRIDS:
        grid = functions.GridWorld(limits, num_points)
        points = query_points(rng, limits, num_points)
        indices = rng.integers(0, grid.nindex, 400)
        rectangles = rng.integers(0, grid.nrectangles, 400)
        out[name + "/limits"] = np.asarray(limits, dtype=np.float64)
        out[name + "/num_points"] = np.asarray(num_points, dtype=np.int64)
        out[name + "/points"] = points
        out[name + "/indices"] = indices
        out[name + "/rectangles"] = rectangles
        out[name + "/unit_maxes"] = np.asarray(grid.unit_maxes)
        out[name + "/index_to_state"] = grid.index_to_state(indices)
        out[name + "/state_to_index"] = np.asarray(grid.state_to_index(points))
        out[name + "/state_to_rectangle"] = np.asarray(grid.state_to_rectangle(points))
        out[name + "/rectangle_to_state"] = grid.rectangle_to_state(rectangles)
        out[name + "/rectangle_corner_index"] = np.asarray(grid.rectangle_corner_index(rectangles))
        # (GridWorld.all_points itself hands np.column_stack a generator, functions.py:635, which
        # NumPy 2 rejects; the vertices below are the same meshgrid of the reference's own
        # discrete_points, functions.py:612-634)
        vertices = None
        if grid.nindex <= 4096:
            mesh = np.meshgrid(*grid.discrete_points, indexing="ij")
            vertices = np.column_stack([col.ravel() for col in mesh])
            out[name + "/all_points"] = vertices
        # value tables: seeded and stored (two output columns) on the small grids; on 64^4 a
        # formula the test re-evaluates exactly (integer arithmetic, one correctly rounded division)
        if grid.nindex > 100000:
            values = big_table(grid.nindex)
        else:
            values = rng.standard_normal((grid.nindex, 2))
            out[name + "/vertex_values"] = values
        for project in (False, True):
            tag = "%s/project%d/" % (name, int(project))
            tri = functions._Triangulation(grid, vertex_values=values, project=project)
            out[tag + "unit_simplices"] = np.asarray(tri.unit_simplices)
            out[tag + "hyperplanes"] = np.asarray(tri.hyperplanes)
            # query order (replayed by the test): find_simplex, evaluation, gradient
            out[tag + "find_simplex"] = np.asarray(tri.find_simplex(points))
            out[tag + "values"] = tri.build_evaluation(points)
            out[tag + "gradient"] = tri.gradient(points)
            if vertices is not None:
                # a table evaluated at its own vertices: the `%` wrap-around makes the reference
                # return something else than the vertex value at some of them
                out[tag + "values_at_vertices"] = tri.build_evaluation(vertices)
    np.savez_compressed(OUT, **out)
    print("wrote %s (%d arrays, %.1f KiB), scipy %s"
          % (OUT, len(out), os.path.getsize(OUT) / 1024.0, scipy.__version__))


if __name__ == "__main__":
    main()
