"""Regions computed BY THE REFERENCE'S OWN ``get_lyapunov_region`` (build container only).

``lyapunov.py:59-139`` is Python-2 / NumPy-1 era code; it runs unmodified from ``/root/reference``
once four things of that era are given back to it (like ``np.int`` and ``collections.Sequence`` in
``numpy_tf.load_reference``):

* ``tiebreaker.next()`` (``:103``): the module's ``itertools`` resolves to a namespace whose ``count``
  objects have a ``.next`` method (``product`` is the real one);
* ``np.bool`` (``:99``) = ``bool``;
* indexing with a LIST of index arrays, ``visited[np.split(neighbors.T, ndim)]`` (``:121, 127, 129``),
  which NumPy < 1.23 read as a tuple: the two arrays that are indexed this way - ``visited``
  (``np.zeros`` inside the function) and the value table (returned by the callable handed in) - are
  ``ndarray`` subclasses that turn such a key into the tuple it meant.

The value tables are piecewise-linear landscapes with several basins (the reference's own
``_Triangulation`` on a coarse grid with seeded vertex values, evaluated on a finer grid whose lines
do not coincide with the table's) and quadratic bowls; every scenario is checked to have no two
equal values among the cells the flood touches (the heap order of equal values depends on push
order, which only the sequential algorithm defines).  Output ``reference_regions.npz``: per scenario
the grid, the value table on it, the start node and the boolean region.  Data only.

    python tests/golden/make_reference_regions.py          (needs /root/reference)
"""

import itertools
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy_tf                                         # noqa: E402

OUT = os.path.join(HERE, "reference_regions.npz")


class LegacyIndexArray(np.ndarray):
    """ndarray that reads a list of index arrays as NumPy < 1.23 did: as a tuple."""

    @staticmethod
    def _key(key):
        return tuple(key) if isinstance(key, list) else key

    def __getitem__(self, key):
        return np.ndarray.__getitem__(self, self._key(key))

    def __setitem__(self, key, value):
        np.ndarray.__setitem__(self, self._key(key), value)


class _Count(object):
    def __init__(self):
        self._it = itertools.count()

    def next(self):
        return next(self._it)

    __next__ = next


def reference_region(ref, grid, values, init_node):
    """Run the reference's function on a value table given on ``grid`` (the reference's GridWorld)."""
    module = ref.lyapunov
    table = np.asarray(values, dtype=np.float64).reshape(-1, 1).view(LegacyIndexArray)

    def function(points):                              # "a tensorflow function": has .eval(feed_dict=)
        assert len(points) == len(table)
        return types.SimpleNamespace(eval=lambda feed_dict=None: table)
    function.feed_dict = {}
    plain_zeros, had_bool = np.zeros, hasattr(np, "bool")
    module.itertools = types.SimpleNamespace(product=itertools.product, count=_Count)
    np.bool = bool
    np.zeros = lambda *a, **k: plain_zeros(*a, **k).view(LegacyIndexArray)
    try:
        region = module.get_lyapunov_region(function, grid, tuple(int(v) for v in init_node))
    finally:
        np.zeros = plain_zeros
        module.itertools = itertools
        if not had_bool:
            del np.bool
    return np.asarray(region, dtype=bool)


def landscapes():
    """-> list of (name, limits, num_points, values[N], init_node)."""
    rng = np.random.default_rng(20260929)
    out = []

    def bumpy(name, limits, num_points, table_points, depth, seed, init=None):
        # a bowl plus seeded bumps on a coarse table, interpolated linearly (several basins)
        r = np.random.default_rng(seed)
        limits = np.asarray(limits, dtype=np.float64)
        axes = [np.linspace(lo, hi, n) for (lo, hi), n in zip(limits, table_points)]
        mesh = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1).reshape(-1, len(limits))
        centre = limits.mean(axis=1)
        vertex = ((mesh - centre) ** 2 / (limits[:, 1] - limits[:, 0]) ** 2).sum(axis=1)
        vertex = vertex + depth * r.random(len(mesh))
        out.append(dict(name=name, limits=limits, num_points=list(num_points),
                        table_points=list(table_points), vertex_values=vertex, init=init))

    bumpy("2d_bowl_smooth", [[-1, 1.03], [-0.97, 1]], (41, 37), (9, 8), 0.002, 1)
    bumpy("2d_bumps_shallow", [[-1, 1.03], [-0.97, 1]], (41, 37), (9, 8), 0.03, 2)
    bumpy("2d_bumps_deep", [[-1, 1.03], [-0.97, 1]], (61, 47), (11, 9), 0.12, 3)
    bumpy("2d_bumps_deep_b", [[-2, 1.0], [-0.5, 1.5]], (53, 64), (7, 12), 0.08, 4)
    bumpy("2d_start_off_minimum", [[-1, 1.03], [-0.97, 1]], (41, 37), (9, 8), 0.03, 2, init=(30, 9))
    # (a start node on the UPPER boundary ends the flood at once.  On the LOWER boundary the reference
    # misses it - `0 == next_node` compares 0 with the start TUPLE, :107 - and then indexes with -1,
    # which wraps to the other side of the grid: garbage that nothing here reproduces)
    bumpy("2d_start_on_boundary", [[-1, 1.03], [-0.97, 1]], (21, 17), (5, 4), 0.03, 5, init=(20, 8))
    bumpy("3d_bumps", [[-1, 1.03], [-0.97, 1], [-1, 1.01]], (17, 15, 16), (5, 4, 5), 0.05, 6)
    bumpy("3d_bowl", [[-1, 1.03], [-0.97, 1], [-1, 1.01]], (13, 15, 11), (4, 4, 4), 0.002, 7)
    bumpy("4d_bumps", [[-1, 1.03], [-0.97, 1], [-1, 1.01], [-0.99, 1]], (8, 7, 9, 8), (3, 3, 4, 3), 0.04, 8)
    bumpy("1d_bumps", [[-1, 1.03]], (101,), (13,), 0.05, 9)
    return out


def main():
    ref = numpy_tf.load_reference(examples=False)
    F = ref.functions
    arrays = {"_numpy_version": np.array(np.__version__)}
    names = []
    for spec in landscapes():
        name = spec["name"]
        grid = F.GridWorld(spec["limits"], spec["num_points"])
        table_grid = F.GridWorld(spec["limits"], spec["table_points"])
        tri = F._Triangulation(table_grid, vertex_values=spec["vertex_values"][:, None], project=True)
        mesh = np.meshgrid(*grid.discrete_points, indexing="ij")
        points = np.column_stack([col.ravel() for col in mesh])
        values = tri.build_evaluation(points)[:, 0]
        shaped = values.reshape(grid.num_points)
        init = spec["init"]
        if init is None:                                    # the grid's lowest interior cell
            inner = shaped[tuple(slice(1, -1) for _ in grid.num_points)]
            init = tuple(int(v) + 1 for v in np.unravel_index(np.argmin(inner), inner.shape))
        region = reference_region(ref, grid, values, init)
        # no ties among the cells the flood touched (region and its rim)
        touched = np.zeros_like(region)
        idx = np.argwhere(region)
        for off in itertools.product((-1, 0, 1), repeat=grid.ndim):
            nb = np.clip(idx + np.array(off), 0, np.array(grid.num_points) - 1)
            touched[tuple(nb.T)] = True
        vals = shaped[touched]
        assert len(np.unique(vals)) == len(vals), name
        names.append(name)
        arrays[name + "/limits"] = np.asarray(spec["limits"], dtype=np.float64)
        arrays[name + "/num_points"] = np.asarray(spec["num_points"], dtype=np.int64)
        arrays[name + "/table_points"] = np.asarray(spec["table_points"], dtype=np.int64)
        arrays[name + "/vertex_values"] = spec["vertex_values"]
        arrays[name + "/values"] = values
        arrays[name + "/init_node"] = np.asarray(init, dtype=np.int64)
        arrays[name + "/region"] = region
        on_rim = [bool((idx == 0).any() or (idx == np.array(grid.num_points) - 1).any())] if len(idx) else [False]
        print("%-24s cells %6d  region %5d  start %s  value range [%.3g, %.3g]" % (
            name, grid.nindex, int(region.sum()), init, values.min(), values.max()), on_rim)
    arrays["_names"] = np.array(names)
    np.savez_compressed(OUT, **arrays)
    print("wrote %s (%d arrays, %.1f KiB)" % (OUT, len(arrays), os.path.getsize(OUT) / 1024.0))


if __name__ == "__main__":
    main()
