"""Unit-cell triangulations of the reference's ``Triangulation`` (``functions.py:1019-1023``), frozen.

The reference triangulates ONE unit cell with SciPy/Qhull (``spatial.Delaunay`` of the corners
``cartesian(*np.diag(unit_maxes))``) and repeats it over the grid.  The corners of a box are
cospherical, so which of the many valid Delaunay triangulations comes back - and in which order
the simplices and their vertices are listed: the order decides ties between simplices
(``functions.py:1103-1158``) and the rounding of the barycentric weights (``:1090-1101``) - is a
property of the Qhull build, not of the mathematics.  No reference test pins d = 4 (d = 2 is
pinned by ``tests/test_functions.py:490-529``, d = 3 by its simplex count, ``:555``).

These tables are what scipy 1.15.3 / Qhull returns for every ``unit_maxes`` tried (checked for 60
random cells per dimension; ``tests/golden/unit_cell_triangulations.json`` holds the same data with
its provenance, ``tests/test_host_logic.py`` compares the two and the SciPy of the machine it runs
on).  ``Triangulation`` uses them for d = 2, 3, 4 instead of calling Qhull, so that another SciPy
cannot move the product and the oracle together unnoticed; d > 4 still asks Qhull.

Entry: dimension -> simplices in Qhull's order, each a list of d + 1 corner codes (bit k set =
coordinate k at ``unit_maxes[k]``, else 0), in Qhull's vertex order."""

UNIT_CELL_SIMPLICES = {
    2: [
        [3, 2, 1],
        [1, 2, 0],
    ],
    3: [
        [7, 3, 1, 0],
        [7, 5, 1, 0],
        [7, 4, 6, 0],
        [7, 2, 6, 0],
        [7, 2, 3, 0],
        [7, 4, 5, 0],
    ],
    4: [
        [13, 5, 3, 1, 0],
        [13, 9, 3, 1, 0],
        [13, 4, 12, 6, 0],
        [13, 11, 15, 10, 3],
        [13, 6, 5, 3, 0],
        [13, 2, 10, 6, 0],
        [13, 8, 12, 10, 0],
        [13, 10, 9, 3, 0],
        [13, 10, 6, 14, 3],
        [13, 10, 6, 14, 0],
        [13, 7, 15, 6, 3],
        [13, 12, 6, 14, 0],
        [13, 12, 10, 14, 0],
        [13, 15, 6, 14, 3],
        [13, 15, 10, 14, 3],
        [13, 2, 10, 6, 3],
        [13, 2, 6, 3, 0],
        [13, 2, 10, 3, 0],
        [13, 4, 6, 5, 0],
        [13, 8, 10, 9, 0],
        [13, 7, 6, 5, 3],
        [13, 11, 10, 9, 3],
    ],
}
