"""``initial_safe_set`` and ``safe_set`` are plain NumPy arrays in the reference, read afresh by every
``update_safe_set`` (``lyapunov.py:500-510``) and written in place (``:598-606``).  The engine keeps
bit-packed copies on the device; these tests pin how the two stay consistent (needs an MI355X):

* an initial mask edited IN PLACE (same object, same number of set cells) is noticed - a drop-in
  must not answer with the safe set of the old mask;
* a read-only mask is trusted by identity (the way to spare a 268 MB mask the hash at 128^4);
* the notebooks' loop - look at ``safe_set`` after every update, then ``update_safe_set(
  can_shrink=False)`` - uploads no mask between iterations, and an in-place edit of ``safe_set``
  is still honoured;
* a closed-form V edited in place after ``update_values()`` keeps the OLD ordering values until
  ``update_values()`` is called again (``lyapunov.py:305-322``).
"""

import numpy as np
import pytest

import cases
import oracle

pytestmark = pytest.mark.gpu


def _pair(case, init):
    import safe_learning_amd as sl
    from safe_learning_amd.benchmarks import build_specs
    policy, dynamics, value, lv = build_specs(case)
    lyap = sl.Lyapunov(sl.GridWorld(case["limits"], case["num_points"]), value, dynamics, case["lf"],
                       lv, case["tau"], policy, initial_set=init)
    olyap = cases.oracle_lyapunov(case)
    olyap.initial_safe_set = init.copy()
    return lyap, olyap


def _case(num_points=(61, 67)):
    # (the level set grows from the 121 initial cells to 2121 cells; the can_shrink=False loop
    # below adds cells at every step)
    return cases.make_case("pendulum", num_points=list(num_points), dynamics="analytic", tau_scale=0.02)


def _initial(case):
    return np.array(cases.initial_safe_mask(case), dtype=bool, copy=True)


def test_initial_mask_edited_in_place_is_noticed():
    case = _case()
    n = int(np.prod(case["num_points"]))
    shape = tuple(case["num_points"])
    centre = np.ravel_multi_index((shape[0] // 2, shape[1] // 2), shape)
    far = np.ravel_multi_index((3, 5), shape)
    init = _initial(case)
    init[far] = True
    lyap, olyap = _pair(case, init)
    lyap.update_safe_set()
    olyap.update_safe_set()
    np.testing.assert_array_equal(lyap.safe_set, olyap.safe_set)
    assert lyap.safe_set.sum() > 2000 and lyap.safe_set[far]
    uploads = lyap.mask_uploads
    lyap.update_safe_set()                                   # nothing changed: no upload
    assert lyap.mask_uploads == uploads
    # move one initial cell: same object, same count
    other = np.ravel_multi_index((shape[0] - 4, 7), shape)
    init[far], init[other] = False, True
    olyap.initial_safe_set = init.copy()
    lyap.update_safe_set()
    olyap.update_safe_set()
    assert lyap.mask_uploads == uploads + 1
    assert lyap.safe_set[other] and not lyap.safe_set[far]
    np.testing.assert_array_equal(lyap.safe_set, olyap.safe_set)
    assert lyap.c_max == olyap.c_max
    # shrink the initial region to the origin, in place: another level set altogether
    init[:] = False
    init[centre] = True
    olyap.initial_safe_set = init.copy()
    lyap.update_safe_set()
    olyap.update_safe_set()
    np.testing.assert_array_equal(lyap.safe_set, olyap.safe_set)
    assert lyap.c_max == olyap.c_max
    # an index list edited in place
    indices = np.array([centre, far])
    lyap.initial_safe_set = indices
    olyap.initial_safe_set = indices
    lyap.update_safe_set()
    olyap.update_safe_set()
    np.testing.assert_array_equal(lyap.safe_set, olyap.safe_set)
    indices[1] = other
    lyap.update_safe_set()
    olyap.update_safe_set()
    assert lyap.safe_set[other] and not lyap.safe_set[far]
    np.testing.assert_array_equal(lyap.safe_set, olyap.safe_set)


def test_read_only_initial_mask_is_trusted_by_identity(monkeypatch):
    from safe_learning_amd import lyapunov as module
    case = _case()
    n = int(np.prod(case["num_points"]))
    init = _initial(case)
    init.flags.writeable = False
    lyap, olyap = _pair(case, init)
    lyap.update_safe_set()
    olyap.update_safe_set()
    np.testing.assert_array_equal(lyap.safe_set, olyap.safe_set)
    calls = []
    real = module._digest
    monkeypatch.setattr(module, "_digest", lambda array: calls.append(1) or real(array))
    uploads = lyap.mask_uploads
    lyap.update_safe_set()
    lyap.update_safe_set()
    assert not calls and lyap.mask_uploads == uploads          # neither hashed nor uploaded
    base = np.zeros(n, dtype=bool)                             # a read-only VIEW of a writable array is not frozen
    base[n // 2] = True
    view = base[:]
    view.flags.writeable = False
    assert not module._frozen(view) and module._frozen(init)


def test_notebook_loop_uploads_no_mask_between_iterations():
    """``adaptive_safety_verification.ipynb`` / ``1d_example.ipynb``: update, look at the set,
    update again without shrinking."""
    case = _case()
    lyap, olyap = _pair(case, _initial(case))
    lyap.update_safe_set()
    olyap.update_safe_set()
    uploads = lyap.mask_uploads
    first = int(olyap.safe_set.sum())
    for tau_scale in (0.9, 0.8, 0.7):
        assert int(lyap.safe_set.sum()) == int(olyap.safe_set.sum())       # the caller READS the set
        lyap.tau = olyap.tau = case["tau"] * tau_scale
        lyap.update_safe_set(can_shrink=False)
        olyap.update_safe_set(can_shrink=False)
        np.testing.assert_array_equal(lyap.safe_set, olyap.safe_set)
    assert lyap.mask_uploads == uploads, "a read of safe_set must not cost a mask upload"
    assert int(olyap.safe_set.sum()) > first                              # (the loop did something)
    # ... but an edit of the array the caller holds is honoured (lyapunov.py:507-510 reads it)
    mask = lyap.safe_set
    extra = np.flatnonzero(~mask)[:3]
    mask[extra] = True
    olyap.safe_set[extra] = True
    lyap.update_safe_set(can_shrink=False)
    olyap.update_safe_set(can_shrink=False)
    assert lyap.mask_uploads == uploads + 1
    np.testing.assert_array_equal(lyap.safe_set, olyap.safe_set)
    # and so is the setter
    lyap.safe_set = olyap.safe_set.copy()
    lyap.update_safe_set(can_shrink=False)
    olyap.update_safe_set(can_shrink=False)
    np.testing.assert_array_equal(lyap.safe_set, olyap.safe_set)


def test_closed_form_values_keep_the_numbers_of_update_values():
    """A ``QuadraticFunction`` whose matrix is replaced IN PLACE after ``update_values()``: the
    reference's ``values`` (the ordering of ``lyapunov.py:512`` and ``c_max``) stay those of the last
    ``update_values()``, while the decrease check uses the new function."""
    case = _case((61, 64))            # (implicit values need a last axis of whole groups of 8 cells)
    lyap, olyap = _pair(case, _initial(case))
    assert lyap._values_implicit
    old_values = olyap.values.copy()
    new_matrix = case["P"] * np.array([[1.0, 0.5], [0.5, 3.0]])
    lyap.lyapunov_function.matrix[...] = new_matrix
    olyap.lyapunov_function.matrix[...] = new_matrix
    lyap.update_safe_set()
    olyap.update_safe_set()
    np.testing.assert_array_equal(olyap.values, old_values)             # the oracle did not refresh them
    np.testing.assert_array_equal(lyap.values, old_values)
    np.testing.assert_array_equal(lyap.safe_set, olyap.safe_set)
    assert lyap.c_max == olyap.c_max
    lyap.update_values()
    olyap.update_values()
    lyap.update_safe_set()
    olyap.update_safe_set()
    np.testing.assert_array_equal(lyap.values, olyap.values)
    np.testing.assert_array_equal(lyap.safe_set, olyap.safe_set)


def test_large_writable_initial_mask_is_hashed_with_a_warning():
    """More than 2^24 bytes of writable initial mask: still hashed on every update (never silently
    stale), and the user is told once how to avoid the cost."""
    import safe_learning_amd as sl
    from safe_learning_amd.benchmarks import build_specs
    case = cases.make_case("pendulum", num_points=[4104, 4096], dynamics="linear", tau_scale=0.02)
    init = _initial(case)
    assert init.nbytes > (1 << 24) and init.flags.writeable
    policy, dynamics, value, lv = build_specs(case)
    lyap = sl.Lyapunov(sl.GridWorld(case["limits"], case["num_points"]), value, dynamics, case["lf"],
                       lv, case["tau"], policy, initial_set=init)
    with pytest.warns(RuntimeWarning, match="read-only"):
        lyap.update_safe_set()
    first = lyap.safe_count
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                       # only once per object
        far = np.ravel_multi_index((5, 7), tuple(case["num_points"]))
        init[far] = True                                     # in place
        lyap.update_safe_set()
    assert lyap.safe_set[far] and lyap.safe_count == first + 1
