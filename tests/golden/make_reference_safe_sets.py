"""Safe sets computed BY THE REFERENCE'S OWN CODE (build container only).

What runs here, unmodified, loaded from ``/root/reference``:

* ``lyapunov.py``: ``Lyapunov.__init__`` / ``update_values`` / ``threshold`` /
  ``v_decrease_confidence`` / ``v_decrease_bound`` / ``update_safe_set`` (``:176-606``: the value
  sort, the batch loop, the prefix rule with its early exit, the ``c_max`` index arithmetic, the
  adaptive branch), ``smallest_boundary_value`` (``:22-56``), ``perturb_actions`` /
  ``get_safe_sample`` (``:609-797``);
* ``functions.py``: ``GridWorld``, ``LinearSystem``, ``QuadraticFunction``, ``Saturation``, the
  ``Triangulation`` graph wrapper with ``_Triangulation`` underneath (value tables, table
  policies, ``gradient`` for L_v) - the policy, dynamics, V and L_v handed to ``Lyapunov`` are
  the reference's own objects;
* ``examples/utilities.py``: ``InvertedPendulum`` and ``CartPole`` (the Euler models).

* ``functions.py``'s GP classes (round 4): a GP dynamics model is the reference's
  ``GaussianProcess(GPRCached(...))`` / ``FunctionStack`` (``:254-307, 357-546``: Cholesky cache,
  ``build_predict``, ``beta sqrt(var)``, ``add_data_point`` with its cache rebuild) on top of
  ``tests/golden/numpy_gpflow.py``, the restatement of the gpflow 0.4.0 pieces underneath (RBF
  kernel, mean functions, parameter plumbing).  Until round 3 this leaf was the oracle's callable.

What does NOT run is TensorFlow (absent from the image): ``tests/golden/numpy_tf.py`` answers the
ops these files request with NumPy (elementwise IEEE-754 double arithmetic; ``matmul`` /
``reduce_sum`` accumulate left to right - see that file for what this does and does not claim;
``tf.cholesky`` / ``tf.matrix_triangular_solve`` are LAPACK).

``tests/test_oracle_reference_safe_sets.py`` replays every scenario on ``oracle.Lyapunov`` with the
oracle's own function classes and requires the safe set, ``c_max``, the value table, the
refinement array and every sample to be equal bit for bit, call after call;
``tests/test_gpu_reference_safe_sets.py`` does the same with the HIP engine.

Tie order: ``lyapunov.py:512`` sorts with NumPy's default argsort, whose order of equal values
depends on the NumPy build and the CPU (AVX-512 / AVX2 / scalar sort).  Scenarios marked
``unique`` use grids that are not symmetric about the origin and the script asserts that no two
cells have the same value; the symmetric scenarios have deterministic odd-symmetric closed
loops, where both cells of a tied pair get the same decision and the outcome does not depend on
their order.  What a tie does in general stays "parity unpinned" (DESIGN.md section 6).

``get_lyapunov_region`` (``lyapunov.py:59-139``) is Python-2 code (``tiebreaker.next()``) and
cannot be run.

    python tests/golden/make_reference_safe_sets.py          (needs /root/reference)
"""

import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy_tf                                         # noqa: E402
import make_reference_gp                                # noqa: E402

OUT = os.path.join(HERE, "reference_safe_sets.npz")


# --------------------------------------------------------------------------------------
# The reference's own function classes as leaves
# --------------------------------------------------------------------------------------

def reference_specs(case, ref):
    """(policy, dynamics, V, L_v) of a case (tests/cases.py parameters) built from the REFERENCE'S
    classes: ``functions.py`` ``LinearSystem`` / ``QuadraticFunction`` / ``Saturation`` /
    ``Triangulation`` / ``GaussianProcess(GPRCached)`` / ``FunctionStack`` and
    ``examples/utilities.py`` ``InvertedPendulum`` / ``CartPole``."""
    F, tf = ref.functions, sys.modules["tensorflow"]
    d = case["d"]
    if "policy_table" in case:
        tab = case["policy_table"]
        policy = F.Triangulation(F.GridWorld(case["limits"], tab["num_points"]), tab["values"])
    else:
        policy = F.LinearSystem((case["K"],))
    if case["saturate"] is not None:
        policy = F.Saturation(policy, *case["saturate"])
    dyn = case["dynamics"]
    if dyn["kind"] == "linear":
        dynamics = F.LinearSystem((dyn["matrix"],))
    elif dyn["kind"] == "pendulum":
        dynamics = ref.examples.InvertedPendulum(dyn["mass"], dyn["length"], dyn["friction"],
                                                 dyn["dt"], dyn["normalization"])
    elif dyn["kind"] == "cartpole":
        dynamics = ref.examples.CartPole(dyn["pendulum_mass"], dyn["cart_mass"], dyn["length"],
                                         dyn["rot_friction"], dyn["dt"], dyn["normalization"])
    else:
        dynamics = make_reference_gp.reference_gp_dynamics(case, ref)
    vspec = case.get("V", {"kind": "quadratic"})
    if vspec["kind"] == "quadratic":
        value = F.QuadraticFunction(case["P"])
    elif vspec["kind"] == "network":
        # examples/utilities.py:48-104; tf.get_variable hands out the case's weights by name
        names, in_dim = [], d
        for i, out_dim in enumerate(vspec["layer_dims"]):
            names.append("weights_posdef_%d" % i)
            if out_dim > in_dim:
                names.append("weights_%d" % i)
            in_dim = out_dim
        numpy_tf.set_named_variables(dict(zip(names, vspec["weights"])))
        activations = {"tanh": tf.tanh, "relu": lambda x, name=None: tf.maximum(x, 0.0)}
        value = ref.examples.LyapunovNetwork(d, vspec["layer_dims"],
                                             [activations[a] for a in vspec["activations"]],
                                             eps=vspec["eps"])
    else:
        value = F.Triangulation(F.GridWorld(case["limits"],
                                            vspec.get("num_points", case["num_points"])),
                                vspec["values"], project=vspec.get("project", False))
    kind, arg = (case["lv"] + (None,))[:2]
    if kind == "const":
        lv = arg
    elif kind == "abs_linear":                   # the notebooks' lambda x: tf.abs(V.gradient(x))
        lv = lambda x: tf.abs(tf.matmul(x, arg.T))                    # noqa: E731
    elif kind == "abs_grad":                     # inverted_pendulum.ipynb cell 14
        def lv(x):
            gradient = tf.abs(value.gradient(x))
            gradient.shape = (None, d)           # static shape of the py_func output
            return gradient
    else:
        raise ValueError(kind)
    return policy, dynamics, value, lv


# --------------------------------------------------------------------------------------
# Scenarios (shared with tests/test_oracle_golden.py, which replays them on the oracle)
# --------------------------------------------------------------------------------------

def scenarios():
    """-> list of dicts: name, case (parameters, tests/cases.py format), batch (cells per
    verification batch, configuration.py:19), adaptive, unique (no two equal values), steps."""
    from safe_learning_amd.benchmarks import GP_VARIANTS, make_case, table_case

    def skew(case, limits):
        case["limits"] = [list(map(float, row)) for row in limits]
        return case

    rng = np.random.default_rng(7)
    out = []

    # test_lyapunov.py:48-74's 1-D system on 1001 cells: one batch, then 64-cell batches
    for batch in (10000, 64):
        case = make_case("1d", tau_scale=0.01)
        case["initial_set"] = np.arange(440, 561)
        out.append(dict(name="1d_batch%d" % batch, case=case, batch=batch,
                        steps=[("update", {}), ("update", {"can_shrink": False})]))

    # deterministic closed loop, symmetric grid (ties between x and -x, same decision for both)
    out.append(dict(name="pendulum_linear_symmetric",
                    case=make_case("pendulum", num_points=41, dynamics="linear", tau_scale=0.02),
                    batch=97, steps=[("update", {}), ("update", {"can_shrink": False})]))

    # nothing passes and there is no initial set: the first cell fails (c_max index -1)
    case = make_case("pendulum", num_points=21, dynamics="linear", tau_scale=50.0)
    case["initial_radius"] = -1.0
    out.append(dict(name="nothing_safe", case=skew(case, [[-1, 1.03], [-0.97, 1]]), batch=50,
                    unique=True, no_initial_set=True, steps=[("update", {})]))

    # everything passes in every batch: no early exit (c_max index = last batch start - 1)
    case = make_case("pendulum", num_points=21, dynamics="linear", tau_scale=1e-9)
    for batch in (50, 10000):
        out.append(dict(name="everything_safe_batch%d" % batch,
                        case=skew(dict(case), [[-0.3, 0.31], [-0.29, 0.3]]),
                        batch=batch, unique=True, steps=[("update", {})]))

    # can_shrink=False with cells marked safe by hand beyond the level set: those in the batch
    # that exits are cleared, those in later batches stay (batch granularity of :585)
    case = skew(make_case("pendulum", num_points=31, dynamics="analytic", tau_scale=0.01),
                [[-1, 1.03], [-0.97, 1]])
    marks = rng.choice(31 * 31, 60, replace=False)
    out.append(dict(name="pendulum_analytic_marks", case=case, batch=100, unique=True,
                    steps=[("update", {}), ("mark_safe", marks),
                           ("update", {"can_shrink": False}), ("update", {"can_shrink": True})]))

    # GP dynamics: grow, add data, re-verify without and with shrinking; then the sampling rule
    hyper = GP_VARIANTS["tight"]
    case = skew(make_case("pendulum", num_points=33, n_gp=40, tau_scale=0.01, **hyper),
                [[-1, 1.03], [-0.97, 1]])
    new_x = rng.uniform(-0.5, 0.5, (5, 3))
    new_y = new_x @ case["dynamics"]["prior"].T + rng.normal(0, 2e-4, (5, 2))
    perturbations = np.array([[-0.2], [-0.05], [0.0], [0.05], [0.2]])
    limits = np.array([[-1.0, 1.0]])
    out.append(dict(name="pendulum_gp", case=case, batch=100, unique=True,
                    steps=[("update", {}),
                           ("sample", dict(perturbations=perturbations, limits=limits, positive=True)),
                           ("sample", dict(perturbations=perturbations, limits=limits, positive=False)),
                           ("add_data", (new_x, new_y)),
                           ("update", {"can_shrink": False}),
                           ("sample", dict(perturbations=perturbations, limits=None, positive=True)),
                           ("update", {"can_shrink": True})]))

    # the notebooks' FunctionStack of one GP per output, 4-D
    case = skew(make_case("cartpole", num_points=7, n_gp=200, tau_scale=0.0001, stack=True, **hyper),
                [[-1, 1.03], [-0.97, 1], [-1, 1.01], [-0.99, 1]])
    out.append(dict(name="cartpole_gp_stack", case=case, batch=500, unique=True,
                    steps=[("update", {})]))

    # table V (projected Triangulation), L_v = |gradient| per dimension (the 1-norm of
    # lyapunov.py:285-286), table policy: inverted_pendulum.ipynb's shape
    # (table intervals coprime with the grid's: interior cells do not sit on table grid lines,
    # where the reference's `%` wrap-around returns something else than the interpolant)
    for n_gp, tau_scale in ((30, 0.001), (120, 0.01)):
        case = table_case(num_points=(41, 31), table_points=(50, 38), n_gp=n_gp,
                          tau_scale=tau_scale, limits=[[-1, 1.03], [-0.97, 1]])
        out.append(dict(name="table_gp%d" % n_gp, case=case, batch=128, unique=True,
                        steps=[("update", {}), ("update", {"can_shrink": False})]))

    # V a LyapunovNetwork (config C3's family): the reference's class multiplies left to right here,
    # the oracle with BLAS - values agree to 1e-13, the safe set must be the same
    from tests import cases as _cases
    case = skew(make_case("pendulum", num_points=27, dynamics="analytic", tau_scale=0.0),
                [[-1, 1.03], [-0.97, 1]])
    dims = [8, 8, 16]
    case["V"] = {"kind": "network", "layer_dims": dims, "activations": ["tanh"] * 3, "eps": 1e-8,
                 "weights": _cases.lyapunov_like_network_weights(case["P"], dims)}
    out.append(dict(name="pendulum_network", case=case, batch=100, unique=True, values_rtol=1e-12,
                    steps=[("update", {}), ("update", {"can_shrink": False})]))

    # adaptive discretisation (lyapunov.py:443-488, 540-582)
    for tau_scale in (0.1, 0.03, 0.003):
        case = skew(make_case("pendulum", num_points=25, dynamics="analytic", tau_scale=tau_scale),
                    [[-1, 1.03], [-0.97, 1]])
        for refinement, factor in ((3, 1.0), (6, 1.4)):
            out.append(dict(name="adaptive_tau%g_%d" % (tau_scale, refinement), case=dict(case),
                            batch=60, unique=True, adaptive=True,
                            steps=[("update", dict(max_refinement=refinement, safety_factor=factor)),
                                   ("update", dict(can_shrink=False, max_refinement=refinement,
                                                   safety_factor=factor))]))
    case = skew(make_case("pendulum", num_points=25, n_gp=40, tau_scale=0.05, **hyper),
                [[-1, 1.03], [-0.97, 1]])
    out.append(dict(name="adaptive_gp", case=case, batch=80, unique=True, adaptive=True,
                    steps=[("update", dict(max_refinement=4, safety_factor=1.2))]))
    # the notebooks' dynamics model (inverted_pendulum.ipynb:145-181): a FunctionStack of
    # single-output GPs with the kernels Linear + Matern32 * Linear, through the notebook's loop -
    # level set, most uncertain safe pair, new observations, level set again
    from safe_learning_amd.benchmarks import notebook_kernels
    for n_gp, name in ((40, "notebook_kernels"), (135, "notebook_kernels_130")):
        case = skew(make_case("pendulum", num_points=25, n_gp=n_gp, tau_scale=0.01, stack=True,
                              noise_std=0.001), [[-1, 1.03], [-0.97, 1]])
        kernels = notebook_kernels(case)
        for spec in kernels:                     # (a prior wide enough for the bound to matter)
            spec[0][0][1]["variance"] = [3e-3, 3e-3, 3e-3]
            spec[1][1][1]["variance"] = 3e-3
        case["dynamics"]["kernels"] = kernels
        new_x = np.random.default_rng(n_gp).uniform(-0.5, 0.5, (3, 3))
        new_y = new_x @ case["dynamics"]["prior"].T + 1e-3
        perturb = dict(perturbations=np.linspace(-0.3, 0.3, 5)[:, None], limits=np.array([[-1.0, 1.0]]),
                       positive=True)
        out.append(dict(name=name, case=case, batch=100, unique=True,
                        steps=[("update", {}), ("sample", perturb), ("add_data", (new_x, new_y)),
                               ("update", {"can_shrink": False}), ("sample", perturb)]))
    # sixteen of the random variations of the live comparison (seed 7), so that the engine meets
    # them too (tests/test_gpu_reference_safe_sets.py)
    for scenario in random_scenarios(16, 7):
        scenario["unique"] = True
        out.append(scenario)
    return out


def replay(scenario, lyap, dynamics, get_safe_sample, c_max_of):
    """Run a scenario's steps on a Lyapunov object (the reference's or the oracle's) -> records."""
    records = []
    for kind, arg in scenario["steps"]:
        if kind == "update":
            lyap.update_safe_set(**arg)
            records.append(dict(safe_set=lyap.safe_set.copy(), c_max=np.float64(c_max_of(lyap)),
                                refinement=np.asarray(lyap._refinement).copy()))
        elif kind == "mark_safe":
            lyap.safe_set[arg] = True
        elif kind == "add_data":
            for x, y in zip(*arg):
                dynamics.add_data_point(x[None, :], y[None, :])
        elif kind == "sample":
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)
                state_action, bound = get_safe_sample(lyap, **arg)
            records.append(dict(state_action=np.asarray(state_action), bound=np.float64(bound)))
        else:
            raise ValueError(kind)
    return records


def jsonable(obj, arrays, prefix):
    """Nested parameters -> JSON structure with the arrays moved into ``arrays`` (npz keys)."""
    if isinstance(obj, dict):
        return {k: jsonable(v, arrays, prefix + "/" + k) for k, v in obj.items()}
    if isinstance(obj, tuple):
        return {"__tuple__": [jsonable(v, arrays, "%s/%d" % (prefix, i)) for i, v in enumerate(obj)]}
    if isinstance(obj, list):
        return [jsonable(v, arrays, "%s/%d" % (prefix, i)) for i, v in enumerate(obj)]
    if isinstance(obj, np.ndarray):
        arrays[prefix] = obj
        return {"__array__": prefix}
    if isinstance(obj, (np.floating, np.integer, np.bool_)):
        return obj.item()
    return obj


def from_jsonable(obj, arrays):
    if isinstance(obj, dict):
        if "__array__" in obj:
            return arrays[obj["__array__"]]
        if "__tuple__" in obj:
            return tuple(from_jsonable(v, arrays) for v in obj["__tuple__"])
        return {k: from_jsonable(v, arrays) for k, v in obj.items()}
    if isinstance(obj, list):
        return [from_jsonable(v, arrays) for v in obj]
    return obj


def main():
    from tests import cases
    ref = numpy_tf.load_reference(examples=True)
    functions, lyapunov, config = ref.functions, ref.lyapunov, ref.config
    arrays, index = {}, []
    for scenario in scenarios():
        name, case = scenario["name"], scenario["case"]
        policy, dynamics, value, lv = reference_specs(case, ref)
        initial = None if scenario.get("no_initial_set") else cases.initial_safe_mask(case)
        config.gp_batch_size = scenario["batch"]
        grid = functions.GridWorld(case["limits"], case["num_points"])
        lyap = lyapunov.Lyapunov(grid, value, dynamics, case["lf"], lv, case["tau"], policy,
                                 initial_set=initial, adaptive=bool(scenario.get("adaptive")))
        if scenario.get("unique"):
            assert len(np.unique(lyap.values)) == len(lyap.values), name
        boundary = lyapunov.smallest_boundary_value(value, grid)
        records = replay(scenario, lyap, dynamics, lyapunov.get_safe_sample,
                         lambda obj: obj.feed_dict[obj.c_max])
        arrays[name + "/values"] = lyap.values
        arrays[name + "/boundary"] = np.float64(boundary)
        for k, record in enumerate(records):
            for key, val in record.items():
                arrays["%s/step%d/%s" % (name, k, key)] = val
        meta = {k: v for k, v in scenario.items() if k != "case"}
        meta["steps"] = [list(step) for step in scenario["steps"]]
        index.append(dict(meta=jsonable(meta, arrays, name + "/meta"),
                          case=jsonable(case, arrays, name + "/case"), records=len(records)))
        updates = [r for r in records if "safe_set" in r]
        print("%-28s cells %6d  safe %s  c_max %s" % (
            name, grid.nindex, [int(r["safe_set"].sum()) for r in updates],
            ["%.4g" % r["c_max"] for r in updates]))
    arrays["_index"] = np.array(json.dumps(index))
    arrays["_numpy_version"] = np.array(np.__version__)
    np.savez_compressed(OUT, **arrays)
    print("wrote %s (%d arrays, %.1f KiB)" % (OUT, len(arrays), os.path.getsize(OUT) / 1024.0))


def random_scenarios(count, seed):
    """Seeded random variations for the LIVE comparison (tests/test_oracle_live_reference.py):
    grid sizes and limits, tau, batch size, dynamics kind, adaptive or not, call sequences."""
    from safe_learning_amd.benchmarks import GP_VARIANTS, make_case
    rng = np.random.default_rng(seed)
    out = []
    for k in range(count):
        kind = ["linear", "analytic", "gp"][int(rng.integers(0, 3))]
        family = "pendulum" if rng.random() < 0.75 else "cartpole"
        if family == "pendulum":
            num_points = [int(v) for v in rng.integers(9, 33, 2)]
        else:
            num_points = [int(v) for v in rng.integers(4, 8, 4)]
        kw = dict(num_points=num_points, tau_scale=float(10.0 ** rng.uniform(-4.5, -1.5)))
        if kind == "gp":
            kw.update(n_gp=int(rng.integers(20, 60)), **GP_VARIANTS["tight"])
            if rng.random() < 0.4:
                kw["stack"] = True
        else:
            kw["dynamics"] = kind
        case = make_case(family, **kw)
        d = case["d"]
        lows = -1.0 + rng.uniform(-0.05, 0.05, d)
        highs = 1.0 + rng.uniform(0.01, 0.09, d)
        case["limits"] = [[float(a), float(b)] for a, b in zip(lows, highs)]
        case["initial_radius"] = float(rng.uniform(0.1, 0.35))
        adaptive = kind != "gp" and rng.random() < 0.35
        update = {}
        if adaptive:
            update = dict(max_refinement=int(rng.integers(2, 7)),
                          safety_factor=float(rng.uniform(1.0, 1.6)))
        n = int(np.prod(num_points))
        steps = [("update", dict(update))]
        if rng.random() < 0.5:
            steps.append(("mark_safe", rng.choice(n, max(n // 20, 1), replace=False)))
        steps.append(("update", dict(update, can_shrink=False)))
        steps.append(("update", dict(update)))
        out.append(dict(name="random_%d_%s_%s" % (k, family, kind), case=case, adaptive=adaptive,
                        batch=int(rng.integers(7, 400)), steps=steps))
    return out


def check_live(count, seed):
    """Reference and oracle side by side on random scenarios, in this process (no fixture)."""
    import oracle
    from oracle import np_lyapunov
    from tests import cases
    ref = numpy_tf.load_reference(examples=True)
    compared = 0
    for scenario in random_scenarios(count, seed):
        name, case = scenario["name"], scenario["case"]
        initial = cases.initial_safe_mask(case)
        # the reference's run
        policy, dynamics, value, lv = reference_specs(case, ref)
        ref.config.gp_batch_size = scenario["batch"]
        grid = ref.functions.GridWorld(case["limits"], case["num_points"])
        lyap = ref.lyapunov.Lyapunov(grid, value, dynamics, case["lf"], lv, case["tau"], policy,
                                     initial_set=initial, adaptive=scenario["adaptive"])
        if len(np.unique(lyap.values)) != len(lyap.values):
            print("%s: equal values on the grid, skipped (tie order is unpinned)" % name)
            continue
        want = replay(scenario, lyap, dynamics, ref.lyapunov.get_safe_sample,
                      lambda obj: obj.feed_dict[obj.c_max])
        # the oracle's
        np_lyapunov.config.gp_batch_size = scenario["batch"]
        opolicy, odynamics, ovalue, olv = cases.oracle_specs(case)
        olyap = oracle.Lyapunov(oracle.GridWorld(case["limits"], case["num_points"]), ovalue,
                                odynamics, case["lf"], olv, case["tau"], opolicy, initial_set=initial)
        olyap.adaptive = scenario["adaptive"]
        got = replay(scenario, olyap, odynamics, oracle.get_safe_sample, lambda obj: obj.c_max)
        assert np.array_equal(olyap.values, lyap.values), name
        for k, (a, b) in enumerate(zip(got, want)):
            for key in b:
                if key == "bound" and case["dynamics"]["kind"] == "gp":     # a posterior std
                    assert np.allclose(a[key], b[key], rtol=1e-10, atol=0), (name, k, key)
                    continue
                assert np.array_equal(a[key], b[key]), "%s step %d %s" % (name, k, key)
        compared += 1
        print("%-34s cells %5d batch %3d safe %s%s" % (
            name, grid.nindex, scenario["batch"], [int(r["safe_set"].sum()) for r in want],
            "  adaptive" if scenario["adaptive"] else ""))
    print("LIVE OK: %d scenarios compared" % compared)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--check-live":
        check_live(int(sys.argv[2]), int(sys.argv[3]))
    else:
        main()
