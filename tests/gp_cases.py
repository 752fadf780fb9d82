"""GP-dynamics workloads of the parity tests (TEST INFRASTRUCTURE).

Every entry is non-degenerate by construction: on the oracle both classes of the decrease mask
occur and the safe set grows beyond the initial set (tests assert it).  SURVEY 8d's literal
hyper-parameters (signal 0.05, noise 0.01, lengthscale 0.5 with 1024 uniformly random training
points in 5-D) leave the posterior at the prior and no cart-pole cell passes the check, so they
are kept only as the ``survey`` variant of ``safe_learning_amd.benchmarks.GP_VARIANTS``.

Columns: family, make_case kwargs, SL_GP_CFG override (None = engine's choice), minimum growth
of the safe set beyond the initial set.
"""

from safe_learning_amd.benchmarks import GP_VARIANTS

INFORMED, TIGHT = GP_VARIANTS["informed"], GP_VARIANTS["tight"]

GP_CASES = [
    # ---- pendulum (d = 2, p = 3) ---------------------------------------------------------
    ("pendulum", dict(num_points=40, n_gp=3, tau_scale=0.0, **TIGHT), None, 5),        # tiny n
    ("pendulum", dict(num_points=64, n_gp=128, tau_scale=0.01, **INFORMED), None, 100),  # cfg 0, 2 panels
    ("pendulum", dict(num_points=[33, 50], n_gp=200, tau_scale=0.01, **INFORMED), None, 100),
    ("pendulum", dict(num_points=48, n_gp=600, tau_scale=0.01, **INFORMED), "1", 100),  # cfg 1, 1 panel
    ("pendulum", dict(num_points=48, n_gp=600, tau_scale=0.01, **INFORMED), "2", 100),  # cfg 2, 2 panels
    ("pendulum", dict(num_points=40, n_gp=2600, tau_scale=0.01, **INFORMED), None, 100),   # inputs read from L2
    # ---- cart-pole (d = 4, p = 5: the headline instantiation) -------------------------
    ("cartpole", dict(num_points=12, n_gp=150, tau_scale=0.0, **TIGHT), None, 100),     # cfg 0
    ("cartpole", dict(num_points=12, n_gp=100, tau_scale=0.0, stack=True, **TIGHT), None, 100),
    ("cartpole", dict(num_points=12, n_gp=1100, tau_scale=0.0, **INFORMED), "1", 100),  # cfg 1, 2 panels
    ("cartpole", dict(num_points=14, n_gp=520, tau_scale=0.0, **TIGHT), "2", 100),      # cfg 2, 2 panels
    ("cartpole", dict(num_points=14, n_gp=520, tau_scale=0.0, **INFORMED), "2", 100),
    ("cartpole", dict(num_points=12, n_gp=300, tau_scale=0.0, stack=True, **TIGHT), None, 100),  # stack on cfg 2
    # largest training set whose inputs fit LDS at p = 5 (alpha' stays in global memory) ...
    ("cartpole", dict(num_points=11, n_gp=1500, tau_scale=0.0, **INFORMED), None, 100),
    # ... and beyond it: the generation phase reads the training inputs from L2
    ("cartpole", dict(num_points=11, n_gp=2000, tau_scale=0.0, **INFORMED), None, 100),
]


# ---- cases of the reference-run GP posterior fixture (tests/golden/make_reference_gp.py) --------
# name, family, make_case kwargs; optional: scale (GPRCached's internal scaling), noise_ratio
# (noise variance / signal variance), zero_mean, add_points (observations added one by one with
# add_data_point), cells (grid cells queried).

def reference_gp_case_list():
    from safe_learning_amd.benchmarks import GP_VARIANTS, make_case
    informed, tight = GP_VARIANTS["informed"], GP_VARIANTS["tight"]
    out = []

    def add(name, family, num_points, n_gp, hyper, cells=384, **extra):
        kw = dict(num_points=num_points, n_gp=n_gp, tau_scale=0.01, **hyper)
        kw.update({k: extra.pop(k) for k in ("stack", "seed") if k in extra})
        out.append(dict(name=name, family=family, kwargs=kw, cells=cells, **extra))

    add("pendulum_n3", "pendulum", 40, 3, tight)
    add("pendulum_n130", "pendulum", 64, 130, informed)
    add("pendulum_n130_scale", "pendulum", 64, 130, informed, scale=7.5)
    add("pendulum_n512_C2", "pendulum", 256, 512, informed)
    add("pendulum_n512_ill", "pendulum", 256, 512, informed, noise_ratio=1e-6)
    add("pendulum_n1024_zero_mean", "pendulum", 48, 1024, informed, zero_mean=True)
    add("cartpole_n3", "cartpole", 12, 3, tight)
    add("cartpole_n130", "cartpole", 12, 130, tight)
    add("cartpole_n130_stack", "cartpole", 12, 130, tight, stack=True)
    add("cartpole_n512_scale", "cartpole", 14, 512, informed, scale=0.04)
    add("cartpole_n1024_C4", "cartpole", 128, 1024, informed, cells=768)
    add("cartpole_n1024_ill", "cartpole", 16, 1024, informed, noise_ratio=1e-6, seed=3)
    add("cartpole_n1024_stack", "cartpole", 16, 1024, informed, stack=True)
    add("pendulum_n130_added", "pendulum", 64, 130, informed, add_points=6)
    return out


def reference_gp_build_case(spec):
    """make_case parameters of a case of ``tests/golden/reference_gp_posterior.npz``."""
    from safe_learning_amd.benchmarks import make_case
    case = make_case(spec["family"], **spec["kwargs"])
    if "noise_ratio" in spec:
        case["dynamics"]["noise_variance"] = float(case["dynamics"]["variance"] * spec["noise_ratio"])
    return case


def reference_gp_tolerance(cond):
    """Relative tolerance of a posterior mean / confidence bound against the reference-run fixture.
    The fixture solves with the Cholesky factor of ``K + noise I`` (``functions.py:408-409, 441``);
    the oracle uses the same LAPACK calls but BLAS dot products in the kernel matrix, the engine
    multiplies by an explicit inverse factor and sums on the matrix cores: rounding differences of
    the kernel entries (1e-16) come back multiplied by up to cond(K) in ``a = L^-1 k_x`` and in the
    difference ``sigma^2 - |a|^2``.  Measured: <= 0.4 eps cond(K) for the oracle; the bound below
    leaves a factor 20 and never exceeds the north star's 1e-5 (cond(K) <= 5e8 in the fixture)."""
    return 1e-13 + 8 * 2.2e-16 * cond


def reference_gp_model(ns, spec, case, fixture):
    """The dynamics model of a fixture case from the function classes of ``ns`` (the ``oracle``
    package or ``safe_learning_amd``): same training data, hyper-parameters, mean function, scale,
    and the observations the fixture added with ``add_data_point``."""
    import numpy as np
    name, d, dyn = spec["name"], case["d"], case["dynamics"]
    assert np.array_equal(dyn["X"], fixture[name + "/X"]) and np.array_equal(dyn["Y"], fixture[name + "/Y"])
    assert dyn["noise_variance"] == float(fixture[name + "/noise_variance"])

    def head(Y, prior, lengthscales):
        kern = ns.RBF(d + 1, dyn["variance"], lengthscales, ARD=True)
        mean = None if spec.get("zero_mean") else ns.LinearSystem((prior,))
        gp = ns.GPRCached(dyn["X"], Y, kern, mean, scale=spec.get("scale", 1.0),
                          likelihood_variance=dyn["noise_variance"])
        return ns.GaussianProcess(gp, dyn["beta"])

    if case["stack"]:
        model = ns.FunctionStack([head(dyn["Y"][:, [k]], dyn["prior"][[k], :], dyn["lengthscales"][k])
                                  for k in range(d)])
    else:
        model = head(dyn["Y"], dyn["prior"], dyn["lengthscales"])
    if "add_points" in spec:
        for x, y in zip(fixture[name + "/added_x"], fixture[name + "/added_y"]):
            model.add_data_point(x[None, :], y[None, :])
    return model


# ---- kernels other than the RBF: Linear + Matern32 * Linear of the reference's notebooks -----------
# A kernel is a list of products, a product a list of leaves (kind, keyword arguments of the
# gpflow 0.4.0 constructor).  The pendulum cases follow examples/inverted_pendulum.ipynb:145-158
# (variances = squared difference of the true and the prior linearisation, clipped at 1e-5).

def _notebook_kernel(variances):
    return [[("linear", dict(input_dim=3, variance=[float(v) for v in variances], ARD=True))],
            [("matern32", dict(input_dim=1, lengthscales=1.0, active_dims=[0])),
             ("linear", dict(input_dim=1, variance=float(variances[1])))]]


def kernel_case_list():
    """-> the cases of ``tests/golden/reference_gp_kernels.npz``: name, grid / policy family, number
    of training points, one kernel per output column (``FunctionStack`` of single-output GPs)."""
    nb = [[2.3e-3, 1.1e-5, 4.0e-4], [1.0e-5, 6.2e-4, 2.5e-3]]
    return [
        dict(name="notebook_n40", n_gp=40, kernels=[_notebook_kernel(nb[0]), _notebook_kernel(nb[1])]),
        dict(name="notebook_n130", n_gp=130, kernels=[_notebook_kernel(nb[0]), _notebook_kernel(nb[1])]),
        dict(name="notebook_n250_added", n_gp=246, add_points=4,
             kernels=[_notebook_kernel(nb[0]), _notebook_kernel(nb[1])]),
        # every leaf kind, ARD lengthscales, active dimensions out of order, a product of three
        dict(name="mixed_n90", n_gp=90, kernels=[
            [[("matern32", dict(input_dim=3, variance=0.02, lengthscales=[0.6, 0.9, 1.3], ARD=True))],
             [("rbf", dict(input_dim=2, variance=0.5, lengthscales=[0.8, 1.1], active_dims=[2, 0], ARD=True)),
              ("linear", dict(input_dim=1, variance=0.01, active_dims=[1])),
              ("matern32", dict(input_dim=1, variance=1.5, lengthscales=0.7, active_dims=[1]))]],
            [[("rbf", dict(input_dim=3, variance=0.01, lengthscales=0.9))],
             [("linear", dict(input_dim=3, variance=[1e-3, 2e-3, 5e-4], ARD=True))]]]),
    ]


def kernel_from_spec(products, lib):
    """One kernel object from a spec: ``lib`` is the ``gpflow`` stand-in module, the ``oracle``
    package or ``safe_learning_amd`` (its ``kernels`` namespace)."""
    import oracle as _oracle
    from safe_learning_amd.benchmarks import kernel_from_products
    if lib is _oracle:
        leaves = {"rbf": _oracle.np_functions.SlicedRBF, "matern32": _oracle.np_functions.Matern32,
                  "linear": _oracle.np_functions.Linear}
    else:
        k = lib.kernels
        leaves = {"rbf": k.RBF, "matern32": k.Matern32, "linear": k.Linear}
    return kernel_from_products(products, leaves)


def kernel_build_case(spec):
    """Pendulum grid, policy and prior of the parity tests with ``n_gp`` observations of the true
    dynamics (``safe_learning_amd.benchmarks.make_case``); the GP models come from ``spec``."""
    from safe_learning_amd.benchmarks import make_case
    return make_case("pendulum", num_points=[65, 48], n_gp=spec["n_gp"], tau_scale=0.01,
                     noise_std=0.001, stack=True)


def kernel_model(ns, spec, case, fixture, kern_lib=None):
    """The ``FunctionStack`` of a kernel case from the classes of ``ns`` (``oracle`` or
    ``safe_learning_amd``), with the observations the fixture added."""
    import numpy as np
    name, d, dyn = spec["name"], case["d"], case["dynamics"]
    assert np.array_equal(dyn["X"], fixture[name + "/X"]) and np.array_equal(dyn["Y"], fixture[name + "/Y"])
    heads = []
    for k in range(d):
        kern = kernel_from_spec(spec["kernels"][k], ns if kern_lib is None else kern_lib)
        gp = ns.GPRCached(dyn["X"], dyn["Y"][:, [k]], kern, ns.LinearSystem((dyn["prior"][[k], :],)),
                          likelihood_variance=dyn["noise_variance"])
        heads.append(ns.GaussianProcess(gp, dyn["beta"]))
    model = ns.FunctionStack(heads)
    if "add_points" in spec:
        for x, y in zip(fixture[name + "/added_x"], fixture[name + "/added_y"]):
            model.add_data_point(x[None, :], y[None, :])
    return model
