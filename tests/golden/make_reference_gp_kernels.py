"""GP posteriors with the kernels of the reference's notebooks, computed BY THE REFERENCE'S GP CODE
(build container only).

``examples/inverted_pendulum.ipynb:152-181`` builds its dynamics model as a ``FunctionStack`` of two
single-output GPs with the kernel ``Linear(3, ARD) + Matern32(1, active_dims=[0]) * Linear(1)`` and a
linear mean function.  Here the reference's ``GPRCached`` / ``GaussianProcess`` / ``FunctionStack``
(``functions.py:254-307, 357-546``) run unmodified on such models - and on a case with every leaf
kind, ARD lengthscales, shuffled active dimensions and a product of three - behind the stand-ins
``numpy_tf.py`` and ``numpy_gpflow.py``.  gpflow 0.4.0 is absent from the checkout: the kernel
formulas (``kernels.py``: ``Stationary.square_dist`` / ``euclid_dist``, ``RBF``, ``Matern32``,
``Linear``, ``Add``, ``Prod``) are the stand-in's restatement of the published sources and no test
of the reference pins them - these fixtures pin the ARITHMETIC AROUND THEM (Cholesky, solves,
``Kdiag - sum a^2``, ``beta sqrt(var)``, the stack, ``add_data_point``) and fix one statement of
the formulas for the oracle and the engine to agree with.

Output ``reference_gp_kernels.npz``: per case X, Y, prior, noise, the query inputs (grid cells by
flat index with their closed-loop actions, training inputs, far-away points), ``mean`` and
``bound = beta * sqrt(var)`` as ``FunctionStack.__call__`` returns them, cond(K).  Data only.

    python tests/golden/make_reference_gp_kernels.py          (needs /root/reference)
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy_tf                                         # noqa: E402
from make_reference_gp import evaluate, query_inputs    # noqa: E402
from tests.gp_cases import kernel_build_case, kernel_case_list, kernel_from_spec   # noqa: E402

OUT = os.path.join(HERE, "reference_gp_kernels.npz")


def reference_kernel_dynamics(spec, case, ref):
    gpflow = sys.modules["gpflow"]
    F = ref.functions
    dyn, d = case["dynamics"], case["d"]
    heads = []
    for k in range(d):
        kern = kernel_from_spec(spec["kernels"][k], gpflow)
        prior = dyn["prior"][[k], :]
        mean = gpflow.mean_functions.Linear(np.ascontiguousarray(prior.T), np.zeros(1))
        gp = F.GPRCached(dyn["X"], dyn["Y"][:, [k]], kern, mean)
        gp.likelihood.variance = dyn["noise_variance"]
        gp.update_cache()
        heads.append(F.GaussianProcess(gp, beta=dyn["beta"]))
    return F.FunctionStack(heads)


def main():
    ref = numpy_tf.load_reference(examples=False)
    rng = np.random.default_rng(20260929)
    arrays = {"_names": np.array([spec["name"] for spec in kernel_case_list()]),
              "_numpy_version": np.array(np.__version__)}
    for spec in kernel_case_list():
        name = spec["name"]
        case = kernel_build_case(spec)
        d, dyn = case["d"], case["dynamics"]
        dynamics = reference_kernel_dynamics(spec, case, ref)
        if "add_points" in spec:
            new_x = rng.uniform(-0.6, 0.6, (spec["add_points"], d + 1))
            new_y = new_x @ dyn["prior"].T + rng.normal(0, 2e-3, (spec["add_points"], d))
            for x, y in zip(new_x, new_y):
                dynamics.add_data_point(x[None, :], y[None, :])
            arrays[name + "/added_x"], arrays[name + "/added_y"] = new_x, new_y
        idx, cell_inputs, extra = query_inputs(case, rng, 1500)
        mean_c, bound_c = evaluate(dynamics, cell_inputs, d)
        mean_e, bound_e = evaluate(dynamics, extra, d)
        cond = max(float(np.linalg.cond(h.gaussian_process.cholesky.value)) ** 2 for h in dynamics.functions)
        arrays.update({name + "/X": dyn["X"], name + "/Y": dyn["Y"], name + "/prior": dyn["prior"],
                       name + "/noise_variance": np.float64(dyn["noise_variance"]),
                       name + "/beta": np.float64(dyn["beta"]),
                       name + "/cell_index": idx, name + "/cell_inputs": cell_inputs,
                       name + "/cell_mean": mean_c, name + "/cell_bound": bound_c,
                       name + "/extra_inputs": extra, name + "/extra_mean": mean_e,
                       name + "/extra_bound": bound_e, name + "/cond": np.float64(cond)})
        print("%-22s n %3d  cond(K) %.1e  bound %.2e .. %.2e" % (
            name, len(dynamics.functions[0].X), cond, bound_c.min(), bound_c.max()))
    np.savez_compressed(OUT, **arrays)
    print("wrote %s (%d arrays, %.1f KiB)" % (OUT, len(arrays), os.path.getsize(OUT) / 1024.0))


if __name__ == "__main__":
    main()
