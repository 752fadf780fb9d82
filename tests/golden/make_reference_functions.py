"""Outputs of the REFERENCE'S OWN function classes on seeded inputs (build container only).

The reference's ``functions.py`` (``LinearSystem``, ``QuadraticFunction`` and its gradient,
``Saturation``, negation through ``MultipliedFunction``, the TensorFlow ``Triangulation`` wrapper
and its gradient), ``examples/utilities.py`` (``InvertedPendulum``, ``CartPole`` with their
``linearize``, ``LyapunovNetwork``) and ``utilities.py`` (``dlqr``, ``batchify``, ``unique_rows``)
are loaded from ``/root/reference`` and called unmodified; ``numpy_tf.py`` answers their TensorFlow
ops (elementwise NumPy, left-to-right accumulation in ``matmul`` / ``reduce_sum``).  gpflow is
absent, so ``GPRCached`` / ``GaussianProcess`` cannot be constructed: the GP posterior stays pinned
by the reference's known-answer test only (``tests/golden/reference_known_answers.json``).

``tests/test_oracle_reference_functions.py`` requires the oracle's classes to reproduce every
array: bit for bit, except the network (BLAS products in the oracle: 1e-13 relative) and
``linearize`` / ``dlqr`` (same SciPy calls: bit for bit given the SciPy version of the fixture).

    python tests/golden/make_reference_functions.py          (needs /root/reference)
"""

import json
import os
import sys

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy_tf                                                     # noqa: E402
from make_reference_safe_sets import jsonable, from_jsonable        # noqa: E402,F401

OUT = os.path.join(HERE, "reference_functions.npz")


def inputs():
    """Seeded parameters and query points of every item (stored in the fixture)."""
    rng = np.random.default_rng(20260928)
    d = 4
    spec = {}
    spec["linear"] = dict(A=rng.normal(size=(d, d)), B=rng.normal(size=(d, 2)),
                          x=rng.uniform(-1, 1, (200, d)), u=rng.uniform(-1, 1, (200, 2)))
    P = rng.normal(size=(d, d))
    spec["quadratic"] = dict(P=P @ P.T + 0.1 * rng.normal(size=(d, d)),     # not symmetric
                             x=rng.uniform(-1, 1, (200, d)))
    spec["saturation"] = dict(K=rng.normal(size=(2, d)), lower=-0.4, upper=0.7,
                              x=rng.uniform(-1, 1, (200, d)))
    theta_max, omega_max = np.deg2rad(30), np.sqrt(9.81 / 0.5)
    u_max = 9.81 * 0.15 * 0.5 * np.sin(theta_max)
    spec["pendulum"] = dict(
        variants=[dict(mass=0.15, length=0.5, friction=0.1, dt=0.01,
                       normalization=[[theta_max, omega_max], [u_max]]),
                  dict(mass=0.1, length=0.4, friction=0.0, dt=1 / 80, normalization=None)],
        xu=rng.uniform(-1.2, 1.2, (300, 3)))
    spec["cartpole"] = dict(
        variants=[dict(pendulum_mass=0.175, cart_mass=1.732, length=0.28, rot_friction=0.01,
                       dt=0.01, normalization=[[0.5, np.deg2rad(30), 2., np.deg2rad(30)],
                                               [(0.175 + 1.732) * 4 / 0.5]]),
                  dict(pendulum_mass=0.3, cart_mass=1.0, length=0.5, rot_friction=0.0, dt=0.02,
                       normalization=None)],
        xu=rng.uniform(-1.2, 1.2, (300, 5)))
    dims = [8, 8, 16]
    from safe_learning_amd.benchmarks import network_weights
    spec["network"] = dict(input_dim=2, layer_dims=dims, activations=["tanh", "relu", "tanh"],
                           eps=1e-6, weights=network_weights(2, dims, seed=3),
                           x=rng.uniform(-1, 1, (150, 2)))
    tables = []
    for limits, num_points, project in (([[-1.0, 1.0], [-2.0, 3.0]], [7, 5], True),
                                        ([[-1.0, 1.0]] * 4, [4, 3, 5, 4], True),
                                        ([[0.0, 1.0], [-1.0, 0.5], [2.0, 3.0]], [3, 4, 3], False)):
        lim = np.asarray(limits)
        n = int(np.prod(num_points))
        inside = lim[:, 0] + rng.uniform(0.02, 0.98, (150, len(lim))) * (lim[:, 1] - lim[:, 0])
        outside = lim[:, 0] + rng.uniform(-0.3, 1.3, (60, len(lim))) * (lim[:, 1] - lim[:, 0])
        tables.append(dict(limits=limits, num_points=num_points, project=project,
                           vertex_values=rng.normal(size=(n, 2)),
                           x=np.concatenate((inside, outside if project else inside[:60]))))
    spec["tables"] = tables
    spec["dlqr"] = [dict(a=np.array([[1.2]]), b=np.array([[0.9]]), q=np.array([[1.0]]),
                         r=np.array([[0.1]])),
                    dict(a=np.eye(3) + 0.1 * rng.normal(size=(3, 3)), b=rng.normal(size=(3, 2)),
                         q=np.diag([1.0, 2.0, 0.5]), r=np.diag([0.3, 1.5]))]
    rows = rng.integers(0, 3, (40, 3)).astype(np.float64)
    spec["unique_rows"] = dict(array=rows)
    spec["batchify"] = dict(arrays=[np.arange(23), rng.normal(size=(23, 2))], batch_size=7)
    return spec


def main():
    ref = numpy_tf.load_reference(examples=True)
    F, E, tf = ref.functions, ref.examples, sys.modules["tensorflow"]
    tf.nn = type(sys)("tensorflow.nn")
    tf.nn.relu = numpy_tf.unary(lambda v: np.maximum(v, 0.0))
    utilities = sys.modules["safe_learning.utilities"]
    spec = inputs()
    out = {}

    def run(node):
        return np.asarray(node.eval({}) if isinstance(node, numpy_tf.Lazy) else node)

    s = spec["linear"]
    system = F.LinearSystem((s["A"], s["B"]))
    out["linear/two_inputs"] = run(system(s["x"], s["u"]))
    out["linear/stacked"] = run(system(np.hstack((s["x"], s["u"]))))
    s = spec["quadratic"]
    quadratic = F.QuadraticFunction(s["P"])
    out["quadratic/values"] = run(quadratic(s["x"]))
    out["quadratic/gradient"] = run(quadratic.gradient(s["x"]))
    out["quadratic/negated"] = run((-quadratic)(s["x"]))
    s = spec["saturation"]
    out["saturation/values"] = run(F.Saturation(F.LinearSystem((s["K"],)), s["lower"],
                                                s["upper"])(s["x"]))
    for name, cls in (("pendulum", E.InvertedPendulum), ("cartpole", E.CartPole)):
        s = spec[name]
        for k, kwargs in enumerate(s["variants"]):
            model = cls(**kwargs)
            width = s["xu"].shape[1] - 1
            out["%s/%d/values" % (name, k)] = run(model(s["xu"][:, :width], s["xu"][:, width:]))
            a, b = model.linearize()
            out["%s/%d/A" % (name, k)], out["%s/%d/B" % (name, k)] = np.asarray(a), np.asarray(b)
    s = spec["network"]
    names = []
    in_dim = s["input_dim"]
    for i, out_dim in enumerate(s["layer_dims"]):            # examples/utilities.py:95-99
        names.append("weights_posdef_%d" % i)
        if out_dim > in_dim:
            names.append("weights_%d" % i)
        in_dim = out_dim
    assert len(names) == len(s["weights"])
    numpy_tf.set_named_variables(dict(zip(names, s["weights"])))
    activations = {"tanh": tf.tanh, "relu": tf.nn.relu}
    network = E.LyapunovNetwork(s["input_dim"], s["layer_dims"],
                                [activations[a] for a in s["activations"]], eps=s["eps"])
    out["network/values"] = run(network(s["x"]))
    for k, s in enumerate(spec["tables"]):
        table = F.Triangulation(F.GridWorld(s["limits"], s["num_points"]), s["vertex_values"],
                                project=s["project"])
        out["tables/%d/values" % k] = run(table(s["x"]))
        out["tables/%d/gradient" % k] = run(table.gradient(s["x"]))
    for k, s in enumerate(spec["dlqr"]):
        gain, cost = utilities.dlqr(s["a"], s["b"], s["q"], s["r"])
        out["dlqr/%d/k" % k], out["dlqr/%d/p" % k] = np.asarray(gain), np.asarray(cost)
    out["unique_rows/result"] = utilities.unique_rows(spec["unique_rows"]["array"])
    s = spec["batchify"]
    for k, (start, batches) in enumerate(utilities.batchify(s["arrays"], s["batch_size"])):
        out["batchify/%d/start" % k] = np.int64(start)
        for j, batch in enumerate(batches):
            out["batchify/%d/%d" % (k, j)] = batch
    out["batchify/count"] = np.int64(k + 1)

    arrays = dict(out)
    arrays["_spec"] = np.array(json.dumps(jsonable(spec, arrays, "_in")))
    arrays["_scipy_version"] = np.array(scipy.__version__)
    np.savez_compressed(OUT, **arrays)
    print("wrote %s (%d arrays, %.1f KiB)" % (OUT, len(arrays), os.path.getsize(OUT) / 1024.0))
    for key in sorted(out):
        print("  %-28s %s" % (key, np.shape(out[key])))


if __name__ == "__main__":
    main()
