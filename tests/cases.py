"""Build ORACLE objects (test infrastructure) from the parameter dicts of
``safe_learning_amd.benchmarks.make_case`` - the same numbers the HIP engine gets."""

import numpy as np

import oracle
from safe_learning_amd.benchmarks import initial_safe_mask, make_case  # noqa: F401


ORACLE_KERNEL_LEAVES = {'rbf': oracle.np_functions.SlicedRBF, 'matern32': oracle.np_functions.Matern32,
                        'linear': oracle.np_functions.Linear}


def oracle_specs(case):
    if 'policy_table' in case:
        tab = case['policy_table']
        policy = oracle.Triangulation(oracle.GridWorld(case['limits'], tab['num_points']),
                                      tab['values'])
    else:
        policy = oracle.LinearSystem((case['K'],))
    if case['saturate'] is not None:
        policy = oracle.Saturation(policy, *case['saturate'])
    dyn = case['dynamics']
    if dyn['kind'] == 'linear':
        dynamics = oracle.LinearSystem((dyn['matrix'],))
    elif dyn['kind'] == 'pendulum':
        dynamics = oracle.InvertedPendulum(dyn['mass'], dyn['length'], dyn['friction'], dyn['dt'],
                                           dyn['normalization'])
    elif dyn['kind'] == 'cartpole':
        dynamics = oracle.CartPole(dyn['pendulum_mass'], dyn['cart_mass'], dyn['length'],
                                   dyn['rot_friction'], dyn['dt'], dyn['normalization'])
    else:
        d = case['d']
        if case['stack']:
            heads = []
            for k in range(d):
                if 'kernels' in dyn:
                    from safe_learning_amd.benchmarks import kernel_from_products
                    kern = kernel_from_products(dyn['kernels'][k], ORACLE_KERNEL_LEAVES)
                else:
                    kern = oracle.RBF(d + 1, dyn['variance'], dyn['lengthscales'][k], ARD=True)
                gp = oracle.GPRCached(dyn['X'], dyn['Y'][:, [k]], kern,
                                      oracle.LinearSystem((dyn['prior'][[k], :],)),
                                      likelihood_variance=dyn['noise_variance'])
                heads.append(oracle.GaussianProcess(gp, dyn['beta']))
            dynamics = oracle.FunctionStack(heads)
        else:
            kern = oracle.RBF(d + 1, dyn['variance'], dyn['lengthscales'], ARD=True)
            gp = oracle.GPRCached(dyn['X'], dyn['Y'], kern, oracle.LinearSystem((dyn['prior'],)),
                                  likelihood_variance=dyn['noise_variance'])
            dynamics = oracle.GaussianProcess(gp, dyn['beta'])
    vspec = case.get('V', {'kind': 'quadratic'})
    if vspec['kind'] == 'quadratic':
        value = oracle.QuadraticFunction(case['P'])
        gradient = None
    elif vspec['kind'] == 'network':
        value = oracle.LyapunovNetwork(case['d'], vspec['layer_dims'], vspec['activations'],
                                       vspec['eps'], vspec['weights'])
        gradient = value.gradient
    else:
        value = oracle.Triangulation(oracle.GridWorld(case['limits'],
                                                      vspec.get('num_points', case['num_points'])),
                                     vspec['values'], project=vspec.get('project', False))
        gradient = value.gradient
    kind, arg = (case['lv'] + (None,))[:2]
    if kind == 'const':
        lv = arg
    elif kind == 'abs_linear':
        lv = oracle.AbsFunction(oracle.LinearSystem((arg,)))
    elif kind == 'norm_linear':
        lv = oracle.Norm1Function(oracle.LinearSystem((arg,)))
    elif kind == 'abs_grad':
        lv = oracle.AbsFunction(gradient)
    else:
        lv = oracle.Norm1Function(gradient)
    return policy, dynamics, value, lv


class _LyapunovWithoutValues(oracle.Lyapunov):
    """For timing samples of huge grids: skip the all_points value table (8.6 GB at 128^4)."""

    def update_values(self):
        self.values = None


def oracle_lyapunov(case, compute_values=True, dynamics=None):
    """``dynamics``: a model to use instead of the case's own (kernel cases of gp_cases.py)."""
    grid = oracle.GridWorld(case['limits'], case['num_points'])
    policy, own_dynamics, value, lv = oracle_specs(case)
    dynamics = own_dynamics if dynamics is None else dynamics
    cls = oracle.Lyapunov if compute_values else _LyapunovWithoutValues
    return cls(grid, value, dynamics, case['lf'], lv, case['tau'], policy,
               initial_set=initial_safe_mask(case) if compute_values else None)


def oracle_cell_records(lyap, indices):
    """Per-cell [decrease, threshold, mean[d], err[d]] of the oracle (same layout as the
    engine's debug records)."""
    states = lyap.discretization.index_to_state(indices)
    actions = lyap.policy(states)
    nxt = lyap.dynamics(states, actions)
    decrease = lyap.v_decrease_bound(states, nxt)
    threshold = lyap.threshold(states, lyap.tau)
    if isinstance(nxt, tuple):
        mean, err = nxt
    else:
        mean, err = nxt, np.zeros_like(nxt)
    threshold = np.broadcast_to(threshold, decrease.shape)
    return np.hstack((decrease, threshold, mean, err))


def lyapunov_like_network_weights(P, layer_dims, scale=0.5, eps=1e-8):
    """LyapunovNetwork weights whose V is close to ``scale^2 x^T P x`` near the origin (a network
    that IS a Lyapunov candidate, unlike a randomly initialised one): the first layer's kernel
    ``W^T W + eps I`` equals ``scale^2 P`` and the remaining kernels pass the leading components
    through.  Variable order of ``examples/utilities.py:95-99``."""
    P = np.asarray(P, dtype=np.float64)
    d = len(P)
    weights, in_dim = [], d
    for i, out_dim in enumerate(layer_dims):
        hidden = int(np.ceil((in_dim + 1) / 2))
        W = np.zeros((hidden, in_dim))
        if i == 0:
            assert hidden >= d
            W[:d, :d] = np.linalg.cholesky(scale ** 2 * P - eps * np.eye(d)).T
        else:
            k = min(hidden, in_dim)
            W[:k, :k] = np.eye(k)
        weights.append(W)
        if out_dim > in_dim:
            weights.append(np.zeros((out_dim - in_dim, in_dim)))
        in_dim = out_dim
    return weights
