"""Synthetic problem definitions of BASELINE.json's configs.

``make_case`` returns a plain dict of NumPy parameters (no engine objects) so that the very same
numbers can be fed to the HIP engine (``build_lyapunov`` below) and, in tests / the CPU baseline,
to the oracle (``tests/cases.py``).  Provenance of the constants: the pendulum and its
normalisation are ``examples/adaptive_safety_verification.ipynb`` cells 7-17, the cart-pole is
``examples/reinforcement_learning_cartpole.ipynb`` cell 7; grid sizes, GP sizes and the RBF kernel
are BASELINE.json's.
"""

import numpy as np

from .functions import _cartpole_linearize, _pendulum_linearize
from .utilities import dlqr

GRAVITY = 9.81

# GP hyper-parameter sets (signal / noise standard deviation, lengthscale of every input).
#   survey   : SURVEY.md 8d's literal values.  With 1024 uniformly random training points in the
#              5-D cart-pole input box the posterior stays at the prior (beta sigma ~ 0.1 per
#              output against a decrease of ~ -5e-3): no cell passes the check.
#   informed : signal = the standard deviation of what the GP has to model (true dynamics minus
#              the linear prior: 0.026 in the fastest cart-pole state), low measurement noise and
#              a lengthscale that matches the smooth residual: the safe set grows.
#   tight    : an almost-known model (tiny residual prior): large safe sets on coarse test grids.
GP_VARIANTS = {
    'survey': dict(signal_std=0.05, noise_std=0.01, lengthscale=0.5),
    'informed': dict(signal_std=0.03, noise_std=0.0005, lengthscale=1.5),
    'tight': dict(signal_std=0.001, noise_std=0.0002, lengthscale=1.0),
}


def _true_dynamics_numpy(case, X):
    """True system used to generate GP targets (host NumPy; data generation only)."""
    dyn = case['true_dynamics']
    x, u = X[:, :case['d']].copy(), X[:, case['d']:].copy()
    norm = dyn['normalization']
    x, u = x * np.asarray(norm[0]), u * np.asarray(norm[1])
    dt = dyn['dt'] / 10
    if dyn['kind'] == 'pendulum':
        inertia = dyn['mass'] * dyn['length'] ** 2
        for _ in range(10):
            acc = GRAVITY / dyn['length'] * np.sin(x[:, [0]]) + u / inertia
            acc = acc - dyn['friction'] / inertia * x[:, [1]]
            x = x + dt * np.concatenate((x[:, [1]], acc), axis=1)
    else:
        m, M, L, b = dyn['pendulum_mass'], dyn['cart_mass'], dyn['length'], dyn['rot_friction']
        for _ in range(10):
            th, v, om = x[:, [1]], x[:, [2]], x[:, [3]]
            s, c = np.sin(th), np.cos(th)
            det = L * (M + m * s * s)
            vd = (u - m * L * om * om * s - b * om * c + 0.5 * m * GRAVITY * L * np.sin(2 * th)) * L / det
            wd = (u * c - 0.5 * m * L * om * om * np.sin(2 * th) - b * (m + M) * om / (m * L)
                  + (m + M) * GRAVITY * s) / det
            x = x + dt * np.concatenate((v, om, vd, wd), axis=1)
    return x / np.asarray(norm[0])


def make_case(name, num_points=None, n_gp=None, dynamics=None, seed=0, stack=False,
              tau_scale=1.0, noise_std=0.01, signal_std=0.05, lengthscale=0.5,
              initial_radius=0.2):
    """Parameters of one synthetic configuration.

    name: '1d' (C1), 'pendulum' (C2/C3 family, d=2) or 'cartpole' (C4/C5 family, d=4).
    dynamics: 'linear', 'analytic' (Euler pendulum / cart-pole) or 'gp' (default per family).
    """
    case = {'name': name, 'stack': bool(stack)}
    if name == '1d':
        # safe_learning/tests/test_lyapunov.py:48-74 scaled to 1001 cells
        n = 1001 if num_points is None else num_points
        case.update(d=1, m=1, limits=[[-1., 1.]], num_points=[n],
                    K=np.array([[-0.1]]), saturate=None,
                    dynamics={'kind': 'linear', 'matrix': np.array([[1., 1.]])},
                    P=np.array([[1.]]), lv=('const', 0.3), lf=0.4, tau=0.5 * tau_scale,
                    initial_set=np.array([n // 2]))
        return case

    if name == 'pendulum':
        d, dt = 2, 0.01
        theta_max, omega_max = np.deg2rad(30), np.sqrt(GRAVITY / 0.5)
        u_max = GRAVITY * 0.15 * 0.5 * np.sin(theta_max)
        norm = [(theta_max, omega_max), (u_max,)]
        true = {'kind': 'pendulum', 'mass': 0.15, 'length': 0.5, 'friction': 0.1, 'dt': dt,
                'normalization': norm}
        A_true, B_true = _pendulum_linearize(0.15, 0.5, 0.1, dt, norm)
        A_prior, B_prior = _pendulum_linearize(0.1, 0.4, 0.0, dt, norm)
        Q, R = np.diag([1., 2.]), 1.2 * np.eye(1)
        default_points = 256
    elif name == 'cartpole':
        d, dt = 4, 0.01
        m_p, m_c, length, fric = 0.175, 1.732, 0.28, 0.01
        norm = [(0.5, np.deg2rad(30), 2., np.deg2rad(30)), ((m_p + m_c) * 4 / 0.5,)]
        true = {'kind': 'cartpole', 'pendulum_mass': m_p, 'cart_mass': m_c, 'length': length,
                'rot_friction': fric, 'dt': dt, 'normalization': norm}
        A_true, B_true = _cartpole_linearize(m_p, m_c, length, fric, dt, norm)
        A_prior, B_prior = A_true, B_true
        Q, R = 0.1 * np.eye(4), 0.1 * np.eye(1)
        default_points = 128
    else:
        raise ValueError(name)

    n = default_points if num_points is None else num_points
    num_points = np.broadcast_to(n, (d,)).astype(int)
    K, P = dlqr(A_true, B_true, Q, R)
    P = P / np.abs(P).max()
    unit = 2.0 / (num_points - 1)
    case.update(d=d, m=1, limits=[[-1., 1.]] * d, num_points=list(int(v) for v in num_points),
                K=-K, saturate=(-1., 1.), P=P, lv=('abs_linear', 2 * P),
                lf=float(np.linalg.norm(A_true, 1) + np.linalg.norm(B_true, 1) * np.linalg.norm(-K, 1)),
                tau=float(np.sum(unit) / 2) * tau_scale, initial_radius=initial_radius,
                true_dynamics=true, A_true=A_true, B_true=B_true)
    kind = dynamics or 'gp'
    if kind == 'linear':
        case['dynamics'] = {'kind': 'linear', 'matrix': np.hstack((A_true, B_true))}
    elif kind == 'analytic':
        case['dynamics'] = dict(true)
    elif kind == 'gp':
        n_gp = (512 if name == 'pendulum' else 1024) if n_gp is None else n_gp
        p = d + 1
        X = np.random.default_rng(seed).uniform(-1, 1, (n_gp, p))
        Y = _true_dynamics_numpy(case, X) + np.random.default_rng(seed + 1).normal(
            0, noise_std, (n_gp, d))
        ls = np.full(p, lengthscale)
        if stack:
            # notebook style: one GP per output with its own lengthscales (functions.py:278-291)
            ls = np.stack([np.full(p, lengthscale * (1 + 0.25 * k)) for k in range(d)])
        case['dynamics'] = {'kind': 'gp', 'X': X, 'Y': Y, 'variance': signal_std ** 2,
                            'lengthscales': ls, 'noise_variance': noise_std ** 2,
                            'prior': np.hstack((A_prior, B_prior)), 'beta': 2.0}
    else:
        raise ValueError(kind)
    return case


HEADLINE_GP = 'informed'
HEADLINE_TAU_SCALE = 0.0005


def headline_case(num_points=128, n_gp=1024, family='cartpole', stack=False, variant=None):
    """BASELINE.json's headline workload (configs[3]): cart-pole 128^4 cells, 1024-point shared
    RBF GP over [x, u].  Grid, GP size, kernel, policy, V, L_v, L_f follow SURVEY.md 8d; the GP
    hyper-parameters are the ``informed`` set and tau = HEADLINE_TAU_SCALE x (sum of the grid
    spacings / 2), because with 8d's literal values (and the full tau) no cell passes the decrease
    check, which would make mask parity vacuous: measured on the 128^4 grid, 8d's values give 1
    passing cell of 2.7e8 and a safe set equal to the initial set; this workload has 8.5e7 passing
    cells and a level set that grows by 1.3e4 cells.  The cost of a sweep does not depend on the
    hyper-parameters (the kernel has no early exit)."""
    hyper = GP_VARIANTS[HEADLINE_GP if variant is None else variant]
    return make_case(family, num_points=num_points, n_gp=n_gp, stack=stack,
                     tau_scale=HEADLINE_TAU_SCALE, **hyper)


def table_case(num_points=(2001, 1501), table_points=(101, 101), n_gp=128, tau_scale=0.0005,
               dynamics=None, stack=False, limits=None):
    """The table-V / table-policy Lyapunov sweep of ``inverted_pendulum.ipynb`` (cell 14:
    ``lyapunov_function = -rl.value_function``, ``L_v = |gradient|``, the policy a Triangulation
    on the value grid; ``:112`` runs it on 2001 x 1501 cells): pendulum, RBF GP dynamics, V a
    piecewise-linear table (a quadratic plus a smooth bump, so the simplices differ), policy the
    saturated LQR law sampled on the same table grid."""
    case = make_case('pendulum', num_points=list(num_points), n_gp=n_gp, tau_scale=tau_scale,
                     dynamics=dynamics, stack=stack, **GP_VARIANTS['informed'])
    if limits is not None:
        case['limits'] = [[float(lo), float(hi)] for lo, hi in limits]
    axes = [np.linspace(lo, hi, n) for (lo, hi), n in zip(case['limits'], table_points)]
    pts = np.stack(np.meshgrid(*axes, indexing='ij'), axis=-1).reshape(-1, case['d'])
    vals = np.einsum('ij,jk,ik->i', pts, case['P'], pts)
    vals = vals * (1.0 + 0.05 * np.sin(3.0 * pts[:, 0]) * np.cos(2.0 * pts[:, 1]))
    case['V'] = {'kind': 'table', 'values': vals[:, None], 'project': True,
                 'num_points': list(table_points)}
    case['lv'] = ('abs_grad',)
    act = pts @ case['K'].T
    if case['saturate'] is not None:
        act = np.clip(act, *case['saturate'])
    case['policy_table'] = {'num_points': list(table_points), 'values': act}
    return case


def kernel_from_products(products, leaves):
    """A kernel object from its description as a sum of products of leaves:
    ``[[(kind, constructor keywords), ...], ...]`` with ``kind`` in ``rbf`` / ``matern32`` /
    ``linear`` and ``leaves`` mapping the kinds to constructors with gpflow 0.4.0's signatures
    (``safe_learning_amd.kernels``, the oracle's classes, the gpflow stand-in of the fixtures)."""
    total = None
    for product in products:
        term = None
        for kind, kwargs in product:
            leaf = leaves[kind](**kwargs)
            term = leaf if term is None else term * leaf
        total = term if total is None else total + term
    return total


def notebook_kernels(case):
    """One kernel per output column as ``examples/inverted_pendulum.ipynb:145-158`` builds them:
    ``Linear(3, variances, ARD) + Matern32(1, active_dims=[0]) * Linear(1, variances[1])`` with
    ``variances`` = squared difference of the true and the prior linearisation, at least 1e-5."""
    m_true = np.hstack((case['A_true'], case['B_true']))
    variances = np.clip((m_true - case['dynamics']['prior']) ** 2, 1e-5, None)
    p = case['d'] + 1
    return [[[('linear', dict(input_dim=p, variance=[float(v) for v in variances[k]], ARD=True))],
             [('matern32', dict(input_dim=1, lengthscales=1.0, active_dims=[0])),
              ('linear', dict(input_dim=1, variance=float(variances[k, 1])))]]
            for k in range(case['d'])]


def initial_safe_mask(case):
    """``||x||_2 <= radius`` on the grid without materialising all points
    (``adaptive_safety_verification.ipynb`` cell 11)."""
    if 'initial_set' in case:
        return case['initial_set']
    axes = [np.linspace(lo, hi, n) ** 2 for (lo, hi), n in zip(case['limits'], case['num_points'])]
    total = axes[0]
    for ax in axes[1:]:
        total = np.add.outer(total, ax)
    return (np.sqrt(total) <= case['initial_radius']).ravel()


def build_specs(case):
    """Engine specs (policy, dynamics, V, L_v) of a case."""
    from . import functions as F
    if 'policy_table' in case:
        # piecewise-linear policy of the RL loop (inverted_pendulum.ipynb cells 9-13)
        tab = case['policy_table']
        policy = F.Triangulation(F.GridWorld(case['limits'], tab['num_points']), tab['values'])
    else:
        policy = F.LinearSystem((case['K'],))
    if case['saturate'] is not None:
        policy = F.Saturation(policy, *case['saturate'])
    dyn = case['dynamics']
    if dyn['kind'] == 'linear':
        dynamics = F.LinearSystem((dyn['matrix'],))
    elif dyn['kind'] == 'pendulum':
        dynamics = F.InvertedPendulum(dyn['mass'], dyn['length'], dyn['friction'], dyn['dt'],
                                      dyn['normalization'])
    elif dyn['kind'] == 'cartpole':
        dynamics = F.CartPole(dyn['pendulum_mass'], dyn['cart_mass'], dyn['length'],
                              dyn['rot_friction'], dyn['dt'], dyn['normalization'])
    else:
        d = case['d']
        if case['stack']:
            heads = []
            for k in range(d):
                if 'kernels' in dyn:                   # notebook_kernels(): Linear + Matern32 * Linear
                    kern = kernel_from_products(dyn['kernels'][k], {'rbf': F.RBF, 'matern32': F.Matern32,
                                                                    'linear': F.Linear})
                else:
                    kern = F.RBF(d + 1, dyn['variance'], dyn['lengthscales'][k], ARD=True)
                gp = F.GPRCached(dyn['X'], dyn['Y'][:, [k]], kern,
                                 F.LinearSystem((dyn['prior'][[k], :],)),
                                 likelihood_variance=dyn['noise_variance'])
                heads.append(F.GaussianProcess(gp, dyn['beta']))
            dynamics = F.FunctionStack(heads)
        else:
            kern = F.RBF(d + 1, dyn['variance'], dyn['lengthscales'], ARD=True)
            gp = F.GPRCached(dyn['X'], dyn['Y'], kern, F.LinearSystem((dyn['prior'],)),
                             likelihood_variance=dyn['noise_variance'])
            dynamics = F.GaussianProcess(gp, dyn['beta'])
    vspec = case.get('V', {'kind': 'quadratic'})
    if vspec['kind'] == 'quadratic':
        value = F.QuadraticFunction(case['P'])
    elif vspec['kind'] == 'network':
        value = F.LyapunovNetwork(case['d'], vspec['layer_dims'], vspec['activations'],
                                  vspec['eps'], vspec['weights'])
    else:
        value = F.Triangulation(F.GridWorld(case['limits'],
                                            vspec.get('num_points', case['num_points'])),
                                vspec['values'], project=vspec.get('project', False))
    kind, arg = (case['lv'] + (None,))[:2]
    if kind == 'const':
        lv = arg
    elif kind == 'abs_linear':
        lv = F.AbsFunction(F.LinearSystem((arg,)))
    elif kind == 'norm_linear':
        lv = F.Norm1Function(F.LinearSystem((arg,)))
    elif kind == 'abs_grad':
        lv = F.AbsFunction(F.Gradient(value))
    else:
        lv = F.Norm1Function(F.Gradient(value))
    return policy, dynamics, value, lv


def network_weights(input_dim, layer_dims, seed=1):
    """Xavier-uniform weights in the variable order of ``examples/utilities.py:95-99``."""
    rng = np.random.default_rng(seed)
    weights, in_dim = [], input_dim
    for out_dim in layer_dims:
        hidden = int(np.ceil((in_dim + 1) / 2))
        shapes = [(hidden, in_dim)] + ([(out_dim - in_dim, in_dim)] if out_dim > in_dim else [])
        for s in shapes:
            weights.append(rng.uniform(-1, 1, s) * np.sqrt(6. / (s[0] + s[1])))
        in_dim = out_dim
    return weights


def build_lyapunov(case):
    """``safe_learning_amd.Lyapunov`` of a case (runs ``update_values`` like the reference)."""
    from .functions import GridWorld
    from .lyapunov import Lyapunov
    grid = GridWorld(case['limits'], case['num_points'])
    policy, dynamics, value, lv = build_specs(case)
    # The initial set as a READ-ONLY array of its own: Lyapunov identifies such a mask by identity
    # instead of hashing its content on every update_safe_set (a writable one - the reference reads
    # it afresh every call - costs 0.15 ms per MB per update: 40 ms for the 268 MB of 128^4).
    initial = np.array(initial_safe_mask(case), copy=True)
    initial.flags.writeable = False
    return Lyapunov(grid, value, dynamics, case['lf'], lv, case['tau'], policy, initial_set=initial)
