"""Translate function specs into the C-ABI model description and upload it.

This is the host-side replacement of the reference's TensorFlow graph construction
(``safe_learning/lyapunov.py:431-443``, ``reinforcement_learning.py:89-104``): instead of
composing callables into a graph, the specs are written into ``sl_model_desc`` and the GP /
table / network payloads are uploaded once (and again only when they change).
"""

import numpy as np

from . import _hip
from .functions import (AbsFunction, CartPole, ConstantFunction, FunctionStack,
                        GaussianProcess, Gradient, InvertedPendulum, LinearSystem, LyapunovNetwork,
                        NeuralNetwork,
                        Norm1Function, QuadraticFunction, Saturation, Triangulation, _gp_heads,
                        _is_plain_rbf)


class ModelBuilder(object):
    """Keeps the engine's model in sync with a set of specs."""

    def __init__(self, ctx, grid):
        self.ctx = ctx
        self.grid = grid
        self._gp_signature = None
        self._gp_placeholder = set()       # heads uploaded for a model without observations
        self._tri_signature = [None, None]
        self._tri_structure = [None, None]
        self._net_signature = None
        self._policy_table = None
        self._policy_net_signature = None

    # ---- pieces --------------------------------------------------------------------------
    def _write_policy(self, desc, policy):
        import torch
        d = self.grid.ndim
        pd = desc.policy
        inner = policy
        pd.saturate = 0
        if isinstance(policy, Saturation):
            inner = policy.fun
        if isinstance(inner, LinearSystem):
            m = inner.output_dim
            if inner.input_dim != d:
                raise ValueError('policy expects %d inputs, the grid has %d dimensions'
                                 % (inner.input_dim, d))
            pd.kind = _hip.POLICY_LINEAR
            for a in range(m):
                for k in range(d):
                    pd.matrix[a][k] = float(inner.matrix[a, k])
        elif isinstance(inner, ConstantFunction):
            m = inner.output_dim
            pd.kind = _hip.POLICY_CONST
            for a in range(m):
                pd.constant[a] = float(inner.constant[a])
        elif isinstance(inner, Triangulation):
            m = inner.output_dim
            # evaluated by interpolation even on its own grid, exactly like the reference
            # (policy(states) in reinforcement_learning.py:92 / lyapunov.py:436)
            pd.kind = _hip.POLICY_TRI
            self._upload_tri(1, inner)
        elif isinstance(inner, NeuralNetwork):
            # evaluated once per cell of a call into an action table (sl_policy_net.hip)
            dims, acts, kernels, biases = inner._layers_for_upload(d)
            m = inner.output_dim
            pd.kind = _hip.POLICY_NETWORK
            signature = (id(inner), inner._version, inner.output_scale)
            if signature != self._policy_net_signature:
                self.ctx.policy_network_set(dims, acts, kernels, biases, inner.output_scale)
                self._policy_net_signature = signature
        elif isinstance(inner, (np.ndarray, torch.Tensor)):
            # one action row per vertex / point; a device tensor is used in place
            if isinstance(inner, torch.Tensor):
                table = inner.to(device=self.ctx.torch_device, dtype=torch.float64)
                table = table.reshape(self.grid.nindex, -1).contiguous()
            else:
                table = torch.from_numpy(np.ascontiguousarray(inner, dtype=np.float64).reshape(
                    self.grid.nindex, -1)).to(self.ctx.torch_device)
            m = table.shape[1]
            pd.kind = _hip.POLICY_TABLE
            self._policy_table = table
            pd.d_table = self._policy_table.data_ptr()
        else:
            raise TypeError('unsupported policy spec %r: use LinearSystem, Saturation(LinearSystem), '
                            'Triangulation, NeuralNetwork, ConstantFunction or a per-vertex ndarray' % (policy,))
        if m > _hip.MAX_ACTION_DIM:
            raise ValueError('at most %d action dimensions are supported' % _hip.MAX_ACTION_DIM)
        pd.m = m
        if isinstance(policy, Saturation):
            pd.saturate = 1
            lower = np.broadcast_to(np.asarray(policy.lower, dtype=np.float64), (m,))
            upper = np.broadcast_to(np.asarray(policy.upper, dtype=np.float64), (m,))
            for a in range(m):
                pd.lower[a], pd.upper[a] = float(lower[a]), float(upper[a])
        return m

    def _write_dynamics(self, desc, dynamics, m):
        d = self.grid.ndim
        dd = desc.dynamics
        if isinstance(dynamics, LinearSystem):
            if dynamics.matrix.shape != (d, d + m):
                raise ValueError('linear dynamics must be %d x %d' % (d, d + m))
            dd.kind = _hip.DYN_LINEAR
            for i in range(d):
                for j in range(d + m):
                    dd.matrix[i][j] = float(dynamics.matrix[i, j])
        elif isinstance(dynamics, (InvertedPendulum, CartPole)):
            dynamics._write_dynamics(dd)
        elif isinstance(dynamics, (GaussianProcess, FunctionStack)):
            dd.kind = _hip.DYN_GP
            heads, beta = _gp_heads(dynamics)
            # prior mean rows m(x*) (functions.py:439), one row per output column
            for gp, _, col0 in heads:
                if gp.mean_function is not None:
                    mat = gp.mean_function.matrix
                    for r in range(mat.shape[0]):
                        for j in range(d + m):
                            dd.matrix[col0 + r][j] = float(mat[r, j])
            self._upload_gp(heads, beta, d + m)
        else:
            raise TypeError('unsupported dynamics spec %r' % (dynamics,))

    def _upload_gp(self, heads, beta, p):
        # _version / _table_version are process-wide unique tokens (functions._TOKENS), never reused
        signature = tuple((gp._version, col0) for gp, _, col0 in heads) + (beta,)
        if signature == self._gp_signature:
            return
        old = self._gp_signature
        same_heads = (old is not None and len(old) == len(signature) and old[-1] == beta
                      and all(o[1] == s[1] for o, s in zip(old[:-1], signature[:-1])))
        for h, (gp, _, col0) in enumerate(heads):
            if gp.X.shape[1] != p:
                raise ValueError('GP inputs have %d columns, expected state+action = %d'
                                 % (gp.X.shape[1], p))
            if same_heads and old[h][0] == gp._version:
                continue                                # this head is already on the device
            if (same_heads and h not in self._gp_placeholder
                    and self._follow_appends(h, gp, old[h][0])):
                continue                                # add_data_point: new rows only
            X, Linv, alpha = gp.X, gp.cholesky_inverse, gp.alpha
            self._gp_placeholder.discard(h)
            if len(X) == 0:
                self._gp_placeholder.add(h)             # nothing to extend by rank-one rows later
                # a model without observations (the notebooks start that way,
                # inverted_pendulum.ipynb:166-176): the posterior is the prior - one training
                # point with a zero row of L^-1 and a zero alpha contributes nothing
                X, Linv, alpha = np.zeros((1, p)), np.zeros((1, 1)), np.zeros((1, gp.Y.shape[1]))
            if _is_plain_rbf(gp.kern, p):
                self.ctx.gp_set_head(h, X, Linv, alpha, col0, gp.kern.variance, gp.kern.lengthscales)
            else:                                       # Linear / Matern32 / sums and products
                self.ctx.gp_set_head_kernel(h, X, Linv, alpha, col0, gp.kern._factors(p))
        self.ctx.gp_configure(len(heads), beta)
        self._gp_signature = signature

    def _follow_appends(self, head, gp, uploaded_version):
        """Bring an uploaded head up to date through ``GPRCached.append_data``'s log: one new row
        of L^-1, one alpha row and one input per point (O(n) bytes) instead of re-packing and
        re-uploading the whole factor (8 MB at n = 1024).  False when the log does not lead from
        the uploaded state to the current one or the head's padded capacity is exhausted."""
        chain, version = [], uploaded_version
        for before, after, x, row, alpha_new in gp._append_log:
            if before == version:
                chain.append((x, row, alpha_new))
                version = after
        if version != gp._version or not chain:
            return False
        for x, row, alpha_new in chain:
            if not self.ctx.gp_append_point(head, x, row, alpha_new):
                return False
        return True

    def _upload_tri(self, slot, tri):
        signature = (tri._table_version, tri.project)
        if signature == self._tri_signature[slot]:
            return
        table = tri._device(self.ctx)
        structure = (tri._structure_token, tri.project, int(table.shape[1]))
        if structure == self._tri_structure[slot]:
            # same grid and simplices, new vertex values (every value-iteration sweep): only the
            # table pointer of the descriptor changes
            self.ctx.tri_set_table(slot, table)
        else:
            tri._upload(self.ctx, slot)
            self._tri_structure[slot] = structure
        self._tri_signature[slot] = signature

    def _write_value(self, vd, fun):
        if isinstance(fun, QuadraticFunction):
            fun._write_value(vd)
        elif isinstance(fun, Triangulation):
            vd.kind = _hip.V_TRI
            vd.negate = int(fun.negate)
            self._upload_tri(0, fun)
        elif isinstance(fun, LyapunovNetwork):
            vd.kind = _hip.V_NETWORK
            vd.negate = int(fun.negate)
            signature = (fun.eps, tuple(fun.activations), tuple(fun.output_dims),
                         tuple(w.tobytes() for w in fun.weights))
            if signature != self._net_signature:
                fun._upload(self.ctx)
                self._net_signature = signature
        else:
            raise TypeError('unsupported value-function spec %r' % (fun,))

    def _write_lipschitz(self, ld, lipschitz_lyapunov, lipschitz_dynamics, tau):
        d = self.grid.ndim
        if isinstance(lipschitz_lyapunov, Norm1Function) and lipschitz_lyapunov.constant != 0.0:
            raise TypeError('a constant offset is only supported for lipschitz_dynamics')
        if (isinstance(lipschitz_lyapunov, (AbsFunction, Norm1Function))
                and isinstance(lipschitz_lyapunov.fun, Gradient)):
            ld.lv_kind = (_hip.LIP_ABS_GRAD if isinstance(lipschitz_lyapunov, AbsFunction)
                          else _hip.LIP_NORM_GRAD)
        elif isinstance(lipschitz_lyapunov, (AbsFunction, Norm1Function)):
            inner = lipschitz_lyapunov.fun
            if not isinstance(inner, LinearSystem) or inner.matrix.shape != (d, d):
                raise TypeError('AbsFunction / Norm1Function need a %d x %d LinearSystem' % (d, d))
            ld.lv_kind = (_hip.LIP_ABS_LINEAR if isinstance(lipschitz_lyapunov, AbsFunction)
                          else _hip.LIP_NORM_LINEAR)
            for i in range(d):
                for j in range(d):
                    ld.lv_matrix[i][j] = float(inner.matrix[i, j])
        elif np.isscalar(lipschitz_lyapunov):
            ld.lv_kind = _hip.LIP_CONST
            ld.lv_const = float(lipschitz_lyapunov)
        else:
            raise TypeError('lipschitz_lyapunov must be a float, AbsFunction(LinearSystem), '
                            'Norm1Function(LinearSystem), AbsFunction/Norm1Function(Gradient(V)); arbitrary Python '
                            'callables cannot run inside a GPU kernel')
        if np.isscalar(lipschitz_dynamics):
            ld.lf_kind = _hip.LF_CONST
            ld.lf_const = float(lipschitz_dynamics)
        elif (isinstance(lipschitz_dynamics, Norm1Function)
              and isinstance(lipschitz_dynamics.fun, LinearSystem)
              and lipschitz_dynamics.fun.matrix.shape[1] == d
              and lipschitz_dynamics.fun.matrix.shape[0] <= _hip.MAX_STATE_DIM):
            # state-dependent L_f(x) = c + ||M x||_1 (lyapunov.py:227-244 allows a callable)
            mat = np.zeros((d, d))
            rows = lipschitz_dynamics.fun.matrix
            if rows.shape[0] > d:
                raise TypeError('the L_f matrix may have at most %d rows' % d)
            mat[:rows.shape[0]] = rows
            ld.lf_kind = _hip.LF_AFFINE_NORM1
            ld.lf_const = float(lipschitz_dynamics.constant)
            for i in range(d):
                for j in range(d):
                    ld.lf_matrix[i][j] = float(mat[i, j])
        else:
            raise TypeError('lipschitz_dynamics must be a scalar or c + Norm1Function(LinearSystem(M)) '
                            '(arbitrary Python callables cannot run inside a GPU kernel)')
        ld.tau = float(tau)

    # ---- whole model -----------------------------------------------------------------------
    def upload(self, policy, dynamics, value_function, lipschitz_lyapunov=0.0,
               lipschitz_dynamics=0.0, tau=0.0, reward=None, gamma=0.0):
        desc = _hip.ModelDesc()
        desc.grid = self.grid._desc()
        m = self._write_policy(desc, policy)
        self._write_dynamics(desc, dynamics, m)
        self._write_value(desc.value, value_function)
        self._write_lipschitz(desc.lipschitz, lipschitz_lyapunov, lipschitz_dynamics, tau)
        if reward is not None:
            if not isinstance(reward, QuadraticFunction):
                raise TypeError('reward_function must be a QuadraticFunction on [x, u]')
            reward._write_value(desc.reward)
        desc.gamma = float(gamma)
        self.ctx.model_set(desc)
        self._desc = desc
        return desc
