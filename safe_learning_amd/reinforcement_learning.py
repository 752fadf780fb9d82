"""``PolicyIteration``: dynamic programming on a GridWorld, executed by the HIP engine.

Drop-in for ``safe_learning/reinforcement_learning.py:26-140, 213-279`` (``future_values``,
``value_iteration``, ``bellmann_error``, ``discrete_policy_optimization``).  The reference
returns TensorFlow ops that the caller runs; here the calls perform the sweep on the GPU.

The value function is a :class:`~safe_learning_amd.functions.Triangulation` whose vertex table
lives on the device; one Jacobi sweep reads the old table (replicated on every GPU) and writes
this rank's shard of the new one, followed by an all-gather.
"""

import numpy as np

from . import _hip
from . import distributed as dist_utils
from ._model import ModelBuilder
from .functions import ConstantFunction, Triangulation

__all__ = ['PolicyIteration', 'OptimizationError']


def _same_grid(a, b):
    return (a is b) or (np.array_equal(a.limits, b.limits)
                        and np.array_equal(a.num_points, b.num_points))


class OptimizationError(Exception):
    """``reinforcement_learning.py:22-23``."""


class PolicyIteration(object):
    """See ``reinforcement_learning.py:26-63`` for the argument meanings."""

    def __init__(self, policy, dynamics, reward_function, value_function, gamma=0.98):
        if not isinstance(value_function, Triangulation):
            raise TypeError('value_function must be a Triangulation')
        self.policy = policy
        self.dynamics = dynamics
        self.reward_function = reward_function
        self.value_function = value_function
        self.gamma = gamma
        self.discretization = value_function.discretization
        self._ctx = _hip.Context()
        self._builder = ModelBuilder(self._ctx, self.discretization)
        n = self.discretization.nindex
        self._rank, self._world = dist_utils.rank_and_world()
        self._bounds = dist_utils.shard_bounds(n, self._world)
        self._lo, self._hi = self._bounds[self._rank], self._bounds[self._rank + 1]
        self.last_residual = None

    def successor_cache(self, max_bytes=-1):
        """Budget of the engine's successor cache (``sl_successor_cache_configure``).

        The next state of (vertex, action) never sees the value table (``:89-104``), so the first
        ``value_iteration(action_space)`` / ``discrete_policy_optimization`` sweep keeps where
        every successor lies in the value grid and the following sweeps - max sweeps over the same
        action set, policy evaluation with a greedy table policy - only gather and combine (bit
        for bit the tables of the uncached sweeps).  ``-1``: default budget (a quarter of the
        GPU's memory), ``0``: recompute every sweep, otherwise a byte limit.  New GP data, other
        dynamics or another action set are noticed by the engine."""
        self._ctx.successor_cache_configure(max_bytes)

    @property
    def successor_cache_info(self):
        """``sl_successor_cache_info``: bytes, validity and hit counters of the cache."""
        return self._ctx.successor_cache_info()

    @property
    def state_space(self):
        """All grid vertices (``reinforcement_learning.py:58-59``)."""
        return self.discretization.all_points

    def _upload(self, policy):
        self._builder.upload(policy, self.dynamics, self.value_function, reward=self.reward_function,
                             gamma=self.gamma)

    def _sweep(self, policy, actions, want_q=False):
        """One pass over this rank's vertices; returns device tensors (v_new, argmax, q, stats)."""
        import torch
        ctx, lo, hi = self._ctx, self._lo, self._hi
        dev = ctx.torch_device
        # a full shard's capacity on every rank: the gathers are one all_gather_into_tensor of
        # equal pieces (distributed.allgather_equal), no pad copy
        count = max(self._bounds[1] - self._bounds[0] if dist_utils.is_distributed() else hi - lo, 1)
        if actions is not None:
            # a sweep over an action set never evaluates the policy: a constant stands in its slot, so
            # that a greedy table adopted by the previous sweep is neither built nor uploaded for it
            width = int(np.atleast_2d(np.asarray(actions)).shape[1])
            if getattr(self, '_placeholder_policy', None) is None or self._placeholder_policy.output_dim != width:
                self._placeholder_policy = ConstantFunction(np.zeros(width))
            policy = self._placeholder_policy
        self._upload(policy)
        v_new = torch.empty(count, dtype=torch.float64, device=dev)
        stats = torch.zeros(2, dtype=torch.float64, device=dev)
        argmax = q = None
        if actions is not None:
            actions = np.atleast_2d(np.asarray(actions, dtype=np.float64))
            argmax = torch.empty(count, dtype=torch.int32, device=dev)
            if want_q:
                q = torch.empty((count, actions.shape[0]), dtype=torch.float64, device=dev)
        ctx.bellman_sweep(lo, hi, actions, v_new, argmax, q, stats)
        return v_new, argmax, q, stats

    def _gather(self, shard, width=1):
        """All ranks' shards (``width`` entries per vertex) as one device tensor."""
        n = self.discretization.nindex
        if not dist_utils.is_distributed():
            return shard.reshape(-1)[:(self._hi - self._lo) * width]
        return dist_utils.allgather_equal(shard, n * width)

    def future_values(self, states=None, policy=None, actions=None, lyapunov=None,
                      lagrange_multiplier=1.):
        """``r(x,u) + gamma V(f(x,u))`` (``:65-114``): one sweep kernel when ``states`` is the grid
        (``None`` or ``state_space``; ``actions`` then is one constant action row), otherwise the
        point-evaluation kernels at the given states."""
        if states is not None and states is not self.state_space:
            states = np.atleast_2d(np.asarray(states, dtype=np.float64))
            if states.shape != self.state_space.shape or not np.array_equal(states, self.state_space):
                return self._future_values_at(states, policy, actions, lyapunov, lagrange_multiplier)
        if actions is not None:
            # the reference evaluates dynamics(states, actions) row by row (:89-104): one row is
            # a constant policy, one row per vertex is a per-vertex action table
            rows = np.atleast_2d(np.asarray(actions, dtype=np.float64))
            if len(rows) == 1 or np.all(rows == rows[0]):
                policy = ConstantFunction(rows[0])
            elif len(rows) == self.discretization.nindex:
                policy = np.ascontiguousarray(rows)
            else:
                raise ValueError('actions must be one row or one row per grid vertex (%d), got %d'
                                 % (self.discretization.nindex, len(rows)))
        elif policy is None:
            policy = self.policy
        v_new, _, _, _ = self._sweep(policy, None)
        updated = self._gather(v_new).cpu().numpy()[:, None]
        if lyapunov is not None:
            # Lyapunov decrease as a soft constraint (:107-112): the sweep above already holds
            # r + gamma V(mean); the penalty needs the posterior (mean, error) of the same
            # dynamics at the vertices, V_lyap and L_v - all through the point-evaluation kernels
            from . import _evaluate
            x = self.state_space
            u = policy if isinstance(policy, np.ndarray) else _evaluate.policy(policy, x)
            next_states = _evaluate.dynamics(self.dynamics, x, u)
            if not isinstance(next_states, tuple):
                raise TypeError('the Lyapunov penalty needs uncertain dynamics (mean, error): the '
                                'reference reads the error bound of the GP (:97-108)')
            decrease = lyapunov.v_decrease_bound(x, next_states)
            constraint = decrease - lyapunov.threshold(x)
            updated = updated - lagrange_multiplier * constraint
        return updated

    def _future_values_at(self, states, policy, actions, lyapunov, lagrange_multiplier):
        """``future_values`` at arbitrary states (``:65-114``) composed from the point-evaluation
        kernels: policy, dynamics (posterior mean), reward, value table."""
        from . import _evaluate
        if actions is not None:
            u = np.array(np.broadcast_to(np.atleast_2d(np.asarray(actions, dtype=np.float64)),
                                         (len(states), np.shape(actions)[-1])))
        else:
            u = _evaluate.policy(self.policy if policy is None else policy, states)
        next_states = _evaluate.dynamics(self.dynamics, states, u)
        mean = next_states[0] if isinstance(next_states, tuple) else next_states
        rewards = _evaluate.value(self.reward_function, np.hstack((states, u)))
        updated = rewards + self.gamma * _evaluate.value(self.value_function, mean)
        if lyapunov is not None:
            if not isinstance(next_states, tuple):
                raise TypeError('the Lyapunov penalty needs uncertain dynamics (mean, error)')
            constraint = lyapunov.v_decrease_bound(states, next_states) - lyapunov.threshold(states)
            updated = updated - lagrange_multiplier * constraint
        return updated

    def value_iteration(self, action_space=None):
        """One Jacobi sweep (``:135-140``); returns ``max |dV|``.

        ``action_space=None`` is the reference's call: ``V <- r + gamma V(f(x, policy(x)))``.
        With a finite ``action_space`` the sweep is the Bellman optimality backup
        ``V <- max_a [r(x, a) + gamma V(f(x, a))]`` and the policy becomes the greedy one - what
        ``discrete_policy_optimization`` followed by ``value_iteration()`` computes in two sweeps.
        Across ranks: each rank writes its shard, the shards are all-gathered (16.8 MB per rank at
        64^4 over 8 GPUs) and the residual is MAX-reduced."""
        if action_space is None:
            v_new, _, _, stats = self._sweep(self.policy, None)
        else:
            action_space = np.atleast_2d(np.asarray(action_space, dtype=np.float64))
            if isinstance(self.policy, Triangulation) and not _same_grid(
                    self.policy.discretization, self.discretization):
                raise ValueError('value_iteration(action_space) adopts the greedy action per VALUE '
                                 'vertex; the policy table lives on another grid - call '
                                 'discrete_policy_optimization() and value_iteration() instead')
            v_new, argmax, _, stats = self._sweep(self.policy, action_space)
            self._adopt_greedy_policy(action_space, argmax)
        full = self._gather(v_new)
        dist_utils.allreduce_max_(stats[:1])
        self.value_function._adopt_device_table(full.reshape(-1, 1).contiguous())
        self.last_residual = float(stats[0])
        return self.last_residual

    def _adopt_greedy_policy(self, action_space, argmax, best=None):
        """Per-vertex action table from arg-max indices: only the int32 indices travel between the
        ranks (4 bytes per vertex), never the ``[N, A]`` table of action values."""
        import torch
        if best is None:
            best = self._gather(argmax)                  # int32 indices index as they are
        # the action table stays on the device across the sweeps of a loop (no upload per sweep)
        key = (action_space.shape, action_space.tobytes(), str(best.device))
        if getattr(self, '_actions_dev_key', None) != key:
            self._actions_dev = torch.from_numpy(np.ascontiguousarray(action_space)).to(best.device)
            self._actions_dev_key = key
        if not isinstance(self.policy, Triangulation):
            self.policy = Triangulation(self.discretization)
        actions_dev, picks = self._actions_dev, best.reshape(-1)
        # The table is built when somebody reads it (the next policy evaluation, `policy.parameters`):
        # in a value-iteration loop the next sweep replaces it unread.  (index_select takes the int32
        # indices as they are; `actions[best]` first converts them to int64 - a second 16.7 M-element
        # kernel at 64^4.)
        self.policy._adopt_lazy_device_table(
            lambda: torch.index_select(actions_dev, 0, picks).contiguous(), actions_dev.shape[1])

    def bellmann_error(self, states=None):
        """``sum (future_values - V)^2`` (``:116-133``): over the grid in one sweep, or at the given
        states through the point-evaluation kernels."""
        if states is not None and states is not self.state_space:
            states = np.atleast_2d(np.asarray(states, dtype=np.float64))
            if states.shape != self.state_space.shape or not np.array_equal(states, self.state_space):
                from . import _evaluate
                target = self._future_values_at(states, None, None, None, 1.)
                return float(np.sum(np.square(target - _evaluate.value(self.value_function, states))))
        _, _, _, stats = self._sweep(self.policy, None)
        dist_utils.allreduce_sum_(stats[1:])
        return float(stats[1])

    def discrete_policy_optimization(self, action_space, constraint=None, return_values=False):
        """Greedy policy over a finite action set (``:213-279``); the first maximiser wins.

        The arg-max is taken inside the sweep kernel and only the indices are all-gathered.  The
        ``[N, A]`` table of action values (``values`` in the reference, 1.2 GB at 64^4 x 9) is
        produced and gathered only on request (``return_values=True``: returned as a device
        tensor) or when a ``constraint`` callback has to veto actions on the host."""
        import torch
        action_space = np.atleast_2d(np.asarray(action_space, dtype=np.float64))
        n_act = action_space.shape[0]
        policy_grid = getattr(self.policy, 'discretization', None)
        if isinstance(self.policy, Triangulation) and not _same_grid(policy_grid,
                                                                     self.discretization):
            return self._discrete_policy_optimization_at(policy_grid, action_space, constraint,
                                                         return_values)
        want_q = return_values or constraint is not None
        _, argmax, q, _ = self._sweep(self.policy, action_space, want_q=want_q)
        from .lyapunov import Lyapunov
        if isinstance(constraint, Lyapunov):
            # The constraint is the Lyapunov decrease condition itself (what the reference's
            # Lyapunov.safety_constraint describes, lyapunov.py:378-406): an action is ruled out at a
            # vertex where the condition fails under it (and the vertex is not in the initial safe
            # set).  One decrease sweep per action writes a bit mask, the arg-max over the rows of the
            # action-value table with those masks runs in a kernel: neither the [N, A] table nor the
            # masks leave the device (sl_argmax_rows_masked).
            if not _same_grid(constraint.discretization, self.discretization):
                raise ValueError('the Lyapunov constraint lives on another grid than the value function')
            count = self._hi - self._lo
            allowed = torch.stack([constraint.decrease_bits(ConstantFunction(action))
                                   for action in action_space]).contiguous()
            argmax = torch.empty_like(argmax)
            # (both contexts enqueue on torch's current stream: the masks are complete when read)
            self._ctx.argmax_rows_masked(count, n_act, q, allowed, allowed.shape[1], argmax)
            self._adopt_greedy_policy(action_space, argmax)
            if not return_values:
                return None
            # `values` of the reference carries -inf at the vetoed entries (:272-275), like the
            # callback path below: unpack this rank's mask words and mask the table before the gather
            shifts = torch.arange(64, dtype=torch.int64, device=q.device)
            bits = ((allowed[:, :, None] >> shifts) & 1).reshape(n_act, -1)[:, :count].t().bool()
            q[:count] = torch.where(bits, q[:count], torch.full_like(q[:count], -float('inf')))
            return self._gather(q, n_act).reshape(-1, n_act)
        q_all = None
        if want_q:
            q_all = self._gather(q, n_act).reshape(-1, n_act)
        best = None
        if constraint is not None:
            # actions whose safety slack is negative at a vertex are ruled out there (:272-275);
            # the callback is the caller's Python, so this part runs on the host like the reference
            n = self.discretization.nindex
            unsafe = np.zeros((n, n_act), dtype=bool)
            for i, action in enumerate(action_space):
                slack = constraint(np.broadcast_to(action, (n, action_space.shape[1])))
                unsafe[:, i] = np.asarray(slack).reshape(-1) < 0
            q_all = torch.where(torch.from_numpy(unsafe).to(q_all.device),
                                torch.full_like(q_all, -float('inf')), q_all)
            best = torch.from_numpy(np.argmax(q_all.cpu().numpy(), axis=1)).to(q_all.device)
        self._adopt_greedy_policy(action_space, argmax, best)
        return q_all if return_values else None

    def _discrete_policy_optimization_at(self, policy_grid, action_space, constraint,
                                         return_values):
        """The policy table lives on another grid than the value table: the reference optimises
        over ``self.policy.discretization.all_points`` (``:227``), which the sweep kernel (one
        thread per VALUE vertex) does not walk - the action values come from the point-evaluation
        kernels, the arg-max and the constraint callback run on the host like the reference's."""
        import torch
        states = policy_grid.all_points
        n, n_act = len(states), action_space.shape[0]
        values = np.empty((n, n_act), dtype=np.float64)
        for i, action in enumerate(action_space):                        # :266-275
            values[:, i] = self._future_values_at(states, None, action[None, :], None, 1.)[:, 0]
            if constraint is not None:
                slack = constraint(np.broadcast_to(action, (n, action_space.shape[1])))
                values[np.asarray(slack).reshape(-1) < 0, i] = -np.inf
        self.policy.parameters = action_space[np.argmax(values, axis=1)]  # :278 first max wins
        if return_values:
            return torch.from_numpy(values).to(self._ctx.torch_device)
        return None

    def optimize_value_function(self, **solver_options):
        """The cvxpy linear program of ``:142-211`` is outside the accelerated path."""
        raise NotImplementedError('optimize_value_function (cvxpy LP) is out of scope')
