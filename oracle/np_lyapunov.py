"""Oracle (test infrastructure): Lyapunov safe-set computation, NumPy float64.

Restates ``safe_learning/lyapunov.py:142-606`` (class ``Lyapunov``) and ``:22-56``
(``smallest_boundary_value``).  The TensorFlow graph is replaced by direct calls of NumPy
callables; the batch loop, the prefix rule, the early exit and the ``c_max`` index
arithmetic of ``update_safe_set`` are kept as they are in the reference, quirks included -
and checked against the reference's own ``update_safe_set`` / ``get_safe_sample`` run in the build
container (``tests/golden/make_reference_safe_sets.py``, 35 scenarios incl. the adaptive branch).

Tie order: the reference sorts with NumPy's default (unstable) argsort
(``lyapunov.py:512``); the oracle defines the canonical order as ascending
``(value, flat index)`` - "parity unpinned" for cells with exactly equal values.
"""

import numpy as np

from .np_utilities import batchify
from .np_functions import ordered_rowsum


class _Config(object):
    """``safe_learning/configuration.py:8-32``: float64, 10 000-cell verification batches."""

    def __init__(self):
        self.np_dtype = np.float64
        self.gp_batch_size = 10000


config = _Config()


def smallest_boundary_value(fun, discretization):
    """Minimum of ``fun`` over the grid's boundary faces.  Reference: ``lyapunov.py:22-56``."""
    min_value = np.inf
    for i in range(discretization.ndim):
        tmp = list(discretization.discrete_points)
        tmp[i] = discretization.discrete_points[i][[0, -1]]
        columns = [x.ravel() for x in np.meshgrid(*tmp, indexing='ij')]
        all_points = np.column_stack(columns)
        min_value = min(min_value, np.min(fun(all_points)))
    return min_value


class Lyapunov(object):
    """Reference: ``lyapunov.py:142-225`` (constructor contract kept)."""

    def __init__(self, discretization, lyapunov_function, dynamics, lipschitz_dynamics,
                 lipschitz_lyapunov, tau, policy, initial_set=None, adaptive=False):
        self.discretization = discretization
        self.policy = policy
        self.safe_set = np.zeros(np.prod(discretization.num_points), dtype=bool)
        self.initial_safe_set = initial_set
        if initial_set is not None:
            self.safe_set[initial_set] = True
        self.tau = tau
        self.dynamics = dynamics
        self.lyapunov_function = lyapunov_function
        self.values = None
        self.c_max = 0.
        self._lipschitz_dynamics = lipschitz_dynamics
        self._lipschitz_lyapunov = lipschitz_lyapunov
        self.update_values()
        self.adaptive = adaptive
        self._refinement = np.zeros(discretization.nindex, dtype=int)
        if initial_set is not None:
            self._refinement[initial_set] = 1

    def lipschitz_dynamics(self, states):
        """Reference: ``lyapunov.py:227-244``."""
        if hasattr(self._lipschitz_dynamics, '__call__'):
            return self._lipschitz_dynamics(states)
        return self._lipschitz_dynamics

    def lipschitz_lyapunov(self, states):
        """Reference: ``lyapunov.py:246-263``."""
        if hasattr(self._lipschitz_lyapunov, '__call__'):
            return self._lipschitz_lyapunov(states)
        return self._lipschitz_lyapunov

    def threshold(self, states, tau=None):
        """``-||L_v||_1 (1 + L_f) tau``.  Reference: ``lyapunov.py:265-288``."""
        if tau is None:
            tau = self.tau
        lv = self.lipschitz_lyapunov(states)
        if hasattr(self._lipschitz_lyapunov, '__call__') and lv.shape[1] > 1:
            lv = ordered_rowsum(np.abs(lv))
        lf = self.lipschitz_dynamics(states)
        return - lv * (1. + lf) * tau

    def is_safe(self, state):
        """Reference: ``lyapunov.py:290-303``."""
        return self.safe_set[self.discretization.state_to_index(state)]

    def update_values(self):
        """Reference: ``lyapunov.py:305-322``."""
        self.values = np.asarray(
            self.lyapunov_function(self.discretization.all_points)).squeeze()

    def v_decrease_confidence(self, states, next_states):
        """Reference: ``lyapunov.py:324-354``."""
        if isinstance(next_states, (tuple, list)):
            next_states, error_bounds = next_states
            lv = self.lipschitz_lyapunov(next_states)
            bound = ordered_rowsum(np.atleast_2d(lv * error_bounds))
        else:
            bound = 0.
        v_decrease = self.lyapunov_function(next_states) - self.lyapunov_function(states)
        return v_decrease, bound

    def v_decrease_bound(self, states, next_states):
        """Reference: ``lyapunov.py:356-376``."""
        v_dot, v_dot_error = self.v_decrease_confidence(states, next_states)
        return v_dot + v_dot_error

    def negative(self, states):
        """The per-batch graph of ``lyapunov.py:436-441``: strict ``decrease < threshold``."""
        actions = self.policy(states)
        next_states = self.dynamics(states, actions)
        decrease = self.v_decrease_bound(states, next_states)
        threshold = self.threshold(states, self.tau)
        return np.squeeze(np.less(decrease, threshold), axis=1)

    def _decrease_and_threshold(self, states):
        actions = self.policy(states)
        next_states = self.dynamics(states, actions)
        decrease = self.v_decrease_bound(states, next_states)
        threshold = np.broadcast_to(self.threshold(states, self.tau), decrease.shape)
        return decrease, threshold

    def n_required(self, states, safety_factor):
        """Refinement ``N(x)`` with ``dv < threshold(tau / N)``.  Reference: ``lyapunov.py:445-454``."""
        decrease, threshold = self._decrease_and_threshold(states)
        with np.errstate(divide='ignore', invalid='ignore'):
            ratio = safety_factor * threshold / decrease
        n_req = np.where(np.isnan(ratio), 0., ratio)
        return np.ceil(np.maximum(n_req, 0))

    def refined_negative(self, states, refinement):
        """``refined_safety_check`` mapped over the fed states.  Reference: ``lyapunov.py:459-485``.

        As written in the reference the refined points are built but never evaluated: every row
        compares the decrease of the WHOLE fed batch with its own refined threshold
        ``threshold(center, tau / n_req)`` and reduces with ``all`` - restated as is."""
        decrease, _ = self._decrease_and_threshold(states)
        out = np.zeros(len(states), dtype=bool)
        for k in range(len(states)):
            n_req = int(refinement[k])
            refined_threshold = self.threshold(states[[k]], self.tau / n_req)
            out[k] = np.all(np.less(decrease, refined_threshold))
        return out

    def update_safe_set(self, can_shrink=True, max_refinement=1, safety_factor=1.,
                        parallel_iterations=1):
        """``lyapunov.py:407-606`` including the adaptive branch (``:540-582``)."""
        safety_factor = np.maximum(safety_factor, 1.)                     # :428
        if can_shrink:                                                    # :500-506
            safe_sorted_src = np.zeros_like(self.safe_set, dtype=bool)
            refinement_src = np.zeros_like(self._refinement, dtype=int)
            if self.initial_safe_set is not None:
                safe_sorted_src[self.initial_safe_set] = True
                refinement_src[self.initial_safe_set] = 1
        else:                                                             # :507-510
            safe_sorted_src = self.safe_set
            refinement_src = self._refinement

        order = np.argsort(self.values, kind='stable')                    # :512 (canonical ties)
        safe_sorted = safe_sorted_src[order]                              # :513 (copy)
        refinement = refinement_src[order]

        start = bound = refine_bound = 0
        for start, (indices, safe_batch, refine_batch) in batchify(
                (order, safe_sorted, refinement), config.gp_batch_size):  # :517-524
            states = self.discretization.index_to_state(indices)
            negative = self.negative(states)                              # :529
            safe_batch |= negative                                        # :530
            refine_batch[negative] = 1                                    # :531
            bound = int(np.argmin(safe_batch))                            # :535 first False, 0 if none
            refine_bound = 0
            if bound > 0 or not safe_batch[0]:                            # :539
                if self.adaptive and max_refinement > 1:                  # :540
                    refine_batch[bound:] = self.n_required(states[bound:], safety_factor).ravel()
                    idx_safe = np.logical_or(negative, self.initial_safe_set[indices])   # :547
                    refine_batch[idx_safe] = 1
                    states_to_check = np.logical_and(refine_batch >= 1,
                                                     refine_batch <= max_refinement)[bound:]
                    stop = len(states_to_check) if np.all(states_to_check) \
                        else int(np.argmin(states_to_check))
                    if stop > 0:                                          # :562-575
                        refined_safe = self.refined_negative(states[bound:bound + stop],
                                                             refine_batch[bound:bound + stop])
                        refine_bound = len(refined_safe) if np.all(refined_safe) \
                            else int(np.argmin(refined_safe))
                        safe_batch[bound:bound + refine_bound] = True
                    if stop < len(states_to_check) or refine_bound < stop:    # :579-582
                        safe_batch[bound + refine_bound:] = False
                        refine_batch[bound + refine_bound:] = 0
                        break
                else:
                    safe_batch[bound:] = False                            # :585
                    refine_batch[bound:] = 0                              # :586
                    break                                                 # :587

        max_index = start + bound + refine_bound - 1                      # :590
        self.c_max = self.values[order[max_index]]                        # :595

        self.safe_set[:] = False                                          # :598-601
        self.safe_set[order[safe_sorted]] = True
        self._refinement[order] = refinement
        if self.initial_safe_set is not None:                             # :604-606
            self.safe_set[self.initial_safe_set] = True
            self._refinement[self.initial_safe_set] = 1


def unique_rows(array):
    """Unique rows through a void view (byte-wise order).  Reference: ``utilities.py:496-516``."""
    array = np.ascontiguousarray(array)
    dtype = np.dtype((np.void, array.dtype.itemsize * array.shape[1]))
    combined = array.view(dtype=dtype)
    _, idx = np.unique(combined, return_index=True)
    return array[idx]


def perturb_actions(states, actions, perturbations, limits=None):
    """All (state, action + perturbation) pairs, clipped and de-duplicated.
    Reference: ``lyapunov.py:609-651``."""
    num_states, state_dim = states.shape
    states_new = np.repeat(states, len(perturbations), axis=0)
    actions_new = (np.repeat(actions, len(perturbations), axis=0)
                   + np.tile(perturbations, (num_states, 1)))
    state_actions = np.column_stack((states_new, actions_new))
    if limits is not None:
        limits = np.asarray(limits)
        view = state_actions[:, state_dim:]
        np.clip(view, limits[:, 0], limits[:, 1], out=view)
        state_actions = unique_rows(state_actions)
    return state_actions


def get_safe_sample(lyapunov, perturbations=None, limits=None, positive=False,
                    num_samples=None, actions=None):
    """Most uncertain state-action pair that provably maps back into the level set.
    Reference: ``lyapunov.py:657-797`` (the warning of ``:782-783`` is kept)."""
    import warnings
    safe_idx = np.where(lyapunov.safe_set)
    safe_states = lyapunov.discretization.index_to_state(safe_idx)
    if num_samples is not None and len(safe_states) > num_samples:
        idx = np.random.choice(len(safe_states), num_samples, replace=True)
        safe_states = safe_states[idx]
    if perturbations is None:
        arrays = [arr.ravel() for arr in np.meshgrid(safe_states, actions, indexing='ij')]
        state_actions = np.column_stack(arrays)
        safe_actions = None
    else:
        safe_actions = lyapunov.policy(safe_states)
        state_actions = perturb_actions(safe_states, safe_actions, perturbations=perturbations,
                                        limits=limits)

    def evaluate(state_actions):
        mean, std = lyapunov.dynamics(state_actions)                    # :714
        bound = ordered_rowsum(std)                                     # :715
        lv = lyapunov.lipschitz_lyapunov(mean)                          # :716
        error = ordered_rowsum(np.atleast_2d(lv * std))                 # :717
        future_values = lyapunov.lyapunov_function(mean) + error        # :718-721
        return mean, bound, np.less(future_values, lyapunov.c_max).squeeze(axis=1)

    mean, bound, maps_inside = evaluate(state_actions)
    if not positive:                                                    # :773-776
        next_state_index = lyapunov.discretization.state_to_index(mean)
        maps_inside = maps_inside & lyapunov.safe_set[next_state_index]
    bound_safe = bound[maps_inside]
    if len(bound_safe) == 0:                                            # :780-793
        warnings.warn("No safe state-action pairs found! Using backup policy ...", RuntimeWarning)
        zero_perturbation = np.array([[0.]], dtype=np.float64)
        state_actions = perturb_actions(safe_states, safe_actions, perturbations=zero_perturbation,
                                        limits=limits)
        _, bound, _ = evaluate(state_actions)
        max_id = np.argmax(bound)
        return state_actions[[max_id]], bound[max_id].squeeze()
    max_id = np.argmax(bound_safe)
    return state_actions[maps_inside, :][[max_id]], bound_safe[max_id].squeeze()


def get_lyapunov_region(lyapunov, discretization, init_node):
    """Flood fill of the region around ``init_node`` in which ``lyapunov`` increases monotonically
    along the fill order.  Reference: ``lyapunov.py:59-139`` (heap order, tie-breaking counter,
    boundary stop and pruning of unvisited queue entries kept as they are)."""
    import itertools
    from heapq import heappush, heappop
    values = np.asarray(lyapunov(discretization.all_points)).reshape(discretization.num_points)
    init_node = tuple(int(v) for v in init_node)
    ndim, num_points = discretization.ndim, discretization.num_points
    offsets = np.array(tuple(itertools.product(*[(0, -1, 1) for _ in range(ndim)]))[1:])
    visited = np.zeros(num_points, dtype=bool)
    visited[init_node] = True
    tiebreaker = itertools.count()
    last_value = values[init_node]
    queue = [(values[init_node], next(tiebreaker), np.array(init_node))]
    while queue:
        value, _, node = heappop(queue)
        if np.any(node == 0) or np.any(node == num_points - 1):       # :107-109
            visited[tuple(node)] = False
            break
        if value < last_value:                                        # :112-113
            break
        last_value = value
        neighbors = node + offsets
        is_new = ~visited[tuple(neighbors.T)]
        neighbors = neighbors[is_new]
        if neighbors.size:
            visited[tuple(neighbors.T)] = True
            for nvalue, neighbor in zip(values[tuple(neighbors.T)], neighbors):
                heappush(queue, (nvalue, next(tiebreaker), neighbor))
    for _, _, node in queue:                                          # :136-137
        visited[tuple(node)] = False
    return visited
