"""Oracle (test infrastructure): dynamic programming on the grid, NumPy float64.

Restates ``safe_learning/reinforcement_learning.py:26-140`` (``future_values``,
``bellmann_error``, ``value_iteration``) and ``:213-279``
(``discrete_policy_optimization``).  ``optimize_value_function`` (cvxpy LP) is out of scope.
The reference returns TensorFlow assign ops; the oracle performs the assignment directly.
"""

import numpy as np


class PolicyIteration(object):
    """Reference: ``reinforcement_learning.py:46-63``."""

    def __init__(self, policy, dynamics, reward_function, value_function, gamma=0.98):
        self.dynamics = dynamics
        self.reward_function = reward_function
        self.value_function = value_function
        self.gamma = gamma
        self.state_space = self.value_function.discretization.all_points
        self.policy = policy

    def future_values(self, states, policy=None, actions=None, lyapunov=None,
                      lagrange_multiplier=1.):
        """``r(x,u) + gamma V(mean f(x,u))``.  Reference: ``reinforcement_learning.py:65-114``."""
        if actions is None:
            if policy is None:
                policy = self.policy
            actions = policy(states)
        next_states = self.dynamics(states, actions)
        rewards = self.reward_function(states, actions)
        var = None
        if isinstance(next_states, tuple):                                # :98-99 mean only
            next_states, var = next_states
        expected_values = self.value_function(next_states)
        updated_values = rewards + self.gamma * expected_values           # :104
        if lyapunov is not None:                                          # :107-112
            decrease = lyapunov.v_decrease_bound(states, (next_states, var))
            constraint = decrease - lyapunov.threshold(states)
            updated_values = updated_values - lagrange_multiplier * constraint
        return updated_values

    def bellmann_error(self, states):
        """Reference: ``reinforcement_learning.py:116-133``."""
        target = self.future_values(states)
        squares = np.square(target - self.value_function(states)).ravel()
        # left-to-right, like every sum of the oracle (np.sum adds pairwise from 8 terms on)
        return np.cumsum(squares)[-1] if len(squares) else 0.0

    def value_iteration(self):
        """One Jacobi sweep: every read sees the old table.  Reference: ``:135-140``."""
        future_values = self.future_values(self.state_space)
        self.value_function.parameters = future_values
        return future_values

    def discrete_policy_optimization(self, action_space, constraint=None):
        """Arg-max over a finite action set per grid vertex.  Reference: ``:213-279``."""
        states = self.policy.discretization.all_points
        n_states = states.shape[0]
        action_space = np.atleast_2d(np.asarray(action_space, dtype=np.float64))
        n_options, n_actions = action_space.shape
        values = np.empty((n_states, n_options), dtype=np.float64)
        for i, action in enumerate(action_space):                         # :266-275
            action_array = np.broadcast_to(action, (n_states, n_actions))
            values[:, i] = self.future_values(states, actions=action_array)[:, 0]
            if constraint is not None:
                unsafe = np.asarray(constraint(action_array) < 0).reshape(-1)
                values[unsafe, i] = -np.inf
        best = np.argmax(values, axis=1)                                  # :278 first max wins
        self.policy.parameters = action_space[best]
        return values, best
