"""Evaluate function specs at explicit points on the GPU (``Function.__call__`` of the reference,
``safe_learning/functions.py:63-82``), through ``sl_eval_points``.

A spec is evaluated by putting it into the slot of a throw-away engine model that matches its
role (value function, policy, dynamics); all arithmetic runs in the same device functions as
the grid sweeps.  The payloads (GP factors, tables, networks) stay on the evaluation context
between calls and are uploaded again only when their version tokens change."""

import numpy as np

from . import _hip
from ._model import ModelBuilder

_context = None


def _ctx():
    global _context
    if _context is None:
        _context = _hip.Context()
    return _context


def _dummy_grid(d):
    from .functions import GridWorld
    return GridWorld([[0., 1.]] * d, 2)


def _to_device(ctx, array):
    """float64 device tensor of a NumPy array or tensor ([n, k])."""
    import torch
    if isinstance(array, torch.Tensor):
        t = array.to(device=ctx.torch_device, dtype=torch.float64)
        return (t if t.dim() == 2 else t.reshape(len(t), -1)).contiguous()
    array = np.atleast_2d(np.asarray(array, dtype=np.float64))
    return torch.from_numpy(np.ascontiguousarray(array)).to(ctx.torch_device)


def _on_device(x):
    """Inputs given as device tensors keep the results on the device (no host round trip)."""
    import torch
    return isinstance(x, torch.Tensor)


_persistent_builder = None
_dummy_grids = {}


def _builder(d):
    """The evaluation context's ONE builder (it remembers which GP heads / tables / networks the
    context holds - version tokens, ``_model.py`` - so a spec that did not change is not packed and
    uploaded again by every call: ``get_safe_sample`` evaluates the same GP, value table and policy
    every iteration of the exploration loop) with a throw-away grid of ``d`` dimensions."""
    global _persistent_builder
    import copy
    ctx = _ctx()
    if _persistent_builder is None:
        _persistent_builder = ModelBuilder(ctx, None)
    if d not in _dummy_grids:
        _dummy_grids[d] = _dummy_grid(d)
    _persistent_builder.grid = copy.copy(_dummy_grids[d])     # (dynamics() edits its nindex)
    return ctx, _persistent_builder


def value(spec, points, lipschitz=None):
    """``V(points)`` -> ``[n, 1]``; with ``lipschitz`` also returns ``L_v(points)``."""
    import torch
    from .functions import ConstantFunction, LinearSystem
    keep = _on_device(points)
    ctx = _ctx()
    d_pts = _to_device(ctx, points)
    n, d = d_pts.shape
    ctx, builder = _builder(d)
    builder.upload(ConstantFunction(np.zeros(1)), LinearSystem((np.eye(d), np.zeros((d, 1)))),
                   spec, 0.0 if lipschitz is None else lipschitz, 0.0, 0.0)
    out = torch.empty((n, 1), dtype=torch.float64, device=ctx.torch_device)
    ctx.eval_points(_hip.EVAL_VALUE, n, d_pts, out)
    if lipschitz is None:
        return out if keep else out.cpu().numpy()
    cols = 1 if builder._desc.lipschitz.lv_kind in (_hip.LIP_CONST, _hip.LIP_NORM_LINEAR,
                                                    _hip.LIP_NORM_GRAD) else d
    lv = torch.empty((n, cols), dtype=torch.float64, device=ctx.torch_device)
    ctx.eval_points(_hip.EVAL_LV, n, d_pts, lv)
    return (out, lv) if keep else (out.cpu().numpy(), lv.cpu().numpy())


def policy(spec, points):
    """``policy(points)`` -> ``[n, m]``."""
    import torch
    from .functions import LinearSystem, QuadraticFunction
    keep = _on_device(points)
    ctx = _ctx()
    d_pts = _to_device(ctx, points)
    n, d = d_pts.shape
    ctx, builder = _builder(d)
    m = _policy_output_dim(spec)
    builder.upload(spec, LinearSystem((np.eye(d), np.zeros((d, m)))), QuadraticFunction(np.eye(d)))
    out = torch.empty((n, m), dtype=torch.float64, device=ctx.torch_device)
    ctx.eval_points(_hip.EVAL_POLICY, n, d_pts, out)
    return out if keep else out.cpu().numpy()


def _policy_output_dim(spec):
    inner = getattr(spec, 'fun', spec) if type(spec).__name__ == 'Saturation' else spec
    return int(inner.output_dim)


def dynamics(spec, states, actions):
    """``dynamics(states, actions)`` -> next states, or ``(mean, error)`` for uncertain specs."""
    import torch
    from .functions import QuadraticFunction, UncertainFunction
    keep = _on_device(states)
    ctx = _ctx()
    d_states = _to_device(ctx, states)
    d_actions = _to_device(ctx, actions)
    n, d = d_states.shape
    ctx, builder = _builder(d)
    builder.grid.nindex = n                      # the action table is indexed by the point index
    builder.upload(d_actions, spec, QuadraticFunction(np.eye(d)))
    out = torch.empty((n, 2 + 2 * d), dtype=torch.float64, device=ctx.torch_device)
    ctx.eval_points(_hip.EVAL_DYNAMICS, n, d_states, out)
    rec = out if keep else out.cpu().numpy()
    mean, err = rec[:, 2:2 + d], rec[:, 2 + d:]
    return (mean, err) if isinstance(spec, UncertainFunction) else mean


def linear_map(spec, points):
    """``[x] M^T`` for a LinearSystem called with already-concatenated inputs."""
    from .functions import ConstantFunction, LinearSystem
    points = np.atleast_2d(np.asarray(points, dtype=np.float64))
    n, k = points.shape
    rows = spec.matrix.shape[0]
    if rows > k:
        raise NotImplementedError('LinearSystem evaluation needs output_dim <= input_dim')
    padded = np.zeros((k, k + 1))
    padded[:rows, :k] = spec.matrix
    out = dynamics(LinearSystem((padded,)), points, np.zeros((n, 1)))
    return out[:, :rows]
