"""The kernels a ``GPRCached`` model can be built from, under the names of gpflow 0.4.0's
``gpflow.kernels`` (``examples/inverted_pendulum.ipynb:152-158`` of the reference builds
``kernels.Linear(...) + kernels.Matern32(...) * kernels.Linear(...)``)."""

from .functions import Add, Kern, Linear, Matern32, Prod, RBF       # noqa: F401

__all__ = ['Kern', 'RBF', 'Matern32', 'Linear', 'Add', 'Prod']
