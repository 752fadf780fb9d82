"""Run-time configuration (reference: ``safe_learning/configuration.py:8-32``)."""

import numpy as np


class Configuration(object):
    """float64 arithmetic and the reference's 10 000-cell verification batch.

    ``gp_batch_size`` no longer sizes any launch - the HIP kernels sweep the whole grid - but it
    still defines the batch boundaries that the reference's ``update_safe_set`` semantics depend
    on (``lyapunov.py:517-519, 585, 590``), so it is kept and honoured."""

    def __init__(self):
        self.np_dtype = np.float64
        self.gp_batch_size = 10000

    @property
    def dtype(self):
        return self.np_dtype

    def __repr__(self):
        return 'Configuration parameters:\n\ndtype: float64\ngp_batch_size: %d' % self.gp_batch_size


config = Configuration()
