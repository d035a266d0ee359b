"""MI355X-native implementation of the JoXSZ per-walker log-posterior hot path.

Host side: plain Python/numpy.  Device side: hand-written HIP kernels for gfx950
plus rocFFT, reached through the C-ABI declared in ``include/joxsz_hip.h``
(``joxsz_amd/csrc/libjoxsz_hip.so``, bound with ctypes).  There is no CPU
fallback: constructing ``JoxszPosterior`` without the built library raises.
"""
from .problem import Problem, default_par_table, PAR_SLOTS  # noqa: F401
from . import datasets, setup_host, chain, profiles  # noqa: F401

__all__ = ['Problem', 'default_par_table', 'PAR_SLOTS', 'datasets', 'setup_host', 'chain', 'profiles']
