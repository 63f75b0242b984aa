"""smcp_amd -- MI355X-native chordal cone-program Newton-KKT kernels behind SMCP's API.

Only the hot path of cvxopt/smcp is implemented natively (HIP, gfx950): the chordal
multifrontal kernels (cholesky, completion, projected_inverse, hessian, llt, trsm, dot) and
the KKT/Schur-complement layer around them.  The interior-point drivers stay in Python.
"""
from .symbolic import Symbolic, symbolic, maxcardsearch, mindegree  # noqa: F401
from .cspmatrix import cspmatrix  # noqa: F401
from . import base, solvers  # noqa: F401
from .base import SDP, band_SDP  # noqa: F401
from .chordal import (cholesky, completion, projected_inverse, hessian, llt, trsm, dot,  # noqa: F401
                      logdiagsum)

__version__ = "0.1.0"
