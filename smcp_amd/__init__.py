"""smcp_amd -- MI355X-native chordal cone-program Newton-KKT kernels behind SMCP's API.

Only the hot path of cvxopt/smcp is implemented natively (HIP, gfx950): the chordal
multifrontal kernels (cholesky, completion, projected_inverse, hessian, llt, trsm, dot) and
the KKT/Schur-complement layer around them.  The interior-point drivers stay in Python.
"""
from .symbolic import Symbolic, symbolic, maxcardsearch, mindegree  # noqa: F401
from .cspmatrix import cspmatrix  # noqa: F401
from . import base, solvers  # noqa: F401
from .base import SDP, band_SDP, mtxnorm_SDP, completion  # noqa: F401  (smcp.__init__: same four names)
from .chordal import cholesky, projected_inverse, hessian, llt, trsm, dot, logdiagsum  # noqa: F401
# the CHOMPACK-level in-place completion (factor of the inverse) is smcp_amd.chordal.completion

__version__ = "0.1.0"
