"""linearoperators.jl_amd — MI355X-native `mul!` hot path behind the LinearOperators.jl operator API.

The directory name contains a dot, so it is loaded through ``__graft_entry__.load_package()``
(which registers it in ``sys.modules`` as ``linearoperators_jl_amd``)::

    from __graft_entry__ import load_package
    lo = load_package()
    D = lo.opDiagonal(d); lo.mul(res, D, v, 2.0, 3.0)

Everything that touches a vector runs in ``csrc/libmxlo.so`` (hand-written HIP for gfx950) through
the C ABI of ``include/mxlo.h``; this package is host control flow only and raises if the library
or a GPU is missing — there is no CPU fallback.
"""
from . import _lib
from ._lib import MxloError, build
from .device import Context, Storage, Timer, get_ctx, storage_of
from .operators import (AbstractLinearOperator, AdjointLinearOperator, ConjugateLinearOperator, LinearOperator,
                        LinearOperatorException, TransposeLinearOperator, add, adjoint, allocate_vectors_args3, apply,
                        compose, conj, eltype, has_args5, hcat, hvcat, isallocated5, ishermitian, issymmetric, mul,
                        nctprod, neg, nprod, ntprod, one, reset, scale_op, size, state_version, storage_type, to_dense, touched,
                        transpose,
                        vcat, zero)
from .leaves import (BlockDiagonalOperator, LinearOperatorFromMatrix, LinearOperatorFromSparse, ShiftedOperator, sparse_csc, jrange, kron, opDiagonal, opExtension, opEye,
                     opHermitian, opHouseholder, opOnes, opRestriction, opZeros)

Matrix = to_dense

try:  # quasi-Newton operators
    from .qn import (InverseLBFGSOperator, LBFGSOperator, LSR1Operator, diag, ldiv, push, solve_shifted_system)
except ImportError:  # pragma: no cover - during bring-up only
    pass
from .diagqn import DiagonalAndrei, DiagonalBFGS, DiagonalPSB, SpectralGradient
from .graph import CapturedSequence, capture_mul
from .utilities import check_ctranspose, check_hermitian, check_positive_definite, normest

try:
    from . import sharded
except ImportError:  # pragma: no cover
    pass
