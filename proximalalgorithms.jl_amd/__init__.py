"""proximalalgorithms.jl_amd -- MI355X-native proximal-gradient engine behind the ProximalAlgorithms.jl
iterator API (ForwardBackward / FastForwardBackward on LeastSquares + NormL1 / IndBox).

Host-side mirror of the reference's interface (same names, keyword arguments and return conventions);
all arithmetic runs in hand-written HIP kernels of libproxgrad_hip.so (csrc/, C ABI in
include/proxgrad_hip.h).  There is no CPU fallback: without the library or a gfx950 device, calls raise.
"""
from ._lib import (PG_ERR_ALLOC, PG_ERR_COLLECTIVE, PG_ERR_HIP, PG_ERR_INVALID, PG_ERR_TIMEOUT, PG_ERR_UNSUPPORTED,
                   PG_FLAG_GAMMA_TOO_SMALL, PG_FLAG_SWEEP_FALLBACK, ProxGradError)
from .algorithm import IterativeAlgorithm
from .device import Context, Graph, HIPMatrix, HIPVector, as_hipvector, get_context, set_default_context
from .douglas_rachford import DouglasRachford, DouglasRachfordIteration, DouglasRachfordState
from .fast_forward_backward import (FastForwardBackward, FastForwardBackwardIteration, FastForwardBackwardState,
                                    FastProximalGradient, FastProximalGradientIteration)
from .fb_tools import backtrack_stepsize_, f_model, lower_bound_smoothness_constant
from .forward_backward import (ForwardBackward, ForwardBackwardIteration, ForwardBackwardState, ProximalGradient,
                               ProximalGradientIteration)
from .lbfgs import LBFGS, LBFGSOperator
from .nesterov import (AdaptiveNesterovSequence, ConstantNesterovSequence, FixedNesterovSequence,
                       NesterovExtrapolation, SimpleNesterovSequence, next_)
from .operators import (Composed, Conjugate, IndAffine, IndBox, IndNonnegative, IndPoint, IndZero, LeastSquares, Linear,
                        LogisticLoss, NormL1, Quadratic, SlicedSeparableSum,
                        SeparableQuadratic, SqrNormL2, SquaredDistance, Zero, convex_conjugate, gradient_, is_convex,
                        is_generalized_quadratic, prox, prox_, value_and_gradient, value_and_gradient_)
from .panoc import PANOC, NoAcceleration, PANOCIteration, PANOCState
from .panocplus import PANOCplus, PANOCplusIteration
from .sharding import (NativeRcclComm, TorchDistributedComm, allreduce_sum_, attach_row_team, native_rccl_available,
                       row_team_geometry, row_team_in_process, row_team_stats, row_team_tune, shard_cols, shard_rows)

from .zerofpr import ZeroFPR, ZeroFPRIteration
from .sfista import SFISTA, SFISTAIteration
from .anderson import AndersonAcceleration, AndersonAccelerationOperator
from .broyden import Broyden, BroydenOperator
from .davis_yin import DavisYin, DavisYinIteration
from .li_lin import LiLin, LiLinIteration
from .drls import DRLS, DRLSIteration
from .primal_dual import (AFBA, AFBAIteration, ChambollePock, ChambollePockIteration, VuCondat, VuCondatIteration,
                          AFBA_default_stepsizes)

__all__ = [
    "IndAffine", "IndNonnegative", "IndPoint", "Linear", "SlicedSeparableSum",
    "AndersonAcceleration", "AndersonAccelerationOperator", "Broyden", "BroydenOperator",
    "SFISTA", "SFISTAIteration", "DavisYin", "DavisYinIteration", "LiLin", "LiLinIteration", "DRLS", "DRLSIteration",
    "AFBA", "AFBAIteration", "VuCondat", "VuCondatIteration", "ChambollePock", "ChambollePockIteration",
    "AFBA_default_stepsizes", "NesterovExtrapolation", "Conjugate", "IndZero", "Quadratic", "SqrNormL2",
    "convex_conjugate", "is_convex", "is_generalized_quadratic",
    "ZeroFPR", "ZeroFPRIteration", "PANOCplus", "PANOCplusIteration",
    "ProxGradError", "PG_ERR_ALLOC", "PG_ERR_COLLECTIVE", "PG_ERR_HIP", "PG_ERR_INVALID", "PG_ERR_TIMEOUT", "PG_ERR_UNSUPPORTED",
    "PG_FLAG_GAMMA_TOO_SMALL", "PG_FLAG_SWEEP_FALLBACK", "IterativeAlgorithm", "DouglasRachford", "DouglasRachfordIteration", "DouglasRachfordState",
    "SeparableQuadratic", "PANOC", "PANOCIteration", "PANOCState", "NoAcceleration", "Composed", "LogisticLoss",
    "SquaredDistance", "Context", "HIPMatrix", "HIPVector", "as_hipvector", "get_context",
    "FastForwardBackward", "FastForwardBackwardIteration", "FastForwardBackwardState", "FastProximalGradient",
    "FastProximalGradientIteration", "backtrack_stepsize_", "f_model", "lower_bound_smoothness_constant",
    "ForwardBackward", "ForwardBackwardIteration", "ForwardBackwardState", "ProximalGradient",
    "ProximalGradientIteration", "LBFGS", "LBFGSOperator", "AdaptiveNesterovSequence", "ConstantNesterovSequence",
    "FixedNesterovSequence", "SimpleNesterovSequence", "next_", "IndBox", "LeastSquares", "NormL1", "Zero",
    "gradient_", "prox", "prox_", "value_and_gradient", "NativeRcclComm", "native_rccl_available", "TorchDistributedComm", "allreduce_sum_",
    "shard_rows", "shard_cols", "attach_row_team", "row_team_in_process", "row_team_stats", "row_team_tune", "row_team_geometry",
]
