"""smooth_feedback_amd: MI355X-native batched QP / MPC / EKF engine behind the smooth::feedback API.

The compute path is libsfb.so (hand-written HIP for gfx950, C-ABI in include/sfb.h).  This package
is the thin host-side mirror of the reference interface used by tests and bench.py.
"""
from . import _capi  # noqa: F401  (fails loudly if libsfb.so is missing)
from ._capi import debug_set, debug_set_from  # noqa: F401  (debug knobs of the library: tests, A/B measurements)
from .qp import (QPBatchSolution, QPSolution, QPSolutionStatus, QPSolver, QPSolverParams,  # noqa: F401
                 QuadraticProgram, pack_colmajor, random_qp_batch, solve_qp, solve_qp_batch_device, solve_qp_batch_device_ws, Workspace,
                 solve_qp_batch_host, QuadraticProgramSparse, SparseQPPlan, solve_qp_sparse)

from .ekf import (ekf_predict_batch_device, ekf_predict_batch_host, ekf_predict_stepper_batch_device,  # noqa: F401
                  ekf_predict_update_batch_device, ekf_step_batch_host, ekf_update_batch_device)
from .mpc import LIE_RN, LIE_SE2, LIE_SO3, MPCLayout, MPCSwarm  # noqa: F401

__version__ = "0.1.0"
