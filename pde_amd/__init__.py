"""pde_amd - MI355X (gfx950) kernels for physics-informed PDE-residual losses and evaluation metrics.

Drop-in classes and functions carrying the signatures of reca1c1trant/PDE's loss library; the
arithmetic runs in hand-written HIP kernels behind a C ABI (include/pde_amd.h).
"""
from .losses import (  # noqa: F401
    StencilPDELoss, CapturedLossStep, Burgers1DPDELoss, Advection1DPDELoss, Heat1DPDELoss, APEBenchBurgers1DPDELoss, KS1DPDELoss,
    KdV1DPDELoss, TaylorGreen2DPDELoss, AdvDiff2DPDELoss, AnisoDiff2DPDELoss, Wave2DPDELoss,
    APEBenchBurgers2DPDELoss, NSPwCPDELoss, KP2DPDELoss, ShearFlowPDELossNPINN, ShearFlowPDELoss, Burgers2DCHPDELoss,
    ActiveMatterNPINNPDELoss, RayleighBenardFullPDELoss, TurbulentRadiativePDELoss, WaveGaussPDELoss,
    GrayScottPDELoss, Advection3DPDELoss, APEBenchBurgers3DPDELoss, SwiftHohenberg3DPDELoss, Heat3DPDELoss,
    TaylorGreenPDELoss, TaylorGreen3DPDELoss, NSIncomPDELoss, ActiveMatterPDELoss, ViscoelasticContinuityLoss,
    burgers_pde_loss, burgers_pde_loss_upwind, flow_mixing_pde_loss_v2, compute_flow_coefficients, pad_with_boundaries_1x,
)

from .boundary import (  # noqa: F401
    compute_boundary_loss, boundary_loss_1d, boundary_loss_2d_strips, boundary_loss_2d_edges, boundary_loss_3d_faces,
    boundary_loss_3d_shells, boundary_loss_3d_xy_strips, compute_noslip_loss, compute_patch_boundary_loss, PeriodicBCLoss,
)
from .metrics import (  # noqa: F401
    compute_vrmse, nrmse_per_sample, nrmse_per_channel, compute_rse, compute_mean_rse, metric_func, NRMSEAccumulator,
    compute_pdebench_nrmse, LpLoss, FftLpLoss, validation_scalars,
)
from .glue import (  # noqa: F401
    gather_fields, create_pde_loss, compute_pde_loss_generic, PDE_LOSS_REGISTRY, compute_pde_loss_taylor_green_2d,
    compute_pde_loss_rayleigh_benard, compute_pde_loss_shear_flow, compute_pde_loss_burgers_1d, compute_pde_loss_gray_scott,
    compute_pde_loss_burgers_3d, compute_pde_loss_heat_3d,
)
from .sink import shared_grad, Denormalised  # noqa: F401
from .scales import ResidualScaleAccumulator  # noqa: F401
from .rollout import evaluate_rollout, summarize_rollout, denormalised  # noqa: F401

__version__ = "0.1.0"
