"""pde_amd - MI355X (gfx950) kernels for physics-informed PDE-residual losses and evaluation metrics.

Drop-in classes and functions carrying the signatures of reca1c1trant/PDE's loss library; the
arithmetic runs in hand-written HIP kernels behind a C ABI (include/pde_amd.h).
"""
from .losses import (  # noqa: F401
    StencilPDELoss, Burgers1DPDELoss, Advection1DPDELoss, Heat1DPDELoss, APEBenchBurgers1DPDELoss, KS1DPDELoss,
    KdV1DPDELoss, TaylorGreen2DPDELoss, AdvDiff2DPDELoss, AnisoDiff2DPDELoss, Wave2DPDELoss,
    APEBenchBurgers2DPDELoss, NSPwCPDELoss, KP2DPDELoss, ShearFlowPDELossNPINN, Burgers2DCHPDELoss,
    ActiveMatterNPINNPDELoss, RayleighBenardFullPDELoss, TurbulentRadiativePDELoss, WaveGaussPDELoss,
    GrayScottPDELoss, Advection3DPDELoss, APEBenchBurgers3DPDELoss, SwiftHohenberg3DPDELoss, Heat3DPDELoss,
    burgers_pde_loss, burgers_pde_loss_upwind,
)

__version__ = "0.1.0"
