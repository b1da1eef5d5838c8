"""Diagnostic: occupancy the runtime reports for the z-column kernels of a PDE_AMD_DIAG_ONLY build (PDE_AMD_LIB)."""
import ctypes, os, torch
torch.zeros(1, device="cuda")
L = ctypes.CDLL(os.environ["PDE_AMD_LIB"])
L.pde_amd_diag_zcol_occupancy()
