/* pde_amd - C ABI of the MI355X (gfx950) PDE-residual loss and evaluation-metric kernels.
 *
 * The reference (reca1c1trant/PDE) is pure PyTorch and has no FFI of its own; each entry point below
 * names the reference Python interface whose device work it replaces (paths relative to the reference
 * tree).  All pointers are device pointers unless stated otherwise; all tensors are contiguous fp32.
 * Every call is asynchronous on `stream` (a hipStream_t passed as void*), never allocates, never
 * synchronises, and returns 0 or a negative pde_amd_status.
 */
#ifndef PDE_AMD_H
#define PDE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDE_AMD_MAX_FIELDS 4
#define PDE_AMD_MAX_EQS 4
#define PDE_AMD_MAX_AUX 1
#define PDE_AMD_MAX_SPARAMS 3
#define PDE_AMD_MAX_CONSTS 16

typedef enum pde_amd_status {
  PDE_AMD_OK = 0,
  PDE_AMD_EINVAL = -1,      /* null pointer / unknown pde id / bad count */
  PDE_AMD_ESHAPE = -2,      /* extents the kernels cannot handle (e.g. periodic axis shorter than a halo) */
  PDE_AMD_EWORKSPACE = -3,  /* workspace too small */
  PDE_AMD_EHIP = -4         /* a HIP launch failed; hipGetLastError has the cause */
} pde_amd_status;

/* One residual-loss problem: which PDE, the field extents [B, T, n0, n1, n2] (unused leading spatial
 * axes = 1), the residual lattice lo[a] <= index < n[a] - hi[a] (crops such as skip_bl / skip_boundary,
 * finetune/pde_loss_verified.py:939-940, :2056-2057, :1560-1561), and the per-instance constants in the
 * order pde_amd_pde_info reports them. */
typedef struct pde_amd_problem {
  int32_t pde_id;
  int32_t batch, frames;
  int32_t n[3];
  int32_t lo[3], hi[3];
  int32_t n_consts;
  float consts[PDE_AMD_MAX_CONSTS];
} pde_amd_problem;

typedef struct pde_amd_pde_info {
  int32_t id, ndim, n_fields, n_eqs, n_aux, n_consts, n_sparams;
  int32_t t_lo, t_hi;       /* residual frames are t_lo <= t < frames - t_hi */
  int32_t halo[3];          /* largest spatial reach of the residual stencil per axis */
  int32_t can_recompute;    /* 1 when pde_amd_residual_backward accepts resid_in == NULL */
  const char* name;
} pde_amd_pde_info;

int pde_amd_num_pdes(void);
int pde_amd_pde_id(const char* name);                     /* < 0 when unknown */
int pde_amd_get_pde_info(int pde_id, pde_amd_pde_info* out);
const char* pde_amd_status_string(int status);

/* Bytes of scratch the two residual entry points need for this problem (partial sums). */
size_t pde_amd_residual_workspace_bytes(const pde_amd_problem* prob);

/* Forward of every `*PDELoss.forward` in finetune/pde_loss_verified.py (e.g. TaylorGreen2DPDELoss.forward
 * :1823-1923, Burgers1DPDELoss.forward :2984-3027) and of burgers_pde_loss, finetune/pde_loss.py:87-216:
 *   sums[b][k] = sum over residual frames t and lattice sites of  weights[k][t] * R_k(b, t, site)^2
 * in fp64.  weights is [n_eqs][T_res]; with weights[k][t] = w_k / (s_k^2 N) the column sums of `sums`
 * are the reference's per-equation losses (per-timestep scales: w_k / (s_t[t]^2 N)).
 * totals (nullable): fp32 [n_eqs + 1] = the column sums of `sums` and, last, their left-to-right fp32
 * sum, i.e. the reference's `losses[k]` and `total = sum(losses.values())` (:1922).
 * resid_out (host array of n_eqs device pointers, or NULL): when given, R_k is also written on the full
 * grid as [B, T_res, n0, n1, n2] with zeros outside the lattice. */
int pde_amd_residual_forward(const pde_amd_problem* prob,
                             const float* const* fields, const float* const* aux,
                             const float* const* sparams, const float* weights,
                             double* sums, float* totals, float* const* resid_out,
                             void* workspace, size_t workspace_bytes, void* stream);

/* Backward of the same functions (autograd replay in the reference).  With
 *   u[b][k] = upstream[k] + upstream[n_eqs] + upstream_sums[b][k]
 * (upstream: host array of n_eqs + 1 device scalars, the gradients w.r.t. totals[k] and totals[n_eqs], NULL
 * entries count as 0; upstream_sums: device [B][n_eqs] fp64 gradient w.r.t. `sums`, nullable) it writes
 *   grads[f] = d( sum_{b,k,t,site} u[b][k] * weights[k][t] * R_k^2 ) / d fields[f]      on every frame.
 * The upstream values are read on the device, so no host synchronisation is needed to chain it behind
 * a loss such as lambda_pde * total (finetune/train_taylor_green_2d_lora_v3.py:506-512).
 * resid_in: the residuals a forward call stored (stored mode), or NULL to recompute them inside the
 * kernel (only when pde_amd_pde_info.can_recompute).  Every element of every grads[f] is written once. */
int pde_amd_residual_backward(const pde_amd_problem* prob,
                              const float* const* fields, const float* const* aux,
                              const float* const* sparams, const float* weights,
                              const float* const* upstream, const double* upstream_sums,
                              const float* const* resid_in, float* const* grads,
                              void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PDE_AMD_H */
