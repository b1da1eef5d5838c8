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
#define PDE_AMD_MAX_AUX 2
#define PDE_AMD_MAX_SPARAMS 3
#define PDE_AMD_MAX_CONSTS 16

typedef enum pde_amd_status {
  PDE_AMD_OK = 0,
  PDE_AMD_EINVAL = -1,      /* null pointer / unknown pde id / bad count */
  PDE_AMD_ESHAPE = -2,      /* extents the kernels cannot handle (e.g. periodic axis shorter than a halo) */
  PDE_AMD_EWORKSPACE = -3,  /* workspace too small */
  PDE_AMD_EHIP = -4,        /* a HIP call failed; hipGetLastError has the cause */
  PDE_AMD_ESCRATCH = -5,    /* pde_amd_residual_forward_backward_static: the problem needs more than the static scratch */
  PDE_AMD_EEVENT = -6       /* an event armed with pde_amd_time_next_adjoint could not be recorded (e.g. under stream capture);
                               the kernels of the call were launched all the same */
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

/* One training step of the same functions in one call: the loss (sums / totals as in pde_amd_residual_forward) and
 * the gradient of `upstream_total * totals[n_eqs]` w.r.t. every field (as pde_amd_residual_backward with only
 * upstream[n_eqs] set), i.e. what `total, losses = loss_fn(...); (c * total).backward()` needs
 * (finetune/train_taylor_green_2d_lora_v3.py:506-530).  Where the vector adjoint applies it is a single pass over
 * the fields: that kernel recomputes every residual anyway and adds up weight * R^2 on the way; elsewhere the
 * forward and the backward kernels run one after the other.  upstream_total is a device scalar, read on the device.
 * Needs pde_amd_pde_info.can_recompute. */
int pde_amd_residual_forward_backward(const pde_amd_problem* prob,
                                      const float* const* fields, const float* const* aux,
                                      const float* const* sparams, const float* weights,
                                      const float* upstream_total, double* sums, float* totals,
                                      float* const* grads, void* workspace, size_t workspace_bytes, void* stream);

/* pde_amd_residual_forward_backward for callers that cannot allocate device memory or hand over a workspace (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same
 * arguments, results and kernels, but the partial sums live in a device array inside the library of
 * PDE_AMD_STATIC_SCRATCH_BYTES bytes, which every GPU the library runs on holds until the process ends; the call makes
 * no allocation of any kind.  Problems with pde_amd_residual_workspace_bytes(prob) above that size are rejected with
 * PDE_AMD_ESCRATCH (every configuration of the reference's trainers fits: the largest, shear flow at B = 64, needs
 * 3.5 MiB).  All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads) race;
 * calls on one stream, or otherwise ordered, do not.  The array used is the one of the GPU that owns grads[0] (the
 * current device where the runtime cannot tell), whichever device is current.  The first call on a device looks the
 * array's address up (hipGetSymbolAddress), so make it outside stream capture. */
#define PDE_AMD_STATIC_SCRATCH_BYTES 4194304
int pde_amd_residual_forward_backward_static(const pde_amd_problem* prob,
                                             const float* const* fields, const float* const* aux,
                                             const float* const* sparams, const float* weights,
                                             const float* upstream_total, double* sums, float* totals,
                                             float* const* grads, void* stream);

/* pde_amd_residual_forward in double precision, for the offline tools that measure the residual scales of ground-truth
 * trajectories in float64 (tools/compute_gt_scales_per_t.py:66-175, tools/test_gt_pde_*.py): near an exact solution the
 * residual is a small difference of large terms and fp32 leaves a floor of about 1e-7 of the largest one.  Same PDEs, same
 * formula (the fp32-rounded numeric literals included), fp64 arithmetic on fp64 tensors: fields, aux, sparams, weights
 * [n_eqs][T_res], resid_out are double; consts: host array of prob->n_consts doubles used instead of prob->consts (which
 * are floats); totals (nullable): double [n_eqs + 1].  No backward: the tools only evaluate. */
int pde_amd_residual_forward_f64(const pde_amd_problem* prob, const double* consts, const double* const* fields,
                                 const double* const* aux, const double* const* sparams, const double* weights, double* sums,
                                 double* totals, double* const* resid_out, void* workspace, size_t workspace_bytes, void* stream);

/* Measurement hook, no counterpart in the reference: the next launch of an adjoint kernel issued from this host thread
 * (by pde_amd_residual_backward in either mode, or by pde_amd_residual_forward_backward) records the two HIP events
 * (hipEvent_t, created with timing enabled) on its stream immediately before and after that one launch; the small
 * reduction launches of the same call stay outside.  One-shot: the events are forgotten afterwards.  NULL, NULL disarms it. */
int pde_amd_time_next_adjoint(void* start_event, void* stop_event);

/* The adjoint kernel the last backward or one-pass call of this host thread launched, spelled as a profiler prints it
 * (e.g. "bwd_march_q_kernel<pde_amd::Pde_tg2d, 4, true, true>"): the key under which a measurement of that kernel is
 * filed.  "" before the first such call.  The string lives in thread-local storage of the library. */
const char* pde_amd_last_adjoint_kernel(void);
/* Diagnostic, no counterpart in the reference: fills the LDS of the current device's CUs with NaNs (a short kernel on
 * `stream`).  LDS keeps its contents between kernels, so a kernel that reads LDS it did not write usually sees finite
 * leftovers; after this call it sees NaNs.  The test-suite runs with it before every launch when PDE_AMD_POISON_LDS=1. */
int pde_amd_diag_poison_lds(void* stream);

/* grads[f][i] *= *upstream / *assumed for n_fields arrays of `elements` floats; nothing is read or written when the
 * two device scalars are equal (decided on the device, no host synchronisation).  It turns the gradients that
 * pde_amd_residual_forward_backward produced for an assumed upstream value into those for the value autograd
 * delivers later (e.g. lambda_pde, or a loss scaler's factor).  No counterpart in the reference. */
int pde_amd_rescale_gradients(float* const* grads, int32_t n_fields, int64_t elements, const float* upstream,
                              const float* assumed, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Channel-last evaluation metrics.
 *
 * One pass over pred and target, both [groups][rows][channels] contiguous fp32 (channels innermost; "groups" are
 * whatever leading axes the caller wants separate results for: samples, samples x frames, or 1).  For every
 * (group, channel) it returns PDE_AMD_NSTAT fp64 numbers:
 *   [0] sum (p-g)^2   [1] sum g^2   [2] sum g   [3] sum p   [4] sum (p-g)^2 / (g^2 + rse_eps)   [5] max |p-g|
 *   ([4] is left 0 when rse_eps < 0: its division is the costliest part of the pass and only compute_mean_rse reads it)
 * from which the host forms, with the reference's own epsilons,
 *   compute_vrmse            finetune/train_taylor_green_2d_lora_v3.py:204-232 (per-channel mse, variance of g, overall rmse)
 *   evaluate() accumulators  eval/eval_nrmse.py:125-155 (per channel and per frame sums of (p-g)^2 and g^2)
 *   nrmse_per_sample/_channel tools/eval_rollout.py:60-90
 *   compute_mean_rse         eval/eval_autoregressive.py:197-219
 *   metric_func              eval/metrics.py:189-226 (RMSE, nRMSE, MPP-nRMSE, conserved-sum error, max error; there the
 *                            caller passes channels = nt * nc of the PDEBench layout [nb, nx.., nt, nc]).
 * channels <= 1024, groups <= 65535. */
#define PDE_AMD_NSTAT 6
size_t pde_amd_pair_stats_workspace_bytes(int64_t groups, int64_t rows, int32_t channels);
int pde_amd_pair_stats(const float* pred, const float* target, int64_t groups, int64_t rows, int32_t channels,
                       float rse_eps, double* stats, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Index-list "term" losses on a channel-last tensor [outer][d0][d1][d2][C] (unused leading spatial axes = 1).
 *
 * A term selects the Cartesian product of four index lists (axes d0, d1, d2 and the channel axis; the lists
 * live in one device int32 array `indices`, list k of a term starting at idx_off[k] with n[k] entries) and
 * evaluates at each selected element   v = sum_j coef[j] * x[index + shift[j] along `axis`],
 * x = pred - target (use_target = 1) or pred (use_target = 0).  sums[t] = sum of v^2 over the term, fp64.
 * The slices the reference concatenates are one term each, so elements shared by two slices count twice,
 * as they do there:
 *   compute_boundary_loss   1-D finetune/train_burgers_1d_lora_v3.py:111-139, 2-D strips
 *                           finetune/train_taylor_green_2d_lora_v3.py:132-166, 2-D edges
 *                           finetune/train_shear_flow_lora_v3.py:111-146, 3-D faces finetune/train_heat_3d_lora_v3.py:109-141
 *                           (its z faces select rows of the y axis by the channel list), 3-D shells
 *                           finetune/train_advection_3d_lora_v3.py:110-159
 *   compute_noslip_loss     finetune/train_ns_incom_lora_v3.py:150-176
 *   compute_patch_boundary_loss finetune/train_rayleigh_benard_lora_v3.py:114-135 (taps 1, -2, 1)
 * Backward: grad += scales[t] * d(sums[t]) / d pred for every term; `grad` has the shape of pred and must
 * hold the values to accumulate into (zeros for a fresh gradient).  scales is a device array [n_terms]. */
#define PDE_AMD_MAX_TERMS 12
typedef struct pde_amd_term {
  int32_t n[4];
  int32_t idx_off[4];
  int32_t axis, n_taps;
  int32_t shift[3];
  float coef[3];
  int32_t use_target;
} pde_amd_term;
size_t pde_amd_terms_workspace_bytes(int32_t n_terms);
int pde_amd_terms_forward(const float* pred, const float* target, int64_t outer, const int32_t* dims /* host [4]: d0 d1 d2 C */,
                          const pde_amd_term* terms /* host */, int32_t n_terms, const int32_t* indices, double* sums,
                          void* workspace, size_t workspace_bytes, void* stream);
int pde_amd_terms_backward(const float* pred, const float* target, int64_t outer, const int32_t* dims,
                           const pde_amd_term* terms, int32_t n_terms, const int32_t* indices, const float* scales,
                           float* grad, void* stream);

/* compute_vrmse's epilogue (finetune/train_taylor_green_2d_lora_v3.py:215-232) on the statistics of ONE group as
 * pde_amd_pair_stats leaves them, stats[channels][PDE_AMD_NSTAT]: over the n_valid <= 64 listed channels (host array)
 *   out[0] = mean_c sqrt( (sum d^2 / rows) / (sum g^2 / rows - (sum g / rows)^2 + 1e-8) )        (vrmse)
 *   out[1] = sqrt( sum_c sum d^2 / (rows * n_valid) + 1e-8 )                                       (rmse)
 * as fp32 device scalars; NaN for an empty list, like the mean of an empty tensor. */
int pde_amd_vrmse_from_stats(const double* stats, int32_t channels, const int32_t* valid, int32_t n_valid, int64_t rows,
                             float* out, void* stream);

/* The radial binning of metric_func's Fourier-band error (eval/metrics.py:258-296: the Python loops over wave numbers
 * that add |F|^2 into bin floor(|k|)).  spectrum: complex64 [groups][points][nt] (re, im interleaved), device;
 * the bins as a CSR list on the device: members of bin i are point_index[bin_start[i] .. bin_start[i+1]), each a
 * point number < points.  out[groups][nbins][nt] (fp32, device) = sum over the bin's members of |spectrum|^2, added
 * in list order with fp64 accumulation (repeatable; no atomics). */
int pde_amd_band_power(const float* spectrum, int64_t groups, int64_t points, int32_t nt, const int32_t* bin_start,
                       const int32_t* point_index, int32_t nbins, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * compute_pde_loss glue (finetune/train_taylor_green_2d_lora_v3.py:169-201, train_rayleigh_benard_lora_v3.py:178-216,
 * train_burgers_1d_lora_v3.py:142-172, train_heat_3d_lora_v3.py:144-172, train_fullft_physonly.py:172-223):
 *   fields[f][b][0][s]     = input [b][0][s][channels[f]]
 *   fields[f][b][1 + t][s] = output[b][t][s][channels[f]]
 * with output [B][T_out][S][C], input [B][T_in][S][C], S the product of the spatial extents.  transpose = 1
 * swaps the last two spatial axes (rows x cols -> cols x rows) in the fields, which is what the
 * Rayleigh-Benard trainer's u.transpose(-1, -2) feeds its loss.  The scatter is the gradient of the gather
 * w.r.t. output: every element of grad_output [B][T_out][S][C] is written, zeros outside the gathered channels
 * (B * T_out <= 65535, S * C < 2^31). */
int pde_amd_gather_fields(const float* output, const float* input, int64_t B, int64_t T_out, int64_t T_in, int64_t S, int32_t C,
                          const int32_t* channels /* host */, int32_t n_fields, int32_t rows, int32_t cols, int32_t transpose,
                          float* const* fields /* host array of device pointers */, void* stream);
int pde_amd_scatter_field_grads(const float* const* field_grads, int64_t B, int64_t T_out, int64_t S, int32_t C,
                                const int32_t* channels, int32_t n_fields, int32_t rows, int32_t cols, int32_t transpose,
                                float* grad_output, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * The same four operations on a model output that is still normalised: the trainers compute
 * `output = output_norm * std + mean` (finetune/train_taylor_green_2d_lora_v3.py:500-501, with mean and std one value
 * per sample and channel, pretrain/model_v3.py:427-431) and evaluate every loss on that product; its forward and
 * its autograd backward are two more passes each over the full 18-channel tensor.  Here `pred` / `output` is
 * output_norm and scale[B][C] = std, shift[B][C] = mean (fp32, device): a value is read as
 * fl(fl(x * scale) + shift), exactly what the product would hold in memory, and gradients come out w.r.t. output_norm
 * (multiplied by scale).  `input` of the gather is raw data and is not mapped.  outer_per_sample = outer / B
 * (the frames of a sample).  scale == shift == NULL gives the plain functions above. */
/* pde_amd_pair_stats with pred read as pred * scale + shift, scale and shift [groups][channels]: a group per sample. */
int pde_amd_pair_stats_affine(const float* pred, const float* target, int64_t groups, int64_t rows, int32_t channels,
                              float rse_eps, const float* scale, const float* shift, double* stats, void* workspace,
                              size_t workspace_bytes, void* stream);
/* The same with a by-product that lets ONE sweep over the model output serve a whole validation batch
 * (finetune/train_taylor_green_2d_lora_v3.py:252-275: boundary RMSE, residual loss, VRMSE and RMSE of one `output`):
 * the channels field_channels[f] of the de-normalised pred are also written as per-field tensors
 * fields_out[f] = [groups][frames + 1][rows / frames] at frames 1..frames, which is the gather of compute_pde_loss
 * (:183-195) minus its frame 0 (the input's first frame: the caller's to fill).  fields_out: host array of n_fields
 * device pointers (n_fields = 0: none); rows must be a multiple of frames, the channels distinct. */
int pde_amd_pair_stats_gather(const float* pred, const float* target, int64_t groups, int64_t rows, int32_t channels,
                              float rse_eps, const float* scale, const float* shift, double* stats, int64_t frames,
                              const int32_t* field_channels, int32_t n_fields, float* const* fields_out, void* workspace,
                              size_t workspace_bytes, void* stream);
int pde_amd_terms_forward_affine(const float* pred, const float* target, int64_t outer, const int32_t* dims,
                                 const pde_amd_term* terms, int32_t n_terms, const int32_t* indices, const float* scale,
                                 const float* shift, int64_t outer_per_sample, double* sums, void* workspace,
                                 size_t workspace_bytes, void* stream);
int pde_amd_terms_backward_affine(const float* pred, const float* target, int64_t outer, const int32_t* dims,
                                  const pde_amd_term* terms, int32_t n_terms, const int32_t* indices, const float* scale,
                                  const float* shift, int64_t outer_per_sample, int32_t disjoint /* 1: the caller guarantees
                                  that no two terms touch a common element of pred, and all terms run in one launch;
                                  0: one launch per term, in order */, const float* scales, float* grad, void* stream);
int pde_amd_gather_fields_affine(const float* output, const float* input, int64_t B, int64_t T_out, int64_t T_in, int64_t S, int32_t C,
                                 const int32_t* channels, int32_t n_fields, int32_t rows, int32_t cols, int32_t transpose,
                                 const float* scale, const float* shift, float* const* fields, void* stream);
int pde_amd_scatter_field_grads_affine(const float* const* field_grads, int64_t B, int64_t T_out, int64_t S, int32_t C,
                                       const int32_t* channels, int32_t n_fields, int32_t rows, int32_t cols, int32_t transpose,
                                       const float* scale, float* grad_output, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Spectral Laplacian of two real fields at once (the Gray-Scott loss's fft_laplacian,
 * finetune/pde_loss_verified.py:200-214, applied to A and to B at :226-227 of the same file):
 *   lap = ifft2(multiplier * fft2(f)).real over the last two axes, multiplier = -K2 of shape [nx][ny].
 * a, b, lap_a, lap_b: [planes][nx][ny] fp32, 16-byte aligned; a plane pair (a[p], b[p]) is transformed as one complex
 * plane inside one workgroup's LDS.  The multiplier must be real and even in both wavenumbers (K2 is), which also makes
 * the operator its own adjoint: the backward pass calls the same entry point on the two upstream gradients.
 * nx, ny in {16, 32, 64, 128}; anything else returns PDE_AMD_ESHAPE (pde_amd_spectral_pair_supported tells beforehand)
 * and the caller keeps its rocFFT path. */
int pde_amd_spectral_pair_supported(int32_t nx, int32_t ny);
int pde_amd_spectral_laplacian_pair(const float* a, const float* b, float* lap_a, float* lap_b, const float* multiplier,
                                    int64_t planes, int32_t nx, int32_t ny, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PDE_AMD_H */
