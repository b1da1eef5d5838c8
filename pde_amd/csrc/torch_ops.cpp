// Torch front end of the residual-loss entry points: one operator, pde_amd::residual_losses, whose autograd node
// lives in C++ and calls the C ABI of libpde_amd.so directly (no Python between autograd and the kernel launch).
// Host code only; the C ABI is bound at run time (pde_amd_torch_bind) so that the same operator library serves
// whichever kernel library the process loaded.  Mirrors pde_amd/functional.py::_ResidualLosses, which stays as the
// path for non-HIP backends of the test-suite.
#include <dlfcn.h>

#include <cstring>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include <ATen/ATen.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include "../../include/pde_amd.h"

namespace {

struct Abi {
  decltype(&pde_amd_residual_workspace_bytes) workspace_bytes = nullptr;
  decltype(&pde_amd_residual_forward) forward = nullptr;
  decltype(&pde_amd_residual_backward) backward = nullptr;
  decltype(&pde_amd_residual_forward_backward) forward_backward = nullptr;
  decltype(&pde_amd_rescale_gradients) rescale = nullptr;
  decltype(&pde_amd_status_string) status_string = nullptr;
  decltype(&pde_amd_get_pde_info) get_pde_info = nullptr;
} abi;

void check(int st, const char* what) {
  TORCH_CHECK(st == 0, "pde_amd: ", what, " failed: ", abi.status_string ? abi.status_string(st) : "?", " (", st, ")");
}

// scratch for the partial sums, one buffer per (device, stream): every use is ordered on the stream it was launched on
at::Tensor& workspace(const at::Device& dev, void* stream, size_t bytes) {
  static std::mutex mu;
  static std::unordered_map<uint64_t, at::Tensor> cache;
  const uint64_t key = (uint64_t)(uintptr_t)stream * 64u + (uint64_t)dev.index();
  std::lock_guard<std::mutex> lock(mu);
  at::Tensor& t = cache[key];
  if (!t.defined() || (size_t)t.numel() < bytes)
    t = at::empty({(int64_t)std::max(bytes, (size_t)1 << 20)}, at::TensorOptions().dtype(at::kByte).device(dev));
  return t;
}

struct Config {      // everything of a call that is not a tensor
  pde_amd_problem prob;
  int n_eqs, t_res, mode;      // mode: 0 recompute, 1 stored, 2 fused
  bool need_grad;
};

template <int N>
struct Ptrs {
  const float* p[N] = {};
  explicit Ptrs(at::TensorList ts) {
    TORCH_CHECK((int)ts.size() <= N, "pde_amd: too many tensors");
    for (size_t i = 0; i < ts.size(); ++i) p[i] = ts[i].defined() ? ts[i].data_ptr<float>() : nullptr;
  }
};

// fp32, contiguous, on `dev`: the tensor itself when it already is; otherwise a converted copy (recorded by autograd
// when the tensor needs gradients, since this runs before the node is built)
at::Tensor prepared(const at::Tensor& t, const at::Device& dev) {
  at::Tensor r = t;
  if (r.scalar_type() != at::kFloat || r.device() != dev) r = r.to(at::TensorOptions().dtype(at::kFloat).device(dev));
  return r.is_contiguous() ? r : r.contiguous();
}

using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

struct ResidualLosses : public torch::autograd::Function<ResidualLosses> {
  // outputs: totals[0..n_eqs] as 0-dim tensors, then sums[B, n_eqs]
  static variable_list forward(AutogradContext* ctx, at::TensorList fields, at::TensorList aux, at::TensorList sparams,
                               const at::Tensor& weights, const std::optional<at::Tensor>& assumed, int64_t cfg_handle,
                               int64_t totals_handle) {
    const Config& c = *reinterpret_cast<const Config*>(cfg_handle);
    const at::Tensor& f0 = fields[0];
    const c10::hip::HIPStream stream = c10::hip::getCurrentHIPStream(f0.device().index());
    c10::hip::HIPGuard guard(f0.device().index());
    const int ne = c.n_eqs, B = c.prob.batch;
    at::Tensor sums = at::empty({B, ne}, f0.options().dtype(at::kDouble));
    // the caller's vector when it gave one (not an input of the node: the returned scalars are plain views of it)
    at::Tensor totals = totals_handle ? *reinterpret_cast<const at::Tensor*>(totals_handle) : at::empty({ne + 1}, f0.options());
    const size_t ws_bytes = abi.workspace_bytes(&c.prob);
    at::Tensor& ws = workspace(f0.device(), stream.stream(), ws_bytes);
    Ptrs<PDE_AMD_MAX_FIELDS> pf(fields);
    Ptrs<PDE_AMD_MAX_AUX> pa(aux);
    Ptrs<PDE_AMD_MAX_SPARAMS> ps(sparams);
    variable_list resid, ghat;
    if (c.need_grad && c.mode == 2) {
      float* pg[PDE_AMD_MAX_FIELDS] = {};
      for (size_t i = 0; i < fields.size(); ++i) {
        ghat.push_back(at::empty_like(fields[i]));
        pg[i] = ghat.back().data_ptr<float>();
      }
      check(abi.forward_backward(&c.prob, pf.p, pa.p, ps.p, weights.data_ptr<float>(), assumed->data_ptr<float>(),
                                 sums.data_ptr<double>(), totals.data_ptr<float>(), pg, ws.data_ptr(), ws_bytes, stream.stream()),
            "residual_forward_backward");
    } else {
      float* pr[PDE_AMD_MAX_EQS] = {};
      const bool store = c.need_grad && c.mode == 1;
      if (store) {
        std::vector<int64_t> shape(f0.sizes().begin(), f0.sizes().end());
        shape[1] = std::max(c.t_res, 0);
        for (int k = 0; k < ne; ++k) {
          resid.push_back(at::empty(shape, f0.options()));
          pr[k] = resid.back().data_ptr<float>();
        }
      }
      check(abi.forward(&c.prob, pf.p, pa.p, ps.p, weights.data_ptr<float>(), sums.data_ptr<double>(), totals.data_ptr<float>(),
                        store ? pr : nullptr, ws.data_ptr(), ws_bytes, stream.stream()),
            "residual_forward");
    }
    if (c.need_grad) {
      variable_list saved(fields.begin(), fields.end());
      saved.insert(saved.end(), aux.begin(), aux.end());
      saved.insert(saved.end(), sparams.begin(), sparams.end());
      saved.push_back(weights);
      ctx->save_for_backward(saved);
      ctx->saved_data["cfg"] = c10::IValue(std::string(reinterpret_cast<const char*>(&c), sizeof(Config)));
      ctx->saved_data["counts"] = c10::IValue(std::vector<int64_t>{(int64_t)fields.size(), (int64_t)aux.size(), (int64_t)sparams.size()});
      if (!resid.empty()) ctx->saved_data["resid"] = c10::IValue(resid);      // not inputs or outputs: kept as plain data
      if (!ghat.empty()) {
        ctx->saved_data["ghat"] = c10::IValue(ghat);
        ctx->saved_data["assumed"] = c10::IValue(*assumed);
      }
    }
    ctx->set_materialize_grads(false);
    variable_list out;
    out.reserve(ne + 2);
    for (int k = 0; k <= ne; ++k) out.push_back(totals.select(0, k));
    out.push_back(sums);
    return out;
  }

  static variable_list backward(AutogradContext* ctx, variable_list gouts) {
    const std::string blob = ctx->saved_data["cfg"].toStringRef();
    Config c;
    std::memcpy(&c, blob.data(), sizeof(Config));
    const auto counts = ctx->saved_data["counts"].toIntVector();
    const int nf = (int)counts[0], na = (int)counts[1], ns = (int)counts[2], ne = c.n_eqs;
    const variable_list saved = ctx->get_saved_variables();
    variable_list result((size_t)(nf + na + ns + 4));      // undefined = no gradient
    bool any_tot = false, only_last = true;
    for (int k = 0; k <= ne; ++k) {
      if (gouts[k].defined()) {
        any_tot = true;
        if (k < ne) only_last = false;
      }
    }
    const at::Tensor& gsums = gouts[ne + 1];
    if (!any_tot && !gsums.defined()) return result;
    const at::Tensor& f0 = saved[0];
    const c10::hip::HIPStream stream = c10::hip::getCurrentHIPStream(f0.device().index());
    c10::hip::HIPGuard guard(f0.device().index());
    variable_list grads;
    auto it = ctx->saved_data.find("ghat");
    if (it != ctx->saved_data.end() && !gsums.defined() && only_last && gouts[ne].defined()) {
      grads = it->second.toTensorVector();      // computed in forward for the assumed upstream value: rescale if it differs
      ctx->saved_data.erase(it);                // handed out once; a second backward recomputes
      const at::Tensor up = gouts[ne].to(at::kFloat).contiguous();
      const at::Tensor assumed = ctx->saved_data["assumed"].toTensor();
      if (up.data_ptr() != assumed.data_ptr()) {      // handed the assumed scalar itself: nothing to compare on the device
        float* pg[PDE_AMD_MAX_FIELDS] = {};
        for (int i = 0; i < nf; ++i) pg[i] = grads[i].data_ptr<float>();
        check(abi.rescale(pg, nf, grads[0].numel(), up.data_ptr<float>(), assumed.data_ptr<float>(), stream.stream()), "rescale_gradients");
      }
    } else {
      std::vector<at::Tensor> ups((size_t)ne + 1);
      const float* pu[PDE_AMD_MAX_EQS + 1] = {};
      for (int k = 0; k <= ne; ++k) {
        if (gouts[k].defined()) {
          ups[k] = gouts[k].to(at::kFloat).contiguous();
          pu[k] = ups[k].data_ptr<float>();
        }
      }
      at::Tensor gs;
      if (gsums.defined()) gs = gsums.to(at::kDouble).contiguous();
      Ptrs<PDE_AMD_MAX_FIELDS> pf(at::TensorList(saved.data(), (size_t)nf));
      Ptrs<PDE_AMD_MAX_AUX> pa(at::TensorList(saved.data() + nf, (size_t)na));
      Ptrs<PDE_AMD_MAX_SPARAMS> ps(at::TensorList(saved.data() + nf + na, (size_t)ns));
      const at::Tensor& weights = saved[(size_t)(nf + na + ns)];
      const float* pr[PDE_AMD_MAX_EQS] = {};
      variable_list resid;
      auto ir = ctx->saved_data.find("resid");
      if (ir != ctx->saved_data.end()) {
        resid = ir->second.toTensorVector();
        for (int k = 0; k < ne; ++k) pr[k] = resid[k].data_ptr<float>();
      }
      float* pg[PDE_AMD_MAX_FIELDS] = {};
      for (int i = 0; i < nf; ++i) {
        grads.push_back(at::empty_like(saved[i]));
        pg[i] = grads.back().data_ptr<float>();
      }
      check(abi.backward(&c.prob, pf.p, pa.p, ps.p, weights.data_ptr<float>(), any_tot ? pu : nullptr,
                         gs.defined() ? gs.data_ptr<double>() : nullptr, resid.empty() ? nullptr : pr, pg, nullptr, 0, stream.stream()),
            "residual_backward");
    }
    for (int i = 0; i < nf; ++i)
      if (ctx->needs_input_grad(i)) result[i] = grads[i];
    return result;
  }
};

// fields: tensors of one shape on one HIP device (converted to contiguous fp32 when they are not); consts/lo/hi as in
// pde_amd_problem; mode 0 recompute, 1 stored, 2 fused (then `assumed` is the device scalar of the upstream value to assume);
// totals_out: where the n_eqs + 1 loss scalars are written instead of a fresh vector (the returned scalars are views of it).
std::vector<at::Tensor> residual_losses(at::TensorList fields, at::TensorList aux, at::TensorList sparams, const at::Tensor& weights,
                                        int64_t pde_id, c10::ArrayRef<double> consts, at::IntArrayRef lo, at::IntArrayRef hi,
                                        int64_t n_eqs, int64_t t_lo, int64_t t_hi, int64_t mode, const std::optional<at::Tensor>& assumed,
                                        const std::optional<at::Tensor>& totals_out) {
  TORCH_CHECK(abi.forward != nullptr, "pde_amd: the kernel library is not bound (pde_amd_torch_bind)");
  TORCH_CHECK(!fields.empty() && fields.size() <= PDE_AMD_MAX_FIELDS && fields[0].dim() >= 3 && fields[0].dim() <= 5, "pde_amd: bad fields");
  TORCH_CHECK(consts.size() <= PDE_AMD_MAX_CONSTS && lo.size() == 3 && hi.size() == 3 && n_eqs <= PDE_AMD_MAX_EQS, "pde_amd: bad problem");
  Config c{};
  const at::Tensor& f0 = fields[0];
  c.prob.pde_id = (int32_t)pde_id;
  c.prob.batch = (int32_t)f0.size(0);
  c.prob.frames = (int32_t)f0.size(1);
  const int nd = (int)f0.dim() - 2;
  for (int a = 0; a < 3; ++a) {
    c.prob.n[a] = a < 3 - nd ? 1 : (int32_t)f0.size(2 + a - (3 - nd));
    c.prob.lo[a] = (int32_t)lo[a];
    c.prob.hi[a] = (int32_t)hi[a];
  }
  c.prob.n_consts = (int32_t)consts.size();
  for (size_t i = 0; i < consts.size(); ++i) c.prob.consts[i] = (float)consts[i];
  c.n_eqs = (int)n_eqs;
  c.t_res = (int)std::max<int64_t>(f0.size(1) - t_lo - t_hi, 0);
  c.mode = (int)mode;
  c.need_grad = false;
  if (at::GradMode::is_enabled())
    for (const auto& f : fields) c.need_grad = c.need_grad || f.requires_grad();
  TORCH_CHECK(c.mode != 2 || !c.need_grad || (assumed.has_value() && assumed->defined()), "pde_amd: fused mode needs the assumed upstream scalar");
  TORCH_CHECK(f0.is_cuda(), "pde_amd: fields must be on a HIP device (there is no CPU fallback)");
  pde_amd_pde_info info;
  check(abi.get_pde_info(c.prob.pde_id, &info), "pde_info");
  TORCH_CHECK((int)fields.size() == info.n_fields && (int)sparams.size() == info.n_sparams && (int)aux.size() == info.n_aux &&
                  n_eqs == info.n_eqs && nd == info.ndim,
              "pde_amd: ", info.name, " takes ", info.n_fields, " fields [B, T, ", info.ndim, " spatial axes], ", info.n_sparams,
              " per-sample parameters and ", info.n_aux, " auxiliary fields");
  const at::Device dev = f0.device();
  std::vector<at::Tensor> fs, ax, sp;
  for (const auto& f : fields) {
    TORCH_CHECK(f.sizes() == f0.sizes() && f.device() == dev, "pde_amd: all fields must share one shape and device");
    fs.push_back(prepared(f, dev));
  }
  for (const auto& a : aux) {
    TORCH_CHECK(a.dim() == f0.dim() - 1 && a.size(0) == f0.size(0) && a.sizes().slice(1) == f0.sizes().slice(2),
                "pde_amd: auxiliary fields must be [B, *spatial]");
    ax.push_back(prepared(a, dev));
  }
  for (const auto& p : sparams) {
    TORCH_CHECK(p.dim() == 1 && p.size(0) == f0.size(0), "pde_amd: per-sample parameters must have shape [", f0.size(0), "]");
    sp.push_back(prepared(p, dev));
  }
  TORCH_CHECK(weights.numel() == n_eqs * (int64_t)c.t_res, "pde_amd: weights must be [n_eqs, T_res]");
  const bool given = totals_out.has_value() && totals_out->defined();
  if (given)
    TORCH_CHECK(totals_out->scalar_type() == at::kFloat && totals_out->device() == dev && totals_out->dim() == 1 &&
                    totals_out->numel() == n_eqs + 1 && totals_out->is_contiguous() && !totals_out->requires_grad(),
                "pde_amd: totals_out must be a contiguous fp32 vector of n_eqs + 1 = ", n_eqs + 1, " elements on the fields' device");
  return ResidualLosses::apply(at::TensorList(fs), at::TensorList(ax), at::TensorList(sp), prepared(weights, dev), assumed,
                               reinterpret_cast<int64_t>(&c), given ? reinterpret_cast<int64_t>(&*totals_out) : (int64_t)0);
}

}  // namespace

TORCH_LIBRARY(pde_amd, m) {
  m.def("residual_losses(Tensor[] fields, Tensor[] aux, Tensor[] sparams, Tensor weights, int pde_id, float[] consts, int[] lo, "
        "int[] hi, int n_eqs, int t_lo, int t_hi, int mode, Tensor? assumed, Tensor? totals_out=None) -> Tensor[]",
        &residual_losses);
}

// Binds the C ABI of the kernel library at `path` (already loaded by the process or not); returns 0, or -1 with the
// reason in dlerror().
extern "C" int pde_amd_torch_bind(const char* path) {
  void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return -1;
#define PDE_AMD_BIND(field, sym)                                        \
  abi.field = reinterpret_cast<decltype(abi.field)>(dlsym(h, #sym));    \
  if (!abi.field) return -1;
  PDE_AMD_BIND(workspace_bytes, pde_amd_residual_workspace_bytes)
  PDE_AMD_BIND(forward, pde_amd_residual_forward)
  PDE_AMD_BIND(backward, pde_amd_residual_backward)
  PDE_AMD_BIND(forward_backward, pde_amd_residual_forward_backward)
  PDE_AMD_BIND(rescale, pde_amd_rescale_gradients)
  PDE_AMD_BIND(status_string, pde_amd_status_string)
  PDE_AMD_BIND(get_pde_info, pde_amd_get_pde_info)
#undef PDE_AMD_BIND
  return 0;
}
