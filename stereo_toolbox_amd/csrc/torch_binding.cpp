// TORCH_LIBRARY binding of the hot path's C-ABI (include/stx_hip.h): the loader form north_star and SURVEY.md 8(b) name --
// `torch.utils.cpp_extension.load(...)` building a module that registers `torch.ops.stx.*` with schemas, device dispatch,
// Meta (shape) kernels and autograd for the volume builder.  It links against the SAME libstx_hip.so the ctypes binding
// loads (stereo_toolbox_amd/_capi.py): no kernel lives here, every op checks its operands with TORCH_CHECK, allocates its
// outputs with at::empty (the caching allocator owns them), launches on the current HIP stream of the operands' device and
// turns a non-zero status into a RuntimeError carrying stx_last_error().  There is no CPU kernel: a CPU tensor gets the
// dispatcher's "could not run ... with arguments from the 'CPU' backend".
//
// Ops (reference functions in the header's comments):
//   stx::cost_volume(Tensor? Lg, Tensor? Rg, Tensor? Lc, Tensor? Rc, int maxdisp, int num_groups, bool mask_left) -> Tensor
//        [B, D, H, W, G + 2 Cc] (dense channels-last volume), differentiable (stx_cost_volume_fwd / _bwd)
//   stx::conv3d_pack_weight(Tensor w, int mode) -> Tensor
//   stx::conv3d(Tensor x, Tensor wp, int cout, int ks, int stride, Tensor? scale, Tensor? bias, Tensor? residual, int act) -> Tensor
//   stx::deconv3d(Tensor x, Tensor wp, int cout, Tensor? scale, Tensor? bias, Tensor? residual, int act) -> Tensor
//   stx::conv3d_wgrad(Tensor fine, Tensor coarse, int ks, int stride) -> Tensor
//   stx::regression_head(Tensor cost, int maxdisp, int H, int W, bool align_corners) -> Tensor
//   stx::softargmax(Tensor x) -> Tensor        stx::argmax_disparity(Tensor x) -> Tensor
#include <torch/extension.h>
#include <torch/csrc/autograd/autograd_not_implemented_fallback.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include "../../include/stx_hip.h"

namespace {

// (PyTorch-ROCm tensors carry the device type "cuda": the guard / stream accessors that accept it are the "masquerading" ones)
using DeviceGuard = c10::hip::HIPGuardMasqueradingAsCUDA;
void* cur_stream() { return (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream(); }

void chk(const at::Tensor& t, const char* name, int64_t dims) {
    TORCH_CHECK(t.is_cuda(), name, ": expected a ROCm device tensor (the cost-volume hot path has no CPU fallback)");
    TORCH_CHECK(t.scalar_type() == at::kFloat, name, ": expected float32, got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), name, ": expected a dense contiguous tensor");
    TORCH_CHECK(dims < 0 || t.dim() == dims, name, ": expected ", dims, " dims, got ", t.dim());
}
const float* ptr(const c10::optional<at::Tensor>& t) { return t.has_value() && t->defined() ? t->data_ptr<float>() : nullptr; }
void status(int rc, const char* what) { TORCH_CHECK(rc == 0, what, " failed (", rc, "): ", stx_last_error()); }

struct VolShape { int64_t B, H, W, Cg, Cc; };
VolShape vol_shape(const c10::optional<at::Tensor>& Lg, const c10::optional<at::Tensor>& Lc, int64_t G) {
    const bool g = Lg.has_value() && Lg->defined(), c = Lc.has_value() && Lc->defined();
    TORCH_CHECK(g || c, "stx::cost_volume: neither gwc nor concat features given");
    const at::Tensor& r = g ? *Lg : *Lc;
    VolShape s{r.size(0), r.size(2), r.size(3), g ? Lg->size(1) : 0, c ? Lc->size(1) : 0};
    if (g) TORCH_CHECK(G > 0 && s.Cg % G == 0, "stx::cost_volume: ", s.Cg, " channels do not split into ", G, " groups");   // submodule.py:46
    return s;
}

at::Tensor cost_volume_fwd(const c10::optional<at::Tensor>& Lg, const c10::optional<at::Tensor>& Rg,
                           const c10::optional<at::Tensor>& Lc, const c10::optional<at::Tensor>& Rc, int64_t D, int64_t G,
                           bool mask_left) {
    for (auto* t : {&Lg, &Rg, &Lc, &Rc})
        if (t->has_value() && (*t)->defined()) chk(**t, "stx::cost_volume feature", 4);
    const VolShape s = vol_shape(Lg, Lc, G);
    const at::Tensor& r = (Lg.has_value() && Lg->defined()) ? *Lg : *Lc;
    const DeviceGuard guard(r.device());
    const int64_t Gn = s.Cg ? G : 0;
    auto vol = at::empty({s.B, D, s.H, s.W, Gn + 2 * s.Cc}, r.options());
    status(stx_cost_volume_fwd(ptr(Lg), ptr(Rg), (int)s.Cg, (int)Gn, ptr(Lc), ptr(Rc), (int)s.Cc, nullptr, vol.data_ptr<float>(),
                               (int)s.B, (int)s.H, (int)s.W, (int)D, mask_left ? 1 : 0, cur_stream()), "stx_cost_volume_fwd");
    return vol;
}

at::Tensor cost_volume_meta(const c10::optional<at::Tensor>& Lg, const c10::optional<at::Tensor>& Rg,
                            const c10::optional<at::Tensor>& Lc, const c10::optional<at::Tensor>& Rc, int64_t D, int64_t G,
                            bool mask_left) {
    const VolShape s = vol_shape(Lg, Lc, G);
    const at::Tensor& r = (Lg.has_value() && Lg->defined()) ? *Lg : *Lc;
    return at::empty({s.B, D, s.H, s.W, (s.Cg ? G : 0) + 2 * s.Cc}, r.options());
}

class CostVolumeFn : public torch::autograd::Function<CostVolumeFn> {
  public:
    static at::Tensor forward(torch::autograd::AutogradContext* ctx, const c10::optional<at::Tensor>& Lg,
                              const c10::optional<at::Tensor>& Rg, const c10::optional<at::Tensor>& Lc,
                              const c10::optional<at::Tensor>& Rc, int64_t D, int64_t G, bool mask_left) {
        at::AutoDispatchBelowADInplaceOrView guard;
        static auto op = c10::Dispatcher::singleton().findSchemaOrThrow("stx::cost_volume", "").typed<decltype(cost_volume_fwd)>();
        auto opt = [](const c10::optional<at::Tensor>& t) { return t.has_value() && t->defined() ? *t : at::Tensor(); };
        ctx->save_for_backward({opt(Lg), opt(Rg)});
        ctx->saved_data["D"] = D; ctx->saved_data["G"] = G; ctx->saved_data["mask_left"] = mask_left;
        ctx->saved_data["Cc"] = (Lc.has_value() && Lc->defined()) ? Lc->size(1) : (int64_t)0;
        return op.call(Lg, Rg, Lc, Rc, D, G, mask_left);
    }
    static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor Lg = saved[0], Rg = saved[1];
        const int64_t D = ctx->saved_data["D"].toInt(), G = ctx->saved_data["G"].toInt(), Cc = ctx->saved_data["Cc"].toInt();
        const bool mask_left = ctx->saved_data["mask_left"].toBool();
        const at::Tensor gvol = grads[0].contiguous();
        chk(gvol, "stx::cost_volume grad", 5);
        const DeviceGuard guard(gvol.device());
        const int64_t B = gvol.size(0), H = gvol.size(2), W = gvol.size(3);
        const int64_t Cg = Lg.defined() ? Lg.size(1) : 0;
        at::Tensor gLg, gRg, gLc, gRc;
        if (Cg) { gLg = at::empty_like(Lg); gRg = at::empty_like(Rg); }
        if (Cc) { gLc = at::empty({B, Cc, H, W}, gvol.options()); gRc = at::empty({B, Cc, H, W}, gvol.options()); }
        status(stx_cost_volume_bwd(gvol.data_ptr<float>(), Cg ? Lg.data_ptr<float>() : nullptr, Cg ? Rg.data_ptr<float>() : nullptr,
                                   (int)Cg, (int)(Cg ? G : 0), (int)Cc, Cg ? gLg.data_ptr<float>() : nullptr,
                                   Cg ? gRg.data_ptr<float>() : nullptr, Cc ? gLc.data_ptr<float>() : nullptr,
                                   Cc ? gRc.data_ptr<float>() : nullptr, (int)B, (int)H, (int)W, (int)D, mask_left ? 1 : 0,
                                   cur_stream()), "stx_cost_volume_bwd");
        return {gLg, gRg, gLc, gRc, at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor cost_volume_autograd(const c10::optional<at::Tensor>& Lg, const c10::optional<at::Tensor>& Rg,
                                const c10::optional<at::Tensor>& Lc, const c10::optional<at::Tensor>& Rc, int64_t D, int64_t G,
                                bool mask_left) {
    return CostVolumeFn::apply(Lg, Rg, Lc, Rc, D, G, mask_left);
}


at::Tensor pack_weight(const at::Tensor& w, int64_t mode) {
    chk(w, "stx::conv3d_pack_weight w", 5);
    const DeviceGuard guard(w.device());
    const int64_t A = w.size(0), Bd = w.size(1), T = w.size(2) * w.size(3) * w.size(4);
    const int64_t K = mode == 0 ? Bd : A, N = mode == 0 ? A : Bd;
    auto wp = at::empty({stx_conv3d_packed_floats((int)K, (int)N, (int)T)}, w.options());
    status(stx_conv3d_pack_weight(w.data_ptr<float>(), wp.data_ptr<float>(), (int)A, (int)Bd, (int)T, (int)mode, cur_stream()),
           "stx_conv3d_pack_weight");
    return wp;
}
at::Tensor pack_weight_meta(const at::Tensor& w, int64_t mode) {
    const int64_t A = w.size(0), Bd = w.size(1), T = w.size(2) * w.size(3) * w.size(4);
    const int64_t K = mode == 0 ? Bd : A, N = mode == 0 ? A : Bd;
    return at::empty({(int64_t)stx_conv3d_packed_floats((int)K, (int)N, (int)T)}, w.options());     // (host-only size query of the library)
}

std::array<int64_t, 3> conv_out(const at::Tensor& x, int64_t ks, int64_t stride) {
    const int64_t pad = ks / 2;
    return {(x.size(1) + 2 * pad - ks) / stride + 1, (x.size(2) + 2 * pad - ks) / stride + 1, (x.size(3) + 2 * pad - ks) / stride + 1};
}

at::Tensor conv3d(const at::Tensor& x, const at::Tensor& wp, int64_t cout, int64_t ks, int64_t stride,
                  const c10::optional<at::Tensor>& scale, const c10::optional<at::Tensor>& bias,
                  const c10::optional<at::Tensor>& residual, int64_t act) {
    chk(x, "stx::conv3d x", 5);
    chk(wp, "stx::conv3d packed weight", 1);
    const DeviceGuard guard(x.device());
    const auto o = conv_out(x, ks, stride);
    auto out = at::empty({x.size(0), o[0], o[1], o[2], cout}, x.options());
    status(stx_conv3d_fwd(x.data_ptr<float>(), wp.data_ptr<float>(), out.data_ptr<float>(), ptr(scale), ptr(bias), ptr(residual),
                          nullptr, (int)x.size(0), (int)x.size(1), (int)x.size(2), (int)x.size(3), (int)x.size(4), (int)cout,
                          (int)ks, (int)stride, (int)act, cur_stream()), "stx_conv3d_fwd");
    return out;
}
at::Tensor conv3d_meta(const at::Tensor& x, const at::Tensor&, int64_t cout, int64_t ks, int64_t stride,
                       const c10::optional<at::Tensor>&, const c10::optional<at::Tensor>&, const c10::optional<at::Tensor>&, int64_t) {
    const auto o = conv_out(x, ks, stride);
    return at::empty({x.size(0), o[0], o[1], o[2], cout}, x.options());
}

at::Tensor deconv3d(const at::Tensor& x, const at::Tensor& wp, int64_t cout, const c10::optional<at::Tensor>& scale,
                    const c10::optional<at::Tensor>& bias, const c10::optional<at::Tensor>& residual, int64_t act) {
    chk(x, "stx::deconv3d x", 5);
    chk(wp, "stx::deconv3d packed weight", 1);
    const DeviceGuard guard(x.device());
    auto out = at::empty({x.size(0), 2 * x.size(1), 2 * x.size(2), 2 * x.size(3), cout}, x.options());
    status(stx_deconv3d_fwd(x.data_ptr<float>(), wp.data_ptr<float>(), out.data_ptr<float>(), ptr(scale), ptr(bias), ptr(residual),
                            nullptr, (int)x.size(0), (int)x.size(1), (int)x.size(2), (int)x.size(3), (int)x.size(4), (int)cout,
                            (int)(2 * x.size(1)), (int)(2 * x.size(2)), (int)(2 * x.size(3)), (int)act, cur_stream()),
           "stx_deconv3d_fwd");
    return out;
}
at::Tensor deconv3d_meta(const at::Tensor& x, const at::Tensor&, int64_t cout, const c10::optional<at::Tensor>&,
                         const c10::optional<at::Tensor>&, const c10::optional<at::Tensor>&, int64_t) {
    return at::empty({x.size(0), 2 * x.size(1), 2 * x.size(2), 2 * x.size(3), cout}, x.options());
}

at::Tensor conv3d_wgrad(const at::Tensor& fine, const at::Tensor& coarse, int64_t ks, int64_t stride) {
    chk(fine, "stx::conv3d_wgrad fine", 5);
    chk(coarse, "stx::conv3d_wgrad coarse", 5);
    const DeviceGuard guard(fine.device());
    const int B = (int)fine.size(0), CF = (int)fine.size(4), CC = (int)coarse.size(4);
    const long long n = stx_conv3d_wgrad_workspace_floats(B, (int)coarse.size(1), (int)coarse.size(2), (int)coarse.size(3), CF, CC,
                                                          (int)ks, (int)stride);
    TORCH_CHECK(n > 0, "stx::conv3d_wgrad: unsupported shape (channel counts must be multiples of 32)");
    auto ws = at::empty({n}, fine.options());
    auto dw = at::empty({CC, CF, ks * ks * ks}, fine.options());
    status(stx_conv3d_wgrad(fine.data_ptr<float>(), coarse.data_ptr<float>(), dw.data_ptr<float>(), ws.data_ptr<float>(), B,
                            (int)fine.size(1), (int)fine.size(2), (int)fine.size(3), CF, (int)coarse.size(1), (int)coarse.size(2),
                            (int)coarse.size(3), CC, (int)ks, (int)stride, cur_stream()), "stx_conv3d_wgrad");
    return dw;
}
at::Tensor conv3d_wgrad_meta(const at::Tensor& fine, const at::Tensor& coarse, int64_t ks, int64_t) {
    return at::empty({coarse.size(4), fine.size(4), ks * ks * ks}, fine.options());
}

at::Tensor regression_head(const at::Tensor& cost, int64_t maxdisp, int64_t H, int64_t W, bool align_corners) {
    chk(cost, "stx::regression_head cost", 4);
    const DeviceGuard guard(cost.device());
    auto disp = at::empty({cost.size(0), H, W}, cost.options());
    status(stx_head_fwd2(cost.data_ptr<float>(), disp.data_ptr<float>(), nullptr, (int)cost.size(0), (int)cost.size(1),
                         (int)cost.size(2), (int)cost.size(3), (int)maxdisp, (int)H, (int)W, align_corners ? 1 : 0, cur_stream()),
           "stx_head_fwd2");
    return disp;
}
at::Tensor regression_head_meta(const at::Tensor& cost, int64_t, int64_t H, int64_t W, bool) {
    return at::empty({cost.size(0), H, W}, cost.options());
}

at::Tensor softargmax(const at::Tensor& x) {
    chk(x, "stx::softargmax x", 4);
    const DeviceGuard guard(x.device());
    auto out = at::empty({x.size(0), 1, x.size(2), x.size(3)}, x.options());
    status(stx_softargmax_fwd(x.data_ptr<float>(), out.data_ptr<float>(), (int)x.size(0), (int)x.size(1),
                              (int)(x.size(2) * x.size(3)), cur_stream()), "stx_softargmax_fwd");
    return out;
}
at::Tensor softargmax_meta(const at::Tensor& x) { return at::empty({x.size(0), 1, x.size(2), x.size(3)}, x.options()); }

at::Tensor argmax_disparity(const at::Tensor& x) {
    chk(x, "stx::argmax_disparity x", 4);
    const DeviceGuard guard(x.device());
    auto out = at::empty({x.size(0), 1, x.size(2), x.size(3)}, x.options().dtype(at::kLong));
    status(stx_argmax_fwd(x.data_ptr<float>(), (long long*)out.data_ptr<int64_t>(), (int)x.size(0), (int)x.size(1),
                          (int)(x.size(2) * x.size(3)), cur_stream()), "stx_argmax_fwd");
    return out;
}
at::Tensor argmax_disparity_meta(const at::Tensor& x) {
    return at::empty({x.size(0), 1, x.size(2), x.size(3)}, x.options().dtype(at::kLong));
}

}  // namespace

TORCH_LIBRARY(stx, m) {
    m.def("cost_volume(Tensor? Lg, Tensor? Rg, Tensor? Lc, Tensor? Rc, int maxdisp, int num_groups, bool mask_left) -> Tensor");
    m.def("conv3d_pack_weight(Tensor w, int mode) -> Tensor");
    m.def("conv3d(Tensor x, Tensor wp, int cout, int ks, int stride, Tensor? scale, Tensor? bias, Tensor? residual, int act) -> Tensor");
    m.def("deconv3d(Tensor x, Tensor wp, int cout, Tensor? scale, Tensor? bias, Tensor? residual, int act) -> Tensor");
    m.def("conv3d_wgrad(Tensor fine, Tensor coarse, int ks, int stride) -> Tensor");
    m.def("regression_head(Tensor cost, int maxdisp, int H, int W, bool align_corners) -> Tensor");
    m.def("softargmax(Tensor x) -> Tensor");
    m.def("argmax_disparity(Tensor x) -> Tensor");
    m.def("build_info() -> str", []() { return std::string(stx_build_info()); });
}
TORCH_LIBRARY_IMPL(stx, CUDA, m) {          // (ROCm devices are the "CUDA" dispatch key of PyTorch-ROCm)
    m.impl("cost_volume", &cost_volume_fwd);
    m.impl("conv3d_pack_weight", &pack_weight);
    m.impl("conv3d", &conv3d);
    m.impl("deconv3d", &deconv3d);
    m.impl("conv3d_wgrad", &conv3d_wgrad);
    m.impl("regression_head", &regression_head);
    m.impl("softargmax", &softargmax);
    m.impl("argmax_disparity", &argmax_disparity);
}
TORCH_LIBRARY_IMPL(stx, Meta, m) {
    m.impl("cost_volume", &cost_volume_meta);
    m.impl("conv3d_pack_weight", &pack_weight_meta);
    m.impl("conv3d", &conv3d_meta);
    m.impl("deconv3d", &deconv3d_meta);
    m.impl("conv3d_wgrad", &conv3d_wgrad_meta);
    m.impl("regression_head", &regression_head_meta);
    m.impl("softargmax", &softargmax_meta);
    m.impl("argmax_disparity", &argmax_disparity_meta);
}
TORCH_LIBRARY_IMPL(stx, Autograd, m) {
    m.impl("cost_volume", &cost_volume_autograd);
    // the other operators are forward entry points of the C-ABI without a registered derivative here (the models differentiate
    // through stereo_toolbox_amd.ops' autograd Functions): inputs that require grad get a grad_fn that raises at backward
    // ("derivative for stx::conv3d is not implemented") instead of an output silently cut off the graph
    for (const char* name : {"conv3d_pack_weight", "conv3d", "deconv3d", "conv3d_wgrad", "regression_head", "softargmax",
                             "argmax_disparity"})
        m.impl(name, torch::autograd::autogradNotImplementedFallback());
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) { m.doc() = "torch.ops.stx: TORCH_LIBRARY binding of libstx_hip.so"; }
