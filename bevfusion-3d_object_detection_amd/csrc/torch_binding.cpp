// torch_binding.cpp -- C++ autograd front-ends of the hottest entry points of libbevfusion_hip.so.
//
// The C ABI (include/bevfusion_hip.h) stays the boundary; this file only replaces the ctypes + Python autograd.Function
// plumbing for the ops that are called ~100 times per training step (fused BatchNorm over channels-last activations and
// over sparse feature matrices): a Python custom Function costs ~50 us of host time per call (forward + backward, GIL
// hand-over in the autograd engine), which made the step host-bound once the kernels themselves were fast.  Here a call
// is: pybind -> torch::autograd::Function::apply -> at::empty -> bfhip_* (plain C call) on the current HIP stream.
// Host-only C++ (no kernels); built by _build.py against the installed torch and linked to libbevfusion_hip.so.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>

#include "../../include/bevfusion_hip.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;

inline void *cur_stream(const Tensor &t) { return (void *)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }
inline int dt_code(const Tensor &t) { return t.scalar_type() == at::kBFloat16 ? 1 : 0; }
inline void check(int rc, const char *what) {
  TORCH_CHECK(rc == 0, what, " failed (rc=", rc, "): ", bfhip_last_error());
}

// y = act(BN_train(x) [+ residual]); x [N, C, H, W] channels-last (or [M, C, 1, 1]), f32 | bf16
class BN2dFn : public torch::autograd::Function<BN2dFn> {
 public:
  static Tensor forward(AutogradContext *ctx, const Tensor &x, const c10::optional<Tensor> &residual, const Tensor &weight,
                        const Tensor &bias, const Tensor &running_mean, const Tensor &running_var, double eps,
                        double momentum, bool relu, const c10::optional<Tensor> &partial, bool relu_bits) {
    const int64_t C = x.size(1), M = x.numel() / C;
    const int dt = dt_code(x);
    Tensor res;
    if (residual.has_value() && residual->defined()) {
      res = residual->scalar_type() == x.scalar_type() ? *residual : residual->to(x.scalar_type());
      res = res.contiguous(at::MemoryFormat::ChannelsLast);
    }
    Tensor y = at::empty_like(x);
    Tensor stats = at::empty({4 * C}, x.options().dtype(at::kFloat));
    Tensor mask;  // residual + ReLU behind a HIP conv: one bit per element for the backward instead of the saved output
    if (partial.has_value() && partial->defined()) {
      // the producing convolution accumulated the column sums in its epilogue: no statistics pass
      const Tensor &pt = *partial;
      if (relu_bits && relu && res.defined() && dt == 1 && C % 8 == 0) {
        mask = at::empty({M, C / 8}, x.options().dtype(at::kByte));
        check(bfhip_bn2d_fwd_partials_mask(x.data_ptr(), res.data_ptr(), weight.data_ptr<float>(), bias.data_ptr<float>(), M, (int)C, dt,
                                           (float)eps, (float)momentum, 1, running_mean.data_ptr<float>(),
                                           running_var.data_ptr<float>(), stats.data_ptr<float>(), y.data_ptr(),
                                           pt.data_ptr<float>(), (int)pt.size(0), nullptr, mask.data_ptr<uint8_t>(), cur_stream(x)),
              "bn2d_fwd_partials_mask");
      } else {
        check(bfhip_bn2d_fwd_partials(x.data_ptr(), res.defined() ? res.data_ptr() : nullptr, weight.data_ptr<float>(),
                                      bias.data_ptr<float>(), M, (int)C, dt, (float)eps, (float)momentum, relu ? 1 : 0,
                                      running_mean.data_ptr<float>(), running_var.data_ptr<float>(), stats.data_ptr<float>(),
                                      y.data_ptr(), pt.data_ptr<float>(), (int)pt.size(0), nullptr, cur_stream(x)),
              "bn2d_fwd_partials");
      }
    } else {
      const size_t wsb = bfhip_bn2d_workspace_bytes(M, (int)C, dt);
      Tensor ws = at::empty({(int64_t)wsb}, x.options().dtype(at::kByte));
      check(bfhip_bn2d_fwd(x.data_ptr(), res.defined() ? res.data_ptr() : nullptr, weight.data_ptr<float>(),
                           bias.data_ptr<float>(), M, (int)C, dt, (float)eps, (float)momentum, relu ? 1 : 0,
                           running_mean.data_ptr<float>(), running_var.data_ptr<float>(), stats.data_ptr<float>(), y.data_ptr(),
                           nullptr, ws.data_ptr(), wsb, cur_stream(x)),
            "bn2d_fwd");
    }
    const bool keep_y = relu && res.defined() && !mask.defined();  // otherwise the ReLU mask is recomputed from x / read from the bits
    ctx->save_for_backward({x, keep_y ? y : Tensor(), stats, weight, mask});
    ctx->saved_data["relu"] = relu;
    ctx->saved_data["has_res"] = res.defined();
    if (res.defined()) ctx->saved_data["res_dtype"] = (int64_t)residual->scalar_type();
    return y;
  }

  static tensor_list backward(AutogradContext *ctx, tensor_list grads) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &y = saved[1], &stats = saved[2], &weight = saved[3], &mask = saved[4];
    const bool relu = ctx->saved_data["relu"].toBool(), has_res = ctx->saved_data["has_res"].toBool();
    const int64_t C = x.size(1), M = x.numel() / C;
    const int dt = dt_code(x);
    Tensor dy = grads[0];
    if (dy.scalar_type() != x.scalar_type()) dy = dy.to(x.scalar_type());
    dy = dy.contiguous(at::MemoryFormat::ChannelsLast);
    Tensor dx = at::empty_like(x);
    Tensor dres = has_res ? at::empty_like(x) : Tensor();
    Tensor dgb = at::empty({2 * C}, x.options().dtype(at::kFloat));
    const size_t wsb = bfhip_bn2d_workspace_bytes(M, (int)C, dt);
    Tensor ws = at::empty({(int64_t)wsb}, x.options().dtype(at::kByte));
    if (mask.defined())
      check(bfhip_bn2d_bwd_mask(dy.data_ptr(), x.data_ptr(), mask.data_ptr<uint8_t>(), stats.data_ptr<float>(),
                                weight.data_ptr<float>(), M, (int)C, dt, dx.data_ptr(), dres.defined() ? dres.data_ptr() : nullptr,
                                dgb.data_ptr<float>(), nullptr, ws.data_ptr(), wsb, cur_stream(x)),
            "bn2d_bwd_mask");
    else
      check(bfhip_bn2d_bwd(dy.data_ptr(), x.data_ptr(), y.defined() ? y.data_ptr() : nullptr, stats.data_ptr<float>(),
                           weight.data_ptr<float>(), M, (int)C, dt, relu ? 1 : 0, dx.data_ptr(),
                           dres.defined() ? dres.data_ptr() : nullptr, dgb.data_ptr<float>(), nullptr, ws.data_ptr(), wsb, cur_stream(x)),
            "bn2d_bwd");
    if (has_res) {
      auto rdt = (at::ScalarType)ctx->saved_data["res_dtype"].toInt();
      if (rdt != dres.scalar_type()) dres = dres.to(rdt);
    }
    return {dx, dres, dgb.slice(0, 0, C), dgb.slice(0, C, 2 * C), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

Tensor bn2d(const Tensor &x, const c10::optional<Tensor> &residual, const Tensor &weight, const Tensor &bias,
            const Tensor &running_mean, const Tensor &running_var, double eps, double momentum, bool relu,
            const c10::optional<Tensor> &partial, bool relu_bits) {
  return BN2dFn::apply(x, residual, weight, bias, running_mean, running_var, eps, momentum, relu, partial, relu_bits);
}

// y = act(BN_train(x) [+ residual]) on sparse feature matrices f32[N, C]  (csrc/bn1d.hip)
class BN1dFn : public torch::autograd::Function<BN1dFn> {
 public:
  static Tensor forward(AutogradContext *ctx, const Tensor &x_, const c10::optional<Tensor> &residual, const Tensor &weight,
                        const Tensor &bias, const Tensor &running_mean, const Tensor &running_var, double eps,
                        double momentum, bool relu) {
    Tensor x = x_.contiguous();
    const int64_t N = x.size(0), C = x.size(1);
    Tensor res;
    if (residual.has_value() && residual->defined()) res = residual->contiguous();
    Tensor y = at::empty_like(x);
    Tensor stats = at::empty({2 * C}, x.options());
    const size_t wsb = bfhip_bn1d_workspace_bytes((int)N, (int)C);
    Tensor ws = at::empty({(int64_t)wsb}, x.options().dtype(at::kByte));
    check(bfhip_bn1d_fwd(x.data_ptr<float>(), res.defined() ? res.data_ptr<float>() : nullptr, weight.data_ptr<float>(),
                         bias.data_ptr<float>(), (int)N, (int)C, (float)eps, (float)momentum, relu ? 1 : 0,
                         running_mean.data_ptr<float>(), running_var.data_ptr<float>(), stats.data_ptr<float>(),
                         y.data_ptr<float>(), nullptr, ws.data_ptr(), wsb, cur_stream(x)),
          "bn1d_fwd");
    ctx->save_for_backward({x, y, stats, weight});
    ctx->saved_data["relu"] = relu;
    ctx->saved_data["has_res"] = res.defined();
    return y;
  }

  static tensor_list backward(AutogradContext *ctx, tensor_list grads) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &y = saved[1], &stats = saved[2], &weight = saved[3];
    const bool relu = ctx->saved_data["relu"].toBool(), has_res = ctx->saved_data["has_res"].toBool();
    const int64_t N = x.size(0), C = x.size(1);
    Tensor dy = grads[0].contiguous();
    Tensor dx = at::empty_like(x);
    Tensor dres = has_res ? at::empty_like(x) : Tensor();
    Tensor dgb = at::empty({2 * C}, x.options());
    const size_t wsb = bfhip_bn1d_workspace_bytes((int)N, (int)C);
    Tensor ws = at::empty({(int64_t)wsb}, x.options().dtype(at::kByte));
    check(bfhip_bn1d_bwd(dy.data_ptr<float>(), y.data_ptr<float>(), x.data_ptr<float>(), stats.data_ptr<float>(),
                         weight.data_ptr<float>(), (int)N, (int)C, relu ? 1 : 0, dx.data_ptr<float>(),
                         dres.defined() ? dres.data_ptr<float>() : nullptr, dgb.data_ptr<float>(), nullptr, ws.data_ptr(), wsb,
                         cur_stream(x)),
          "bn1d_bwd");
    return {dx, dres, dgb.slice(0, 0, C), dgb.slice(0, C, 2 * C), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

Tensor bn1d(const Tensor &x, const c10::optional<Tensor> &residual, const Tensor &weight, const Tensor &bias,
            const Tensor &running_mean, const Tensor &running_var, double eps, double momentum, bool relu) {
  return BN1dFn::apply(x, residual, weight, bias, running_mean, running_var, eps, momentum, relu);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "C++ autograd front-ends over libbevfusion_hip.so (include/bevfusion_hip.h)";
  m.def("bn2d", &bn2d, "fused BatchNorm2d(+residual)(+ReLU), channels-last, training mode; optional statistics partials of the producing conv",
        pybind11::arg("x"), pybind11::arg("residual"), pybind11::arg("weight"), pybind11::arg("bias"), pybind11::arg("running_mean"),
        pybind11::arg("running_var"), pybind11::arg("eps"), pybind11::arg("momentum"), pybind11::arg("relu"),
        pybind11::arg("partial") = c10::optional<Tensor>(), pybind11::arg("relu_bits") = false);
  m.def("bn1d", &bn1d, "fused BatchNorm1d(+residual)(+ReLU) on f32[N, C], training mode");
  m.def("abi_version", []() { return bfhip_abi_version(); });
}
