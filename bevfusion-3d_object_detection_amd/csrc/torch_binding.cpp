// torch_binding.cpp -- C++ autograd front-ends of the hottest entry points of libbevfusion_hip.so.
//
// The C ABI (include/bevfusion_hip.h) stays the boundary; this file only replaces the ctypes + Python autograd.Function
// plumbing for the ops that are called ~100 times per training step (fused BatchNorm over channels-last activations and
// over sparse feature matrices; the dense convolution Functions with the end-of-pass group of their weight gradients): a
// Python custom Function costs ~50 us of host time per call (forward + backward, GIL hand-over in the autograd engine),
// which made the step host-bound once the kernels themselves were fast.  Here a call is: pybind ->
// torch::autograd::Function::apply -> at::empty -> bfhip_* (plain C call) on the current HIP stream.
// Host-only C++ (no kernels); built by _build.py against the installed torch and linked to libbevfusion_hip.so.
#include <torch/extension.h>
#include <torch/csrc/autograd/engine.h>
#include <torch/csrc/autograd/graph_task.h>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPGuard.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>

#include <map>
#include <mutex>
#include <vector>

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


// ------------------------------------------------------------------------------------------------ dense convolution
// The same three Functions as conv2d.py (_Conv2dFunction, _LibConvHipWgradFunction and the end-of-pass group of the weight
// gradients): ~110 calls per training step, forward and backward, each of which cost 20-30 us of interpreter time in Python.
// Semantics, argument meaning and fall-backs follow the Python classes line by line; conv2d.py keeps them as the path without
// this extension (BFHIP_TORCH_EXT=0) and as the documentation of record.

// pixel pitch (elements) when t [N, C, H, W] is channels-last dense or a channel slice of such a tensor, else 0
inline int64_t nhwc_pitch(const Tensor &t) {
  const auto sz = t.sizes();
  const auto st = t.strides();
  const int64_t N = sz[0], C = sz[1], H = sz[2], W = sz[3], sn = st[0], sc = st[1], sh = st[2], sw = st[3];
  if (sc == 1 && sw >= C && sh == W * sw && (sn == H * W * sw || N == 1) && sw % 8 == 0 && ((uintptr_t)t.data_ptr()) % 16 == 0) return sw;
  return 0;
}

inline Tensor as_nhwc_bf16(Tensor t) {
  if (t.scalar_type() != at::kBFloat16) t = t.to(at::kBFloat16);
  if (!nhwc_pitch(t)) {
    t = t.contiguous(at::MemoryFormat::ChannelsLast);
    if (!nhwc_pitch(t)) t = t.permute({0, 2, 3, 1}).contiguous().permute({0, 3, 1, 2});  // C == 1 / W == 1 stride normalisation
  }
  return t;
}

// bf16 [Cout][KH][KW][Cin] memory of a conv weight [Cout, Cin, KH, KW]
inline Tensor weight_ohwi(const Tensor &w_) {
  Tensor w = w_.scalar_type() == at::kBFloat16 ? w_ : w_.to(at::kBFloat16);
  Tensor p = w.permute({0, 2, 3, 1});
  return p.is_contiguous() ? p : p.contiguous();
}

inline Tensor empty_nhwc(int64_t N, int64_t C, int64_t H, int64_t W, const at::TensorOptions &opt) {
  return at::empty({N, H, W, C}, opt).permute({0, 3, 1, 2});
}

// ---- weight gradients of a backward pass, grouped (conv2d.py: WGRAD_GROUPED; include/bevfusion_hip.h: bfhip_conv2d_wgrad_group_*)
struct PendingWgrad {
  Tensor x, dy, weight;
  bfhip_wgrad_layer row;
  void *stream;
};
struct GroupState {  // per device: two pinned table images (alternating: a copy may still be in flight), the device copy, the slabs
  Tensor host[2], dev, slab;
  int flip = 0;
};
std::mutex g_mu;
// (heap objects that are never destroyed: their tensors must not be released after the HIP runtime / torch's allocators at exit)
std::map<int, std::vector<PendingWgrad>> &g_pending = *new std::map<int, std::vector<PendingWgrad>>();  // graph task id -> records
std::map<int, GroupState> &g_group = *new std::map<int, GroupState>();                                  // device index -> state
bool g_grouped = true;

void flush_wgrads(int tid) {
  std::vector<PendingWgrad> pend;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_pending.find(tid);
    if (it == g_pending.end()) return;
    pend = std::move(it->second);
    g_pending.erase(it);
  }
  if (pend.empty()) return;
  static std::mutex flush_mu;  // the per-device buffers below are shared by all passes: one flush at a time
  std::lock_guard<std::mutex> one_at_a_time(flush_mu);
  at::NoGradGuard no_grad;
  std::map<int, std::vector<size_t>> by_dev;
  for (size_t i = 0; i < pend.size(); ++i) by_dev[pend[i].x.get_device()].push_back(i);
  for (auto &kv : by_dev) {
    const int dev = kv.first;
    const auto &idx = kv.second;
    const int n = (int)idx.size();
    c10::hip::HIPGuard guard((c10::DeviceIndex)dev);
    // torch's tensors on ROCm carry the CUDA device type: the stream handed to record_stream must wear it too
    auto cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA((c10::DeviceIndex)dev);
    void *raw = (void *)cur.stream();
    GroupState &st = g_group[dev];
    std::vector<bfhip_wgrad_layer> rows((size_t)n);
    std::vector<Tensor> dws((size_t)n);
    for (int k = 0; k < n; ++k) {
      PendingWgrad &e = pend[idx[k]];
      if (e.stream != raw) {  // produced on another stream than the one the group runs on (the engine has already joined them)
        e.x.record_stream(cur);
        e.dy.record_stream(cur);
      }
      const auto ws = e.weight.sizes();
      dws[k] = empty_nhwc(ws[0], ws[1], ws[2], ws[3], e.weight.options());
      rows[k] = e.row;
      rows[k].dw = dws[k].data_ptr();
    }
    const size_t nbytes = bfhip_conv2d_wgrad_group_table_bytes(n);
    if (!st.dev.defined() || (size_t)st.dev.numel() < nbytes) {
      const int64_t cap = std::max<int64_t>((int64_t)nbytes * 2, 1 << 16);
      for (auto &h : st.host) h = at::empty({cap}, at::TensorOptions().dtype(at::kByte).pinned_memory(true));
      st.dev = at::empty({cap}, at::TensorOptions().dtype(at::kByte).device(at::kCUDA, dev));
    }
    st.flip ^= 1;
    Tensor &host = st.host[st.flip];
    size_t slab_bytes = 0;
    check(bfhip_conv2d_wgrad_group_plan(rows.data(), n, 0, host.data_ptr(), nbytes, &slab_bytes), "conv2d_wgrad_group_plan");
    if (!st.slab.defined() || (size_t)st.slab.numel() < slab_bytes) {
      st.slab = Tensor();
      st.slab = at::empty({(int64_t)(slab_bytes + slab_bytes / 4) + 256}, at::TensorOptions().dtype(at::kByte).device(at::kCUDA, dev));
    }
    st.dev.narrow(0, 0, (int64_t)nbytes).copy_(host.narrow(0, 0, (int64_t)nbytes), /*non_blocking=*/true);
    check(bfhip_conv2d_wgrad_group_launch(host.data_ptr(), st.dev.data_ptr(), st.slab.data_ptr(), (size_t)st.slab.numel(), raw),
          "conv2d_wgrad_group_launch");
    for (int k = 0; k < n; ++k) {
      Tensor &w = pend[idx[k]].weight;
      if (!w.grad().defined()) w.mutable_grad() = dws[k];
      else w.mutable_grad().add_(dws[k]);
    }
  }
}

// true when the layer's weight gradient was queued for the grouped launch at the end of the running backward pass
bool defer_wgrad(const Tensor &x, const Tensor &dy, const Tensor &weight, int64_t stride, int64_t pad, int64_t dil) {
  if (!g_grouped) return false;
  const int64_t N = x.size(0), Cin = x.size(1), H = x.size(2), W = x.size(3);
  const int64_t Cout = weight.size(0), KH = weight.size(2), KW = weight.size(3);
  if (!weight.is_leaf() || !weight.requires_grad()) return false;
  if (weight.scalar_type() != at::kBFloat16 && weight.scalar_type() != at::kFloat) return false;
  if (!bfhip_conv2d_wgrad_groupable((int)N, (int)H, (int)W, (int)Cin, (int)Cout, (int)KH, (int)KW, (int)stride, (int)pad, (int)dil)) return false;
  const int64_t ldx = nhwc_pitch(x), ldg = nhwc_pitch(dy);
  if (!ldx || !ldg) return false;
  void *stream = cur_stream(x);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (cap != hipStreamCaptureStatusNone) return false;
  const int tid = torch::autograd::get_current_graph_task_id();
  if (tid < 0) return false;
  // torch.autograd.grad(..., inputs) / backward(inputs=...): the engine captures the requested gradients from the graph and must
  // not touch .grad -- the pass has a non-empty execution plan then; every layer launches its own weight gradient and returns it
  const auto *plan = torch::autograd::get_current_graph_task_exec_info();
  if (plan && !plan->empty()) return false;
  PendingWgrad e;
  e.x = x; e.dy = dy; e.weight = weight; e.stream = stream;
  e.row = bfhip_wgrad_layer{x.data_ptr(), dy.data_ptr(), nullptr, (int32_t)ldx, (int32_t)ldg, (int32_t)N, (int32_t)H, (int32_t)W,
                            (int32_t)Cin, (int32_t)Cout, (int32_t)KH, (int32_t)KW, (int32_t)stride, (int32_t)pad, (int32_t)dil,
                            weight.scalar_type() == at::kBFloat16 ? 1 : 0, 0};
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_pending.find(tid);
  if (it == g_pending.end()) {
    // one list per backward pass: a pass that died with an exception never runs its callback; its records must not leak into
    // the next pass, which queues its own callback
    torch::autograd::Engine::get_default_engine().queue_callback([tid]() { flush_wgrads(tid); });
    for (auto old = g_pending.begin(); old != g_pending.end();) old = old->first < tid - 8 ? g_pending.erase(old) : std::next(old);
    it = g_pending.emplace(tid, std::vector<PendingWgrad>()).first;
  }
  it->second.push_back(std::move(e));
  return true;
}

int64_t pending_wgrads() {
  std::lock_guard<std::mutex> lock(g_mu);
  int64_t n = 0;
  for (auto &kv : g_pending) n += (int64_t)kv.second.size();
  return n;
}

// dW [Cout, Cin, KH, KW] (channels-last memory) in line, or undefined when the layer joined the group
Tensor launch_wgrad(const Tensor &x, const Tensor &dy, const Tensor &weight, int64_t stride, int64_t pad, int64_t dil) {
  if (defer_wgrad(x, dy, weight, stride, pad, dil)) return Tensor();
  const int64_t N = x.size(0), Cin = x.size(1), H = x.size(2), W = x.size(3);
  const int64_t Cout = weight.size(0), KH = weight.size(2), KW = weight.size(3), OH = dy.size(2), OW = dy.size(3);
  const bool out_bf16 = weight.scalar_type() == at::kBFloat16;
  Tensor dw = empty_nhwc(Cout, Cin, KH, KW, x.options().dtype(out_bf16 ? at::kBFloat16 : at::kFloat));
  const size_t wsb = bfhip_conv2d_wgrad_workspace_bytes((int)N, (int)OH, (int)OW, (int)Cin, (int)Cout, (int)KH, (int)KW);
  Tensor ws = at::empty({(int64_t)std::max<size_t>(wsb, 256)}, x.options().dtype(at::kByte));
  check(bfhip_conv2d_wgrad(x.data_ptr(), (int)nhwc_pitch(x), dy.data_ptr(), (int)nhwc_pitch(dy), dw.data_ptr(), (int)N, (int)H, (int)W,
                           (int)Cin, (int)Cout, (int)KH, (int)KW, (int)stride, (int)pad, (int)dil, out_bf16 ? 1 : 0, ws.data_ptr(),
                           (size_t)ws.numel(), cur_stream(x)),
        "conv2d_wgrad");
  return dw.scalar_type() == weight.scalar_type() ? dw : dw.to(weight.scalar_type());
}

inline Tensor add_grad(Tensor dx, const Tensor &addend, int fork) {
  if (!addend.defined()) return dx;
  if (fork == 2) {
    using torch::indexing::Slice;
    dx.index({Slice(), Slice(), Slice(0, c10::nullopt, 2), Slice(0, c10::nullopt, 2)}).add_(addend);
    return dx;
  }
  return dx + addend;
}

// dx of a convolution on csrc/conv2d.hip (conv2d.py::_hip_dgrad): wt = the cached transposed weight (or undefined)
Tensor hip_dgrad(const Tensor &dy, const Tensor &x, const Tensor &weight, const Tensor &wt, int64_t stride, int64_t pad, int64_t dil,
                 const Tensor &addend, int addend_stride) {
  const int64_t N = x.size(0), Cin = x.size(1), H = x.size(2), W = x.size(3);
  const int64_t Cout = weight.size(0), KH = weight.size(2), KW = weight.size(3);
  Tensor dx = empty_nhwc(N, Cin, H, W, x.options().dtype(at::kBFloat16));
  if (wt.defined()) {
    bool fuse = false;
    if (addend.defined() && addend.scalar_type() == at::kBFloat16 && addend.dim() == 4 && addend.size(0) == N && addend.size(1) == Cin &&
        addend.size(2) == (addend_stride == 2 ? (H + 1) / 2 : H) && addend.size(3) == (addend_stride == 2 ? (W + 1) / 2 : W) &&
        addend.is_contiguous(at::MemoryFormat::ChannelsLast) && ((uintptr_t)addend.data_ptr()) % 16 == 0)
      fuse = bfhip_conv2d_dgrad_fuses_addend((int)KH, (int)KW, (int)stride, (int)pad, 0) != 0;
    check(bfhip_conv2d_dgrad_wt(dy.data_ptr(), (int)nhwc_pitch(dy), wt.data_ptr(), fuse ? addend.data_ptr() : nullptr, addend_stride,
                                dx.data_ptr(), (int)Cin, (int)N, (int)H, (int)W, (int)Cin, (int)Cout, (int)KH, (int)KW, (int)stride,
                                (int)pad, (int)dil, 0, cur_stream(x)),
          "conv2d_dgrad_wt");
    return fuse ? dx : add_grad(dx, addend, addend_stride);
  }
  const size_t wsb = bfhip_conv2d_dgrad_workspace_bytes((int)Cin, (int)Cout, (int)KH, (int)KW);
  Tensor ws = at::empty({(int64_t)std::max<size_t>(wsb, 256)}, x.options().dtype(at::kByte));
  Tensor w = weight_ohwi(weight);
  check(bfhip_conv2d_dgrad(dy.data_ptr(), (int)nhwc_pitch(dy), w.data_ptr(), dx.data_ptr(), (int)Cin, (int)N, (int)H, (int)W, (int)Cin,
                           (int)Cout, (int)KH, (int)KW, (int)stride, (int)pad, (int)dil, 0, ws.data_ptr(), (size_t)ws.numel(), cur_stream(x)),
        "conv2d_dgrad");
  return add_grad(dx, addend, addend_stride);
}

inline Tensor lib_dgrad(const Tensor &dy, const Tensor &x, const Tensor &weight, int64_t stride, int64_t pad, int64_t dil) {
  Tensor w = weight.scalar_type() == at::kBFloat16 ? weight : weight.to(at::kBFloat16);
  return std::get<0>(at::convolution_backward(dy, x, w, c10::nullopt, {stride, stride}, {pad, pad}, {dil, dil}, false, {0, 0}, 1,
                                              {true, false, false}));
}

// f32[C] = sum of dy over batch and pixels (conv2d.py::_bias_grad)
inline Tensor bias_grad(const Tensor &dy) {
  const int64_t N = dy.size(0), C = dy.size(1), H = dy.size(2), W = dy.size(3), M = N * H * W;
  if (nhwc_pitch(dy) == C && M >= 4096 && bfhip_bn2d_supported(M, (int)C, 1)) {
    const size_t wsb = bfhip_bn2d_workspace_bytes(M, (int)C, 1);
    if (wsb > 0) {
      Tensor out = at::empty({C}, dy.options().dtype(at::kFloat));
      Tensor ws = at::empty({(int64_t)wsb}, dy.options().dtype(at::kByte));
      check(bfhip_colsum(dy.data_ptr(), M, (int)C, 1, out.data_ptr<float>(), ws.data_ptr(), wsb, cur_stream(dy)), "colsum");
      return out;
    }
  }
  return dy.sum({0, 2, 3}, false, at::kFloat);
}

// conv2d.py::_Conv2dFunction.  Outputs: y, partial (an EMPTY tensor when no statistics were asked for) [, x' with `fork`]
class ConvFn : public torch::autograd::Function<ConvFn> {
 public:
  static tensor_list forward(AutogradContext *ctx, const Tensor &x_, const Tensor &weight, const c10::optional<Tensor> &bias,
                             int64_t stride, int64_t pad, int64_t dil, bool emit_stats, bool dgrad_lib, int64_t fork,
                             const c10::optional<Tensor> &wt) {
    ctx->set_materialize_grads(false);
    Tensor x = as_nhwc_bf16(x_);
    const int64_t N = x.size(0), Cin = x.size(1), H = x.size(2), W = x.size(3);
    const int64_t Cout = weight.size(0), KH = weight.size(2), KW = weight.size(3);
    const int64_t OH = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, OW = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
    Tensor w = weight_ohwi(weight);
    Tensor y = empty_nhwc(N, Cout, OH, OW, x.options());
    Tensor partial = emit_stats ? at::empty({(int64_t)bfhip_conv2d_stat_rows((int)N, (int)OH, (int)OW), 2, Cout}, x.options().dtype(at::kFloat))
                                : at::empty({0}, x.options().dtype(at::kFloat));
    Tensor b32;
    const bool has_bias = bias.has_value() && bias->defined();
    if (has_bias) b32 = bias->scalar_type() == at::kFloat ? *bias : bias->to(at::kFloat);
    check(bfhip_conv2d_fwd(x.data_ptr(), (int)nhwc_pitch(x), w.data_ptr(), has_bias ? b32.data_ptr<float>() : nullptr, y.data_ptr(),
                           (int)Cout, (int)N, (int)H, (int)W, (int)Cin, (int)Cout, (int)KH, (int)KW, (int)stride, (int)pad, (int)dil, 0,
                           emit_stats ? partial.data_ptr<float>() : nullptr, cur_stream(x)),
          "conv2d_fwd");
    ctx->save_for_backward({x, weight, (wt.has_value() && wt->defined()) ? *wt : Tensor()});
    ctx->saved_data["geom"] = std::vector<int64_t>{stride, pad, dil, fork, dgrad_lib ? 1 : 0, has_bias ? 1 : 0,
                                                   has_bias ? (int64_t)bias->scalar_type() : 0};
    ctx->mark_non_differentiable({partial});
    if (fork == 2) {
      using torch::indexing::Slice;
      // third output: x at its even pixels (what a stride-2 1x1 shortcut reads); its compact gradient comes back to THIS node
      return {y, partial, x.index({Slice(), Slice(), Slice(0, c10::nullopt, 2), Slice(0, c10::nullopt, 2)}).contiguous(at::MemoryFormat::ChannelsLast)};
    }
    if (fork) return {y, partial, x.view_as(x)};  // the input again, as the second consumer's handle (identity branch)
    return {y, partial};
  }

  static tensor_list backward(AutogradContext *ctx, tensor_list grads) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &weight = saved[1], &wt = saved[2];
    const auto g = ctx->saved_data["geom"].toIntVector();
    const int64_t stride = g[0], pad = g[1], dil = g[2];
    const int fork = (int)g[3];
    const bool dgrad_lib = g[4] != 0, has_bias = g[5] != 0;
    Tensor dy = grads[0];
    Tensor d_alias = grads.size() > 2 ? grads[2] : Tensor();
    tensor_list out(10);
    if (!dy.defined()) {
      if (d_alias.defined() && fork == 2) {
        using torch::indexing::Slice;
        Tensor full = at::zeros_like(x);
        full.index_put_({Slice(), Slice(), Slice(0, c10::nullopt, 2), Slice(0, c10::nullopt, 2)}, d_alias);
        d_alias = full;
      }
      out[0] = d_alias;
      return out;
    }
    dy = as_nhwc_bf16(dy);
    if (ctx->needs_input_grad(0)) {
      if (dgrad_lib) out[0] = add_grad(lib_dgrad(dy, x, weight, stride, pad, dil), d_alias, fork);
      else out[0] = hip_dgrad(dy, x, weight, wt, stride, pad, dil, d_alias, fork == 2 ? 2 : 1);
    }
    if (ctx->needs_input_grad(1)) out[1] = launch_wgrad(x, dy, weight, stride, pad, dil);
    if (has_bias && ctx->needs_input_grad(2)) out[2] = bias_grad(dy).to((at::ScalarType)g[6]);
    return out;
  }
};

tensor_list conv2d(const Tensor &x, const Tensor &weight, const c10::optional<Tensor> &bias, int64_t stride, int64_t pad, int64_t dil,
                   bool emit_stats, bool dgrad_lib, int64_t fork, const c10::optional<Tensor> &wt) {
  return ConvFn::apply(x, weight, bias, stride, pad, dil, emit_stats, dgrad_lib, fork, wt);
}

// conv2d.py::_LibConvHipWgradFunction: forward (and by default the data gradient) by the library, weight gradient on csrc/conv2d.hip
class LibConvFn : public torch::autograd::Function<LibConvFn> {
 public:
  static Tensor forward(AutogradContext *ctx, const Tensor &x_, const Tensor &weight, int64_t stride, int64_t pad, int64_t dil,
                        bool dgrad_hip, const c10::optional<Tensor> &wt) {
    Tensor x = as_nhwc_bf16(x_);
    Tensor w = weight.scalar_type() == at::kBFloat16 ? weight : weight.to(at::kBFloat16);
    Tensor y;
    {
      c10::impl::ExcludeDispatchKeyGuard no_autocast(c10::DispatchKey::AutocastCUDA);
      y = at::conv2d(x, w, {}, {stride, stride}, {pad, pad}, {dil, dil});
    }
    ctx->save_for_backward({x, weight, (wt.has_value() && wt->defined()) ? *wt : Tensor()});
    ctx->saved_data["geom"] = std::vector<int64_t>{stride, pad, dil, dgrad_hip ? 1 : 0};
    return y;
  }

  static tensor_list backward(AutogradContext *ctx, tensor_list grads) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &weight = saved[1], &wt = saved[2];
    const auto g = ctx->saved_data["geom"].toIntVector();
    const int64_t stride = g[0], pad = g[1], dil = g[2];
    tensor_list out(7);
    if (!grads[0].defined()) return out;
    Tensor dy = as_nhwc_bf16(grads[0]);
    if (ctx->needs_input_grad(0))
      out[0] = g[3] ? hip_dgrad(dy, x, weight, wt, stride, pad, dil, Tensor(), 1) : lib_dgrad(dy, x, weight, stride, pad, dil);
    if (ctx->needs_input_grad(1)) out[1] = launch_wgrad(x, dy, weight, stride, pad, dil);
    return out;
  }
};

Tensor lib_conv2d(const Tensor &x, const Tensor &weight, int64_t stride, int64_t pad, int64_t dil, bool dgrad_hip,
                  const c10::optional<Tensor> &wt) {
  return LibConvFn::apply(x, weight, stride, pad, dil, dgrad_hip, wt);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "C++ autograd front-ends over libbevfusion_hip.so (include/bevfusion_hip.h)";
  m.def("bn2d", &bn2d, "fused BatchNorm2d(+residual)(+ReLU), channels-last, training mode; optional statistics partials of the producing conv",
        pybind11::arg("x"), pybind11::arg("residual"), pybind11::arg("weight"), pybind11::arg("bias"), pybind11::arg("running_mean"),
        pybind11::arg("running_var"), pybind11::arg("eps"), pybind11::arg("momentum"), pybind11::arg("relu"),
        pybind11::arg("partial") = c10::optional<Tensor>(), pybind11::arg("relu_bits") = false);
  m.def("bn1d", &bn1d, "fused BatchNorm1d(+residual)(+ReLU) on f32[N, C], training mode");
  m.def("conv2d", &conv2d, "dense convolution on csrc/conv2d.hip (conv2d.py::_Conv2dFunction): [y, partial (empty without statistics)[, x']]",
        pybind11::arg("x"), pybind11::arg("weight"), pybind11::arg("bias"), pybind11::arg("stride"), pybind11::arg("pad"), pybind11::arg("dil"),
        pybind11::arg("emit_stats"), pybind11::arg("dgrad_lib"), pybind11::arg("fork"), pybind11::arg("wt"));
  m.def("lib_conv2d", &lib_conv2d, "library forward, HIP weight gradient (conv2d.py::_LibConvHipWgradFunction)", pybind11::arg("x"),
        pybind11::arg("weight"), pybind11::arg("stride"), pybind11::arg("pad"), pybind11::arg("dil"), pybind11::arg("dgrad_hip"), pybind11::arg("wt"));
  m.def("defer_wgrad", &defer_wgrad, "queue a layer's weight gradient for the grouped launch at the end of the running backward pass");
  m.def("set_wgrad_grouped", [](bool on) { g_grouped = on; });
  m.def("pending_wgrads", &pending_wgrads);
  m.def("abi_version", []() { return bfhip_abi_version(); });
}
