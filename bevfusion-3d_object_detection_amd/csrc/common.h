// common.h -- shared helpers of libbevfusion_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/bevfusion_hip.h"

#define BFHIP_EXPORT extern "C" __attribute__((visibility("default")))

namespace bfhip {

constexpr int kWave = 64;  // CDNA wavefront

void set_error(const char *fmt, ...);

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return BFHIP_E_LAUNCH;
  }
  return BFHIP_OK;
}

// optional profiler (capi.hip): events on the op's stream around its dominant kernel
struct ProfScope { bool active; int op; hipStream_t stream; hipEvent_t a, b; };
bool prof_enabled();
void prof_begin(int op, hipStream_t s, ProfScope *sc);
void prof_end(ProfScope *sc);

inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// bump allocator over the caller-provided workspace
struct Workspace {
  char *base;
  size_t size, off;
  Workspace(void *p, size_t n) : base((char *)p), size(n), off(0) {}
  template <typename T>
  T *take(size_t count) {
    size_t bytes = align_up(count * sizeof(T), 256);
    if (off + bytes > size) { off = size + 1; return nullptr; }
    T *r = (T *)(base + off);
    off += bytes;
    return r;
  }
  bool ok() const { return off <= size; }
};

// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Remap so that each
// XCD walks one CONTIGUOUS eighth of the work: consecutive intervals (adjacent BEV cells) gather
// neighbouring feature rows, and consecutive pixels gather neighbouring out_grad rows, so each XCD's
// 4 MiB L2 then holds its own slice of the gathered table.  Bijective for any grid size; speed only.
__device__ __forceinline__ long long xcd_chunked_block(long long bid, long long nblocks) {
  const long long per = (nblocks + 7) / 8;      // blocks per XCD slice (last slices may be short)
  const long long xcd = bid & 7, slot = bid >> 3;
  const long long full = nblocks - (per - 1) * 8;  // number of slices that hold `per` blocks (1..8)
  // slices [0, full) have `per` blocks, the rest have per-1
  long long base = xcd < full ? xcd * per : full * per + (xcd - full) * (per - 1);
  return base + slot;
}

// sparse weight gradient on the bf16 matrix cores (csrc/conv2d.hip: the transposing-LDS-read machinery of the dense conv
// weight gradient with the rulebook as the gather); called by bfhip_spconv_wgrad for bf16 features
size_t spconv_wgrad_tr_workspace_bytes(int KV, int Cin, int Cout, int n_rows);
bool spconv_wgrad_tr_supported(int KV, int Cin, int Cout);
int spconv_wgrad_tr(const void *in, const void *dout, const int32_t *pairs, int ld, int KV, int n_rows, int Cin, int Cout,
                    float *dW, void *workspace, size_t workspace_bytes, hipStream_t stream);

}  // namespace bfhip

#define BFHIP_REQUIRE(cond, ...)                \
  do {                                          \
    if (!(cond)) {                              \
      bfhip::set_error(__VA_ARGS__);            \
      return BFHIP_E_INVALID;                   \
    }                                           \
  } while (0)
