// capi.hip -- error reporting, ABI version and the optional per-op HIP-event profiler.
#include "common.h"

#include <mutex>
#include <vector>

namespace bfhip {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- profiler: HIP events recorded on the SAME stream as the kernel, around the dominant
// kernel of each op only (not around memsets / helper launches).
static int g_prof_on = 0;
struct Pending { hipEvent_t a, b; };
static std::vector<Pending> g_pending[BFHIP_OP_COUNT];
static double g_sum_ms[BFHIP_OP_COUNT];
static long long g_count[BFHIP_OP_COUNT];
static std::mutex g_mu;

bool prof_enabled() { return g_prof_on != 0; }

void prof_begin(int op, hipStream_t s, ProfScope *sc) {
  sc->active = false;
  if (!g_prof_on || op < 0 || op >= BFHIP_OP_COUNT) return;
  if (op >= BFHIP_OP_CONV2D_FWD && g_prof_on < 2) return;  // dense ops: level 2 only
  if (hipEventCreate(&sc->a) != hipSuccess) return;
  if (hipEventCreate(&sc->b) != hipSuccess) { hipEventDestroy(sc->a); return; }
  hipEventRecord(sc->a, s);
  sc->active = true;
  sc->op = op;
  sc->stream = s;
}

void prof_end(ProfScope *sc) {
  if (!sc->active) return;
  hipEventRecord(sc->b, sc->stream);
  std::lock_guard<std::mutex> lk(g_mu);
  g_pending[sc->op].push_back({sc->a, sc->b});
}
}  // namespace bfhip

using namespace bfhip;

BFHIP_EXPORT int bfhip_abi_version(void) { return 2; }
BFHIP_EXPORT const char *bfhip_last_error(void) { return bfhip::g_err; }

BFHIP_EXPORT void bfhip_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof_on = on;
}

BFHIP_EXPORT int bfhip_profile_read(int op, double *sum_ms, long long *count, int reset) {
  BFHIP_REQUIRE(op >= 0 && op < BFHIP_OP_COUNT, "profile_read: bad op %d", op);
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto &p : g_pending[op]) {
    float ms = 0.f;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      g_sum_ms[op] += ms;
      g_count[op] += 1;
    }
    hipEventDestroy(p.a);
    hipEventDestroy(p.b);
  }
  g_pending[op].clear();
  if (sum_ms) *sum_ms = g_sum_ms[op];
  if (count) *count = g_count[op];
  if (reset) { g_sum_ms[op] = 0.0; g_count[op] = 0; }
  return BFHIP_OK;
}
