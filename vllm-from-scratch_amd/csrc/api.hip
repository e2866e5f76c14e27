// Error reporting and version entry points of libmi355_paged.so.
#include <limits.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "common.h"

namespace mi355 {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  return fail((int)e, "%s: %s", what, hipGetErrorString(e));
}

namespace {
struct KnobDef {
  const char* name;
  int dflt;
};
const KnobDef kKnobDefs[kKnobCount] = {
    {"MI355_DECODE_GQA_MFMA", 1},   {"MI355_PAGED_TARGET_WGS", 256}, {"MI355_PAGED_HPW", 0},
    {"MI355_PAGED_PREFILL_MIN_Q", 48}, {"MI355_PREFILL_VALU", 0},    {"MI355_PREFILL_F32_TILED", 1},
    {"MI355_PREFILL_ROW_GROUPS", 1}, {"MI355_PREFILL_WAVES", 0},     {"MI355_MLA_MANY_HEADS", 1},
    {"MI355_MLA_TARGET_WGS", 0},      {"MI355_PREFILL_F32_MFMA", 1},    {"MI355_DECODE_FUSED_MERGE", 8},
    {"MI355_PAGED_BALANCE_WGS", 2048}, {"MI355_DP_EXCHANGE_PROBE", 0},  {"MI355_MLA_ROW_GROUPS", 1},
    {"MI355_MLA_WIDE", 17},          {"MI355_MLA_WIDE_SPLIT_TOKENS", 0}, {"MI355_MLA_WIDE_FUSED_MERGE", 1},
    {"MI355_MLA_WIDE_SKEW_TOKENS", 96},
};
std::atomic<int> g_knob[kKnobCount];
int g_knob_start[kKnobCount];   // the process's values: defaults overridden by the environment
std::once_flag g_knob_once;

void load_knobs() {
  std::call_once(g_knob_once, [] {
    for (int i = 0; i < kKnobCount; ++i) {
      const char* s = getenv(kKnobDefs[i].name);
      g_knob_start[i] = (s && *s) ? atoi(s) : kKnobDefs[i].dflt;
      g_knob[i].store(g_knob_start[i], std::memory_order_relaxed);
    }
  });
}
}  // namespace

int knob(Knob k) {
  load_knobs();
  return g_knob[k].load(std::memory_order_relaxed);
}

}  // namespace mi355

extern "C" int mi355_debug_set_knob(const char* name, int32_t value) {
  using namespace mi355;
  if (!name) return fail(MI355_EINVAL, "debug_set_knob: null name");
  load_knobs();
  for (int i = 0; i < kKnobCount; ++i) {
    if (strcmp(name, kKnobDefs[i].name) == 0) {
      g_knob[i].store(value == INT32_MIN ? g_knob_start[i] : value, std::memory_order_relaxed);
      return 0;
    }
  }
  return fail(MI355_EINVAL, "debug_set_knob: unknown knob '%s'", name);
}

extern "C" int mi355_abi_version(void) { return MI355_ABI_VERSION; }

extern "C" const char* mi355_last_error(void) { return mi355::g_err; }
