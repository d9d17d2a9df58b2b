// thread-local error text, ABI version, tuning options
#include <stdarg.h>

#include "pp_common.h"

static thread_local char g_err[512] = "";

void pp_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* pp_last_error(void) { return g_err; }
extern "C" int pp_abi_version(void) { return 1; }

// ------------------------------------------------------------------------------------------------ options
#include <atomic>
struct PPOptionDef { const char* name; int dflt, lo, hi; };
static const PPOptionDef g_opt_def[PP_OPT_COUNT] = {
    {"mlp_fused", 1, 0, 1},        {"wgrad_split", 0, 0, 1},         {"grid_chunks", 0, 0, 4096},
    {"nerf_split", 1, 0, 1},       {"nerf_split_tn", 1, 0, 1},       {"nerf_bitmask", 1, 0, 1},
    {"nerf_gemm_wgs", 256, 1, 4096}, {"nerf_tn_ch", 64, 32, 64},     {"nerf_tn_split_wgs", 128, 1, 4096},
    {"nerf_tn_wgs", 128, 1, 4096}, {"nerf_bn", 128, 128, 256},       {"nerf_planes", 1, 0, 1},
    {"sdf_index_exact", 0, 0, 1},   {"mlp_split", 31, 0, 31},
    {"nerf_tn256", 0, 0, 1},       {"mlp_wgs", 0, 0, 4096},
    {"wgrad_side_wgs", 0, 0, 4096},
};
static std::atomic<int> g_opt[PP_OPT_COUNT];
static std::atomic<bool> g_opt_init{false};
static void opt_init() {
  if (g_opt_init.load(std::memory_order_acquire)) return;
  static std::atomic<bool> busy{false};
  bool expected = false;
  if (busy.compare_exchange_strong(expected, true)) {
    for (int i = 0; i < PP_OPT_COUNT; ++i) g_opt[i].store(g_opt_def[i].dflt);
    g_opt_init.store(true, std::memory_order_release);
  } else {
    while (!g_opt_init.load(std::memory_order_acquire)) {}
  }
}
int pp_opt(int id) { opt_init(); return g_opt[id].load(std::memory_order_relaxed); }
static int opt_find(const char* name) {
  for (int i = 0; i < PP_OPT_COUNT; ++i) if (name && strcmp(name, g_opt_def[i].name) == 0) return i;
  return -1;
}
extern "C" int pp_set_option(const char* name, int32_t value) {
  opt_init();
  const int i = opt_find(name);
  if (i < 0) { pp_set_error("pp_set_option: unknown option '%s'", name ? name : "(null)"); return PP_ERR_INVALID_ARG; }
  if (value < g_opt_def[i].lo || value > g_opt_def[i].hi) {
    pp_set_error("pp_set_option: %s = %d outside [%d, %d]", name, value, g_opt_def[i].lo, g_opt_def[i].hi);
    return PP_ERR_INVALID_ARG;
  }
  g_opt[i].store(value);
  return PP_OK;
}
extern "C" int pp_get_option(const char* name, int32_t* value) {
  opt_init();
  const int i = opt_find(name);
  if (i < 0 || !value) { pp_set_error("pp_get_option: unknown option '%s'", name ? name : "(null)"); return PP_ERR_INVALID_ARG; }
  *value = g_opt[i].load();
  return PP_OK;
}
