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
extern "C" int pp_abi_version(void) { return PP_ABI_VERSION; }

// ------------------------------------------------------------------------------------------------ options / context
struct PPOptionDef { const char* name; int dflt, lo, hi; };
static const PPOptionDef g_opt_def[PP_OPT_COUNT] = {
    {"mlp_fused", 1, 0, 1},        {"wgrad_split", 0, 0, 1},         {"grid_chunks", 0, 0, 4096},
    {"nerf_split", 1, 0, 1},       {"nerf_split_tn", 1, 0, 1},       {"nerf_bitmask", 1, 0, 1},
    {"nerf_gemm_wgs", 256, 1, 4096}, {"nerf_tn_ch", 64, 32, 64},     {"nerf_tn_split_wgs", 128, 1, 4096},
    {"nerf_tn_wgs", 128, 1, 4096}, {"nerf_bn", 128, 128, 256},       {"nerf_planes", 1, 0, 1},
    {"mlp_split", 31, 0, 31},
    {"nerf_tn256", 0, 0, 1},       {"mlp_wgs", 0, 0, 4096},
    {"wgrad_side_wgs", 0, 0, 4096}, {"side_stream", 0, 0, 2},
    {"nerf_chain", 3, 0, 3},       {"nerf_chain_nw", 4, 4, 8},
    {"nerf_chain_head", 1, 0, 1},
    {"nerf_tn_tr", 1, 0, 1},
};
// compiled-in defaults: constants, never written after static initialisation
static const struct PPDefaults {
  int v[PP_OPT_COUNT];
  PPDefaults() { for (int i = 0; i < PP_OPT_COUNT; ++i) v[i] = g_opt_def[i].dflt; }
} g_defaults;

static thread_local const int* t_opts = nullptr;      // options of the entry point running on this thread (PPOptScope)
PPOptScope::PPOptScope(const void* ctx) : prev(t_opts) { t_opts = ctx ? static_cast<const PPContext*>(ctx)->opt : g_defaults.v; }
PPOptScope::~PPOptScope() { t_opts = prev; }
int pp_opt(int id) { return (t_opts ? t_opts : g_defaults.v)[id]; }

int pp_num_cus() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
        cus > 0)
      return cus;
    return 256;
  }();
  return n;
}

static int opt_find(const char* name) {
  for (int i = 0; i < PP_OPT_COUNT; ++i) if (name && strcmp(name, g_opt_def[i].name) == 0) return i;
  return -1;
}

extern "C" int pp_context_create(void** ctx) {
  PP_REQUIRE(ctx, "null pointer");
  PPContext* c = new PPContext;
  for (int i = 0; i < PP_OPT_COUNT; ++i) c->opt[i] = g_opt_def[i].dflt;
  c->have_aux = false;
  c->pending = 0;
  *ctx = c;
  return PP_OK;
}

bool pp_context_aux(PPContext* c) {
  if (c->have_aux) return true;
  if (hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking) != hipSuccess) return false;
  for (int i = 0; i < 16; ++i) {
    hipEventCreateWithFlags(&c->fork[i], hipEventDisableTiming);
    hipEventCreateWithFlags(&c->join[i], hipEventDisableTiming);
  }
  for (int i = 0; i < 4; ++i) {
    hipEventCreateWithFlags(&c->dfork[i], hipEventDisableTiming);
    hipEventCreateWithFlags(&c->djoin[i], hipEventDisableTiming);
  }
  c->have_aux = true;
  return true;
}

extern "C" int pp_context_destroy(void* ctx) {
  if (!ctx) return PP_OK;
  PPContext* c = static_cast<PPContext*>(ctx);
  if (c->have_aux) {
    hipStreamSynchronize(c->aux);
    for (int i = 0; i < 16; ++i) { hipEventDestroy(c->fork[i]); hipEventDestroy(c->join[i]); }
    for (int i = 0; i < 4; ++i) { hipEventDestroy(c->dfork[i]); hipEventDestroy(c->djoin[i]); }
    hipStreamDestroy(c->aux);
  }
  delete c;
  return PP_OK;
}

extern "C" int pp_context_set_option(void* ctx, const char* name, int32_t value) {
  PP_REQUIRE(ctx, "null context (the compiled-in defaults cannot be changed: create a context)");
  const int i = opt_find(name);
  if (i < 0) { pp_set_error("pp_context_set_option: unknown option '%s'", name ? name : "(null)"); return PP_ERR_INVALID_ARG; }
  if (value < g_opt_def[i].lo || value > g_opt_def[i].hi) {
    pp_set_error("pp_context_set_option: %s = %d outside [%d, %d]", name, value, g_opt_def[i].lo, g_opt_def[i].hi);
    return PP_ERR_INVALID_ARG;
  }
  static_cast<PPContext*>(ctx)->opt[i] = value;
  return PP_OK;
}

extern "C" int pp_context_get_option(const void* ctx, const char* name, int32_t* value) {
  const int i = opt_find(name);
  if (i < 0 || !value) { pp_set_error("pp_context_get_option: unknown option '%s'", name ? name : "(null)"); return PP_ERR_INVALID_ARG; }
  *value = ctx ? static_cast<const PPContext*>(ctx)->opt[i] : g_opt_def[i].dflt;
  return PP_OK;
}
