// Library identity and the option table (the library's process-wide switches: include/myriad_hip.h, "options").
#include "common.h"
#include <stdlib.h>
#include <string.h>
extern "C" const char* mh_version(void) { return "myriad_hip 0.1 (gfx950)"; }
extern "C" int mh_target_arch(void) { return 950; }

struct MhOptEntry { const char* name; const char* env; int def; int value; };
static MhOptEntry g_opts[MH_OPT_COUNT] = {
    {"slab_bf16", "MYRIAD_SLAB_BF16", 1, -1},
    {"gemm_skinny", "MYRIAD_GEMM_SKINNY", 1, -1},
    {"swiglu_fused", "MYRIAD_SWIGLU_FUSED", 1, -1},
    {"gelu_fused", "MYRIAD_GELU_FUSED", 1, -1},
    {"attn_bwd_split", "MYRIAD_ATTN_BWD_SPLIT", 1, -1},
    {"gemm_zero_pad", "MYRIAD_GEMM_ZERO_PAD", 1, -1},
    {"gemm256_impl", "MYRIAD_GEMM256_IMPL", 1, -1},
    {"lora_norm_fused", "MYRIAD_LORA_NORM_FUSED", 1, -1},
    {"attn_full", "MYRIAD_ATTN_FULL", 1, -1},
    {"lora_wgrad_mfma", "MYRIAD_LORA_WGRAD_MFMA", 1, -1},
    {"gemm_skip_pad", "MYRIAD_GEMM_SKIP_PAD", 1, -1},
    {"gemm_split_xcd", "MYRIAD_GEMM_SPLIT_XCD", 1, -1},
};

int mh_opt(int id) {
  MhOptEntry& o = g_opts[id];
  if (o.value < 0) {
    const char* e = getenv(o.env);
    o.value = e ? (e[0] == '0' ? 0 : 1) : o.def;
  }
  return o.value;
}

static int find_opt(const char* name) {
  if (!name) return -1;
  for (int i = 0; i < MH_OPT_COUNT; ++i)
    if (!strcmp(name, g_opts[i].name)) return i;
  return -1;
}
extern "C" int mh_get_option(const char* name) {
  const int i = find_opt(name);
  return i < 0 ? MH_ERR_ARG : mh_opt(i);
}
// returns the previous value (>= 0) or MH_ERR_ARG for an unknown name / a value other than 0 or 1
extern "C" int mh_set_option(const char* name, int value) {
  const int i = find_opt(name);
  if (i < 0 || (value != 0 && value != 1)) return MH_ERR_ARG;
  const int prev = mh_opt(i);
  g_opts[i].value = value;
  return prev;
}
