// Library identity + error strings for the C ABI declared in include/pocketflow_hip.h.
#include "pf_common.h"

extern "C" int pf_version(void) { return 100; }   // 0.1.0 (round 1)

extern "C" const char* pf_error_string(int err) {
  if (err == 0) return "ok";
  return hipGetErrorString((hipError_t)err);
}

// ---- tuning switches: one read of the environment (pf_common.h) -------------------------------------------------------------------
#include <stdlib.h>
#include <stdio.h>
static PfTuning g_tuning;
static bool g_tuning_loaded = false;

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e == nullptr || *e == 0) ? dflt : atoi(e);
}

static void tuning_load() {
  PfTuning t;
  t.conv_bn = env_int("PF_CONV_BN", 0);
  t.conv_igemm = env_int("PF_CONV_IGEMM", 1);
  t.conv_igemm_pro = env_int("PF_CONV_IGEMM_PRO", 1);
  t.conv_stream = env_int("PF_CONV_STREAM", 1);
  t.conv_stream_maxsplit = env_int("PF_CONV_STREAM_MAXSPLIT", 2);
  t.igemm_pro3 = env_int("PF_IGEMM_PRO3", 1);
  t.igemm_tile_bm = t.igemm_tile_bn = 0;
  const char* e = getenv("PF_IGEMM_TILE");
  if (e != nullptr) {
    int bm = 0, bn = 0;
    if (sscanf(e, "%dx%d", &bm, &bn) == 2) { t.igemm_tile_bm = bm; t.igemm_tile_bn = bn; }
  }
  t.pool3s2 = env_int("PF_POOL3S2", 1);
  t.wrw_tr = env_int("PF_WRW_TR", 0);
  t.wrw2 = env_int("PF_WRW2", 1);
  t.wrw2_target = env_int("PF_WRW2_TARGET", 0);
  t.splitk = env_int("PF_IGEMM_SPLITK", 1);
  t.wrw2_tk256 = env_int("PF_WRW2_TK256", 1);
  t.conv3x3_c64 = env_int("PF_CONV3X3_C64", 1);
  g_tuning = t;
  g_tuning_loaded = true;
}

const PfTuning& pf_tuning() {
  if (!g_tuning_loaded) tuning_load();
  return g_tuning;
}

extern "C" int pf_tuning_reload(void) { tuning_load(); return 0; }

// ---- per-(device, kernel) dynamic-LDS ceiling (declared in pf_common.h) --------------------------------------------------------
#include <map>
#include <mutex>
#include <utility>
int pf_require_lds(const void* fn, size_t lds) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, size_t> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = done[std::make_pair(dev, fn)];
  if (lds > have) {
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    have = lds;
  }
  return 0;
}
