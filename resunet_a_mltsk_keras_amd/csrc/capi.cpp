// C-ABI plumbing: error text, version, device query.  (Kernels live in the .hip files.)
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdint.h>
#include "common.h"

static thread_local char g_err[512] = "";
RuaTuning g_tune;

void rua_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* rua_last_error(void) { return g_err; }
extern "C" int rua_version(void) { return 200; }

extern "C" int rua_device_info(int* cu_count, int* lds_bytes, char* arch, int arch_len) {
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    rua_set_error("rua_device_info: no HIP device");
    return RUA_ERR_LAUNCH;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
  if (arch && arch_len > 0) { strncpy(arch, prop.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
  return RUA_OK;
}

// Timing events for bench.py's per-kernel roofline.  Created with hipEventDisableSystemFence | hipEventReleaseToDevice:
// a default event performs a system-scope release when it is recorded (L2 write-back of everything the kernel before it
// wrote), which lands inside the bracket of a kernel with a large output and is not part of the kernel's own duration.
extern "C" void* rua_prof_event_create(void) {
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence | hipEventReleaseToDevice) != hipSuccess) {
    (void)hipGetLastError();
    if (hipEventCreate(&e) != hipSuccess) { rua_set_error("rua_prof_event_create: hipEventCreate failed"); return nullptr; }
  }
  return e;
}
extern "C" int rua_prof_event_record(void* ev, void* stream) {
  return hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) == hipSuccess ? RUA_OK : RUA_ERR_LAUNCH;
}
extern "C" int rua_prof_event_elapsed_us(void* start, void* stop, double* us) {
  float ms = 0.f;
  if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess || hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) {
    rua_set_error("rua_prof_event_elapsed_us: %s", hipGetErrorString(hipGetLastError()));
    return RUA_ERR_LAUNCH;
  }
  *us = 1e3 * (double)ms;
  return RUA_OK;
}
extern "C" void rua_prof_event_destroy(void* ev) { (void)hipEventDestroy((hipEvent_t)ev); }

// Kernel nodes of a captured HIP graph (bench.py: dispatches per step of the whole-step graph, counted instead of read off a profile).
extern "C" int rua_graph_kernel_nodes(void* graph, int* kernels, int* total) {
  size_t n = 0;
  if (!graph || hipGraphGetNodes((hipGraph_t)graph, nullptr, &n) != hipSuccess) { (void)hipGetLastError(); rua_set_error("rua_graph_kernel_nodes: hipGraphGetNodes failed"); return RUA_ERR_ARG; }
  hipGraphNode_t* nodes = n ? new hipGraphNode_t[n] : nullptr;
  int k = 0;
  if (n && hipGraphGetNodes((hipGraph_t)graph, nodes, &n) == hipSuccess)
    for (size_t i = 0; i < n; ++i) {
      hipGraphNodeType t;
      if (hipGraphNodeGetType(nodes[i], &t) == hipSuccess && t == hipGraphNodeTypeKernel) ++k;
    }
  delete[] nodes;
  if (kernels) *kernels = k;
  if (total) *total = (int)n;
  return RUA_OK;
}

int rua_device_index() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) { (void)hipGetLastError(); dev = 0; }
  return dev;
}

// Compute units the launchers size their one-round grids by.  Tuning key cu_reserve (set by dist.DataParallel for world > 1): that many
// CUs are left to the RCCL kernels of a gradient bucket in flight - a grid of exactly one block per CU would push its last blocks into
// a second round whenever a collective's workgroups hold some CUs.
int rua_cu_count() {
  static int cached[64] = {0};
  const int dev = rua_device_index();
  if (dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    cached[dev] = n;
  }
  int n = cached[dev] - (g_tune.cu_reserve > 0 ? g_tune.cu_reserve : 0);
  return n < 16 ? 16 : n;
}

// ---- tuning: the only way a launch heuristic changes (no environment reads anywhere in the library) ---------------------
struct TuneKey { const char* name; int* i; long long* ll; };
static const TuneKey* tune_table(int* n) {
  static const TuneKey t[] = {
    {"conv_force_bn", &g_tune.conv_force_bn, nullptr}, {"conv_force_bm", &g_tune.conv_force_bm, nullptr}, {"conv_dma", &g_tune.conv_dma, nullptr},
    {"conv_pw", &g_tune.conv_pw, nullptr}, {"conv_pw_minm", nullptr, &g_tune.conv_pw_minm}, {"conv_pw_blocks", &g_tune.conv_pw_blocks, nullptr},
    {"conv_halo", &g_tune.conv_halo, nullptr}, {"halo64_maxd", &g_tune.halo64_maxd, nullptr}, {"conv_dmap", &g_tune.conv_dmap, nullptr},
    {"dmap_target", &g_tune.dmap_target, nullptr}, {"dmap_fused_finish", &g_tune.dmap_fused_finish, nullptr}, {"dmap_rowb", &g_tune.dmap_rowb, nullptr},
    {"dmap_bm64", &g_tune.dmap_bm64, nullptr}, {"wgrad_pw", &g_tune.wgrad_pw, nullptr}, {"wgpw_blocks", &g_tune.wgpw_blocks, nullptr},
    {"wgpw_r", &g_tune.wgpw_r, nullptr}, {"wgd_blocks", &g_tune.wgd_blocks, nullptr}, {"wgrad_dmap", &g_tune.wgrad_dmap, nullptr},
    {"wgd_mintiles", &g_tune.wgd_mintiles, nullptr}, {"wgrad_blocks", &g_tune.wgrad_blocks, nullptr}, {"bn_grid", &g_tune.bn_grid, nullptr},
    {"tani_vec", &g_tune.tani_vec, nullptr}, {"metrics_blocks", &g_tune.metrics_blocks, nullptr}, {"stem_blocks", &g_tune.stem_blocks, nullptr},
    {"head_blocks", &g_tune.head_blocks, nullptr}, {"conv_img", &g_tune.conv_img, nullptr}, {"conv_strip", &g_tune.conv_strip, nullptr}, {"wgrad_slabs", &g_tune.wgrad_slabs, nullptr}, {"strip_narrow_maxd", &g_tune.strip_narrow_maxd, nullptr}, {"conv_group", &g_tune.conv_group, nullptr}, {"wgrad_group", &g_tune.wgrad_group, nullptr}, {"wgd_ks_slow", &g_tune.wgd_ks_slow, nullptr}, {"head_fwd2", &g_tune.head_fwd2, nullptr}, {"head_fwd3", &g_tune.head_fwd3, nullptr}, {"head_fwd3_bpc", &g_tune.head_fwd3_bpc, nullptr}, {"stem_reg", &g_tune.stem_reg, nullptr}, {"stats_blocks", &g_tune.stats_blocks, nullptr}, {"conv_band", &g_tune.conv_band, nullptr}, {"conv_band64", &g_tune.conv_band64, nullptr}, {"conv_band64m", &g_tune.conv_band64m, nullptr}, {"strip_group_share", &g_tune.strip_group_share, nullptr}, {"band_dbg", &g_tune.band_dbg, nullptr}, {"fill_kernel", &g_tune.fill_kernel, nullptr}, {"bn_regs", &g_tune.bn_regs, nullptr}, {"wgrad_taps_share", &g_tune.wgrad_taps_share, nullptr}, {"wgrad_kernel_share", &g_tune.wgrad_kernel_share, nullptr}, {"dmap_group_bm128", &g_tune.dmap_group_bm128, nullptr}, {"dmap_chain", &g_tune.dmap_chain, nullptr}, {"epi_fast", &g_tune.epi_fast, nullptr}, {"dmap_spread", &g_tune.dmap_spread, nullptr}, {"bn_bwd_group", &g_tune.bn_bwd_group, nullptr}, {"conv_small", &g_tune.conv_small, nullptr}, {"strip_stag", &g_tune.strip_stag, nullptr}, {"cu_reserve", &g_tune.cu_reserve, nullptr}, {"wgrad_rows", &g_tune.wgrad_rows, nullptr}, {"strip_seglen", &g_tune.strip_seglen, nullptr}, {"band_stag", &g_tune.band_stag, nullptr}, {"conv_band128m", &g_tune.conv_band128m, nullptr}, {"conv_img2", &g_tune.conv_img2, nullptr}, {"dbg_ptr", nullptr, &g_tune.dbg_ptr},
  };
  *n = (int)(sizeof(t) / sizeof(t[0]));
  return t;
}
extern "C" int rua_set_tuning(const char* key, int64_t value) {
  int n; const TuneKey* t = tune_table(&n);
  for (int i = 0; key && i < n; ++i)
    if (strcmp(t[i].name, key) == 0) { if (t[i].i) *t[i].i = (int)value; else *t[i].ll = (long long)value; return RUA_OK; }
  rua_set_error("rua_set_tuning: unknown key '%s'", key ? key : "(null)");
  return RUA_ERR_ARG;
}
extern "C" int rua_get_tuning(const char* key, int64_t* value) {
  int n; const TuneKey* t = tune_table(&n);
  for (int i = 0; key && value && i < n; ++i)
    if (strcmp(t[i].name, key) == 0) { *value = t[i].i ? (int64_t)*t[i].i : (int64_t)*t[i].ll; return RUA_OK; }
  rua_set_error("rua_get_tuning: unknown key '%s'", key ? key : "(null)");
  return RUA_ERR_ARG;
}
extern "C" const char* rua_tuning_key(int index) {
  int n; const TuneKey* t = tune_table(&n);
  return (index >= 0 && index < n) ? t[index].name : nullptr;
}

// ---- data parallel: the library exports NO collective (round 4).  The engine's one path for the gradient all-reduce is
// torch.distributed's RCCL communicator (dist.py: buckets issued from the launch stream, run on the process group's stream); the
// thin rua_comm_* / rua_allreduce_bucket wrappers of rounds 1-3 were reached by nothing but a one-rank test - a second, unexercised
// communicator beside the first is what a first multi-GPU run does not need.  A C embedder calls ncclAllReduce itself on the
// contiguous slices INTEGRATION.md section 2 describes.
