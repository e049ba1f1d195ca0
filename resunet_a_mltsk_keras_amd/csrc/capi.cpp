// C-ABI plumbing: error text, version, device query.  (Kernels live in the .hip files.)
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <dlfcn.h>
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

int rua_cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    cached[dev] = n;
  }
  return cached[dev];
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
    {"head_blocks", &g_tune.head_blocks, nullptr}, {"conv_strip", &g_tune.conv_strip, nullptr}, {"wgrad_slabs", &g_tune.wgrad_slabs, nullptr}, {"strip_narrow_maxd", &g_tune.strip_narrow_maxd, nullptr}, {"conv_group", &g_tune.conv_group, nullptr}, {"wgrad_group", &g_tune.wgrad_group, nullptr}, {"wgd_ks_slow", &g_tune.wgd_ks_slow, nullptr}, {"head_fwd2", &g_tune.head_fwd2, nullptr}, {"conv_band", &g_tune.conv_band, nullptr}, {"conv_band64", &g_tune.conv_band64, nullptr}, {"conv_band64m", &g_tune.conv_band64m, nullptr}, {"strip_group_share", &g_tune.strip_group_share, nullptr}, {"band_dbg", &g_tune.band_dbg, nullptr}, {"fill_kernel", &g_tune.fill_kernel, nullptr}, {"bn_regs", &g_tune.bn_regs, nullptr}, {"wgrad_taps_share", &g_tune.wgrad_taps_share, nullptr}, {"wgrad_kernel_share", &g_tune.wgrad_kernel_share, nullptr}, {"dmap_group_bm128", &g_tune.dmap_group_bm128, nullptr}, {"dmap_chain", &g_tune.dmap_chain, nullptr}, {"epi_fast", &g_tune.epi_fast, nullptr}, {"dmap_spread", &g_tune.dmap_spread, nullptr}, {"bn_bwd_group", &g_tune.bn_bwd_group, nullptr}, {"conv_small", &g_tune.conv_small, nullptr}, {"strip_stag", &g_tune.strip_stag, nullptr}, {"strip_seglen", &g_tune.strip_seglen, nullptr},
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

// ---- data parallel: thin RCCL entry points (SURVEY 8b "rua_allreduce_bucket").  RCCL is bound at first use with dlopen, so
// the library has no link-time dependency on it (a host that already carries an RCCL - PyTorch does - keeps using its own
// copy: dlopen by soname returns the loaded one).  Replaces the implicit NCCL all-reduce of tf.distribute.MirroredStrategy
// (train_ISPRS.py:347,432).  The communicator handle is the caller's; nothing is cached here.
namespace {
typedef struct { char internal[128]; } rccl_uid;
typedef int (*fn_uid)(rccl_uid*);
typedef int (*fn_init)(void**, int, rccl_uid, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*fn_errstr)(int);
struct Rccl { void* h = nullptr; fn_uid uid = nullptr; fn_init init = nullptr; fn_destroy destroy = nullptr; fn_allreduce allreduce = nullptr; fn_errstr errstr = nullptr; };
Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
    if (r.h) {
      r.uid = (fn_uid)dlsym(r.h, "ncclGetUniqueId"); r.init = (fn_init)dlsym(r.h, "ncclCommInitRank");
      r.destroy = (fn_destroy)dlsym(r.h, "ncclCommDestroy"); r.allreduce = (fn_allreduce)dlsym(r.h, "ncclAllReduce");
      r.errstr = (fn_errstr)dlsym(r.h, "ncclGetErrorString");
      if (!r.uid || !r.init || !r.destroy || !r.allreduce) { dlclose(r.h); r.h = nullptr; }
    }
  }
  return r.h ? &r : nullptr;
}
int rccl_fail(const char* what, int rc) {
  Rccl* r = rccl();
  rua_set_error("%s: RCCL error %d (%s)", what, rc, (r && r->errstr) ? r->errstr(rc) : "?");
  return RUA_ERR_LAUNCH;
}
}  // namespace

extern "C" int rua_comm_unique_id(void* id128) {
  Rccl* r = rccl();
  RUA_CHECK_ARG(id128, "rua_comm_unique_id: null buffer");
  if (!r) { rua_set_error("rua_comm_unique_id: librccl.so not found"); return RUA_ERR_LAUNCH; }
  rccl_uid u;
  const int rc = r->uid(&u);
  if (rc != 0) return rccl_fail("rua_comm_unique_id", rc);
  memcpy(id128, u.internal, 128);
  return RUA_OK;
}
extern "C" int rua_comm_init(void** comm, int world, int rank, const void* id128) {
  Rccl* r = rccl();
  RUA_CHECK_ARG(comm && id128 && world >= 1 && rank >= 0 && rank < world, "rua_comm_init: bad arguments (world %d, rank %d)", world, rank);
  if (!r) { rua_set_error("rua_comm_init: librccl.so not found"); return RUA_ERR_LAUNCH; }
  rccl_uid u;
  memcpy(u.internal, id128, 128);
  const int rc = r->init(comm, world, u, rank);
  return rc == 0 ? RUA_OK : rccl_fail("rua_comm_init", rc);
}
extern "C" int rua_comm_destroy(void* comm) {
  Rccl* r = rccl();
  if (!r || !comm) return RUA_OK;
  const int rc = r->destroy(comm);
  return rc == 0 ? RUA_OK : rccl_fail("rua_comm_destroy", rc);
}
extern "C" int rua_allreduce_bucket(void* comm, float* grads, int64_t count, void* stream) {
  Rccl* r = rccl();
  RUA_CHECK_ARG(comm && grads && count > 0, "rua_allreduce_bucket: bad arguments");
  if (!r) { rua_set_error("rua_allreduce_bucket: librccl.so not found"); return RUA_ERR_LAUNCH; }
  const int rc = r->allreduce(grads, grads, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, (hipStream_t)stream);
  return rc == 0 ? RUA_OK : rccl_fail("rua_allreduce_bucket", rc);
}
