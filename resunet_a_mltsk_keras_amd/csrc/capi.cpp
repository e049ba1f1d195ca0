// C-ABI plumbing: error text, version, device query.  (Kernels live in the .hip files.)
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include "../../include/rua_hip.h"

static thread_local char g_err[512] = "";

void rua_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* rua_last_error(void) { return g_err; }
extern "C" int rua_version(void) { return 100; }

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
