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
