// Error channel and device probe of the C-ABI (include/smml.h).  No exceptions cross the boundary:
// every entry point returns 0 or a negative code and leaves a message here (thread-local).
#include "smml_common.h"

#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void smml_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* smml_last_error(void) { return g_err; }

int smml_abi_version(void) { return 2; }   // 2: per-call SmmlDeformOpts instead of per-thread setters, region entry points (round 5)

// 0 when a gfx950 device is usable by this process, negative (with message) otherwise
int smml_device_check(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    smml_set_error("smml_device_check: no HIP device visible (%s)", hipGetErrorString(e));
    return SMML_ERR_HIP;
  }
  SMML_REQUIRE(device >= 0 && device < n, "smml_device_check: device %d out of range (%d visible)", device, n);
  hipDeviceProp_t p;
  e = hipGetDeviceProperties(&p, device);
  if (e != hipSuccess) {
    smml_set_error("smml_device_check: hipGetDeviceProperties failed (%s)", hipGetErrorString(e));
    return SMML_ERR_HIP;
  }
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
    smml_set_error("smml_device_check: device %d is %s, this library is built for gfx950 only", device, p.gcnArchName);
    return SMML_ERR_HIP;
  }
  return SMML_OK;
}

// HIP events as plain handles, so that a host can time single kernels on the stream they run on
// (bench.py's roofline leg).  ev_* arguments of the kernel entry points accept these handles.
void* smml_event_create(void) {
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) { smml_set_error("smml_event_create: hipEventCreate failed"); return nullptr; }
  return (void*)e;
}
int smml_event_destroy(void* ev) {
  if (ev && hipEventDestroy((hipEvent_t)ev) != hipSuccess) { smml_set_error("smml_event_destroy failed"); return SMML_ERR_HIP; }
  return SMML_OK;
}
int smml_event_record(void* ev, void* stream) {
  SMML_REQUIRE(ev, "smml_event_record: null event");
  if (hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) != hipSuccess) { smml_set_error("smml_event_record failed"); return SMML_ERR_HIP; }
  return SMML_OK;
}
// waits for `stop`, then returns the time between the two records in milliseconds
int smml_event_elapsed_ms(void* start, void* stop, float* ms) {
  SMML_REQUIRE(start && stop && ms, "smml_event_elapsed_ms: null argument");
  hipError_t e = hipEventSynchronize((hipEvent_t)stop);
  if (e == hipSuccess) e = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
  if (e != hipSuccess) { smml_set_error("smml_event_elapsed_ms: %s", hipGetErrorString(e)); return SMML_ERR_HIP; }
  return SMML_OK;
}

}  // extern "C"
