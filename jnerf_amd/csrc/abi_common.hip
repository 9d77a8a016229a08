// Error plumbing + device info for libngp_hip.so.
#include "ngp_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void ngp_set_error(const char *fmt, ...) {
	va_list ap; va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}
NGP_API int ngp_abi_version(void) { return NGP_ABI_VERSION; }
NGP_API const char *ngp_last_error(void) { return g_err; }
NGP_API int ngp_device_info(int device, int64_t *out4) {
	hipDeviceProp_t p;
	hipError_t e = hipGetDeviceProperties(&p, device);
	if (e != hipSuccess) { ngp_set_error("hipGetDeviceProperties: %s", hipGetErrorString(e)); return (int)e; }
	out4[0] = p.multiProcessorCount; out4[1] = p.clockRate; out4[2] = p.l2CacheSize; out4[3] = (int64_t)(p.totalGlobalMem >> 20);
	return 0;
}
