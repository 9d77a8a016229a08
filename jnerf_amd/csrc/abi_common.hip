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

// Host helper: the hash-grid level table every encode call takes (position_encoders/hash_encoder/grid_encode.py:17-40 for offsets/sizes in fp64,
// op_header/HashEncode.h:149-151 for the fp32 per-level scale/resolution).  out: u32[16][4] = offset, size, resolution, scale bits.
#include <math.h>
#include <string.h>
NGP_API uint32_t ngp_level_table(double aabb_scale, uint32_t *out64_host) {
	const double s = exp(log(2048.0 * aabb_scale / 16.0) / 15.0);
	const float log2s = (float)log2(s);
	uint32_t off = 0;
	for (uint32_t l = 0; l < 16; ++l) {
		const double scale_h = pow(2.0, (double)l * log2(s)) * 16.0 - 1.0;
		const uint32_t res_h = (uint32_t)ceil(scale_h) + 1;
		uint64_t p = (uint64_t)res_h * res_h * res_h;
		p = (p + 7) / 8 * 8; if (p > (1u << 19)) p = 1u << 19;
		const float arg = (float)l * log2s;
		const float scale_d = (float)exp2((double)arg) * 16.0f - 1.0f;
		const uint32_t res_d = (uint32_t)ceilf(scale_d) + 1;
		out64_host[4 * l] = off; out64_host[4 * l + 1] = (uint32_t)p; out64_host[4 * l + 2] = res_d; memcpy(&out64_host[4 * l + 3], &scale_d, 4);
		off += (uint32_t)p;
	}
	return off * 2;
}
