// Shared device/host helpers for libngp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ngp_hip.h"

#define NGP_API extern "C" __attribute__((visibility("default")))

void ngp_set_error(const char *fmt, ...);
#define NGP_REQUIRE(cond, code, ...) do { if (!(cond)) { ngp_set_error(__VA_ARGS__); return (code); } } while (0)
#define NGP_LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { ngp_set_error("%s: %s", name, hipGetErrorString(e_)); return (int)e_; } } while (0)

static inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// Every kernel launch goes through NGP_LAUNCH: identical to hipLaunchKernelGGL unless the kernel was enabled with ngp_prof_enable, in which case the launch is
// bracketed by a HIP event pair on its own stream (csrc/prof.hip).
extern int g_ngp_prof_on;
int ngp_prof_register(const char *kernel_expr);
struct NgpProfScope { int id; hipStream_t s; hipEvent_t a, b; NgpProfScope(int id, hipStream_t s); ~NgpProfScope(); };
#define NGP_LAUNCH(kernel, grid, block, shmem, stream, ...) do { \
	static const int ngp_kid_ = ngp_prof_register(#kernel); \
	NgpProfScope ngp_ps_(ngp_kid_, (hipStream_t)(stream)); \
	hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__); } while (0)

// internal cross-file entry points (not exported)
int ngp_hash_encode_bwd_ws_marked(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *dLdy, const uint32_t *level_table_host, void *grad, uint64_t n_params, int dtype,
                                  int grad_dtype, int in_layout, int zero_first, const uint32_t *n_valid, void *workspace, uint64_t workspace_bytes, hipEvent_t after_coarse);
int ngp_dp_reduce(void *comm, hipStream_t s, const NgpDpPlan *plan, void *grad, int dtype, uint32_t first_bucket, uint32_t last_bucket, float *tail_f32, float *extra_f32, uint64_t extra_count);

struct LevelTable { uint32_t v[64]; };   // [16][4] = offset, size, res, scale bits — passed by value (256 B of kernarg)

// ------------------------------------------------------------------ pcg32 (ops/op_include/pcg32/pcg32.h semantics)
struct Pcg32 {
	uint64_t state, inc;
	__host__ __device__ uint32_t next_uint() {
		uint64_t old = state;
		state = old * 0x5851f42d4c957f2dULL + inc;
		uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
		uint32_t rot = (uint32_t)(old >> 59u);
		return (xs >> rot) | (xs << ((~rot + 1u) & 31));
	}
	__host__ __device__ float next_float() {
		uint32_t u = (next_uint() >> 9) | 0x3f800000u;
		float f;
#if defined(__HIP_DEVICE_COMPILE__)
		f = __uint_as_float(u);
#else
		__builtin_memcpy(&f, &u, 4);
#endif
		return f - 1.0f;
	}
	__host__ __device__ void advance(uint64_t delta) {
		uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
		while (delta > 0) {
			if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
			cur_plus = (cur_mult + 1) * cur_plus;
			cur_mult *= cur_mult;
			delta >>= 1;
		}
		state = acc_mult * state + acc_plus;
	}
};

// ------------------------------------------------------------------ morton (ray_sampler_header.h:642-667)
__host__ __device__ static inline uint32_t expand_bits(uint32_t v) {
	v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u; return v;
}
__host__ __device__ static inline uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) { return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2); }
__host__ __device__ static inline uint32_t morton3D_invert(uint32_t x) {
	x = x & 0x49249249; x = (x | (x >> 2)) & 0xc30c30c3; x = (x | (x >> 4)) & 0x0f00f00f;
	x = (x | (x >> 8)) & 0xff0000ff; x = (x | (x >> 16)) & 0x0000ffff; return x;
}

// sampler constants (density_grid_sampler.py:35-39, 96-116)
#define NGP_GRIDSIZE 128u
#define NGP_STEPS 1024u
#define NGP_SQRT3 1.73205080757f
__host__ __device__ static inline float min_cone_stepsize() { return NGP_SQRT3 / NGP_STEPS; }
