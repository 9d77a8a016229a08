// Shared by the hash-grid translation units (hash_encode.hip: forward + table-gradient scatter; hash_order2.hip: input gradient and second-order terms; probes.hip):
// index arithmetic of HashEncode.h:68-115, the level-major block map, pair types.
#pragma once
#include "ngp_common.h"

template <typename T> struct Pair;
template <> struct Pair<float> { using type = float2; };
template <> struct Pair<__half> { using type = __half2; };

__device__ __forceinline__ float2 to_f2(float2 v) { return v; }
__device__ __forceinline__ float2 to_f2(__half2 v) { return __half22float2(v); }
__device__ __forceinline__ void from_f2(float2 &o, float2 v) { o = v; }
__device__ __forceinline__ void from_f2(__half2 &o, float2 v) { o = __floats2half2_rn(v.x, v.y); }

// HashEncode.h:68-94 with get_index(p0,p1,p2) = p0 ^ p1*19349663 ^ p2*83492791 (projects/ngp/configs/ngp_base.py:69)
__device__ __forceinline__ uint32_t grid_index(uint32_t size, uint32_t res, bool dense, uint32_t gx, uint32_t gy, uint32_t gz) {
	uint32_t index = dense ? gx + gy * res + gz * res * res : (gx ^ gy * 19349663u ^ gz * 83492791u);
	if ((size & (size - 1)) == 0) return index & (size - 1);   // hashed levels are 2^19 entries
	if (index >= size) { index -= size; if (index >= size) index %= size; }   // dense levels wrap only at the +1 boundary corner, and then by < size (res(1+res+res^2) < 2 res^3): the division is never executed for in-range positions
	return index;
}
// the reference decides "dense" by letting the stride loop run while stride <= size (HashEncode.h:82-91)
__device__ __forceinline__ bool level_is_dense(uint32_t size, uint32_t res) {
	uint32_t stride = 1;
#pragma unroll
	for (int d = 0; d < 3; ++d) if (stride <= size) stride *= res;
	return !(size < stride);
}

__device__ __forceinline__ void block_to_level_chunk(uint32_t nblk, uint32_t &level, uint32_t &chunk) {
#ifdef NGP_PROBE_LINEAR_MAP      // (diagnosis build only, tools/probe_shared_gpu.sh: levels laid end to end, no XCD-aware placement)
	level = blockIdx.x / nblk; chunk = blockIdx.x - level * nblk;
#else
	const uint32_t b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
	const uint32_t phase = slot / nblk;
	chunk = slot - phase * nblk;
	level = phase == 0 ? 15u - xcd : xcd;
#endif
}

static LevelTable load_table(const uint32_t *host) { LevelTable lt; for (int i = 0; i < 64; ++i) lt.v[i] = host[i]; return lt; }

struct Corner { uint32_t g[3]; float w[3]; };
__device__ __forceinline__ Corner locate(const float *pos, uint32_t stride, uint32_t i, float scale) {
	Corner c;
#pragma unroll
	for (int d = 0; d < 3; ++d) {            // pos_fract, HashEncode.h:106-115
		float p = pos[(size_t)i * stride + d] * scale + 0.5f;
		float fl = floorf(p);
		c.g[d] = (uint32_t)(int)fl;
		c.w[d] = p - fl;
	}
	return c;
}

__device__ __forceinline__ void atomic_add_pair(float *p, float2 v) {
	__hip_atomic_fetch_add(p, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__hip_atomic_fetch_add(p + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void atomic_add_pair(__half *p, float2 v) {
	typedef _Float16 __attribute__((ext_vector_type(2))) h2;
	h2 x; x[0] = (_Float16)v.x; x[1] = (_Float16)v.y;
	(void)__builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2 *)p, x);   // global_atomic_pk_add_f16
}
