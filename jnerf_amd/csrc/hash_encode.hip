// Multiresolution hash-grid encode, forward + backward, for gfx950.
//
// What it computes: position_encoders/hash_encoder/op_header/HashEncode.h:117-203 (kernel_grid) and :299-396
// (kernel_grid_backward) of the reference, with the level table (offset/size/resolution/scale) precomputed on the host.
//
// MI355X design (not the reference's launch shape):
//  * one thread = one (sample, level); a workgroup works on ONE level so its gathers stay inside that level's slice of the
//    table (<= 2 MiB fp16 / 4 MiB fp32) ...
//  * ... and the blockIdx -> (level, chunk) map is XCD-aware: the dispatcher places block b on XCD b%8 and each XCD has a
//    private 4 MiB L2, so XCD x is handed level 15-x for ALL sample chunks first and level x afterwards.  A level's table
//    slice is then fetched from HBM/Infinity-Cache once per XCD and served out of that XCD's L2 for the rest of the pass
//    instead of being bounced between eight L2s.  This mapping affects speed only, never results.
//  * no extract_position / transpose kernels: positions are read strided from the caller's buffer and features are written
//    either as the [n,32] rows HashEncoder returns or as a level-major [16][n] stream of pairs that the fused MLP consumes with
//    fully coalesced 256-B wave accesses (no 4-byte-per-64-byte-line partial writes from sixteen different XCDs).
//  * backward uses hardware float atomics (global_atomic_add_f32 / global_atomic_pk_add_f16), gradient zeroing is a fused memset.
#include "ngp_common.h"
#pragma clang fp contract(off)

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
	return index < size ? index : index % size;                // dense levels wrap only at the +1 boundary corner
}
// the reference decides "dense" by letting the stride loop run while stride <= size (HashEncode.h:82-91)
__device__ __forceinline__ bool level_is_dense(uint32_t size, uint32_t res) {
	uint32_t stride = 1;
#pragma unroll
	for (int d = 0; d < 3; ++d) if (stride <= size) stride *= res;
	return !(size < stride);
}

__device__ __forceinline__ void block_to_level_chunk(uint32_t nblk, uint32_t &level, uint32_t &chunk) {
	const uint32_t b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
	const uint32_t phase = slot / nblk;
	chunk = slot - phase * nblk;
	level = phase == 0 ? 15u - xcd : xcd;
}

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

template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void k_hash_fwd(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ table, LevelTable lt,
                                                  T *__restrict__ out, uint32_t nblk, const uint32_t *__restrict__ n_valid) {
	using P = typename Pair<T>::type;
	uint32_t level, chunk; block_to_level_chunk(nblk, level, chunk);
	const uint32_t i = chunk * 256u + threadIdx.x;
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	if (i >= lim) return;
	const uint32_t off = lt.v[4 * level], size = lt.v[4 * level + 1], res = lt.v[4 * level + 2];
	const float scale = __uint_as_float(lt.v[4 * level + 3]);
	const bool dense = level_is_dense(size, res);
	const Corner c = locate(pos, stride, i, scale);
	const P *tab = reinterpret_cast<const P *>(table) + off;
	P v[8]; float w[8];
#pragma unroll
	for (uint32_t k = 0; k < 8; ++k) {        // issue all eight gathers before the first use
		float weight = 1; uint32_t g[3];
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			if ((k & (1u << d)) == 0) { weight *= 1 - c.w[d]; g[d] = c.g[d]; }
			else { weight *= c.w[d]; g[d] = c.g[d] + 1; }
		}
		w[k] = weight;
		v[k] = tab[grid_index(size, res, dense, g[0], g[1], g[2])];
	}
	float2 acc = make_float2(0.f, 0.f);
#pragma unroll
	for (uint32_t k = 0; k < 8; ++k) { float2 f = to_f2(v[k]); acc.x += w[k] * f.x; acc.y += w[k] * f.y; }
	P r; from_f2(r, acc);
	P *o = reinterpret_cast<P *>(out);
	if (LAYOUT == NGP_LAYOUT_SOA) o[(size_t)level * n + i] = r;
	else o[(size_t)i * 16 + level] = r;
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

template <typename T, typename G, int LAYOUT>
__global__ __launch_bounds__(256) void k_hash_bwd(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ dLdy, LevelTable lt,
                                                  G *__restrict__ grad, uint32_t nblk, const uint32_t *__restrict__ n_valid) {
	using P = typename Pair<T>::type;
	uint32_t level, chunk; block_to_level_chunk(nblk, level, chunk);
	const uint32_t i = chunk * 256u + threadIdx.x;
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	if (i >= lim) return;
	const uint32_t off = lt.v[4 * level], size = lt.v[4 * level + 1], res = lt.v[4 * level + 2];
	const float scale = __uint_as_float(lt.v[4 * level + 3]);
	const bool dense = level_is_dense(size, res);
	const P *dy = reinterpret_cast<const P *>(dLdy);
	const float2 g2 = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)level * n + i] : dy[(size_t)i * 16 + level]);
	if (g2.x == 0.f && g2.y == 0.f) return;   // zero-padded rows add exact zeros in the reference; skipping them is value-identical
	const Corner c = locate(pos, stride, i, scale);
	G *gl = grad + (size_t)off * 2;
#pragma unroll
	for (uint32_t k = 0; k < 8; ++k) {
		float weight = 1; uint32_t g[3];
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			if ((k & (1u << d)) == 0) { weight *= 1 - c.w[d]; g[d] = c.g[d]; }
			else { weight *= c.w[d]; g[d] = c.g[d] + 1; }
		}
		const uint32_t idx = grid_index(size, res, dense, g[0], g[1], g[2]);
		atomic_add_pair(gl + (size_t)idx * 2, make_float2(g2.x * weight, g2.y * weight));
	}
}

static LevelTable load_table(const uint32_t *host) { LevelTable lt; for (int i = 0; i < 64; ++i) lt.v[i] = host[i]; return lt; }

NGP_API int ngp_hash_encode_fwd(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *table, const uint32_t *level_table_host,
                                void *out, int dtype, int out_layout, const uint32_t *n_valid) {
	NGP_REQUIRE(n == 0 || (pos && table && level_table_host && out), NGP_E_ARG, "ngp_hash_encode_fwd: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_hash_encode_fwd: bad dtype %d", dtype);
	if (n == 0) return 0;
	NGP_REQUIRE(pos_stride >= 3, NGP_E_ARG, "ngp_hash_encode_fwd: pos stride %u < 3", pos_stride);
	const uint32_t nblk = div_up(n, 256);
	const dim3 grid(16 * nblk), block(256);
	const LevelTable lt = load_table(level_table_host);
	hipStream_t s = (hipStream_t)stream;
#define GO(T, L) hipLaunchKernelGGL((k_hash_fwd<T, L>), grid, block, 0, s, n, pos, pos_stride, (const T *)table, lt, (T *)out, nblk, n_valid)
	if (dtype == NGP_F32) { if (out_layout == NGP_LAYOUT_SOA) GO(float, NGP_LAYOUT_SOA); else GO(float, NGP_LAYOUT_AOS); }
	else { if (out_layout == NGP_LAYOUT_SOA) GO(__half, NGP_LAYOUT_SOA); else GO(__half, NGP_LAYOUT_AOS); }
#undef GO
	NGP_LAUNCH_CHECK("ngp_hash_encode_fwd");
	return 0;
}

NGP_API int ngp_hash_encode_bwd(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *dLdy, const uint32_t *level_table_host,
                                void *grad, uint64_t n_params, int dtype, int grad_dtype, int in_layout, int zero_first, const uint32_t *n_valid) {
	NGP_REQUIRE(grad && (n == 0 || (pos && dLdy && level_table_host)), NGP_E_ARG, "ngp_hash_encode_bwd: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_hash_encode_bwd: bad dtype %d", dtype);
	NGP_REQUIRE(grad_dtype == NGP_F32 || (grad_dtype == NGP_F16 && dtype == NGP_F16), NGP_E_DTYPE, "ngp_hash_encode_bwd: bad grad dtype %d for dtype %d", grad_dtype, dtype);
	hipStream_t s = (hipStream_t)stream;
	if (zero_first) {
		hipError_t e = hipMemsetAsync(grad, 0, n_params * (grad_dtype == NGP_F16 ? 2 : 4), s);
		if (e != hipSuccess) { ngp_set_error("ngp_hash_encode_bwd memset: %s", hipGetErrorString(e)); return (int)e; }
	}
	if (n == 0) return 0;
	const uint32_t nblk = div_up(n, 256);
	const dim3 grid(16 * nblk), block(256);
	const LevelTable lt = load_table(level_table_host);
#define GO(T, G, L) hipLaunchKernelGGL((k_hash_bwd<T, G, L>), grid, block, 0, s, n, pos, pos_stride, (const T *)dLdy, lt, (G *)grad, nblk, n_valid)
	if (dtype == NGP_F32) { if (in_layout == NGP_LAYOUT_SOA) GO(float, float, NGP_LAYOUT_SOA); else GO(float, float, NGP_LAYOUT_AOS); }
	else if (grad_dtype == NGP_F32) { if (in_layout == NGP_LAYOUT_SOA) GO(__half, float, NGP_LAYOUT_SOA); else GO(__half, float, NGP_LAYOUT_AOS); }
	else { if (in_layout == NGP_LAYOUT_SOA) GO(__half, __half, NGP_LAYOUT_SOA); else GO(__half, __half, NGP_LAYOUT_AOS); }
#undef GO
	NGP_LAUNCH_CHECK("ngp_hash_encode_bwd");
	return 0;
}
