// Input gradient of the hash-grid encoding and its second-order terms (hash-grid SDF network, BASELINE configs[4]): split from hash_encode.hip in round 5 - none of this
// is on the Instant-NGP training path.
#include "hash_common.h"
#include <stdlib.h>
#pragma clang fp contract(off)

// dL/dx[i][d] = sum_k dL/dy[i][k] * dy_dx[i][d][k] (fp32, k ascending): the contraction GridEncode.grad needs to return a position gradient.  The reference returns
// None there (grid_encode.py:190) and has no kernel for it - restated from the chain rule (tiny-cuda-nn's kernel_grid_backward_input computes the same sum).
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void k_hash_bwd_input(uint32_t n, const T *__restrict__ dLdy, const float *__restrict__ dy_dx, float *__restrict__ dLdx, const uint32_t *__restrict__ n_valid) {
	using P = typename Pair<T>::type;
	const uint32_t t = blockIdx.x * 256u + threadIdx.x, i = t / 3u, d = t - 3u * i;
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	if (i >= lim) return;
	const P *dy = reinterpret_cast<const P *>(dLdy);
	const float2 *row = reinterpret_cast<const float2 *>(dy_dx + (size_t)i * 96 + d * 32);
	float a = 0.f;
#pragma unroll
	for (uint32_t l = 0; l < 16; ++l) {
		const float2 g = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)l * n + i] : dy[(size_t)i * 16 + l]);
		const float2 r = row[l];
		a += g.x * r.x; a += g.y * r.y;
	}
	dLdx[(size_t)i * 3 + d] = a;
}

NGP_API int ngp_hash_encode_bwd_input(void *stream, uint32_t n, const void *dLdy, int dtype, int in_layout, const float *dy_dx, float *dLdx, const uint32_t *n_valid) {
	NGP_REQUIRE(n == 0 || (dLdy && dy_dx && dLdx), NGP_E_ARG, "ngp_hash_encode_bwd_input: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_hash_encode_bwd_input: bad dtype %d", dtype);
	NGP_REQUIRE(((uintptr_t)dy_dx & 7) == 0, NGP_E_ALIGN, "ngp_hash_encode_bwd_input: dy_dx must be 8-byte aligned");
	if (n == 0) return 0;
	const dim3 grid(div_up(n * 3u, 256)), block(256);
	hipStream_t s = (hipStream_t)stream;
#define GO(T, L) NGP_LAUNCH((k_hash_bwd_input<T, L>), grid, block, 0, s, n, (const T *)dLdy, dy_dx, dLdx, n_valid)
	if (dtype == NGP_F32) { if (in_layout == NGP_LAYOUT_SOA) GO(float, NGP_LAYOUT_SOA); else GO(float, NGP_LAYOUT_AOS); }
	else { if (in_layout == NGP_LAYOUT_SOA) GO(__half, NGP_LAYOUT_SOA); else GO(__half, NGP_LAYOUT_AOS); }
#undef GO
	NGP_LAUNCH_CHECK("ngp_hash_encode_bwd_input");
	return 0;
}


// ---------------------------------------------------------------------------------------------------------------- second-order terms (hash-grid SDF network, r3)
// A network that is trained on its own input gradient (NeuS' eikonal term and normal-fed colour network over a hash-grid SDF, BASELINE configs[4]) back-propagates
// through g = dL/dx = sum_k dLdy_k * dy_k/dx (k_hash_bwd_input).  For an upstream gradient u = d loss / d g [n,3] that needs
//   (i)  d loss / d dLdy[i][k]      = sum_d u[i][d] * dy_dx[i][d][k]                                   (k_hash_bwd_input_bwd_dy: the transposed contraction)
//   (ii) d loss / d table[e][f]    += dLdy[i][2l+f] * sum_d u[i][d] * d w_c(x_i) / d x_d               for every corner c of (sample i, level l) that lands on entry e
//        (k_hash_bwd_input_bwd_grid: the table scatter of k_hash_bwd with the interpolation weight replaced by its directional derivative along u; d w_c / d x_d is
//        the weight the dy_dx branch uses - scale * w(first other dim) * w(second other dim), positive for the corner on the right of dimension d, negative on the left)
// The reference has neither (its dy_dx branch is never enabled); tiny-cuda-nn's kernel_grid_backward_input_backward_grid computes (ii).  The mixed second derivative
// w.r.t. the position itself is not produced (NeuS' sample positions carry no parameters).  Few samples (512 rays x 128), fp32 float atomics: not a hot-path kernel.
template <typename T>
__global__ __launch_bounds__(256) void k_hash_bwd_input_bwd_dy(uint32_t n, const float *__restrict__ u, const float *__restrict__ dy_dx, T *__restrict__ ddy) {
	using P = typename Pair<T>::type;
	const uint32_t t = blockIdx.x * 256u + threadIdx.x, i = t >> 4, l = t & 15u;
	if (i >= n) return;
	float2 a = make_float2(0.f, 0.f);
#pragma unroll
	for (uint32_t d = 0; d < 3; ++d) {
		const float ud = u[(size_t)i * 3 + d];
		const float2 r = *reinterpret_cast<const float2 *>(dy_dx + (size_t)i * 96 + d * 32 + 2 * l);
		a.x += ud * r.x; a.y += ud * r.y;
	}
	P o; from_f2(o, a);
	reinterpret_cast<P *>(ddy)[(size_t)i * 16 + l] = o;
}
template <typename T>
__global__ __launch_bounds__(256) void k_hash_bwd_input_bwd_grid(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ dLdy, const float *__restrict__ u,
                                                                 LevelTable lt, float *__restrict__ grad, uint32_t nblk) {
	using P = typename Pair<T>::type;
	uint32_t level, chunk0; block_to_level_chunk(nblk, level, chunk0);
	const uint32_t off = lt.v[4 * level], size = lt.v[4 * level + 1], res = lt.v[4 * level + 2];
	const float scale = __uint_as_float(lt.v[4 * level + 3]);
	const bool dense = level_is_dense(size, res);
	float *gl = grad + (size_t)off * 2;
	for (uint32_t i = chunk0 * 256u + threadIdx.x; i < n; i += nblk * 256u) {
		const float2 g2 = to_f2(reinterpret_cast<const P *>(dLdy)[(size_t)i * 16 + level]);
		const float u0 = u[(size_t)i * 3], u1 = u[(size_t)i * 3 + 1], u2 = u[(size_t)i * 3 + 2];
		if ((g2.x == 0.f && g2.y == 0.f) || (u0 == 0.f && u1 == 0.f && u2 == 0.f)) continue;
		const Corner c = locate(pos, stride, i, scale);
		const float uu[3] = {u0, u1, u2};
#pragma unroll
		for (uint32_t k = 0; k < 8; ++k) {
			float wk = 0.f;
#pragma unroll
			for (uint32_t gd = 0; gd < 3; ++gd) {
				const uint32_t d0 = gd == 0 ? 1u : 0u, d1 = gd == 2 ? 1u : 2u;
				float weight = scale;
				weight *= (k >> d0) & 1u ? c.w[d0] : 1 - c.w[d0];
				weight *= (k >> d1) & 1u ? c.w[d1] : 1 - c.w[d1];
				wk += uu[gd] * ((k >> gd) & 1u ? weight : -weight);
			}
			const uint32_t idx = grid_index(size, res, dense, c.g[0] + (k & 1u), c.g[1] + ((k >> 1) & 1u), c.g[2] + (k >> 2));
			atomic_add_pair(gl + (size_t)idx * 2, make_float2(g2.x * wk, g2.y * wk));
		}
	}
}
NGP_API int ngp_hash_encode_bwd_input_bwd_dy(void *stream, uint32_t n, const float *u, const float *dy_dx, void *ddLdy, int dtype) {
	NGP_REQUIRE(n == 0 || (u && dy_dx && ddLdy), NGP_E_ARG, "ngp_hash_encode_bwd_input_bwd_dy: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_hash_encode_bwd_input_bwd_dy: bad dtype %d", dtype);
	NGP_REQUIRE(((uintptr_t)dy_dx & 7) == 0, NGP_E_ALIGN, "ngp_hash_encode_bwd_input_bwd_dy: dy_dx must be 8-byte aligned");
	if (n == 0) return 0;
	NGP_REQUIRE(n <= (1u << 27), NGP_E_ARG, "ngp_hash_encode_bwd_input_bwd_dy: n = %u too large", n);
	const dim3 grid(div_up(n * 16u, 256)), block(256);
	hipStream_t s = (hipStream_t)stream;
	if (dtype == NGP_F32) NGP_LAUNCH((k_hash_bwd_input_bwd_dy<float>), grid, block, 0, s, n, u, dy_dx, (float *)ddLdy);
	else NGP_LAUNCH((k_hash_bwd_input_bwd_dy<__half>), grid, block, 0, s, n, u, dy_dx, (__half *)ddLdy);
	NGP_LAUNCH_CHECK("ngp_hash_encode_bwd_input_bwd_dy");
	return 0;
}
NGP_API int ngp_hash_encode_bwd_input_bwd_grid(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *dLdy, int dtype, const float *u,
                                               const uint32_t *level_table_host, float *grad, uint64_t n_params) {
	NGP_REQUIRE(n == 0 || (pos && dLdy && u && level_table_host && grad), NGP_E_ARG, "ngp_hash_encode_bwd_input_bwd_grid: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_hash_encode_bwd_input_bwd_grid: bad dtype %d", dtype);
	if (n == 0) return 0;
	NGP_REQUIRE(pos_stride >= 3, NGP_E_ARG, "ngp_hash_encode_bwd_input_bwd_grid: pos stride %u < 3", pos_stride);
	const LevelTable lt = load_table(level_table_host);
	NGP_REQUIRE((uint64_t)(lt.v[4 * 15] + lt.v[4 * 15 + 1]) * 2u <= n_params, NGP_E_ARG, "ngp_hash_encode_bwd_input_bwd_grid: level table needs %llu parameters, grad has %llu",
	            (unsigned long long)(lt.v[4 * 15] + lt.v[4 * 15 + 1]) * 2ull, (unsigned long long)n_params);
	const uint32_t nblk = min(div_up(n, 256), 2048u);
	const dim3 grid(16 * nblk), block(256);
	hipStream_t s = (hipStream_t)stream;
	if (dtype == NGP_F32) NGP_LAUNCH((k_hash_bwd_input_bwd_grid<float>), grid, block, 0, s, n, pos, pos_stride, (const float *)dLdy, u, lt, grad, nblk);
	else NGP_LAUNCH((k_hash_bwd_input_bwd_grid<__half>), grid, block, 0, s, n, pos, pos_stride, (const __half *)dLdy, u, lt, grad, nblk);
	NGP_LAUNCH_CHECK("ngp_hash_encode_bwd_input_bwd_grid");
	return 0;
}
