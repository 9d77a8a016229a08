// fp32 field network for gfx950: what NGPNetworks.execute_ computes when cfg.fp16 is unset (models/networks/ngp_network.py:57-67, 77-84 — the
// configuration projects/ngp/configs/ngp_base.py, i.e. the lego headline, runs): SH direction encoding + density MLP (32->64->16) + colour MLP
// (32->64->64->3), no biases, fp32 weights, fp32 activations, fp32 accumulation; forward and backward (dL/dfeatures + the five weight gradients).
// The reference executes this as five cuBLAS GEMMs + elementwise kernels + three concats forward and twice that backward, with every [n,64]
// intermediate written to and re-read from memory (2.3 ms per 2^18-sample batch through rocBLAS here).
//
// CDNA4 design (same structure as field_mlp.hip, different matrix instruction):
//  * v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate (an fma chain; no xf32/TF32 exists on gfx950).  157 TFLOP/s dense peak: one 2^18-sample
//    batch is 5.4 GFLOP forward / 16.1 GFLOP backward, so the kernels are MFMA-issue-bound, not HBM-bound.
//  * "transposed" formulation: Y^T[neurons x samples] = W[neurons x k] * X^T[k x samples]; A = weights, B = 16 samples of a wave tile.  The C fragment of a
//    layer (lane = sample lane&15, registers = neurons 4*(lane>>4)+r of every 16-neuron tile) IS the B operand of the next layer's k-steps: k-step (tile t, r)
//    takes register r of tile t, i.e. lane group g supplies neuron 16t+4g+r.  The matching A operand W[o][16t+4g+r], r = 0..3, is four CONSECUTIVE floats of
//    a weight row, so every weight fragment is one 16-byte LDS read per lane serving four MFMAs.  Activations never leave registers between layers.
//  * weights are staged in LDS as pre-permuted 1-KiB fragments (lane-contiguous: conflict-free ds_read_b128), built once per step by k_pack_frags32.
//  * backward recomputes the forward, runs dgrad with transposed fragments, and contracts the weight gradients over SAMPLES through LDS ([neuron][sample]
//    rows, five staging phases that reuse one 66-KiB region), one fp32 slab per workgroup, summed by ngp_reduce_slabs (deterministic, no atomics).
#include "ngp_common.h"
#include "field_split.h"
#include "mlp_tail.h"
#include <stdlib.h>
#include <map>
#include <mutex>
#include <utility>

typedef float floatx4 __attribute__((ext_vector_type(4)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define NGP_FIELD32_BWD_DEFAULT 3       // 3 = split fp16 operands (field_split.hip): 109 us in the training step vs 139 us for 2 = two free-running groups on fp32 MFMAs (+4.8 % it/s, A/B on one box)
// NF32_FWD / NF32_BWD / NF32_ALL and frag_value32 (the fp32 fragment layout): mlp_tail.h, shared with the hash backward's record kernels that carry the MLP tail (r6)
__global__ __launch_bounds__(256) void k_pack_frags32(const float *__restrict__ wd, const float *__restrict__ wc, float *__restrict__ out, int n_frags) {
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= n_frags * 256) return;
	const int f = idx >> 8, lane = (idx >> 2) & 63, j = idx & 3;
	out[idx] = frag_value32(wd, wc, f, lane & 15, lane >> 4, j);
}
__device__ __forceinline__ void stage_weights32(float *lds, const float *__restrict__ packed, int n_frags) {
	const float4 *src = reinterpret_cast<const float4 *>(packed);
	float4 *dst = reinterpret_cast<float4 *>(lds);
	for (int idx = threadIdx.x; idx < n_frags * 64; idx += blockDim.x) dst[idx] = src[idx];
}
__device__ __forceinline__ floatx4 ld_frag32(const float *lds, int f, int lane) { return *reinterpret_cast<const floatx4 *>(lds + f * 256 + lane * 4); }

__device__ __forceinline__ floatx4 relu4(floatx4 a) { floatx4 r; r[0] = fmaxf(a[0], 0.f); r[1] = fmaxf(a[1], 0.f); r[2] = fmaxf(a[2], 0.f); r[3] = fmaxf(a[3], 0.f); return r; }
// relu'(activation) * grad; the activation is the exact fp32 pre-activation clamped at zero, so act > 0 <=> pre-activation > 0
__device__ __forceinline__ floatx4 mask4(floatx4 grad, floatx4 act) {
	floatx4 r;
#pragma unroll
	for (int k = 0; k < 4; ++k) r[k] = act[k] > 0.f ? grad[k] : 0.f;
	return r;
}

// degree-4 SH of (2d-1), components 4g..4g+3 (SphericalEncode.h:77-95)
__device__ __forceinline__ void sh4_32(const float d[3], int g, float o[4]) {
	const float x = d[0] * 2.f - 1.f, y = d[1] * 2.f - 1.f, z = d[2] * 2.f - 1.f;
	const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
	if (g == 0) { o[0] = 0.28209479177387814f; o[1] = -0.48860251190291987f * y; o[2] = 0.48860251190291987f * z; o[3] = -0.48860251190291987f * x; }
	else if (g == 1) { o[0] = 1.0925484305920792f * xy; o[1] = -1.0925484305920792f * yz; o[2] = 0.94617469575755997f * z2 - 0.31539156525251999f; o[3] = -1.0925484305920792f * xz; }
	else if (g == 2) { o[0] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2; o[1] = 0.59004358992664352f * y * (-3.0f * x2 + y2); o[2] = 2.8906114426405538f * xy * z; o[3] = 0.45704579946446572f * y * (1.0f - 5.0f * z2); }
	else { o[0] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f); o[1] = 0.45704579946446572f * x * (1.0f - 5.0f * z2); o[2] = 1.4453057213202769f * z * (x2 - y2); o[3] = 0.59004358992664352f * x * (-x2 + 3.0f * y2); }
}

// features 8g..8g+7 of sample i (levels 4g..4g+3)
template <int LAYOUT>
__device__ __forceinline__ void load_feat32(const float *__restrict__ feat, uint32_t n, uint32_t i, int g, float f[8]) {
	if (LAYOUT == NGP_LAYOUT_SOA) {
		const float2 *p = reinterpret_cast<const float2 *>(feat);
#pragma unroll
		for (int q = 0; q < 4; ++q) { const float2 v = p[(size_t)(4 * g + q) * n + i]; f[2 * q] = v.x; f[2 * q + 1] = v.y; }
	} else {
		const float4 *p = reinterpret_cast<const float4 *>(feat + (size_t)i * 32 + 8 * g);
		const float4 a = p[0], b = p[1];
		f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
	}
}

struct Fwd32 { floatx4 h[4], den, g0[4], g1[4], rgb; };      // h, g0, g1: post-ReLU activations (C layout: [tile][r] = neuron 16*tile + 4g + r of sample lane&15)

// one layer: acc[u] (NU output tiles) += sum over NK k-groups of 4 k-steps; A fragments f0 + u*NK + kq, B value of k-group kq, step j = b(kq, j)
template <bool DENSITY_ONLY>
__device__ __forceinline__ void forward32(const float *wl, int lane, const float feat[8], const float sh[4], Fwd32 &st) {
	const floatx4 z = {0.f, 0.f, 0.f, 0.f};
	floatx4 acc[4] = {z, z, z, z};
#pragma unroll
	for (int kq = 0; kq < 2; ++kq) {                       // L0: 32 -> 64
		floatx4 a[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) a[u] = ld_frag32(wl, 2 * u + kq, lane);
#pragma unroll
		for (int j = 0; j < 4; ++j)
#pragma unroll
			for (int u = 0; u < 4; ++u) acc[u] = MFMA32(a[u][j], feat[4 * kq + j], acc[u]);
	}
#pragma unroll
	for (int u = 0; u < 4; ++u) st.h[u] = relu4(acc[u]);
	floatx4 d0 = z, d1 = z;                                // L1: 64 -> 16, two partial accumulators (a single 16-wide output tile would be a 16-deep dependent chain)
#pragma unroll
	for (int kq = 0; kq < 4; ++kq) {
		const floatx4 a = ld_frag32(wl, 8 + kq, lane);
		d0 = MFMA32(a[0], st.h[kq][0], d0); d1 = MFMA32(a[1], st.h[kq][1], d1);
		d0 = MFMA32(a[2], st.h[kq][2], d0); d1 = MFMA32(a[3], st.h[kq][3], d1);
	}
	st.den = d0 + d1;
	if (DENSITY_ONLY) return;
#pragma unroll
	for (int u = 0; u < 4; ++u) acc[u] = z;
#pragma unroll
	for (int kq = 0; kq < 2; ++kq) {                       // L2: [density(16) | SH(16)] -> 64
		floatx4 a[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) a[u] = ld_frag32(wl, 12 + 2 * u + kq, lane);
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const float b = kq == 0 ? st.den[j] : sh[j];
#pragma unroll
			for (int u = 0; u < 4; ++u) acc[u] = MFMA32(a[u][j], b, acc[u]);
		}
	}
#pragma unroll
	for (int u = 0; u < 4; ++u) { st.g0[u] = relu4(acc[u]); acc[u] = z; }
#pragma unroll
	for (int kq = 0; kq < 4; ++kq) {                       // L3: 64 -> 64
		floatx4 a[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) a[u] = ld_frag32(wl, 20 + 4 * u + kq, lane);
#pragma unroll
		for (int j = 0; j < 4; ++j)
#pragma unroll
			for (int u = 0; u < 4; ++u) acc[u] = MFMA32(a[u][j], st.g0[kq][j], acc[u]);
	}
#pragma unroll
	for (int u = 0; u < 4; ++u) st.g1[u] = relu4(acc[u]);
	d0 = z; d1 = z;                                        // L4: 64 -> 16 (3 used)
#pragma unroll
	for (int kq = 0; kq < 4; ++kq) {
		const floatx4 a = ld_frag32(wl, 36 + kq, lane);
		d0 = MFMA32(a[0], st.g1[kq][0], d0); d1 = MFMA32(a[1], st.g1[kq][1], d1);
		d0 = MFMA32(a[2], st.g1[kq][2], d0); d1 = MFMA32(a[3], st.g1[kq][3], d1);
	}
	st.rgb = d0 + d1;
}

template <int LAYOUT, bool DENSITY_ONLY>
__global__ __launch_bounds__(256) void k_field32_fwd(uint32_t n, const float *__restrict__ feat, const float *__restrict__ dir, uint32_t dir_stride,
                                                     const float *__restrict__ packed, float *__restrict__ out, const uint32_t *__restrict__ n_valid) {
	__shared__ __attribute__((aligned(16))) float wl[NF32_FWD * 256];
	stage_weights32(wl, packed, DENSITY_ONLY ? 12 : NF32_FWD);
	__syncthreads();
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	const int lane = threadIdx.x & 63, s = lane & 15, g = lane >> 4;
	const uint32_t n_tiles = (lim + 15u) / 16u;
	const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
	auto fetch = [&](uint32_t tile, float f[8], float d[3]) {
		const uint32_t i = tile * 16u + s;
		const uint32_t ic = i < lim ? i : lim - 1;
		load_feat32<LAYOUT>(feat, n, ic, g, f);
		if (!DENSITY_ONLY) { d[0] = dir[(size_t)ic * dir_stride]; d[1] = dir[(size_t)ic * dir_stride + 1]; d[2] = dir[(size_t)ic * dir_stride + 2]; }
	};
	float f[8], fn[8], d[3] = {0.f, 0.f, 0.f}, dn[3] = {0.f, 0.f, 0.f};
	if (wave < n_tiles) fetch(wave, f, d);
	for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
		const uint32_t i = tile * 16u + s;
		const bool more = tile + n_waves < n_tiles;
		if (more) fetch(tile + n_waves, fn, dn);                   // the next tile's inputs are in flight during this tile's 160 MFMAs
		float sh[4] = {0.f, 0.f, 0.f, 0.f};
		if (!DENSITY_ONLY) sh4_32(d, g, sh);
		Fwd32 st;
		forward32<DENSITY_ONLY>(wl, lane, f, sh, st);
		if (g == 0 && i < lim) {
			if (DENSITY_ONLY) out[i] = st.den[0];
			else *reinterpret_cast<float4 *>(out + (size_t)i * 4) = make_float4(st.rgb[0], st.rgb[1], st.rgb[2], st.den[0]);
		}
		if (more) {
#pragma unroll
			for (int q = 0; q < 8; ++q) f[q] = fn[q];
			d[0] = dn[0]; d[1] = dn[1]; d[2] = dn[2];
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------- backward
#define BT32 128               // samples per workgroup trip: 8 waves x 16
// (r5) The lock-step kernel of round 2 (145 us) and the ping-pong experiment of round 3 (151 us) are gone from the binary: measurements and what they taught are in
// DESIGN.md 6.  One exact-product backward remains - two free-running groups, below - as the fallback of the split-operand kernels (ngp_field32_select).
#define HT32 64                 // samples per half trip: 4 waves x 16
#define RSH32 (HT32 + 4)        // LDS row stride in floats (272 B: 16-byte aligned rows, consecutive rows shift by one 16-byte slot)
__device__ __forceinline__ void st_tiles_h(float *stage, int row0, int col, int g, const floatx4 *v) {
#pragma unroll
	for (int t = 0; t < 4; ++t)
#pragma unroll
		for (int r = 0; r < 4; ++r) stage[(row0 + 16 * t + 4 * g + r) * RSH32 + col] = v[t][r];
}
__device__ __forceinline__ floatx4 wgrad_tile_h(const float *stage, int row_a, int row_b, int o, int g, floatx4 acc) {
#pragma unroll
	for (int c = 0; c < HT32; c += 16) {
		const floatx4 a = *reinterpret_cast<const floatx4 *>(stage + (row_a + o) * RSH32 + c + 4 * g), b = *reinterpret_cast<const floatx4 *>(stage + (row_b + o) * RSH32 + c + 4 * g);
#pragma unroll
		for (int j = 0; j < 4; ++j) acc = MFMA32(a[j], b[j], acc);
	}
	return acc;
}

// ---------------------------------------------------------------------------------------------------------------- backward, two free-running groups (r3)
// What the ping-pong variant taught: pairing the forward/dgrad chain with the other half's staging phases through SHARED barriers chops the chain into ten blocks, and
// every block start exposes the latency of its fragment loads (151 us vs 144).  What it needs is the overlap without the coupling: here the two halves of the workgroup
// (waves 0-3 / 4-7, one wave of each per SIMD) are two INDEPENDENT groups that share nothing but the read-only weight fragments - own staging region (2 x 34 KiB), own
// barrier (an LDS arrival counter the group's four waves poll: s_barrier would stop the other group too), own half trips.  Each group runs the plain sequence - 284
// MFMAs of forward + dgrad without any synchronisation, then the five staging phases - and group 1 starts half an iteration late, so one group's LDS-write / barrier
// phases fall into the other group's MFMA stretch on the same SIMD.  It is what two workgroups per CU would do if 76 KiB of fragments fitted twice.
template <int LAYOUT>
__global__ __launch_bounds__(512, 2) void k_field32_bwd_2g(uint32_t n, const float *__restrict__ feat, const float *__restrict__ dir, uint32_t dir_stride,
                                                           const float *__restrict__ packed, const float *__restrict__ dout,
                                                           float *__restrict__ dfeat, float *__restrict__ slabs, const uint32_t *__restrict__ n_valid, AbsmaxOut am) {
	extern __shared__ __attribute__((aligned(16))) float smem32[];
	float *wl = smem32;                                   // 76 fragments
	float *stage = smem32 + NF32_ALL * 256 + (size_t)((threadIdx.x >> 8) * 128 * RSH32);      // [128][RSH32], one region per GROUP (waves 0-3 / 4-7)
	__shared__ uint32_t gctr[2];                            // arrival counters of the two groups' barriers (monotonic)
	if (threadIdx.x < 2) gctr[threadIdx.x] = 0u;
	__shared__ uint32_t gstart;                             // group 0 -> group 1: "my first forward/dgrad block is done" (the two groups run half an iteration apart)
	if (threadIdx.x == 2) gstart = 0u;
	stage_weights32(wl, packed, NF32_ALL);
	const float *wb = wl + NF32_FWD * 256;
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	const int lane = threadIdx.x & 63, s = lane & 15, g = lane >> 4, w = threadIdx.x >> 6, half = w >> 2, wq = w & 3;
	const uint32_t n_ht = (lim + HT32 - 1) / HT32;
	const uint32_t K = blockIdx.x < n_ht ? (n_ht - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;      // half trips of this workgroup: blockIdx.x, + gridDim.x, ...
	const floatx4 z = {0.f, 0.f, 0.f, 0.f};
	floatx4 aV1[4] = {z, z, z, z}, aW0[2] = {z, z}, aV0[2] = {z, z}, aW1 = z, aV2 = z;            // this wave's ten weight-gradient tiles (summed over its half's samples)
	float lmax[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
	__syncthreads();
	struct Inputs { float f[8]; float d3[3]; float go[4]; };
	auto fetch = [&](uint32_t k, Inputs &in) {
		const uint32_t i = (blockIdx.x + k * gridDim.x) * HT32 + 16u * wq + s;
		const bool valid = i < lim;
		const uint32_t ic = valid ? i : lim - 1;
		load_feat32<LAYOUT>(feat, n, ic, g, in.f);
		in.d3[0] = dir[(size_t)ic * dir_stride]; in.d3[1] = dir[(size_t)ic * dir_stride + 1]; in.d3[2] = dir[(size_t)ic * dir_stride + 2];
		in.go[0] = in.go[1] = in.go[2] = in.go[3] = 0.f;
		if (valid) { const float4 v = *reinterpret_cast<const float4 *>(dout + (size_t)i * 4); in.go[0] = v.x; in.go[1] = v.y; in.go[2] = v.z; in.go[3] = v.w; }
	};
	// what role X leaves in registers for role Y
	Inputs cur;
	float sh[4] = {0.f, 0.f, 0.f, 0.f};
	Fwd32 st;
	floatx4 dO = z, dG1[4], dG0[4], dH[4], dD = z;
#pragma unroll
	for (int u = 0; u < 4; ++u) { st.h[u] = z; st.g0[u] = z; st.g1[u] = z; dG1[u] = z; dG0[u] = z; dH[u] = z; }
	st.den = z; st.rgb = z;
#pragma unroll
	for (int q = 0; q < 8; ++q) cur.f[q] = 0.f;
	if ((uint32_t)half < K) fetch((uint32_t)half, cur);           // half h does X on k = h, h + 2, ...
	const int o = lane & 15, col = 16 * wq + s;
	uint32_t gepoch = 0;
#define GROUP_BAR() do { gepoch += 4u; if (lane == 0) __hip_atomic_fetch_add(&gctr[half], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); \
	for (uint32_t spin_ = 0; __hip_atomic_load(&gctr[half], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < gepoch && spin_ < (1u << 24); ++spin_) __builtin_amdgcn_s_sleep(1); } while (0)   /* (bounded: a bug must not hang the GPU) */
	if (half == 1 && K >= 1u) for (uint32_t spin_ = 0; __hip_atomic_load(&gstart, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u && spin_ < (1u << 22); ++spin_) __builtin_amdgcn_s_sleep(4);      // start half an iteration late
	for (uint32_t it = (uint32_t)half; it < K; it += 2u) {
		{
			// ------------------------------------------------------------ forward recompute + dgrad of half trip `it`: 284 MFMAs per wave, no synchronisation at all
			const bool work = true;
			const uint32_t i = (blockIdx.x + it * gridDim.x) * HT32 + 16u * wq + s;
			const bool valid = work && i < lim;
			floatx4 acc[4] = {z, z, z, z};
			if (work) {
				sh4_32(cur.d3, g, sh);
#pragma unroll
				for (int kq = 0; kq < 2; ++kq) {                       // L0: 32 -> 64
					floatx4 a[4];
#pragma unroll
					for (int u = 0; u < 4; ++u) a[u] = ld_frag32(wl, 2 * u + kq, lane);
#pragma unroll
					for (int j = 0; j < 4; ++j)
#pragma unroll
						for (int u = 0; u < 4; ++u) acc[u] = MFMA32(a[u][j], cur.f[4 * kq + j], acc[u]);
				}
#pragma unroll
				for (int u = 0; u < 4; ++u) st.h[u] = relu4(acc[u]);
			}
			if (work) {
				floatx4 d0 = z, d1 = z;                                // L1: 64 -> 16
#pragma unroll
				for (int kq = 0; kq < 4; ++kq) {
					const floatx4 a = ld_frag32(wl, 8 + kq, lane);
					d0 = MFMA32(a[0], st.h[kq][0], d0); d1 = MFMA32(a[1], st.h[kq][1], d1);
					d0 = MFMA32(a[2], st.h[kq][2], d0); d1 = MFMA32(a[3], st.h[kq][3], d1);
				}
				st.den = d0 + d1;
			}
			if (work) {
#pragma unroll
				for (int u = 0; u < 4; ++u) acc[u] = z;
#pragma unroll
				for (int kq = 0; kq < 2; ++kq) {                       // L2: [density(16) | SH(16)] -> 64
					floatx4 a[4];
#pragma unroll
					for (int u = 0; u < 4; ++u) a[u] = ld_frag32(wl, 12 + 2 * u + kq, lane);
#pragma unroll
					for (int j = 0; j < 4; ++j) {
						const float b = kq == 0 ? st.den[j] : sh[j];
#pragma unroll
						for (int u = 0; u < 4; ++u) acc[u] = MFMA32(a[u][j], b, acc[u]);
					}
				}
#pragma unroll
				for (int u = 0; u < 4; ++u) st.g0[u] = relu4(acc[u]);
			}
			if (work) {
#pragma unroll
				for (int u = 0; u < 4; ++u) acc[u] = z;
#pragma unroll
				for (int kq = 0; kq < 4; ++kq) {                       // L3: 64 -> 64
					floatx4 a[4];
#pragma unroll
					for (int u = 0; u < 4; ++u) a[u] = ld_frag32(wl, 20 + 4 * u + kq, lane);
#pragma unroll
					for (int j = 0; j < 4; ++j)
#pragma unroll
						for (int u = 0; u < 4; ++u) acc[u] = MFMA32(a[u][j], st.g0[kq][j], acc[u]);
				}
#pragma unroll
				for (int u = 0; u < 4; ++u) st.g1[u] = relu4(acc[u]);
			}
			// (L4, the rgb output, is not needed by the backward pass: dL/d(rgb logits) comes in from the compositor)
			if (work) {
				dO = z;                                                // register j <-> gradient of output neuron 4g+j; only neurons 0..2 (g == 0) are non-zero
				if (g == 0) { dO[0] = cur.go[0]; dO[1] = cur.go[1]; dO[2] = cur.go[2]; }
				floatx4 a[4];
#pragma unroll
				for (int u = 0; u < 4; ++u) { a[u] = ld_frag32(wb, u, lane); dG1[u] = z; }
#pragma unroll
				for (int j = 0; j < 3; ++j)
#pragma unroll
					for (int u = 0; u < 4; ++u) dG1[u] = MFMA32(a[u][j], dO[j], dG1[u]);
#pragma unroll
				for (int u = 0; u < 4; ++u) dG1[u] = mask4(dG1[u], st.g1[u]);
			}
			if (work) {
#pragma unroll
				for (int u = 0; u < 4; ++u) dG0[u] = z;
#pragma unroll
				for (int t = 0; t < 2; ++t) {
					floatx4 a[4];
#pragma unroll
					for (int u = 0; u < 4; ++u) a[u] = ld_frag32(wb, 4 + 4 * u + t, lane);
#pragma unroll
					for (int j = 0; j < 4; ++j)
#pragma unroll
						for (int u = 0; u < 4; ++u) dG0[u] = MFMA32(a[u][j], dG1[t][j], dG0[u]);
				}
			}
			if (work) {
#pragma unroll
				for (int t = 2; t < 4; ++t) {
					floatx4 a[4];
#pragma unroll
					for (int u = 0; u < 4; ++u) a[u] = ld_frag32(wb, 4 + 4 * u + t, lane);
#pragma unroll
					for (int j = 0; j < 4; ++j)
#pragma unroll
						for (int u = 0; u < 4; ++u) dG0[u] = MFMA32(a[u][j], dG1[t][j], dG0[u]);
				}
#pragma unroll
				for (int u = 0; u < 4; ++u) dG0[u] = mask4(dG0[u], st.g0[u]);
			}
			if (work) {
				floatx4 d0 = z, d1 = z;
#pragma unroll
				for (int t = 0; t < 4; ++t) {
					const floatx4 a = ld_frag32(wb, 20 + t, lane);
					d0 = MFMA32(a[0], dG0[t][0], d0); d1 = MFMA32(a[1], dG0[t][1], d1);
					d0 = MFMA32(a[2], dG0[t][2], d0); d1 = MFMA32(a[3], dG0[t][3], d1);
				}
				dD = d0 + d1;
				if (g == 0) dD[0] += cur.go[3];                       // out[:,3] = den[:,0]  (ngp_network.py:83)
				floatx4 a[4];
#pragma unroll
				for (int u = 0; u < 4; ++u) { a[u] = ld_frag32(wb, 24 + u, lane); dH[u] = z; }
#pragma unroll
				for (int j = 0; j < 4; ++j)
#pragma unroll
					for (int u = 0; u < 4; ++u) dH[u] = MFMA32(a[u][j], dD[j], dH[u]);
#pragma unroll
				for (int u = 0; u < 4; ++u) dH[u] = mask4(dH[u], st.h[u]);
			}
			floatx4 dF[2] = {z, z};
			if (work) {
#pragma unroll
				for (int t = 0; t < 2; ++t) {
					const floatx4 a0 = ld_frag32(wb, 28 + t, lane), a1 = ld_frag32(wb, 32 + t, lane);
#pragma unroll
					for (int j = 0; j < 4; ++j) { dF[0] = MFMA32(a0[j], dH[t][j], dF[0]); dF[1] = MFMA32(a1[j], dH[t][j], dF[1]); }
				}
			}
			if (work) {
#pragma unroll
				for (int t = 2; t < 4; ++t) {
					const floatx4 a0 = ld_frag32(wb, 28 + t, lane), a1 = ld_frag32(wb, 32 + t, lane);
#pragma unroll
					for (int j = 0; j < 4; ++j) { dF[0] = MFMA32(a0[j], dH[t][j], dF[0]); dF[1] = MFMA32(a1[j], dH[t][j], dF[1]); }
				}
				if (valid) {                                          // feature 16u+4g+r  ->  level 8u+2g+(r>>1), component r&1
#pragma unroll
					for (int u = 0; u < 2; ++u)
#pragma unroll
						for (int pr = 0; pr < 2; ++pr) {
							const float2 v = make_float2(dF[u][2 * pr], dF[u][2 * pr + 1]);
							const uint32_t level = 8 * u + 2 * g + pr;
							lmax[u][pr] = fmaxf(lmax[u][pr], fmaxf(fabsf(v.x), fabsf(v.y)));
							if (LAYOUT == NGP_LAYOUT_SOA) reinterpret_cast<float2 *>(dfeat)[(size_t)level * n + i] = v;
							else *reinterpret_cast<float2 *>(dfeat + (size_t)i * 32 + 2 * level) = v;
						}
				}
			}
		}
		if (half == 0 && it == 0u && lane == 0 && wq == 0) __hip_atomic_store(&gstart, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		{
			// ------------------------------------------------------------ weight gradients of the same half trip: five phases through the GROUP's staging region, barriers among the group's four waves only
			const bool work = true;
			// phase A: dG1 rows 0..63 | G0 rows 64..127 -> V1: wave wq owns output tile `wq` against the four input tiles
			if (work) { st_tiles_h(stage, 0, col, g, dG1); st_tiles_h(stage, 64, col, g, st.g0); }
			GROUP_BAR();                                                  // 1
			if (work) {
#pragma unroll
				for (int ti = 0; ti < 4; ++ti) aV1[ti] = wgrad_tile_h(stage, 16 * wq, 64 + 16 * ti, o, g, aV1[ti]);
			}
			GROUP_BAR();                                                  // 2
			// phase B1: dH 0..63 | F 64..95 -> W0
			if (work) {
				st_tiles_h(stage, 0, col, g, dH);
#pragma unroll
				for (int q = 0; q < 8; ++q) stage[(64 + 8 * g + q) * RSH32 + col] = cur.f[q];
			}
			GROUP_BAR();                                                  // 3
			if (work) {
				aW0[0] = wgrad_tile_h(stage, 16 * wq, 64, o, g, aW0[0]);
				aW0[1] = wgrad_tile_h(stage, 16 * wq, 80, o, g, aW0[1]);
			}
			GROUP_BAR();                                                  // 4
			// phase B2: dG0 0..63 | IN2 = [density(16) | SH(16)] 64..95 -> V0
			if (work) {
				st_tiles_h(stage, 0, col, g, dG0);
#pragma unroll
				for (int r = 0; r < 4; ++r) { stage[(64 + 4 * g + r) * RSH32 + col] = st.den[r]; stage[(80 + 4 * g + r) * RSH32 + col] = sh[r]; }
			}
			GROUP_BAR();                                                  // 5
			if (work) {
				aV0[0] = wgrad_tile_h(stage, 16 * wq, 64, o, g, aV0[0]);
				aV0[1] = wgrad_tile_h(stage, 16 * wq, 80, o, g, aV0[1]);
			}
			GROUP_BAR();                                                  // 6
			// phase C1: dD 0..15 | H 16..79 -> W1: input tile wq
			if (work) {
#pragma unroll
				for (int r = 0; r < 4; ++r) stage[(4 * g + r) * RSH32 + col] = dD[r];
				st_tiles_h(stage, 16, col, g, st.h);
			}
			GROUP_BAR();                                                  // 7
			if (work) aW1 = wgrad_tile_h(stage, 0, 16 + 16 * wq, o, g, aW1);
			GROUP_BAR();                                                  // 8
			// phase C2: dO 0..15 | G1 16..79 -> V2
			if (work) {
#pragma unroll
				for (int r = 0; r < 4; ++r) stage[(4 * g + r) * RSH32 + col] = dO[r];
				st_tiles_h(stage, 16, col, g, st.g1);
			}
			GROUP_BAR();                                                  // 9
			if (work) aV2 = wgrad_tile_h(stage, 0, 16 + 16 * wq, o, g, aV2);
			if (it + 2u < K) fetch(it + 2u, cur);                     // inputs of this group's next half trip
			GROUP_BAR();                                                  // 10
		}
	}
	// ---- the two halves hold partial sums of the same ten tiles per wave index: half 1 hands its sums over through LDS (the fragment region is free now), half 0
	// adds and writes the workgroup's slab, packed like the weights (wd part 0..3071, wc part 3072..10239); C rows = 4g+r, cols = lane&15
	float *xch = wl + (size_t)wq * 10 * 256;                    // [wave][10 tiles][64 lanes][4]
	auto put = [&](int tile, const floatx4 &v) { *reinterpret_cast<floatx4 *>(xch + tile * 256 + lane * 4) = v; };
	auto get = [&](int tile) { return *reinterpret_cast<const floatx4 *>(xch + tile * 256 + lane * 4); };
	__syncthreads();
	if (half == 1) {
#pragma unroll
		for (int t = 0; t < 4; ++t) put(t, aV1[t]);
		put(4, aW0[0]); put(5, aW0[1]); put(6, aV0[0]); put(7, aV0[1]); put(8, aW1); put(9, aV2);
	}
	__syncthreads();
	if (half == 0) {
#pragma unroll
		for (int t = 0; t < 4; ++t) aV1[t] += get(t);
		aW0[0] += get(4); aW0[1] += get(5); aV0[0] += get(6); aV0[1] += get(7); aW1 += get(8); aV2 += get(9);
		float *slab = slabs + (size_t)blockIdx.x * 10240;
		const int ci = lane & 15;
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int ro = 4 * g + r;
#pragma unroll
			for (int ti = 0; ti < 4; ++ti) slab[3072 + 2048 + (16 * wq + ro) * 64 + 16 * ti + ci] = aV1[ti][r];
#pragma unroll
			for (int tj = 0; tj < 2; ++tj) { slab[(16 * wq + ro) * 32 + 16 * tj + ci] = aW0[tj][r]; slab[3072 + (16 * wq + ro) * 32 + 16 * tj + ci] = aV0[tj][r]; }
			slab[2048 + ro * 64 + 16 * wq + ci] = aW1[r];
			slab[3072 + 6144 + ro * 64 + 16 * wq + ci] = aV2[r];
		}
	}
	if (am.parts) { __syncthreads(); absmax_epilogue(am, lmax, smem32 + NF32_ALL * 256, 8); }      // (one scratch for both groups: group 0's staging region)
}

#undef GROUP_BAR

// ---------------------------------------------------------------------------------------------------------------- fused tail of the fp32 step (r3)
// Adam+EMA sweep of the flat weight pack (10240 floats, EMA aliasing the parameter like k_adam_ema<float, 2>) and the MFMA fragments of the UPDATED weights for the
// next iteration, one single-workgroup launch instead of k_adam_ema (pack) + next step's k_pack_frags32: two launches and their boundaries less per iteration.
__global__ __launch_bounds__(1024) void k_mlp32_sweep_pack(float *__restrict__ pack, const float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v, AdamConsts c,
                                                           float *__restrict__ packed_out, const uint16_t *__restrict__ table) {
	__shared__ float w[10240];
	tail_mlp32_sweep_pack_1024(pack, grad, m, v, c, packed_out, w, table);       // (mlp_tail.h: the same job rides in k_bin_pairs' grid on the single-GPU training path)
}
// the slot -> weight table: the layout functions evaluated on a ramp (weight i holds the value i + 1, exact in fp32; a constant-zero slot reads 0)
__global__ void k_pack_table_ramp(float *ramp) { const int i = blockIdx.x * 256 + threadIdx.x; if (i < 10240) ramp[i] = (float)(i + 1); }
__global__ void k_pack_table_build(const float *__restrict__ ramp, uint16_t *__restrict__ table) {
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx < PACK_TABLE_32) {
		const int f = idx >> 8, lane = (idx >> 2) & 63, j = idx & 3;
		table[idx] = (uint16_t)frag_value32(ramp, ramp + 3072, f, lane & 15, lane >> 4, j);
	} else if (idx < PACK_TABLE_32 + PACK_TABLE_SPLIT) {
		const int r = idx - PACK_TABLE_32, f = r >> 9, lane = (r >> 3) & 63, j = r & 7;
		table[idx] = (uint16_t)split_frag_weight(ramp, ramp + 3072, f, lane & 15, lane >> 4, j);
	}
}
const uint16_t *ngp_mlp32_pack_table(void *stream) {
	static std::mutex mu;
	static uint16_t *tab[64] = {nullptr};
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
	std::lock_guard<std::mutex> lk(mu);
	if (tab[dev]) return tab[dev];
	float *ramp = nullptr; uint16_t *t = nullptr;
	const uint32_t n = PACK_TABLE_32 + PACK_TABLE_SPLIT;
	if (hipMalloc((void **)&ramp, 10240 * sizeof(float)) != hipSuccess || hipMalloc((void **)&t, n * sizeof(uint16_t)) != hipSuccess) { ngp_set_error("ngp_mlp32_pack_table: hipMalloc failed"); if (ramp) (void)hipFree(ramp); return nullptr; }
	hipStream_t s = (hipStream_t)stream;
	hipLaunchKernelGGL(k_pack_table_ramp, dim3(40), dim3(256), 0, s, ramp);
	hipLaunchKernelGGL(k_pack_table_build, dim3(div_up(n, 256)), dim3(256), 0, s, (const float *)ramp, t);
	// built once per device: wait for it, so that every stream may use the table from here on, and release the ramp
	if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) { ngp_set_error("ngp_mlp32_pack_table: building the table failed"); (void)hipFree(ramp); (void)hipFree(t); return nullptr; }
	(void)hipFree(ramp);
	tab[dev] = t;
	return t;
}
int ngp_mlp32_sweep_pack(void *stream, float *pack, const float *grad, float *m, float *v, float lr, float beta0, float beta1, float eps, uint32_t step, float ema_decay, float *packed_out) {
	NGP_REQUIRE(pack && grad && m && v && packed_out && step >= 1, NGP_E_ARG, "ngp_mlp32_sweep_pack: bad arguments");
	const uint16_t *table = ngp_mlp32_pack_table(stream);
	if (!table) return NGP_E_ARG;
	const AdamConsts c = adam_consts(lr, beta0, beta1, eps, step, ema_decay, 1.0f);
	NGP_LAUNCH(k_mlp32_sweep_pack, dim3(1), dim3(1024), 0, (hipStream_t)stream, pack, grad, m, v, c, packed_out, table);
	NGP_LAUNCH_CHECK("ngp_mlp32_sweep_pack");
	return 0;
}

// ---------------------------------------------------------------------------------------------------------------- C ABI
static int check_field32(const char *fn, const void *feat, const void *wd, const void *wc, int layout) {
	const bool prepacked = (layout & NGP_WEIGHTS_PACKED) != 0;
	layout &= ~NGP_WEIGHTS_PACKED;
	NGP_REQUIRE(feat && wd && (wc || prepacked), NGP_E_ARG, "%s: null pointer", fn);
	NGP_REQUIRE(!prepacked || ((uintptr_t)wd & 15) == 0, NGP_E_ALIGN, "%s: packed weight buffer must be 16-byte aligned", fn);
	NGP_REQUIRE(layout == NGP_LAYOUT_AOS || layout == NGP_LAYOUT_SOA, NGP_E_ARG, "%s: bad layout %d", fn, layout);
	NGP_REQUIRE(((uintptr_t)feat & 15) == 0, NGP_E_ALIGN, "%s: feature pointer must be 16-byte aligned", fn);
	return 0;
}
// forward kernel: split fp16 operands on v_mfma_f32_16x16x32_f16 (field_split.hip, default) | exact fp32 products on v_mfma_f32_16x16x4_f32 (NGP_FIELD32_FWD=mfma32)
static int g_fwd_split = -1, g_bwd_variant = -1;     // -1: not decided yet (environment, then the default); set by ngp_field32_select
static bool fwd_split() {
	if (g_fwd_split < 0) { const char *e = getenv("NGP_FIELD32_FWD"); g_fwd_split = (e && (e[0] == 'm' || e[0] == '0')) ? 0 : 1; }
	return g_fwd_split == 1;
}
static bool dens_split() {       // (probe hook: NGP_DENSITY32_FWD=mfma32|split selects the density-only forward independently)
	static int v = -1;
	if (v < 0) { const char *e = getenv("NGP_DENSITY32_FWD"); v = !e ? -2 : ((e[0] == 'm' || e[0] == '0') ? 0 : 1); }
	return v == -2 ? fwd_split() : v == 1;
}
static int bwd_variant() {
	if (g_bwd_variant < 0) { const char *e = getenv("NGP_FIELD32_BWD"); g_bwd_variant = e ? atoi(e) : NGP_FIELD32_BWD_DEFAULT; }
	return g_bwd_variant;
}
// Which kernels ngp_field32_fwd / ngp_density32_fwd / ngp_field32_bwd launch from now on in this process: exact = 1 -> the exact-product fp32-MFMA kernels (no operand
// range), exact = 0 -> the split-operand kernels (default).  Both fragment sets are always present in a packed weight buffer.  Returns the previous setting.
NGP_API int ngp_field32_select(int exact) {
	const int was = (fwd_split() ? 0 : 1);
	// the choice the environment made (NGP_FIELD32_FWD / NGP_FIELD32_BWD, else the split default), remembered at the first call: select(0) goes back to THAT, not to a
	// hard-coded pair (a process started on the exact-product kernels was silently moved to the split ones by a select(1) ... select(0) round trip)
	static const int env_fwd = fwd_split() ? 1 : 0, env_bwd = bwd_variant();
	if (exact) { g_fwd_split = 0; g_bwd_variant = 2; }
	else { g_fwd_split = env_fwd; g_bwd_variant = env_bwd; }
	return was;
}
// n_frags fp32 fragments (n_frags < 0: the first -n_frags split fp16 fragments of the forward instead) of raw weight packs in a per-(device, stream) scratch, or the caller's packed buffer
static const float *pack_weights32(const char *fn, hipStream_t s, const void *wd, const void *wc, int n_frags, int layout_flags) {
	if (layout_flags & NGP_WEIGHTS_PACKED) return (const float *)wd;
	static std::mutex mu;
	static std::map<std::pair<int, hipStream_t>, float *> pool;
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) { ngp_set_error("%s: hipGetDevice failed", fn); return nullptr; }
	float *buf;
	{
		std::lock_guard<std::mutex> lk(mu);
		float *&slot = pool[{dev, s}];
		if (!slot) { hipError_t e = hipMalloc((void **)&slot, (size_t)NGP_PACKED32_WEIGHT_FLOATS * sizeof(float)); if (e != hipSuccess) { slot = nullptr; ngp_set_error("%s: hipMalloc(fragment scratch): %s", fn, hipGetErrorString(e)); return nullptr; } }
		buf = slot;
	}
	if (n_frags < 0) { if (ngp_field32_pack_split(s, (const float *)wd, (const float *)wc, buf + NF32_ALL * 256, -n_frags)) return nullptr; return buf; }
	NGP_LAUNCH(k_pack_frags32, dim3(div_up((uint32_t)n_frags * 256u, 256u)), dim3(256), 0, s, (const float *)wd, (const float *)wc, buf, n_frags);
	return buf;
}
static uint32_t fwd32_grid(uint32_t n) { uint32_t b = div_up(div_up(n, 16), 4); return b < 1024 ? (b ? b : 1) : 1024; }

NGP_API int ngp_field32_pack_weights(void *stream, const float *wd, const float *wc, float *packed_out) {
	NGP_REQUIRE(wd && wc && packed_out, NGP_E_ARG, "ngp_field32_pack_weights: null pointer");
	NGP_REQUIRE(((uintptr_t)packed_out & 15) == 0, NGP_E_ALIGN, "ngp_field32_pack_weights: output must be 16-byte aligned");
	NGP_LAUNCH(k_pack_frags32, dim3(div_up((uint32_t)NF32_ALL * 256u, 256u)), dim3(256), 0, (hipStream_t)stream, wd, wc, packed_out, NF32_ALL);
	NGP_LAUNCH_CHECK("ngp_field32_pack_weights");
	return ngp_field32_pack_split(stream, wd, wc, packed_out + NF32_ALL * 256, NSPLIT_FRAGS);
}
NGP_API int ngp_field32_fwd(void *stream, uint32_t n, const float *feat, int layout, const float *dir, uint32_t dir_stride, const float *wd, const float *wc,
                            float *out, const uint32_t *n_valid) {
	int rc = check_field32("ngp_field32_fwd", feat, wd, wc, layout); if (rc) return rc;
	const int layout_flags = layout; layout &= ~NGP_WEIGHTS_PACKED;
	NGP_REQUIRE(dir && out && dir_stride >= 3, NGP_E_ARG, "ngp_field32_fwd: bad dir/out");
	if (n == 0) return 0;
	hipStream_t s = (hipStream_t)stream;
	const float *packed = pack_weights32("ngp_field32_fwd", s, wd, wc, fwd_split() ? -NSPLIT_FRAGS : NF32_FWD, layout_flags); if (!packed) return NGP_E_ARG;
	if (fwd_split()) return ngp_field32_fwd_split(stream, n, feat, layout, dir, dir_stride, packed + NF32_ALL * 256, out, n_valid, 0);
	const dim3 grid(fwd32_grid(n)), block(256);
	if (layout == NGP_LAYOUT_SOA) NGP_LAUNCH((k_field32_fwd<NGP_LAYOUT_SOA, false>), grid, block, 0, s, n, feat, dir, dir_stride, packed, out, n_valid);
	else NGP_LAUNCH((k_field32_fwd<NGP_LAYOUT_AOS, false>), grid, block, 0, s, n, feat, dir, dir_stride, packed, out, n_valid);
	NGP_LAUNCH_CHECK("ngp_field32_fwd");
	return 0;
}
NGP_API int ngp_density32_fwd(void *stream, uint32_t n, const float *feat, int layout, const float *wd, float *out) {
	int rc = check_field32("ngp_density32_fwd", feat, wd, wd, layout); if (rc) return rc;
	const int layout_flags = layout; layout &= ~NGP_WEIGHTS_PACKED;
	NGP_REQUIRE(out, NGP_E_ARG, "ngp_density32_fwd: null out");
	if (n == 0) return 0;
	hipStream_t s = (hipStream_t)stream;
	const float *packed = pack_weights32("ngp_density32_fwd", s, wd, wd, dens_split() ? -6 : 12, layout_flags); if (!packed) return NGP_E_ARG;
	if (dens_split()) return ngp_field32_fwd_split(stream, n, feat, layout, nullptr, 3u, packed + NF32_ALL * 256, out, nullptr, 1);
	const dim3 grid(fwd32_grid(n)), block(256);
	if (layout == NGP_LAYOUT_SOA) NGP_LAUNCH((k_field32_fwd<NGP_LAYOUT_SOA, true>), grid, block, 0, s, n, feat, (const float *)nullptr, 3u, packed, out, (const uint32_t *)nullptr);
	else NGP_LAUNCH((k_field32_fwd<NGP_LAYOUT_AOS, true>), grid, block, 0, s, n, feat, (const float *)nullptr, 3u, packed, out, (const uint32_t *)nullptr);
	NGP_LAUNCH_CHECK("ngp_density32_fwd");
	return 0;
}
NGP_API int ngp_field32_bwd_slabs(uint32_t n) { uint32_t b = div_up(n, BT32); return (int)(b < 256 ? (b ? b : 1) : 256); }
NGP_API int ngp_field32_bwd(void *stream, uint32_t n, const float *feat, int layout, const float *dir, uint32_t dir_stride, const float *wd, const float *wc,
                            const float *dLdout, float *dLdfeat, float *wgrad_slabs, uint32_t n_slabs, const uint32_t *n_valid) {
	return ngp_field32_bwd_am(stream, n, feat, layout, dir, dir_stride, wd, wc, dLdout, dLdfeat, wgrad_slabs, n_slabs, n_valid, nullptr);
}
int ngp_field32_bwd_am(void *stream, uint32_t n, const float *feat, int layout, const float *dir, uint32_t dir_stride, const float *wd, const float *wc,
                       const float *dLdout, float *dLdfeat, float *wgrad_slabs, uint32_t n_slabs, const uint32_t *n_valid, const AbsmaxOut *am_in) {
	const AbsmaxOut am = am_in ? *am_in : AbsmaxOut{nullptr, nullptr, 0u, nullptr};
	int rc = check_field32("ngp_field32_bwd", feat, wd, wc, layout); if (rc) return rc;
	const int layout_flags = layout; layout &= ~NGP_WEIGHTS_PACKED;
	NGP_REQUIRE(dir && dLdout && dLdfeat && wgrad_slabs && dir_stride >= 3, NGP_E_ARG, "ngp_field32_bwd: null pointer");
	NGP_REQUIRE((int)n_slabs == ngp_field32_bwd_slabs(n), NGP_E_ARG, "ngp_field32_bwd: n_slabs %u != ngp_field32_bwd_slabs(%u)", n_slabs, n);
	if (n == 0) return 0;
	const dim3 grid(n_slabs), block(512);
	hipStream_t s = (hipStream_t)stream;
	// 3 = split fp16 operands on the fp16 matrix cores (field_split.hip, r3: the default)
	const int variant = bwd_variant();
	const float *packed = pack_weights32("ngp_field32_bwd", s, wd, wc, variant == 3 ? -NSPLIT_FRAGS : NF32_ALL, layout_flags); if (!packed) return NGP_E_ARG;
	if (variant == 3) return ngp_field32_bwd_split(stream, n, feat, layout, dir, dir_stride, packed + NF32_ALL * 256, dLdout, dLdfeat, wgrad_slabs, n_slabs, n_valid, am_in);
	// anything else = the exact-product kernel: two free-running groups on v_mfma_f32_16x16x4_f32 (r3: 138 us; the fallback ngp_field32_select(1) switches to)
	const size_t shmem_2g = ((size_t)NF32_ALL * 256 + (size_t)2 * 128 * RSH32) * sizeof(float);
#define GO2G(L) do { \
	static bool attr_set = false; \
	if (!attr_set) { hipError_t e = hipFuncSetAttribute((const void *)k_field32_bwd_2g<L>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem_2g); \
		if (e != hipSuccess) { ngp_set_error("ngp_field32_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; } attr_set = true; } \
	NGP_LAUNCH((k_field32_bwd_2g<L>), grid, block, shmem_2g, s, n, feat, dir, dir_stride, packed, dLdout, dLdfeat, wgrad_slabs, n_valid, am); } while (0)
	if (layout == NGP_LAYOUT_SOA) GO2G(NGP_LAYOUT_SOA); else GO2G(NGP_LAYOUT_AOS);
#undef GO2G
	NGP_LAUNCH_CHECK("ngp_field32_bwd");
	return 0;
}
