// (r6) The two small launches at the end of the fp32 configuration's backward - the slab reduction of the MLP weight gradients (k_reduce_slabs: 160 workgroups, ~10 us)
// and the Adam + EMA sweep of the 10240-float weight pack with the fragment packing (k_mlp32_sweep_pack: ONE workgroup, ~12 us) - as device functions, so that they can
// RIDE in the grid of a launch that is in the stream anyway: kernel boundaries are at the hardware's floor (1.2 us), so the only way left to shorten the main stream is
// fewer and fuller launches.  Both jobs are tiny grids that leave the chip idle; as extra workgroups of the hash backward's record kernels (k_bin_runs2: 256 threads,
// 26 KB of LDS; k_bin_pairs: 1024 threads, 66 KB) they run beside ~1300 latency-bound workgroups instead of in front of them.  Same arithmetic in the same order as the
// standalone kernels (which remain: other precisions, data parallel, level tables without both record kernels) - bit-identical results.
#pragma once
#include "ngp_common.h"
#include "field_split.h"
#include <string.h>

#define NF32_FWD 40
#define NF32_BWD 36
#define NF32_ALL (NF32_FWD + NF32_BWD)          // fp32 fragments; the packed buffer continues with the split fp16 fragments (field_split.h: forward + transposed): NGP_PACKED32_WEIGHT_FLOATS = NF32_ALL * 256 + NSPLIT_HALVES / 2
static_assert(NGP_PACKED32_WEIGHT_FLOATS == NF32_ALL * 256 + NSPLIT_HALVES / 2, "packed fp32 weight buffer layout");

// value j (0..3) of weight fragment f for lane (s = lane&15: row of the A tile, g = lane>>4: k index of the MFMA).  fp32 packs, (out,in) row-major:
// wd: W0 @0 [64][32], W1 @2048 [16][64];  wc: V0 @0 [64][32], V1 @2048 [64][64], V2 @6144 [16][64]   (ngp_network.py:21-29)
__device__ __forceinline__ float frag_value32(const float *__restrict__ wd, const float *__restrict__ wc, int f, int s, int g, int j) {
	if (f < 8) { const int u = f >> 1, kq = f & 1; return wd[(16 * u + s) * 32 + 8 * g + 4 * kq + j]; }                         // L0: lane group g holds features 8g..8g+7
	if (f < 12) { const int kq = f - 8; return wd[2048 + s * 64 + 16 * kq + 4 * g + j]; }                                        // L1
	if (f < 20) { const int u = (f - 12) >> 1, kq = (f - 12) & 1; return wc[(16 * u + s) * 32 + 16 * kq + 4 * g + j]; }           // L2: input = [density(16) | SH(16)]
	if (f < 36) { const int u = (f - 20) >> 2, kq = (f - 20) & 3; return wc[2048 + (16 * u + s) * 64 + 16 * kq + 4 * g + j]; }    // L3
	if (f < 40) { const int kq = f - 36; return wc[6144 + s * 64 + 16 * kq + 4 * g + j]; }                                       // L4
	f -= 40;                                                                                                                   // backward: A = W^T
	if (f < 4) return wc[6144 + (4 * g + j) * 64 + 16 * f + s];                                                                 // dG1 = V2^T dO
	if (f < 20) { const int u = (f - 4) >> 2, t = (f - 4) & 3; return wc[2048 + (16 * t + 4 * g + j) * 64 + 16 * u + s]; }       // dG0 = V1^T dG1
	if (f < 24) { const int t = f - 20; return wc[(16 * t + 4 * g + j) * 32 + s]; }                                             // dD  = (V0^T dG0)[0:16]
	if (f < 28) { const int u = f - 24; return wd[2048 + (4 * g + j) * 64 + 16 * u + s]; }                                      // dH  = W1^T dD
	{ const int u = (f - 28) >> 2, t = (f - 28) & 3; return wd[(16 * t + 4 * g + j) * 32 + 16 * u + s]; }                        // dF  = W0^T dH
}

static inline TailJobs no_tail_jobs() { TailJobs t; memset(&t, 0, sizeof(t)); return t; }
#define TAIL_REDUCE_COLS 64u
#define TAIL_REDUCE_LDS_FLOATS (16u * 65u)

// Adam + EMA of the parameter that column `col` of the flat weight gradient belongs to (its gradient: the column sum t); fp16 configuration, two packs with their fp16 shadows
__device__ __forceinline__ void pack_sweep_column(const PackSweep &a, const PackSweep &b, const AdamConsts &c, uint32_t col, float t) {
	const PackSweep &w = col >= b.begin ? b : a;
	if (col >= w.begin && col - w.begin < w.count) {
		const uint32_t e = col - w.begin;
		float P = w.p[e], M = w.m[e], V = w.v[e], E = 0.f;
		const bool has_ema = w.ema != nullptr, alias = w.ema == w.p;
		if (has_ema) E = alias ? P : w.ema[e];
		if (has_ema) adam_ema_update<true>(P, M, V, E, t, c); else adam_ema_update<false>(P, M, V, E, t, c);
		w.p[e] = P; w.m[e] = M; w.v[e] = V;
		if (has_ema && !alias) w.ema[e] = E;
		if (w.p_half) w.p_half[e] = __float2half_rn(P);
	}
}

// the slab reduction for columns [64 unit, 64 unit + 64) by ONE 256-thread workgroup: thread = (column, quarter q); partial sums of slab groups q, q + 4, q + 8, q + 12 of
// k_reduce_slabs' sixteen (independent chains: four loads in flight), then the sixteen partials in its order.  lds: TAIL_REDUCE_LDS_FLOATS floats.
__device__ __forceinline__ void tail_reduce_slabs_256(const TailJobs &tj, float *lds, uint32_t unit) {
	float (*part)[65] = reinterpret_cast<float (*)[65]>(lds);
	const uint32_t lc = threadIdx.x & 63u, q = threadIdx.x >> 6, col = unit * TAIL_REDUCE_COLS + lc;
	float s[4] = {0.f, 0.f, 0.f, 0.f};
	if (col < tj.width) {
		for (uint32_t k0 = 0; k0 < tj.n_slabs; k0 += 16u) {
#pragma unroll
			for (uint32_t u = 0; u < 4; ++u) { const uint32_t k = k0 + q + 4u * u; if (k < tj.n_slabs) s[u] += tj.slabs[(size_t)k * tj.width + col]; }
		}
	}
#pragma unroll
	for (uint32_t u = 0; u < 4; ++u) part[q + 4u * u][lc] = s[u];
	__syncthreads();
	if (q == 0 && col < tj.width) {
		float t = 0.f;
#pragma unroll
		for (int g = 0; g < 16; ++g) t += part[g][lc];
		tj.reduce_out[col] = t;
		if (tj.do_sweep16) pack_sweep_column(tj.a16, tj.b16, tj.c, col, t);          // fp16 configuration: k_reduce_slabs_sweep's second half
	}
}

#define PACK_TABLE_32 (NF32_ALL * 256)
#define PACK_TABLE_SPLIT (NSPLIT_FRAGS * 512)
// k_mlp32_sweep_pack's job by ONE 1024-thread workgroup; w: 10240 floats of LDS; table: ngp_mlp32_pack_table()
__device__ __forceinline__ void tail_mlp32_sweep_pack_1024(float *__restrict__ pack, const float *__restrict__ grad, float *__restrict__ m, float *__restrict__ v, const AdamConsts &c,
                                                           float *__restrict__ packed_out, float *w, const uint16_t *__restrict__ table) {
	// ten elements per thread: all forty loads of a thread are issued before the first use (a loop of dependent load -> update -> store round trips made this 15 us long for 160 KB of traffic)
	float P[10], M[10], V[10], G[10];
#pragma unroll
	for (int k = 0; k < 10; ++k) { const int i = threadIdx.x + 1024 * k; P[k] = pack[i]; M[k] = m[i]; V[k] = v[i]; G[k] = grad[i]; }
#pragma unroll
	for (int k = 0; k < 10; ++k) {
		const int i = threadIdx.x + 1024 * k;
		float E = P[k];
		adam_ema_update<true>(P[k], M[k], V[k], E, G[k], c);
		pack[i] = P[k]; m[i] = M[k]; v[i] = V[k]; w[i] = P[k];
	}
	__syncthreads();
	// (r6) the fragments of the UPDATED weights, gathered through the slot -> weight table: the index arithmetic of frag_value32 / split_frag_weight (~60 instructions per
	// slot, 105 k slots) made this ONE workgroup VALU-bound - 12 us alone, ~25 us beside the record workgroups it now shares a CU with
#pragma unroll
	for (int k = 0; k < PACK_TABLE_32 / 1024; ++k) {          // 19 rounds: table loads and stores coalesced, the LDS gathers independent
		const int idx = threadIdx.x + 1024 * k;
		const uint32_t src = table[idx];
		packed_out[idx] = src ? w[src - 1u] : 0.f;
	}
	_Float16 *split_out = reinterpret_cast<_Float16 *>(packed_out + NF32_ALL * 256);      // the split fp16 fragments (field_split.h): h part, then the m part
#pragma unroll
	for (int k = 0; k < PACK_TABLE_SPLIT / 1024; ++k) {       // 21 rounds: one gather serves a slot's h and m half
		const int r = threadIdx.x + 1024 * k;
		const uint32_t src = table[PACK_TABLE_32 + r];
		const float wv = src ? w[src - 1u] : 0.f;
		const _Float16 h = (_Float16)wv;
		split_out[r] = h; split_out[PACK_TABLE_SPLIT + r] = (_Float16)((wv - (float)h) * SPLIT_SCALE);
	}
}
