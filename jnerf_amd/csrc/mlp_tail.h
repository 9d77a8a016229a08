// (r6) The two small launches at the end of the fp32 configuration's backward - the slab reduction of the MLP weight gradients (k_reduce_slabs: 160 workgroups, ~10 us)
// and the Adam + EMA sweep of the 10240-float weight pack with the fragment packing (k_mlp32_sweep_pack: ONE workgroup, ~12 us) - as device functions, so that they can
// RIDE in launches that are in the stream anyway: kernel boundaries are at the hardware's floor (1.2 us), so the only way left to shorten the main stream is fewer and fuller
// launches.  First form (profiles/r06b_ab_lines.txt: +2.6 % it/s): an extra ROW of workgroups in front of the hash backward's two record kernels.  But those kernels are
// tuned to be exactly resident (k_bin_runs2: 1280 workgroups = 5 per CU), so 128 extra workgroups of ~15 us pushed 128 real ones into a second round (41 -> 52 us alone), and
// the single sweep-and-pack workgroup sharing a CU with record workgroups ran 2-3x its stand-alone time (k_bin_pairs 27 -> 42 us: profiles/r06d_lego_timeline.txt).  Second
// form (this file): the jobs are DISTRIBUTED over the record workgroups themselves -
//   every workgroup of the run kernel reduces 8 columns of the slabs (k_reduce_slabs' order: 16 partial sums over slabs k, k + 16, ..., then their sum in order) and the
//   thread that holds a column's sum applies Adam + EMA to the parameter it is the gradient of (fp32: the flat pack; fp16: the two packs and their shadows);
//   every workgroup of the edge kernel - the NEXT launch, so every parameter is updated - gathers 32 slots of the fragment buffer through the slot -> weight table.
// One dependent load round at the top of each workgroup instead of rows of foreign workgroups.  Same arithmetic in the same order as the stand-alone kernels (which
// remain: data parallel, level tables without both record kernels, the per-stage API) - bit-identical results.
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

// Adam + EMA of element `col` of the flat fp32 weight pack (EMA aliasing the parameter): k_mlp32_sweep_pack's update of that element
__device__ __forceinline__ void pack32_sweep_column(const TailJobs &tj, uint32_t col, float t) {
	float P = tj.pack[col], M = tj.m[col], V = tj.v[col], E = P;
	adam_ema_update<true>(P, M, V, E, t, tj.c);
	tj.pack[col] = P; tj.m[col] = M; tj.v[col] = V;
}
// the slab reduction for columns [8 unit, 8 unit + 8) by the first 128 threads of a workgroup (any size >= 128): thread = (column c, slab group q of k_reduce_slabs' sixteen);
// a group's slabs q, q + 16, ... are loaded together and added in order, then the sixteen partials in order - k_reduce_slabs' sums bit for bit.  The thread that ends up with a
// column's sum stores it and (do_sweep / do_sweep16) updates the parameter it is the gradient of.  lds: TAIL_REDUCE_LDS_FLOATS floats; contains one __syncthreads().
#define TAIL_REDUCE_COLS 8u
#define TAIL_REDUCE_LDS_FLOATS (16u * 9u)
__device__ __forceinline__ void tail_reduce_cols8(const TailJobs &tj, float *lds, uint32_t unit) {
	float (*part)[9] = reinterpret_cast<float (*)[9]>(lds);
	const uint32_t c = threadIdx.x & 7u, q = (threadIdx.x >> 3) & 15u, col = unit * TAIL_REDUCE_COLS + c;
	if (threadIdx.x < 128u) {
		float s = 0.f;
		if (col < tj.width) {
			for (uint32_t k0 = q; k0 < tj.n_slabs; k0 += 256u) {          // sixteen loads in flight, added in slab order
				float v[16];
#pragma unroll
				for (uint32_t j = 0; j < 16; ++j) { const uint32_t k = k0 + 16u * j; v[j] = k < tj.n_slabs ? tj.slabs[(size_t)k * tj.width + col] : 0.f; }
#pragma unroll
				for (uint32_t j = 0; j < 16; ++j) if (k0 + 16u * j < tj.n_slabs) s += v[j];
			}
		}
		part[q][c] = s;
	}
	__syncthreads();
	if (threadIdx.x < 8u && col < tj.width) {
		float t = 0.f;
#pragma unroll
		for (int g = 0; g < 16; ++g) t += part[g][c];
		tj.reduce_out[col] = t;
		if (tj.do_sweep16) pack_sweep_column(tj.a16, tj.b16, tj.c, col, t);          // fp16 configuration: k_reduce_slabs_sweep's second half
		if (tj.do_sweep) pack32_sweep_column(tj, col, t);                            // fp32 configuration: k_mlp32_sweep_pack's first half
	}
}
// every workgroup of a launch with `n_wg` workgroups, this one being number `wg`: its share of the reduction.  Call at the top of the kernel, before its own LDS use, by ALL threads.
__device__ __forceinline__ void tail_reduce_share(const TailJobs &tj, float *lds, uint32_t wg, uint32_t n_wg) {
	const uint32_t n_units = (tj.width + TAIL_REDUCE_COLS - 1u) / TAIL_REDUCE_COLS;
	for (uint32_t unit = wg; unit < n_units; unit += n_wg) { tail_reduce_cols8(tj, lds, unit); __syncthreads(); }
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

// k_mlp32_sweep_pack's second half, distributed: 32 slots of the fragment buffer per unit, gathered from the UPDATED pack in memory (the launch before this one swept it).
#define TAIL_PACK_SLOTS 32u
__device__ __forceinline__ void tail_pack_share(const TailJobs &tj, uint32_t wg, uint32_t n_wg) {
	if (threadIdx.x >= TAIL_PACK_SLOTS) return;
	const uint32_t n_units = (PACK_TABLE_32 + PACK_TABLE_SPLIT) / TAIL_PACK_SLOTS;
	_Float16 *split_out = reinterpret_cast<_Float16 *>(tj.packed_out + NF32_ALL * 256);
	for (uint32_t unit = wg; unit < n_units; unit += n_wg) {
		const uint32_t idx = unit * TAIL_PACK_SLOTS + threadIdx.x;
		const uint32_t src = tj.pack_table[idx];
		const float wv = src ? tj.pack[src - 1u] : 0.f;
		if (idx < PACK_TABLE_32) tj.packed_out[idx] = wv;
		else {
			const uint32_t r = idx - PACK_TABLE_32;
			const _Float16 h = (_Float16)wv;
			split_out[r] = h; split_out[PACK_TABLE_SPLIT + r] = (_Float16)((wv - (float)h) * SPLIT_SCALE);
		}
	}
}
static_assert((PACK_TABLE_32 + PACK_TABLE_SPLIT) % TAIL_PACK_SLOTS == 0 && PACK_TABLE_32 % TAIL_PACK_SLOTS == 0, "whole units of fragment slots");
