// fp32 field network FORWARD on fp16 matrix cores with fp32-level accuracy: every operand x is carried as two halves, h = fp16(x) and m = fp16((x - h) * 2^11) (the
// residual of the first rounding, pre-scaled so that it sits in fp16's normal range), and a product sum is three MFMAs: sum a_h b_h + 2^-11 (sum a_h b_m + sum a_m b_h).
// h + m 2^-11 carries 22 bits of x, the dropped a_m b_m term is 2^-22 relative: measured 3e-7 of the output scale over the five-layer chain, the same as an fp32
// matrix product (fp32 accumulation in both cases).  v_mfma_f32_16x16x32_f16 runs at 16x the rate of v_mfma_f32_16x16x4_f32 on gfx950, so three of them are 5.3x the
// fp32 MFMA peak - and two halves per weight are the same 4 bytes of LDS as one float.  (csrc/field_split.hip; shared with field32.hip's fused sweep + pack tail.)
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#define NSPLIT_FWD 20                           // forward fragments of the five layers, 512 halves each, in field_mlp.hip's order and permutation
#define NSPLIT_BWD 22                           // transposed fragments of the dgrad chain (field_mlp.hip's fragments 20..41)
#define NSPLIT_FRAGS (NSPLIT_FWD + NSPLIT_BWD)
#define NSPLIT_HALVES (2 * NSPLIT_FRAGS * 512)  // [part: h | m][fragment][lane][8]
#define SPLIT_SCALE 2048.0f

__device__ __forceinline__ int sp_k32(int g, int j) { return 8 * g + j; }
__device__ __forceinline__ int sp_k64(int kb, int g, int j) { return 32 * kb + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4)); }
// fp32 weight behind slot j of fragment f for lane (o = lane & 15: row of the 16-row A tile, g = lane >> 4).  Packs, (out,in) row-major:
// wd: W0 @0 [64][32], W1 @2048 [16][64];  wc: V0 @0 [64][32], V1 @2048 [64][64], V2 @6144 [16][64]   (ngp_network.py:21-29).  f < 20: A = W (forward), else A = W^T.
__device__ __forceinline__ float split_frag_weight(const float *wd, const float *wc, int f, int o, int g, int j) {
	if (f < 4) return wd[(16 * f + o) * 32 + sp_k32(g, j)];                                          // L0  tile f
	if (f < 6) return wd[2048 + o * 64 + sp_k64(f - 4, g, j)];                                       // L1
	if (f < 10) return wc[(16 * (f - 6) + o) * 32 + sp_k64(0, g, j)];                                // L2  input = [density(16) | SH(16)]
	if (f < 18) { const int t = (f - 10) >> 1, kb = (f - 10) & 1; return wc[2048 + (16 * t + o) * 64 + sp_k64(kb, g, j)]; }    // L3
	if (f < 20) return wc[6144 + o * 64 + sp_k64(f - 18, g, j)];                                     // L4
	f -= 20;
	if (f < 4) return j < 4 ? wc[6144 + (4 * g + j) * 64 + 16 * f + o] : 0.f;                        // dG1 = V2^T dO   (K = 16, upper slots zero)
	if (f < 12) { const int t = (f - 4) >> 1, kb = (f - 4) & 1; return wc[2048 + sp_k64(kb, g, j) * 64 + 16 * t + o]; }         // dG0 = V1^T dG1
	if (f < 14) return wc[sp_k64(f - 12, g, j) * 32 + o];                                            // dD  = (V0^T dG0)[0:16]
	if (f < 18) return j < 4 ? wd[2048 + (4 * g + j) * 64 + 16 * (f - 14) + o] : 0.f;                // dH  = W1^T dD   (K = 16)
	{ const int t = (f - 18) >> 1, kb = (f - 18) & 1; return wd[sp_k64(kb, g, j) * 32 + 16 * t + o]; }                          // dF  = W0^T dH
}
// element idx of the split fragment buffer: idx = (part * NSPLIT_FRAGS + f) * 512 + lane * 8 + j
__device__ __forceinline__ _Float16 split_frag_half(const float *wd, const float *wc, int idx) {
	const int part = idx / (NSPLIT_FRAGS * 512), r = idx - part * (NSPLIT_FRAGS * 512);
	const int f = r >> 9, lane = (r >> 3) & 63, j = r & 7;
	const float w = split_frag_weight(wd, wc, f, lane & 15, lane >> 4, j);
	const _Float16 h = (_Float16)w;
	return part == 0 ? h : (_Float16)((w - (float)h) * SPLIT_SCALE);
}
