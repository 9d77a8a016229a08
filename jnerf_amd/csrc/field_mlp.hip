// Fused field network for gfx950: SH direction encoding + density MLP (32->64->16) + colour MLP (32->64->64->3),
// forward and backward, on fp16 MFMA (v_mfma_f32_16x16x32_f16, fp32 accumulate) with all weights resident in LDS.
//
// What it computes: NGPNetworks.execute_ / .density (models/networks/ngp_network.py:77-89) with the FMLP weight pack
// (ngp_network.py:21-29, ops/code_ops/fully_fused_mlp.py:26-41) and SHEncoder (position_encoders/sh_encoder/op_header/
// SphericalEncode.h:45-95); backward = what FullyFusedMlp_weight.grad (fully_fused_mlp.py:88-145) returns: dL/dinput and the
// five weight gradients.  The reference's implementation is a binary-only tiny-cuda-nn object; nothing here derives from it.
//
// CDNA4 design
//  * "Transposed" formulation: every layer is  Y^T[neurons x samples] = W[neurons x k] * X^T[k x samples], i.e. the weights are
//    the MFMA A operand and the 16 samples of a wave tile are the B columns.  The C fragment of one layer (lane = sample
//    lane&15, registers = neurons 4*(lane>>4)+r) is then *already* a B fragment of the next layer up to a permutation of the k
//    index — and k order is free as long as A is loaded with the same permutation.  The weight fragments are therefore staged
//    into LDS pre-permuted, one conflict-free ds_read_b128 per lane per fragment, and activations never leave registers
//    between layers: no LDS round trip, no [n,64] intermediates in HBM (the reference writes 3 x 128 B/sample of them).
//  * Backward recomputes the forward (20 MFMAs per 16 samples — cheaper than re-reading saved activations), runs the dgrad chain
//    the same register-resident way with transposed weight fragments, and only the weight gradients (a contraction over
//    SAMPLES) go through LDS: activations/gradients are written once as [neuron][sample] and each wave accumulates 10 of
//    the 40 16x16 weight-gradient tiles over the whole persistent loop; one fp32 slab per workgroup is written at the end and
//    summed by ngp_reduce_slabs (no atomics, deterministic).
//  * Features arrive level-major ([16][n] pairs) from the XCD-aware hash kernel: every wave load is four 64-B segments.
#include "ngp_common.h"
#include "mlp_tail.h"
#include <stdlib.h>
#include <map>
#include <mutex>
#include <utility>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// packed weights (row-major (out,in), f16):  wd: W0 @0 [64][32], W1 @2048 [16][64];  wc: V0 @0 [64][32], V1 @2048 [64][64], V2 @6144 [16][64]
#define N_FWD_FRAGS 20
#define N_BWD_FRAGS 22
// logical k of MFMA slot (g = lane>>4, j) for the three kinds of B operand
__device__ __forceinline__ int k32(int g, int j) { return 8 * g + j; }                                            // natural order (features)
__device__ __forceinline__ int k64(int kb, int g, int j) { return 32 * kb + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4)); }   // two C tiles 2kb, 2kb+1

// value of weight fragment `f`, lane (o = lane&15, g = lane>>4), slot j.  f < 20: forward (A = W), f >= 20: backward (A = W^T).
__device__ __forceinline__ _Float16 frag_value(const _Float16 *__restrict__ wd, const _Float16 *__restrict__ wc, int f, int o, int g, int j) {
	if (f < 4) return wd[(16 * f + o) * 32 + k32(g, j)];                                         // L0  tile t=f
	if (f < 6) return wd[2048 + o * 64 + k64(f - 4, g, j)];                                      // L1  kb
	if (f < 10) return wc[(16 * (f - 6) + o) * 32 + k64(0, g, j)];                               // L2  tile t, input = [D(16), SH(16)]
	if (f < 18) { int t = (f - 10) >> 1, kb = (f - 10) & 1; return wc[2048 + (16 * t + o) * 64 + k64(kb, g, j)]; }   // L3
	if (f < 20) return wc[6144 + o * 64 + k64(f - 18, g, j)];                                    // L4
	f -= 20;
	if (f < 4) return j < 4 ? wc[6144 + (4 * g + j) * 64 + 16 * f + o] : (_Float16)0;            // dG1 = V2^T dO   (K = 16, upper half zero)
	if (f < 12) { int t = (f - 4) >> 1, kb = (f - 4) & 1; return wc[2048 + k64(kb, g, j) * 64 + 16 * t + o]; }       // dG0 = V1^T dG1
	if (f < 14) return wc[k64(f - 12, g, j) * 32 + o];                                           // dD  = (V0^T dG0)[0:16]
	if (f < 18) return j < 4 ? wd[2048 + (4 * g + j) * 64 + 16 * (f - 14) + o] : (_Float16)0;    // dH  = W1^T dD   (K = 16)
	{ int t = (f - 18) >> 1, kb = (f - 18) & 1; return wd[k64(kb, g, j) * 32 + 16 * t + o]; }    // dF  = W0^T dH
}
// The permuted fragments are built once per call by a small kernel (k_pack_frags, 42 x 512 halves) into a per-stream scratch; every workgroup then stages them
// into LDS with plain 16-byte copies.  (Building them per workgroup cost 40-84 dependent 2-byte gathers per thread - more than the MFMA work of the whole launch.)
__global__ __launch_bounds__(256) void k_pack_frags(const _Float16 *__restrict__ wd, const _Float16 *__restrict__ wc, _Float16 *__restrict__ out, int n_frags) {
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= n_frags * 512) return;
	const int f = idx >> 9, lane = (idx >> 3) & 63, j = idx & 7;
	out[idx] = frag_value(wd, wc, f, lane & 15, lane >> 4, j);
}
__device__ __forceinline__ void stage_weights(_Float16 *lds, const _Float16 *__restrict__ packed, int n_frags) {
	const uint4 *src = reinterpret_cast<const uint4 *>(packed);
	uint4 *dst = reinterpret_cast<uint4 *>(lds);
	for (int idx = threadIdx.x; idx < n_frags * 64; idx += blockDim.x) dst[idx] = src[idx];
}
__device__ __forceinline__ half8 ld_frag(const _Float16 *lds, int f, int lane) { return *reinterpret_cast<const half8 *>(lds + f * 512 + lane * 8); }

// (r6b) packed register arithmetic, written out: two accumulator values are ONE v_cvt_pk_f16_f32, and the ReLU is one v_pk_max_f16 on the pair - rounding is monotone and
// keeps the sign, so max(fp16(x), 0) is fp16(max(x, 0)).  (Left to the compiler an activation was two v_max_f32 - one to quiet a NaN the matrix instruction might have
// produced - and a scalar conversion, then a re-packing permute: 3.5 vector instructions per value, 1 now.)
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half2v cvt_pk(float a, float b) { return __builtin_convertvector((float2v){a, b}, half2v); }
__device__ __forceinline__ half8 join4(half2v a, half2v b, half2v c, half2v d) {
	return __builtin_bit_cast(half8, (uint4v){__builtin_bit_cast(unsigned int, a), __builtin_bit_cast(unsigned int, b), __builtin_bit_cast(unsigned int, c), __builtin_bit_cast(unsigned int, d)});
}
__device__ __forceinline__ half8 pack_relu(floatx4 a, floatx4 b) {
	const half2v z = {(_Float16)0, (_Float16)0};
	return join4(__builtin_elementwise_max(cvt_pk(a[0], a[1]), z), __builtin_elementwise_max(cvt_pk(a[2], a[3]), z),
	             __builtin_elementwise_max(cvt_pk(b[0], b[1]), z), __builtin_elementwise_max(cvt_pk(b[2], b[3]), z));
}
// relu'(pre-activation) * grad; the mask is taken from the fp32 accumulator (bit k of `mask` <-> slot k), not from the rounded fp16 activation
__device__ __forceinline__ half8 pack_masked(floatx4 a, floatx4 b, uint32_t mask) {
	floatx4 va, vb;
#pragma unroll
	for (int k = 0; k < 4; ++k) { va[k] = (mask >> k) & 1u ? a[k] : 0.f; vb[k] = (mask >> (4 + k)) & 1u ? b[k] : 0.f; }
	return join4(cvt_pk(va[0], va[1]), cvt_pk(va[2], va[3]), cvt_pk(vb[0], vb[1]), cvt_pk(vb[2], vb[3]));
}
__device__ __forceinline__ uint32_t relu_mask(floatx4 a, floatx4 b) {
	uint32_t m = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) { m |= a[k] > 0.f ? (1u << k) : 0u; m |= b[k] > 0.f ? (1u << (4 + k)) : 0u; }
	return m;
}

// degree-4 SH of (2d-1), components 4g..4g+3 (SphericalEncode.h:77-95)
__device__ __forceinline__ void sh4(const float d[3], int g, float o[4]) {
	const float x = d[0] * 2.f - 1.f, y = d[1] * 2.f - 1.f, z = d[2] * 2.f - 1.f;
	const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
	if (g == 0) { o[0] = 0.28209479177387814f; o[1] = -0.48860251190291987f * y; o[2] = 0.48860251190291987f * z; o[3] = -0.48860251190291987f * x; }
	else if (g == 1) { o[0] = 1.0925484305920792f * xy; o[1] = -1.0925484305920792f * yz; o[2] = 0.94617469575755997f * z2 - 0.31539156525251999f; o[3] = -1.0925484305920792f * xz; }
	else if (g == 2) { o[0] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2; o[1] = 0.59004358992664352f * y * (-3.0f * x2 + y2); o[2] = 2.8906114426405538f * xy * z; o[3] = 0.45704579946446572f * y * (1.0f - 5.0f * z2); }
	else { o[0] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f); o[1] = 0.45704579946446572f * x * (1.0f - 5.0f * z2); o[2] = 1.4453057213202769f * z * (x2 - y2); o[3] = 0.59004358992664352f * x * (-x2 + 3.0f * y2); }
}

// B fragment of the first layer: features of sample `i` for levels 4g..4g+3
template <int LAYOUT>
__device__ __forceinline__ half8 load_feat(const _Float16 *__restrict__ feat, uint32_t n, uint32_t i, int g) {
	half8 r;
	if (LAYOUT == NGP_LAYOUT_SOA) {
		const uint32_t *f32 = reinterpret_cast<const uint32_t *>(feat);
		uint32_t v[4];
#pragma unroll
		for (int q = 0; q < 4; ++q) v[q] = f32[(size_t)(4 * g + q) * n + i];
		r = *reinterpret_cast<half8 *>(v);
	} else {
		r = *reinterpret_cast<const half8 *>(feat + (size_t)i * 32 + 8 * g);
	}
	return r;
}

struct FwdState { half8 feat, hfrag[2], in2, g0[2], g1[2]; floatx4 den, rgb; uint32_t mh[2], mg0[2], mg1[2]; };

template <bool DENSITY_ONLY>
__device__ __forceinline__ void forward_tile(const _Float16 *wl, int lane, half8 feat, const float sh[4], FwdState &st) {
	const floatx4 z = {0.f, 0.f, 0.f, 0.f};
	floatx4 c0[4];
#pragma unroll
	for (int t = 0; t < 4; ++t) c0[t] = MFMA(ld_frag(wl, t, lane), feat, z);
	st.feat = feat;
	st.hfrag[0] = pack_relu(c0[0], c0[1]); st.hfrag[1] = pack_relu(c0[2], c0[3]);
	st.mh[0] = relu_mask(c0[0], c0[1]); st.mh[1] = relu_mask(c0[2], c0[3]);
	floatx4 d = MFMA(ld_frag(wl, 4, lane), st.hfrag[0], z);
	d = MFMA(ld_frag(wl, 5, lane), st.hfrag[1], d);
	st.den = d;
	if (DENSITY_ONLY) return;
	half8 in2;
#pragma unroll
	for (int k = 0; k < 4; ++k) { in2[k] = (_Float16)d[k]; in2[4 + k] = (_Float16)sh[k]; }
	st.in2 = in2;
	floatx4 c2[4];
#pragma unroll
	for (int t = 0; t < 4; ++t) c2[t] = MFMA(ld_frag(wl, 6 + t, lane), in2, z);
	st.g0[0] = pack_relu(c2[0], c2[1]); st.g0[1] = pack_relu(c2[2], c2[3]);
	st.mg0[0] = relu_mask(c2[0], c2[1]); st.mg0[1] = relu_mask(c2[2], c2[3]);
	floatx4 c3[4];
#pragma unroll
	for (int t = 0; t < 4; ++t) { c3[t] = MFMA(ld_frag(wl, 10 + 2 * t, lane), st.g0[0], z); c3[t] = MFMA(ld_frag(wl, 11 + 2 * t, lane), st.g0[1], c3[t]); }
	st.g1[0] = pack_relu(c3[0], c3[1]); st.g1[1] = pack_relu(c3[2], c3[3]);
	st.mg1[0] = relu_mask(c3[0], c3[1]); st.mg1[1] = relu_mask(c3[2], c3[3]);
	floatx4 o = MFMA(ld_frag(wl, 18, lane), st.g1[0], z);
	st.rgb = MFMA(ld_frag(wl, 19, lane), st.g1[1], o);
}

template <typename T> __device__ __forceinline__ void store_out4(T *p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store_out4<float>(float *p, float a, float b, float c, float d) { *reinterpret_cast<float4 *>(p) = make_float4(a, b, c, d); }
template <> __device__ __forceinline__ void store_out4<__half>(__half *p, float a, float b, float c, float d) {
	half4 v = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
	*reinterpret_cast<half4 *>(p) = v;
}
template <typename T> __device__ __forceinline__ void store_out1(T *p, float a);
template <> __device__ __forceinline__ void store_out1<float>(float *p, float a) { *p = a; }
template <> __device__ __forceinline__ void store_out1<__half>(__half *p, float a) { *p = __float2half(a); }

template <typename T, int LAYOUT, bool DENSITY_ONLY>
__global__ __launch_bounds__(256, 2) void k_field_fwd(uint32_t n, const _Float16 *__restrict__ feat, const float *__restrict__ dir, uint32_t dir_stride,
                                                   const _Float16 *__restrict__ packed, T *__restrict__ out,
                                                   const uint32_t *__restrict__ n_valid) {
	__shared__ __attribute__((aligned(16))) _Float16 wl[N_FWD_FRAGS * 512];
	stage_weights(wl, packed, DENSITY_ONLY ? 6 : N_FWD_FRAGS);
	__syncthreads();
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	const int lane = threadIdx.x & 63, s = lane & 15, g = lane >> 4;
	const uint32_t n_tiles = (lim + 15u) / 16u;
	const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
	// two tiles in flight per wave: the next tile's features / direction are requested before the current tile's MFMA chain starts
	auto fetch = [&](uint32_t tile, half8 &f, float d[3]) {
		const uint32_t i = tile * 16u + s;
		const uint32_t ic = i < lim ? i : lim - 1;
		f = load_feat<LAYOUT>(feat, n, ic, g);
		if (!DENSITY_ONLY) { d[0] = dir[(size_t)ic * dir_stride]; d[1] = dir[(size_t)ic * dir_stride + 1]; d[2] = dir[(size_t)ic * dir_stride + 2]; }
	};
	half8 f, fn; float d[3] = {0.f, 0.f, 0.f}, dn[3] = {0.f, 0.f, 0.f};
	if (wave < n_tiles) fetch(wave, f, d);
	for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
		const uint32_t i = tile * 16u + s;
		const bool more = tile + n_waves < n_tiles;
		if (more) fetch(tile + n_waves, fn, dn);
		float sh[4] = {0.f, 0.f, 0.f, 0.f};
		if (!DENSITY_ONLY) sh4(d, g, sh);
		FwdState st;
		forward_tile<DENSITY_ONLY>(wl, lane, f, sh, st);
		if (g == 0 && i < lim) {
			if (DENSITY_ONLY) store_out1<T>(out + i, st.den[0]);
			else store_out4<T>(out + (size_t)i * 4, st.rgb[0], st.rgb[1], st.rgb[2], st.den[0]);
		}
		if (more) { f = fn; d[0] = dn[0]; d[1] = dn[1]; d[2] = dn[2]; }
	}
}

// ---------------------------------------------------------------------------------------------------------------- backward
#define BT 128                // samples per workgroup tile: 8 waves x 16 samples
#define RS (BT + 8)           // LDS row stride in halves (272 B: 16-B aligned rows, breaks the 128-B bank period)
// The weight-gradient contraction is staged in three phases that reuse one 192-row region (each phase: 8 waves write their 16 sample columns, barrier,
// every wave accumulates its share of the phase's 16x16 tiles over the 128 samples, barrier).  A single 480-row region for everything would cost 130 KB and
// leave room for only four waves per CU; with 52 KB the workgroup has eight waves = two per SIMD, which is what hides the latency of the 42 dependent
// MFMA steps of the forward-recompute + dgrad chain (97 % of this kernel's time with one wave per SIMD).
//   phase A: dG1 0..63 | G0 64..127                      -> V1  (16 tiles, two per wave)
//   phase B: dH 0..63 | F 64..95 | dG0 96..159 | IN2 160..191   -> W0, V0 (8 + 8 tiles, one of each per wave)
//   phase C: dD 0..15 | H 16..79 | dO 80..95 | G1 96..159       -> W1 (waves 0-3), V2 (waves 4-7)
#define N_ROWS 192

__device__ __forceinline__ void st_rows64(_Float16 *stage, int row0, int col, int g, half8 lo, half8 hi) {   // two k64 fragments = 64 neurons
#pragma unroll
	for (int j = 0; j < 8; ++j) { stage[(row0 + k64(0, g, j)) * RS + col] = lo[j]; stage[(row0 + k64(1, g, j)) * RS + col] = hi[j]; }
}
__device__ __forceinline__ half8 ld_rows(const _Float16 *stage, int row, int col) { return *reinterpret_cast<const half8 *>(stage + row * RS + col); }

// ---- (r6) the transposed staging image (field_split.hip's, one plane): [sample row][neuron], 512-byte rows (256 neuron slots, 192 used).  A lane's four consecutive neurons
// are ONE 8-byte store (30 per lane and trip instead of 120 ds_write_b16), and the weight-gradient phases read the operands back through the hardware transpose read
// (ds_read_b64_tr_b16: sixteen lanes fetch a [4 samples][16 neurons] block, lane i receives neuron i's four samples) - the same values in the same k slots of the same
// MFMAs, so the same bits.  The chunk index (four neurons = 8 bytes) is XORed with a function of the row so that both access patterns spread over the banks
// (tools/microbench_trstage.hip: the analysis and the measurement behind the function).
#define FT_ROW 256
// Unrolling of the weight-gradient loops (four k steps per phase).  History: fully unrolled, the transposed image's reads (two per operand) pushed round 6a's kernel to
// 256 VGPRs + scratch; unrolled by two it needed 242 - and a 512-thread workgroup at 2 x 242 registers per SIMD lane leaves room for ONE 24-register wavefront of another
// kernel, so on the sparse many-ray batches whose serial marcher (k_march_count, 300 us) holds two wavefronts on many SIMDs the workgroups had to wait for CUs to drain
// (90 us in the procedural-fox trace against 45 us alone, profiles/r06z_fox_kernel_trace.md); not unrolled: 226.  (r6b) With the packed register arithmetic and the reads
// behind lane bases the kernel needs 194 not unrolled and 202 - 206 FULLY unrolled (-DFIELD_WG_UNROLL=4: no scratch, 419 -> 346 vector instructions per trip, same bits);
// measured together with the split kernel's: not faster (profiles/r06y_ab_unroll.txt), so not unrolled stays - the fewest registers beside the serial marcher.
#ifndef FIELD_WG_UNROLL
#define FIELD_WG_UNROLL 1
#endif
typedef short ft_short4 __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ int ft_f(int s) { return ((s & 3) << 2) | ((s >> 2) & 3) | (((s >> 3) & 1) << 4); }
__device__ __forceinline__ int ft_off(int s, int c) { return s * FT_ROW + ((c ^ ft_f(s)) << 2); }                          // halves; c = chunk of four neurons, 0..63
struct alignas(8) FtH4 { _Float16 v[4]; };
__device__ __forceinline__ void st_chunk_T(_Float16 *stage, int s, int c, const half8 &h, int hi) {                        // slots 4 hi .. 4 hi + 3 of an operand
	FtH4 a;
#pragma unroll
	for (int j = 0; j < 4; ++j) a.v[j] = h[4 * hi + j];
	*reinterpret_cast<FtH4 *>(stage + ft_off(s, c)) = a;
}
// the lane's eight slots of two k64 half-fragments (neurons nrow0 .. nrow0 + 63) / of one k32- or k64(0)-ordered fragment / the low four slots, as row `s`
__device__ __forceinline__ void st_rows64_T(_Float16 *stage, int nrow0, int s, int g, const half8 &lo, const half8 &hi) {
	st_chunk_T(stage, s, (nrow0 >> 2) + g, lo, 0); st_chunk_T(stage, s, (nrow0 >> 2) + 4 + g, lo, 1);                          // k64(0, g, j): 4g + j | 16 + 4g + (j - 4)
	st_chunk_T(stage, s, (nrow0 >> 2) + 8 + g, hi, 0); st_chunk_T(stage, s, (nrow0 >> 2) + 12 + g, hi, 1);                    // k64(1, g, j): 32 + ...
}
__device__ __forceinline__ void st_rows32_T(_Float16 *stage, int nrow0, int s, int g, const half8 &v, bool k32_order) {
	if (k32_order) { st_chunk_T(stage, s, (nrow0 >> 2) + 2 * g, v, 0); st_chunk_T(stage, s, (nrow0 >> 2) + 2 * g + 1, v, 1); }   // k32(g, j) = 8g + j
	else { st_chunk_T(stage, s, (nrow0 >> 2) + g, v, 0); st_chunk_T(stage, s, (nrow0 >> 2) + 4 + g, v, 1); }
}
__device__ __forceinline__ void st_rows16_T(_Float16 *stage, int nrow0, int s, int g, const half8 &v) { st_chunk_T(stage, s, (nrow0 >> 2) + g, v, 0); }
// (r6b) An operand's transpose reads as immediates behind two lane-dependent addresses (sample rows +0 | +4 of the first k step) that are formed once per phase and made
// opaque to the compiler's constant re-association: the staging image lies 42 KiB into the workgroup's LDS and spans 64 KiB, so "lane part + region base + step" mostly
// exceeds a DS instruction's 16-bit immediate and every read had its own shift and three-operand add.  A k step is 32 sample rows = 16 KiB; ft_f depends on the row's low
// four bits only, which a step keeps.
struct TrAddr { uint32_t a0, a4; };
__device__ __forceinline__ TrAddr ft_lane_addr(const _Float16 *stage, int nrow, int o, int s0) {    // neuron nrow + o (nrow a multiple of 4), samples s0 .. s0 + 7 (s0 a multiple of 8)
	TrAddr r;
	r.a0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const _Float16 *)(stage + ft_off(s0 + (o >> 2), (nrow >> 2) + (o & 3)));
	r.a4 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const _Float16 *)(stage + ft_off(s0 + 4 + (o >> 2), (nrow >> 2) + (o & 3)));
	asm("" : "+v"(r.a0)); asm("" : "+v"(r.a4));
	return r;
}
__device__ __forceinline__ half8 ld_rows_T_at(TrAddr t, uint32_t byte_off) {
	typedef __attribute__((address_space(3))) ft_short4 lds_ft_short4;
	const ft_short4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ft_short4 *)(uintptr_t)(t.a0 + byte_off));
	const ft_short4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ft_short4 *)(uintptr_t)(t.a4 + byte_off));
	typedef short ft_short8 __attribute__((ext_vector_type(8)));
	ft_short8 r;
#pragma unroll
	for (int j = 0; j < 4; ++j) { r[j] = a[j]; r[4 + j] = b[j]; }
	return __builtin_bit_cast(half8, r);
}
#define FT_KSTEP (32 * FT_ROW * 2)                                                               // bytes per k step

template <typename T> __device__ __forceinline__ void load_dout(const T *p, float o[4]);
template <> __device__ __forceinline__ void load_dout<float>(const float *p, float o[4]) { float4 v = *reinterpret_cast<const float4 *>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <> __device__ __forceinline__ void load_dout<__half>(const __half *p, float o[4]) { half4 v = *reinterpret_cast<const half4 *>(p); o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3]; }

template <typename T, int LAYOUT, bool TR /* (r6) the transposed staging image: 8-byte stores + transpose reads */>
__global__ __launch_bounds__(512, 1) void k_field_bwd(uint32_t n, const _Float16 *__restrict__ feat, const float *__restrict__ dir, uint32_t dir_stride,
                                                      const _Float16 *__restrict__ packed, const T *__restrict__ dout,
                                                      _Float16 *__restrict__ dfeat, float *__restrict__ slabs, const uint32_t *__restrict__ n_valid, AbsmaxOut am) {
	float lmax[2][2] = {{0.f, 0.f}, {0.f, 0.f}};          // running max |dL/dfeature| (of the fp16 values as stored) of levels 8t + 2g + pr over this lane's samples (absmax_epilogue)
	extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
	_Float16 *wl = smem;                                         // 42 fragments
	_Float16 *stage = smem + (N_FWD_FRAGS + N_BWD_FRAGS) * 512;  // [N_ROWS][RS] | TR: [BT][FT_ROW]
#define LDR(nrow, cs) ld_rows(stage, (nrow) + o, (cs))
	stage_weights(wl, packed, N_FWD_FRAGS + N_BWD_FRAGS);
	const _Float16 *wb = wl + N_FWD_FRAGS * 512;
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	const int lane = threadIdx.x & 63, s = lane & 15, g = lane >> 4, w = threadIdx.x >> 6;
	const uint32_t n_bt = (lim + BT - 1) / BT;
	const floatx4 z = {0.f, 0.f, 0.f, 0.f};
	// this wave's 5 weight-gradient tiles: V1 (to, ti0), (to, ti0+1) | W0 (to, tj) | V0 (to, tj) | W1 (ti=w) for waves 0-3, V2 (ti=w-4) for waves 4-7
	const int to = w >> 1, ti0 = 2 * (w & 1), tj = w & 1;
	floatx4 aV1[2] = {z, z}, aW0 = z, aV0 = z, aX = z;
	__syncthreads();
	// the next tile's inputs are requested while the current tile is in its MFMA / LDS phases (one workgroup per CU: nothing else hides the HBM round trip)
	struct Inputs { half8 f; float d3[3]; float go[4]; };
	auto fetch = [&](uint32_t bt, Inputs &in) {
		const uint32_t i = bt * BT + 16u * w + s;
		const bool valid = i < lim;
		const uint32_t ic = valid ? i : lim - 1;
		in.f = load_feat<LAYOUT>(feat, n, ic, g);
		in.d3[0] = dir[(size_t)ic * dir_stride]; in.d3[1] = dir[(size_t)ic * dir_stride + 1]; in.d3[2] = dir[(size_t)ic * dir_stride + 2];
		in.go[0] = in.go[1] = in.go[2] = in.go[3] = 0.f;
		if (valid) load_dout<T>(dout + (size_t)i * 4, in.go);
	};
	Inputs cur, nxt;
	if (blockIdx.x < n_bt) fetch(blockIdx.x, cur);
	for (uint32_t bt = blockIdx.x; bt < n_bt; bt += gridDim.x) {
		const uint32_t i = bt * BT + 16u * w + s;
		const bool valid = i < lim;
		const bool more = bt + gridDim.x < n_bt;
		if (more) fetch(bt + gridDim.x, nxt);
		const half8 f = cur.f;
		float sh[4]; sh4(cur.d3, g, sh);
		float go[4] = {cur.go[0], cur.go[1], cur.go[2], cur.go[3]};
		FwdState st;
		forward_tile<false>(wl, lane, f, sh, st);
		// ---- dgrad chain (register resident, transposed weights)
		half8 dO;                                                 // slots j<4 <-> dO neuron 4g+j; only neurons 0..2 (g==0) are non-zero
#pragma unroll
		for (int j = 0; j < 8; ++j) dO[j] = (_Float16)0;
		if (g == 0) { dO[0] = (_Float16)go[0]; dO[1] = (_Float16)go[1]; dO[2] = (_Float16)go[2]; }
		floatx4 c[4];
#pragma unroll
		for (int t = 0; t < 4; ++t) c[t] = MFMA(ld_frag(wb, t, lane), dO, z);
		const half8 dG1lo = pack_masked(c[0], c[1], st.mg1[0]), dG1hi = pack_masked(c[2], c[3], st.mg1[1]);
#pragma unroll
		for (int t = 0; t < 4; ++t) { c[t] = MFMA(ld_frag(wb, 4 + 2 * t, lane), dG1lo, z); c[t] = MFMA(ld_frag(wb, 5 + 2 * t, lane), dG1hi, c[t]); }
		const half8 dG0lo = pack_masked(c[0], c[1], st.mg0[0]), dG0hi = pack_masked(c[2], c[3], st.mg0[1]);
		floatx4 dD = MFMA(ld_frag(wb, 12, lane), dG0lo, z);
		dD = MFMA(ld_frag(wb, 13, lane), dG0hi, dD);
		if (g == 0) dD[0] += go[3];                               // out[:,3] = den[:,0]  (ngp_network.py:83)
		half8 dDf;
#pragma unroll
		for (int j = 0; j < 4; ++j) { dDf[j] = (_Float16)dD[j]; dDf[4 + j] = (_Float16)0; }
#pragma unroll
		for (int t = 0; t < 4; ++t) c[t] = MFMA(ld_frag(wb, 14 + t, lane), dDf, z);
		const half8 dHlo = pack_masked(c[0], c[1], st.mh[0]), dHhi = pack_masked(c[2], c[3], st.mh[1]);
		floatx4 dF[2];
#pragma unroll
		for (int t = 0; t < 2; ++t) { dF[t] = MFMA(ld_frag(wb, 18 + 2 * t, lane), dHlo, z); dF[t] = MFMA(ld_frag(wb, 19 + 2 * t, lane), dHhi, dF[t]); }
		if (valid) {                                              // feature 16t+4g+r  ->  level 8t+2g+(r>>1), component r&1
#pragma unroll
			for (int t = 0; t < 2; ++t)
#pragma unroll
				for (int pr = 0; pr < 2; ++pr) {
					half2v v = {(_Float16)dF[t][2 * pr], (_Float16)dF[t][2 * pr + 1]};
					const uint32_t level = 8 * t + 2 * g + pr;
					lmax[t][pr] = fmaxf(lmax[t][pr], fmaxf(fabsf((float)v[0]), fabsf((float)v[1])));
					if (LAYOUT == NGP_LAYOUT_SOA) *reinterpret_cast<half2v *>(dfeat + ((size_t)level * n + i) * 2) = v;
					else *reinterpret_cast<half2v *>(dfeat + (size_t)i * 32 + 2 * level) = v;
				}
		}
		// ---- weight gradients: dW[o][i] += sum_s dY[s][o] X[s][i]   (A = dY^T rows o, B = X columns i, k = sample), three staging phases
		const int col = 16 * w + s, o = lane & 15;
		// phase A
		if (TR) { st_rows64_T(stage, 0, col, g, dG1lo, dG1hi); st_rows64_T(stage, 64, col, g, st.g0[0], st.g0[1]); }
		else { st_rows64(stage, 0, col, g, dG1lo, dG1hi); st_rows64(stage, 64, col, g, st.g0[0], st.g0[1]); }
		__syncthreads();
		if (TR) {
			const TrAddr ta = ft_lane_addr(stage, 16 * to, o, 8 * g), tb0 = ft_lane_addr(stage, 64 + 16 * ti0, o, 8 * g), tb1 = ft_lane_addr(stage, 64 + 16 * (ti0 + 1), o, 8 * g);
#pragma unroll FIELD_WG_UNROLL
			for (int kb = 0; kb < BT / 32; ++kb) {
				const half8 a_dg1 = ld_rows_T_at(ta, kb * FT_KSTEP);
				aV1[0] = MFMA(a_dg1, ld_rows_T_at(tb0, kb * FT_KSTEP), aV1[0]);
				aV1[1] = MFMA(a_dg1, ld_rows_T_at(tb1, kb * FT_KSTEP), aV1[1]);
			}
		} else {
#pragma unroll FIELD_WG_UNROLL
			for (int kb = 0; kb < BT / 32; ++kb) {
				const int cs = 32 * kb + 8 * g;
				const half8 a_dg1 = LDR(16 * to, cs);
				aV1[0] = MFMA(a_dg1, LDR(64 + 16 * ti0, cs), aV1[0]);
				aV1[1] = MFMA(a_dg1, LDR(64 + 16 * (ti0 + 1), cs), aV1[1]);
			}
		}
		__syncthreads();
		// phase B
		if (TR) {
			st_rows64_T(stage, 0, col, g, dHlo, dHhi); st_rows64_T(stage, 96, col, g, dG0lo, dG0hi);
			st_rows32_T(stage, 64, col, g, st.feat, true); st_rows32_T(stage, 160, col, g, st.in2, false);
		} else {
			st_rows64(stage, 0, col, g, dHlo, dHhi);
			st_rows64(stage, 96, col, g, dG0lo, dG0hi);
#pragma unroll
			for (int j = 0; j < 8; ++j) { stage[(64 + k32(g, j)) * RS + col] = st.feat[j]; stage[(160 + k64(0, g, j)) * RS + col] = st.in2[j]; }
		}
		__syncthreads();
		if (TR) {
			const TrAddr ta = ft_lane_addr(stage, 16 * to, o, 8 * g), tb = ft_lane_addr(stage, 64 + 16 * tj, o, 8 * g);
			const TrAddr tc = ft_lane_addr(stage, 96 + 16 * to, o, 8 * g), td = ft_lane_addr(stage, 160 + 16 * tj, o, 8 * g);
#pragma unroll FIELD_WG_UNROLL
			for (int kb = 0; kb < BT / 32; ++kb) {
				aW0 = MFMA(ld_rows_T_at(ta, kb * FT_KSTEP), ld_rows_T_at(tb, kb * FT_KSTEP), aW0);
				aV0 = MFMA(ld_rows_T_at(tc, kb * FT_KSTEP), ld_rows_T_at(td, kb * FT_KSTEP), aV0);
			}
		} else {
#pragma unroll FIELD_WG_UNROLL
			for (int kb = 0; kb < BT / 32; ++kb) {
				const int cs = 32 * kb + 8 * g;
				aW0 = MFMA(LDR(16 * to, cs), LDR(64 + 16 * tj, cs), aW0);
				aV0 = MFMA(LDR(96 + 16 * to, cs), LDR(160 + 16 * tj, cs), aV0);
			}
		}
		__syncthreads();
		// phase C
		if (TR) {
			st_rows16_T(stage, 0, col, g, dDf); st_rows16_T(stage, 80, col, g, dO);
			st_rows64_T(stage, 16, col, g, st.hfrag[0], st.hfrag[1]); st_rows64_T(stage, 96, col, g, st.g1[0], st.g1[1]);
		} else {
#pragma unroll
			for (int j = 0; j < 4; ++j) { stage[(4 * g + j) * RS + col] = dDf[j]; stage[(80 + 4 * g + j) * RS + col] = dO[j]; }
			st_rows64(stage, 16, col, g, st.hfrag[0], st.hfrag[1]);
			st_rows64(stage, 96, col, g, st.g1[0], st.g1[1]);
		}
		__syncthreads();
		if (TR) {                                                                     // W1: dD^T x H tile w (waves 0-3) | V2: dO^T x G1 tile w - 4
			const TrAddr ta = ft_lane_addr(stage, w < 4 ? 0 : 80, o, 8 * g), tb = ft_lane_addr(stage, w < 4 ? 16 + 16 * w : 96 + 16 * (w - 4), o, 8 * g);
#pragma unroll FIELD_WG_UNROLL
			for (int kb = 0; kb < BT / 32; ++kb) aX = MFMA(ld_rows_T_at(ta, kb * FT_KSTEP), ld_rows_T_at(tb, kb * FT_KSTEP), aX);
		} else {
#pragma unroll FIELD_WG_UNROLL
			for (int kb = 0; kb < BT / 32; ++kb) {
				const int cs = 32 * kb + 8 * g;
				if (w < 4) aX = MFMA(LDR(0, cs), LDR(16 + 16 * w, cs), aX);               // W1: dD^T x H tile w
				else aX = MFMA(LDR(80, cs), LDR(96 + 16 * (w - 4), cs), aX);              // V2: dO^T x G1 tile w-4
			}
		}
		__syncthreads();
		if (more) cur = nxt;
	}
	// ---- one fp32 slab per workgroup, packed like the weights (wd part 0..3071, wc part 3072..10239); C rows = 4g+r, cols = lane&15
	float *slab = slabs + (size_t)blockIdx.x * 10240;
	const int ci = lane & 15;
#pragma unroll
	for (int r = 0; r < 4; ++r) {
		const int ro = 4 * g + r;
#pragma unroll
		for (int q = 0; q < 2; ++q) slab[3072 + 2048 + (16 * to + ro) * 64 + 16 * (ti0 + q) + ci] = aV1[q][r];
		slab[(16 * to + ro) * 32 + 16 * tj + ci] = aW0[r];
		slab[3072 + (16 * to + ro) * 32 + 16 * tj + ci] = aV0[r];
		if (w < 4) slab[2048 + ro * 64 + 16 * w + ci] = aX[r];
		else slab[3072 + 6144 + ro * 64 + 16 * (w - 4) + ci] = aX[r];
	}
	if (am.parts) absmax_epilogue(am, lmax, reinterpret_cast<float *>(stage), 8);      // (the trip loop ends with a barrier: the staging region is free)
#undef LDR
}

// (r5) The free-running-groups variant of this kernel (k_field_bwd_g, NGP_FIELD_BWD_GROUPS = 2 | 3 | 4: round 3) is gone: alone it ran 61 -> 50 us, in the training step it
// never gained - re-measured in round 5 on the real fox scene, in one call: 1924 vs 1936 it/s (profiles/r05a_fox_ab.txt).  The chain needs ~220 VGPRs, so more than two groups spill.

__global__ __launch_bounds__(1024) void k_reduce_slabs(const float *__restrict__ slabs, uint32_t n_slabs, uint32_t width, float *__restrict__ out, int accumulate) {
	// 64 columns x 16 slab groups per workgroup; LDS tree over the groups
	__shared__ float part[16][65];
	const uint32_t col = blockIdx.x * 64u + (threadIdx.x & 63u), grp = threadIdx.x >> 6;
	float s = 0.f;
	if (col < width) for (uint32_t k = grp; k < n_slabs; k += 16) s += slabs[(size_t)k * width + col];
	part[grp][threadIdx.x & 63u] = s;
	__syncthreads();
	if (grp == 0 && col < width) {
		float t = 0.f;
#pragma unroll
		for (int g = 0; g < 16; ++g) t += part[g][threadIdx.x];
		out[col] = accumulate ? out[col] + t : t;
	}
}

// (r5) The tail of the fp16 configuration's iteration in ONE launch: k_reduce_slabs' column sums, and - in the thread that holds a column's total - the Adam + EMA
// update of the parameter that column is the gradient of (optim.hip: adam_ema_update, the same function on the same values: the gradient a separate sweep would load
// back is the sum this thread just stored).  The weight gradients tile one flat buffer (3072 of the density MLP's pack, then 7168 of the colour MLP's), each pack with its
// own fp32 master, moments, EMA state and fp16 shadow.  Replaces three launches (k_reduce_slabs + two 3 k / 7 k-element k_adam_ema: 8 + 5 + 5 us of an iteration).
// (struct PackSweep: ngp_common.h - the same job also rides in the hash backward's run-record launch on the single-GPU training path, mlp_tail.h)
__global__ __launch_bounds__(1024) void k_reduce_slabs_sweep(const float *__restrict__ slabs, uint32_t n_slabs, uint32_t width, float *__restrict__ out, PackSweep a, PackSweep b, AdamConsts c) {
	__shared__ float part[16][65];
	const uint32_t col = blockIdx.x * 64u + (threadIdx.x & 63u), grp = threadIdx.x >> 6;
	float s = 0.f;
	if (col < width) for (uint32_t k = grp; k < n_slabs; k += 16) s += slabs[(size_t)k * width + col];
	part[grp][threadIdx.x & 63u] = s;
	__syncthreads();
	if (grp == 0 && col < width) {
		float t = 0.f;
#pragma unroll
		for (int g = 0; g < 16; ++g) t += part[g][threadIdx.x];
		out[col] = t;                                                      // (overwrite: the fused tail only runs with grad_overwrite)
		pack_sweep_column(a, b, c, col, t);
	}
}

// ---------------------------------------------------------------------------------------------------------------- standalone SH
template <typename T>
__global__ void k_sh(uint32_t n, const float *__restrict__ dir, uint32_t stride, T *__restrict__ out) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float d[3] = {dir[(size_t)i * stride], dir[(size_t)i * stride + 1], dir[(size_t)i * stride + 2]};
#pragma unroll
	for (int g = 0; g < 4; ++g) {
		float o[4]; sh4(d, g, o);
#pragma unroll
		for (int k = 0; k < 4; ++k) store_out1<T>(out + (size_t)i * 16 + 4 * g + k, o[k]);
	}
}

// ---------------------------------------------------------------------------------------------------------------- MFMA layout self-test
__global__ void k_selftest_mfma(uint32_t *result) {
	// C = A * B with A[i][k] = (i*3 + k) % 7 - 3, B[k][j] = (k*5 + j*2) % 9 - 4 (asymmetric), both loaded with the layout assumed above.
	const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
	half8 a, b;
	for (int j = 0; j < 8; ++j) { const int k = 8 * g + j; a[j] = (_Float16)((i * 3 + k) % 7 - 3); b[j] = (_Float16)((k * 5 + i * 2) % 9 - 4); }
	const floatx4 z = {0.f, 0.f, 0.f, 0.f};
	const floatx4 c = MFMA(a, b, z);
	uint32_t bad = 0;
	for (int r = 0; r < 4; ++r) {
		const int row = 4 * g + r, col = i;
		float ref = 0.f;
		for (int k = 0; k < 32; ++k) ref += (float)((row * 3 + k) % 7 - 3) * (float)((k * 5 + col * 2) % 9 - 4);
		if (c[r] != ref) bad++;
	}
	if (bad) atomicAdd(&result[0], bad);
	if (lane == 0) result[1] = 0xC0FFEEu;
}

// ---------------------------------------------------------------------------------------------------------------- C ABI
static int check_field(const char *fn, const void *feat, const void *wd, const void *wc, int layout, int out_dtype) {
	const bool prepacked = (layout & NGP_WEIGHTS_PACKED) != 0;
	layout &= ~NGP_WEIGHTS_PACKED;
	NGP_REQUIRE(feat && wd && (wc || prepacked), NGP_E_ARG, "%s: null pointer", fn);
	NGP_REQUIRE(!prepacked || ((uintptr_t)wd & 15) == 0, NGP_E_ALIGN, "%s: packed weight buffer must be 16-byte aligned", fn);
	NGP_REQUIRE(layout == NGP_LAYOUT_AOS || layout == NGP_LAYOUT_SOA, NGP_E_ARG, "%s: bad layout %d", fn, layout);
	NGP_REQUIRE(out_dtype == NGP_F32 || out_dtype == NGP_F16, NGP_E_DTYPE, "%s: bad dtype %d", fn, out_dtype);
	NGP_REQUIRE(((uintptr_t)feat & 15) == 0, NGP_E_ALIGN, "%s: feature pointer must be 16-byte aligned", fn);
	return 0;
}
// per-(device, stream) scratch for the packed fragments; launches on one stream are ordered, so one buffer per stream is enough
static const _Float16 *pack_weights(const char *fn, hipStream_t s, const void *wd, const void *wc, int n_frags, int layout_flags) {
	if (layout_flags & NGP_WEIGHTS_PACKED) return (const _Float16 *)wd;
	static std::mutex mu;
	static std::map<std::pair<int, hipStream_t>, _Float16 *> pool;
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) { ngp_set_error("%s: hipGetDevice failed", fn); return nullptr; }
	_Float16 *buf;
	{
		std::lock_guard<std::mutex> lk(mu);
		_Float16 *&slot = pool[{dev, s}];
		if (!slot) { hipError_t e = hipMalloc((void **)&slot, (size_t)(N_FWD_FRAGS + N_BWD_FRAGS) * 512 * sizeof(_Float16)); if (e != hipSuccess) { slot = nullptr; ngp_set_error("%s: hipMalloc(fragment scratch): %s", fn, hipGetErrorString(e)); return nullptr; } }
		buf = slot;
	}
	NGP_LAUNCH(k_pack_frags, dim3(div_up((uint32_t)n_frags * 512u, 256u)), dim3(256), 0, s, (const _Float16 *)wd, (const _Float16 *)wc, buf, n_frags);
	return buf;
}
static uint32_t fwd_grid(uint32_t n) { uint32_t b = div_up(div_up(n, 16), 4); return b < 2048 ? (b ? b : 1) : 2048; }

NGP_API int ngp_field_fwd(void *stream, uint32_t n, const void *feat, int layout, const float *dir, uint32_t dir_stride, const void *wd, const void *wc,
                          void *out, int out_dtype, const uint32_t *n_valid) {
	int rc = check_field("ngp_field_fwd", feat, wd, wc, layout, out_dtype); if (rc) return rc;
	const int layout_flags = layout; layout &= ~NGP_WEIGHTS_PACKED;
	NGP_REQUIRE(dir && out && dir_stride >= 3, NGP_E_ARG, "ngp_field_fwd: bad dir/out");
	if (n == 0) return 0;
	const dim3 grid(fwd_grid(n)), block(256);
	hipStream_t s = (hipStream_t)stream;
	const _Float16 *packed = pack_weights("ngp_field_fwd", s, wd, wc, N_FWD_FRAGS, layout_flags); if (!packed) return NGP_E_ARG;
#define GO(T, L) NGP_LAUNCH((k_field_fwd<T, L, false>), grid, block, 0, s, n, (const _Float16 *)feat, dir, dir_stride, packed, (T *)out, n_valid)
	if (out_dtype == NGP_F32) { if (layout == NGP_LAYOUT_SOA) GO(float, NGP_LAYOUT_SOA); else GO(float, NGP_LAYOUT_AOS); }
	else { if (layout == NGP_LAYOUT_SOA) GO(__half, NGP_LAYOUT_SOA); else GO(__half, NGP_LAYOUT_AOS); }
#undef GO
	NGP_LAUNCH_CHECK("ngp_field_fwd");
	return 0;
}
NGP_API int ngp_density_fwd(void *stream, uint32_t n, const void *feat, int layout, const void *wd, void *out, int out_dtype) {
	int rc = check_field("ngp_density_fwd", feat, wd, wd, layout, out_dtype); if (rc) return rc;
	const int layout_flags = layout; layout &= ~NGP_WEIGHTS_PACKED;
	NGP_REQUIRE(out, NGP_E_ARG, "ngp_density_fwd: null out");
	if (n == 0) return 0;
	const dim3 grid(fwd_grid(n)), block(256);
	hipStream_t s = (hipStream_t)stream;
	const _Float16 *packed = pack_weights("ngp_density_fwd", s, wd, wd, 6, layout_flags); if (!packed) return NGP_E_ARG;
#define GO(T, L) NGP_LAUNCH((k_field_fwd<T, L, true>), grid, block, 0, s, n, (const _Float16 *)feat, (const float *)nullptr, 3u, packed, (T *)out, (const uint32_t *)nullptr)
	if (out_dtype == NGP_F32) { if (layout == NGP_LAYOUT_SOA) GO(float, NGP_LAYOUT_SOA); else GO(float, NGP_LAYOUT_AOS); }
	else { if (layout == NGP_LAYOUT_SOA) GO(__half, NGP_LAYOUT_SOA); else GO(__half, NGP_LAYOUT_AOS); }
#undef GO
	NGP_LAUNCH_CHECK("ngp_density_fwd");
	return 0;
}
NGP_API int ngp_field_pack_weights(void *stream, const void *wd, const void *wc, void *packed_out) {
	NGP_REQUIRE(wd && wc && packed_out, NGP_E_ARG, "ngp_field_pack_weights: null pointer");
	NGP_REQUIRE(((uintptr_t)packed_out & 15) == 0, NGP_E_ALIGN, "ngp_field_pack_weights: output must be 16-byte aligned");
	const int n_frags = N_FWD_FRAGS + N_BWD_FRAGS;
	NGP_LAUNCH(k_pack_frags, dim3(div_up((uint32_t)n_frags * 512u, 256u)), dim3(256), 0, (hipStream_t)stream, (const _Float16 *)wd, (const _Float16 *)wc, (_Float16 *)packed_out, n_frags);
	NGP_LAUNCH_CHECK("ngp_field_pack_weights");
	return 0;
}
NGP_API int ngp_field_bwd_slabs(uint32_t n) { uint32_t b = div_up(n, BT); return (int)(b < 256 ? (b ? b : 1) : 256); }
NGP_API int ngp_field_bwd(void *stream, uint32_t n, const void *feat, int layout, const float *dir, uint32_t dir_stride, const void *wd, const void *wc,
                          const void *dLdout, int out_dtype, void *dLdfeat, float *wgrad_slabs, uint32_t n_slabs, const uint32_t *n_valid) {
	return ngp_field_bwd_am(stream, n, feat, layout, dir, dir_stride, wd, wc, dLdout, out_dtype, dLdfeat, wgrad_slabs, n_slabs, n_valid, nullptr);
}
int ngp_field_bwd_am(void *stream, uint32_t n, const void *feat, int layout, const float *dir, uint32_t dir_stride, const void *wd, const void *wc,
                     const void *dLdout, int out_dtype, void *dLdfeat, float *wgrad_slabs, uint32_t n_slabs, const uint32_t *n_valid, const AbsmaxOut *am_in) {
	const AbsmaxOut am = am_in ? *am_in : AbsmaxOut{nullptr, nullptr, 0u, nullptr};
	int rc = check_field("ngp_field_bwd", feat, wd, wc, layout, out_dtype); if (rc) return rc;
	const int layout_flags = layout; layout &= ~NGP_WEIGHTS_PACKED;
	NGP_REQUIRE(dir && dLdout && dLdfeat && wgrad_slabs && dir_stride >= 3, NGP_E_ARG, "ngp_field_bwd: null pointer");
	NGP_REQUIRE((int)n_slabs == ngp_field_bwd_slabs(n), NGP_E_ARG, "ngp_field_bwd: n_slabs %u != ngp_field_bwd_slabs(%u)", n_slabs, n);
	if (n == 0) return 0;
	static const bool tr = [] { const char *e = getenv("NGP_FIELD_TRSTAGE"); return !(e && e[0] == '0'); }();      // A/B hook: NGP_FIELD_TRSTAGE=0 = rounds 1-5's [neuron][sample] staging image (same bits)
	const size_t shmem = ((N_FWD_FRAGS + N_BWD_FRAGS) * 512 + (tr ? BT * FT_ROW : N_ROWS * RS)) * sizeof(_Float16);
	const dim3 grid(n_slabs), block(512);
	hipStream_t s = (hipStream_t)stream;
	const _Float16 *packed = pack_weights("ngp_field_bwd", s, wd, wc, N_FWD_FRAGS + N_BWD_FRAGS, layout_flags); if (!packed) return NGP_E_ARG;
#define GOT(T, L, R) do { \
	static bool attr_set = false; \
	if (!attr_set) { hipError_t e = hipFuncSetAttribute((const void *)k_field_bwd<T, L, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
		if (e != hipSuccess) { ngp_set_error("ngp_field_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; } attr_set = true; } \
	NGP_LAUNCH((k_field_bwd<T, L, R>), grid, block, shmem, s, n, (const _Float16 *)feat, dir, dir_stride, packed, (const T *)dLdout, (_Float16 *)dLdfeat, wgrad_slabs, n_valid, am); } while (0)
#define GO(T, L) do { if (tr) GOT(T, L, true); else GOT(T, L, false); } while (0)
	if (out_dtype == NGP_F32) { if (layout == NGP_LAYOUT_SOA) GO(float, NGP_LAYOUT_SOA); else GO(float, NGP_LAYOUT_AOS); }
	else { if (layout == NGP_LAYOUT_SOA) GO(__half, NGP_LAYOUT_SOA); else GO(__half, NGP_LAYOUT_AOS); }
#undef GO
#undef GOT
	NGP_LAUNCH_CHECK("ngp_field_bwd");
	return 0;
}
NGP_API int ngp_reduce_slabs(void *stream, const float *slabs, uint32_t n_slabs, uint32_t width, float *out, int accumulate) {
	NGP_REQUIRE(slabs && out, NGP_E_ARG, "ngp_reduce_slabs: null pointer");
	NGP_LAUNCH(k_reduce_slabs, dim3(div_up(width, 64)), dim3(1024), 0, (hipStream_t)stream, slabs, n_slabs, width, out, accumulate);
	NGP_LAUNCH_CHECK("ngp_reduce_slabs");
	return 0;
}
// internal (csrc/train_step.hip): ngp_reduce_slabs (overwrite) + ngp_adam_ema_step of the two weight packs whose gradients tile `out`, one launch
int ngp_reduce_slabs_sweep(void *stream, const float *slabs, uint32_t n_slabs, uint32_t width, float *out, const float *const pk[2][5] /* p, m, v, ema, p_half */, const uint32_t begin[2],
                           const uint32_t count[2], float lr, float beta0, float beta1, float eps, uint32_t step, float ema_decay) {
	NGP_REQUIRE(slabs && out && pk && step >= 1, NGP_E_ARG, "ngp_reduce_slabs_sweep: bad arguments");
	PackSweep w[2];
	for (int k = 0; k < 2; ++k) {
		NGP_REQUIRE(pk[k][0] && pk[k][1] && pk[k][2] && begin[k] + count[k] <= width, NGP_E_ARG, "ngp_reduce_slabs_sweep: bad pack %d", k);
		w[k] = PackSweep{(float *)pk[k][0], (float *)pk[k][1], (float *)pk[k][2], (float *)pk[k][3], (__half *)pk[k][4], begin[k], count[k]};
	}
	NGP_REQUIRE(w[0].begin + w[0].count <= w[1].begin, NGP_E_ARG, "ngp_reduce_slabs_sweep: packs must be ordered and disjoint");
	const AdamConsts c = adam_consts(lr, beta0, beta1, eps, step, ema_decay, 1.0f);
	NGP_LAUNCH(k_reduce_slabs_sweep, dim3(div_up(width, 64)), dim3(1024), 0, (hipStream_t)stream, slabs, n_slabs, width, out, w[0], w[1], c);
	NGP_LAUNCH_CHECK("ngp_reduce_slabs_sweep");
	return 0;
}
NGP_API int ngp_sh_encode(void *stream, uint32_t n, const float *dir, uint32_t stride, void *out, int dtype) {
	NGP_REQUIRE(dir && out && stride >= 3, NGP_E_ARG, "ngp_sh_encode: bad arguments");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_sh_encode: bad dtype %d", dtype);
	if (n == 0) return 0;
	if (dtype == NGP_F32) NGP_LAUNCH(k_sh<float>, dim3(div_up(n, 128)), dim3(128), 0, (hipStream_t)stream, n, dir, stride, (float *)out);
	else NGP_LAUNCH(k_sh<__half>, dim3(div_up(n, 128)), dim3(128), 0, (hipStream_t)stream, n, dir, stride, (__half *)out);
	NGP_LAUNCH_CHECK("ngp_sh_encode");
	return 0;
}
NGP_API int ngp_selftest_mfma(void *stream, uint32_t *result) {
	NGP_REQUIRE(result, NGP_E_ARG, "ngp_selftest_mfma: null pointer");
	hipError_t e = hipMemsetAsync(result, 0, 16, (hipStream_t)stream);
	if (e != hipSuccess) { ngp_set_error("ngp_selftest_mfma: %s", hipGetErrorString(e)); return (int)e; }
	NGP_LAUNCH(k_selftest_mfma, dim3(1), dim3(64), 0, (hipStream_t)stream, result);
	NGP_LAUNCH_CHECK("ngp_selftest_mfma");
	return 0;
}
