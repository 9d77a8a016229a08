// The fp32 field network (NGPNetworks.execute_ / .density without cfg.fp16: models/networks/ngp_network.py:57-67, 77-89 - what ngp_base.py runs), forward AND backward,
// on the fp16 matrix cores at fp32 accuracy: split operands, three MFMAs per product sum (field_split.h).  Same "transposed" register-resident formulation as
// field_mlp.hip (weights = A operand from pre-permuted LDS fragments, 16 samples of a wave tile = B columns, a layer's C fragment is the next layer's B fragment),
// fp32 features / gradients in, fp32 outputs out; between layers the fp32 accumulator is ReLU'd (masked) and split again in registers.
// Default kernels of ngp_field32_fwd / ngp_density32_fwd / ngp_field32_bwd since round 3 (forward 55 -> 31 us, backward 139 -> 109 us per 2^18-sample batch); the
// exact-product kernels of field32.hip (v_mfma_f32_16x16x4_f32) stay selectable: NGP_FIELD32_FWD=mfma32, NGP_FIELD32_BWD=2.  Against an fp64 evaluation both pairs
// are fp32-accurate in the forward (2.2e-7 of the output scale) and the split pair is the closer one in the gradients (profiles/r03_split_accuracy.md).
#include "ngp_common.h"
#include "field_split.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
// fp16 spans 6e-5 .. 65504 (normal range): B operands are multiplied by a power of two before the split and the factor is taken out of the fp32 accumulator again (both
// exact).  Hash features start at ~1e-4 and stay below ~10: x 256 (safe up to 255); hidden activations / density logits / SH: x 16 (safe up to 4094; accurate down to a
// tensor scale of ~4e-6).  Values beyond the safe maxima would overflow to infinity - the exact-product kernels of field32.hip (NGP_FIELD32_FWD=mfma32) have no such
// range.  (r4) No silent infinity: the forward kernels keep the largest prescaled operand they split (four v_max3 per eight operands) and raise a device-side flag -
// bit 0 once an operand comes within a factor four of fp16's largest finite value (features > 63.9, activations > 1023: nothing has overflowed yet), bit 1 when one
// exceeded it (the results of that launch contain infinities); the backward raises bit 1 when a feature gradient it stores is not finite.  ngp_field32_range_check
// reads (and optionally clears) the flag; Runner polls it where it synchronises anyway (every 16th step, after every rendered image): bit 0 switches the process to
// the exact-product kernels through ngp_field32_select before anything overflowed, bit 1 is an error.
#define FEAT_PRESCALE 256.0f
#define HID_PRESCALE 16.0f
#define SPLIT_RANGE_MAX 65504.0f
#define SPLIT_RANGE_NEAR (SPLIT_RANGE_MAX / 4.0f)
__device__ uint32_t g_split_range_flag = 0u;
// rmax: two packed 16-bit maxima of |h| bit patterns (fp16: 0x7bff = 65504, 0x7c00 = infinity, above = NaN; 0x73ff = 16376 = a quarter of the range)
__device__ __forceinline__ void range_report(uint32_t rmax) {
	const uint32_t m = max(rmax & 0xffffu, rmax >> 16);
	if (m > 0x73ffu) atomicOr(&g_split_range_flag, m > 0x7bffu ? 3u : 1u);
}

__global__ __launch_bounds__(256) void k_pack_split(const float *__restrict__ wd, const float *__restrict__ wc, _Float16 *__restrict__ out, int n_frags) {
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx < NSPLIT_HALVES && ((idx % (NSPLIT_FRAGS * 512)) >> 9) < n_frags) out[idx] = split_frag_half(wd, wc, idx);
}
// n_frags = 6: the density network only (wc is not read), NSPLIT_FRAGS: everything (forward + transposed)
int ngp_field32_pack_split(void *stream, const float *wd, const float *wc, void *out_halves, int n_frags) {
	NGP_LAUNCH(k_pack_split, dim3(div_up(NSPLIT_HALVES, 256)), dim3(256), 0, (hipStream_t)stream, wd, wc, (_Float16 *)out_halves, n_frags);
	NGP_LAUNCH_CHECK("ngp_field32_pack_split");
	return 0;
}

struct B2 { half8 h, m; };
// (r6b) Packed register arithmetic, written out.  Left to the compiler the split of eight operands was ~3 vector instructions per operand plus re-packing moves (its SLP pass
// pairs the residuals' fused multiply-adds into v_pk_fma_f32, which takes no fp16 operand: a conversion of h back to fp32 and a second conversion of the residual per operand,
// and pairs taken across register-pair boundaries).  Here two operands are: one v_cvt_pk_f16_f32 (h, h'), and one v_fma_mix{lo,hi}_f16 each for m = fp16(fma(h, -2^11, v 2^11))
// - the fp16 h read in place, the fp16 result written in place - on v 2^11 from a packed multiply.  Same bits as rounds 3-6a (same roundings in the same order).
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t cvt_pk(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector((float2v){a, b}, half2v)); }
// The residuals are ONE asm statement per operand set, for three reasons the compiler cannot be told otherwise: (1) v_fma_mixhi_f16 must land in the register its mixlo
// partner wrote; (2) gfx950 wants two wait states between a vector instruction's register write and a matrix instruction that reads it, and the compiler - which inserts
// them for its own instructions - does not look into asm: the statement ends in s_nop 1 (2 cycles per eight operands); (3) a 16-bit partial write followed at once by a read
// of the register is a forwarding hazard the compiler also handles only for its own instructions: all low halves first, then all high halves (>= 2 instructions apart).
// The inputs come from v_cvt_pk_f16_f32 / v_pk_mul_f32 (vector ALU -> vector ALU is interlocked in hardware), never straight from a matrix instruction.
// slots 0..3 <- a, 4..7 <- b  (wa / wb: the same values times 2^11)
__device__ __forceinline__ B2 split_x4(floatx4 a, floatx4 b, floatx4 wa, floatx4 wb) {
	const uint32_t h0 = cvt_pk(a[0], a[1]), h1 = cvt_pk(a[2], a[3]), h2 = cvt_pk(b[0], b[1]), h3 = cvt_pk(b[2], b[3]);
	uint32_t m0, m1, m2, m3;
	const float ns = -SPLIT_SCALE;
	asm("v_fma_mixlo_f16 %0, %4, %16, %8 op_sel_hi:[1,0,0]\n\t"
	    "v_fma_mixlo_f16 %1, %5, %16, %10 op_sel_hi:[1,0,0]\n\t"
	    "v_fma_mixlo_f16 %2, %6, %16, %12 op_sel_hi:[1,0,0]\n\t"
	    "v_fma_mixlo_f16 %3, %7, %16, %14 op_sel_hi:[1,0,0]\n\t"
	    "v_fma_mixhi_f16 %0, %4, %16, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
	    "v_fma_mixhi_f16 %1, %5, %16, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
	    "v_fma_mixhi_f16 %2, %6, %16, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
	    "v_fma_mixhi_f16 %3, %7, %16, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
	    "s_nop 1"
	    : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3)
	    : "v"(h0), "v"(h1), "v"(h2), "v"(h3), "v"(wa[0]), "v"(wa[1]), "v"(wa[2]), "v"(wa[3]), "v"(wb[0]), "v"(wb[1]), "v"(wb[2]), "v"(wb[3]), "s"(ns));
	B2 r; r.h = __builtin_bit_cast(half8, (uint4v){h0, h1, h2, h3}); r.m = __builtin_bit_cast(half8, (uint4v){m0, m1, m2, m3});
	return r;
}
// slots 0..3 <- a, 4..7 zero
__device__ __forceinline__ B2 split_x4_low(floatx4 a, floatx4 wa) {
	const uint32_t h0 = cvt_pk(a[0], a[1]), h1 = cvt_pk(a[2], a[3]);
	uint32_t m0, m1;
	const float ns = -SPLIT_SCALE;
	asm("v_fma_mixlo_f16 %0, %2, %8, %4 op_sel_hi:[1,0,0]\n\t"
	    "v_fma_mixlo_f16 %1, %3, %8, %6 op_sel_hi:[1,0,0]\n\t"
	    "s_nop 0\n\t"
	    "v_fma_mixhi_f16 %0, %2, %8, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
	    "v_fma_mixhi_f16 %1, %3, %8, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
	    "s_nop 1"
	    : "=&v"(m0), "=&v"(m1)
	    : "v"(h0), "v"(h1), "v"(wa[0]), "v"(wa[1]), "v"(wa[2]), "v"(wa[3]), "s"(ns));
	B2 r; r.h = __builtin_bit_cast(half8, (uint4v){h0, h1, 0u, 0u}); r.m = __builtin_bit_cast(half8, (uint4v){m0, m1, 0u, 0u});
	return r;
}
__device__ __forceinline__ B2 split8(const float v[8]) {
	const floatx4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
	return split_x4(a, b, a * SPLIT_SCALE, b * SPLIT_SCALE);
}
// ReLU + prescale + split of two accumulator tiles given RAW (main + corr 2^-11, see combine_raw): operand = max(t, 0) 2^E.  Rounds 3-6a formed ((t S) relu) P with the
// layer's combine scale S and the next layer's prescale P, two exact multiplications by powers of two; 2^E = S P is the same number (E = -4 behind the first layer: 2^-8 2^4;
// 0 behind the hidden layers: 2^-4 2^4 - no multiplication at all).
template <int E>
__device__ __forceinline__ B2 split_relu_raw(floatx4 ta, floatx4 tb) {
	const floatx4 z = {0.f, 0.f, 0.f, 0.f};
	floatx4 va = __builtin_elementwise_max(ta, z), vb = __builtin_elementwise_max(tb, z);
	const floatx4 wa = va * (SPLIT_SCALE * __builtin_ldexpf(1.0f, E)), wb = vb * (SPLIT_SCALE * __builtin_ldexpf(1.0f, E));
	if (E != 0) { va = va * __builtin_ldexpf(1.0f, E); vb = vb * __builtin_ldexpf(1.0f, E); }
	return split_x4(va, vb, wa, wb);
}
// the operands' magnitudes folded into rmax: |h| as bit patterns order like the magnitudes (an overflowed operand reads 0x7c00): a packed unsigned 16-bit maximum over the
// four registers of the h part
typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void range_fold(const B2 &r, uint32_t &rmax) {
	const uint4v q = __builtin_bit_cast(uint4v, r.h) & 0x7fff7fffu;
	ushort2v m = __builtin_bit_cast(ushort2v, rmax);
#pragma unroll
	for (int k = 0; k < 4; ++k) m = __builtin_elementwise_max(m, __builtin_bit_cast(ushort2v, q[k]));
	rmax = __builtin_bit_cast(uint32_t, m);
}
struct Acc { floatx4 main, corr; };
__device__ __forceinline__ half8 ld_half8(const _Float16 *lds, int f, int lane) { return *reinterpret_cast<const half8 *>(lds + f * 512 + lane * 8); }
// acc += W-fragment f x B: the leading product and the two cross terms (kept in an accumulator of their own: they are 2^-11 smaller)
// (wl: the h parts of a fragment set in LDS, the m parts `mstride` halves behind them)
template <int MSTRIDE>
__device__ __forceinline__ void mma3(const _Float16 *wl, int f, int lane, const B2 &b, Acc &acc) {
	const half8 ah = ld_half8(wl, f, lane), am = ld_half8(wl + MSTRIDE, f, lane);
	acc.main = MFMA16(ah, b.h, acc.main);
	acc.corr = MFMA16(ah, b.m, acc.corr);
	acc.corr = MFMA16(am, b.h, acc.corr);
}
// main + corr 2^-11 as ONE fused multiply-add per register pair (v_pk_fma_f32): the product by a power of two is exact, so this is the separate multiply and add of rounds
// 3-6a bit for bit; a power-of-two scale on top commutes with the rounding
__device__ __forceinline__ floatx4 combine_raw(const Acc &a) {
	const floatx4 k = {1.0f / SPLIT_SCALE, 1.0f / SPLIT_SCALE, 1.0f / SPLIT_SCALE, 1.0f / SPLIT_SCALE};
	return __builtin_elementwise_fma(a.corr, k, a.main);
}
__device__ __forceinline__ floatx4 combine(const Acc &a, float scale) { return combine_raw(a) * scale; }

// degree-4 SH of (2d-1), components 4g..4g+3 (SphericalEncode.h:77-95)
__device__ __forceinline__ void sh4_split(const float d[3], int g, float o[4]) {
	const float x = d[0] * 2.f - 1.f, y = d[1] * 2.f - 1.f, z = d[2] * 2.f - 1.f;
	const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
	if (g == 0) { o[0] = 0.28209479177387814f; o[1] = -0.48860251190291987f * y; o[2] = 0.48860251190291987f * z; o[3] = -0.48860251190291987f * x; }
	else if (g == 1) { o[0] = 1.0925484305920792f * xy; o[1] = -1.0925484305920792f * yz; o[2] = 0.94617469575755997f * z2 - 0.31539156525251999f; o[3] = -1.0925484305920792f * xz; }
	else if (g == 2) { o[0] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2; o[1] = 0.59004358992664352f * y * (-3.0f * x2 + y2); o[2] = 2.8906114426405538f * xy * z; o[3] = 0.45704579946446572f * y * (1.0f - 5.0f * z2); }
	else { o[0] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f); o[1] = 0.45704579946446572f * x * (1.0f - 5.0f * z2); o[2] = 1.4453057213202769f * z * (x2 - y2); o[3] = 0.59004358992664352f * x * (-x2 + 3.0f * y2); }
}

// features 8g..8g+7 of sample i (levels 4g..4g+3): slot j of the first layer's B fragment (k order sp_k32)
template <int LAYOUT>
__device__ __forceinline__ void load_feat_split(const float *__restrict__ feat, uint32_t n, uint32_t i, int g, float f[8]) {
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

template <bool DENSITY_ONLY, int MSTRIDE>
__device__ __forceinline__ void forward_split(const _Float16 *wl, int lane, const float feat[8], const float sh[4], floatx4 &den, floatx4 &rgb, uint32_t &rmax) {
	const floatx4 z = {0.f, 0.f, 0.f, 0.f};
	float fs[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) fs[k] = feat[k] * FEAT_PRESCALE;
	const B2 b0 = split8(fs); range_fold(b0, rmax);
	floatx4 c0[4];
#pragma unroll
	for (int t = 0; t < 4; ++t) { Acc a = {z, z}; mma3<MSTRIDE>(wl, t, lane, b0, a); c0[t] = combine_raw(a); }                                     // L0: 32 -> 64  (raw: x 2^8)
	const B2 h0 = split_relu_raw<-4>(c0[0], c0[1]), h1 = split_relu_raw<-4>(c0[2], c0[3]); range_fold(h0, rmax); range_fold(h1, rmax);
	floatx4 dr;                                                                                                                                      // L1: 64 -> 16  (raw: x 2^4)
	{ Acc a = {z, z}; mma3<MSTRIDE>(wl, 4, lane, h0, a); mma3<MSTRIDE>(wl, 5, lane, h1, a); dr = combine_raw(a); den = dr * (1.0f / HID_PRESCALE); }
	if (DENSITY_ONLY) return;
	const floatx4 shv = {sh[0], sh[1], sh[2], sh[3]};
	const floatx4 shs = shv * HID_PRESCALE;
	const B2 b2 = split_x4(dr, shs, dr * SPLIT_SCALE, shs * SPLIT_SCALE); range_fold(b2, rmax);                                                      // den 2^4 = the raw value
	floatx4 c2[4];
#pragma unroll
	for (int t = 0; t < 4; ++t) { Acc a = {z, z}; mma3<MSTRIDE>(wl, 6 + t, lane, b2, a); c2[t] = combine_raw(a); }                                  // L2: [density(16) | SH(16)] -> 64
	const B2 g00 = split_relu_raw<0>(c2[0], c2[1]), g01 = split_relu_raw<0>(c2[2], c2[3]); range_fold(g00, rmax); range_fold(g01, rmax);
	floatx4 c3[4];
#pragma unroll
	for (int t = 0; t < 4; ++t) { Acc a = {z, z}; mma3<MSTRIDE>(wl, 10 + 2 * t, lane, g00, a); mma3<MSTRIDE>(wl, 11 + 2 * t, lane, g01, a); c3[t] = combine_raw(a); }   // L3: 64 -> 64
	const B2 g10 = split_relu_raw<0>(c3[0], c3[1]), g11 = split_relu_raw<0>(c3[2], c3[3]); range_fold(g10, rmax); range_fold(g11, rmax);
	{ Acc a = {z, z}; mma3<MSTRIDE>(wl, 18, lane, g10, a); mma3<MSTRIDE>(wl, 19, lane, g11, a); rgb = combine(a, 1.0f / HID_PRESCALE); }             // L4: 64 -> 16 (3 used)
}

template <int LAYOUT, bool DENSITY_ONLY>
__global__ __launch_bounds__(256, 2) void k_field32_fwd_split(uint32_t n, const float *__restrict__ feat, const float *__restrict__ dir, uint32_t dir_stride,
                                                           const _Float16 *__restrict__ packed, float *__restrict__ out, const uint32_t *__restrict__ n_valid) {
	__shared__ __attribute__((aligned(16))) _Float16 wl[2 * NSPLIT_FWD * 512];
	{	// both parts of the forward fragments this variant reads (density only: layers 0 and 1); the buffer holds NSPLIT_FRAGS fragments per part
		const int nf = DENSITY_ONLY ? 6 : NSPLIT_FWD;
		const uint4 *src = reinterpret_cast<const uint4 *>(packed);
		uint4 *dst = reinterpret_cast<uint4 *>(wl);
		for (int idx = threadIdx.x; idx < nf * 64; idx += 256) { dst[idx] = src[idx]; dst[NSPLIT_FWD * 64 + idx] = src[NSPLIT_FRAGS * 64 + idx]; }
	}
	__syncthreads();
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	const int lane = threadIdx.x & 63, s = lane & 15, g = lane >> 4;
	const uint32_t n_tiles = (lim + 15u) / 16u;
	const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
	auto fetch = [&](uint32_t tile, float f[8], float d[3]) {
		const uint32_t i = tile * 16u + s;
		const uint32_t ic = i < lim ? i : lim - 1;
		load_feat_split<LAYOUT>(feat, n, ic, g, f);
		if (!DENSITY_ONLY) { d[0] = dir[(size_t)ic * dir_stride]; d[1] = dir[(size_t)ic * dir_stride + 1]; d[2] = dir[(size_t)ic * dir_stride + 2]; }
	};
	float f[8], fn[8], d[3] = {0.f, 0.f, 0.f}, dn[3] = {0.f, 0.f, 0.f};
	uint32_t rmax = 0u;                                             // largest prescaled operand this lane split, as packed fp16 bit patterns (range_report)
	if (wave < n_tiles) fetch(wave, f, d);
	for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
		const uint32_t i = tile * 16u + s;
		const bool more = tile + n_waves < n_tiles;
		if (more) fetch(tile + n_waves, fn, dn);                   // the next tile's inputs are in flight during this tile's chain
		float sh[4] = {0.f, 0.f, 0.f, 0.f};
		if (!DENSITY_ONLY) sh4_split(d, g, sh);
		floatx4 den, rgb;
		forward_split<DENSITY_ONLY, NSPLIT_FWD * 512>(wl, lane, f, sh, den, rgb, rmax);
		if (g == 0 && i < lim) {
			if (DENSITY_ONLY) out[i] = den[0];
			else *reinterpret_cast<float4 *>(out + (size_t)i * 4) = make_float4(rgb[0], rgb[1], rgb[2], den[0]);
		}
		if (more) {
#pragma unroll
			for (int q = 0; q < 8; ++q) f[q] = fn[q];
			d[0] = dn[0]; d[1] = dn[1]; d[2] = dn[2];
		}
	}
	range_report(rmax);
}

// ---------------------------------------------------------------------------------------------------------------- backward
// dL/dfeatures + the five weight gradients of the fp32 network on split fp16 operands: forward recompute (as above, without the colour head's last layer), the
// register-resident dgrad chain on the transposed fragments, and the weight-gradient contraction over samples through LDS - every matrix product three MFMAs.
// Gradients have no fixed magnitude (dL/dout carries the loss scale and the compositing weights: 1e-8 .. 1e-2), so every 128-sample trip derives a power of two
// sigma from the largest |dL/dout| of ITS samples (one extra workgroup barrier), runs the chain on sigma-scaled gradients (max ~2^3: room for a 4000-fold growth
// through the three transposed layers before fp16 overflows) and takes sigma out again - exactly - when dL/dfeatures is stored and when a trip's weight-gradient
// tiles are folded into the persistent fp32 accumulators.  Activations carry the forward kernel's prescales (features 2^8, everything else 2^4).
// LDS: 2 x 42 fragments (84 KiB) + a staging region of 2 planes (h, m) x 128 rows x 136 halves (68 KiB); five staging phases per trip like k_field32_bwd:
//   A : dG1 0..63 | G0 64..127  -> V1     B1: dH 0..63 | F 64..95 -> W0     B2: dG0 0..63 | IN2 64..95 -> V0
//   C1: dD 0..15 | H 16..79     -> W1     C2: dO 0..15 | G1 16..79 -> V2    (W1 / V2: tile w & 3 over the samples of half w >> 2, joined at the end)
#define SBT 128
#define SRS (SBT + 8)                         // row stride in halves (272 B: 16-byte aligned rows, breaks the 128-byte bank period)
#define SROWS 128
#define SPLANE (SROWS * SRS)
#define GRAD_TARGET_EXP 3                     // sigma brings the trip's largest |dL/dout| to [2^3, 2^4): room for a 4000-fold growth through the three transposed layers before fp16
                                              // overflows (2^7 was as accurate - powers of two are exact - but left only a 250-fold margin)

__device__ __forceinline__ B2 split_masked(floatx4 a, floatx4 b, uint32_t mask) {           // relu'(pre-activation) * gradient, split
	floatx4 va, vb;
#pragma unroll
	for (int k = 0; k < 4; ++k) { va[k] = (mask >> k) & 1u ? a[k] : 0.f; vb[k] = (mask >> (4 + k)) & 1u ? b[k] : 0.f; }
	return split_x4(va, vb, va * SPLIT_SCALE, vb * SPLIT_SCALE);
}
__device__ __forceinline__ uint32_t relu_mask8(floatx4 a, floatx4 b) {
	uint32_t m = 0;
#pragma unroll
	for (int k = 0; k < 4; ++k) { m |= a[k] > 0.f ? (1u << k) : 0u; m |= b[k] > 0.f ? (1u << (4 + k)) : 0u; }
	return m;
}
// the lane's eight slots of two k64 half-fragments (64 neurons) / one k32 fragment / the low four slots into the rows of both planes, column `col`
__device__ __forceinline__ void st_rows64_2(_Float16 *stage, int row0, int col, int g, const B2 &lo, const B2 &hi) {
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const int r0 = (row0 + sp_k64(0, g, j)) * SRS + col, r1 = (row0 + sp_k64(1, g, j)) * SRS + col;
		stage[r0] = lo.h[j]; stage[SPLANE + r0] = lo.m[j]; stage[r1] = hi.h[j]; stage[SPLANE + r1] = hi.m[j];
	}
}
__device__ __forceinline__ void st_rows32_2(_Float16 *stage, int row0, int col, int g, const B2 &v, bool k32_order) {
#pragma unroll
	for (int j = 0; j < 8; ++j) { const int r = (row0 + (k32_order ? sp_k32(g, j) : sp_k64(0, g, j))) * SRS + col; stage[r] = v.h[j]; stage[SPLANE + r] = v.m[j]; }
}
__device__ __forceinline__ void st_rows16_2(_Float16 *stage, int row0, int col, int g, const B2 &v) {
#pragma unroll
	for (int j = 0; j < 4; ++j) { const int r = (row0 + 4 * g + j) * SRS + col; stage[r] = v.h[j]; stage[SPLANE + r] = v.m[j]; }
}
// one 16x16 weight-gradient tile over the staged samples [c0, c1): A = gradient rows row_a + o, B = activation rows row_b + o, k = 32 samples per step
__device__ __forceinline__ floatx4 wgrad3(const _Float16 *stage, int row_a, int row_b, int o, int g, int c0, int c1) {
	const floatx4 z = {0.f, 0.f, 0.f, 0.f};
	Acc acc = {z, z};
#pragma unroll 2
	for (int c = c0; c < c1; c += 32) {      // (full unrolling lets the scheduler hoist every step's four fragment loads of every tile of a phase: 290 VGPRs wanted, 36 spilled)
		const int cs = c + 8 * g;
		const half8 ah = *reinterpret_cast<const half8 *>(stage + (row_a + o) * SRS + cs), am = *reinterpret_cast<const half8 *>(stage + SPLANE + (row_a + o) * SRS + cs);
		const half8 bh = *reinterpret_cast<const half8 *>(stage + (row_b + o) * SRS + cs), bm = *reinterpret_cast<const half8 *>(stage + SPLANE + (row_b + o) * SRS + cs);
		acc.main = MFMA16(ah, bh, acc.main);
		acc.corr = MFMA16(ah, bm, acc.corr);
		acc.corr = MFMA16(am, bh, acc.corr);
	}
	return combine_raw(acc);
}
// ---- (r6) the transposed staging image.  The image above is [neuron row][sample column]: a lane owns ONE sample and eight neurons of every operand, so it writes eight 2-byte
// stores per operand half - 240 ds_write_b16 per lane and trip, 26 of the kernel's 120 us (profiles/r05a_split_probe.txt).  Here the image is [sample row][neuron]: a lane's four
// consecutive neurons are ONE 8-byte store (60 per lane and trip), and the weight-gradient phases read it back through the hardware transpose read (ds_read_b64_tr_b16:
// sixteen lanes fetch a [4 samples][16 neurons] block, lane i receives neuron i's four samples) - the same operand in the same k slots as before, so the results are the
// same bits.  Rows are 256 bytes (128 neurons), no padding: 2 planes x 128 x 256 B = 64 KiB; the chunk index (four neurons = 8 bytes) is XORed with a function of the row so
// that both access patterns spread over the banks: stores (16 lanes = 16 consecutive rows, one chunk) need f mod 16 distinct over 16 rows, transpose reads (16 lanes = 4 rows
// x 4 chunks, two such groups 8 rows apart per cycle) need c ^ f distinct over the 32 (tools/microbench_trstage.hip measures both against the linear image).
#define TPLANE (SROWS * 128)
typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ int tr_f(int s) { return ((s & 3) << 2) | ((s >> 2) & 3) | (((s >> 3) & 1) << 4); }
__device__ __forceinline__ int tr_off(int plane, int s, int c) { return plane * TPLANE + s * 128 + ((c ^ tr_f(s)) << 2); }      // halves
struct alignas(8) H4 { _Float16 v[4]; };
__device__ __forceinline__ void st_chunk(_Float16 *stage, int s, int c, const half8 &h, const half8 &m, int hi) {               // slots 4 hi .. 4 hi + 3 of both halves of an operand
	H4 a, b;
#pragma unroll
	for (int j = 0; j < 4; ++j) { a.v[j] = h[4 * hi + j]; b.v[j] = m[4 * hi + j]; }
	*reinterpret_cast<H4 *>(stage + tr_off(0, s, c)) = a; *reinterpret_cast<H4 *>(stage + tr_off(1, s, c)) = b;
}
// the lane's eight slots of two k64 half-fragments (neurons nrow0 .. nrow0 + 63) / of one k64 or k32 fragment / the low four slots, as the row `s` of both planes
__device__ __forceinline__ void st_rows64_T(_Float16 *stage, int nrow0, int s, int g, const B2 &lo, const B2 &hi) {
	st_chunk(stage, s, (nrow0 >> 2) + g, lo.h, lo.m, 0); st_chunk(stage, s, (nrow0 >> 2) + 4 + g, lo.h, lo.m, 1);                  // sp_k64(0, g, j): 4g + j | 16 + 4g + (j - 4)
	st_chunk(stage, s, (nrow0 >> 2) + 8 + g, hi.h, hi.m, 0); st_chunk(stage, s, (nrow0 >> 2) + 12 + g, hi.h, hi.m, 1);            // sp_k64(1, g, j): 32 + ...
}
__device__ __forceinline__ void st_rows32_T(_Float16 *stage, int nrow0, int s, int g, const B2 &v, bool k32_order) {
	if (k32_order) { st_chunk(stage, s, (nrow0 >> 2) + 2 * g, v.h, v.m, 0); st_chunk(stage, s, (nrow0 >> 2) + 2 * g + 1, v.h, v.m, 1); }       // sp_k32(g, j) = 8g + j
	else { st_chunk(stage, s, (nrow0 >> 2) + g, v.h, v.m, 0); st_chunk(stage, s, (nrow0 >> 2) + 4 + g, v.h, v.m, 1); }
}
__device__ __forceinline__ void st_rows16_T(_Float16 *stage, int nrow0, int s, int g, const B2 &v) { st_chunk(stage, s, (nrow0 >> 2) + g, v.h, v.m, 0); }
__device__ __forceinline__ half8 ld_tr8(const _Float16 *stage, int plane, int nrow, int o, int s0) {          // neuron nrow + o, samples s0 .. s0 + 7 (s0 a multiple of 8)
	const short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v *)(stage + tr_off(plane, s0 + (o >> 2), (nrow >> 2) + (o & 3))));
	const short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v *)(stage + tr_off(plane, s0 + 4 + (o >> 2), (nrow >> 2) + (o & 3))));
	typedef short short8v __attribute__((ext_vector_type(8)));
	short8v r;
#pragma unroll
	for (int j = 0; j < 4; ++j) { r[j] = a[j]; r[4 + j] = b[j]; }
	return __builtin_bit_cast(half8, r);
}
// (r6b) The staging region starts 84 KiB into the workgroup's LDS and a DS instruction's immediate offset is 16 bits: left to the compiler, every transpose read's address was
// "lane part + 0x15000 + block" with the constant beyond the immediate's reach - one v_add_u32 per read, 20 per loop iteration.  Here the four lane-dependent addresses of a
// tile (operand a | b, sample rows +0 | +4 of the first step, plane 0) are formed once, made opaque to constant re-association, and every other block of the tile is an
// immediate behind one of them: a step = 32 sample rows = 8 KiB, plane 1 = 32 KiB, at most 0xE000.  (tr_f depends on the row's low four bits only, which a step keeps.)
// Unrolling of the weight-gradient loops (k steps of 32 samples: four per full tile).  Unrolled by two since round 3 - fully unrolled the kernel wanted 290 VGPRs then.  With the
// register arithmetic written out (above) the full unrolling (-DSPLIT_WG_UNROLL=4) fits the 256 without scratch, loses the loops' accumulator zeroing and counters (746 -> 652
// vector instructions per trip) and lets the V1 tile pair share its gradient operand's reads: measured, same bits (the same MFMAs in the same order), kernel 83.0 -> 81.9 us, step
// 2146 vs 2140 it/s (profiles/r06y_ab_unroll.txt) - not faster, so the smaller form stays.
#ifndef SPLIT_WG_UNROLL
#define SPLIT_WG_UNROLL 2
#endif
typedef __attribute__((address_space(3))) short4v lds_short4v;
__device__ __forceinline__ uint32_t tr_lane_addr(const _Float16 *stage, int s, int c) {
	uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const _Float16 *)(stage + tr_off(0, s, c));
	asm("" : "+v"(a));
	return a;
}
__device__ __forceinline__ half8 ld_tr8_at(uint32_t a0, uint32_t a4, uint32_t byte_off) {           // samples s .. s + 3 (a0) and s + 4 .. s + 7 (a4) of one neuron
	const short4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4v *)(uintptr_t)(a0 + byte_off));
	const short4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4v *)(uintptr_t)(a4 + byte_off));
	typedef short short8v __attribute__((ext_vector_type(8)));
	short8v r;
#pragma unroll
	for (int j = 0; j < 4; ++j) { r[j] = a[j]; r[4 + j] = b[j]; }
	return __builtin_bit_cast(half8, r);
}
__device__ __forceinline__ floatx4 wgrad3_T(const _Float16 *stage, int row_a, int row_b, int o, int g, int c0, int c1) {
	const floatx4 z = {0.f, 0.f, 0.f, 0.f};
	Acc acc = {z, z};
	const int s0 = c0 + 8 * g + (o >> 2);
	const uint32_t a0 = tr_lane_addr(stage, s0, (row_a >> 2) + (o & 3)), a4 = tr_lane_addr(stage, s0 + 4, (row_a >> 2) + (o & 3));
	const uint32_t b0 = tr_lane_addr(stage, s0, (row_b >> 2) + (o & 3)), b4 = tr_lane_addr(stage, s0 + 4, (row_b >> 2) + (o & 3));
	constexpr uint32_t P1 = TPLANE * 2, STEP = 32 * 128 * 2;                                          // bytes
#pragma unroll SPLIT_WG_UNROLL
	for (int c = c0; c < c1; c += 32) {
		const uint32_t off = (uint32_t)((c - c0) >> 5) * STEP;
		const half8 ah = ld_tr8_at(a0, a4, off), am = ld_tr8_at(a0, a4, off + P1);
		const half8 bh = ld_tr8_at(b0, b4, off), bm = ld_tr8_at(b0, b4, off + P1);
		acc.main = MFMA16(ah, bh, acc.main);
		acc.corr = MFMA16(ah, bm, acc.corr);
		acc.corr = MFMA16(am, bh, acc.corr);
	}
	return combine_raw(acc);
}
__device__ __forceinline__ floatx4 fma4(floatx4 acc, floatx4 v, float s) {
#pragma unroll
	for (int k = 0; k < 4; ++k) acc[k] += v[k] * s;
	return acc;
}

// PROBE (timing experiments only, a -DNGP_PROBE_SPLIT=n build, tools/probe_split_bwd.py; results are wrong for PROBE != 0): 1 = no staging stores, 2 = no weight-gradient
// loads / MFMAs, 3 = neither (chain + barriers only), 4 = no barriers either
template <int LAYOUT, int PROBE, bool TR /* (r6) the transposed staging image: 8-byte stores + transpose reads */>
__global__ __launch_bounds__(512, 1) void k_field32_bwd_split(uint32_t n, const float *__restrict__ feat, const float *__restrict__ dir, uint32_t dir_stride,
                                                              const _Float16 *__restrict__ packed, const float *__restrict__ dout,
                                                              float *__restrict__ dfeat, float *__restrict__ slabs, const uint32_t *__restrict__ n_valid, AbsmaxOut am) {
#define WG3(...) (TR ? wgrad3_T(__VA_ARGS__) : wgrad3(__VA_ARGS__))
	extern __shared__ __attribute__((aligned(16))) _Float16 smem_split[];
	__shared__ float smax[8];
	float lmax[2][2] = {{0.f, 0.f}, {0.f, 0.f}};          // running max |dL/dfeature| of levels 8t + 2g + pr over this lane's samples (absmax_epilogue)
	bool bad = false;                                      // a feature gradient that is not finite: an operand of the chain left fp16's range (g_split_range_flag bit 1)
	_Float16 *wl = smem_split;                             // [2 parts][42 fragments][512]
	_Float16 *stage = smem_split + NSPLIT_HALVES;          // [2 planes][SROWS][SRS]
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(packed);
		uint4 *dst = reinterpret_cast<uint4 *>(wl);
		for (int idx = threadIdx.x; idx < NSPLIT_HALVES / 8; idx += 512) dst[idx] = src[idx];
	}
	const _Float16 *wb = wl + NSPLIT_FWD * 512;            // transposed fragments (h parts; the m parts NSPLIT_FRAGS * 512 halves behind, like the forward ones)
	constexpr int MS = NSPLIT_FRAGS * 512;
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	const int lane = threadIdx.x & 63, s = lane & 15, g = lane >> 4, w = threadIdx.x >> 6;
	const uint32_t n_bt = (lim + SBT - 1) / SBT;
	const floatx4 z = {0.f, 0.f, 0.f, 0.f};
	const int to = w >> 1, ti0 = 2 * (w & 1), tj = w & 1, tx = w & 3, half = w >> 2;
	floatx4 aV1[2] = {z, z}, aW0 = z, aV0 = z, aW1 = z, aV2 = z;
	__syncthreads();
	struct Inputs { float f[8]; float d3[3]; float go[4]; };
	auto fetch = [&](uint32_t bt, Inputs &in) {
		const uint32_t i = bt * SBT + 16u * w + s;
		const bool valid = i < lim;
		const uint32_t ic = valid ? i : lim - 1;
		load_feat_split<LAYOUT>(feat, n, ic, g, in.f);
		in.d3[0] = dir[(size_t)ic * dir_stride]; in.d3[1] = dir[(size_t)ic * dir_stride + 1]; in.d3[2] = dir[(size_t)ic * dir_stride + 2];
		in.go[0] = in.go[1] = in.go[2] = in.go[3] = 0.f;
		if (valid) { const float4 v = *reinterpret_cast<const float4 *>(dout + (size_t)i * 4); in.go[0] = v.x; in.go[1] = v.y; in.go[2] = v.z; in.go[3] = v.w; }
	};
	Inputs cur, nxt;
	if (blockIdx.x < n_bt) fetch(blockIdx.x, cur);
	for (uint32_t bt = blockIdx.x; bt < n_bt; bt += gridDim.x) {
		const uint32_t i = bt * SBT + 16u * w + s;
		const bool valid = i < lim;
		const bool more = bt + gridDim.x < n_bt;
		if (more) fetch(bt + gridDim.x, nxt);
		// ---- the trip's gradient scale: sigma = 2^(GRAD_TARGET_EXP - floor(log2(max |dL/dout|)))
		{
			float mx = fmaxf(fmaxf(fabsf(cur.go[0]), fabsf(cur.go[1])), fmaxf(fabsf(cur.go[2]), fabsf(cur.go[3])));     // (every lane group loaded the sample's four values)
#pragma unroll
			for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
			if (lane == 0) smax[w] = mx;
		}
		__syncthreads();
		float sigma = 1.0f, inv_sigma = 1.0f;
		{
			float mx = 0.f;
#pragma unroll
			for (int k = 0; k < 8; ++k) mx = fmaxf(mx, smax[k]);
			const uint32_t e = (__float_as_uint(mx) >> 23) & 0xffu;                   // biased exponent (0: zero / subnormal maximum - leave the gradients alone)
			if (e != 0u && e != 0xffu) {
				int sb = 127 + GRAD_TARGET_EXP - ((int)e - 127);
				sb = sb < 1 ? 1 : (sb > 253 ? 253 : sb);
				sigma = __uint_as_float((uint32_t)sb << 23); inv_sigma = __uint_as_float((uint32_t)(254 - sb) << 23);
			}
		}
		// ---- forward recompute (the colour head's output layer is not needed)
		float sh[4]; sh4_split(cur.d3, g, sh);
		float fs[8];
#pragma unroll
		for (int k = 0; k < 8; ++k) fs[k] = cur.f[k] * FEAT_PRESCALE;
		const B2 b0 = split8(fs);
		floatx4 c[4];                                                             // RAW accumulator values from here on (combine_raw; the scales are folded into split_relu_raw)
#pragma unroll
		for (int t = 0; t < 4; ++t) { Acc a = {z, z}; mma3<MS>(wl, t, lane, b0, a); c[t] = combine_raw(a); }
		const B2 h0 = split_relu_raw<-4>(c[0], c[1]), h1 = split_relu_raw<-4>(c[2], c[3]);
		const uint32_t mh0 = relu_mask8(c[0], c[1]), mh1 = relu_mask8(c[2], c[3]);
		floatx4 dr;
		{ Acc a = {z, z}; mma3<MS>(wl, 4, lane, h0, a); mma3<MS>(wl, 5, lane, h1, a); dr = combine_raw(a); }
		const floatx4 shs = (floatx4){sh[0], sh[1], sh[2], sh[3]} * HID_PRESCALE;
		const B2 b2 = split_x4(dr, shs, dr * SPLIT_SCALE, shs * SPLIT_SCALE);
#pragma unroll
		for (int t = 0; t < 4; ++t) { Acc a = {z, z}; mma3<MS>(wl, 6 + t, lane, b2, a); c[t] = combine_raw(a); }
		const B2 g00 = split_relu_raw<0>(c[0], c[1]), g01 = split_relu_raw<0>(c[2], c[3]);
		const uint32_t mg00 = relu_mask8(c[0], c[1]), mg01 = relu_mask8(c[2], c[3]);
#pragma unroll
		for (int t = 0; t < 4; ++t) { Acc a = {z, z}; mma3<MS>(wl, 10 + 2 * t, lane, g00, a); mma3<MS>(wl, 11 + 2 * t, lane, g01, a); c[t] = combine_raw(a); }
		const B2 g10 = split_relu_raw<0>(c[0], c[1]), g11 = split_relu_raw<0>(c[2], c[3]);
		const uint32_t mg10 = relu_mask8(c[0], c[1]), mg11 = relu_mask8(c[2], c[3]);
		// ---- dgrad chain on sigma-scaled gradients (register resident, transposed fragments), interleaved with the five weight-gradient phases in the order that
		// releases registers soonest: a phase runs as soon as its gradient exists, and its activations (only the ReLU masks feed the chain) die with it.
		// dW[o][i] += sum_s dY[s][o] X[s][i]   (A = dY^T rows o, B = X^T rows i, k = sample); a trip's tiles carry sigma and X's prescale.
		const int col = 16 * w + s, o = lane & 15;
		const float uH = inv_sigma * (1.0f / HID_PRESCALE), uF = inv_sigma * (1.0f / FEAT_PRESCALE);
		floatx4 dov = {0.f, 0.f, 0.f, 0.f};                                              // slots j < 4 <-> output neuron 4g + j; only neurons 0..2 (g == 0) are non-zero
		if (g == 0) { dov[0] = cur.go[0] * sigma; dov[1] = cur.go[1] * sigma; dov[2] = cur.go[2] * sigma; }
		const B2 dO = split_x4_low(dov, dov * SPLIT_SCALE);
		// phase C2: V2 = dO x G1 (W1 / V2: each tile's samples split between waves w and w + 4)
		if (PROBE != 1 && PROBE < 3) (TR ? st_rows16_T(stage, 0, col, g, dO) : st_rows16_2(stage, 0, col, g, dO));
		if (PROBE != 1 && PROBE < 3) (TR ? st_rows64_T(stage, 16, col, g, g10, g11) : st_rows64_2(stage, 16, col, g, g10, g11));
		if (PROBE < 4) __syncthreads();
		if (PROBE < 2) aV2 = fma4(aV2, WG3(stage, 0, 16 + 16 * tx, o, g, 64 * half, 64 * half + 64), uH);
#pragma unroll
		for (int t = 0; t < 4; ++t) { Acc a = {z, z}; mma3<MS>(wb, t, lane, dO, a); c[t] = combine_raw(a); }
		const B2 dG1lo = split_masked(c[0], c[1], mg10), dG1hi = split_masked(c[2], c[3], mg11);
		if (PROBE < 4) __syncthreads();
		// phase A: V1 = dG1 x G0
		if (PROBE != 1 && PROBE < 3) (TR ? st_rows64_T(stage, 0, col, g, dG1lo, dG1hi) : st_rows64_2(stage, 0, col, g, dG1lo, dG1hi));
		if (PROBE != 1 && PROBE < 3) (TR ? st_rows64_T(stage, 64, col, g, g00, g01) : st_rows64_2(stage, 64, col, g, g00, g01));
		if (PROBE < 4) __syncthreads();
		if (PROBE < 2) aV1[0] = fma4(aV1[0], WG3(stage, 16 * to, 64 + 16 * ti0, o, g, 0, SBT), uH);
		if (PROBE < 2) aV1[1] = fma4(aV1[1], WG3(stage, 16 * to, 64 + 16 * (ti0 + 1), o, g, 0, SBT), uH);
#pragma unroll
		for (int t = 0; t < 4; ++t) { Acc a = {z, z}; mma3<MS>(wb, 4 + 2 * t, lane, dG1lo, a); mma3<MS>(wb, 5 + 2 * t, lane, dG1hi, a); c[t] = combine_raw(a); }
		const B2 dG0lo = split_masked(c[0], c[1], mg00), dG0hi = split_masked(c[2], c[3], mg01);
		if (PROBE < 4) __syncthreads();
		// phase B2: V0 = dG0 x [density | SH]
		if (PROBE != 1 && PROBE < 3) (TR ? st_rows64_T(stage, 0, col, g, dG0lo, dG0hi) : st_rows64_2(stage, 0, col, g, dG0lo, dG0hi));
		if (PROBE != 1 && PROBE < 3) (TR ? st_rows32_T(stage, 64, col, g, b2, false) : st_rows32_2(stage, 64, col, g, b2, false));
		if (PROBE < 4) __syncthreads();
		if (PROBE < 2) aV0 = fma4(aV0, WG3(stage, 16 * to, 64 + 16 * tj, o, g, 0, SBT), uH);
		floatx4 dD;
		{ Acc a = {z, z}; mma3<MS>(wb, 12, lane, dG0lo, a); mma3<MS>(wb, 13, lane, dG0hi, a); dD = combine_raw(a); }
		if (g == 0) dD[0] += cur.go[3] * sigma;                                      // out[:,3] = den[:,0]  (ngp_network.py:83)
		const B2 dDf = split_x4_low(dD, dD * SPLIT_SCALE);
		if (PROBE < 4) __syncthreads();
		// phase C1: W1 = dD x H
		if (PROBE != 1 && PROBE < 3) (TR ? st_rows16_T(stage, 0, col, g, dDf) : st_rows16_2(stage, 0, col, g, dDf));
		if (PROBE != 1 && PROBE < 3) (TR ? st_rows64_T(stage, 16, col, g, h0, h1) : st_rows64_2(stage, 16, col, g, h0, h1));
		if (PROBE < 4) __syncthreads();
		if (PROBE < 2) aW1 = fma4(aW1, WG3(stage, 0, 16 + 16 * tx, o, g, 64 * half, 64 * half + 64), uH);
#pragma unroll
		for (int t = 0; t < 4; ++t) { Acc a = {z, z}; mma3<MS>(wb, 14 + t, lane, dDf, a); c[t] = combine_raw(a); }
		const B2 dHlo = split_masked(c[0], c[1], mh0), dHhi = split_masked(c[2], c[3], mh1);
		if (PROBE < 4) __syncthreads();
		// phase B1: W0 = dH x features
		if (PROBE != 1 && PROBE < 3) (TR ? st_rows64_T(stage, 0, col, g, dHlo, dHhi) : st_rows64_2(stage, 0, col, g, dHlo, dHhi));
		if (PROBE != 1 && PROBE < 3) (TR ? st_rows32_T(stage, 64, col, g, b0, true) : st_rows32_2(stage, 64, col, g, b0, true));
		if (PROBE < 4) __syncthreads();
		if (PROBE < 2) aW0 = fma4(aW0, WG3(stage, 16 * to, 64 + 16 * tj, o, g, 0, SBT), uF);
		floatx4 dF[2];
#pragma unroll
		for (int t = 0; t < 2; ++t) { Acc a = {z, z}; mma3<MS>(wb, 18 + 2 * t, lane, dHlo, a); mma3<MS>(wb, 19 + 2 * t, lane, dHhi, a); dF[t] = combine(a, inv_sigma); }
		if (valid) {                                                                 // feature 16t + 4g + r  ->  level 8t + 2g + (r >> 1), component r & 1
#pragma unroll
			for (int t = 0; t < 2; ++t)
#pragma unroll
				for (int pr = 0; pr < 2; ++pr) {
					const float2 v = make_float2(dF[t][2 * pr], dF[t][2 * pr + 1]);
					const uint32_t level = 8 * t + 2 * g + pr;
					lmax[t][pr] = fmaxf(lmax[t][pr], fmaxf(fabsf(v.x), fabsf(v.y)));
					bad |= !(fabsf(v.x) <= 3.0e38f) || !(fabsf(v.y) <= 3.0e38f);
					if (LAYOUT == NGP_LAYOUT_SOA) reinterpret_cast<float2 *>(dfeat)[(size_t)level * n + i] = v;
					else *reinterpret_cast<float2 *>(dfeat + (size_t)i * 32 + 2 * level) = v;
				}
		}
		if (PROBE < 4) __syncthreads();
		if (more) cur = nxt;
	}
	// ---- W1 / V2 partial sums of waves 4..7 join those of waves 0..3 through LDS, then one fp32 slab per workgroup, packed like the weights
	float *xch = reinterpret_cast<float *>(stage);            // [4 tiles][2][64 lanes][4]
	if (half == 1) {
#pragma unroll
		for (int r = 0; r < 4; ++r) { xch[((tx * 2 + 0) * 64 + lane) * 4 + r] = aW1[r]; xch[((tx * 2 + 1) * 64 + lane) * 4 + r] = aV2[r]; }
	}
	__syncthreads();
	float *slab = slabs + (size_t)blockIdx.x * 10240;
	const int ci = lane & 15;
#pragma unroll
	for (int r = 0; r < 4; ++r) {
		const int ro = 4 * g + r;
#pragma unroll
		for (int q = 0; q < 2; ++q) slab[3072 + 2048 + (16 * to + ro) * 64 + 16 * (ti0 + q) + ci] = aV1[q][r];
		slab[(16 * to + ro) * 32 + 16 * tj + ci] = aW0[r];
		slab[3072 + (16 * to + ro) * 32 + 16 * tj + ci] = aV0[r];
		if (half == 0) {
			slab[2048 + ro * 64 + 16 * tx + ci] = aW1[r] + xch[((tx * 2 + 0) * 64 + lane) * 4 + r];
			slab[3072 + 6144 + ro * 64 + 16 * tx + ci] = aV2[r] + xch[((tx * 2 + 1) * 64 + lane) * 4 + r];
		}
	}
	if (bad) atomicOr(&g_split_range_flag, 2u);
	if (am.parts) { __syncthreads(); absmax_epilogue(am, lmax, reinterpret_cast<float *>(stage), 8); }
}

#undef WG3
// bit 0: an operand of a split forward came within 4x of fp16's largest finite value since the last reset; bit 1: one left the range (infinities in that launch's
// results) or a split backward stored a non-finite feature gradient.  A 4-byte read-back through the null stream; it does not wait for other streams' kernels - a launch
// that has not reported yet reports at the next check.  reset clears EXACTLY the bits that were read, atomically on the device (r5, ADVICE r4: a read followed by a
// plain store of zero could wipe a bit a kernel on another stream raised in between - the one way the "an error, never a silent one" guarantee could be lost).
__global__ void k_range_flag_clear(uint32_t bits) { atomicAnd(&g_split_range_flag, ~bits); }
NGP_API int ngp_field32_range_check(int reset) {
	uint32_t v = 0u;
	hipError_t e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_split_range_flag), sizeof(v), 0, hipMemcpyDeviceToHost);
	if (e != hipSuccess) { ngp_set_error("ngp_field32_range_check: %s", hipGetErrorString(e)); return NGP_E_ARG; }
	if (reset && v) {
		NGP_LAUNCH(k_range_flag_clear, dim3(1), dim3(1), 0, (hipStream_t)0, v);
		NGP_LAUNCH_CHECK("ngp_field32_range_check");
	}
	return (int)(v & 3u);
}

static uint32_t split_grid(uint32_t n) { uint32_t b = div_up(div_up(n, 16), 4); return b < 1024 ? (b ? b : 1) : 1024; }
// launched by ngp_field32_fwd / ngp_density32_fwd (field32.hip) with the split fragments that follow the fp32 fragments in the packed weight buffer
int ngp_field32_fwd_split(void *stream, uint32_t n, const float *feat, int layout, const float *dir, uint32_t dir_stride, const void *split_frags, float *out,
                          const uint32_t *n_valid, int density_only) {
	hipStream_t s = (hipStream_t)stream;
	const dim3 grid(split_grid(n)), block(256);
	const _Float16 *p = (const _Float16 *)split_frags;
	if (density_only) {
		if (layout == NGP_LAYOUT_SOA) NGP_LAUNCH((k_field32_fwd_split<NGP_LAYOUT_SOA, true>), grid, block, 0, s, n, feat, (const float *)nullptr, 3u, p, out, n_valid);
		else NGP_LAUNCH((k_field32_fwd_split<NGP_LAYOUT_AOS, true>), grid, block, 0, s, n, feat, (const float *)nullptr, 3u, p, out, n_valid);
	} else {
		if (layout == NGP_LAYOUT_SOA) NGP_LAUNCH((k_field32_fwd_split<NGP_LAYOUT_SOA, false>), grid, block, 0, s, n, feat, dir, dir_stride, p, out, n_valid);
		else NGP_LAUNCH((k_field32_fwd_split<NGP_LAYOUT_AOS, false>), grid, block, 0, s, n, feat, dir, dir_stride, p, out, n_valid);
	}
	NGP_LAUNCH_CHECK("ngp_field32_fwd_split");
	return 0;
}

size_t ngp_field32_bwd_split_shmem() { return (size_t)(NSPLIT_HALVES + 2 * SPLANE) * sizeof(_Float16); }
// launched by ngp_field32_bwd_am (field32.hip) when NGP_FIELD32_BWD selects the split variant
int ngp_field32_bwd_split(void *stream, uint32_t n, const float *feat, int layout, const float *dir, uint32_t dir_stride, const void *split_frags, const float *dout,
                          float *dfeat, float *slabs, uint32_t n_slabs, const uint32_t *n_valid, const AbsmaxOut *am_in) {
	const AbsmaxOut am = am_in ? *am_in : AbsmaxOut{nullptr, nullptr, 0u, nullptr};
	hipStream_t s = (hipStream_t)stream;
	const size_t shmem = ngp_field32_bwd_split_shmem();
	const dim3 grid(n_slabs), block(512);
	const _Float16 *p = (const _Float16 *)split_frags;
	static const bool tr = [] { const char *e = getenv("NGP_SPLIT_TRSTAGE"); return !(e && e[0] == '0'); }();      // A/B hook: NGP_SPLIT_TRSTAGE=0 = rounds 3-5's [neuron][sample] staging image
#define GOPT(L, P, T) do { \
	static bool attr_set = false; \
	if (!attr_set) { hipError_t e = hipFuncSetAttribute((const void *)k_field32_bwd_split<L, P, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
		if (e != hipSuccess) { ngp_set_error("ngp_field32_bwd(split): hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; } attr_set = true; } \
	NGP_LAUNCH((k_field32_bwd_split<L, P, T>), grid, block, shmem, s, n, feat, dir, dir_stride, p, dout, dfeat, slabs, n_valid, am); } while (0)
#define GOP(L, P) do { if (tr) GOPT(L, P, true); else GOPT(L, P, false); } while (0)
	// (r6) the timing probes (PROBE != 0: parts of the kernel compiled out, RESULTS WRONG) are a compile-time build - EXTRA=-DNGP_PROBE_SPLIT=1..4 bash csrc/build.sh,
	// tools/probe_split_bwd.py - no longer an environment variable a user could set on the product binary
#ifdef NGP_PROBE_SPLIT
#define GO(L) GOP(L, NGP_PROBE_SPLIT)
#else
#define GO(L) GOP(L, 0)
#endif
	if (layout == NGP_LAYOUT_SOA) GO(NGP_LAYOUT_SOA); else GO(NGP_LAYOUT_AOS);
#undef GO
#undef GOP
#undef GOPT
	NGP_LAUNCH_CHECK("ngp_field32_bwd(split)");
	return 0;
}
