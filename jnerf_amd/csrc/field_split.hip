// Forward of the fp32 field network (NGPNetworks.execute_ / .density without cfg.fp16: models/networks/ngp_network.py:57-67, 77-89 - what ngp_base.py runs) on the
// fp16 matrix cores at fp32 accuracy: split operands, three MFMAs per product sum (field_split.h).  Same "transposed" register-resident formulation as
// field_mlp.hip (weights = A operand from pre-permuted LDS fragments, 16 samples of a wave tile = B columns, a layer's C fragment is the next layer's B fragment),
// fp32 features in, fp32 outputs out; between layers the fp32 accumulator is ReLU'd and split again in registers.
// Replaces k_field32_fwd (v_mfma_f32_16x16x4_f32, 52 us per 2^18-sample batch at 0.58 of the fp32 MFMA peak) in ngp_field32_fwd / ngp_density32_fwd;
// NGP_FIELD32_FWD=mfma32 selects the exact-product kernel again.  The backward kernel (field32.hip) is unchanged and recomputes its forward with fp32 MFMAs: the
// two forwards agree to ~3e-7 of the output scale (tests/test_hip_parity.py::test_field32_split_forward...).
#include "ngp_common.h"
#include "field_split.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
// fp16 spans 6e-5 .. 65504 (normal range): B operands are multiplied by a power of two before the split and the factor is taken out of the fp32 accumulator again (both
// exact).  Hash features start at ~1e-4 and stay below ~10: x 256 (safe up to 255); hidden activations / density logits / SH: x 16 (safe up to 4094; accurate down to a
// tensor scale of ~4e-6).  Values beyond the safe maxima overflow to infinity - NGP_FIELD32_FWD=mfma32 is the kernel without such a range.
#define FEAT_PRESCALE 256.0f
#define HID_PRESCALE 16.0f

__global__ __launch_bounds__(256) void k_pack_split(const float *__restrict__ wd, const float *__restrict__ wc, _Float16 *__restrict__ out, int n_frags) {
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx < NSPLIT_HALVES && ((idx % (NSPLIT_FRAGS * 512)) >> 9) < n_frags) out[idx] = split_frag_half(wd, wc, idx);
}
// n_frags = 6: the density network only (wc is not read), NSPLIT_FRAGS: everything
int ngp_field32_pack_split(void *stream, const float *wd, const float *wc, void *out_halves, int n_frags) {
	NGP_LAUNCH(k_pack_split, dim3(div_up(NSPLIT_HALVES, 256)), dim3(256), 0, (hipStream_t)stream, wd, wc, (_Float16 *)out_halves, n_frags);
	NGP_LAUNCH_CHECK("ngp_field32_pack_split");
	return 0;
}

struct B2 { half8 h, m; };
__device__ __forceinline__ B2 split8(const float v[8]) {
	B2 r;
#pragma unroll
	for (int k = 0; k < 8; ++k) { const _Float16 h = (_Float16)v[k]; r.h[k] = h; r.m[k] = (_Float16)((v[k] - (float)h) * SPLIT_SCALE); }
	return r;
}
__device__ __forceinline__ B2 split_relu(floatx4 a, floatx4 b) {
	float v[8];
#pragma unroll
	for (int k = 0; k < 4; ++k) { v[k] = fmaxf(a[k], 0.f) * HID_PRESCALE; v[4 + k] = fmaxf(b[k], 0.f) * HID_PRESCALE; }
	return split8(v);
}
struct Acc { floatx4 main, corr; };
__device__ __forceinline__ half8 ld_half8(const _Float16 *lds, int f, int lane) { return *reinterpret_cast<const half8 *>(lds + f * 512 + lane * 8); }
// acc += W-fragment f x B: the leading product and the two cross terms (kept in an accumulator of their own: they are 2^-11 smaller)
__device__ __forceinline__ void mma3(const _Float16 *wl, int f, int lane, const B2 &b, Acc &acc) {
	const half8 ah = ld_half8(wl, f, lane), am = ld_half8(wl + NSPLIT_FRAGS * 512, f, lane);
	acc.main = MFMA16(ah, b.h, acc.main);
	acc.corr = MFMA16(ah, b.m, acc.corr);
	acc.corr = MFMA16(am, b.h, acc.corr);
}
__device__ __forceinline__ floatx4 combine(const Acc &a, float scale) {
	floatx4 r;
#pragma unroll
	for (int k = 0; k < 4; ++k) r[k] = (a.main[k] + a.corr[k] * (1.0f / SPLIT_SCALE)) * scale;
	return r;
}

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

template <bool DENSITY_ONLY>
__device__ __forceinline__ void forward_split(const _Float16 *wl, int lane, const float feat[8], const float sh[4], floatx4 &den, floatx4 &rgb) {
	const floatx4 z = {0.f, 0.f, 0.f, 0.f};
	float fs[8];
#pragma unroll
	for (int k = 0; k < 8; ++k) fs[k] = feat[k] * FEAT_PRESCALE;
	const B2 b0 = split8(fs);
	floatx4 c0[4];
#pragma unroll
	for (int t = 0; t < 4; ++t) { Acc a = {z, z}; mma3(wl, t, lane, b0, a); c0[t] = combine(a, 1.0f / FEAT_PRESCALE); }        // L0: 32 -> 64
	const B2 h0 = split_relu(c0[0], c0[1]), h1 = split_relu(c0[2], c0[3]);
	{ Acc a = {z, z}; mma3(wl, 4, lane, h0, a); mma3(wl, 5, lane, h1, a); den = combine(a, 1.0f / HID_PRESCALE); }               // L1: 64 -> 16
	if (DENSITY_ONLY) return;
	float in2[8];
#pragma unroll
	for (int k = 0; k < 4; ++k) { in2[k] = den[k] * HID_PRESCALE; in2[4 + k] = sh[k] * HID_PRESCALE; }
	const B2 b2 = split8(in2);
	floatx4 c2[4];
#pragma unroll
	for (int t = 0; t < 4; ++t) { Acc a = {z, z}; mma3(wl, 6 + t, lane, b2, a); c2[t] = combine(a, 1.0f / HID_PRESCALE); }                       // L2: [density(16) | SH(16)] -> 64
	const B2 g00 = split_relu(c2[0], c2[1]), g01 = split_relu(c2[2], c2[3]);
	floatx4 c3[4];
#pragma unroll
	for (int t = 0; t < 4; ++t) { Acc a = {z, z}; mma3(wl, 10 + 2 * t, lane, g00, a); mma3(wl, 11 + 2 * t, lane, g01, a); c3[t] = combine(a, 1.0f / HID_PRESCALE); }   // L3: 64 -> 64
	const B2 g10 = split_relu(c3[0], c3[1]), g11 = split_relu(c3[2], c3[3]);
	{ Acc a = {z, z}; mma3(wl, 18, lane, g10, a); mma3(wl, 19, lane, g11, a); rgb = combine(a, 1.0f / HID_PRESCALE); }                          // L4: 64 -> 16 (3 used)
}

template <int LAYOUT, bool DENSITY_ONLY>
__global__ __launch_bounds__(256) void k_field32_fwd_split(uint32_t n, const float *__restrict__ feat, const float *__restrict__ dir, uint32_t dir_stride,
                                                           const _Float16 *__restrict__ packed, float *__restrict__ out, const uint32_t *__restrict__ n_valid) {
	__shared__ __attribute__((aligned(16))) _Float16 wl[NSPLIT_HALVES];
	{	// both parts of the fragments this variant reads (density only: layers 0 and 1)
		const int nf = DENSITY_ONLY ? 6 : NSPLIT_FRAGS;
		const uint4 *src = reinterpret_cast<const uint4 *>(packed);
		uint4 *dst = reinterpret_cast<uint4 *>(wl);
		for (int idx = threadIdx.x; idx < nf * 64; idx += 256) { dst[idx] = src[idx]; dst[NSPLIT_FRAGS * 64 + idx] = src[NSPLIT_FRAGS * 64 + idx]; }
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
	if (wave < n_tiles) fetch(wave, f, d);
	for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
		const uint32_t i = tile * 16u + s;
		const bool more = tile + n_waves < n_tiles;
		if (more) fetch(tile + n_waves, fn, dn);                   // the next tile's inputs are in flight during this tile's chain
		float sh[4] = {0.f, 0.f, 0.f, 0.f};
		if (!DENSITY_ONLY) sh4_split(d, g, sh);
		floatx4 den, rgb;
		forward_split<DENSITY_ONLY>(wl, lane, f, sh, den, rgb);
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
