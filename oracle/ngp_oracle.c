/*
 * TEST INFRASTRUCTURE ONLY — CPU oracle for the Instant-NGP hot path of Jittor/JNeRF.
 *
 * A plain-C restatement (scalar, single thread, fp32 arithmetic in the reference's
 * evaluation order, no FMA contraction) of every kernel on the path named by
 * BASELINE.json `north_star`.  Each function cites the reference file:line it follows
 * (paths relative to /root/reference/python/jnerf/).  It is pinned against the
 * reference's own kernel source compiled for the host (oracle/_ref, built by
 * oracle/ref_shim/Makefile) and against the committed fixtures in tests/golden/ —
 * see tests/test_oracle_vs_ref.py and tests/test_oracle_golden.py.  End to end it is
 * pinned by tests/test_refrun_golden.py: 18 iterations of the reference's unmodified Python
 * package (Runner.train over a Jittor stand-in, every CUDA launch bound to oracle/_ref;
 * tests/golden/make_golden_refrun.py) are replayed through this file alone - losses to 1e-6
 * up to the second occupancy refresh, identical sample counts and generator state.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (jnerf_amd/) never does.
 *
 * Parts with NO compilable source in the reference ("parity unpinned"):
 *   - the fully-fused MLP (binary-only tiny-cuda-nn object) -> restated from the
 *     fallback nn.Linear chain models/networks/ngp_network.py:59-67 (that chain itself IS
 *     executed against this file, forward and backward: tests/test_pyref_golden.py; what
 *     stays unpinned is the fp16 rounding of the binary kernels);
 *   - Adam (inside Jittor, external)                         -> standard bias-corrected Adam.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------ fp16 helpers */
/* IEEE binary16 <-> binary32, round-to-nearest-even (what __half(float) does). */
static uint16_t f2h(float f) {
	uint32_t x; memcpy(&x, &f, 4);
	uint32_t sign = (x >> 16) & 0x8000u;
	uint32_t mant = x & 0x007fffffu;
	int32_t exp = (int32_t)((x >> 23) & 0xff);
	if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u | (mant >> 13) : 0));
	exp = exp - 127 + 15;
	if (exp >= 0x1f) return (uint16_t)(sign | 0x7c00u);
	if (exp <= 0) {
		if (exp < -10) return (uint16_t)sign;
		mant |= 0x00800000u;
		uint32_t shift = (uint32_t)(14 - exp);
		uint32_t h = mant >> shift;
		uint32_t rem = mant & ((1u << shift) - 1u);
		uint32_t half = 1u << (shift - 1);
		if (rem > half || (rem == half && (h & 1u))) h++;
		return (uint16_t)(sign | h);
	}
	uint32_t h = ((uint32_t)exp << 10) | (mant >> 13);
	uint32_t rem = mant & 0x1fffu;
	if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
	return (uint16_t)(sign | h);
}
static float h2f(uint16_t h) {
	uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
	uint32_t exp = (h >> 10) & 0x1f;
	uint32_t mant = h & 0x3ffu;
	uint32_t x;
	if (exp == 0) {
		if (mant == 0) x = sign;
		else {
			int e = -1;
			do { mant <<= 1; e++; } while (!(mant & 0x400u));
			x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((mant & 0x3ffu) << 13);
		}
	} else if (exp == 0x1f) x = sign | 0x7f800000u | (mant << 13);
	else x = sign | ((exp - 15 + 127) << 23) | (mant << 13);
	float f; memcpy(&f, &x, 4); return f;
}
EXPORT uint16_t orc_f2h(float f) { return f2h(f); }
EXPORT float orc_h2f(uint16_t h) { return h2f(h); }

/* element access for T in {f32 (is_half=0), f16 (is_half=1)} */
static inline float ldT(const void *p, size_t i, int is_half) { return is_half ? h2f(((const uint16_t *)p)[i]) : ((const float *)p)[i]; }
static inline void stT(void *p, size_t i, float v, int is_half) { if (is_half) ((uint16_t *)p)[i] = f2h(v); else ((float *)p)[i] = v; }

/* ------------------------------------------------------------------ pcg32 */
/* ops/op_include/pcg32/pcg32.h:39-166 */
#define PCG32_MULT 0x5851f42d4c957f2dULL
typedef struct { uint64_t state, inc; } pcg32_t;
static uint32_t pcg_next_uint(pcg32_t *r) {                       /* pcg32.h:62-68 */
	uint64_t old = r->state;
	r->state = old * PCG32_MULT + r->inc;
	uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
	uint32_t rot = (uint32_t)(old >> 59u);
	return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
static void pcg_seed(pcg32_t *r, uint64_t initstate, uint64_t initseq) { /* pcg32.h:53-59 */
	r->state = 0; r->inc = (initseq << 1u) | 1u;
	pcg_next_uint(r); r->state += initstate; pcg_next_uint(r);
}
static float pcg_next_float(pcg32_t *r) {                          /* pcg32.h:103-112 */
	union { uint32_t u; float f; } x;
	x.u = (pcg_next_uint(r) >> 9) | 0x3f800000u;
	return x.f - 1.0f;
}
static void pcg_advance(pcg32_t *r, int64_t delta_) {              /* pcg32.h:145-166 */
	uint64_t cur_mult = PCG32_MULT, cur_plus = r->inc, acc_mult = 1u, acc_plus = 0u;
	uint64_t delta = (uint64_t)delta_;
	while (delta > 0) {
		if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
		cur_plus = (cur_mult + 1) * cur_plus;
		cur_mult *= cur_mult;
		delta /= 2;
	}
	r->state = acc_mult * r->state + acc_plus;
}
EXPORT void orc_pcg32_seed(uint64_t initstate, uint64_t initseq, uint64_t *st) { pcg32_t r; pcg_seed(&r, initstate, initseq); st[0] = r.state; st[1] = r.inc; }
EXPORT uint32_t orc_pcg32_next_uint(uint64_t *st) { pcg32_t r = {st[0], st[1]}; uint32_t v = pcg_next_uint(&r); st[0] = r.state; return v; }
EXPORT float orc_pcg32_next_float(uint64_t *st) { pcg32_t r = {st[0], st[1]}; float v = pcg_next_float(&r); st[0] = r.state; return v; }
EXPORT void orc_pcg32_advance(uint64_t *st, int64_t delta) { pcg32_t r = {st[0], st[1]}; pcg_advance(&r, delta); st[0] = r.state; }

/* ------------------------------------------------------------------ hash-grid level table */
/* Host half: position_encoders/hash_encoder/grid_encode.py:17-40 (fp64 python).
 * Device half: the per-level scale/resolution of HashEncode.h:149-151 (fp32: exp2f(level*log2f(s))*16-1),
 * evaluated here as the correctly-rounded fp32 value so that host python, this oracle and the HIP
 * kernels all consume ONE table (SURVEY.md §7 "fp32 exp2f level-scale hazard").
 * table layout per level (4 x u32): offset(entries), size(entries), resolution, scale (f32 bits). */
EXPORT uint32_t orc_level_table(double aabb_scale, uint32_t *table /*[16*4]*/, uint32_t *offsets /*[17]*/) {
	const double per_level_scale = exp(log(2048.0 * aabb_scale / 16.0) / 15.0);
	const float log2s = (float)log2(per_level_scale);
	uint32_t offset = 0;
	for (uint32_t l = 0; l < 16; ++l) {
		double scale_h = pow(2.0, (double)l * log2(per_level_scale)) * 16.0 - 1.0;        /* grid_encode.py:27 */
		uint32_t res_h = (uint32_t)ceil(scale_h) + 1;                                      /* :28 */
		uint64_t p = (uint64_t)res_h * res_h * res_h;                                      /* :29 */
		p = (p + 7) / 8 * 8; if (p > (1u << 19)) p = 1u << 19;                             /* :30-32 */
		float arg = (float)l * log2s;
		float scale_d = (float)exp2((double)arg) * 16.0f - 1.0f;                          /* HashEncode.h:149 */
		uint32_t res_d = (uint32_t)ceilf(scale_d) + 1;                                     /* :151 */
		offsets[l] = offset;
		table[4 * l + 0] = offset; table[4 * l + 1] = (uint32_t)p; table[4 * l + 2] = res_d; memcpy(&table[4 * l + 3], &scale_d, 4);
		offset += (uint32_t)p;
	}
	offsets[16] = offset;
	return offset * 2; /* m_n_params, grid_encode.py:36 */
}

/* HashEncode.h:68-94 with get_index = ngp_base.py:69 */
static inline uint32_t grid_index(uint32_t hashmap_size, uint32_t res, const uint32_t g[3]) {
	uint32_t stride = 1, index = 0;
	for (uint32_t dim = 0; dim < 3 && stride <= hashmap_size; ++dim) { index += g[dim] * stride; stride *= res; }
	if (hashmap_size < stride) index = g[0] ^ g[1] * 19349663u ^ g[2] * 83492791u;
	return (index % hashmap_size) * 2;
}

/* HashEncode.h:106-115 (identity interpolation) */
static inline void pos_fract(float in, float scale, float *w, uint32_t *g) {
	float p = in * scale + 0.5f;
	int tmp = (int)floorf(p);
	*g = (uint32_t)tmp;
	*w = p - (float)tmp;
}

/* Forward: HashEncode.h:117-203 (+extract_position :36-50, transpose :254-268). Output [n,32] level-major, accumulated IN T (:199). */
EXPORT void orc_hash_encode_fwd(uint32_t n, const float *x /*[n,3]*/, const void *grid, const uint32_t *table, void *out /*[n,32]*/, int is_half) {
	for (uint32_t i = 0; i < n; ++i) for (uint32_t l = 0; l < 16; ++l) {
		const uint32_t off = table[4 * l], size = table[4 * l + 1], res = table[4 * l + 2];
		float scale; memcpy(&scale, &table[4 * l + 3], 4);
		float w[3]; uint32_t g[3];
		for (int d = 0; d < 3; ++d) pos_fract(x[3 * i + d], scale, &w[d], &g[d]);
		float r0 = 0.f, r1 = 0.f; /* holds a T-representable value */
		for (uint32_t c = 0; c < 8; ++c) {
			float weight = 1; uint32_t gl[3];
			for (int d = 0; d < 3; ++d) {
				if ((c & (1u << d)) == 0) { weight *= 1 - w[d]; gl[d] = g[d]; }
				else { weight *= w[d]; gl[d] = g[d] + 1; }
			}
			size_t idx = (size_t)off * 2 + grid_index(size, res, gl);
			float t0 = weight * ldT(grid, idx, is_half), t1 = weight * ldT(grid, idx + 1, is_half);
			if (is_half) { r0 = h2f(f2h(r0 + h2f(f2h(t0)))); r1 = h2f(f2h(r1 + h2f(f2h(t1)))); }
			else { r0 += t0; r1 += t1; }
		}
		stT(out, (size_t)i * 32 + 2 * l, r0, is_half); stT(out, (size_t)i * 32 + 2 * l + 1, r1, is_half);
	}
}

/* Forward with d(encoding)/d(position): the `dy_dx` branch of kernel_grid, HashEncode.h:205-251 (compiled in the reference but never enabled by grid_encode.py:96 -
 * the prerequisite of a hash-grid NeuS).  dydx [n][3][32] fp32: dydx[i][d][2l+f] = d out[i][2l+f] / d x[i][d]; per derivative dimension the
 * four (left, right) corner pairs in the reference's idx order, weight = scale * w(non-grad dim 0) * w(non-grad dim 1), pos_derivative = 1 (identity interpolation). */
EXPORT void orc_hash_encode_fwd_dydx(uint32_t n, const float *x /*[n,3]*/, const void *grid, const uint32_t *table, void *out /*[n,32]*/, float *dydx /*[n,3,32]*/, int is_half) {
	orc_hash_encode_fwd(n, x, grid, table, out, is_half);
	for (uint32_t i = 0; i < n; ++i) for (uint32_t l = 0; l < 16; ++l) {
		const uint32_t off = table[4 * l], size = table[4 * l + 1], res = table[4 * l + 2];
		float scale; memcpy(&scale, &table[4 * l + 3], 4);
		float w[3]; uint32_t g[3];
		for (int d = 0; d < 3; ++d) pos_fract(x[3 * i + d], scale, &w[d], &g[d]);
		for (uint32_t gd = 0; gd < 3; ++gd) {
			float a0 = 0.f, a1 = 0.f;
			for (uint32_t idx = 0; idx < 4; ++idx) {
				float weight = scale; uint32_t gl[3];
				for (uint32_t nd = 0; nd < 2; ++nd) {
					const uint32_t dim = nd >= gd ? nd + 1 : nd;
					if ((idx & (1u << nd)) == 0) { weight *= 1 - w[dim]; gl[dim] = g[dim]; }
					else { weight *= w[dim]; gl[dim] = g[dim] + 1; }
				}
				gl[gd] = g[gd];
				const size_t il = (size_t)off * 2 + grid_index(size, res, gl);
				gl[gd] = g[gd] + 1;
				const size_t ir = (size_t)off * 2 + grid_index(size, res, gl);
				a0 += weight * (ldT(grid, ir, is_half) - ldT(grid, il, is_half)) * 1.0f;
				a1 += weight * (ldT(grid, ir + 1, is_half) - ldT(grid, il + 1, is_half)) * 1.0f;
			}
			dydx[(size_t)i * 96 + gd * 32 + 2 * l] = a0; dydx[(size_t)i * 96 + gd * 32 + 2 * l + 1] = a1;
		}
	}
}
/* dL/dx[i][d] = sum_k dL/dy[i][k] * dydx[i][d][k], fp32, k ascending.  The reference has NO kernel for this (GridEncode.grad returns None for the positions,
 * grid_encode.py:190): this is the contraction its autograd would need - parity unpinned, restated from the chain rule. */
EXPORT void orc_hash_encode_bwd_input(uint32_t n, const void *dy /*[n,32]*/, const float *dydx, float *dLdx /*[n,3]*/, int is_half) {
	for (uint32_t i = 0; i < n; ++i) for (uint32_t d = 0; d < 3; ++d) {
		float a = 0.f;
		for (uint32_t k = 0; k < 32; ++k) a += ldT(dy, (size_t)i * 32 + k, is_half) * dydx[(size_t)i * 96 + d * 32 + k];
		dLdx[3 * i + d] = a;
	}
}

/* Backward: HashEncode.h:299-396 (+memset grid_encode.py:153, transpose_gradients :270-284).
 * Serial accumulation in thread order (level-major, sample-minor) like the serial launcher; T adds (:345-356). */
EXPORT void orc_hash_encode_bwd(uint32_t n, const float *x, const void *dy /*[n,32]*/, const uint32_t *table, void *grad, uint64_t n_params, int is_half) {
	memset(grad, 0, n_params * (is_half ? 2 : 4));
	for (uint32_t l = 0; l < 16; ++l) {
		const uint32_t off = table[4 * l], size = table[4 * l + 1], res = table[4 * l + 2];
		float scale; memcpy(&scale, &table[4 * l + 3], 4);
		for (uint32_t i = 0; i < n; ++i) {
			float w[3]; uint32_t g[3];
			for (int d = 0; d < 3; ++d) pos_fract(x[3 * i + d], scale, &w[d], &g[d]);
			float g0 = ldT(dy, (size_t)i * 32 + 2 * l, is_half), g1 = ldT(dy, (size_t)i * 32 + 2 * l + 1, is_half);
			for (uint32_t c = 0; c < 8; ++c) {
				float weight = 1; uint32_t gl[3];
				for (int d = 0; d < 3; ++d) {
					if ((c & (1u << d)) == 0) { weight *= 1 - w[d]; gl[d] = g[d]; }
					else { weight *= w[d]; gl[d] = g[d] + 1; }
				}
				size_t idx = (size_t)off * 2 + grid_index(size, res, gl);
				if (is_half) {
					uint16_t *gp = (uint16_t *)grad;
					gp[idx] = f2h(h2f(gp[idx]) + h2f(f2h(g0 * weight)));
					gp[idx + 1] = f2h(h2f(gp[idx + 1]) + h2f(f2h(g1 * weight)));
				} else {
					float *gp = (float *)grad; gp[idx] += g0 * weight; gp[idx + 1] += g1 * weight;
				}
			}
		}
	}
}

/* ------------------------------------------------------------------ spherical harmonics */
/* position_encoders/sh_encoder/op_header/SphericalEncode.h:45-95, degree 4 (sh_encoder.py:15-16) */
EXPORT void orc_sh_encode(uint32_t n, const float *din /*[n,3] in [0,1]*/, void *out /*[n,16]*/, int is_half) {
	for (uint32_t i = 0; i < n; ++i) {
		float x = din[3 * i] * 2.f - 1.f, y = din[3 * i + 1] * 2.f - 1.f, z = din[3 * i + 2] * 2.f - 1.f;
		float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
		float o[16];
		o[0] = 0.28209479177387814f;
		o[1] = -0.48860251190291987f * y;
		o[2] = 0.48860251190291987f * z;
		o[3] = -0.48860251190291987f * x;
		o[4] = 1.0925484305920792f * xy;
		o[5] = -1.0925484305920792f * yz;
		o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
		o[7] = -1.0925484305920792f * xz;
		o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
		o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
		o[10] = 2.8906114426405538f * xy * z;
		o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
		o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
		o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
		o[14] = 1.4453057213202769f * z * (x2 - y2);
		o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
		for (int k = 0; k < 16; ++k) stT(out, (size_t)i * 16 + k, o[k], is_half);
	}
}

/* ------------------------------------------------------------------ field MLPs (fp32 restatement) */
/* models/networks/ngp_network.py:59-67 + :77-84.  Weight pack = FMLP/FullyFusedMlp_weight layout
 * (ngp_network.py:21-29, fully_fused_mlp.py:26-41): every layer stored (out,in) row-major, last layer
 * zero-padded to 16 rows, all concatenated:  density: W0[64x32] W1[16x64];  rgb: V0[64x32] V1[64x64] V2[16x64].
 * acts (optional, may be NULL) saves post-ReLU hidden activations: h[n,64], g0[n,64], g1[n,64], plus den[n,16]. */
static void matvec(const float *W, int out, int in, const float *x, float *y, int relu) {
	for (int o = 0; o < out; ++o) {
		float s = 0.f;
		for (int k = 0; k < in; ++k) s += W[o * in + k] * x[k];
		y[o] = (relu && s < 0.f) ? 0.f : s;
	}
}
EXPORT void orc_field_fwd(uint32_t n, const float *feat /*[n,32]*/, const float *sh /*[n,16]*/, const float *wd /*[3072]*/, const float *wc /*[7168]*/,
                          float *out /*[n,4]*/, float *h /*[n,64]*/, float *den /*[n,16]*/, float *g0 /*[n,64]*/, float *g1 /*[n,64]*/) {
	for (uint32_t i = 0; i < n; ++i) {
		float hh[64], dd[16], in2[32], a0[64], a1[64], rgb[16];
		matvec(wd, 64, 32, feat + (size_t)i * 32, hh, 1);
		matvec(wd + 2048, 16, 64, hh, dd, 0);
		for (int k = 0; k < 16; ++k) { in2[k] = dd[k]; in2[16 + k] = sh[(size_t)i * 16 + k]; }    /* ngp_network.py:81 */
		matvec(wc, 64, 32, in2, a0, 1);
		matvec(wc + 2048, 64, 64, a0, a1, 1);
		matvec(wc + 2048 + 4096, 16, 64, a1, rgb, 0);
		out[4 * i] = rgb[0]; out[4 * i + 1] = rgb[1]; out[4 * i + 2] = rgb[2]; out[4 * i + 3] = dd[0];   /* :83 */
		if (h) memcpy(h + (size_t)i * 64, hh, 256);
		if (den) memcpy(den + (size_t)i * 16, dd, 64);
		if (g0) memcpy(g0 + (size_t)i * 64, a0, 256);
		if (g1) memcpy(g1 + (size_t)i * 64, a1, 256);
	}
}
/* density-only path: NGPNetworks.density, ngp_network.py:86-89 */
EXPORT void orc_density_fwd(uint32_t n, const float *feat, const float *wd, float *out /*[n]*/) {
	for (uint32_t i = 0; i < n; ++i) {
		float hh[64], dd[16];
		matvec(wd, 64, 32, feat + (size_t)i * 32, hh, 1);
		matvec(wd + 2048, 16, 64, hh, dd, 0);
		out[i] = dd[0];
	}
}
/* Backward of the chain above (what Jittor autograd / fully_fused_mlp.py:88-145 produce):
 * dL/dfeat [n,32], dL/dwd [3072], dL/dwc [7168] (rows >= 3 of V2 stay zero, fully_fused_mlp.py:136). */
EXPORT void orc_field_bwd(uint32_t n, const float *feat, const float *sh, const float *wd, const float *wc, const float *dout /*[n,4]*/,
                          float *dfeat /*[n,32]*/, float *dwd, float *dwc) {
	memset(dwd, 0, 3072 * 4); memset(dwc, 0, 7168 * 4);
	const float *W0 = wd, *W1 = wd + 2048, *V0 = wc, *V1 = wc + 2048, *V2 = wc + 6144;
	float *dW0 = dwd, *dW1 = dwd + 2048, *dV0 = dwc, *dV1 = dwc + 2048, *dV2 = dwc + 6144;
	for (uint32_t i = 0; i < n; ++i) {
		const float *f = feat + (size_t)i * 32;
		float hh[64], dd[16], in2[32], a0[64], a1[64];
		matvec(W0, 64, 32, f, hh, 1);
		matvec(W1, 16, 64, hh, dd, 0);
		for (int k = 0; k < 16; ++k) { in2[k] = dd[k]; in2[16 + k] = sh[(size_t)i * 16 + k]; }
		matvec(V0, 64, 32, in2, a0, 1);
		matvec(V1, 64, 64, a0, a1, 1);
		const float *go = dout + (size_t)i * 4;
		float da1[64], da0[64], din2[32], ddd[16], dh[64];
		for (int k = 0; k < 64; ++k) { float s = 0; for (int o = 0; o < 3; ++o) s += V2[o * 64 + k] * go[o]; da1[k] = a1[k] > 0 ? s : 0; }
		for (int o = 0; o < 3; ++o) for (int k = 0; k < 64; ++k) dV2[o * 64 + k] += go[o] * a1[k];
		for (int k = 0; k < 64; ++k) { float s = 0; for (int o = 0; o < 64; ++o) s += V1[o * 64 + k] * da1[o]; da0[k] = a0[k] > 0 ? s : 0; }
		for (int o = 0; o < 64; ++o) for (int k = 0; k < 64; ++k) dV1[o * 64 + k] += da1[o] * a0[k];
		for (int k = 0; k < 32; ++k) { float s = 0; for (int o = 0; o < 64; ++o) s += V0[o * 32 + k] * da0[o]; din2[k] = s; }
		for (int o = 0; o < 64; ++o) for (int k = 0; k < 32; ++k) dV0[o * 32 + k] += da0[o] * in2[k];
		for (int k = 0; k < 16; ++k) ddd[k] = din2[k];
		ddd[0] += go[3];
		for (int k = 0; k < 64; ++k) { float s = 0; for (int o = 0; o < 16; ++o) s += W1[o * 64 + k] * ddd[o]; dh[k] = hh[k] > 0 ? s : 0; }
		for (int o = 0; o < 16; ++o) for (int k = 0; k < 64; ++k) dW1[o * 64 + k] += ddd[o] * hh[k];
		for (int k = 0; k < 32; ++k) { float s = 0; for (int o = 0; o < 64; ++o) s += W0[o * 32 + k] * dh[o]; dfeat[(size_t)i * 32 + k] = s; }
		for (int o = 0; o < 64; ++o) for (int k = 0; k < 32; ++k) dW0[o * 32 + k] += dh[o] * f[k];
	}
}

/* ------------------------------------------------------------------ sampler constants */
/* density_grid_sampler.py:35-39, 96-116 */
#define NERF_GRIDSIZE 128u
#define NERF_STEPS 1024u
static const float SQRT3 = 1.73205080757f;
static inline float min_cone_stepsize(void) { return SQRT3 / NERF_STEPS; }
static inline float max_cone_stepsize(int cascades) { return min_cone_stepsize() * (1 << (cascades - 1)) * NERF_STEPS / NERF_GRIDSIZE; }
static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); }
static inline float calc_dt(float t, float cone_angle, int const_dt, int cascades) {     /* density_grid_sampler.py:107-115 */
	if (const_dt) return min_cone_stepsize() * 0.5;
	return clampf(t * cone_angle, min_cone_stepsize(), max_cone_stepsize(cascades));
}
/* ray_sampler_header.h:642-667 */
static inline uint32_t expand_bits(uint32_t v) {
	v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u; return v;
}
static inline uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) { return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2); }
static inline uint32_t morton3D_invert(uint32_t x) {
	x = x & 0x49249249; x = (x | (x >> 2)) & 0xc30c30c3; x = (x | (x >> 4)) & 0x0f00f00f;
	x = (x | (x >> 8)) & 0xff0000ff; x = (x | (x >> 16)) & 0x0000ffff; return x;
}
EXPORT uint32_t orc_morton3D(uint32_t x, uint32_t y, uint32_t z) { return morton3D(x, y, z); }
EXPORT uint32_t orc_morton3D_invert(uint32_t x) { return morton3D_invert(x); }

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
/* ray_sampler_header.h:60-77 */
static inline int mip_from_pos(const float p[3], int cascades) {
	int exponent;
	float m = fmaxf(fmaxf(fabsf(p[0] - 0.5f), fabsf(p[1] - 0.5f)), fabsf(p[2] - 0.5f));
	frexpf(m, &exponent);
	return imin(cascades - 1, imax(0, exponent + 1));
}
static inline int mip_from_dt(float dt, const float p[3], int cascades) {
	int mip = mip_from_pos(p, cascades);
	dt *= 2 * NERF_GRIDSIZE;
	if (dt < 1.f) return mip;
	int exponent; frexpf(dt, &exponent);
	return imin(cascades - 1, imax(exponent, mip));
}
/* ray_sampler_header.h:755-776 */
static inline uint32_t cascaded_grid_idx_at(const float pin[3], uint32_t mip) {
	float mip_scale = scalbnf(1.0f, -(int)mip);
	int c[3];
	for (int d = 0; d < 3; ++d) {
		float p = pin[d] - 0.5f; p *= mip_scale; p += 0.5f;
		int i = (int)(p * NERF_GRIDSIZE);
		c[d] = i < 0 ? 0 : (i > (int)NERF_GRIDSIZE - 1 ? (int)NERF_GRIDSIZE - 1 : i);
	}
	return morton3D((uint32_t)c[0], (uint32_t)c[1], (uint32_t)c[2]);
}
static inline int occupied_at(const float p[3], const uint8_t *bitfield, uint32_t mip) {
	uint32_t idx = cascaded_grid_idx_at(p, mip);
	return bitfield[idx / 8 + (NERF_GRIDSIZE * NERF_GRIDSIZE * NERF_GRIDSIZE * mip) / 8] & (1 << (idx % 8));
}
/* ray_sampler_header.h:728-753 */
static inline float distance_to_next_voxel(const float pos[3], const float dir[3], const float idir[3], uint32_t res) {
	float t3[3];
	for (int d = 0; d < 3; ++d) {
		float p = res * pos[d];
		t3[d] = (floorf(p + 0.5f + 0.5f * copysignf(1.0f, dir[d])) - p) * idir[d];
	}
	float t = fminf(fminf(t3[0], t3[1]), t3[2]);
	return fmaxf(t / res, 0.0f);
}
static inline float advance_to_next_voxel(float t, float cone, const float pos[3], const float dir[3], const float idir[3], uint32_t res, int const_dt, int cascades) {
	float t_target = t + distance_to_next_voxel(pos, dir, idir, res);
	do { t += calc_dt(t, cone, const_dt, cascades); } while (t < t_target);
	return t;
}
/* BoundingBox::ray_intersect, ray_sampler_header.h:408-465 */
static inline void ray_intersect(float a0, float a1, const float o[3], const float d[3], float *tmin_o, float *tmax_o) {
	float tmin = (a0 - o[0]) / d[0], tmax = (a1 - o[0]) / d[0];
	if (tmin > tmax) { float c = tmin; tmin = tmax; tmax = c; }
	float tymin = (a0 - o[1]) / d[1], tymax = (a1 - o[1]) / d[1];
	if (tymin > tymax) { float c = tymin; tymin = tymax; tymax = c; }
	if (tmin > tymax || tymin > tmax) { *tmin_o = FLT_MAX; *tmax_o = FLT_MAX; return; }
	if (tymin > tmin) tmin = tymin;
	if (tymax < tmax) tmax = tymax;
	float tzmin = (a0 - o[2]) / d[2], tzmax = (a1 - o[2]) / d[2];
	if (tzmin > tzmax) { float c = tzmin; tzmin = tzmax; tzmax = c; }
	if (tmin > tzmax || tzmin > tmax) { *tmin_o = FLT_MAX; *tmax_o = FLT_MAX; return; }
	if (tzmin > tmin) tmin = tzmin;
	if (tzmax < tmax) tmax = tzmax;
	*tmin_o = tmin; *tmax_o = tmax;
}
static inline int aabb_contains(float a0, float a1, const float p[3]) {
	return p[0] >= a0 && p[0] <= a1 && p[1] >= a0 && p[1] <= a1 && p[2] >= a0 && p[2] <= a1;
}
static inline float warp_dt(float dt, int cascades) {                                    /* ray_sampler_header.h:839-843 */
	float max_stepsize = min_cone_stepsize() * (1 << (cascades - 1));
	return (dt - min_cone_stepsize()) / (max_stepsize - min_cone_stepsize());
}
static inline float unwarp_dt(float dt, int cascades) {                                  /* calc_rgb.h:4-8 */
	float max_stepsize = min_cone_stepsize() * (1 << (cascades - 1));
	return dt * (max_stepsize - min_cone_stepsize()) + min_cone_stepsize();
}

/* ------------------------------------------------------------------ ray marcher */
/* samplers/density_grid_sampler/op_header/ray_sampler.h:4-114, launched as ray_sampler.py:34-62:
 * coords zeroed, serial ray order (what a one-thread-at-a-time launcher gives the atomics), rng advanced by 2^32 on return.
 * counters[0] = rays that got a slot, counters[1] = total steps reserved (incl. overflowed rays). */
EXPORT void orc_march_rays(uint32_t n_rays, float a0, float a1, uint32_t max_samples, const float *rays_o, const float *rays_d,
                           const uint8_t *bitfield, float cone_angle, float near_distance, int const_dt, int cascades,
                           uint64_t *rng_state, uint32_t *counters, int32_t *ray_indices, uint32_t *numsteps_out, float *coords) {
	memset(coords, 0, (size_t)max_samples * 7 * 4);
	counters[0] = counters[1] = 0;
	for (uint32_t i = 0; i < n_rays; ++i) {
		pcg32_t rng = {rng_state[0], rng_state[1]};
		pcg_advance(&rng, (int64_t)(uint32_t)(i * 8u));                                   /* ray_sampler.h:30 */
		const float *o = rays_o + 3 * i, *d = rays_d + 3 * i;
		float tmin, tmax; ray_intersect(a0, a1, o, d, &tmin, &tmax);
		tmin = fmaxf(tmin, near_distance);
		float startt = tmin;
		startt += calc_dt(startt, cone_angle, const_dt, cascades) * pcg_next_float(&rng); /* :48 */
		float idir[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
		uint32_t j = 0; float t = startt; float pos[3];
		for (;;) {
			for (int k = 0; k < 3; ++k) pos[k] = o[k] + t * d[k];
			if (!(aabb_contains(a0, a1, pos) && j < NERF_STEPS)) break;
			float dt = calc_dt(t, cone_angle, const_dt, cascades);
			uint32_t mip = (uint32_t)mip_from_dt(dt, pos, cascades);
			if (occupied_at(pos, bitfield, mip)) { ++j; t += dt; }
			else t = advance_to_next_voxel(t, cone_angle, pos, d, idir, NERF_GRIDSIZE >> mip, const_dt, cascades);
		}
		uint32_t numsteps = j;
		uint32_t base = counters[1]; counters[1] += numsteps;
		if (base + numsteps > max_samples) { numsteps_out[2 * i] = 0; numsteps_out[2 * i + 1] = base; continue; }
		float *out = coords + (size_t)base * 7;
		uint32_t ray_idx = counters[0]++;
		ray_indices[i] = (int32_t)ray_idx;
		numsteps_out[2 * i] = numsteps; numsteps_out[2 * i + 1] = base;
		if (j == 0) { ray_indices[i] = -1; continue; }
		float wdir[3] = {(d[0] + 1.0f) * 0.5f, (d[1] + 1.0f) * 0.5f, (d[2] + 1.0f) * 0.5f};
		t = startt; j = 0;
		for (;;) {
			for (int k = 0; k < 3; ++k) pos[k] = o[k] + t * d[k];
			if (!(aabb_contains(a0, a1, pos) && j < numsteps)) break;
			float dt = calc_dt(t, cone_angle, const_dt, cascades);
			uint32_t mip = (uint32_t)mip_from_dt(dt, pos, cascades);
			if (occupied_at(pos, bitfield, mip)) {
				float *c = out + (size_t)j * 7;
				for (int k = 0; k < 3; ++k) c[k] = (pos[k] - a0) / (a1 - a0);
				c[3] = warp_dt(dt, cascades);
				c[4] = wdir[0]; c[5] = wdir[1]; c[6] = wdir[2];
				++j; t += dt;
			} else t = advance_to_next_voxel(t, cone_angle, pos, d, idir, NERF_GRIDSIZE >> mip, const_dt, cascades);
		}
	}
	pcg32_t rng = {rng_state[0], rng_state[1]};
	pcg_advance(&rng, 1ll << 32);                                                         /* ray_sampler.py:61 */
	rng_state[0] = rng.state;
}

/* ------------------------------------------------------------------ compaction */
/* op_header/compacted_coord.h:4-76 (the transmittance loop has no observable effect, :40-43), compacted_coord.py:38 zero-fills. */
EXPORT void orc_compact_coords(uint32_t n_rays, uint32_t cap, const float *coords_in, const uint32_t *numsteps_in,
                               float *coords_out /*[cap,7]*/, uint32_t *numsteps_out, uint32_t *counter /*[1]*/) {
	memset(coords_out, 0, (size_t)cap * 28);
	counter[0] = 0;
	for (uint32_t i = 0; i < n_rays; ++i) {
		uint32_t numsteps = numsteps_in[2 * i], base = numsteps_in[2 * i + 1];
		uint32_t cbase = counter[0]; counter[0] += numsteps;
		uint32_t cn = cap - (cap < cbase ? cap : cbase); if (numsteps < cn) cn = numsteps;
		numsteps_out[2 * i] = cn; numsteps_out[2 * i + 1] = cbase;
		if (cn == 0) continue;
		memcpy(coords_out + (size_t)cbase * 7, coords_in + (size_t)base * 7, (size_t)cn * 28);
	}
}

/* ------------------------------------------------------------------ compositing */
static inline float logistic(float x) { return 1.0f / (1.0f + expf(-x)); }              /* ray_sampler_header.h:895-898 */
/* op_header/calc_rgb.h:10-74 */
EXPORT void orc_composite_fwd(uint32_t n_rays, const void *net /*[N,4] T*/, const float *coords, const uint32_t *numsteps /*uncompacted*/,
                              const uint32_t *numsteps_c, const float *bg /*[n,3]*/, int cascades, float *rgb_out, int is_half) {
	for (uint32_t i = 0; i < n_rays; ++i) {
		uint32_t ns = numsteps_c[2 * i], base = numsteps_c[2 * i + 1];
		if (ns == 0) { for (int c = 0; c < 3; ++c) rgb_out[3 * i + c] = bg[3 * i + c]; continue; }
		float T = 1.f, ray[3] = {0, 0, 0};
		for (uint32_t k = 0; k < ns; ++k) {
			size_t s = (size_t)base + k;
			float rgb[3]; for (int c = 0; c < 3; ++c) rgb[c] = logistic(ldT(net, s * 4 + c, is_half));
			float dt = unwarp_dt(coords[s * 7 + 3], cascades);
			float density = expf(ldT(net, s * 4 + 3, is_half));
			float alpha = 1.f - expf(-density * dt);
			float weight = alpha * T;
			for (int c = 0; c < 3; ++c) ray[c] += weight * rgb[c];
			T *= (1.f - alpha);
		}
		if (ns == numsteps[2 * i]) for (int c = 0; c < 3; ++c) ray[c] += T * bg[3 * i + c];
		for (int c = 0; c < 3; ++c) rgb_out[3 * i + c] = ray[c];
	}
}
/* op_header/calc_rgb.h:76-148; dout zeroed first (calc_rgb.py:93) */
EXPORT void orc_composite_bwd(uint32_t n_rays, uint32_t n_elems, const void *net, const float *coords, const uint32_t *numsteps_c,
                              const float *loss_grad /*[n,3]*/, const float *rgb_ray /*[n,3] forward output*/, float density_grid_mean,
                              int cascades, void *dout /*[N,4] T*/, int is_half) {
	memset(dout, 0, (size_t)n_elems * 4 * (is_half ? 2 : 4));
	float loss_scale = 128; loss_scale /= n_rays;
	const float l1_reg = density_grid_mean < 0.01f ? 1e-4f : 0.0f;
	for (uint32_t i = 0; i < n_rays; ++i) {
		uint32_t ns = numsteps_c[2 * i], base = numsteps_c[2 * i + 1];
		const float *G = loss_grad + 3 * i, *R = rgb_ray + 3 * i;
		float T = 1.f, ray2[3] = {0, 0, 0};
		for (uint32_t k = 0; k < ns; ++k) {
			size_t s = (size_t)base + k;
			float o[4]; for (int c = 0; c < 4; ++c) o[c] = ldT(net, s * 4 + c, is_half);
			float rgb[3]; for (int c = 0; c < 3; ++c) rgb[c] = logistic(o[c]);
			float dt = unwarp_dt(coords[s * 7 + 3], cascades);
			float density = expf(o[3]);
			float alpha = 1.f - expf(-density * dt);
			float weight = alpha * T;
			for (int c = 0; c < 3; ++c) ray2[c] += weight * rgb[c];
			T *= (1.f - alpha);
			float dv[3];
			for (int c = 0; c < 3; ++c) {
				float suffix = R[c] - ray2[c];
				float dl = weight * G[c];
				float sg = logistic(o[c]);
				stT(dout, s * 4 + c, loss_scale * (dl * (sg * (1 - sg)) + fmaxf(0.0f, 0.0f * o[c])), is_half);
				dv[c] = G[c] * (T * rgb[c] - suffix);
			}
			float dotv = dv[0] + (dv[1] + dv[2]); /* Eigen's unrolled 3-vector dot() reduces as a0 + (a1 + a2) */
			float dd = expf(clampf(o[3], -15.0f, 15.0f));
			float dmlp = dd * (dt * dotv);
			stT(dout, s * 4 + 3, loss_scale * dmlp + (o[3] < 0 ? -l1_reg : 0.0f), is_half);
		}
	}
}
/* op_header/calc_rgb.h:151-212 */
EXPORT void orc_composite_inference(uint32_t n_rays, const void *net, const float *coords, const uint32_t *numsteps, int cascades,
                                    float *rgb_out, float *alpha_out, int is_half) {
	for (uint32_t i = 0; i < n_rays; ++i) {
		uint32_t ns = numsteps[2 * i], base = numsteps[2 * i + 1];
		if (ns == 0) { rgb_out[3 * i] = rgb_out[3 * i + 1] = rgb_out[3 * i + 2] = 0; alpha_out[i] = 0; continue; }
		float T = 1.f, ray[3] = {0, 0, 0};
		for (uint32_t k = 0; k < ns; ++k) {
			size_t s = (size_t)base + k;
			float dt = unwarp_dt(coords[s * 7 + 3], cascades);
			float density = expf(ldT(net, s * 4 + 3, is_half));
			float alpha = 1.f - expf(-density * dt);
			float weight = alpha * T;
			for (int c = 0; c < 3; ++c) ray[c] += weight * logistic(ldT(net, s * 4 + c, is_half));
			T *= (1.f - alpha);
		}
		for (int c = 0; c < 3; ++c) rgb_out[3 * i + c] = ray[c];
		alpha_out[i] = 1 - T;
	}
}

/* ------------------------------------------------------------------ density grid maintenance */
/* op_header/mark_untrained_density_grid.h:3-47; xforms [n_img,4,3] = column-major 3x4 (dataset.py:165) */
EXPORT void orc_grid_mark_untrained(uint32_t n_elements, float *grid, uint32_t n_images, const float *focal /*[n,2]*/, const float *xforms, int W, int H) {
	memset(grid, 0, (size_t)n_elements * 4);   /* mark_untrained_density_grid.py:21: output starts as zeros */
	const uint32_t G3 = NERF_GRIDSIZE * NERF_GRIDSIZE * NERF_GRIDSIZE;
	float half_resx = W * 0.5f, half_resy = H * 0.5f;
	for (uint32_t i = 0; i < n_elements; ++i) {
		uint32_t level = i / G3, pos_idx = i % G3;
		uint32_t x = morton3D_invert(pos_idx >> 0), y = morton3D_invert(pos_idx >> 1), z = morton3D_invert(pos_idx >> 2);
		float sc = scalbnf(1.0f, (int)level);
		float pos[3] = {(((float)x + 0.5f) / NERF_GRIDSIZE - 0.5f) * sc + 0.5f, (((float)y + 0.5f) / NERF_GRIDSIZE - 0.5f) * sc + 0.5f,
		                (((float)z + 0.5f) / NERF_GRIDSIZE - 0.5f) * sc + 0.5f};
		float voxel_radius = 0.5f * SQRT3 * sc / NERF_GRIDSIZE;
		int count = 0;
		for (uint32_t j = 0; j < n_images; ++j) {
			const float *m = xforms + (size_t)j * 12; /* col c at m[3c..3c+2] */
			float pl[3] = {pos[0] - m[9], pos[1] - m[10], pos[2] - m[11]};
			float xx = pl[0] * m[0] + pl[1] * m[1] + pl[2] * m[2];
			float yy = pl[0] * m[3] + pl[1] * m[4] + pl[2] * m[5];
			float zz = pl[0] * m[6] + pl[1] * m[7] + pl[2] * m[8];
			if (zz > 0.f) {
				if (fabsf(xx) - voxel_radius < zz / focal[2 * j] * half_resx && fabsf(yy) - voxel_radius < zz / focal[2 * j + 1] * half_resy) { count++; break; }
			}
		}
		if ((grid[i] < 0) != (count <= 0)) grid[i] = (count > 0) ? 0.f : -1.f;
	}
}
/* op_header/generate_grid_samples_nerf_nonuniform.h:3-35; host rng.advance() after (generate_grid_samples…py:44) */
EXPORT void orc_grid_generate_samples(uint32_t n, uint64_t *rng_state, uint32_t step, float a0, float a1, const float *grid_in,
                                      float *out_pos /*[n,3]*/, uint32_t *indices, uint32_t n_cascades, float thresh) {
	const uint32_t G3 = NERF_GRIDSIZE * NERF_GRIDSIZE * NERF_GRIDSIZE;
	for (uint32_t i = 0; i < n; ++i) {
		pcg32_t rng = {rng_state[0], rng_state[1]};
		pcg_advance(&rng, (int64_t)(uint32_t)(i * 4u));
		uint32_t level = (uint32_t)(pcg_next_float(&rng) * n_cascades) % n_cascades;
		uint32_t idx = 0;
		for (uint32_t j = 0; j < 10; ++j) {
			idx = ((i + step * n) * 56924617u + j * 19349663u + 96925573u) % G3;
			idx += level * G3;
			if (grid_in[idx] > thresh) break;
		}
		uint32_t pos_idx = idx % G3;
		uint32_t x = morton3D_invert(pos_idx >> 0), y = morton3D_invert(pos_idx >> 1), z = morton3D_invert(pos_idx >> 2);
		float r0 = pcg_next_float(&rng), r1 = pcg_next_float(&rng), r2 = pcg_next_float(&rng);
		float sc = scalbnf(1.0f, (int)level);
		float pos[3] = {(((float)x + r0) / NERF_GRIDSIZE - 0.5f) * sc + 0.5f, (((float)y + r1) / NERF_GRIDSIZE - 0.5f) * sc + 0.5f,
		                (((float)z + r2) / NERF_GRIDSIZE - 0.5f) * sc + 0.5f};
		for (int k = 0; k < 3; ++k) out_pos[3 * i + k] = (pos[k] - a0) / (a1 - a0);
		indices[i] = idx;
	}
	pcg32_t rng = {rng_state[0], rng_state[1]};
	pcg_advance(&rng, 1ll << 32);
	rng_state[0] = rng.state;
}
/* op_header/splat_grid_samples_nerf_max_nearest_neighbor.h:5-23 (width 1) */
EXPORT void orc_grid_splat_max(uint32_t n, const uint32_t *indices, const void *mlp_out, float *grid_tmp, int is_half) {
	for (uint32_t i = 0; i < n; ++i) {
		float mlp = expf(ldT(mlp_out, i, is_half));
		float thick = mlp * scalbnf(min_cone_stepsize(), 0);
		uint32_t u, cur; memcpy(&u, &thick, 4); memcpy(&cur, &grid_tmp[indices[i]], 4);
		if (u > cur) memcpy(&grid_tmp[indices[i]], &u, 4);
	}
}
/* op_header/ema_grid_samples_nerf.h:3-25 */
EXPORT void orc_grid_ema(uint32_t n, float decay, float *grid, const float *grid_tmp) {
	for (uint32_t i = 0; i < n; ++i) {
		float prev = grid[i];
		grid[i] = (prev < 0.f) ? prev : fmaxf(prev * decay, grid_tmp[i]);
	}
}
/* update_bitfield.py:15-37 + op_header/update_bitfield.h:23-69; mean over cascade 0 only */
EXPORT void orc_grid_update_bitfield(const float *grid, int cascades, float *mean /*[1]*/, uint8_t *bitfield) {
	const uint32_t G3 = NERF_GRIDSIZE * NERF_GRIDSIZE * NERF_GRIDSIZE;
	/* reduce_sum (update_bitfield.py:25-28) is a block-wise tree reduction on the GPU, accurate to ~1e-7.  A SERIAL fp32 sum of 2 M nearly equal terms is not (measured:
	   0.7 % high on the first refresh of a training run, when every trained cell holds ~0.0017 - enough to put the threshold above every cell and empty the bitfield),
	   so the restatement accumulates in double and rounds once. */
	double acc = 0.0;
	for (uint32_t i = 0; i < G3; ++i) acc += (double)(fmaxf(grid[i], 0.f) / (G3));
	const float s = (float)acc;
	mean[0] = s;
	float thresh = 0.01f < s ? 0.01f : s;
	for (uint32_t i = 0; i < G3 / 8 * (uint32_t)cascades; ++i) {
		uint8_t bits = 0;
		for (uint8_t j = 0; j < 8; ++j) bits |= grid[(size_t)i * 8 + j] > thresh ? ((uint8_t)1 << j) : 0;
		bitfield[i] = bits;
	}
	for (int level = 1; level < cascades; ++level) {
		const uint8_t *prev = bitfield + (size_t)G3 * (level - 1) / 8;
		uint8_t *next = bitfield + (size_t)G3 * level / 8;
		for (uint32_t i = 0; i < G3 / 64; ++i) {
			uint8_t bits = 0;
			for (uint8_t j = 0; j < 8; ++j) bits |= prev[i * 8 + j] > 0 ? ((uint8_t)1 << j) : 0;
			uint32_t x = morton3D_invert(i >> 0) + NERF_GRIDSIZE / 8, y = morton3D_invert(i >> 1) + NERF_GRIDSIZE / 8, z = morton3D_invert(i >> 2) + NERF_GRIDSIZE / 8;
			next[morton3D(x, y, z)] |= bits;
		}
	}
}

/* ------------------------------------------------------------------ loss + optimiser */
/* models/losses/huber_loss.py:6-14 (unreduced) and its elementwise derivative (autograd of sum) */
EXPORT void orc_huber(uint32_t n, const float *x, const float *target, float delta, float *loss, float *grad) {
	for (uint32_t i = 0; i < n; ++i) {
		float d = x[i] - target[i], rel = fabsf(d);
		if (loss) loss[i] = rel > delta ? rel - 0.5f * delta : 0.5f / delta * rel * rel;
		if (grad) grad[i] = rel > delta ? (d > 0 ? 1.0f : -1.0f) : d / delta;
	}
}
/* Adam (Jittor nn.Adam, external — standard bias-corrected form; hyper-parameters ngp_base.py:21-26), then
 * EMA.ema_step (optims/ema.py:26-37) which OVERWRITES the live parameter.  step is 1-based for both. */
EXPORT void orc_adam_ema_step(uint64_t n, float *p, const float *g, float *m, float *v, float *ema, float lr, float b0, float b1,
                              float eps, uint32_t step, float ema_decay) {
	double bc0 = 1.0 - pow((double)b0, (double)step), bc1 = 1.0 - pow((double)b1, (double)step);
	float step_size = (float)((double)lr * sqrt(bc1) / bc0);
	float debias_old = (float)(1.0 - pow((double)ema_decay, (double)step - 1.0));
	float debias_new = (float)(1.0 / (1.0 - pow((double)ema_decay, (double)step)));
	for (uint64_t i = 0; i < n; ++i) {
		float gi = g[i];
		float mi = b0 * m[i] + (1 - b0) * gi;
		float vi = b1 * v[i] + (1 - b1) * gi * gi;
		m[i] = mi; v[i] = vi;
		float pi = p[i] - mi * step_size / (sqrtf(vi) + eps);
		if (ema) {
			pi = ((1 - ema_decay) * pi + ema_decay * ema[i] * debias_old) * debias_new;
			ema[i] = pi;
		}
		p[i] = pi;
	}
}

/* ------------------------------------------------------------------ ray generation */
/* dataset/dataset.py:172-188 (generate_random_data): pixel index -> ray; xforms [n_img,4,3] col-major 3x4 */
EXPORT void orc_generate_rays(uint32_t n, const int64_t *index, int W, int H, const float *focal /*[n_img,2]*/, const float *pp /*[n_img,2]*/,
                              const float *xforms, int32_t *img_id, float *rays_o, float *rays_d) {
	for (uint32_t i = 0; i < n; ++i) {
		int64_t id = index[i] / ((int64_t)H * W), off = index[i] % ((int64_t)H * W);
		const float *m = xforms + (size_t)id * 12;
		float x = ((float)(off % W) + 0.5f) / W, y = ((float)(off / W) + 0.5f) / H;
		float dc[3] = {(x - pp[2 * id]) * W / focal[2 * id], (y - pp[2 * id + 1]) * H / focal[2 * id + 1], 1.0f};
		float d[3];
		for (int r = 0; r < 3; ++r) d[r] = m[r] * dc[0] + m[3 + r] * dc[1] + m[6 + r] * dc[2];
		float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
		for (int r = 0; r < 3; ++r) { rays_d[3 * i + r] = d[r] / nrm; rays_o[3 * i + r] = m[9 + r]; }
		img_id[i] = (int32_t)id;
	}
}
