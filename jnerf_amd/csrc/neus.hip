// NeuS' SDF -> opacity -> weights -> colour chain (python/jnerf/models/samplers/neus_render/renderer.py:216-252, the second half of NeuSRenderer.render_core) as ONE
// kernel per direction - BASELINE.json configs[4]'s "SDF-to-density render path".  The reference expresses it as ~25 Jittor tensor ops over [batch, n] tensors plus
// their autograd; here one wavefront owns one ray (rays have a FIXED number of sections: 128 inside the unit sphere + 32 of the background model), lanes own sections,
// the transmittance product and the backward's suffix sums are wavefront scans, and nothing but the inputs and the per-section weights touches memory.
//
//   iter_cos = -(relu(-cos/2 + 1/2) (1 - r) + relu(-cos) r)                r = cos_anneal_ratio                    renderer.py:218-219
//   prev/next = sigmoid((sdf -/+ iter_cos * dist / 2) * inv_s);  p = prev - next;  c = prev;  a = clip((p + 1e-5) / (c + 1e-5), 0, 1)      :222-231
//   with a background model:  alpha_i = a_i inside_i + bg_alpha_i (1 - inside_i)  (i < n),  = bg_alpha_i  (n <= i < n_total); colours alike                 :238-244
//   w_i = alpha_i * prod_{j<i} (1 - alpha_j + 1e-6);   colour = sum_i w_i colour_i                                                                     :248-251
// The clip is Jittor's safe_clip: the VALUE is clamped, the gradient passes through (jnerf_amd/neus_network.py:safe_clip) - the backward below treats it as identity.
// fp32 throughout; the scans multiply / add in tree order, so weights differ from a serial cumprod by a few ulp (tests: 1e-5 relative).
#include "ngp_common.h"

#define NEUS_MAX_SECTIONS 512u       // per ray; LDS of the backward: 2 floats per section and wavefront
#define NEUS_RAYS_PER_BLOCK 4u

struct NeusIn {
	const float *sdf, *cosv, *dists, *inv_s, *color, *inside, *bg_alpha, *bg_color;
	float ratio;
	uint32_t n_rays, n, n_total;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

struct Section { float alpha, col[3], a_raw, p, c, s_prev, s_next, ins; };
// the opacity and colour of section i of ray r after the blend with the background model
__device__ __forceinline__ Section neus_section(const NeusIn &in, uint32_t r, uint32_t i, float inv_s) {
	Section o;
	o.ins = 1.0f; o.p = o.c = o.a_raw = o.s_prev = o.s_next = 0.f;
	if (i < in.n) {
		const size_t k = (size_t)r * in.n + i;
		const float sdf = in.sdf[k], cs = in.cosv[k], d = in.dists[k];
		const float iter_cos = -(fmaxf(-cs * 0.5f + 0.5f, 0.f) * (1.0f - in.ratio) + fmaxf(-cs, 0.f) * in.ratio);
		o.s_next = sigmoidf_((sdf + iter_cos * d * 0.5f) * inv_s);
		o.s_prev = sigmoidf_((sdf - iter_cos * d * 0.5f) * inv_s);
		o.p = o.s_prev - o.s_next; o.c = o.s_prev;
		o.a_raw = (o.p + 1e-5f) / (o.c + 1e-5f);
		const float a = fminf(fmaxf(o.a_raw, 0.f), 1.f);
		const float *col = in.color + k * 3;
		if (in.bg_alpha) {
			o.ins = in.inside[k];
			const size_t kb = (size_t)r * in.n_total + i;
			o.alpha = a * o.ins + in.bg_alpha[kb] * (1.0f - o.ins);
#pragma unroll
			for (int ch = 0; ch < 3; ++ch) o.col[ch] = col[ch] * o.ins + in.bg_color[kb * 3 + ch] * (1.0f - o.ins);
		} else {
			o.alpha = a;
#pragma unroll
			for (int ch = 0; ch < 3; ++ch) o.col[ch] = col[ch];
		}
	} else {
		const size_t kb = (size_t)r * in.n_total + i;
		o.alpha = in.bg_alpha[kb]; o.ins = 0.f;
#pragma unroll
		for (int ch = 0; ch < 3; ++ch) o.col[ch] = in.bg_color[kb * 3 + ch];
	}
	return o;
}

__device__ __forceinline__ float wave_incl_prod(float v, uint32_t lane) {
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const float y = __shfl_up(v, off); if (lane >= (uint32_t)off) v *= y; }
	return v;
}
__device__ __forceinline__ float wave_incl_suffix_sum(float v, uint32_t lane) {
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const float y = __shfl_down(v, off); if (lane + (uint32_t)off < 64u) v += y; }
	return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
	return v;
}

__global__ __launch_bounds__(256) void k_neus_composite_fwd(NeusIn in, float *__restrict__ out_color, float *__restrict__ weights, float *__restrict__ alpha_out,
                                                            float *__restrict__ p_out, float *__restrict__ c_out) {
	const uint32_t lane = threadIdx.x & 63u, r = blockIdx.x * NEUS_RAYS_PER_BLOCK + (threadIdx.x >> 6);
	if (r >= in.n_rays) return;
	const float inv_s = *in.inv_s;
	float T = 1.0f, acc[3] = {0.f, 0.f, 0.f};
	for (uint32_t base = 0; base < in.n_total; base += 64u) {
		const uint32_t i = base + lane;
		const bool live = i < in.n_total;
		Section s; s.alpha = 0.f; s.col[0] = s.col[1] = s.col[2] = 0.f;
		if (live) s = neus_section(in, r, i, inv_s);
		const float keep = live ? 1.0f - s.alpha + 1e-6f : 1.0f;
		const float incl = wave_incl_prod(keep, lane);
		float excl = __shfl_up(incl, 1); if (lane == 0) excl = 1.0f;
		const float w = s.alpha * (T * excl);
		if (live) {
			const size_t kb = (size_t)r * in.n_total + i;
			weights[kb] = w; alpha_out[kb] = s.alpha;
			if (i < in.n) { p_out[(size_t)r * in.n + i] = s.p; c_out[(size_t)r * in.n + i] = s.c; }
#pragma unroll
			for (int ch = 0; ch < 3; ++ch) acc[ch] += w * s.col[ch];
		}
		T *= __shfl(incl, 63);
	}
#pragma unroll
	for (int ch = 0; ch < 3; ++ch) acc[ch] = wave_sum(acc[ch]);
	if (lane == 0) { out_color[(size_t)r * 3] = acc[0]; out_color[(size_t)r * 3 + 1] = acc[1]; out_color[(size_t)r * 3 + 2] = acc[2]; }
}

// backward: dL/dw_i = G_i = g_color . colour_i + g_weights_i;   dL/dalpha_i = G_i T_i - (sum_{k>i} G_k w_k) / (1 - alpha_i + 1e-6);   dL/dcolour_i = w_i g_color
__global__ __launch_bounds__(256) void k_neus_composite_bwd(NeusIn in, const float *__restrict__ g_color, const float *__restrict__ g_weights, float *__restrict__ d_sdf,
                                                            float *__restrict__ d_cos, float *__restrict__ d_inv_s_partial, float *__restrict__ d_color,
                                                            float *__restrict__ d_bg_alpha, float *__restrict__ d_bg_color) {
	__shared__ float sh[NEUS_RAYS_PER_BLOCK][2][NEUS_MAX_SECTIONS];          // per wavefront: T_i and G_i w_i
	const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, r = blockIdx.x * NEUS_RAYS_PER_BLOCK + wv;
	if (r >= in.n_rays) return;
	const float inv_s = *in.inv_s;
	const float gc[3] = {g_color[(size_t)r * 3], g_color[(size_t)r * 3 + 1], g_color[(size_t)r * 3 + 2]};
	float *shT = sh[wv][0], *shGw = sh[wv][1];
	float T = 1.0f;
	for (uint32_t base = 0; base < in.n_total; base += 64u) {                  // pass 1, front to back: transmittance and G_i w_i
		const uint32_t i = base + lane;
		const bool live = i < in.n_total;
		Section s; s.alpha = 0.f; s.col[0] = s.col[1] = s.col[2] = 0.f;
		if (live) s = neus_section(in, r, i, inv_s);
		const float keep = live ? 1.0f - s.alpha + 1e-6f : 1.0f;
		const float incl = wave_incl_prod(keep, lane);
		float excl = __shfl_up(incl, 1); if (lane == 0) excl = 1.0f;
		const float Ti = T * excl;
		if (live) {
			const float G = gc[0] * s.col[0] + gc[1] * s.col[1] + gc[2] * s.col[2] + (g_weights ? g_weights[(size_t)r * in.n_total + i] : 0.f);
			shT[i] = Ti; shGw[i] = G * (s.alpha * Ti);
		}
		T *= __shfl(incl, 63);
	}
	float tail = 0.f, ds_acc = 0.f;                                              // pass 2, back to front: suffix sums, then everything per section
	const uint32_t n_chunks = (in.n_total + 63u) / 64u;
	for (uint32_t ck = n_chunks; ck-- > 0;) {
		const uint32_t i = ck * 64u + lane;
		const bool live = i < in.n_total;
		const float gw = live ? shGw[i] : 0.f;
		const float incl = wave_incl_suffix_sum(gw, lane);
		const float after = incl - gw + tail;                                    // sum over k > i
		tail += __shfl(incl, 0);
		if (!live) continue;
		const Section s = neus_section(in, r, i, inv_s);
		const float Ti = shT[i], w = s.alpha * Ti;
		const float G = gc[0] * s.col[0] + gc[1] * s.col[1] + gc[2] * s.col[2] + (g_weights ? g_weights[(size_t)r * in.n_total + i] : 0.f);
		const float d_alpha = G * Ti - after / (1.0f - s.alpha + 1e-6f);
		const size_t kb = (size_t)r * in.n_total + i;
		if (i < in.n) {
			const size_t k = (size_t)r * in.n + i;
			const float ins = in.bg_alpha ? s.ins : 1.0f;
#pragma unroll
			for (int ch = 0; ch < 3; ++ch) d_color[k * 3 + ch] = w * gc[ch] * ins;
			if (in.bg_alpha) {
				d_bg_alpha[kb] = d_alpha * (1.0f - ins);
#pragma unroll
				for (int ch = 0; ch < 3; ++ch) d_bg_color[kb * 3 + ch] = w * gc[ch] * (1.0f - ins);
			}
			const float d_a = d_alpha * ins;                                        // through safe_clip: identity
			const float ce = s.c + 1e-5f;
			const float d_p = d_a / ce, d_c = -d_a * (s.p + 1e-5f) / (ce * ce);
			const float d_xprev = (d_p + d_c) * s.s_prev * (1.0f - s.s_prev), d_xnext = -d_p * s.s_next * (1.0f - s.s_next);
			const float sdf = in.sdf[k], cs = in.cosv[k], d = in.dists[k];
			const float u = -cs * 0.5f + 0.5f, v = -cs;
			const float iter_cos = -(fmaxf(u, 0.f) * (1.0f - in.ratio) + fmaxf(v, 0.f) * in.ratio);
			d_sdf[k] = (d_xprev + d_xnext) * inv_s;
			const float d_ic = (d_xnext - d_xprev) * d * 0.5f * inv_s;
			d_cos[k] = d_ic * ((u > 0.f ? 0.5f * (1.0f - in.ratio) : 0.f) + (v > 0.f ? in.ratio : 0.f));
			ds_acc += d_xprev * (sdf - iter_cos * d * 0.5f) + d_xnext * (sdf + iter_cos * d * 0.5f);
		} else {
			d_bg_alpha[kb] = d_alpha;
#pragma unroll
			for (int ch = 0; ch < 3; ++ch) d_bg_color[kb * 3 + ch] = w * gc[ch];
		}
	}
	ds_acc = wave_sum(ds_acc);
	if (lane == 0) d_inv_s_partial[r] = ds_acc;
}

static int neus_check(const char *who, uint32_t n_rays, uint32_t n, uint32_t n_total, const float *sdf, const float *cosv, const float *dists, const float *inv_s, const float *color,
                      const float *inside, const float *bg_alpha, const float *bg_color) {
	NGP_REQUIRE(n_rays == 0 || (sdf && cosv && dists && inv_s && color), NGP_E_ARG, "%s: null pointer", who);
	NGP_REQUIRE(n >= 1 && n_total >= n && n_total <= NEUS_MAX_SECTIONS, NGP_E_ARG, "%s: need 1 <= n (%u) <= n_total (%u) <= %u sections per ray", who, n, n_total, NEUS_MAX_SECTIONS);
	NGP_REQUIRE((bg_alpha != nullptr) == (bg_color != nullptr), NGP_E_ARG, "%s: bg_alpha and bg_color come together", who);
	NGP_REQUIRE(bg_alpha ? inside != nullptr : n_total == n, NGP_E_ARG, "%s: a background model needs `inside`; without one n_total must equal n", who);
	return 0;
}
NGP_API int ngp_neus_composite_fwd(void *stream, uint32_t n_rays, uint32_t n, uint32_t n_total, const float *sdf, const float *cosv, const float *dists, const float *inv_s,
                                   const float *color, const float *inside, const float *bg_alpha, const float *bg_color, float cos_anneal_ratio,
                                   float *out_color, float *weights, float *alpha, float *p, float *c) {
	if (int e = neus_check("ngp_neus_composite_fwd", n_rays, n, n_total, sdf, cosv, dists, inv_s, color, inside, bg_alpha, bg_color)) return e;
	NGP_REQUIRE(n_rays == 0 || (out_color && weights && alpha && p && c), NGP_E_ARG, "ngp_neus_composite_fwd: null output");
	if (n_rays == 0) return 0;
	const NeusIn in = {sdf, cosv, dists, inv_s, color, inside, bg_alpha, bg_color, cos_anneal_ratio, n_rays, n, n_total};
	NGP_LAUNCH(k_neus_composite_fwd, dim3(div_up(n_rays, NEUS_RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, in, out_color, weights, alpha, p, c);
	NGP_LAUNCH_CHECK("ngp_neus_composite_fwd");
	return 0;
}
NGP_API int ngp_neus_composite_bwd(void *stream, uint32_t n_rays, uint32_t n, uint32_t n_total, const float *sdf, const float *cosv, const float *dists, const float *inv_s,
                                   const float *color, const float *inside, const float *bg_alpha, const float *bg_color, float cos_anneal_ratio,
                                   const float *g_color, const float *g_weights, float *d_sdf, float *d_cos, float *d_inv_s_partial, float *d_color, float *d_bg_alpha, float *d_bg_color) {
	if (int e = neus_check("ngp_neus_composite_bwd", n_rays, n, n_total, sdf, cosv, dists, inv_s, color, inside, bg_alpha, bg_color)) return e;
	NGP_REQUIRE(n_rays == 0 || (g_color && d_sdf && d_cos && d_inv_s_partial && d_color), NGP_E_ARG, "ngp_neus_composite_bwd: null pointer");
	NGP_REQUIRE(!bg_alpha || (d_bg_alpha && d_bg_color), NGP_E_ARG, "ngp_neus_composite_bwd: a background model needs d_bg_alpha / d_bg_color");
	if (n_rays == 0) return 0;
	const NeusIn in = {sdf, cosv, dists, inv_s, color, inside, bg_alpha, bg_color, cos_anneal_ratio, n_rays, n, n_total};
	NGP_LAUNCH(k_neus_composite_bwd, dim3(div_up(n_rays, NEUS_RAYS_PER_BLOCK)), dim3(256), 0, (hipStream_t)stream, in, g_color, g_weights, d_sdf, d_cos, d_inv_s_partial, d_color,
	           d_bg_alpha, d_bg_color);
	NGP_LAUNCH_CHECK("ngp_neus_composite_bwd");
	return 0;
}
