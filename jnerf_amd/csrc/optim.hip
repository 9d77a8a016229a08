// Parameter update and ray generation.
//  * ngp_adam_ema_step: Adam (Jittor nn.Adam via optims/adam.py; hyper-parameters projects/ngp/configs/ngp_base.py:21-26) followed by
//    EMA.ema_step (optims/ema.py:26-37, which overwrites the live parameter) fused into ONE streaming sweep that also refreshes the fp16
//    copy the gather kernels read and zeroes the gradient for the next step.  The reference spends dozens of elementwise kernels
//    (>= 12 passes over 12-13 M parameters) on this; here it is 5 reads + 5 writes per parameter with 16-byte accesses — the one
//    part of the step that runs against the HBM roofline.
//  * ngp_generate_rays: dataset/dataset.py:172-188 + the target compositing of runner/runner.py:66-68.
#include "ngp_common.h"
#pragma clang fp contract(off)


template <typename G, int EMA /*0 none, 1 separate buffer, 2 the EMA state IS the parameter (v == p at every step boundary)*/, bool HALF, bool ZERO>
__global__ __launch_bounds__(256) void k_adam_ema(uint64_t n4, float4 *__restrict__ p, G *__restrict__ g, float4 *__restrict__ m, float4 *__restrict__ v, float4 *__restrict__ ema,
                                                  uint2 *__restrict__ p_half, AdamConsts c) {
	for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) {
		float gi[4];
		if (sizeof(G) == 4) { float4 t = reinterpret_cast<float4 *>(g)[i]; gi[0] = t.x; gi[1] = t.y; gi[2] = t.z; gi[3] = t.w; if (ZERO) reinterpret_cast<float4 *>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
		else {
			uint2 t = reinterpret_cast<uint2 *>(g)[i];
			float2 a = __half22float2(*reinterpret_cast<__half2 *>(&t.x)), b = __half22float2(*reinterpret_cast<__half2 *>(&t.y));
			gi[0] = a.x * c.g_mul; gi[1] = a.y * c.g_mul; gi[2] = b.x * c.g_mul; gi[3] = b.y * c.g_mul;
			if (ZERO) reinterpret_cast<uint2 *>(g)[i] = make_uint2(0u, 0u);
		}
		float4 P = p[i], M = m[i], V = v[i], E;
		if (EMA == 1) E = ema[i];
		if (EMA == 2) E = P;                       // ema.py:26-37 ends with v <- p, so the stored EMA equals the parameter it is about to blend with
		float *pp = &P.x, *mm = &M.x, *vv = &V.x, *ee = &E.x;
#pragma unroll
		for (int k = 0; k < 4; ++k) adam_ema_update<EMA != 0>(pp[k], mm[k], vv[k], ee[k], gi[k], c);
		p[i] = P; m[i] = M; v[i] = V;
		if (EMA == 1) ema[i] = E;
		if (HALF) {
			__half2 a = __floats2half2_rn(P.x, P.y), b = __floats2half2_rn(P.z, P.w);
			p_half[i] = make_uint2(*reinterpret_cast<uint32_t *>(&a), *reinterpret_cast<uint32_t *>(&b));
		}
	}
}

// fp32 gradient -> fp16 communication buffer (data-parallel all-reduce at half the bytes; the reference's gradients are fp16 to begin with), optionally
// zeroing the source for the next backward in the same sweep.  Streaming, 32 B in / 16 B out per thread.
__global__ __launch_bounds__(256) void k_grad_to_half(uint64_t n8, float4 *__restrict__ src, uint4 *__restrict__ dst, int zero_src, float scale) {
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (uint64_t)gridDim.x * 256) {
		float4 a = src[2 * i], b = src[2 * i + 1];
		a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale; b.x *= scale; b.y *= scale; b.z *= scale; b.w *= scale;
		const __half2 h0 = __floats2half2_rn(a.x, a.y), h1 = __floats2half2_rn(a.z, a.w), h2 = __floats2half2_rn(b.x, b.y), h3 = __floats2half2_rn(b.z, b.w);
		dst[i] = make_uint4(*reinterpret_cast<const uint32_t *>(&h0), *reinterpret_cast<const uint32_t *>(&h1), *reinterpret_cast<const uint32_t *>(&h2), *reinterpret_cast<const uint32_t *>(&h3));
		if (zero_src) { src[2 * i] = make_float4(0.f, 0.f, 0.f, 0.f); src[2 * i + 1] = make_float4(0.f, 0.f, 0.f, 0.f); }
	}
}
NGP_API int ngp_grad_to_half_scaled(void *stream, uint64_t n, float *grad_f32, void *grad_f16, int zero_src, float scale);
NGP_API int ngp_grad_to_half(void *stream, uint64_t n, float *grad_f32, void *grad_f16, int zero_src) { return ngp_grad_to_half_scaled(stream, n, grad_f32, grad_f16, zero_src, 1.0f); }
NGP_API int ngp_grad_to_half_scaled(void *stream, uint64_t n, float *grad_f32, void *grad_f16, int zero_src, float scale) {
	NGP_REQUIRE(n == 0 || (grad_f32 && grad_f16), NGP_E_ARG, "ngp_grad_to_half: null pointer");
	NGP_REQUIRE(n % 8 == 0, NGP_E_ALIGN, "ngp_grad_to_half: n (%llu) must be a multiple of 8", (unsigned long long)n);
	NGP_REQUIRE((((uintptr_t)grad_f32 | (uintptr_t)grad_f16) & 15) == 0, NGP_E_ALIGN, "ngp_grad_to_half: buffers must be 16-byte aligned");
	if (n == 0) return 0;
	const uint64_t n8 = n / 8;
	uint32_t blocks = (uint32_t)((n8 + 255) / 256); if (blocks > 2048 * 4) blocks = 2048 * 4;
	NGP_LAUNCH(k_grad_to_half, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n8, (float4 *)grad_f32, (uint4 *)grad_f16, zero_src, scale);
	NGP_LAUNCH_CHECK("ngp_grad_to_half");
	return 0;
}

NGP_API int ngp_adam_ema_step_scaled(void *stream, uint64_t n, float *p, void *g, int g_dtype, float *m, float *v, float *ema, void *p_half,
                                     float lr, float beta0, float beta1, float eps, uint32_t step, float ema_decay, int zero_grad, float grad_mul);
NGP_API int ngp_adam_ema_step(void *stream, uint64_t n, float *p, void *g, int g_dtype, float *m, float *v, float *ema, void *p_half,
                              float lr, float beta0, float beta1, float eps, uint32_t step, float ema_decay, int zero_grad) {
	return ngp_adam_ema_step_scaled(stream, n, p, g, g_dtype, m, v, ema, p_half, lr, beta0, beta1, eps, step, ema_decay, zero_grad, 1.0f);
}
NGP_API int ngp_adam_ema_step_scaled(void *stream, uint64_t n, float *p, void *g, int g_dtype, float *m, float *v, float *ema, void *p_half,
                                     float lr, float beta0, float beta1, float eps, uint32_t step, float ema_decay, int zero_grad, float grad_mul) {
	NGP_REQUIRE(p && g && m && v && step >= 1, NGP_E_ARG, "ngp_adam_ema_step: bad arguments");
	NGP_REQUIRE(grad_mul == 1.0f || g_dtype == NGP_F16, NGP_E_ARG, "ngp_adam_ema_step: a gradient multiplier is only applied to fp16 (communication) gradients");
	NGP_REQUIRE(g_dtype == NGP_F32 || g_dtype == NGP_F16, NGP_E_DTYPE, "ngp_adam_ema_step: bad gradient dtype %d", g_dtype);
	NGP_REQUIRE(n % 4 == 0, NGP_E_ALIGN, "ngp_adam_ema_step: n (%llu) must be a multiple of 4", (unsigned long long)n);
	NGP_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema | (uintptr_t)p_half) & 15) == 0, NGP_E_ALIGN, "ngp_adam_ema_step: buffers must be 16-byte aligned");
	if (n == 0) return 0;
	const AdamConsts c = adam_consts(lr, beta0, beta1, eps, step, ema_decay, grad_mul);
	const uint64_t n4 = n / 4;
	uint32_t blocks = (uint32_t)((n4 + 255) / 256); if (blocks > 2048 * 4) blocks = 2048 * 4;
	hipStream_t s = (hipStream_t)stream;
#define GO(G, E, H, Z) NGP_LAUNCH((k_adam_ema<G, E, H, Z>), dim3(blocks), dim3(256), 0, s, n4, (float4 *)p, (G *)g, (float4 *)m, (float4 *)v, (float4 *)ema, (uint2 *)p_half, c)
#define GO_Z(G, E, H) do { if (zero_grad) GO(G, E, H, true); else GO(G, E, H, false); } while (0)
#define GO_H(G, E) do { if (p_half) GO_Z(G, E, true); else GO_Z(G, E, false); } while (0)
#define GO_E(G) do { if (ema == p) GO_H(G, 2); else if (ema) GO_H(G, 1); else GO_H(G, 0); } while (0)
	if (g_dtype == NGP_F32) GO_E(float); else GO_E(__half);
#undef GO
	NGP_LAUNCH_CHECK("ngp_adam_ema_step");
	return 0;
}

__global__ void k_generate_rays(uint32_t n, const int64_t *__restrict__ index, int W, int H, const float *__restrict__ focal, const float *__restrict__ meta,
                                const float *__restrict__ xforms, const float *__restrict__ images, const float *__restrict__ bg, int32_t *__restrict__ img_id,
                                float *__restrict__ rays_o, float *__restrict__ rays_d, float *__restrict__ target) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const int64_t hw = (int64_t)H * W;
	const int64_t id = index[i] / hw, off = index[i] % hw;
	const float *m = xforms + (size_t)id * 12;
	const float x = ((float)(off % W) + 0.5f) / W, y = ((float)(off / W) + 0.5f) / H;
	const float dc[3] = {(x - meta[id * 11 + 4]) * W / focal[2 * id], (y - meta[id * 11 + 5]) * H / focal[2 * id + 1], 1.0f};
	float d[3];
#pragma unroll
	for (int r = 0; r < 3; ++r) d[r] = m[r] * dc[0] + m[3 + r] * dc[1] + m[6 + r] * dc[2];
	const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
#pragma unroll
	for (int r = 0; r < 3; ++r) { rays_d[3 * (size_t)i + r] = d[r] / nrm; rays_o[3 * (size_t)i + r] = m[9 + r]; }
	img_id[i] = (int32_t)id;
	if (target) {
		const float4 px = reinterpret_cast<const float4 *>(images)[index[i]];
		target[3 * (size_t)i + 0] = px.x * px.w + bg[3 * (size_t)i + 0] * (1 - px.w);
		target[3 * (size_t)i + 1] = px.y * px.w + bg[3 * (size_t)i + 1] * (1 - px.w);
		target[3 * (size_t)i + 2] = px.z * px.w + bg[3 * (size_t)i + 2] * (1 - px.w);
	}
}
NGP_API int ngp_generate_rays(void *stream, uint32_t n, const int64_t *pixel_index, int W, int H, const float *focal, const float *metadata, const float *xforms,
                              const float *images, const float *bg, int32_t *img_id, float *rays_o, float *rays_d, float *target) {
	NGP_REQUIRE(pixel_index && focal && metadata && xforms && img_id && rays_o && rays_d, NGP_E_ARG, "ngp_generate_rays: null pointer");
	NGP_REQUIRE(!target || (images && bg), NGP_E_ARG, "ngp_generate_rays: target requested without images/bg");
	if (n == 0) return 0;
	NGP_LAUNCH(k_generate_rays, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, pixel_index, W, H, focal, metadata, xforms, images, bg, img_id, rays_o, rays_d, target);
	NGP_LAUNCH_CHECK("ngp_generate_rays");
	return 0;
}
