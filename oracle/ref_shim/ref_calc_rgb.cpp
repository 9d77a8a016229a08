// TEST INFRASTRUCTURE ONLY. Host build of compute_rgbs / compute_rgbs_grad / compute_rgbs_inference
// (op_header/calc_rgb.h), launched as calc_rgb.py:45-68, 78-104, 120-144 (activations: rgb=Logistic(2), density=Exponential(3)).
#define CONST_DT 0
#include "nerf_prelude.h"
#include "calc_rgb.h"
template <typename T>
static void f_fwd(uint32_t n_rays, float a0, float a1, const T *net, const float *coords, const uint32_t *numsteps, float *rgb,
                  const uint32_t *numsteps_c, const float *bg) {
	BoundingBox aabb(Eigen::Vector3f::Constant(a0), Eigen::Vector3f::Constant(a1));
	cpu_linear(compute_rgbs<T>, n_rays, aabb, 4, net, ENerfActivation(2), ENerfActivation(3),
	           PitchedPtr<NerfCoordinate>((NerfCoordinate *)coords, 1, 0, 0), (uint32_t *)numsteps, (Array3f *)rgb, (uint32_t *)numsteps_c,
	           (const Array3f *)bg, (int)NERF_CASCADES(), MIN_CONE_STEPSIZE());
}
template <typename T>
static void f_bwd(uint32_t n_rays, uint32_t n_elems, float a0, float a1, T *dout, const T *net, const uint32_t *numsteps_c, const float *coords,
                  const float *loss_grad, const float *rgb_ray, const float *mean) {
	std::memset(dout, 0, (size_t)n_elems * 4 * sizeof(T));
	BoundingBox aabb(Eigen::Vector3f::Constant(a0), Eigen::Vector3f::Constant(a1));
	cpu_linear(compute_rgbs_grad<T>, n_rays, aabb, 4, dout, net, (uint32_t *)numsteps_c,
	           PitchedPtr<NerfCoordinate>((NerfCoordinate *)coords, 1, 0, 0), ENerfActivation(2), ENerfActivation(3), (Array3f *)loss_grad,
	           (Array3f *)rgb_ray, (float *)mean, (int)NERF_CASCADES(), MIN_CONE_STEPSIZE());
}
template <typename T>
static void f_inf(uint32_t n_rays, float a0, float a1, const T *net, const float *coords, const uint32_t *numsteps, float *rgb, float *alpha) {
	BoundingBox aabb(Eigen::Vector3f::Constant(a0), Eigen::Vector3f::Constant(a1));
	cpu_linear(compute_rgbs_inference<T>, n_rays, aabb, 4, Array3f(0, 0, 0), net, ENerfActivation(2), ENerfActivation(3),
	           PitchedPtr<NerfCoordinate>((NerfCoordinate *)coords, 1, 0, 0), (uint32_t *)numsteps, (Array3f *)rgb, (int)NERF_CASCADES(),
	           MIN_CONE_STEPSIZE(), alpha);
}
#define EXP extern "C" __attribute__((visibility("default")))
EXP void ref_rgb_fwd_f32(uint32_t n, float a0, float a1, const float *net, const float *c, const uint32_t *ns, float *rgb, const uint32_t *nsc, const float *bg) { f_fwd<float>(n, a0, a1, net, c, ns, rgb, nsc, bg); }
EXP void ref_rgb_fwd_f16(uint32_t n, float a0, float a1, const void *net, const float *c, const uint32_t *ns, float *rgb, const uint32_t *nsc, const float *bg) { f_fwd<__half>(n, a0, a1, (const __half *)net, c, ns, rgb, nsc, bg); }
EXP void ref_rgb_bwd_f32(uint32_t n, uint32_t ne, float a0, float a1, float *dout, const float *net, const uint32_t *nsc, const float *c, const float *lg, const float *rr, const float *mean) { f_bwd<float>(n, ne, a0, a1, dout, net, nsc, c, lg, rr, mean); }
EXP void ref_rgb_bwd_f16(uint32_t n, uint32_t ne, float a0, float a1, void *dout, const void *net, const uint32_t *nsc, const float *c, const float *lg, const float *rr, const float *mean) { f_bwd<__half>(n, ne, a0, a1, (__half *)dout, (const __half *)net, nsc, c, lg, rr, mean); }
EXP void ref_rgb_inf_f32(uint32_t n, float a0, float a1, const float *net, const float *c, const uint32_t *ns, float *rgb, float *alpha) { f_inf<float>(n, a0, a1, net, c, ns, rgb, alpha); }
EXP void ref_rgb_inf_f16(uint32_t n, float a0, float a1, const void *net, const float *c, const uint32_t *ns, float *rgb, float *alpha) { f_inf<__half>(n, a0, a1, (const __half *)net, c, ns, rgb, alpha); }
