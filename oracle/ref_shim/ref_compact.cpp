// TEST INFRASTRUCTURE ONLY. Host build of compacted_coord (op_header/compacted_coord.h:4-76), launched as compacted_coord.py:39-64.
#define CONST_DT 0
#include "nerf_prelude.h"
#include "compacted_coord.h"
template <typename T>
static void run(uint32_t n_rays, float a0, float a1, uint32_t cap, const T *net, const float *coords_in, float *coords_out,
                const uint32_t *numsteps_in, uint32_t *counter, uint32_t *numsteps_out, uint32_t *rays_counter) {
	std::memset(coords_out, 0, (size_t)cap * 7 * sizeof(float));
	BoundingBox aabb(Eigen::Vector3f::Constant(a0), Eigen::Vector3f::Constant(a1));
	cpu_linear(compacted_coord<T>, n_rays, aabb, cap, 4, Array4f(1, 1, 1, 1), net, ENerfActivation(2), ENerfActivation(3),
	           (const NerfCoordinate *)coords_in, (NerfCoordinate *)coords_out, numsteps_in, counter, numsteps_out, rays_counter);
}
extern "C" {
__attribute__((visibility("default"))) void ref_compact_f32(uint32_t n, float a0, float a1, uint32_t cap, const float *net, const float *ci, float *co, const uint32_t *ni, uint32_t *cnt, uint32_t *no, uint32_t *rc) { run<float>(n, a0, a1, cap, net, ci, co, ni, cnt, no, rc); }
__attribute__((visibility("default"))) void ref_compact_f16(uint32_t n, float a0, float a1, uint32_t cap, const void *net, const float *ci, float *co, const uint32_t *ni, uint32_t *cnt, uint32_t *no, uint32_t *rc) { run<__half>(n, a0, a1, cap, (const __half *)net, ci, co, ni, cnt, no, rc); }
}
