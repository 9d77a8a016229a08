// TEST INFRASTRUCTURE ONLY. Host build of the reference's rays_sampler
// (density_grid_sampler/op_header/ray_sampler.h:4-114), launched as ray_sampler.py:34-62 does:
// memset of coords, lin128 over rays, host-side rng.advance() afterwards.
#include "nerf_prelude.h"
#include "ray_sampler.h"
extern "C" __attribute__((visibility("default")))
void REF_MARCH_NAME(uint32_t n_rays, float aabb0, float aabb1, uint32_t max_samples, const float *rays_o, const float *rays_d,
                    const uint8_t *bitfield, float cone_angle_constant, const float *metadata, const uint32_t *img_ids,
                    uint32_t *counters /*[2]: rays, samples*/, uint32_t *ray_indices, uint32_t *numsteps /*[n,2]*/,
                    float *coords /*[max_samples,7]*/, const float *xforms /*[n_img,4,3] == col-major 3x4*/, float near_distance,
                    uint64_t *rng_state /*[2]: state, inc; advanced by 2^32 on return*/) {
	pcg32 rng; rng.state = rng_state[0]; rng.inc = rng_state[1];
	std::memset(coords, 0, (size_t)max_samples * 7 * sizeof(float));
	BoundingBox aabb(Eigen::Vector3f::Constant(aabb0), Eigen::Vector3f::Constant(aabb1));
	cpu_linear(rays_sampler, n_rays, aabb, max_samples, (const Vector3f *)rays_o, (const Vector3f *)rays_d, bitfield, cone_angle_constant,
	           (const TrainingImageMetadata *)metadata, img_ids, counters, counters + 1, ray_indices, numsteps,
	           PitchedPtr<NerfCoordinate>((NerfCoordinate *)coords, 1, 0, 0), (const Matrix<float, 3, 4> *)xforms, near_distance, rng);
	rng.advance();
	rng_state[0] = rng.state; rng_state[1] = rng.inc;
}
