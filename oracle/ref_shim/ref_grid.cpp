// TEST INFRASTRUCTURE ONLY. Host build of the reference's five density-grid op headers
// (density_grid_sampler/op_header/{mark_untrained_density_grid,generate_grid_samples_nerf_nonuniform,
//  splat_grid_samples_nerf_max_nearest_neighbor,ema_grid_samples_nerf,update_bitfield}.h). The headers have no include guards and
// redefine each other's types, so this file is compiled once per op with -DOP_<NAME> (as Jittor compiles one TU per jt.code site).
#define CONST_DT 0
#include "nerf_prelude.h"
#define EXP extern "C" __attribute__((visibility("default")))
#if defined(OP_MARK)
#include "mark_untrained_density_grid.h"
// launch: mark_untrained_density_grid.py:19-36 (output starts as zeros)
EXP void ref_grid_mark(uint32_t n_elements, float *grid, uint32_t n_images, const float *focal, const float *xforms, int W, int H) {
	std::memset(grid, 0, (size_t)n_elements * 4);
	cpu_linear(mark_untrained_density_grid, n_elements, grid, n_images, (const Vector2f *)focal, (const Matrix<float, 3, 4> *)xforms, Vector2i(W, H));
}
#elif defined(OP_GEN)
#include "generate_grid_samples_nerf_nonuniform.h"
// launch: generate_grid_samples_nerf_nonuniform.py:22-47 (host rng.advance() afterwards)
EXP void ref_grid_gen(uint32_t n, uint64_t *rng_state, uint32_t ema_step, float a0, float a1, const float *grid, float *pos, uint32_t *idx, uint32_t n_cascades, float thresh) {
	pcg32 rng; rng.state = rng_state[0]; rng.inc = rng_state[1];
	BoundingBox aabb{Vector3f::Constant(a0), Vector3f::Constant(a1)};
	cpu_linear(generate_grid_samples_nerf_nonuniform, n, rng, (const uint32_t *)&ema_step, aabb, grid, (NerfPosition *)pos, idx, n_cascades, thresh);
	rng.advance();
	rng_state[0] = rng.state; rng_state[1] = rng.inc;
}
#elif defined(OP_SPLAT)
#include "splat_grid_samples_nerf_max_nearest_neighbor.h"
// launch: splat_grid_samples_nerf_max_nearest_neighbor.py:19-41 (padded_output_width = 1, density_grid_sampler.py:52)
EXP void ref_grid_splat_f32(uint32_t n, const uint32_t *idx, const float *mlp, float *tmp) {
	cpu_linear(splat_grid_samples_nerf_max_nearest_neighbor<float>, n, idx, 1, mlp, tmp, ENerfActivation::Logistic, ENerfActivation::Exponential);
}
EXP void ref_grid_splat_f16(uint32_t n, const uint32_t *idx, const void *mlp, float *tmp) {
	cpu_linear(splat_grid_samples_nerf_max_nearest_neighbor<__half>, n, idx, 1, (const __half *)mlp, tmp, ENerfActivation::Logistic, ENerfActivation::Exponential);
}
#elif defined(OP_EMA)
#include "ema_grid_samples_nerf.h"
// launch: ema_grid_samples_nerf.py:17-28 (decay 0.95)
EXP void ref_grid_ema(uint32_t n, float decay, float *grid, const float *tmp) { cpu_linear(ema_grid_samples_nerf, n, decay, grid, tmp); }
#elif defined(OP_BITFIELD)
#include "update_bitfield.h"
// launch: update_bitfield.py:15-37. block_reduce is block-cooperative (shuffles + shared memory) and cannot run under a serial
// launcher, so reduce_sum is restated as what it computes: sum over cascade 0 of max(g,0)/128^3 (update_bitfield.py:25-28).
EXP void ref_grid_bitfield(const float *grid, float *mean /*[1]*/, uint8_t *bitfield) {
	const uint32_t n_elements = NERF_GRIDSIZE() * NERF_GRIDSIZE() * NERF_GRIDSIZE();
	// (double accumulator: the GPU's tree reduction is accurate to ~1e-7, a serial fp32 sum of 2 M nearly equal terms is off by up to 1 % - see oracle/ngp_oracle.c)
	double acc = 0.0;
	for (uint32_t i = 0; i < n_elements; ++i) acc += (double)(fmaxf(grid[i], 0.f) / (n_elements));
	mean[0] = (float)acc;
	cpu_linear(grid_to_bitfield, n_elements / 8 * NERF_CASCADES(), grid, bitfield, (const float *)mean);
	for (uint32_t level = 1; level < NERF_CASCADES(); ++level)
		cpu_linear(bitfield_max_pool, n_elements / 64, (const uint8_t *)(bitfield + grid_mip_offset(level - 1) / 8), bitfield + grid_mip_offset(level) / 8);
}
#endif
