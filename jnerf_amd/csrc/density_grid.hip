// Occupancy-grid maintenance (every update_den_freq = 16 training steps).
// Reference: samplers/density_grid_sampler/op_header/{mark_untrained_density_grid.h:3-47, generate_grid_samples_nerf_nonuniform.h:3-35,
// splat_grid_samples_nerf_max_nearest_neighbor.h:5-23, ema_grid_samples_nerf.h:3-25, update_bitfield.h:3-69}, orchestrated by
// density_grid_sampler.py:204-264.  The reference's mean uses a 32-lane-warp block_reduce (density_grid_sampler_header.h:310-406);
// here it is a wave64 reduction + one atomic per workgroup, and the four max-pool launches stay separate because each level
// depends on the previous one across workgroups.
#include "ngp_common.h"
#include <map>
#include <mutex>
#include <utility>
#pragma clang fp contract(off)

#define G3 (NGP_GRIDSIZE * NGP_GRIDSIZE * NGP_GRIDSIZE)

__global__ void k_grid_mark(uint32_t n_elements, float *__restrict__ grid, uint32_t n_images, const float *__restrict__ focal, const float *__restrict__ xforms, float half_resx, float half_resy) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_elements) return;
	const uint32_t level = i / G3, pos_idx = i % G3;
	const uint32_t x = morton3D_invert(pos_idx >> 0), y = morton3D_invert(pos_idx >> 1), z = morton3D_invert(pos_idx >> 2);
	const float sc = scalbnf(1.0f, (int)level);
	const float pos[3] = {(((float)x + 0.5f) / NGP_GRIDSIZE - 0.5f) * sc + 0.5f, (((float)y + 0.5f) / NGP_GRIDSIZE - 0.5f) * sc + 0.5f, (((float)z + 0.5f) / NGP_GRIDSIZE - 0.5f) * sc + 0.5f};
	const float voxel_radius = 0.5f * NGP_SQRT3 * sc / NGP_GRIDSIZE;
	int count = 0;
	for (uint32_t j = 0; j < n_images; ++j) {
		const float *m = xforms + (size_t)j * 12;                // column-major 3x4: col c at m[3c..3c+2]
		const float pl[3] = {pos[0] - m[9], pos[1] - m[10], pos[2] - m[11]};
		const float xx = pl[0] * m[0] + pl[1] * m[1] + pl[2] * m[2];
		const float yy = pl[0] * m[3] + pl[1] * m[4] + pl[2] * m[5];
		const float zz = pl[0] * m[6] + pl[1] * m[7] + pl[2] * m[8];
		if (zz > 0.f && fabsf(xx) - voxel_radius < zz / focal[2 * j] * half_resx && fabsf(yy) - voxel_radius < zz / focal[2 * j + 1] * half_resy) { count = 1; break; }
	}
	grid[i] = count > 0 ? 0.f : -1.f;                            // the reference starts from zeros (mark_untrained_density_grid.py:21), so this is the net effect of :43-46
}

// slot -> sample: with perm_mask = P - 1 (P a power of two dividing n, P <= 128^3) slot s holds sample i = (s & ~(P-1)) | ((s & (P-1)) * A^-1 mod P), A = 56924617:
// the cell the reference assigns to sample i (first try: ((i + step n) A + c) mod 128^3) then advances by ONE Morton index from slot to slot, so the points of
// neighbouring slots are neighbours in space and the 16-level gather that follows hits the same cache lines instead of 8 x 16 random ones per point.
// The multiset of (position, cell) pairs is exactly the reference's; only the order in the output arrays differs (the consumer is an atomic max per cell).
__global__ void k_grid_generate(uint32_t n, Pcg32 rng0, const uint32_t *__restrict__ step_p, float a0, float a1, const float *__restrict__ grid_in,
                                float *__restrict__ out, uint32_t *__restrict__ indices, uint32_t n_cascades, float thresh, uint32_t perm_mask) {
	const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
	if (slot >= n) return;
	const uint32_t i = perm_mask ? ((slot & ~perm_mask) | (((slot & perm_mask) * 53369u) & perm_mask)) : slot;          // 56924617 * 53369 == 1 (mod 2^21)
	Pcg32 rng = rng0;
	rng.advance((uint64_t)(uint32_t)(i * 4u));
	const uint32_t level = (uint32_t)(rng.next_float() * n_cascades) % n_cascades;
	const uint32_t step = *step_p;
	uint32_t idx = 0;
	for (uint32_t j = 0; j < 10; ++j) {
		idx = ((i + step * n) * 56924617u + j * 19349663u + 96925573u) % G3;
		idx += level * G3;
		if (grid_in[idx] > thresh) break;
	}
	const uint32_t pos_idx = idx % G3;
	const uint32_t x = morton3D_invert(pos_idx >> 0), y = morton3D_invert(pos_idx >> 1), z = morton3D_invert(pos_idx >> 2);
	const float r0 = rng.next_float(), r1 = rng.next_float(), r2 = rng.next_float();
	const float sc = scalbnf(1.0f, (int)level);
	const float pos[3] = {(((float)x + r0) / NGP_GRIDSIZE - 0.5f) * sc + 0.5f, (((float)y + r1) / NGP_GRIDSIZE - 0.5f) * sc + 0.5f, (((float)z + r2) / NGP_GRIDSIZE - 0.5f) * sc + 0.5f};
#pragma unroll
	for (int k = 0; k < 3; ++k) out[3 * (size_t)slot + k] = (pos[k] - a0) / (a1 - a0);
	indices[slot] = idx;
}

template <typename T>
__global__ void k_grid_splat(uint32_t n, const uint32_t *__restrict__ indices, const T *__restrict__ density, float *__restrict__ grid_tmp) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float mlp = __expf((float)density[i]);
	const float thick = mlp * min_cone_stepsize();
	atomicMax(reinterpret_cast<uint32_t *>(grid_tmp) + indices[i], __float_as_uint(thick));
}

__global__ void k_grid_ema(uint32_t n, float decay, float *__restrict__ grid, const float *__restrict__ tmp) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float prev = grid[i];
	grid[i] = (prev < 0.f) ? prev : fmaxf(prev * decay, tmp[i]);
}

// Mean of cascade 0 in two deterministic stages (r3): every workgroup writes its partial sum to a slot of its own, and every workgroup of the bitfield kernel adds the
// MEAN_PARTS partials in the same fixed order.  (Rounds 1-2 added the partials with one float atomic per workgroup: the sum then depended on the order the 2048 workgroups
// retired in, the threshold min(0.01, mean) moved by an ulp from run to run, now and then a cell of the bitfield flipped - and two runs of the same seed drifted apart.
// It was the only order-dependent float operation of the training path.)
#define MEAN_PARTS (G3 / 4 / 256)
__global__ __launch_bounds__(256) void k_grid_mean(const float *__restrict__ grid, float *__restrict__ partials) {
	__shared__ float sh[4];
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;        // G3/4 float4 elements, grid = G3/4/256 blocks
	const float4 v = reinterpret_cast<const float4 *>(grid)[i];
	float s = fmaxf(v.x, 0.f) / (G3) + fmaxf(v.y, 0.f) / (G3) + fmaxf(v.z, 0.f) / (G3) + fmaxf(v.w, 0.f) / (G3);
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
	__syncthreads();
	if (threadIdx.x == 0) partials[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void k_grid_to_bitfield(uint32_t n, const float *__restrict__ grid, uint8_t *__restrict__ bitfield, const float *__restrict__ partials, float *__restrict__ mean) {
	__shared__ float sh[4];
	float ps = 0.f;
#pragma unroll
	for (uint32_t k = 0; k < MEAN_PARTS / 256u; ++k) ps += partials[threadIdx.x + 256u * k];
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) ps += __shfl_xor(ps, off);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ps;
	__syncthreads();
	const float m = (sh[0] + sh[1]) + (sh[2] + sh[3]);
	if (blockIdx.x == 0 && threadIdx.x == 0) *mean = m;
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float thresh = 0.01f < m ? 0.01f : m;
	const float4 a = reinterpret_cast<const float4 *>(grid)[2 * (size_t)i], b = reinterpret_cast<const float4 *>(grid)[2 * (size_t)i + 1];
	uint8_t bits = 0;
	bits |= a.x > thresh ? 1 : 0; bits |= a.y > thresh ? 2 : 0; bits |= a.z > thresh ? 4 : 0; bits |= a.w > thresh ? 8 : 0;
	bits |= b.x > thresh ? 16 : 0; bits |= b.y > thresh ? 32 : 0; bits |= b.z > thresh ? 64 : 0; bits |= b.w > thresh ? 128 : 0;
	bitfield[i] = bits;
}
__global__ void k_bitfield_max_pool(uint32_t n, const uint8_t *__restrict__ prev, uint8_t *__restrict__ next) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint2 p = reinterpret_cast<const uint2 *>(prev)[i];
	uint8_t bits = 0;
#pragma unroll
	for (int j = 0; j < 4; ++j) { bits |= ((p.x >> (8 * j)) & 0xffu) ? (1u << j) : 0u; bits |= ((p.y >> (8 * j)) & 0xffu) ? (1u << (4 + j)) : 0u; }
	const uint32_t x = morton3D_invert(i >> 0) + NGP_GRIDSIZE / 8, y = morton3D_invert(i >> 1) + NGP_GRIDSIZE / 8, z = morton3D_invert(i >> 2) + NGP_GRIDSIZE / 8;
	next[morton3D(x, y, z)] |= bits;
}

NGP_API int ngp_grid_mark_untrained(void *stream, uint32_t n_elements, float *grid, uint32_t n_images, const float *focal, const float *xforms, int W, int H) {
	NGP_REQUIRE(grid && focal && xforms, NGP_E_ARG, "ngp_grid_mark_untrained: null pointer");
	if (n_elements == 0) return 0;
	NGP_LAUNCH(k_grid_mark, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, grid, n_images, focal, xforms, W * 0.5f, H * 0.5f);
	NGP_LAUNCH_CHECK("ngp_grid_mark_untrained");
	return 0;
}
static int grid_generate_impl(void *stream, uint32_t n, uint64_t *rng_state_host, const uint32_t *ema_step, float aabb0, float aabb1, const float *grid,
                              float *positions, uint32_t *indices, uint32_t n_cascades, float thresh, int morton_order);
NGP_API int ngp_grid_generate_samples(void *stream, uint32_t n, uint64_t *rng_state_host, const uint32_t *ema_step, float aabb0, float aabb1, const float *grid,
                                      float *positions, uint32_t *indices, uint32_t n_cascades, float thresh) {
	return grid_generate_impl(stream, n, rng_state_host, ema_step, aabb0, aabb1, grid, positions, indices, n_cascades, thresh, 0);
}
NGP_API int ngp_grid_generate_samples_ordered(void *stream, uint32_t n, uint64_t *rng_state_host, const uint32_t *ema_step, float aabb0, float aabb1, const float *grid,
                                              float *positions, uint32_t *indices, uint32_t n_cascades, float thresh, int morton_order) {
	return grid_generate_impl(stream, n, rng_state_host, ema_step, aabb0, aabb1, grid, positions, indices, n_cascades, thresh, morton_order);
}
static int grid_generate_impl(void *stream, uint32_t n, uint64_t *rng_state_host, const uint32_t *ema_step, float aabb0, float aabb1, const float *grid,
                              float *positions, uint32_t *indices, uint32_t n_cascades, float thresh, int morton_order) {
	NGP_REQUIRE(rng_state_host && ema_step && grid && positions && indices && n_cascades >= 1, NGP_E_ARG, "ngp_grid_generate_samples: bad arguments");
	Pcg32 rng{rng_state_host[0], rng_state_host[1]};
	Pcg32 adv = rng; adv.advance(1ull << 32); rng_state_host[0] = adv.state;          // generate_grid_samples_nerf_nonuniform.py:44
	if (n == 0) return 0;
	uint32_t perm_mask = 0;
	if (morton_order) { uint32_t P = n & (0u - n); if (P > G3) P = G3; perm_mask = P >= 65536u ? P - 1u : 0u; }      // largest power of two dividing n (<= 128^3); the refresh uses multiples of 2^19 - below 2^16 the order gains nothing
	NGP_LAUNCH(k_grid_generate, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, rng, ema_step, aabb0, aabb1, grid, positions, indices, n_cascades, thresh, perm_mask);
	NGP_LAUNCH_CHECK("ngp_grid_generate_samples");
	return 0;
}
NGP_API int ngp_grid_splat_max(void *stream, uint32_t n, const uint32_t *indices, const void *density, int dtype, float *grid_tmp) {
	NGP_REQUIRE(indices && density && grid_tmp, NGP_E_ARG, "ngp_grid_splat_max: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_grid_splat_max: bad dtype %d", dtype);
	if (n == 0) return 0;
	if (dtype == NGP_F32) NGP_LAUNCH(k_grid_splat<float>, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, indices, (const float *)density, grid_tmp);
	else NGP_LAUNCH(k_grid_splat<__half>, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, indices, (const __half *)density, grid_tmp);
	NGP_LAUNCH_CHECK("ngp_grid_splat_max");
	return 0;
}
NGP_API int ngp_grid_ema(void *stream, uint32_t n_elements, float decay, float *grid, const float *grid_tmp) {
	NGP_REQUIRE(grid && grid_tmp, NGP_E_ARG, "ngp_grid_ema: null pointer");
	if (n_elements == 0) return 0;
	NGP_LAUNCH(k_grid_ema, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, decay, grid, grid_tmp);
	NGP_LAUNCH_CHECK("ngp_grid_ema");
	return 0;
}
// ---------------------------------------------------------------------------------------------------------------- occupied bounds (r3)
// Per cascade, the integer bounding box of the occupied cells of the bitfield: bounds[c] = {min x, min y, min z, max x, max y, max z} in cell coordinates (min > max: the
// cascade is empty).  One byte of the Morton-ordered bitfield is a 2x2x2 block of cells; a non-zero byte counts as a whole block (conservative).  Integer atomicMin / Max:
// order-independent.  The marcher uses the boxes to drop rays that cannot meet an occupied cell and to stop a ray behind the last box (csrc/sampler.hip: occ_range) -
// in the training batches of an object-centred scene most rays only see background, and the cooperative marcher used to evaluate all ~1400 candidates of such a ray.
__global__ void k_occ_bounds_reset(int32_t *__restrict__ bounds, int cascades) {
	const int i = threadIdx.x;
	if (i < cascades * 6) bounds[i] = (i % 6) < 3 ? (int)NGP_GRIDSIZE : -1;
}
// (r5) 16 bytes (128 cells) per thread, reduced over the wavefront and then over the workgroup's four wavefronts in LDS: six atomics per WORKGROUP with an occupied cell -
// 64 workgroups per cascade.  Round 3's version issued them per wavefront (4096 per cascade): ~120 k same-address atomics, which the L2 retires one at a time - 72 us per
// refresh for ngp_base.py's bitfield, 585 us for ngp_fox.py's (profiles/r05a_realfox_kernel_trace.md: 7 % of that configuration's iteration).  min / max: same result.
__global__ __launch_bounds__(256) void k_occ_bounds(const uint8_t *__restrict__ bitfield, int32_t *__restrict__ bounds) {
	__shared__ int red[4][6];
	const uint32_t q = blockIdx.x * 256u + threadIdx.x;              // 16-byte group within the cascade blockIdx.y (G3 / 8 bytes each)
	const uint32_t c = blockIdx.y;
	int lo[3] = {(int)NGP_GRIDSIZE, (int)NGP_GRIDSIZE, (int)NGP_GRIDSIZE}, hi[3] = {-1, -1, -1};
	if (q < G3 / 128) {
		const uint4 v = reinterpret_cast<const uint4 *>(bitfield + (size_t)c * (G3 / 8))[q];
		const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
		for (uint32_t j = 0; j < 16; ++j) {
			if ((w[j >> 2] >> (8u * (j & 3u))) & 0xffu) {
				const uint32_t m = (q * 16u + j) * 8u;                   // Morton index of the byte's first cell: the 2 x 2 x 2 block at (x, y, z)
				const int x = (int)morton3D_invert(m), y = (int)morton3D_invert(m >> 1), z = (int)morton3D_invert(m >> 2);
				lo[0] = min(lo[0], x); lo[1] = min(lo[1], y); lo[2] = min(lo[2], z);
				hi[0] = max(hi[0], x + 1); hi[1] = max(hi[1], y + 1); hi[2] = max(hi[2], z + 1);
			}
		}
	}
#pragma unroll
	for (int k = 0; k < 3; ++k) {
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) { lo[k] = min(lo[k], __shfl_xor(lo[k], off)); hi[k] = max(hi[k], __shfl_xor(hi[k], off)); }
	}
	if ((threadIdx.x & 63u) == 0) {
#pragma unroll
		for (int k = 0; k < 3; ++k) { red[threadIdx.x >> 6][k] = lo[k]; red[threadIdx.x >> 6][3 + k] = hi[k]; }
	}
	__syncthreads();
	if (threadIdx.x < 6) {
		const int k = (int)threadIdx.x;
		int r = red[0][k];
#pragma unroll
		for (int wv = 1; wv < 4; ++wv) r = k < 3 ? min(r, red[wv][k]) : max(r, red[wv][k]);
		if (k < 3) { if (r < (int)NGP_GRIDSIZE) atomicMin(&bounds[c * 6 + k], r); }
		else if (r >= 0) atomicMax(&bounds[c * 6 + k], r);
	}
}
// Coarse map of the unit cube (NGP_OCC_COARSE^3 = 32^3 cells of 4^3 fine cells): a coarse cell is marked when any cascade-0 cell inside it, or any cascade-1 cell
// overlapping it, is occupied (those are the only cascades a constant-step traversal of a scene box inside the unit cube can consult, see sampler.hip); then the map is
// DILATED by one coarse cell.  A ray sampled every 1/32 that finds no sample in the dilated map cannot pass through an occupied cell (csrc/sampler.hip: occ_coarse_range).
// Both groups of 64 (cascade 0) / 8 (cascade 1) fine cells are contiguous in Morton order: one 8-byte / one 1-byte load per coarse cell.
__global__ __launch_bounds__(256) void k_occ_coarse(const uint8_t *__restrict__ bitfield, int cascades, uint8_t *__restrict__ raw) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;              // coarse cell, x fastest
	if (i >= NGP_OCC_COARSE * NGP_OCC_COARSE * NGP_OCC_COARSE) return;
	const uint32_t X = i % NGP_OCC_COARSE, Y = (i / NGP_OCC_COARSE) % NGP_OCC_COARSE, Z = i / (NGP_OCC_COARSE * NGP_OCC_COARSE);
	const uint2 b0 = reinterpret_cast<const uint2 *>(bitfield)[morton3D(X, Y, Z)];                       // cascade 0: Morton indices 64 m .. 64 m + 63 = the 4x4x4 block at (4X, 4Y, 4Z)
	bool occ = (b0.x | b0.y) != 0u;
	if (cascades > 1) occ = occ || bitfield[G3 / 8 + morton3D(16u + X, 16u + Y, 16u + Z)] != 0;        // cascade 1: the 2x2x2 block at (32 + 2X, ...), Morton index 8 m'
	raw[i] = occ ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_occ_dilate(const uint8_t *__restrict__ raw, uint8_t *__restrict__ dilated) {
	const int i = blockIdx.x * 256 + threadIdx.x, C = NGP_OCC_COARSE;
	if (i >= C * C * C) return;
	const int X = i % C, Y = (i / C) % C, Z = i / (C * C);
	uint8_t v = 0;
	for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
		const int x = X + dx, y = Y + dy, z = Z + dz;
		if (x >= 0 && x < C && y >= 0 && y < C && z >= 0 && z < C) v |= raw[(z * C + y) * C + x];
	}
	dilated[i] = v;
}
NGP_API int ngp_grid_occupied_bounds(void *stream, const uint8_t *bitfield, int cascades, int32_t *bounds) {
	NGP_REQUIRE(bitfield && bounds && cascades >= 1 && cascades <= 8, NGP_E_ARG, "ngp_grid_occupied_bounds: bad arguments");
	NGP_REQUIRE(((uintptr_t)bitfield & 15) == 0, NGP_E_ALIGN, "ngp_grid_occupied_bounds: bitfield must be 16-byte aligned");
	hipStream_t s = (hipStream_t)stream;
	NGP_LAUNCH(k_occ_bounds_reset, dim3(1), dim3(64), 0, s, bounds, cascades);
	NGP_LAUNCH(k_occ_bounds, dim3(div_up(G3 / 128, 256), cascades), dim3(256), 0, s, bitfield, bounds);
	uint8_t *dil = reinterpret_cast<uint8_t *>(bounds + NGP_OCC_COARSE_OFFSET_INTS), *raw = dil + NGP_OCC_COARSE * NGP_OCC_COARSE * NGP_OCC_COARSE;
	const uint32_t nc = NGP_OCC_COARSE * NGP_OCC_COARSE * NGP_OCC_COARSE;
	NGP_LAUNCH(k_occ_coarse, dim3(div_up(nc, 256)), dim3(256), 0, s, bitfield, cascades, raw);
	NGP_LAUNCH(k_occ_dilate, dim3(div_up(nc, 256)), dim3(256), 0, s, (const uint8_t *)raw, dil);
	NGP_LAUNCH_CHECK("ngp_grid_occupied_bounds");
	return 0;
}

NGP_API int ngp_grid_update_bitfield(void *stream, const float *grid, int cascades, float *mean, uint8_t *bitfield) {
	NGP_REQUIRE(grid && mean && bitfield && cascades >= 1 && cascades <= 8, NGP_E_ARG, "ngp_grid_update_bitfield: bad arguments");
	NGP_REQUIRE(((uintptr_t)grid & 15) == 0, NGP_E_ALIGN, "ngp_grid_update_bitfield: grid must be 16-byte aligned (update_bitfield.h:14-17)");
	hipStream_t s = (hipStream_t)stream;
	// the partial sums of the mean: 8 KiB of library-owned scratch per (device, stream) - launches on one stream are ordered, so one buffer per stream is enough
	static std::mutex mu;
	static std::map<std::pair<int, hipStream_t>, float *> pool;
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) { ngp_set_error("ngp_grid_update_bitfield: hipGetDevice failed"); return NGP_E_ARG; }
	float *partials;
	{
		std::lock_guard<std::mutex> lk(mu);
		float *&slot = pool[{dev, s}];
		if (!slot) { hipError_t e = hipMalloc((void **)&slot, (size_t)MEAN_PARTS * sizeof(float)); if (e != hipSuccess) { slot = nullptr; ngp_set_error("ngp_grid_update_bitfield: hipMalloc(mean partials): %s", hipGetErrorString(e)); return (int)e; } }
		partials = slot;
	}
	NGP_LAUNCH(k_grid_mean, dim3(MEAN_PARTS), dim3(256), 0, s, grid, partials);
	NGP_LAUNCH(k_grid_to_bitfield, dim3(div_up(G3 / 8 * cascades, 256)), dim3(256), 0, s, G3 / 8 * (uint32_t)cascades, grid, bitfield, (const float *)partials, mean);
	for (int level = 1; level < cascades; ++level)
		NGP_LAUNCH(k_bitfield_max_pool, dim3(div_up(G3 / 64, 256)), dim3(256), 0, s, G3 / 64, (const uint8_t *)(bitfield + (size_t)G3 * (level - 1) / 8), bitfield + (size_t)G3 * level / 8);
	NGP_LAUNCH_CHECK("ngp_grid_update_bitfield");
	return 0;
}
