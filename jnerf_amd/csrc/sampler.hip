// Occupancy-grid ray marching, sample compaction, alpha compositing (fwd / bwd / inference), Huber loss.
//
// What it computes (reference, relative to python/jnerf/models/samplers/density_grid_sampler/):
//   rays_sampler            op_header/ray_sampler.h:4-114      (helpers op_header/ray_sampler_header.h)
//   compacted_coord         op_header/compacted_coord.h:4-76
//   compute_rgbs[_grad|_inference]   op_header/calc_rgb.h:10-212
//
// MI355X design: the reference reserves output slots with one global atomicAdd per ray, which makes the sample order (and everything
// downstream) run-to-run non-deterministic and forces a 117 MB memset + host read-back per call.  Here slot reservation is a
// deterministic exclusive scan in ray order (count pass -> scan -> write pass), all counters stay on the device, nothing is memset in
// the training path, and march + compaction are one pass (ngp_march_rays_compacted).  Per-ray arithmetic is kept in the reference's
// evaluation order with FMA contraction off, so sample counts and records are bit-identical to the reference's kernels.
#include "ngp_common.h"
#pragma clang fp contract(off)

struct MarchParams {
	float a0, a1, near_distance, cone_angle;
	int const_dt, cascades;
	uint64_t rng_state, rng_inc;
};

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); }
__device__ __forceinline__ float max_cone_stepsize(int cascades) { return min_cone_stepsize() * (1 << (cascades - 1)) * NGP_STEPS / NGP_GRIDSIZE; }
__device__ __forceinline__ float calc_dt(float t, const MarchParams &p) {          // density_grid_sampler.py:107-115
	if (p.const_dt) return min_cone_stepsize() * 0.5;
	return clampf(t * p.cone_angle, min_cone_stepsize(), max_cone_stepsize(p.cascades));
}
__device__ __forceinline__ int mip_from_pos(const float pos[3], int cascades) {    // ray_sampler_header.h:60-66
	int e;
	float m = fmaxf(fmaxf(fabsf(pos[0] - 0.5f), fabsf(pos[1] - 0.5f)), fabsf(pos[2] - 0.5f));
	frexpf(m, &e);
	return min(cascades - 1, max(0, e + 1));
}
__device__ __forceinline__ int mip_from_dt(float dt, const float pos[3], int cascades) {   // :68-77
	int mip = mip_from_pos(pos, cascades);
	dt *= 2 * NGP_GRIDSIZE;
	if (dt < 1.f) return mip;
	int e; frexpf(dt, &e);
	return min(cascades - 1, max(e, mip));
}
__device__ __forceinline__ bool occupied_at(const float pos[3], const uint8_t *__restrict__ bitfield, uint32_t mip) {   // :755-776
	const float mip_scale = scalbnf(1.0f, -(int)mip);
	uint32_t c[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		float q = pos[d] - 0.5f; q *= mip_scale; q += 0.5f;
		int i = (int)(q * NGP_GRIDSIZE);
		c[d] = (uint32_t)min(max(i, 0), (int)NGP_GRIDSIZE - 1);
	}
	const uint32_t idx = morton3D(c[0], c[1], c[2]);
	return bitfield[idx / 8 + (NGP_GRIDSIZE * NGP_GRIDSIZE * NGP_GRIDSIZE * mip) / 8] & (1 << (idx % 8));
}
__device__ __forceinline__ float advance_to_next_voxel(float t, const float pos[3], const float dir[3], const float idir[3], uint32_t res, const MarchParams &p) {   // :728-753
	float t3[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		float q = res * pos[d];
		t3[d] = (floorf(q + 0.5f + 0.5f * copysignf(1.0f, dir[d])) - q) * idir[d];
	}
	float tt = fminf(fminf(t3[0], t3[1]), t3[2]);
	float t_target = t + fmaxf(tt / res, 0.0f);
	do { t += calc_dt(t, p); } while (t < t_target);
	return t;
}
__device__ __forceinline__ bool contains(const MarchParams &p, const float q[3]) {
	return q[0] >= p.a0 && q[0] <= p.a1 && q[1] >= p.a0 && q[1] <= p.a1 && q[2] >= p.a0 && q[2] <= p.a1;
}
// BoundingBox::ray_intersect (:408-465) + near clamp + jittered start (ray_sampler.h:42-48)
__device__ __forceinline__ float ray_start(const MarchParams &p, uint32_t i, const float o[3], const float d[3]) {
	const float big = 3.402823466e+38f;
	float tmin = (p.a0 - o[0]) / d[0], tmax = (p.a1 - o[0]) / d[0];
	if (tmin > tmax) { float c = tmin; tmin = tmax; tmax = c; }
	float tymin = (p.a0 - o[1]) / d[1], tymax = (p.a1 - o[1]) / d[1];
	if (tymin > tymax) { float c = tymin; tymin = tymax; tymax = c; }
	bool miss = (tmin > tymax || tymin > tmax);
	if (!miss) {
		if (tymin > tmin) tmin = tymin;
		if (tymax < tmax) tmax = tymax;
		float tzmin = (p.a0 - o[2]) / d[2], tzmax = (p.a1 - o[2]) / d[2];
		if (tzmin > tzmax) { float c = tzmin; tzmin = tzmax; tzmax = c; }
		miss = (tmin > tzmax || tzmin > tmax);
		if (!miss && tzmin > tmin) tmin = tzmin;
	}
	if (miss) tmin = big;
	tmin = fmaxf(tmin, p.near_distance);
	Pcg32 rng{p.rng_state, p.rng_inc};
	rng.advance((uint64_t)(uint32_t)(i * 8u));                                    // N_MAX_RANDOM_SAMPLES_PER_RAY = 8 (ray_sampler_header.h:694)
	float startt = tmin;
	startt += calc_dt(startt, p) * rng.next_float();
	return startt;
}

// One traversal of a ray.  WRITE=false: count occupied steps (limit NERF_STEPS).  WRITE=true: emit the first `limit` records.
#define NGP_TCACHE NGP_STEPS   // per-ray cache [n_rays][NGP_STEPS] of the sample parameters t (only touched entries cost anything): the write pass never marches again

template <bool WRITE>
__device__ __forceinline__ uint32_t march(const MarchParams &p, const uint8_t *__restrict__ bitfield, const float o[3], const float d[3], float startt,
                                          uint32_t limit, float *__restrict__ out, float *__restrict__ tcache = nullptr, uint32_t tstride = 0) {
	const float idir[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
	float wdir[3];
	if (WRITE) { wdir[0] = (d[0] + 1.0f) * 0.5f; wdir[1] = (d[1] + 1.0f) * 0.5f; wdir[2] = (d[2] + 1.0f) * 0.5f; }
	const float dtmin = min_cone_stepsize();
	const float dtspan = dtmin * (1 << (p.cascades - 1)) - dtmin;                   // warp_dt, ray_sampler_header.h:839-843
	uint32_t j = 0; float t = startt; float pos[3];
	for (;;) {
#pragma unroll
		for (int k = 0; k < 3; ++k) pos[k] = o[k] + t * d[k];
		if (!(contains(p, pos) && j < limit)) break;
		const float dt = calc_dt(t, p);
		const uint32_t mip = (uint32_t)mip_from_dt(dt, pos, p.cascades);
		if (occupied_at(pos, bitfield, mip)) {
			if (WRITE) {
				float *c = out + (size_t)j * 7;
#pragma unroll
				for (int k = 0; k < 3; ++k) c[k] = (pos[k] - p.a0) / (p.a1 - p.a0);   // warp_position
				c[3] = (dt - dtmin) / dtspan;
				c[4] = wdir[0]; c[5] = wdir[1]; c[6] = wdir[2];
			} else if (tcache) tcache[j] = t;
			++j; t += dt;
		} else {
			t = advance_to_next_voxel(t, pos, d, idir, NGP_GRIDSIZE >> mip, p);
		}
	}
	return j;
}

__global__ __launch_bounds__(128) void k_march_count(uint32_t n_rays, MarchParams p, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                     const uint8_t *__restrict__ bitfield, uint32_t *__restrict__ steps, float *__restrict__ startts) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_rays) return;
	const float o[3] = {rays_o[3 * i], rays_o[3 * i + 1], rays_o[3 * i + 2]}, d[3] = {rays_d[3 * i], rays_d[3 * i + 1], rays_d[3 * i + 2]};
	const float startt = ray_start(p, i, o, d);
	steps[i] = march<false>(p, bitfield, o, d, startt, NGP_STEPS, nullptr, startts ? startts + (size_t)i * NGP_TCACHE : nullptr, 1);
}

// Single-workgroup exclusive scans in ray order (n_rays <= 2^18 in this path; 1024 threads x <=256 rays each).
//  base[i]  = sum_{k<i} steps[k]                       (what atomicAdd(numsteps_counter) yields under a serial launch, ray_sampler.h:73)
//  ok[i]    = base[i] + steps[i] <= max_samples        (:74-80; overflowed rays keep their reservation but get numsteps 0)
//  ridx[i]  = #ok rays before i                        (:84)
//  cbase[i] = sum_{k<i} (ok[k] ? steps[k] : 0)         (compacted_coord.h:62), cn[i] = min(cap - min(cap, cbase), steps)  (:63)
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t *sh /*[17]*/, uint32_t &total) {
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t x = v;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { uint32_t y = __shfl_up(x, off); if (lane >= (uint32_t)off) x += y; }
	if (lane == 63) sh[wave] = x;
	__syncthreads();
	if (threadIdx.x == 0) { uint32_t acc = 0; for (int w = 0; w < 16; ++w) { uint32_t tcur = sh[w]; sh[w] = acc; acc += tcur; } sh[16] = acc; }
	__syncthreads();
	const uint32_t res = sh[wave] + x - v;
	total = sh[16];
	__syncthreads();
	return res;
}
#define SCAN_TILE 8192u     // rays per tile: coalesced load -> LDS -> each thread scans 8 consecutive rays -> workgroup scan -> coalesced store
__global__ __launch_bounds__(1024) void k_march_scan(uint32_t n_rays, uint32_t max_samples, uint32_t cap, const uint32_t *__restrict__ steps,
                                                     uint32_t *__restrict__ numsteps, uint32_t *__restrict__ numsteps_c, int32_t *__restrict__ ray_indices,
                                                     uint32_t *__restrict__ counters, int n_counters) {
	__shared__ uint32_t sh[17];
	__shared__ uint32_t tile[SCAN_TILE + SCAN_TILE / 8];      // +1 word per 8 to dodge the 8-stride bank conflict
	uint32_t run_base = 0, run_ok = 0, run_sumok = 0;           // running totals of the three scans (wave-uniform, identical in every thread)
	for (uint32_t t0 = 0; t0 < n_rays; t0 += SCAN_TILE) {
		const uint32_t cnt = min(SCAN_TILE, n_rays - t0);
		for (uint32_t e = threadIdx.x; e < SCAN_TILE; e += 1024) tile[e + (e >> 3)] = e < cnt ? steps[t0 + e] : 0u;
		__syncthreads();
		uint32_t v[8], sum = 0;
#pragma unroll
		for (int k = 0; k < 8; ++k) { v[k] = tile[threadIdx.x * 9 + k]; sum += v[k]; }
		uint32_t total;
		uint32_t base = run_base + block_exclusive_scan_1024(sum, sh, total);
		uint32_t nok = 0, sumok = 0, okmask = 0, bases[8];
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const bool ok = (t0 + threadIdx.x * 8 + k < n_rays) && base + v[k] <= max_samples;
			bases[k] = base; okmask |= ok ? (1u << k) : 0u;
			nok += ok; sumok += ok ? v[k] : 0u;
			base += v[k];
		}
		uint32_t total_ok, total_sumok;
		uint32_t ridx = run_ok + block_exclusive_scan_1024(nok, sh, total_ok);
		uint32_t cbase = run_sumok + block_exclusive_scan_1024(sumok, sh, total_sumok);
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const uint32_t i = t0 + threadIdx.x * 8 + k;
			const bool ok = (okmask >> k) & 1u;
			const uint32_t sk = ok ? v[k] : 0u;
			if (i < n_rays) {
				numsteps[2 * i] = sk; numsteps[2 * i + 1] = bases[k];
				if (ray_indices) ray_indices[i] = !ok ? 0 /*left untouched by the reference*/ : (sk == 0 ? -1 : (int32_t)ridx);
				if (numsteps_c) { numsteps_c[2 * i] = min(cap - min(cap, cbase), sk); numsteps_c[2 * i + 1] = cbase; }
			}
			ridx += ok; cbase += sk;
		}
		run_base += total; run_ok += total_ok; run_sumok += total_sumok;
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		counters[0] = run_ok; counters[1] = run_base;
		if (n_counters == 4) { counters[2] = run_sumok; counters[3] = min(run_sumok, cap); }
	}
}

template <bool COMPACTED>
__global__ __launch_bounds__(128) void k_march_write(uint32_t n_rays, MarchParams p, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                     const uint8_t *__restrict__ bitfield, const uint32_t *__restrict__ numsteps, float *__restrict__ coords) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_rays) return;
	const uint32_t ns = numsteps[2 * i], base = numsteps[2 * i + 1];
	if (ns == 0) return;
	const float o[3] = {rays_o[3 * i], rays_o[3 * i + 1], rays_o[3 * i + 2]}, d[3] = {rays_d[3 * i], rays_d[3 * i + 1], rays_d[3 * i + 2]};
	const float startt = ray_start(p, i, o, d);
	march<true>(p, bitfield, o, d, startt, ns, coords + (size_t)base * 7);
}

// Write pass from the t-cache ([n_rays][NGP_STEPS]), one thread per SAMPLE: the 28-byte records of consecutive samples are consecutive in
// memory, so the stores are fully coalesced; the owning ray is found by binary search over the (monotonic) compacted bases.
// Every record is an independent function of (o, d, t): pos = o + t*d and dt = calc_dt(t) are the very expressions the marcher evaluates,
// so the records are bit-identical to a second traversal.
__global__ __launch_bounds__(256) void k_march_write_cached(uint32_t n_rays, MarchParams p, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                            const uint32_t *__restrict__ numsteps, const uint32_t *__restrict__ counters, uint32_t total_idx,
                                                            const float *__restrict__ tcache, float *__restrict__ coords, float *__restrict__ pos_out) {
	const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= counters[total_idx]) return;
	// last ray whose base <= s and that owns s (rays with zero steps share a base with their successor)
	uint32_t lo = 0, hi = n_rays - 1;
	while (lo < hi) {
		const uint32_t mid = (lo + hi + 1) >> 1;
		if (numsteps[2 * mid + 1] <= s) lo = mid; else hi = mid - 1;
	}
	uint32_t i = lo;
	while (numsteps[2 * i] == 0 || s - numsteps[2 * i + 1] >= numsteps[2 * i]) { if (i == 0) return; --i; }   // skip empty / truncated rays sharing the base
	const uint32_t j = s - numsteps[2 * i + 1];
	const float o[3] = {rays_o[3 * i], rays_o[3 * i + 1], rays_o[3 * i + 2]}, d[3] = {rays_d[3 * i], rays_d[3 * i + 1], rays_d[3 * i + 2]};
	const float t = tcache[(size_t)i * NGP_TCACHE + j];
	const float dt = calc_dt(t, p);
	const float dtmin = min_cone_stepsize();
	const float dtspan = dtmin * (1 << (p.cascades - 1)) - dtmin;
	float *c = coords + (size_t)s * 7;
#pragma unroll
	for (int k = 0; k < 3; ++k) c[k] = ((o[k] + t * d[k]) - p.a0) / (p.a1 - p.a0);
	c[3] = (dt - dtmin) / dtspan;
	c[4] = (d[0] + 1.0f) * 0.5f; c[5] = (d[1] + 1.0f) * 0.5f; c[6] = (d[2] + 1.0f) * 0.5f;
	if (pos_out) { pos_out[(size_t)s * 3] = c[0]; pos_out[(size_t)s * 3 + 1] = c[1]; pos_out[(size_t)s * 3 + 2] = c[2]; }   // compact [n,3] copy for the hash-grid kernels
}

static int check_march_args(const char *fn, uint32_t n_rays, const void *a, const void *b, const void *c, const void *d, int cascades) {
	NGP_REQUIRE(n_rays == 0 || (a && b && c && d), NGP_E_ARG, "%s: null pointer", fn);
	NGP_REQUIRE(cascades >= 1 && cascades <= 8, NGP_E_ARG, "%s: cascades %d out of range", fn, cascades);
	NGP_REQUIRE(n_rays <= (1u << 18), NGP_E_CAPACITY, "%s: n_rays %u exceeds 2^18", fn, n_rays);
	return 0;
}
static MarchParams make_params(float a0, float a1, float near_distance, float cone, int const_dt, int cascades, uint64_t *rng_state_host) {
	MarchParams p{a0, a1, near_distance, cone, const_dt, cascades, rng_state_host[0], rng_state_host[1]};
	Pcg32 r{rng_state_host[0], rng_state_host[1]};
	r.advance(1ull << 32);                                                          // host-side rng.advance(), ray_sampler.py:61
	rng_state_host[0] = r.state;
	return p;
}

NGP_API int ngp_march_rays(void *stream, uint32_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *bitfield, float aabb0, float aabb1,
                           float near_distance, float cone_angle, int const_dt, int cascades, uint64_t *rng_state_host, uint32_t max_samples,
                           float *coords, uint32_t *numsteps, uint32_t *counters, int32_t *ray_indices, uint32_t *scratch, int zero_coords) {
	int rc = check_march_args("ngp_march_rays", n_rays, rays_o, rays_d, bitfield, coords, cascades); if (rc) return rc;
	NGP_REQUIRE(counters && rng_state_host && (n_rays == 0 || (numsteps && scratch)), NGP_E_ARG, "ngp_march_rays: null pointer");
	hipStream_t s = (hipStream_t)stream;
	const MarchParams p = make_params(aabb0, aabb1, near_distance, cone_angle, const_dt, cascades, rng_state_host);
	if (zero_coords && coords) { hipError_t e = hipMemsetAsync(coords, 0, (size_t)max_samples * 28, s); if (e != hipSuccess) { ngp_set_error("ngp_march_rays memset: %s", hipGetErrorString(e)); return (int)e; } }
	if (n_rays == 0) { hipMemsetAsync(counters, 0, 8, s); return 0; }
	NGP_LAUNCH(k_march_count, dim3(div_up(n_rays, 128)), dim3(128), 0, s, n_rays, p, rays_o, rays_d, bitfield, scratch, (float *)nullptr);
	NGP_LAUNCH(k_march_scan, dim3(1), dim3(1024), 0, s, n_rays, max_samples, 0u, (const uint32_t *)scratch, numsteps, (uint32_t *)nullptr, ray_indices, counters, 2);
	NGP_LAUNCH(k_march_write<false>, dim3(div_up(n_rays, 128)), dim3(128), 0, s, n_rays, p, rays_o, rays_d, bitfield, (const uint32_t *)numsteps, coords);
	NGP_LAUNCH_CHECK("ngp_march_rays");
	return 0;
}

NGP_API int ngp_march_rays_compacted_pos(void *stream, uint32_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *bitfield, float aabb0, float aabb1,
                                         float near_distance, float cone_angle, int const_dt, int cascades, uint64_t *rng_state_host, uint32_t max_samples,
                                         uint32_t cap, float *coords_out, uint32_t *numsteps, uint32_t *numsteps_compacted, uint32_t *counters, uint32_t *scratch, float *pos_out);
NGP_API uint64_t ngp_march_scratch_elems(uint32_t n_rays) { return (uint64_t)((n_rays + 1023u) & ~1023u) + (uint64_t)NGP_TCACHE * n_rays + 1024u; }

NGP_API int ngp_march_rays_compacted(void *stream, uint32_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *bitfield, float aabb0, float aabb1,
                                     float near_distance, float cone_angle, int const_dt, int cascades, uint64_t *rng_state_host, uint32_t max_samples,
                                     uint32_t cap, float *coords_out, uint32_t *numsteps, uint32_t *numsteps_compacted, uint32_t *counters, uint32_t *scratch) {
	return ngp_march_rays_compacted_pos(stream, n_rays, rays_o, rays_d, bitfield, aabb0, aabb1, near_distance, cone_angle, const_dt, cascades, rng_state_host, max_samples,
	                                    cap, coords_out, numsteps, numsteps_compacted, counters, scratch, nullptr);
}
NGP_API int ngp_march_rays_compacted_pos(void *stream, uint32_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *bitfield, float aabb0, float aabb1,
                                         float near_distance, float cone_angle, int const_dt, int cascades, uint64_t *rng_state_host, uint32_t max_samples,
                                         uint32_t cap, float *coords_out, uint32_t *numsteps, uint32_t *numsteps_compacted, uint32_t *counters, uint32_t *scratch, float *pos_out) {
	int rc = check_march_args("ngp_march_rays_compacted", n_rays, rays_o, rays_d, bitfield, coords_out, cascades); if (rc) return rc;
	NGP_REQUIRE(counters && rng_state_host && (n_rays == 0 || (numsteps && numsteps_compacted && scratch)), NGP_E_ARG, "ngp_march_rays_compacted: null pointer");
	hipStream_t s = (hipStream_t)stream;
	const MarchParams p = make_params(aabb0, aabb1, near_distance, cone_angle, const_dt, cascades, rng_state_host);
	if (n_rays == 0) { hipMemsetAsync(counters, 0, 16, s); return 0; }
	// scratch = steps[n_rays] | pad to 1024 | t-cache[NGP_TCACHE][n_rays]  (ngp_march_scratch_elems(n_rays) u32 elements)
	float *tcache = reinterpret_cast<float *>(scratch + ((n_rays + 1023u) & ~1023u));
	NGP_LAUNCH(k_march_count, dim3(div_up(n_rays, 128)), dim3(128), 0, s, n_rays, p, rays_o, rays_d, bitfield, scratch, tcache);
	NGP_LAUNCH(k_march_scan, dim3(1), dim3(1024), 0, s, n_rays, max_samples, cap, (const uint32_t *)scratch, numsteps, numsteps_compacted, (int32_t *)nullptr, counters, 4);
	NGP_LAUNCH(k_march_write_cached, dim3(div_up(cap, 256)), dim3(256), 0, s, n_rays, p, rays_o, rays_d, (const uint32_t *)numsteps_compacted,
	                   (const uint32_t *)counters, 3u, (const float *)tcache, coords_out, pos_out);
	NGP_LAUNCH_CHECK("ngp_march_rays_compacted");
	return 0;
}

// ------------------------------------------------------------------ compaction (compacted_coord.h:4-76)
__global__ __launch_bounds__(1024) void k_compact_scan(uint32_t n_rays, uint32_t cap, const uint32_t *__restrict__ numsteps_in, uint32_t *__restrict__ numsteps_out,
                                                       uint32_t *__restrict__ counter) {
	__shared__ uint32_t sh[17];
	const uint32_t per = (n_rays + 1023u) / 1024u;
	const uint32_t lo = min(threadIdx.x * per, n_rays), hi = min(lo + per, n_rays);
	uint32_t sum = 0;
	for (uint32_t i = lo; i < hi; ++i) sum += numsteps_in[2 * i];
	uint32_t total;
	uint32_t cbase = block_exclusive_scan_1024(sum, sh, total);
	for (uint32_t i = lo; i < hi; ++i) {
		const uint32_t s = numsteps_in[2 * i];
		numsteps_out[2 * i] = min(cap - min(cap, cbase), s); numsteps_out[2 * i + 1] = cbase;
		cbase += s;
	}
	if (threadIdx.x == 0) counter[0] = total;
}
__global__ __launch_bounds__(256) void k_compact_copy(uint32_t n_rays, const float *__restrict__ coords_in, const uint32_t *__restrict__ numsteps_in,
                                                      const uint32_t *__restrict__ numsteps_out, float *__restrict__ coords_out) {
	// one wave per ray, lanes stride over the 7*cn floats of the record run (coalesced both ways)
	const uint32_t ray = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
	if (ray >= n_rays) return;
	const uint32_t cn = numsteps_out[2 * ray];
	if (cn == 0) return;
	const float *src = coords_in + (size_t)numsteps_in[2 * ray + 1] * 7;
	float *dst = coords_out + (size_t)numsteps_out[2 * ray + 1] * 7;
	for (uint32_t k = lane; k < cn * 7; k += 64) dst[k] = src[k];
}
NGP_API int ngp_compact_coords(void *stream, uint32_t n_rays, uint32_t cap, const float *coords_in, const uint32_t *numsteps_in, float *coords_out,
                               uint32_t *numsteps_out, uint32_t *counter, uint32_t *scratch) {
	NGP_REQUIRE(coords_in && numsteps_in && coords_out && numsteps_out && counter, NGP_E_ARG, "ngp_compact_coords: null pointer");
	NGP_REQUIRE(n_rays <= (1u << 18), NGP_E_CAPACITY, "ngp_compact_coords: n_rays %u exceeds 2^18", n_rays);
	hipStream_t s = (hipStream_t)stream;
	hipError_t e = hipMemsetAsync(coords_out, 0, (size_t)cap * 28, s);               // compacted_coord.py:38 zero-fills
	if (e != hipSuccess) { ngp_set_error("ngp_compact_coords memset: %s", hipGetErrorString(e)); return (int)e; }
	if (n_rays == 0) { hipMemsetAsync(counter, 0, 4, s); return 0; }
	NGP_LAUNCH(k_compact_scan, dim3(1), dim3(1024), 0, s, n_rays, cap, numsteps_in, numsteps_out, counter);
	NGP_LAUNCH(k_compact_copy, dim3(div_up(n_rays * 64, 256)), dim3(256), 0, s, n_rays, coords_in, numsteps_in, (const uint32_t *)numsteps_out, coords_out);
	NGP_LAUNCH_CHECK("ngp_compact_coords");
	(void)scratch;
	return 0;
}

// ------------------------------------------------------------------ compositing (calc_rgb.h)
template <typename T> __device__ __forceinline__ void load4(const T *p, float o[4]);
template <> __device__ __forceinline__ void load4<float>(const float *p, float o[4]) { float4 v = *reinterpret_cast<const float4 *>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <> __device__ __forceinline__ void load4<__half>(const __half *p, float o[4]) {
	uint2 raw = *reinterpret_cast<const uint2 *>(p);
	float2 a = __half22float2(*reinterpret_cast<__half2 *>(&raw.x)), b = __half22float2(*reinterpret_cast<__half2 *>(&raw.y));
	o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
}
template <typename T> __device__ __forceinline__ void store4(T *p, const float o[4]);
template <> __device__ __forceinline__ void store4<float>(float *p, const float o[4]) { *reinterpret_cast<float4 *>(p) = make_float4(o[0], o[1], o[2], o[3]); }
template <> __device__ __forceinline__ void store4<__half>(__half *p, const float o[4]) {
	__half2 a = __floats2half2_rn(o[0], o[1]), b = __floats2half2_rn(o[2], o[3]);
	uint2 raw; raw.x = *reinterpret_cast<uint32_t *>(&a); raw.y = *reinterpret_cast<uint32_t *>(&b);
	*reinterpret_cast<uint2 *>(p) = raw;
}
__device__ __forceinline__ float logistic(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float unwarp_dt(float dt, int cascades) {                 // calc_rgb.h:4-8
	float max_stepsize = min_cone_stepsize() * (1 << (cascades - 1));
	return dt * (max_stepsize - min_cone_stepsize()) + min_cone_stepsize();
}

// Sixteen lanes per ray (four rays per wavefront).  The reference walks a ray's samples serially in one thread (calc_rgb.h:20-60) - tens of thousands of threads,
// each a chain of dependent, uncoalesced loads.  Here a ray's lanes load 16 consecutive samples coalesced and evaluate the transcendental part (exp, logistic)
// in parallel; the transmittance / colour recurrences are then replayed in the reference's exact serial order (so results stay bit-identical) with the per-sample
// terms broadcast inside the 16-lane group (ds_bpermute).  The replay is ~13 VALU instructions per sample with no memory access in it.  16 rather than 64 lanes
// because the adaptive ray count settles at ~7 samples per ray: a 64-lane group would idle 90 % of its lanes.
// Inference chunks hold only rays that hit something (dozens of samples each) and keep one wavefront per ray.
template <uint32_t CG> __device__ __forceinline__ float bcast(float v, uint32_t k) { return __shfl(v, (int)k, (int)CG); }
template <> __device__ __forceinline__ float bcast<64>(float v, uint32_t k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)k)); }
constexpr uint32_t CG_TRAIN = 16, CG_INFER = 64;                   // lanes per ray

// Huber loss + its gradient of one ray's three channels (models/losses/huber_loss.py:6-14), the expressions of k_huber
__device__ __forceinline__ void huber3(const float *__restrict__ target, float delta, float *__restrict__ loss, float *__restrict__ grad, uint32_t i, uint32_t c, float x) {
	const float d = x - target[3 * i + c], rel = fabsf(d);
	if (loss) loss[3 * i + c] = rel > delta ? rel - 0.5f * delta : 0.5f / delta * rel * rel;
	grad[3 * i + c] = rel > delta ? (d > 0 ? 1.0f : -1.0f) : d / delta;
}
struct HuberArgs { const float *target; float delta; float *loss, *grad; };     // target == nullptr: no loss stage

template <typename T, bool INFERENCE>
__global__ __launch_bounds__(256) void k_composite_fwd(uint32_t n_rays, const T *__restrict__ net, const float *__restrict__ coords, const uint32_t *__restrict__ numsteps,
                                                       const uint32_t *__restrict__ numsteps_c, const float *__restrict__ bg, int cascades,
                                                       float *__restrict__ rgb_out, float *__restrict__ alpha_out, HuberArgs hub) {
	constexpr uint32_t CG = INFERENCE ? CG_INFER : CG_TRAIN;
	const uint32_t lane = threadIdx.x & (CG - 1u), i = blockIdx.x * (256u / CG) + threadIdx.x / CG;
	if (i >= n_rays) return;
	const uint32_t *nsrc = INFERENCE ? numsteps : numsteps_c;
	const uint32_t ns = nsrc[2 * i], base = nsrc[2 * i + 1];
	if (ns == 0) {
		if (lane < 3) {
			const float v = INFERENCE ? 0.f : bg[3 * i + lane];
			rgb_out[3 * i + lane] = v;
			if (!INFERENCE && hub.target) huber3(hub.target, hub.delta, hub.loss, hub.grad, i, lane, v);
		}
		if (INFERENCE && lane == 0) alpha_out[i] = 0.f;
		return;
	}
	float T_ = 1.f, ray[3] = {0.f, 0.f, 0.f};
	for (uint32_t c0 = 0; c0 < ns; c0 += CG) {                        // trip counts differ between the four rays of a wavefront; a ray's lanes stay together
		const uint32_t m = min(CG, ns - c0);
		float rgb[3] = {0.f, 0.f, 0.f}, alpha = 0.f;
		if (lane < m) {
			const size_t s = (size_t)base + c0 + lane;
			float o[4]; load4<T>(net + s * 4, o);
			const float dt = unwarp_dt(coords[s * 7 + 3], cascades);
			const float density = __expf(o[3]);
			alpha = 1.f - __expf(-density * dt);
#pragma unroll
			for (int c = 0; c < 3; ++c) rgb[c] = logistic(o[c]);
		}
#pragma unroll
		for (uint32_t k = 0; k < CG; ++k) {
			if (k >= m) break;
			const float a = bcast<CG>(alpha, k);
			const float weight = a * T_;
#pragma unroll
			for (int c = 0; c < 3; ++c) ray[c] += weight * bcast<CG>(rgb[c], k);
			T_ *= (1.f - a);
		}
	}
	if (!INFERENCE && ns == numsteps[2 * i]) {
#pragma unroll
		for (int c = 0; c < 3; ++c) ray[c] += T_ * bg[3 * i + c];
	}
	if (lane == 0) {
#pragma unroll
		for (int c = 0; c < 3; ++c) rgb_out[3 * i + c] = ray[c];
		if (INFERENCE) alpha_out[i] = 1 - T_;
		if (!INFERENCE && hub.target) {
#pragma unroll
			for (int c = 0; c < 3; ++c) huber3(hub.target, hub.delta, hub.loss, hub.grad, i, (uint32_t)c, ray[c]);
		}
	}
}

template <typename T>
__global__ __launch_bounds__(256) void k_composite_bwd(uint32_t n_rays, const T *__restrict__ net, const float *__restrict__ coords, const uint32_t *__restrict__ numsteps_c,
                                                       const float *__restrict__ loss_grad, const float *__restrict__ rgb_ray, const float *__restrict__ density_grid_mean,
                                                       int cascades, T *__restrict__ dout) {
	constexpr uint32_t CG = CG_TRAIN;
	const uint32_t lane = threadIdx.x & (CG - 1u), i = blockIdx.x * (256u / CG) + threadIdx.x / CG;
	if (i >= n_rays) return;
	float loss_scale = 128; loss_scale /= n_rays;                                    // calc_rgb.h:100-101
	const uint32_t ns = numsteps_c[2 * i], base = numsteps_c[2 * i + 1];
	if (ns == 0) return;
	const float l1 = *density_grid_mean < 0.01f ? 1e-4f : 0.0f;                       // :112
	const float G[3] = {loss_grad[3 * i], loss_grad[3 * i + 1], loss_grad[3 * i + 2]}, R[3] = {rgb_ray[3 * i], rgb_ray[3 * i + 1], rgb_ray[3 * i + 2]};
	float T_ = 1.f, ray2[3] = {0.f, 0.f, 0.f};
	for (uint32_t c0 = 0; c0 < ns; c0 += CG) {
		const uint32_t m = min(CG, ns - c0);
		const size_t s = (size_t)base + c0 + lane;
		float o[4] = {0.f, 0.f, 0.f, 0.f}, rgb[3] = {0.f, 0.f, 0.f}, alpha = 0.f, dt = 0.f;
		if (lane < m) {
			load4<T>(net + s * 4, o);
#pragma unroll
			for (int c = 0; c < 3; ++c) rgb[c] = logistic(o[c]);
			dt = unwarp_dt(coords[s * 7 + 3], cascades);
			const float density = __expf(o[3]);
			alpha = 1.f - __expf(-density * dt);
		}
		float my_w = 0.f, my_T = 0.f, my_r2[3] = {0.f, 0.f, 0.f};                      // the recurrence's state right after this lane's sample
#pragma unroll
		for (uint32_t k = 0; k < CG; ++k) {
			if (k >= m) break;
			const float a = bcast<CG>(alpha, k);
			const float weight = a * T_;
#pragma unroll
			for (int c = 0; c < 3; ++c) ray2[c] += weight * bcast<CG>(rgb[c], k);
			T_ *= (1.f - a);
			if (lane == k) { my_w = weight; my_T = T_; my_r2[0] = ray2[0]; my_r2[1] = ray2[1]; my_r2[2] = ray2[2]; }
		}
		if (lane < m) {
			float dl[4], dv[3];
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				const float suffix = R[c] - my_r2[c];
				dl[c] = loss_scale * ((my_w * G[c]) * (rgb[c] * (1 - rgb[c])) + fmaxf(0.0f, 0.0f * o[c]));
				dv[c] = G[c] * (my_T * rgb[c] - suffix);
			}
			const float dotv = dv[0] + (dv[1] + dv[2]);                                  // Eigen's 3-vector dot() order
			const float dd = __expf(clampf(o[3], -15.0f, 15.0f));
			dl[3] = loss_scale * (dd * (dt * dotv)) + (o[3] < 0 ? -l1 : 0.0f);
			store4<T>(dout + s * 4, dl);
		}
	}
}

static int composite_fwd_impl(void *stream, uint32_t n_rays, const void *net, int dtype, const float *coords, const uint32_t *numsteps,
                              const uint32_t *numsteps_c, const float *bg, int cascades, float *rgb_out, HuberArgs hub);
NGP_API int ngp_composite_fwd(void *stream, uint32_t n_rays, const void *net, int dtype, const float *coords, const uint32_t *numsteps,
                              const uint32_t *numsteps_c, const float *bg, int cascades, float *rgb_out) {
	return composite_fwd_impl(stream, n_rays, net, dtype, coords, numsteps, numsteps_c, bg, cascades, rgb_out, HuberArgs{nullptr, 0.f, nullptr, nullptr});
}
NGP_API int ngp_composite_fwd_huber(void *stream, uint32_t n_rays, const void *net, int dtype, const float *coords, const uint32_t *numsteps,
                                    const uint32_t *numsteps_c, const float *bg, int cascades, float *rgb_out, const float *target, float delta, float *loss, float *loss_grad) {
	NGP_REQUIRE(target && loss_grad, NGP_E_ARG, "ngp_composite_fwd_huber: null pointer");
	return composite_fwd_impl(stream, n_rays, net, dtype, coords, numsteps, numsteps_c, bg, cascades, rgb_out, HuberArgs{target, delta, loss, loss_grad});
}
static int composite_fwd_impl(void *stream, uint32_t n_rays, const void *net, int dtype, const float *coords, const uint32_t *numsteps,
                              const uint32_t *numsteps_c, const float *bg, int cascades, float *rgb_out, HuberArgs hub) {
	NGP_REQUIRE(net && coords && numsteps && numsteps_c && bg && rgb_out, NGP_E_ARG, "ngp_composite_fwd: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_composite_fwd: bad dtype %d", dtype);
	if (n_rays == 0) return 0;
	const dim3 grid(div_up(n_rays, 256u / CG_TRAIN)), block(256);
	hipStream_t s = (hipStream_t)stream;
	if (dtype == NGP_F32) NGP_LAUNCH((k_composite_fwd<float, false>), grid, block, 0, s, n_rays, (const float *)net, coords, numsteps, numsteps_c, bg, cascades, rgb_out, (float *)nullptr, hub);
	else NGP_LAUNCH((k_composite_fwd<__half, false>), grid, block, 0, s, n_rays, (const __half *)net, coords, numsteps, numsteps_c, bg, cascades, rgb_out, (float *)nullptr, hub);
	NGP_LAUNCH_CHECK("ngp_composite_fwd");
	return 0;
}
NGP_API int ngp_composite_inference(void *stream, uint32_t n_rays, const void *net, int dtype, const float *coords, const uint32_t *numsteps, int cascades,
                                    float *rgb_out, float *alpha_out) {
	NGP_REQUIRE(net && coords && numsteps && rgb_out && alpha_out, NGP_E_ARG, "ngp_composite_inference: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_composite_inference: bad dtype %d", dtype);
	if (n_rays == 0) return 0;
	const dim3 grid(div_up(n_rays, 256u / CG_INFER)), block(256);
	hipStream_t s = (hipStream_t)stream;
	if (dtype == NGP_F32) NGP_LAUNCH((k_composite_fwd<float, true>), grid, block, 0, s, n_rays, (const float *)net, coords, numsteps, (const uint32_t *)nullptr, (const float *)nullptr, cascades, rgb_out, alpha_out, HuberArgs{nullptr, 0.f, nullptr, nullptr});
	else NGP_LAUNCH((k_composite_fwd<__half, true>), grid, block, 0, s, n_rays, (const __half *)net, coords, numsteps, (const uint32_t *)nullptr, (const float *)nullptr, cascades, rgb_out, alpha_out, HuberArgs{nullptr, 0.f, nullptr, nullptr});
	NGP_LAUNCH_CHECK("ngp_composite_inference");
	return 0;
}
NGP_API int ngp_composite_bwd(void *stream, uint32_t n_rays, uint32_t n_elems, const void *net, int dtype, const float *coords, const uint32_t *numsteps_c,
                              const float *loss_grad, const float *rgb_ray, const float *density_grid_mean, int cascades, void *dout, int zero_first) {
	NGP_REQUIRE(net && coords && numsteps_c && loss_grad && rgb_ray && density_grid_mean && dout, NGP_E_ARG, "ngp_composite_bwd: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_composite_bwd: bad dtype %d", dtype);
	hipStream_t s = (hipStream_t)stream;
	if (zero_first) { hipError_t e = hipMemsetAsync(dout, 0, (size_t)n_elems * 4 * (dtype == NGP_F16 ? 2 : 4), s); if (e != hipSuccess) { ngp_set_error("ngp_composite_bwd memset: %s", hipGetErrorString(e)); return (int)e; } }
	if (n_rays == 0) return 0;
	const dim3 grid(div_up(n_rays, 256u / CG_TRAIN)), block(256);
	if (dtype == NGP_F32) NGP_LAUNCH((k_composite_bwd<float>), grid, block, 0, s, n_rays, (const float *)net, coords, numsteps_c, loss_grad, rgb_ray, density_grid_mean, cascades, (float *)dout);
	else NGP_LAUNCH((k_composite_bwd<__half>), grid, block, 0, s, n_rays, (const __half *)net, coords, numsteps_c, loss_grad, rgb_ray, density_grid_mean, cascades, (__half *)dout);
	NGP_LAUNCH_CHECK("ngp_composite_bwd");
	return 0;
}

// ------------------------------------------------------------------ Huber (models/losses/huber_loss.py:6-14)
__global__ void k_huber(uint32_t n, const float *__restrict__ x, const float *__restrict__ target, float delta, float *__restrict__ loss, float *__restrict__ grad) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float d = x[i] - target[i], rel = fabsf(d);
	if (loss) loss[i] = rel > delta ? rel - 0.5f * delta : 0.5f / delta * rel * rel;
	if (grad) grad[i] = rel > delta ? (d > 0 ? 1.0f : -1.0f) : d / delta;
}
NGP_API int ngp_huber(void *stream, uint32_t n, const float *x, const float *target, float delta, float *loss, float *grad) {
	NGP_REQUIRE(x && target && (loss || grad), NGP_E_ARG, "ngp_huber: null pointer");
	if (n == 0) return 0;
	NGP_LAUNCH(k_huber, dim3(div_up(n, 256)), dim3(256), 0, (hipStream_t)stream, n, x, target, delta, loss, grad);
	NGP_LAUNCH_CHECK("ngp_huber");
	return 0;
}
