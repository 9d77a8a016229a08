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
#include <stdlib.h>
#pragma clang fp contract(off)

struct MarchParams {
	float a0, a1, near_distance, cone_angle;
	int const_dt, cascades;
	uint64_t rng_state, rng_inc;
	const int32_t *occ_bounds;        // device i32[cascades][6] from ngp_grid_occupied_bounds, or nullptr (no culling)
	int occ_cascades;                 // cascades 0 .. occ_cascades-1 can be selected for a candidate inside the scene box
	int occ_coarse;                   // the dilated coarse map behind the boxes may be used: only cascades 0 / 1 can be consulted and the scene box lies inside the unit cube
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
// the parameter t_target the marcher skips to when the cell at `pos` is empty (:728-748); the stepping itself is `do t += calc_dt(t) while (t < t_target)`
__device__ __forceinline__ float next_voxel_target(float t, const float pos[3], const float dir[3], const float idir[3], uint32_t res) {
	float t3[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		float q = res * pos[d];
		t3[d] = (floorf(q + 0.5f + 0.5f * copysignf(1.0f, dir[d])) - q) * idir[d];
	}
	float tt = fminf(fminf(t3[0], t3[1]), t3[2]);
	return t + fmaxf(tt / res, 0.0f);
}
__device__ __forceinline__ float advance_to_next_voxel(float t, const float pos[3], const float dir[3], const float idir[3], uint32_t res, const MarchParams &p) {   // :728-753
	const float t_target = next_voxel_target(t, pos, dir, idir, res);
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

// Occupied-bounds culling (r3), EXACT: a candidate emits a sample only if the cell it falls into at its mip level is occupied, i.e. only at positions inside that cascade's
// occupied box.  occ_range intersects the ray with every cascade's box, grown by one cell (plus the clamp: a box that touches the grid's border extends to infinity there,
// because positions beyond the grid are clamped into the border cells) - conservative against every rounding of pos = o + t d.  No intersection at or after t_start: the
// ray emits nothing, whatever the walk does.  Otherwise nothing is emitted behind t_stop, the last exit, so the traversal may end there.  (The stretch BEFORE the first box
// still has to be walked: where the skip chain lands at the box decides which candidates are visited inside.)
__device__ __forceinline__ bool occ_range(const MarchParams &p, const float o[3], const float d[3], float t_start, float &t_stop) {
	const float inf = __builtin_inff();
	bool any = false; t_stop = -inf;
	for (int c = 0; c < p.occ_cascades; ++c) {
		const int32_t *b = p.occ_bounds + c * 6;
		if (b[3] < b[0]) continue;                                            // empty cascade
		const float sc = scalbnf(1.0f, c), cell = sc / NGP_GRIDSIZE;
		float tmin = t_start, tmax = inf;
		bool hit = true;
#pragma unroll
		for (int k = 0; k < 3; ++k) {
			const float lo = b[k] <= 0 ? -inf : 0.5f + ((float)b[k] / NGP_GRIDSIZE - 0.5f) * sc - 1.001f * cell;
			const float hi = b[3 + k] >= (int)NGP_GRIDSIZE - 1 ? inf : 0.5f + ((float)(b[3 + k] + 1) / NGP_GRIDSIZE - 0.5f) * sc + 1.001f * cell;
			if (d[k] == 0.0f) { if (o[k] < lo || o[k] > hi) hit = false; continue; }
			float t0 = (lo - o[k]) / d[k], t1 = (hi - o[k]) / d[k];
			if (t0 > t1) { const float t = t0; t0 = t1; t1 = t; }
			// (an infinite bound gives +-inf, never NaN: o and d are finite and d[k] != 0)
			tmin = fmaxf(tmin, t0); tmax = fminf(tmax, t1);
		}
		if (hit && tmin <= tmax) { any = true; t_stop = fmaxf(t_stop, tmax); }
	}
	if (any) t_stop = t_stop + fabsf(t_stop) * 1e-5f + 1e-4f;               // slack for the rounding of the slab arithmetic itself (the boxes are already a cell too large)
	return any;
}

// Second, tighter stage of the culling for the wave-per-ray kernel when only cascades 0 / 1 can be consulted and the scene box lies inside the unit cube (ngp_base.py):
// the ray is sampled every h = 1/32 between the box entry and t_stop (64 lanes: one or two rounds), each sample looked up in the DILATED coarse map.  An emitting
// position p lies in an occupied coarse cell C; the sample nearest to it along the ray is at most h / 2 away (Euclidean, hence Chebyshev), so - clamped into the cube -
// it lies in C or one of C's 26 neighbours, which the dilation marked.  No sample in the map: no emission (exact).  Otherwise nothing is emitted more than h / 2 behind the
// last marked sample: t_stop shrinks to it (+ h).  The AABB test alone keeps every ray that passes the object's bounding box - about twice as many as meet the object.
__device__ __forceinline__ bool occ_coarse_range(const MarchParams &p, const float o[3], const float d[3], float t_start, float &t_stop, uint32_t lane) {
	const uint8_t *dil = reinterpret_cast<const uint8_t *>(p.occ_bounds + NGP_OCC_COARSE_OFFSET_INTS);
	{   // the map covers the unit cube only; a cascade-1 cell beyond it (never produced by the max-pool of cascade 0, but a loaded bitfield may hold anything) is consulted
		// by candidates exactly on the cube's faces: then this stage stands aside
		const int32_t *b1 = p.occ_bounds + 6;
		if (p.cascades > 1 && b1[3] >= b1[0] && (b1[0] < 32 || b1[1] < 32 || b1[2] < 32 || b1[3] > 95 || b1[4] > 95 || b1[5] > 95)) return true;
	}
	const float dn = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
	if (!(dn > 0.f) || !(t_stop < 3.0e38f)) return true;                     // (degenerate direction / unbounded range: keep the ray)
	const float h = (1.0f / NGP_OCC_COARSE) / dn * 0.999f;                    // spacing in t for a spatial spacing just under one coarse cell
	const float span = t_stop - t_start;
	const uint32_t n = (uint32_t)fminf(ceilf(span / h) + 2.0f, 4096.0f);
	float last = -__builtin_inff();
	for (uint32_t base = 0; base < n; base += 64u) {
		const float t = t_start + (float)(base + lane) * h;
		bool in = false;
		if (base + lane < n) {
			int c[3];
#pragma unroll
			for (int k = 0; k < 3; ++k) { const int q = (int)floorf((o[k] + t * d[k]) * NGP_OCC_COARSE); c[k] = min(max(q, 0), NGP_OCC_COARSE - 1); }
			in = dil[(c[2] * NGP_OCC_COARSE + c[1]) * NGP_OCC_COARSE + c[0]] != 0;
		}
		const unsigned long long m = __ballot(in);
		if (m) last = t_start + (float)(base + 63u - (uint32_t)__builtin_clzll(m)) * h;
	}
	if (last == -__builtin_inff()) return false;
	t_stop = fminf(t_stop, last + 2.0f * h + fabsf(last) * 1e-5f);
	return true;
}

// One traversal of a ray.  WRITE=false: count occupied steps (limit NERF_STEPS).  WRITE=true: emit the first `limit` records.
#define NGP_TCACHE NGP_STEPS   // per-ray cache [n_rays][NGP_STEPS] of the sample parameters t (only touched entries cost anything): the write pass never marches again

template <bool WRITE>
__device__ __forceinline__ uint32_t march(const MarchParams &p, const uint8_t *__restrict__ bitfield, const float o[3], const float d[3], float startt,
                                          uint32_t limit, float *__restrict__ out, float *__restrict__ tcache = nullptr, uint32_t tstride = 0, float t_stop = __builtin_inff()) {
	const float idir[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
	float wdir[3];
	if (WRITE) { wdir[0] = (d[0] + 1.0f) * 0.5f; wdir[1] = (d[1] + 1.0f) * 0.5f; wdir[2] = (d[2] + 1.0f) * 0.5f; }
	const float dtmin = min_cone_stepsize();
	const float dtspan = dtmin * (1 << (p.cascades - 1)) - dtmin;                   // warp_dt, ray_sampler_header.h:839-843
	uint32_t j = 0; float t = startt; float pos[3];
	for (;;) {
#pragma unroll
		for (int k = 0; k < 3; ++k) pos[k] = o[k] + t * d[k];
		if (!(contains(p, pos) && j < limit) || t > t_stop) break;                   // (t_stop: nothing is emitted behind the last occupied box, occ_range)
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
	float t_stop = __builtin_inff();
	if (p.occ_bounds && !occ_range(p, o, d, startt, t_stop)) { steps[i] = 0; return; }
	steps[i] = march<false>(p, bitfield, o, d, startt, NGP_STEPS, nullptr, startts ? startts + (size_t)i * NGP_TCACHE : nullptr, 1, t_stop);
}

// ---------------------------------------------------------------------------------------------------------------- wave-cooperative count pass
// One thread per ray (k_march_count above, kept for ngp_march_rays and for mostly-empty rays) is a chain of dependent loads per ray - hundreds of microseconds for
// the longest ray while the chip idles (8 k rays = 124 waves on 1024 SIMDs).  The cooperative pass rests on one observation about the reference's loop
// (ray_sampler.h:52-69, ray_sampler_header.h:728-753): whether a cell is occupied or not, t only ever advances by t += calc_dt(t).  The sequence t_0 = start,
// t_{k+1} = t_k + calc_dt(t_k) is therefore FIXED per ray, independent of the occupancy grid; the marcher visits a subsequence of it (occupied: emit, go to k+1;
// empty: go to the first m > k with !(t_m < t_target(k))).  One WAVEFRONT per ray works through that sequence NC = 256 candidates at a time:
//   chain : the next NC values of the sequence, in LDS.  With a CONSTANT step the recurrence t_{k+1} = fl(t_k + dt) has a closed form inside a binade: t = a*U
//           (U the binade's ulp, a an integer in [2^23, 2^24)), and as long as the exact sum stays below the binade's end every step adds the same integer
//           q = rint(dt/U) (the fractional part of dt/U is the same at every step, so every step rounds the same way; a tie disables the shortcut).  Lane i writes
//           (a0 + i*q)*U - exact integer arithmetic, bit-identical to the serial sum - and only the 1-3 steps around a binade crossing are real fp32 adds.
//           (Cone stepping has no closed form: lane 0 runs the recurrence.)
//   eval  : lane = candidate (position, box test, mip level, occupancy bit, skip target and - by a lower-bound search in the LDS chain - the candidate nx the
//           visit would continue at): one memory round trip per 64 candidates instead of one per visited candidate;
//   walk  : the visited candidates are the orbit of the start under c -> nx[c].  G[c] = "no earlier candidate of this round skips past c" (prefix-max of nx,
//           one wavefront scan per window) is a guess that is right unless float fuzz makes two candidates of one empty cell land differently; it is CHECKED
//           exactly - every member of G must continue at the next member of G - and if the check holds G IS the orbit (induction from the start).  Otherwise
//           lane 0 walks the links serially.  Either way the result is bit-identical to the serial traversal;
//   emit  : the emitting lanes store their t into the ray's t-cache at slot j + popcount(emitters below me) - the wavefront ballot / prefix-sum compaction of
//           north_star.
// No workgroup-level synchronisation at all: a ray's rounds depend on nothing but the ray.  History (ngp_base.py sampling, 7.8 k rays x 33 samples, alone on the
// GPU): serial 665 us; four rays per workgroup with one THREAD per ray for the chain and the walk and four barriers per round (round 2's first version) 169 us -
// those single-thread phases of the eight workgroups resident on a CU queued up behind each other; this kernel 94 us.
#define MC_WIN 64u           // candidates per wavefront-wide window

__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v, uint32_t lane) {
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(v, off); if (lane >= (uint32_t)off) v = max(v, y); }
	return v;
}
__device__ __forceinline__ unsigned long long bits_below(uint32_t n) { return n >= 64u ? ~0ull : ((1ull << n) - 1ull); }

template <uint32_t NW, bool CONST_DT>
__global__ __launch_bounds__(64) void k_march_wave(uint32_t n_rays, MarchParams p, const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                   const uint8_t *__restrict__ bitfield, uint32_t *__restrict__ steps, float *__restrict__ tcache) {
	constexpr uint32_t NC = NW * MC_WIN;                      // chain values per round
	__shared__ float tch[NC + 1];
	__shared__ uint32_t mlut[NGP_GRIDSIZE];
	__shared__ uint16_t wnext[NC];                            // slow path only: links, skip targets, masks, results
	__shared__ float wtarget[NC];
	__shared__ unsigned long long wmask[NW][4];
	__shared__ uint32_t wres[3];
	const uint32_t lane = threadIdx.x, i = blockIdx.x;
	if (i >= n_rays) return;
	mlut[lane] = expand_bits(lane); mlut[lane + 64u] = expand_bits(lane + 64u);
	float o[3], d[3], idir[3], hs[3];
#pragma unroll
	for (int k = 0; k < 3; ++k) { o[k] = rays_o[3 * (size_t)i + k]; d[k] = rays_d[3 * (size_t)i + k]; idir[k] = 1.0f / d[k]; hs[k] = 0.5f * copysignf(1.0f, d[k]); }
	float t_round = ray_start(p, i, o, d);
	{
		float pos[3];
#pragma unroll
		for (int k = 0; k < 3; ++k) pos[k] = o[k] + t_round * d[k];
		if (!contains(p, pos)) { if (lane == 0) steps[i] = 0; return; }     // the ray misses the box (or starts behind it): the reference's loop does not run
	}
	float t_stop = __builtin_inff();
	if (p.occ_bounds && !occ_range(p, o, d, t_round, t_stop)) { if (lane == 0) steps[i] = 0; return; }      // cannot meet an occupied cell: no samples (exact)
	if (p.occ_bounds && p.occ_coarse && !occ_coarse_range(p, o, d, t_round, t_stop, lane)) { if (lane == 0) steps[i] = 0; return; }
	const float big_neg = -__builtin_inff();
	uint32_t j0 = 0;
	float pend = big_neg;
	for (;;) {
		if (t_round > t_stop) { if (lane == 0) steps[i] = j0; return; }      // every remaining candidate lies behind the last occupied box: nothing more is emitted
		__syncthreads();                                      // (one wavefront: orders this round's LDS writes after the last round's reads; also publishes mlut)
		// ---- chain: tch[0 .. NC] = the next NC + 1 values of the ray's fixed sequence
		if (CONST_DT) {
			const float dtc = calc_dt(t_round, p);
			uint32_t k = 0; float t = t_round;
			while (k <= NC) {
				int e; (void)frexpf(t, &e);
				const float sc = ldexpf(1.0f, 24 - e), a0f = t * sc;             // t = a0 * 2^(e-24), a0 in [2^23, 2^24) for a normal positive t
				uint32_t m = 0, a0 = 0, q = 0;
				if (e > -100 && e < 100 && a0f >= 8388608.0f && a0f < 16777216.0f) {
					const float delta = dtc * sc;
					if (delta < 16777216.0f) {
						const float fl = floorf(delta);
						a0 = (uint32_t)a0f; q = (uint32_t)rintf(delta);
						const int X = (int)(16777215u - a0) - (int)(uint32_t)ceilf(delta);       // steps 0 .. X/q keep the exact sum below the binade's end
						if (delta - fl != 0.5f && X >= 0 && q > 0u) m = (uint32_t)X / q + 1u;
					}
				}
				if (m == 0u) { if (lane == 0) tch[k] = t; t += dtc; ++k; continue; }             // near a binade crossing (or a degenerate t): one real step
				const uint32_t last = min(m, NC - k);
				const float U = ldexpf(1.0f, e - 24);
				for (uint32_t c = lane; c <= last; c += 64u) tch[k + c] = (float)(a0 + c * q) * U;
				t = (float)(a0 + last * q) * U; k += last;
				if (k == NC) break;
			}
		} else if (lane == 0) {
			const float lo = min_cone_stepsize(), hi = max_cone_stepsize(p.cascades);
			float t = t_round;
#pragma unroll 8
			for (uint32_t k = 0; k <= NC; ++k) { tch[k] = t; t += clampf(t * p.cone_angle, lo, hi); }
		}
		__syncthreads();
		float tl[NW];
#pragma unroll
		for (uint32_t w = 0; w < NW; ++w) tl[w] = tch[w * MC_WIN + lane];
		t_round = tch[NC];
		// ---- where the round starts: candidate 0, or the landing of the skip that is still running
		uint32_t s = 0;
		if (pend != big_neg) {
			s = NC;
#pragma unroll
			for (int w = (int)NW - 1; w >= 0; --w) { const unsigned long long m = __ballot(!(tl[w] < pend)); if (m) s = (uint32_t)w * MC_WIN + (uint32_t)__builtin_ctzll(m); }
			if (s == NC) continue;                            // the whole round lies before the target
		}
		// ---- eval: lane = candidate, NW candidates per lane
		unsigned long long INm[NW], OCCm[NW];
		uint32_t nxw[NW]; float tgt[NW];
#pragma unroll
		for (uint32_t w = 0; w < NW; ++w) {
			const uint32_t self = w * MC_WIN + lane;
			nxw[w] = 0; tgt[w] = 0.f; INm[w] = 0ull; OCCm[w] = 0ull;
			if ((w + 1u) * MC_WIN <= s) continue;             // (wave-uniform) window before the start
			float pos[3];
#pragma unroll
			for (int k = 0; k < 3; ++k) pos[k] = o[k] + tl[w] * d[k];
			const bool inside = contains(p, pos);
			bool occ = false;
			float target = 0.f;
			uint32_t nx = self + 1u;
			if (inside) {
				const float dt = calc_dt(tl[w], p);
				const uint32_t mip = (uint32_t)mip_from_dt(dt, pos, p.cascades);
				const float mip_scale = scalbnf(1.0f, -(int)mip);                 // occupied_at (ray_sampler_header.h:755-776), morton code from the LDS table
				uint32_t c[3];
#pragma unroll
				for (int k = 0; k < 3; ++k) {
					float q = pos[k] - 0.5f; q *= mip_scale; q += 0.5f;
					const int ci = (int)(q * NGP_GRIDSIZE);
					c[k] = (uint32_t)min(max(ci, 0), (int)NGP_GRIDSIZE - 1);
				}
				const uint32_t idx = mlut[c[0]] | (mlut[c[1]] << 1) | (mlut[c[2]] << 2);
				occ = bitfield[idx / 8 + (NGP_GRIDSIZE * NGP_GRIDSIZE * NGP_GRIDSIZE * mip) / 8] & (1 << (idx % 8));
				if (!occ) {                                              // next_voxel_target with the per-ray constants (same expressions: x / 2^k == x * 2^-k exactly)
					const uint32_t res = NGP_GRIDSIZE >> mip;
					const float resf = (float)res, inv_res = scalbnf(1.0f, (int)mip - 7);      // 1 / res, exactly: res = 128 >> mip is a power of two
					float t3[3];
#pragma unroll
					for (int k = 0; k < 3; ++k) { const float q = resf * pos[k]; t3[k] = (floorf(q + 0.5f + hs[k]) - q) * idir[k]; }
					const float tt = fminf(fminf(t3[0], t3[1]), t3[2]);
					target = tl[w] + fmaxf(tt * inv_res, 0.0f);
					// the skip lands on the first candidate m > self with !(t_m < target): estimate from the local step, settle with LDS probes (NC = beyond this round)
					const float est = (target - tl[w]) * __frcp_rn(dt);                        // (only where the search below STARTS: it settles on the same candidate from any start)
					uint32_t g = self + 1u;
					if (est > 1.0f) g = est >= (float)NC ? NC : self + (uint32_t)est;
					if (g > NC) g = NC;
					while (g > self + 1u && !(tch[g - 1u] < target)) --g;
					while (g < NC && tch[g] < target) ++g;
					nx = g;
				}
			}
			INm[w] = __ballot(inside); OCCm[w] = __ballot(occ);
			nxw[w] = nx; tgt[w] = target;
		}
		// ---- walk, in parallel: G = candidates no earlier candidate of the round skips past
		unsigned long long G[NW];
		{
			uint32_t carry = 0;
#pragma unroll
			for (uint32_t w = 0; w < NW; ++w) {
				const uint32_t self = w * MC_WIN + lane;
				const uint32_t incl = wave_incl_max(self >= s ? nxw[w] : 0u, lane);
				const uint32_t up = __shfl_up(incl, 1);
				const uint32_t ex = max(carry, lane ? up : 0u);
				G[w] = __ballot(self >= s && (self == s || ex == self));
				carry = max(carry, (uint32_t)__builtin_amdgcn_readlane((int)incl, 63));
			}
		}
		uint32_t FO = NC;                                     // first visited candidate outside the box: the ray ends there
#pragma unroll
		for (int w = (int)NW - 1; w >= 0; --w) { const unsigned long long out = G[w] & ~INm[w]; if (out) FO = (uint32_t)w * MC_WIN + (uint32_t)__builtin_ctzll(out); }
		unsigned long long E[NW];
		bool bad = false;
		{
			uint32_t nfirst = NC;                             // first member of G in a later window
#pragma unroll
			for (int w = (int)NW - 1; w >= 0; --w) {
				const uint32_t self = (uint32_t)w * MC_WIN + lane;
				const unsigned long long V = G[w] & INm[w] & bits_below(FO > (uint32_t)w * MC_WIN ? FO - (uint32_t)w * MC_WIN : 0u);
				const unsigned long long rest = lane == 63u ? 0ull : G[w] >> (lane + 1u);
				const uint32_t ng = rest ? self + 1u + (uint32_t)__builtin_ctzll(rest) : nfirst;
				if (__ballot(((V >> lane) & 1ull) && ng != nxw[w])) bad = true;
				E[w] = V & OCCm[w];
				if (G[w]) nfirst = (uint32_t)w * MC_WIN + (uint32_t)__builtin_ctzll(G[w]);
			}
		}
		uint32_t emitted = 0, fin = 0;
		if (!bad) {
			uint32_t total = 0;
#pragma unroll
			for (uint32_t w = 0; w < NW; ++w) total += (uint32_t)__builtin_popcountll(E[w]);
			const uint32_t room = NGP_STEPS - j0;
			if (total >= room) {                              // NERF_STEPS is reached in this round: keep the first `room` emitters, the ray ends
				uint32_t cum = 0;
#pragma unroll
				for (uint32_t w = 0; w < NW; ++w) {
					const uint32_t c = (uint32_t)__builtin_popcountll(E[w]);
					if (cum + c > room) {
						const uint32_t pre = (uint32_t)__builtin_popcountll(E[w] & bits_below(lane));
						E[w] = __ballot(((E[w] >> lane) & 1ull) && cum + pre < room);
					}
					cum += c;
				}
				emitted = room; fin = 1; pend = big_neg;
			} else {
				emitted = total; fin = FO < NC ? 1u : 0u; pend = big_neg;
				if (!fin) {                                   // the last visited candidate: an empty cell there leaves its skip target pending
#pragma unroll
					for (int w = (int)NW - 1; w >= 0; --w) {
						if (G[w]) {
							const uint32_t ll = 63u - (uint32_t)__builtin_clzll(G[w]);
							if (!((OCCm[w] >> ll) & 1ull)) pend = __shfl(tgt[w], (int)ll);
							break;
						}
					}
				}
			}
		} else {
			// ---- slow path (float fuzz inside an empty cell): lane 0 follows the links
#pragma unroll
			for (uint32_t w = 0; w < NW; ++w) {
				const uint32_t self = w * MC_WIN + lane;
				const bool inside = (INm[w] >> lane) & 1ull, occ = (OCCm[w] >> lane) & 1ull;
				const unsigned long long NX = __ballot(inside && nxw[w] == self + 1u && (occ || nxw[w] < NC));
				wnext[self] = (uint16_t)nxw[w]; wtarget[self] = tgt[w];
				if (lane == 0) { wmask[w][0] = INm[w]; wmask[w][1] = OCCm[w]; wmask[w][2] = NX; wmask[w][3] = 0ull; }
			}
			__syncthreads();
			if (lane == 0) {
				uint32_t cur = s, em = 0, fn = 0;
				float new_pend = big_neg;
				while (cur < NC) {
					const uint32_t w = cur / MC_WIN, c = cur % MC_WIN;
					const unsigned long long IN = wmask[w][0], OCC = wmask[w][1], NX = wmask[w][2];
					if (!((IN >> c) & 1ull) || j0 + em >= NGP_STEPS) { fn = 1; break; }
					if ((NX >> c) & 1ull) {
						const unsigned long long rest = ~(NX >> c);
						uint32_t run = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;
						if (run > MC_WIN - c) run = MC_WIN - c;
						const unsigned long long runmask = bits_below(run) << c;
						unsigned long long emm = OCC & runmask;
						const uint32_t n_em = (uint32_t)__builtin_popcountll(emm), room = NGP_STEPS - (j0 + em);
						if (n_em > room) {
							unsigned long long keep = 0ull;
							for (uint32_t q = 0; q < room; ++q) { const unsigned long long low = emm & (0ull - emm); keep |= low; emm ^= low; }
							wmask[w][3] |= keep; em += room; fn = 1;
							break;
						}
						wmask[w][3] |= emm; em += n_em; cur += run;
					} else {
						const uint32_t nx = wnext[cur];
						if (nx >= NC) { new_pend = wtarget[cur]; break; }
						cur = nx;
					}
				}
				wres[0] = em; wres[1] = fn; wres[2] = __float_as_uint(new_pend);
			}
			__syncthreads();
#pragma unroll
			for (uint32_t w = 0; w < NW; ++w) E[w] = wmask[w][3];
			emitted = wres[0]; fin = wres[1]; pend = __uint_as_float(wres[2]);
		}
		// ---- emit: emitters store their t at slot j0 + (number of emitters before me): wavefront ballot / popcount compaction
		{
			uint32_t cum = j0;
#pragma unroll
			for (uint32_t w = 0; w < NW; ++w) {
				if (tcache && ((E[w] >> lane) & 1ull)) tcache[(size_t)i * NGP_TCACHE + cum + (uint32_t)__builtin_popcountll(E[w] & bits_below(lane))] = tl[w];
				cum += (uint32_t)__builtin_popcountll(E[w]);
			}
		}
		j0 += emitted;
		if (fin) { if (lane == 0) steps[i] = j0; return; }
	}
}

// Exclusive scans in ray order (n_rays <= 2^18 in this path):
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
// The three scans, multi-block (round 1's single 1024-thread workgroup took 78-99 us for ~40 k rays on a 256-CU chip): tiles of 2048 rays, one workgroup
// each, three short launches - tile totals of the step counts; tile totals of the (ok, ok-steps) pairs, which need the global bases; final values.
// Every launch re-derives what it needs from `steps` and the <= 128 tile totals (integer sums: the result does not depend on the tiling).
#define MS_TILE 2048u
__device__ __forceinline__ uint32_t tiles_before(const uint32_t *__restrict__ tot, uint32_t b, uint32_t *sh) {      // sum of tot[0..b)
	uint32_t total;
	(void)block_exclusive_scan_1024(threadIdx.x < b ? tot[threadIdx.x] : 0u, sh, total);
	return total;
}
__global__ __launch_bounds__(1024) void k_mscan_totals(uint32_t n_rays, const uint32_t *__restrict__ steps, uint32_t *__restrict__ tot1) {
	__shared__ uint32_t sh[17];
	const uint32_t i = blockIdx.x * MS_TILE + threadIdx.x * 2u;
	const uint32_t v = (i < n_rays ? steps[i] : 0u) + (i + 1 < n_rays ? steps[i + 1] : 0u);
	uint32_t total;
	(void)block_exclusive_scan_1024(v, sh, total);
	if (threadIdx.x == 0) tot1[blockIdx.x] = total;
}
__global__ __launch_bounds__(1024) void k_mscan_ok(uint32_t n_rays, uint32_t max_samples, const uint32_t *__restrict__ steps, const uint32_t *__restrict__ tot1,
                                                   uint32_t *__restrict__ tot2, uint32_t *__restrict__ tot3) {
	__shared__ uint32_t sh[17];
	const uint32_t i = blockIdx.x * MS_TILE + threadIdx.x * 2u;
	const uint32_t v0 = i < n_rays ? steps[i] : 0u, v1 = i + 1 < n_rays ? steps[i + 1] : 0u;
	const uint32_t before = tiles_before(tot1, blockIdx.x, sh);
	uint32_t total;
	const uint32_t base = before + block_exclusive_scan_1024(v0 + v1, sh, total);
	const bool ok0 = i < n_rays && base + v0 <= max_samples, ok1 = i + 1 < n_rays && base + v0 + v1 <= max_samples;
	uint32_t t2, t3;
	(void)block_exclusive_scan_1024((uint32_t)ok0 + (uint32_t)ok1, sh, t2);
	(void)block_exclusive_scan_1024((ok0 ? v0 : 0u) + (ok1 ? v1 : 0u), sh, t3);
	if (threadIdx.x == 0) { tot2[blockIdx.x] = t2; tot3[blockIdx.x] = t3; }
}
__global__ __launch_bounds__(1024) void k_mscan_final(uint32_t n_rays, uint32_t max_samples, uint32_t cap, const uint32_t *__restrict__ steps, const uint32_t *__restrict__ tot1,
                                                      const uint32_t *__restrict__ tot2, const uint32_t *__restrict__ tot3, uint32_t *__restrict__ numsteps,
                                                      uint32_t *__restrict__ numsteps_c, int32_t *__restrict__ ray_indices, uint32_t *__restrict__ counters, int n_counters) {
	__shared__ uint32_t sh[17];
	const uint32_t i = blockIdx.x * MS_TILE + threadIdx.x * 2u;
	const uint32_t v[2] = {i < n_rays ? steps[i] : 0u, i + 1 < n_rays ? steps[i + 1] : 0u};
	const uint32_t b1 = tiles_before(tot1, blockIdx.x, sh), b2 = tiles_before(tot2, blockIdx.x, sh), b3 = tiles_before(tot3, blockIdx.x, sh);
	uint32_t total, t2, t3;
	uint32_t base = b1 + block_exclusive_scan_1024(v[0] + v[1], sh, total);
	const bool ok[2] = {i < n_rays && base + v[0] <= max_samples, i + 1 < n_rays && base + v[0] + v[1] <= max_samples};
	uint32_t ridx = b2 + block_exclusive_scan_1024((uint32_t)ok[0] + (uint32_t)ok[1], sh, t2);
	uint32_t cbase = b3 + block_exclusive_scan_1024((ok[0] ? v[0] : 0u) + (ok[1] ? v[1] : 0u), sh, t3);
#pragma unroll
	for (int k = 0; k < 2; ++k) {
		const uint32_t r = i + k;
		const uint32_t sk = ok[k] ? v[k] : 0u;
		if (r < n_rays) {
			numsteps[2 * r] = sk; numsteps[2 * r + 1] = base;
			if (ray_indices) ray_indices[r] = !ok[k] ? 0 /*left untouched by the reference*/ : (sk == 0 ? -1 : (int32_t)ridx);
			if (numsteps_c) { numsteps_c[2 * r] = min(cap - min(cap, cbase), sk); numsteps_c[2 * r + 1] = cbase; }
		}
		base += v[k]; ridx += ok[k]; cbase += sk;
	}
	if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
		counters[0] = b2 + t2; counters[1] = b1 + total;
		if (n_counters == 4) { counters[2] = b3 + t3; counters[3] = min(b3 + t3, cap); }
	}
}
// tot: 3 x 128 u32 of scratch
static void launch_march_scan(hipStream_t s, uint32_t n_rays, uint32_t max_samples, uint32_t cap, const uint32_t *steps, uint32_t *tot, uint32_t *numsteps, uint32_t *numsteps_c,
                              int32_t *ray_indices, uint32_t *counters, int n_counters) {
	const uint32_t nb = div_up(n_rays, MS_TILE);
	NGP_LAUNCH(k_mscan_totals, dim3(nb), dim3(1024), 0, s, n_rays, steps, tot);
	NGP_LAUNCH(k_mscan_ok, dim3(nb), dim3(1024), 0, s, n_rays, max_samples, steps, (const uint32_t *)tot, tot + 128, tot + 256);
	NGP_LAUNCH(k_mscan_final, dim3(nb), dim3(1024), 0, s, n_rays, max_samples, cap, steps, (const uint32_t *)tot, (const uint32_t *)(tot + 128), (const uint32_t *)(tot + 256), numsteps,
	           numsteps_c, ray_indices, counters, n_counters);
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
	MarchParams p{a0, a1, near_distance, cone, const_dt, cascades, rng_state_host[0], rng_state_host[1], nullptr, cascades, 0};
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
	launch_march_scan(s, n_rays, max_samples, 0u, (const uint32_t *)scratch, scratch + n_rays, numsteps, (uint32_t *)nullptr, ray_indices, counters, 2);
	NGP_LAUNCH(k_march_write<false>, dim3(div_up(n_rays, 128)), dim3(128), 0, s, n_rays, p, rays_o, rays_d, bitfield, (const uint32_t *)numsteps, coords);
	NGP_LAUNCH_CHECK("ngp_march_rays");
	return 0;
}

NGP_API int ngp_march_rays_compacted_pos(void *stream, uint32_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *bitfield, float aabb0, float aabb1,
                                         float near_distance, float cone_angle, int const_dt, int cascades, uint64_t *rng_state_host, uint32_t max_samples,
                                         uint32_t cap, float *coords_out, uint32_t *numsteps, uint32_t *numsteps_compacted, uint32_t *counters, uint32_t *scratch, float *pos_out);
static int g_march_count_mode = -1;                                      // 0 = by samples per ray, 1 = serial, 2 = cooperative
NGP_API void ngp_x_march_count_mode(int mode) { g_march_count_mode = mode; }      // probe hook (not part of the ABI): tests run both count passes on the same rays
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
	return ngp_march_rays_compacted_bounds(stream, n_rays, rays_o, rays_d, bitfield, aabb0, aabb1, near_distance, cone_angle, const_dt, cascades, rng_state_host, max_samples,
	                                       cap, coords_out, numsteps, numsteps_compacted, counters, scratch, pos_out, nullptr);
}
NGP_API int ngp_march_rays_compacted_bounds(void *stream, uint32_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *bitfield, float aabb0, float aabb1,
                                            float near_distance, float cone_angle, int const_dt, int cascades, uint64_t *rng_state_host, uint32_t max_samples,
                                            uint32_t cap, float *coords_out, uint32_t *numsteps, uint32_t *numsteps_compacted, uint32_t *counters, uint32_t *scratch, float *pos_out,
                                            const int32_t *occ_bounds) {
	int rc = check_march_args("ngp_march_rays_compacted", n_rays, rays_o, rays_d, bitfield, coords_out, cascades); if (rc) return rc;
	NGP_REQUIRE(counters && rng_state_host && (n_rays == 0 || (numsteps && numsteps_compacted && scratch)), NGP_E_ARG, "ngp_march_rays_compacted: null pointer");
	hipStream_t s = (hipStream_t)stream;
	MarchParams p = make_params(aabb0, aabb1, near_distance, cone_angle, const_dt, cascades, rng_state_host);
	p.occ_bounds = getenv("NGP_MARCH_NO_BOUNDS") ? nullptr : occ_bounds;
	// Which cascades can a candidate select?  mip = max(mip_from_pos, mip_from_dt) (ray_sampler_header.h:60-77).  Candidates lie inside the scene box, so
	// mip_from_pos <= the value at the box's largest |coordinate - 0.5| (frexp is monotone); with the constant step mip_from_dt adds nothing (dt * 256 < 1).  The max-pooled
	// coarser cascades are occupied around the object too, and one-cell margins of THEIR cells would cover the whole unit cube - they must not enter the union when the
	// traversal can never consult them (ngp_base.py: cascades 0 and 1 only).  Cone stepping can raise the mip anywhere along the ray: all cascades.
	if (const_dt) {
		const float ext = fmaxf(fabsf(aabb0 - 0.5f), fabsf(aabb1 - 0.5f));
		int e; (void)frexpf(ext, &e);
		int top = e + 1; if (top < 1) top = 1;            // (>= 1: frexp(0) has exponent 0, so the exact centre of the grid selects cascade 1)
		if (top > cascades - 1) top = cascades - 1;
		p.occ_cascades = top + 1;
		p.occ_coarse = (top <= 1 && aabb0 >= 0.0f && aabb1 <= 1.0f && getenv("NGP_MARCH_NO_COARSE") == nullptr) ? 1 : 0;
	}
	if (n_rays == 0) { hipMemsetAsync(counters, 0, 16, s); return 0; }
	// scratch = steps[n_rays] | pad to 1024 | t-cache[NGP_TCACHE][n_rays]  (ngp_march_scratch_elems(n_rays) u32 elements)
	float *tcache = reinterpret_cast<float *>(scratch + ((n_rays + 1023u) & ~1023u));
	// Which count pass?  The cooperative one evaluates EVERY candidate of the ray's fixed t sequence (64 per wavefront instruction), the serial one only the
	// candidates the loop visits (one per empty cell + one per sample) but as a chain of dependent loads per ray.  Measured alone on the GPU, steady state
	// (tools/bench_march.py): ngp_base.py sampling (7.8 k rays x 33 samples, constant step, 9 candidates per cell) serial 800 us / cooperative 190 us;
	// ngp_fox.py sampling (34 k rays x 7 samples, cone stepping through 4 cascades, 2-4 candidates per cell) serial 370 us on a nearly idle chip /
	// cooperative 380-540 us on a busy one.  So: cooperative when rays carry many samples (>= 16 on average: also the first few hundred iterations of any
	// run and every inference chunk), serial when they are mostly empty space.  Both produce identical bits (tests/test_hip_parity.py runs both).
	if (g_march_count_mode < 0) { const char *e = getenv("NGP_MARCH_COUNT"); g_march_count_mode = !e ? 0 : (e[0] == 's' ? 1 : 2); }      // NGP_MARCH_COUNT=serial|coop overrides the choice
	const bool coop = g_march_count_mode ? g_march_count_mode == 2 : (uint64_t)cap >= (uint64_t)16 * n_rays;
	if (coop && const_dt) NGP_LAUNCH((k_march_wave<4u, true>), dim3(n_rays), dim3(64), 0, s, n_rays, p, rays_o, rays_d, bitfield, scratch, tcache);
	else if (coop) NGP_LAUNCH((k_march_wave<4u, false>), dim3(n_rays), dim3(64), 0, s, n_rays, p, rays_o, rays_d, bitfield, scratch, tcache);
	else NGP_LAUNCH(k_march_count, dim3(div_up(n_rays, 128)), dim3(128), 0, s, n_rays, p, rays_o, rays_d, bitfield, scratch, tcache);
	launch_march_scan(s, n_rays, max_samples, cap, (const uint32_t *)scratch, scratch + ((n_rays + 1023u) & ~1023u) + (size_t)NGP_TCACHE * n_rays, numsteps, numsteps_compacted, (int32_t *)nullptr, counters, 4);
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

// A group of lanes per ray (16 = four rays per wavefront when the adaptive ray count has settled at a handful of samples per ray, 64 = one wavefront per ray when
// rays are long).  The reference walks a ray's samples serially in one thread (calc_rgb.h:20-60) - tens of thousands of threads, each a chain of dependent,
// uncoalesced loads.  Here a ray's lanes load consecutive samples coalesced, evaluate the transcendental part (exp, logistic) in parallel, and obtain the
// recurrences by SCANS over the lanes: transmittance = carried T x prefix product of (1 - alpha), colour so far = prefix sum of weight x colour (log2 lanes shuffle
// steps per chunk of samples).  Round 1 replayed the recurrences in the reference's serial order to stay bit-identical; a ray with ~900 samples (through the object,
// constant step) then cost ~90 k dependent cycles and set the kernel's duration (106 + 83 us per lego iteration).  The scans change only the order of fp32
// products / sums: results agree with the serial order to ~1e-7 relative (tests: golden <= 2e-5).
template <uint32_t CG> __device__ __forceinline__ float bcast(float v, uint32_t k) { return __shfl(v, (int)k, (int)CG); }
template <> __device__ __forceinline__ float bcast<64>(float v, uint32_t k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)k)); }
constexpr uint32_t CG_TRAIN = 16, CG_INFER = 64;                   // lanes per ray
// inclusive scans / sum over the CG lanes of a ray (log2 CG shuffle steps)
template <uint32_t CG> __device__ __forceinline__ float group_scan_mul(float v, uint32_t lane) {
#pragma unroll
	for (uint32_t d = 1; d < CG; d <<= 1) { const float u = __shfl_up(v, d, (int)CG); if (lane >= d) v *= u; }
	return v;
}
template <uint32_t CG> __device__ __forceinline__ float group_scan_add(float v, uint32_t lane) {
#pragma unroll
	for (uint32_t d = 1; d < CG; d <<= 1) { const float u = __shfl_up(v, d, (int)CG); if (lane >= d) v += u; }
	return v;
}
template <uint32_t CG> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
	for (uint32_t d = CG / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, (int)CG);
	return v;
}

// Huber loss + its gradient of one ray's three channels (models/losses/huber_loss.py:6-14), the expressions of k_huber
__device__ __forceinline__ void huber3(const float *__restrict__ target, float delta, float *__restrict__ loss, float *__restrict__ grad, uint32_t i, uint32_t c, float x) {
	const float d = x - target[3 * i + c], rel = fabsf(d);
	if (loss) loss[3 * i + c] = rel > delta ? rel - 0.5f * delta : 0.5f / delta * rel * rel;
	grad[3 * i + c] = rel > delta ? (d > 0 ? 1.0f : -1.0f) : d / delta;
}
struct HuberArgs { const float *target; float delta; float *loss, *grad; };     // target == nullptr: no loss stage

template <typename T, bool INFERENCE, uint32_t CG /*lanes per ray*/>
__global__ __launch_bounds__(256) void k_composite_fwd(uint32_t n_rays, const T *__restrict__ net, const float *__restrict__ coords, const uint32_t *__restrict__ numsteps,
                                                       const uint32_t *__restrict__ numsteps_c, const float *__restrict__ bg, int cascades,
                                                       float *__restrict__ rgb_out, float *__restrict__ alpha_out, HuberArgs hub) {
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
		// transmittance in front of every sample of the chunk = carried T x exclusive product of (1 - alpha) over the lanes before it (inactive lanes hold 1)
		const float P = group_scan_mul<CG>(1.f - alpha, lane);
		float Pex = __shfl_up(P, 1, (int)CG); if (lane == 0) Pex = 1.f;
		const float weight = alpha * (T_ * Pex);
#pragma unroll
		for (int c = 0; c < 3; ++c) ray[c] += weight * rgb[c];            // per-lane partial sums over the chunks, folded once at the end
		T_ *= bcast<CG>(P, CG - 1u);
	}
#pragma unroll
	for (int c = 0; c < 3; ++c) ray[c] = group_sum<CG>(ray[c]);
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

template <typename T, uint32_t CG /*lanes per ray*/>
__global__ __launch_bounds__(256) void k_composite_bwd(uint32_t n_rays, const T *__restrict__ net, const float *__restrict__ coords, const uint32_t *__restrict__ numsteps_c,
                                                       const float *__restrict__ loss_grad, const float *__restrict__ rgb_ray, const float *__restrict__ density_grid_mean,
                                                       int cascades, T *__restrict__ dout) {
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
		// the recurrence's state right after this lane's sample, by scans over the ray's lanes: T after = carried T x inclusive product of (1 - alpha);
		// colour so far = carried sum + inclusive sum of weight x colour
		const float P = group_scan_mul<CG>(1.f - alpha, lane);
		float Pex = __shfl_up(P, 1, (int)CG); if (lane == 0) Pex = 1.f;
		const float my_w = alpha * (T_ * Pex), my_T = T_ * P;
		float my_r2[3];
#pragma unroll
		for (int c = 0; c < 3; ++c) { const float S = group_scan_add<CG>(my_w * rgb[c], lane); my_r2[c] = ray2[c] + S; ray2[c] += bcast<CG>(S, CG - 1u); }
		T_ *= bcast<CG>(P, CG - 1u);
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

// (r5) The training path's compositing in ONE launch: forward (k_composite_fwd's loop), Huber on the ray's colour, backward (k_composite_bwd's loop) - what
// ngp_composite_fwd_huber followed by ngp_composite_bwd compute, expression for expression: the forward leaves a ray's colour and loss gradient in EVERY lane of its group
// (the butterfly sum adds the same pairs in every lane), so the backward reads from registers what the two-launch sequence stores and reloads as the same fp32 values.
// rgb / loss / loss_grad are still written (the step returns the loss; the tests read all three).  Bit-identical to the two launches (tests/test_hip_parity.py).
template <typename T, uint32_t CG /*lanes per ray*/>
__global__ __launch_bounds__(256) void k_composite_train(uint32_t n_rays, const T *__restrict__ net, const float *__restrict__ coords, const uint32_t *__restrict__ numsteps,
                                                         const uint32_t *__restrict__ numsteps_c, const float *__restrict__ bg, int cascades, float *__restrict__ rgb_out,
                                                         HuberArgs hub, const float *__restrict__ density_grid_mean, T *__restrict__ dout) {
	const uint32_t lane = threadIdx.x & (CG - 1u), i = blockIdx.x * (256u / CG) + threadIdx.x / CG;
	if (i >= n_rays) return;
	const uint32_t ns = numsteps_c[2 * i], base = numsteps_c[2 * i + 1];
	if (ns == 0) {
		if (lane < 3) {
			const float v = bg[3 * i + lane];
			rgb_out[3 * i + lane] = v;
			huber3(hub.target, hub.delta, hub.loss, hub.grad, i, lane, v);
		}
		return;
	}
	// ---- forward: k_composite_fwd<T, false, CG>
	// (r6) the kernel's duration is its longest ray's (up to 1024 samples = 16 rounds of 64, forward and again backward): every round's loads are issued one round AHEAD,
	// so a round costs max(load, arithmetic) instead of their sum.  Same values, same arithmetic, same order - only the loads move.
	float T_ = 1.f, ray[3] = {0.f, 0.f, 0.f};
	float on[4] = {0.f, 0.f, 0.f, 0.f}, wn = 0.f;                       // the next round's network outputs and warped dt of this lane's sample
	auto fetch = [&](uint32_t c0) { if (c0 + lane < ns) { const size_t s = (size_t)base + c0 + lane; load4<T>(net + s * 4, on); wn = coords[s * 7 + 3]; } };
	fetch(0);
	for (uint32_t c0 = 0; c0 < ns; c0 += CG) {
		const uint32_t m = min(CG, ns - c0);
		float rgb[3] = {0.f, 0.f, 0.f}, alpha = 0.f;
		const float o[4] = {on[0], on[1], on[2], on[3]}, wdt = wn;
		if (c0 + CG < ns) fetch(c0 + CG);
		if (lane < m) {
			const float dt = unwarp_dt(wdt, cascades);
			const float density = __expf(o[3]);
			alpha = 1.f - __expf(-density * dt);
#pragma unroll
			for (int c = 0; c < 3; ++c) rgb[c] = logistic(o[c]);
		}
		const float P = group_scan_mul<CG>(1.f - alpha, lane);
		float Pex = __shfl_up(P, 1, (int)CG); if (lane == 0) Pex = 1.f;
		const float weight = alpha * (T_ * Pex);
#pragma unroll
		for (int c = 0; c < 3; ++c) ray[c] += weight * rgb[c];
		T_ *= bcast<CG>(P, CG - 1u);
	}
#pragma unroll
	for (int c = 0; c < 3; ++c) ray[c] = group_sum<CG>(ray[c]);
	if (ns == numsteps[2 * i]) {
#pragma unroll
		for (int c = 0; c < 3; ++c) ray[c] += T_ * bg[3 * i + c];
	}
	// ---- Huber (huber3's expressions), in every lane; lane 0 stores
	float G[3];
#pragma unroll
	for (int c = 0; c < 3; ++c) {
		const float d = ray[c] - hub.target[3 * i + c], rel = fabsf(d);
		G[c] = rel > hub.delta ? (d > 0 ? 1.0f : -1.0f) : d / hub.delta;
		if (lane == 0) {
			rgb_out[3 * i + c] = ray[c];
			if (hub.loss) hub.loss[3 * i + c] = rel > hub.delta ? rel - 0.5f * hub.delta : 0.5f / hub.delta * rel * rel;
			hub.grad[3 * i + c] = G[c];
		}
	}
	// ---- backward: k_composite_bwd<T, CG> with R = ray, G from above
	float loss_scale = 128; loss_scale /= n_rays;
	const float l1 = *density_grid_mean < 0.01f ? 1e-4f : 0.0f;
	T_ = 1.f;
	float ray2[3] = {0.f, 0.f, 0.f};
	if (ns > CG) fetch(0);                                              // (a ray of one round still holds its values from the forward: on / wn were not overwritten)
	for (uint32_t c0 = 0; c0 < ns; c0 += CG) {
		const uint32_t m = min(CG, ns - c0);
		const size_t s = (size_t)base + c0 + lane;
		float o[4] = {0.f, 0.f, 0.f, 0.f}, rgb[3] = {0.f, 0.f, 0.f}, alpha = 0.f, dt = 0.f;
		const float oc[4] = {on[0], on[1], on[2], on[3]}, wdt = wn;
		if (c0 + CG < ns) fetch(c0 + CG);
		if (lane < m) {
			o[0] = oc[0]; o[1] = oc[1]; o[2] = oc[2]; o[3] = oc[3];
#pragma unroll
			for (int c = 0; c < 3; ++c) rgb[c] = logistic(o[c]);
			dt = unwarp_dt(wdt, cascades);
			const float density = __expf(o[3]);
			alpha = 1.f - __expf(-density * dt);
		}
		const float P = group_scan_mul<CG>(1.f - alpha, lane);
		float Pex = __shfl_up(P, 1, (int)CG); if (lane == 0) Pex = 1.f;
		const float my_w = alpha * (T_ * Pex), my_T = T_ * P;
		float my_r2[3];
#pragma unroll
		for (int c = 0; c < 3; ++c) { const float S = group_scan_add<CG>(my_w * rgb[c], lane); my_r2[c] = ray2[c] + S; ray2[c] += bcast<CG>(S, CG - 1u); }
		T_ *= bcast<CG>(P, CG - 1u);
		if (lane < m) {
			float dl[4], dv[3];
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				const float suffix = ray[c] - my_r2[c];
				dl[c] = loss_scale * ((my_w * G[c]) * (rgb[c] * (1 - rgb[c])) + fmaxf(0.0f, 0.0f * o[c]));
				dv[c] = G[c] * (my_T * rgb[c] - suffix);
			}
			const float dotv = dv[0] + (dv[1] + dv[2]);
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
	hipStream_t s = (hipStream_t)stream;
	// lanes per ray: 16 when the adaptive ray count has settled at a handful of samples per ray (fox: ~7), a whole wavefront when rays are long (lego: ~33 samples,
	// the first iterations of any run: hundreds) - one coalesced load round trip per 64 samples instead of four.  Same arithmetic order either way.  The
	// training batch is 2^18 samples, so the ray count alone tells which regime this is.
	const bool wide = (uint64_t)n_rays * 24u <= (1u << 18);
	const uint32_t cg = wide ? CG_INFER : CG_TRAIN;
	const dim3 grid(div_up(n_rays, 256u / cg)), block(256);
#define CF_GO(T, W) NGP_LAUNCH((k_composite_fwd<T, false, W>), grid, block, 0, s, n_rays, (const T *)net, coords, numsteps, numsteps_c, bg, cascades, rgb_out, (float *)nullptr, hub)
	if (dtype == NGP_F32) { if (wide) CF_GO(float, CG_INFER); else CF_GO(float, CG_TRAIN); }
	else { if (wide) CF_GO(__half, CG_INFER); else CF_GO(__half, CG_TRAIN); }
#undef CF_GO
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
	if (dtype == NGP_F32) NGP_LAUNCH((k_composite_fwd<float, true, CG_INFER>), grid, block, 0, s, n_rays, (const float *)net, coords, numsteps, (const uint32_t *)nullptr, (const float *)nullptr, cascades, rgb_out, alpha_out, HuberArgs{nullptr, 0.f, nullptr, nullptr});
	else NGP_LAUNCH((k_composite_fwd<__half, true, CG_INFER>), grid, block, 0, s, n_rays, (const __half *)net, coords, numsteps, (const uint32_t *)nullptr, (const float *)nullptr, cascades, rgb_out, alpha_out, HuberArgs{nullptr, 0.f, nullptr, nullptr});
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
	const bool wide = (uint64_t)n_rays * 24u <= (uint64_t)n_elems;          // >= 24 samples per ray on average: a wavefront per ray (see composite_fwd_impl)
	const uint32_t cg = wide ? CG_INFER : CG_TRAIN;
	const dim3 grid(div_up(n_rays, 256u / cg)), block(256);
#define CB_GO(T, W) NGP_LAUNCH((k_composite_bwd<T, W>), grid, block, 0, s, n_rays, (const T *)net, coords, numsteps_c, loss_grad, rgb_ray, density_grid_mean, cascades, (T *)dout)
	if (dtype == NGP_F32) { if (wide) CB_GO(float, CG_INFER); else CB_GO(float, CG_TRAIN); }
	else { if (wide) CB_GO(__half, CG_INFER); else CB_GO(__half, CG_TRAIN); }
#undef CB_GO
	NGP_LAUNCH_CHECK("ngp_composite_bwd");
	return 0;
}

// ngp_composite_fwd_huber + ngp_composite_bwd (without zero_first) as one launch - the training step's form (csrc/train_step.hip); same arguments, same results bit for bit
NGP_API int ngp_composite_train(void *stream, uint32_t n_rays, uint32_t n_elems, const void *net, int dtype, const float *coords, const uint32_t *numsteps, const uint32_t *numsteps_c,
                                const float *bg, int cascades, float *rgb_out, const float *target, float delta, float *loss, float *loss_grad, const float *density_grid_mean, void *dout) {
	NGP_REQUIRE(net && coords && numsteps && numsteps_c && bg && rgb_out && target && loss_grad && density_grid_mean && dout, NGP_E_ARG, "ngp_composite_train: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_composite_train: bad dtype %d", dtype);
	if (n_rays == 0) return 0;
	hipStream_t s = (hipStream_t)stream;
	const bool wide = (uint64_t)n_rays * 24u <= (uint64_t)n_elems;          // >= 24 samples per ray on average: a wavefront per ray (see composite_fwd_impl)
	const HuberArgs hub{target, delta, loss, loss_grad};
	// (r6) the split forward picks its lanes per ray from the ray count alone (it has no n_elems), the backward from n_rays and n_elems.  One launch has one
	// width, and the width sets the order of the per-lane partial sums: where the two launches would disagree (n_elems != 2^18 and a ray count between the two
	// thresholds) the call IS the two launches, so "bit-identical to ngp_composite_fwd_huber + ngp_composite_bwd" holds for every argument.
	if (wide != ((uint64_t)n_rays * 24u <= (1u << 18))) {
		const int e = composite_fwd_impl(stream, n_rays, net, dtype, coords, numsteps, numsteps_c, bg, cascades, rgb_out, hub);
		return e ? e : ngp_composite_bwd(stream, n_rays, n_elems, net, dtype, coords, numsteps_c, loss_grad, rgb_out, density_grid_mean, cascades, dout, 0);
	}
	const uint32_t cg = wide ? CG_INFER : CG_TRAIN;
	const dim3 grid(div_up(n_rays, 256u / cg)), block(256);
#define CT_GO(T, W) NGP_LAUNCH((k_composite_train<T, W>), grid, block, 0, s, n_rays, (const T *)net, coords, numsteps, numsteps_c, bg, cascades, rgb_out, hub, density_grid_mean, (T *)dout)
	if (dtype == NGP_F32) { if (wide) CT_GO(float, CG_INFER); else CT_GO(float, CG_TRAIN); }
	else { if (wide) CT_GO(__half, CG_INFER); else CT_GO(__half, CG_TRAIN); }
#undef CT_GO
	NGP_LAUNCH_CHECK("ngp_composite_train");
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
