// Multiresolution hash-grid encode, forward + backward, for gfx950.
//
// What it computes: position_encoders/hash_encoder/op_header/HashEncode.h:117-203 (kernel_grid) and :299-396
// (kernel_grid_backward) of the reference, with the level table (offset/size/resolution/scale) precomputed on the host.
//
// MI355X design (not the reference's launch shape):
//  * one thread = one (sample, level); a workgroup works on ONE level so its gathers stay inside that level's slice of the
//    table (<= 2 MiB fp16 / 4 MiB fp32) ...
//  * ... and the blockIdx -> (level, chunk) map is XCD-aware: the dispatcher places block b on XCD b%8 and each XCD has a
//    private 4 MiB L2, so XCD x is handed level 15-x for ALL sample chunks first and level x afterwards.  A level's table
//    slice is then fetched from HBM/Infinity-Cache once per XCD and served out of that XCD's L2 for the rest of the pass
//    instead of being bounced between eight L2s.  This mapping affects speed only, never results.
//  * no extract_position / transpose kernels: positions are read strided from the caller's buffer and features are written
//    either as the [n,32] rows HashEncoder returns or as a level-major [16][n] stream of pairs that the fused MLP consumes with
//    fully coalesced 256-B wave accesses (no 4-byte-per-64-byte-line partial writes from sixteen different XCDs).
//  * backward: chosen by what the caller hands over (hash_bwd_impl).  With a workspace (the training path): a binned scatter without any float atomic - every
//    contribution computed once, written as a record for the bin its entry lives in, summed per bin in 64-bit integer LDS accumulators, coarse levels with per-thread run
//    combining; bit-reproducible.  fp32 dL/dy (ngp_base.py), round 4: record REGIONS - no global atomic of any kind, one 16-byte record per cell edge on the fine levels, one
//    accumulate kernel for all levels (k_bin_runs2 / k_bin_pairs / k_bin_accumulate2).  fp16 dL/dy (ngp_fox.py): round 2/3's per-corner record lists with cursor
//    reservations (k_bin_records_runs / k_bin_records / k_bin_accumulate).  Without a workspace: an owner-computes scan (a workgroup owns a slice of a level in LDS and
//    filters the sample stream).  NGP_HASH_BWD_ATOMICS=1 or a non-power-of-two hashed table: the reference's scheme, one global float atomic per corner.
#include "ngp_common.h"
#include <stdlib.h>
#include <string.h>
#pragma clang fp contract(off)

template <typename T> struct Pair;
template <> struct Pair<float> { using type = float2; };
template <> struct Pair<__half> { using type = __half2; };

__device__ __forceinline__ float2 to_f2(float2 v) { return v; }
__device__ __forceinline__ float2 to_f2(__half2 v) { return __half22float2(v); }
__device__ __forceinline__ void from_f2(float2 &o, float2 v) { o = v; }
__device__ __forceinline__ void from_f2(__half2 &o, float2 v) { o = __floats2half2_rn(v.x, v.y); }

// HashEncode.h:68-94 with get_index(p0,p1,p2) = p0 ^ p1*19349663 ^ p2*83492791 (projects/ngp/configs/ngp_base.py:69)
__device__ __forceinline__ uint32_t grid_index(uint32_t size, uint32_t res, bool dense, uint32_t gx, uint32_t gy, uint32_t gz) {
	uint32_t index = dense ? gx + gy * res + gz * res * res : (gx ^ gy * 19349663u ^ gz * 83492791u);
	if ((size & (size - 1)) == 0) return index & (size - 1);   // hashed levels are 2^19 entries
	if (index >= size) { index -= size; if (index >= size) index %= size; }   // dense levels wrap only at the +1 boundary corner, and then by < size (res(1+res+res^2) < 2 res^3): the division is never executed for in-range positions
	return index;
}
// the reference decides "dense" by letting the stride loop run while stride <= size (HashEncode.h:82-91)
__device__ __forceinline__ bool level_is_dense(uint32_t size, uint32_t res) {
	uint32_t stride = 1;
#pragma unroll
	for (int d = 0; d < 3; ++d) if (stride <= size) stride *= res;
	return !(size < stride);
}

__device__ __forceinline__ void block_to_level_chunk(uint32_t nblk, uint32_t &level, uint32_t &chunk) {
	const uint32_t b = blockIdx.x, xcd = b & 7u, slot = b >> 3;
	const uint32_t phase = slot / nblk;
	chunk = slot - phase * nblk;
	level = phase == 0 ? 15u - xcd : xcd;
}

static LevelTable load_table(const uint32_t *host) { LevelTable lt; for (int i = 0; i < 64; ++i) lt.v[i] = host[i]; return lt; }

// Balanced variant of the map above (r3).  With "level 15-x, then level x" the XCDs whose first level is fine - samples in different cells, every gather a line of its
// own - work ~60 us on it while the XCDs that drew two coherent levels are done after ~10 us and idle (lego: six fine levels on XCDs 0..5, XCDs 6 and 7 wait).
// FwdMap hands every XCD a list of (level, chunk range) segments of equal estimated COST instead: fine levels (resolution above the run-combining limit, where
// consecutive samples of a ray stop sharing cells) weigh 1 per chunk, the others `light`; the levels are laid end to end, finest first, and cut into eight equal
// shares, so an XCD still works on at most two fine levels (their slices stay in its L2) but none waits for the others.  Speed only, results unchanged.
#define FWD_MAP_SEGS 17
#define FWD_MAP_LIGHT_SPAN 8u       // a block of a coherent level takes this many consecutive chunks (they cost ~1/8 of a fine level's: equal work per block, no swarm of tiny blocks)
struct FwdMap { uint32_t level[8][FWD_MAP_SEGS], begin[8][FWD_MAP_SEGS], count[8][FWD_MAP_SEGS] /* chunks */; uint32_t slots; };     // slots = blocks per XCD in the launch (the longest list)
__host__ __device__ static inline uint32_t fwd_map_span(const LevelTable &lt, uint32_t level) { return lt.v[4 * level + 2] > 300u ? 1u : FWD_MAP_LIGHT_SPAN; }
static FwdMap fwd_map_balanced(const LevelTable &lt, uint32_t nblk, float light) {
	FwdMap m; memset(&m, 0, sizeof(m));
	float w[16], total = 0.f;
	for (int l = 0; l < 16; ++l) { w[l] = lt.v[4 * l + 2] > 300u ? 1.0f : light; total += w[l] * (float)nblk; }
	const float share = total / 8.0f;
	uint32_t xcd = 0, seg = 0; float used = 0.f;
	for (int l = 15; l >= 0; --l) {
		uint32_t done = 0;
		while (done < nblk) {
			const float room = share - used;
			uint32_t take = xcd == 7u ? nblk - done : (uint32_t)(room / w[l] + 0.5f);
			if (take > nblk - done) take = nblk - done;
			if (take == 0 && xcd < 7u) { ++xcd; seg = 0; used = 0.f; continue; }
			if (seg == FWD_MAP_SEGS) { if (xcd < 7u) { ++xcd; seg = 0; used = 0.f; continue; } --seg; m.count[xcd][seg] += take; done += take; ++seg; continue; }   // (cannot happen with 16 levels / 8 shares; keeps the map total anyway)
			m.level[xcd][seg] = (uint32_t)l; m.begin[xcd][seg] = done; m.count[xcd][seg] = take; ++seg;
			done += take; used += (float)take * w[l];
			if (used >= share - 0.5f * w[l] && xcd < 7u) { ++xcd; seg = 0; used = 0.f; }
		}
	}
	for (int x = 0; x < 8; ++x) {
		uint32_t c = 0;
		for (int g = 0; g < FWD_MAP_SEGS; ++g) if (m.count[x][g]) c += div_up(m.count[x][g], fwd_map_span(lt, m.level[x][g]));
		if (c > m.slots) m.slots = c;
	}
	return m;
}
// Variant 2 (measured after variant 1 lost: 75 -> 114 us - an XCD that works on TWO fine levels thrashes its 4 MiB L2 between two 4 MiB tables): the round-1 map
// (XCD x: level 15-x, then level x) stays, so a fine level's table lives in ONE XCD's L2, but the XCDs whose two levels are both coherent ("helpers": 6 and 7 for the
// ngp_base.py table) additionally take the last `help` fraction of the chunks of every fine level, dealt round-robin - they thrash, but only on a small share.
static FwdMap fwd_map_helpers(const LevelTable &lt, uint32_t nblk, float help) {
	FwdMap m; memset(&m, 0, sizeof(m));
	bool heavy[16]; for (int l = 0; l < 16; ++l) heavy[l] = lt.v[4 * l + 2] > 300u;
	int helpers[8], n_help = 0;
	for (int x = 0; x < 8; ++x) if (!heavy[15 - x] && !heavy[x]) helpers[n_help++] = x;
	uint32_t seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	auto add = [&](int x, uint32_t level, uint32_t begin, uint32_t count) { if (count && seg[x] < FWD_MAP_SEGS) { m.level[x][seg[x]] = level; m.begin[x][seg[x]] = begin; m.count[x][seg[x]] = count; ++seg[x]; } };
	uint32_t given = (n_help && nblk >= 16u) ? (uint32_t)((float)nblk * help) : 0u;
	int rr = 0;
	for (int phase = 0; phase < 2; ++phase)
		for (int x = 0; x < 8; ++x) {
			const uint32_t l = phase == 0 ? 15u - x : (uint32_t)x;
			const uint32_t keep = heavy[l] ? nblk - given : nblk;
			add(x, l, 0u, keep);
		}
	if (given) for (int l = 15; l >= 0; --l) if (heavy[l]) { add(helpers[rr % n_help], (uint32_t)l, nblk - given, given); ++rr; }
	for (int x = 0; x < 8; ++x) {
		uint32_t c = 0;
		for (int g = 0; g < FWD_MAP_SEGS; ++g) if (m.count[x][g]) c += div_up(m.count[x][g], fwd_map_span(lt, m.level[x][g]));
		if (c > m.slots) m.slots = c;
	}
	return m;
}
// Variant 3: every fine level keeps its own XCD and ALL its chunks (L2 residency untouched); only the coherent levels move - instead of following a fine level on the same
// XCD (the second phase of the round-1 map) they are dealt to the XCDs that hold no fine level at all, which would otherwise idle.  No helper XCD (fox: eight fine levels): unchanged.
static FwdMap fwd_map_light_aside(const LevelTable &lt, uint32_t nblk) {
	FwdMap m; memset(&m, 0, sizeof(m));
	bool heavy[16]; for (int l = 0; l < 16; ++l) heavy[l] = lt.v[4 * l + 2] > 300u;
	int helpers[8], n_help = 0;
	for (int x = 0; x < 8; ++x) if (!heavy[15 - x] && !heavy[x]) helpers[n_help++] = x;
	uint32_t seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	auto add = [&](int x, uint32_t level) { if (seg[x] < FWD_MAP_SEGS) { m.level[x][seg[x]] = level; m.begin[x][seg[x]] = 0u; m.count[x][seg[x]] = nblk; ++seg[x]; } };
	int rr = 0;
	for (int phase = 0; phase < 2; ++phase)
		for (int x = 0; x < 8; ++x) {
			const uint32_t l = phase == 0 ? 15u - x : (uint32_t)x;
			if (heavy[l] || n_help == 0) add(x, l);
			else { add(helpers[rr % n_help], l); ++rr; }
		}
	for (int x = 0; x < 8; ++x) {
		uint32_t c = 0;
		for (int g = 0; g < FWD_MAP_SEGS; ++g) if (m.count[x][g]) c += div_up(m.count[x][g], fwd_map_span(lt, m.level[x][g]));
		if (c > m.slots) m.slots = c;
	}
	return m;
}
__device__ __forceinline__ bool block_to_level_chunk_map(const FwdMap &m, const LevelTable &lt, uint32_t &level, uint32_t &chunk, uint32_t &chunk_end) {
	const uint32_t b = blockIdx.x, xcd = b & 7u;
	uint32_t slot = b >> 3;
	for (int g = 0; g < FWD_MAP_SEGS; ++g) {
		const uint32_t c = m.count[xcd][g];
		if (!c) continue;
		const uint32_t span = fwd_map_span(lt, m.level[xcd][g]), blocks = (c + span - 1u) / span;
		if (slot < blocks) { level = m.level[xcd][g]; chunk = m.begin[xcd][g] + slot * span; chunk_end = min(chunk + span, m.begin[xcd][g] + c); return true; }
		slot -= blocks;
	}
	return false;
}

struct Corner { uint32_t g[3]; float w[3]; };
__device__ __forceinline__ Corner locate(const float *pos, uint32_t stride, uint32_t i, float scale) {
	Corner c;
#pragma unroll
	for (int d = 0; d < 3; ++d) {            // pos_fract, HashEncode.h:106-115
		float p = pos[(size_t)i * stride + d] * scale + 0.5f;
		float fl = floorf(p);
		c.g[d] = (uint32_t)(int)fl;
		c.w[d] = p - fl;
	}
	return c;
}

template <typename T, int LAYOUT, bool MAPPED>
__device__ __forceinline__ void hash_fwd_body(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ table, const LevelTable &lt,
                                              T *__restrict__ out, uint32_t nblk, const uint32_t *__restrict__ n_valid, uint32_t level, uint32_t chunk, float *__restrict__ dy_dx = nullptr);
// forward with d(encoding)/d(position) - the dy_dx branch of the reference's kernel_grid (HashEncode.h:205-251): same gathers, three more outputs per (sample, level)
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void k_hash_fwd_dydx(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ table, LevelTable lt,
                                                       T *__restrict__ out, uint32_t nblk, const uint32_t *__restrict__ n_valid, float *__restrict__ dy_dx) {
	uint32_t level, chunk; block_to_level_chunk(nblk, level, chunk);
	hash_fwd_body<T, LAYOUT, false>(n, pos, stride, table, lt, out, nblk, n_valid, level, chunk, dy_dx);
}
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void k_hash_fwd(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ table, LevelTable lt,
                                                  T *__restrict__ out, uint32_t nblk, const uint32_t *__restrict__ n_valid) {
	uint32_t level, chunk; block_to_level_chunk(nblk, level, chunk);
	hash_fwd_body<T, LAYOUT, false>(n, pos, stride, table, lt, out, nblk, n_valid, level, chunk);
}
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void k_hash_fwd_bal(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ table, LevelTable lt,
                                                      T *__restrict__ out, uint32_t nblk, const uint32_t *__restrict__ n_valid, FwdMap map) {
	uint32_t level, chunk, chunk_end;
	if (!block_to_level_chunk_map(map, lt, level, chunk, chunk_end)) return;
	for (; chunk < chunk_end; ++chunk) hash_fwd_body<T, LAYOUT, true>(n, pos, stride, table, lt, out, nblk, n_valid, level, chunk);
}
template <typename T, int LAYOUT, bool MAPPED>
__device__ __forceinline__ void hash_fwd_body(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ table, const LevelTable &lt,
                                              T *__restrict__ out, uint32_t nblk, const uint32_t *__restrict__ n_valid, uint32_t level, uint32_t chunk, float *__restrict__ dy_dx) {
	using P = typename Pair<T>::type;
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	const uint32_t off = lt.v[4 * level], size = lt.v[4 * level + 1], res = lt.v[4 * level + 2];
	const float scale = __uint_as_float(lt.v[4 * level + 3]);
	const bool dense = level_is_dense(size, res);
	const P *tab = reinterpret_cast<const P *>(table) + off;
	// nblk is capped by the host: a block takes chunks chunk, chunk + nblk, ... of its level (one trip for a training batch; the fixed-capacity inference buffers
	// - 4 M rows of which a device-side count says how many are valid - used to launch 260 k blocks of which 87 % found nothing to do)
	for (uint32_t i = chunk * 256u + threadIdx.x; i < lim; i += nblk * 256u) {
	const Corner c = locate(pos, stride, i, scale);
	P v[8]; float w[8];
	// The kernel is bound by the L2 request rate (one gather = one request), so the two x-neighbours of a cell edge are fetched with ONE double-width load
	// whenever they are adjacent in memory: always on dense levels (index x + ...; not across the wrap), and on hashed levels when x is even
	// ((x+1) ^ h == (x ^ h) ^ 1 then) - 6 requests per (sample, level) on average instead of 8.  The second single load is issued only by the other lanes.
	struct alignas(sizeof(P)) PP { P a, b; };
	const bool pow2 = (size & (size - 1)) == 0;
#pragma unroll
	for (uint32_t j = 0; j < 4; ++j) {        // j = (y corner, z corner); all gathers are issued before the first use
		const uint32_t gy = c.g[1] + (j & 1u), gz = c.g[2] + (j >> 1);
		const float wy = (j & 1u) ? c.w[1] : 1 - c.w[1], wz = (j >> 1) ? c.w[2] : 1 - c.w[2];
		w[2 * j] = ((1 - c.w[0]) * wy) * wz; w[2 * j + 1] = (c.w[0] * wy) * wz;          // the reference's x, y, z multiplication order
		const uint32_t i0 = grid_index(size, res, dense, c.g[0], gy, gz), i1 = grid_index(size, res, dense, c.g[0] + 1, gy, gz);
		const bool adjacent_up = i1 == i0 + 1u && (dense || pow2), adjacent_dn = i0 == i1 + 1u && !dense && pow2;     // (x^h)^1 is either one above or one below
		if (adjacent_up) { const PP t = *reinterpret_cast<const PP *>(tab + i0); v[2 * j] = t.a; v[2 * j + 1] = t.b; }
		else if (adjacent_dn) { const PP t = *reinterpret_cast<const PP *>(tab + i1); v[2 * j] = t.b; v[2 * j + 1] = t.a; }
		else { v[2 * j] = tab[i0]; v[2 * j + 1] = tab[i1]; }
	}
	float2 acc = make_float2(0.f, 0.f);
#pragma unroll
	for (uint32_t k = 0; k < 8; ++k) { float2 f = to_f2(v[k]); acc.x += w[k] * f.x; acc.y += w[k] * f.y; }
	P r; from_f2(r, acc);
	P *o = reinterpret_cast<P *>(out);
	if (LAYOUT == NGP_LAYOUT_SOA) o[(size_t)level * n + i] = r;
	else o[(size_t)i * 16 + level] = r;
	if (dy_dx) {
		// HashEncode.h:205-251: per derivative dimension the four (left, right) pairs along it, weight = scale * w(first other dim) * w(second other dim) in that order,
		// summed in the reference's idx order (bit 0 = first other dim).  v[k]: corner k = x + 2 y + 4 z, already in registers - no extra gathers.
		float2 f[8];
#pragma unroll
		for (uint32_t k = 0; k < 8; ++k) f[k] = to_f2(v[k]);
#pragma unroll
		for (uint32_t gd = 0; gd < 3; ++gd) {
			const uint32_t d0 = gd == 0 ? 1u : 0u, d1 = gd == 2 ? 1u : 2u;           // the two non-derivative dimensions, ascending
			float2 a = make_float2(0.f, 0.f);
#pragma unroll
			for (uint32_t idx = 0; idx < 4; ++idx) {
				const uint32_t b0 = idx & 1u, b1 = idx >> 1;
				float weight = scale;
				weight *= b0 ? c.w[d0] : 1 - c.w[d0];
				weight *= b1 ? c.w[d1] : 1 - c.w[d1];
				const uint32_t left = (b0 << d0) | (b1 << d1), right = left | (1u << gd);
				a.x += weight * (f[right].x - f[left].x) * 1.0f;
				a.y += weight * (f[right].y - f[left].y) * 1.0f;
			}
			*reinterpret_cast<float2 *>(dy_dx + (size_t)i * 96 + gd * 32 + 2 * level) = a;
		}
	}
	}
}

// dL/dx[i][d] = sum_k dL/dy[i][k] * dy_dx[i][d][k] (fp32, k ascending): the contraction GridEncode.grad needs to return a position gradient.  The reference returns
// None there (grid_encode.py:190) and has no kernel for it - restated from the chain rule (tiny-cuda-nn's kernel_grid_backward_input computes the same sum).
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void k_hash_bwd_input(uint32_t n, const T *__restrict__ dLdy, const float *__restrict__ dy_dx, float *__restrict__ dLdx, const uint32_t *__restrict__ n_valid) {
	using P = typename Pair<T>::type;
	const uint32_t t = blockIdx.x * 256u + threadIdx.x, i = t / 3u, d = t - 3u * i;
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	if (i >= lim) return;
	const P *dy = reinterpret_cast<const P *>(dLdy);
	const float2 *row = reinterpret_cast<const float2 *>(dy_dx + (size_t)i * 96 + d * 32);
	float a = 0.f;
#pragma unroll
	for (uint32_t l = 0; l < 16; ++l) {
		const float2 g = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)l * n + i] : dy[(size_t)i * 16 + l]);
		const float2 r = row[l];
		a += g.x * r.x; a += g.y * r.y;
	}
	dLdx[(size_t)i * 3 + d] = a;
}

__device__ __forceinline__ void atomic_add_pair(float *p, float2 v) {
	__hip_atomic_fetch_add(p, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	__hip_atomic_fetch_add(p + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void atomic_add_pair(__half *p, float2 v) {
	typedef _Float16 __attribute__((ext_vector_type(2))) h2;
	h2 x; x[0] = (_Float16)v.x; x[1] = (_Float16)v.y;
	(void)__builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2 *)p, x);   // global_atomic_pk_add_f16
}

template <typename T, typename G, int LAYOUT>
__global__ __launch_bounds__(256) void k_hash_bwd(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ dLdy, LevelTable lt,
                                                  G *__restrict__ grad, uint32_t nblk, const uint32_t *__restrict__ n_valid) {
	using P = typename Pair<T>::type;
	uint32_t level, chunk; block_to_level_chunk(nblk, level, chunk);
	const uint32_t i = chunk * 256u + threadIdx.x;
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	if (i >= lim) return;
	const uint32_t off = lt.v[4 * level], size = lt.v[4 * level + 1], res = lt.v[4 * level + 2];
	const float scale = __uint_as_float(lt.v[4 * level + 3]);
	const bool dense = level_is_dense(size, res);
	const P *dy = reinterpret_cast<const P *>(dLdy);
	const float2 g2 = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)level * n + i] : dy[(size_t)i * 16 + level]);
	if (g2.x == 0.f && g2.y == 0.f) return;   // zero-padded rows add exact zeros in the reference; skipping them is value-identical
	const Corner c = locate(pos, stride, i, scale);
	G *gl = grad + (size_t)off * 2;
#pragma unroll
	for (uint32_t k = 0; k < 8; ++k) {
		float weight = 1; uint32_t g[3];
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			if ((k & (1u << d)) == 0) { weight *= 1 - c.w[d]; g[d] = c.g[d]; }
			else { weight *= c.w[d]; g[d] = c.g[d] + 1; }
		}
		const uint32_t idx = grid_index(size, res, dense, g[0], g[1], g[2]);
		atomic_add_pair(gl + (size_t)idx * 2, make_float2(g2.x * weight, g2.y * weight));
	}
}


// ---------------------------------------------------------------------------------------------------------------- owner-computes scatter
// (Since round 2 this scan is the FALLBACK: calls without a workspace, the fixed-point request of ngp_hash_encode_bwd_fx, levels beyond 2^19 entries.  The training
// path hands over a workspace and every level goes through the binned scatter further down.)
// Measured on MI355X (tools/microbench_hash.py, profiles/): float atomics to global memory retire at ~20 G instructions/s chip-wide no
// matter how local they are, i.e. >= 3.3 ms for the 2 x 33.5 M updates of one 2^18-sample batch (7.8 ms on real, spatially concentrated
// samples).  This kernel removes them: every workgroup OWNS a contiguous slice of one level's table (16384 entries = 128 KiB of fp32
// pairs in its LDS — a CU has 160 KiB), scans the samples, recomputes the eight corner indices and accumulates only the corners that
// fall inside its slice with LDS atomics (ds_add_f32, orders of magnitude faster than memory-side atomics).  A slice owned by a
// single workgroup is written back with plain coalesced stores: no global atomics, no memset of the 50 MB gradient, and the table
// gradient becomes deterministic up to the fp32 add order inside one workgroup.  The small dense levels (whose whole table fits one
// slice and whose updates collide heavily) are instead split over up to 32 sample chunks with a private LDS copy each and a short
// atomic flush (a few thousand adds per workgroup).  The redundant index arithmetic (each sample is visited by every slice owner of a
// level) is ~1e10 lane-ops per batch — about 0.15 ms of VALU time on 256 CUs — and the sample stream is re-read from L2, not HBM.
#define OWN_SLICE 16384u

__device__ __forceinline__ float bin_scale(uint32_t absmax_bits) {     // power of two s with 2^13 <= max*s < 2^14 (0 if the level has no gradient)
	const float m = __uint_as_float(absmax_bits);
	if (!(m > 0.f) || !(m < 3.0e38f)) return 0.f;
	int ex; frexpf(m, &ex);                                           // m = f * 2^ex, f in [0.5, 1)
	return ldexpf(1.0f, 14 - ex);
}


// One accumulation into the owned LDS slice.  FX = false: two ds_add_f32 (the LDS float-atomic path retires ~1 lane / 3 cycles / CU on gfx950).
// FX = true: both features as 32-bit fixed-point fields of ONE ds_add_u64 (16.6 cycles per wave instruction, tools/microbench_lds.py):
// sum = (sum_y << 32) + sum_x in two's complement, decoded exactly at the flush.  The scale is the power of two with scale * L1(level) <= 2^30,
// where L1(level) = sum over all samples of |dL/dy| bounds any entry's |sum| — overflow is impossible by construction, integer adds commute,
// so the exclusive slices become bit-reproducible.
template <int FX>
__device__ __forceinline__ void acc_add(float *acc, uint32_t l, float vx, float vy, float fx_scale) {
	if (FX == 1) {
		const int ix = __float2int_rn(vx * fx_scale), iy = __float2int_rn(vy * fx_scale);
		const unsigned long long add = (unsigned long long)(long long)ix + ((unsigned long long)(uint32_t)iy << 32);
		__hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(acc) + l, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	} else {
		__hip_atomic_fetch_add(&acc[2 * l], vx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		__hip_atomic_fetch_add(&acc[2 * l + 1], vy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
}
template <int FX>
__device__ __forceinline__ float2 acc_read(const float *acc, uint32_t e, float fx_inv) {
	if (FX == 1) {
		const unsigned long long t = reinterpret_cast<const unsigned long long *>(acc)[e];
		const int lo = (int)(uint32_t)(t & 0xffffffffull);
		const int hi = (int)(uint32_t)((t - (unsigned long long)(long long)lo) >> 32);
		return make_float2((float)lo * fx_inv, (float)hi * fx_inv);
	}
	return make_float2(acc[2 * e], acc[2 * e + 1]);
}

struct OwnerPlan { uint32_t first_unit[17]; uint32_t chunks[16]; uint32_t order[16]; uint32_t slab_off[16]; uint32_t level_mask; uint32_t coarse_res; };   // slab_off: float2 offset of the level's [chunks][size] partial slabs, ~0u = none

template <typename T, typename G, int LAYOUT, bool HASHED, bool COMBINE, int FX>
__device__ __forceinline__ void owner_unit(float fx_scale, uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ dLdy, const LevelTable &lt, uint32_t level,
                                           uint32_t slice, uint32_t chunk, uint32_t n_chunks, G *__restrict__ grad, int accumulate, uint32_t lim, float *acc, float2 *__restrict__ slab) {
	using P = typename Pair<T>::type;
	using GP = typename Pair<G>::type;
	const uint32_t off = lt.v[4 * level], size = lt.v[4 * level + 1], res = lt.v[4 * level + 2];
	const float scale = __uint_as_float(lt.v[4 * level + 3]);
	constexpr uint32_t SLICE = OWN_SLICE;
	const uint32_t lo = slice * SLICE;
	const uint32_t cnt = min(SLICE, size - lo);
	for (uint32_t e = threadIdx.x; e < cnt * 2; e += 1024) acc[e] = 0.f;
	__syncthreads();
	const uint32_t per = ((lim + n_chunks - 1) / n_chunks + 7u) & ~7u;
	const uint32_t begin = min(chunk * per, lim), end = min(begin + per, lim);
	const P *dy = reinterpret_cast<const P *>(dLdy);
	const uint32_t res2 = res * res;
	// Each thread takes OWN_K consecutive samples per trip: (a) all of their loads are issued before the first use (the loop is otherwise
	// latency-bound: two dependent L2 round trips per sample), (b) neighbouring lanes are OWN_K samples apart, so the consecutive samples
	// of one ray — which share cells on the coarse levels — never meet in the same ds_add and LDS same-address serialisation disappears.
	constexpr uint32_t OWN_K = 8;
	for (uint32_t base = begin + threadIdx.x * OWN_K; base < end; base += 1024 * OWN_K) {
		float px[OWN_K][3]; float2 gk[OWN_K];
		if (base + OWN_K <= end && stride == 3) {
			const float4 *p4 = reinterpret_cast<const float4 *>(pos + (size_t)base * 3);    // 24 floats, 16-byte aligned (base % 8 == 0)
			float4 v[6];
#pragma unroll
			for (int r = 0; r < 6; ++r) v[r] = p4[r];
			const float *f = reinterpret_cast<const float *>(v);
#pragma unroll
			for (uint32_t kk = 0; kk < OWN_K; ++kk) { px[kk][0] = f[3 * kk]; px[kk][1] = f[3 * kk + 1]; px[kk][2] = f[3 * kk + 2]; }
#pragma unroll
			for (uint32_t kk = 0; kk < OWN_K; ++kk) gk[kk] = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)level * n + base + kk] : dy[(size_t)(base + kk) * 16 + level]);
		} else {
#pragma unroll
			for (uint32_t kk = 0; kk < OWN_K; ++kk) {
				const uint32_t i = base + kk;
				if (i < end) {
					px[kk][0] = pos[(size_t)i * stride]; px[kk][1] = pos[(size_t)i * stride + 1]; px[kk][2] = pos[(size_t)i * stride + 2];
					gk[kk] = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)level * n + i] : dy[(size_t)i * 16 + level]);
				} else { px[kk][0] = px[kk][1] = px[kk][2] = 0.f; gk[kk] = make_float2(0.f, 0.f); }
			}
		}
		if (COMBINE) {
			// Coarse levels: the consecutive samples a thread holds (one or two rays) mostly sit in ONE cell.  Their eight corner
			// contributions are summed in registers and sent to LDS once per run — ds_add_f32 retires ~1 lane per 3 cycles on gfx950
			// (tools/microbench_lds.py), so the hot slices of the small dense levels would otherwise serialise for milliseconds.
			uint32_t key[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, local[8], hits = 0;
			float ax[8], ay[8];
#pragma unroll
			for (uint32_t q = 0; q < 8; ++q) { ax[q] = 0.f; ay[q] = 0.f; local[q] = 0; }
#pragma unroll
			for (uint32_t kk = 0; kk <= OWN_K; ++kk) {
				Corner c;
				bool same = false;
				float2 g2 = make_float2(0.f, 0.f);
				if (kk < OWN_K) {
					g2 = gk[kk];
#pragma unroll
					for (int d = 0; d < 3; ++d) { const float p = px[kk][d] * scale + 0.5f; const float fl = floorf(p); c.g[d] = (uint32_t)(int)fl; c.w[d] = p - fl; }
					same = c.g[0] == key[0] && c.g[1] == key[1] && c.g[2] == key[2];
				}
				if (!same) {                                  // run ends (or final flush at kk == OWN_K)
#pragma unroll
					for (uint32_t q = 0; q < 8; ++q) {
						if ((hits >> q) & 1u) {
							if ((ax[q] != 0.f || ay[q] != 0.f) && !(accumulate & 2)) acc_add<FX>(acc, local[q], ax[q], ay[q], fx_scale);   // (& 2: probe, everything but the LDS atomics)
						}
						ax[q] = 0.f; ay[q] = 0.f;
					}
					if (kk < OWN_K) {
						key[0] = c.g[0]; key[1] = c.g[1]; key[2] = c.g[2];
						uint32_t tx[2], ty[2], tz[2];
						tx[0] = c.g[0]; tx[1] = c.g[0] + 1;
						if (HASHED) { ty[0] = c.g[1] * 19349663u; ty[1] = ty[0] + 19349663u; tz[0] = c.g[2] * 83492791u; tz[1] = tz[0] + 83492791u; }
						else { ty[0] = c.g[1] * res; ty[1] = ty[0] + res; tz[0] = c.g[2] * res2; tz[1] = tz[0] + res2; }
						hits = 0;
#pragma unroll
						for (uint32_t q = 0; q < 8; ++q) {
							uint32_t idx;
							if (HASHED) idx = (tx[q & 1] ^ ty[(q >> 1) & 1] ^ tz[q >> 2]) & (size - 1);
							else { idx = tx[q & 1] + ty[(q >> 1) & 1] + tz[q >> 2]; if (idx >= size) { idx -= size; if (idx >= size) idx %= size; } }
							local[q] = idx - lo;
							hits |= (local[q] < cnt) ? (1u << q) : 0u;
						}
					}
				}
				if (kk < OWN_K && hits) {
					const float x1 = c.w[0], x0 = 1 - c.w[0], y1 = c.w[1], y0 = 1 - c.w[1], z1 = c.w[2], z0 = 1 - c.w[2];
					const float xy[4] = {x0 * y0, x1 * y0, x0 * y1, x1 * y1};
#pragma unroll
					for (uint32_t q = 0; q < 8; ++q) { const float wq = xy[q & 3] * ((q >> 2) ? z1 : z0); ax[q] += wq * g2.x; ay[q] += wq * g2.y; }
				}
			}
			continue;
		}
#pragma unroll
		for (uint32_t kk = 0; kk < OWN_K; ++kk) {
			const float2 g2 = gk[kk];
			Corner c;
#pragma unroll
			for (int d = 0; d < 3; ++d) { const float p = px[kk][d] * scale + 0.5f; const float fl = floorf(p); c.g[d] = (uint32_t)(int)fl; c.w[d] = p - fl; }
			// the three per-axis terms of the index are shared by the eight corners
			uint32_t tx[2], ty[2], tz[2];
			tx[0] = c.g[0]; tx[1] = c.g[0] + 1;
			if (HASHED) { ty[0] = c.g[1] * 19349663u; ty[1] = ty[0] + 19349663u; tz[0] = c.g[2] * 83492791u; tz[1] = tz[0] + 83492791u; }
			else { ty[0] = c.g[1] * res; ty[1] = ty[0] + res; tz[0] = c.g[2] * res2; tz[1] = tz[0] + res2; }
			uint32_t hits = 0;
			if (HASHED) {
				// Slice test for all eight corners at once: the slice id of a corner is bits 14..18 of tx^ty^tz and XOR commutes with bit
				// extraction, so it is sx_i ^ sy_j ^ sz_k of three 5-bit fields.  Four (i,j) combinations are packed one per byte, the two
				// k values are XORed in with a byte-replicating multiply, and "byte == slice" becomes a zero-byte test (no cross-byte borrow
				// because every byte is < 32).  ~30 VALU ops instead of ~100 for eight separate index computations.
				const uint32_t sx0 = (tx[0] >> 14) & 31u, sx1 = (tx[1] >> 14) & 31u, sy0 = (ty[0] >> 14) & 31u, sy1 = (ty[1] >> 14) & 31u;
				const uint32_t sz0 = (tz[0] >> 14) & 31u, sz1 = (tz[1] >> 14) & 31u;
				const uint32_t A = (sx0 ^ sy0) | ((sx1 ^ sy0) << 8) | ((sx0 ^ sy1) << 16) | ((sx1 ^ sy1) << 24);
				const uint32_t S = slice * 0x01010101u;
				const uint32_t X0 = A ^ (sz0 * 0x01010101u) ^ S, X1 = A ^ (sz1 * 0x01010101u) ^ S;
				const uint32_t m0 = ~((X0 | 0x80808080u) - 0x01010101u) & 0x80808080u, m1 = ~((X1 | 0x80808080u) - 0x01010101u) & 0x80808080u;
				hits = ((m0 * 0x00204081u) >> 28) | (((m1 * 0x00204081u) >> 28) << 4);     // msb of each byte -> one bit per corner
			} else {
#pragma unroll
				for (uint32_t q = 0; q < 8; ++q) {
					uint32_t idx = tx[q & 1] + ty[(q >> 1) & 1] + tz[q >> 2];
					if (idx >= size) { idx -= size; if (idx >= size) idx %= size; }                                            // wraps only at the +1 boundary corner
					hits |= (idx - lo < cnt) ? (1u << q) : 0u;
				}
			}
			if (g2.x == 0.f && g2.y == 0.f) hits = 0;       // zero-padded rows add exact zeros in the reference; skipping them is value-identical
			while (hits) {                      // ~8/32 corners per sample land in this slice: one short divergent loop instead of eight regions
				const uint32_t q = __builtin_ctz(hits);
				hits &= hits - 1;
				const uint32_t ex = (q & 1u) ? tx[1] : tx[0], ey = (q & 2u) ? ty[1] : ty[0], ez = (q & 4u) ? tz[1] : tz[0];
				uint32_t idx;
				if (HASHED) idx = (ex ^ ey ^ ez) & (size - 1);
				else { idx = ex + ey + ez; if (idx >= size) { idx -= size; if (idx >= size) idx %= size; } }
				const uint32_t l = idx - lo;
				const float wx = (q & 1u) ? c.w[0] : 1 - c.w[0], wy = (q & 2u) ? c.w[1] : 1 - c.w[1], wz = (q & 4u) ? c.w[2] : 1 - c.w[2];
				const float weight = wx * wy * wz;
				if (accumulate & 2) { if (weight == 123.f) acc[2 * l] = g2.x; continue; }   // probe: everything but the LDS atomics
				acc_add<FX>(acc, l, g2.x * weight, g2.y * weight, fx_scale);
			}
		}
	}
	__syncthreads();
	accumulate &= 1;
	const float fx_inv = FX != 0 ? 1.0f / fx_scale : 1.0f;
	G *gl = grad + ((size_t)off + lo) * 2;
	if (n_chunks == 1) {            // exclusive owner: plain stores (or a private read-modify-write when accumulating)
		for (uint32_t e = threadIdx.x; e < cnt; e += 1024) {
			float2 v = acc_read<FX>(acc, e, fx_inv);
			GP *dst = reinterpret_cast<GP *>(gl) + e;
			if (accumulate) { const float2 old = to_f2(*dst); v.x += old.x; v.y += old.y; }
			GP o; from_f2(o, v);
			*dst = o;
		}
	} else if (slab) {              // shared slice, workspace given: plain store of this chunk's partial slab ([chunk][level entries]); k_reduce_dense sums the chunks
		float2 *dst = slab + (size_t)chunk * size + lo;
		for (uint32_t e = threadIdx.x; e < cnt; e += 1024) dst[e] = acc_read<FX>(acc, e, fx_inv);
	} else {                        // shared slice, no workspace: flush touched entries with global atomics; the host side zeroed the level unless accumulating
		for (uint32_t e = threadIdx.x; e < cnt; e += 1024) {
			const float2 v = acc_read<FX>(acc, e, fx_inv);
			if (v.x != 0.f || v.y != 0.f) atomic_add_pair(gl + (size_t)e * 2, v);
		}
	}
}

template <typename T, typename G, int LAYOUT>
__global__ __launch_bounds__(1024) void k_hash_bwd_owner(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ dLdy, LevelTable lt,
                                                         OwnerPlan plan, G *__restrict__ grad, int accumulate, const uint32_t *__restrict__ n_valid,
                                                         const float *__restrict__ level_l1, float2 *__restrict__ slabs) {
	extern __shared__ __attribute__((aligned(16))) float acc[];          // [slice entries][2]
	// block -> (level, slice, chunk); plan.order lists the chunked dense levels first, then the exclusive-owner (hashed) levels
	uint32_t k = 0;
	while (k < 15 && blockIdx.x >= plan.first_unit[k + 1]) ++k;
	const uint32_t level = plan.order[k];
	const uint32_t u = blockIdx.x - plan.first_unit[k];
	const uint32_t n_chunks = plan.chunks[level];
	const uint32_t slice = u / n_chunks, chunk = u - slice * n_chunks;
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	float2 *slab = (slabs && plan.slab_off[level] != ~0u) ? slabs + plan.slab_off[level] : nullptr;
	if (!((plan.level_mask >> level) & 1u)) return;      // probe hook (tools/microbench_hash.py); all ones in production
	const uint32_t size = lt.v[4 * level + 1], res = lt.v[4 * level + 2];
	const bool coarse = res <= plan.coarse_res;     // cells much longer than a marching step: consecutive samples of a ray share them
	const bool dense = level_is_dense(size, res);
#define OWNER_GO(H, C, F, SC) owner_unit<T, G, LAYOUT, H, C, F>(SC, n, pos, stride, dLdy, lt, level, slice, chunk, n_chunks, grad, accumulate, lim, acc, slab)
	if (level_l1) {
		const float l1 = level_l1[level];
		if (!(l1 > 0.f)) {                                       // nothing to add on this level: write zeros / leave the accumulating buffer alone
			if (n_chunks == 1 && !(accumulate & 1)) {
				const uint32_t off = lt.v[4 * level], lo = slice * OWN_SLICE, cnt = min(OWN_SLICE, size - lo);
				typename Pair<G>::type z; from_f2(z, make_float2(0.f, 0.f));
				for (uint32_t e = threadIdx.x; e < cnt; e += 1024) reinterpret_cast<typename Pair<G>::type *>(grad + ((size_t)off + lo) * 2)[e] = z;
			}
			return;
		}
		int ex; frexpf(l1, &ex);                                 // l1 < 2^ex  =>  scale = 2^(30-ex) keeps |sum| * scale < 2^30
		const float sc = ldexpf(1.0f, 30 - ex);
		if (dense) OWNER_GO(false, true, 1, sc); else if (coarse) OWNER_GO(true, true, 1, sc); else OWNER_GO(true, false, 1, sc);
	} else {
		if (dense) OWNER_GO(false, true, 0, 1.0f); else if (coarse) OWNER_GO(true, true, 0, 1.0f); else OWNER_GO(true, false, 0, 1.0f);
	}
#undef OWNER_GO
	// (hashed levels with a non-power-of-two table never reach this kernel: the host routes them to the atomic kernel)
}


// per-level L1 norm of dL/dy (the overflow bound of the fixed-point accumulation): l1[level] += sum |dy|
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void k_level_l1(uint32_t n, const T *__restrict__ dLdy, float *__restrict__ l1, const uint32_t *__restrict__ n_valid) {
	using P = typename Pair<T>::type;
	const uint32_t level = blockIdx.y;
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	const P *dy = reinterpret_cast<const P *>(dLdy);
	float s = 0.f;
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < lim; i += gridDim.x * 256u) {
		const float2 g = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)level * n + i] : dy[(size_t)i * 16 + level]);
		s += fabsf(g.x) + fabsf(g.y);
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
	if ((threadIdx.x & 63u) == 0 && s != 0.f) __hip_atomic_fetch_add(&l1[level], s * 1.0001f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // 1e-4 slack for the fp32 summation error
}


// grad[level entries] (=|+=) sum over chunks of the partial slabs written by the shared (dense-level) units
template <typename G>
__global__ __launch_bounds__(256) void k_reduce_dense(LevelTable lt, OwnerPlan plan, const float2 *__restrict__ slabs, G *__restrict__ grad, int accumulate) {
	const uint32_t level = blockIdx.y;
	if (plan.slab_off[level] == ~0u) return;
	const uint32_t off = lt.v[4 * level], size = lt.v[4 * level + 1], n_chunks = plan.chunks[level];
	const float2 *sl = slabs + plan.slab_off[level];
	using GP = typename Pair<G>::type;
	for (uint32_t e = blockIdx.x * 256u + threadIdx.x; e < size; e += gridDim.x * 256u) {
		float2 v = make_float2(0.f, 0.f);
		if (n_chunks == 32) {                                  // all 32 loads in flight at once
			float2 t[32];
#pragma unroll
			for (uint32_t c = 0; c < 32; ++c) t[c] = sl[(size_t)c * size + e];
#pragma unroll
			for (uint32_t c = 0; c < 32; ++c) { v.x += t[c].x; v.y += t[c].y; }
		} else
		for (uint32_t c = 0; c < n_chunks; ++c) { const float2 t = sl[(size_t)c * size + e]; v.x += t.x; v.y += t.y; }
		GP *dst = reinterpret_cast<GP *>(grad) + off + e;
		if (accumulate) { const float2 old = to_f2(*dst); v.x += old.x; v.y += old.y; }
		GP o; from_f2(o, v);
		*dst = o;
	}
}


NGP_API int ngp_hash_encode_fwd(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *table, const uint32_t *level_table_host,
                                void *out, int dtype, int out_layout, const uint32_t *n_valid) {
	NGP_REQUIRE(n == 0 || (pos && table && level_table_host && out), NGP_E_ARG, "ngp_hash_encode_fwd: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_hash_encode_fwd: bad dtype %d", dtype);
	if (n == 0) return 0;
	NGP_REQUIRE(pos_stride >= 3, NGP_E_ARG, "ngp_hash_encode_fwd: pos stride %u < 3", pos_stride);
	const uint32_t nblk = min(div_up(n, 256), 2048u);         // chunks per level in flight (k_hash_fwd strides over the rest)
	const dim3 block(256);
	const LevelTable lt = load_table(level_table_host);
	hipStream_t s = (hipStream_t)stream;
	// NGP_HASH_FWD_BALANCE=0 selects the round-1 map (probe hook); NGP_HASH_FWD_LIGHT = relative cost of a chunk of a coherent level
	// (measured, gpurun_out/r3a_*: 0 = 75 us per training launch, 1 = 114 us at light 0.12 - see fwd_map_helpers)
	static const int balance = [] { const char *e = getenv("NGP_HASH_FWD_BALANCE"); return e ? atoi(e) : 0; }();
	static const float light = [] { const char *e = getenv("NGP_HASH_FWD_LIGHT"); return e ? (float)atof(e) : 0.12f; }();
	static const float help = [] { const char *e = getenv("NGP_HASH_FWD_HELP"); return e ? (float)atof(e) : 0.15f; }();
	if (balance) {
		const FwdMap map = balance == 3 ? fwd_map_light_aside(lt, nblk) : balance == 2 ? fwd_map_helpers(lt, nblk, help) : fwd_map_balanced(lt, nblk, light);
		const dim3 grid(8 * map.slots);
#define GO(T, L) NGP_LAUNCH((k_hash_fwd_bal<T, L>), grid, block, 0, s, n, pos, pos_stride, (const T *)table, lt, (T *)out, nblk, n_valid, map)
		if (dtype == NGP_F32) { if (out_layout == NGP_LAYOUT_SOA) GO(float, NGP_LAYOUT_SOA); else GO(float, NGP_LAYOUT_AOS); }
		else { if (out_layout == NGP_LAYOUT_SOA) GO(__half, NGP_LAYOUT_SOA); else GO(__half, NGP_LAYOUT_AOS); }
#undef GO
		NGP_LAUNCH_CHECK("ngp_hash_encode_fwd");
		return 0;
	}
	const dim3 grid(16 * nblk);
#define GO(T, L) NGP_LAUNCH((k_hash_fwd<T, L>), grid, block, 0, s, n, pos, pos_stride, (const T *)table, lt, (T *)out, nblk, n_valid)
	if (dtype == NGP_F32) { if (out_layout == NGP_LAYOUT_SOA) GO(float, NGP_LAYOUT_SOA); else GO(float, NGP_LAYOUT_AOS); }
	else { if (out_layout == NGP_LAYOUT_SOA) GO(__half, NGP_LAYOUT_SOA); else GO(__half, NGP_LAYOUT_AOS); }
#undef GO
	NGP_LAUNCH_CHECK("ngp_hash_encode_fwd");
	return 0;
}

NGP_API int ngp_hash_encode_fwd_dydx(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *table, const uint32_t *level_table_host,
                                     void *out, int dtype, int out_layout, const uint32_t *n_valid, float *dy_dx) {
	NGP_REQUIRE(n == 0 || (pos && table && level_table_host && out && dy_dx), NGP_E_ARG, "ngp_hash_encode_fwd_dydx: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_hash_encode_fwd_dydx: bad dtype %d", dtype);
	NGP_REQUIRE(((uintptr_t)dy_dx & 7) == 0, NGP_E_ALIGN, "ngp_hash_encode_fwd_dydx: dy_dx must be 8-byte aligned");
	if (n == 0) return 0;
	NGP_REQUIRE(pos_stride >= 3, NGP_E_ARG, "ngp_hash_encode_fwd_dydx: pos stride %u < 3", pos_stride);
	const uint32_t nblk = min(div_up(n, 256), 2048u);
	const dim3 grid(16 * nblk), block(256);
	const LevelTable lt = load_table(level_table_host);
	hipStream_t s = (hipStream_t)stream;
#define GO(T, L) NGP_LAUNCH((k_hash_fwd_dydx<T, L>), grid, block, 0, s, n, pos, pos_stride, (const T *)table, lt, (T *)out, nblk, n_valid, dy_dx)
	if (dtype == NGP_F32) { if (out_layout == NGP_LAYOUT_SOA) GO(float, NGP_LAYOUT_SOA); else GO(float, NGP_LAYOUT_AOS); }
	else { if (out_layout == NGP_LAYOUT_SOA) GO(__half, NGP_LAYOUT_SOA); else GO(__half, NGP_LAYOUT_AOS); }
#undef GO
	NGP_LAUNCH_CHECK("ngp_hash_encode_fwd_dydx");
	return 0;
}
NGP_API int ngp_hash_encode_bwd_input(void *stream, uint32_t n, const void *dLdy, int dtype, int in_layout, const float *dy_dx, float *dLdx, const uint32_t *n_valid) {
	NGP_REQUIRE(n == 0 || (dLdy && dy_dx && dLdx), NGP_E_ARG, "ngp_hash_encode_bwd_input: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_hash_encode_bwd_input: bad dtype %d", dtype);
	NGP_REQUIRE(((uintptr_t)dy_dx & 7) == 0, NGP_E_ALIGN, "ngp_hash_encode_bwd_input: dy_dx must be 8-byte aligned");
	if (n == 0) return 0;
	const dim3 grid(div_up(n * 3u, 256)), block(256);
	hipStream_t s = (hipStream_t)stream;
#define GO(T, L) NGP_LAUNCH((k_hash_bwd_input<T, L>), grid, block, 0, s, n, (const T *)dLdy, dy_dx, dLdx, n_valid)
	if (dtype == NGP_F32) { if (in_layout == NGP_LAYOUT_SOA) GO(float, NGP_LAYOUT_SOA); else GO(float, NGP_LAYOUT_AOS); }
	else { if (in_layout == NGP_LAYOUT_SOA) GO(__half, NGP_LAYOUT_SOA); else GO(__half, NGP_LAYOUT_AOS); }
#undef GO
	NGP_LAUNCH_CHECK("ngp_hash_encode_bwd_input");
	return 0;
}


// ---------------------------------------------------------------------------------------------------------------- second-order terms (hash-grid SDF network, r3)
// A network that is trained on its own input gradient (NeuS' eikonal term and normal-fed colour network over a hash-grid SDF, BASELINE configs[4]) back-propagates
// through g = dL/dx = sum_k dLdy_k * dy_k/dx (k_hash_bwd_input).  For an upstream gradient u = d loss / d g [n,3] that needs
//   (i)  d loss / d dLdy[i][k]      = sum_d u[i][d] * dy_dx[i][d][k]                                   (k_hash_bwd_input_bwd_dy: the transposed contraction)
//   (ii) d loss / d table[e][f]    += dLdy[i][2l+f] * sum_d u[i][d] * d w_c(x_i) / d x_d               for every corner c of (sample i, level l) that lands on entry e
//        (k_hash_bwd_input_bwd_grid: the table scatter of k_hash_bwd with the interpolation weight replaced by its directional derivative along u; d w_c / d x_d is
//        the weight the dy_dx branch uses - scale * w(first other dim) * w(second other dim), positive for the corner on the right of dimension d, negative on the left)
// The reference has neither (its dy_dx branch is never enabled); tiny-cuda-nn's kernel_grid_backward_input_backward_grid computes (ii).  The mixed second derivative
// w.r.t. the position itself is not produced (NeuS' sample positions carry no parameters).  Few samples (512 rays x 128), fp32 float atomics: not a hot-path kernel.
template <typename T>
__global__ __launch_bounds__(256) void k_hash_bwd_input_bwd_dy(uint32_t n, const float *__restrict__ u, const float *__restrict__ dy_dx, T *__restrict__ ddy) {
	using P = typename Pair<T>::type;
	const uint32_t t = blockIdx.x * 256u + threadIdx.x, i = t >> 4, l = t & 15u;
	if (i >= n) return;
	float2 a = make_float2(0.f, 0.f);
#pragma unroll
	for (uint32_t d = 0; d < 3; ++d) {
		const float ud = u[(size_t)i * 3 + d];
		const float2 r = *reinterpret_cast<const float2 *>(dy_dx + (size_t)i * 96 + d * 32 + 2 * l);
		a.x += ud * r.x; a.y += ud * r.y;
	}
	P o; from_f2(o, a);
	reinterpret_cast<P *>(ddy)[(size_t)i * 16 + l] = o;
}
template <typename T>
__global__ __launch_bounds__(256) void k_hash_bwd_input_bwd_grid(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ dLdy, const float *__restrict__ u,
                                                                 LevelTable lt, float *__restrict__ grad, uint32_t nblk) {
	using P = typename Pair<T>::type;
	uint32_t level, chunk0; block_to_level_chunk(nblk, level, chunk0);
	const uint32_t off = lt.v[4 * level], size = lt.v[4 * level + 1], res = lt.v[4 * level + 2];
	const float scale = __uint_as_float(lt.v[4 * level + 3]);
	const bool dense = level_is_dense(size, res);
	float *gl = grad + (size_t)off * 2;
	for (uint32_t i = chunk0 * 256u + threadIdx.x; i < n; i += nblk * 256u) {
		const float2 g2 = to_f2(reinterpret_cast<const P *>(dLdy)[(size_t)i * 16 + level]);
		const float u0 = u[(size_t)i * 3], u1 = u[(size_t)i * 3 + 1], u2 = u[(size_t)i * 3 + 2];
		if ((g2.x == 0.f && g2.y == 0.f) || (u0 == 0.f && u1 == 0.f && u2 == 0.f)) continue;
		const Corner c = locate(pos, stride, i, scale);
		const float uu[3] = {u0, u1, u2};
#pragma unroll
		for (uint32_t k = 0; k < 8; ++k) {
			float wk = 0.f;
#pragma unroll
			for (uint32_t gd = 0; gd < 3; ++gd) {
				const uint32_t d0 = gd == 0 ? 1u : 0u, d1 = gd == 2 ? 1u : 2u;
				float weight = scale;
				weight *= (k >> d0) & 1u ? c.w[d0] : 1 - c.w[d0];
				weight *= (k >> d1) & 1u ? c.w[d1] : 1 - c.w[d1];
				wk += uu[gd] * ((k >> gd) & 1u ? weight : -weight);
			}
			const uint32_t idx = grid_index(size, res, dense, c.g[0] + (k & 1u), c.g[1] + ((k >> 1) & 1u), c.g[2] + (k >> 2));
			atomic_add_pair(gl + (size_t)idx * 2, make_float2(g2.x * wk, g2.y * wk));
		}
	}
}
NGP_API int ngp_hash_encode_bwd_input_bwd_dy(void *stream, uint32_t n, const float *u, const float *dy_dx, void *ddLdy, int dtype) {
	NGP_REQUIRE(n == 0 || (u && dy_dx && ddLdy), NGP_E_ARG, "ngp_hash_encode_bwd_input_bwd_dy: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_hash_encode_bwd_input_bwd_dy: bad dtype %d", dtype);
	NGP_REQUIRE(((uintptr_t)dy_dx & 7) == 0, NGP_E_ALIGN, "ngp_hash_encode_bwd_input_bwd_dy: dy_dx must be 8-byte aligned");
	if (n == 0) return 0;
	NGP_REQUIRE(n <= (1u << 27), NGP_E_ARG, "ngp_hash_encode_bwd_input_bwd_dy: n = %u too large", n);
	const dim3 grid(div_up(n * 16u, 256)), block(256);
	hipStream_t s = (hipStream_t)stream;
	if (dtype == NGP_F32) NGP_LAUNCH((k_hash_bwd_input_bwd_dy<float>), grid, block, 0, s, n, u, dy_dx, (float *)ddLdy);
	else NGP_LAUNCH((k_hash_bwd_input_bwd_dy<__half>), grid, block, 0, s, n, u, dy_dx, (__half *)ddLdy);
	NGP_LAUNCH_CHECK("ngp_hash_encode_bwd_input_bwd_dy");
	return 0;
}
NGP_API int ngp_hash_encode_bwd_input_bwd_grid(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *dLdy, int dtype, const float *u,
                                               const uint32_t *level_table_host, float *grad, uint64_t n_params) {
	NGP_REQUIRE(n == 0 || (pos && dLdy && u && level_table_host && grad), NGP_E_ARG, "ngp_hash_encode_bwd_input_bwd_grid: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_hash_encode_bwd_input_bwd_grid: bad dtype %d", dtype);
	if (n == 0) return 0;
	NGP_REQUIRE(pos_stride >= 3, NGP_E_ARG, "ngp_hash_encode_bwd_input_bwd_grid: pos stride %u < 3", pos_stride);
	const LevelTable lt = load_table(level_table_host);
	NGP_REQUIRE((uint64_t)(lt.v[4 * 15] + lt.v[4 * 15 + 1]) * 2u <= n_params, NGP_E_ARG, "ngp_hash_encode_bwd_input_bwd_grid: level table needs %llu parameters, grad has %llu",
	            (unsigned long long)(lt.v[4 * 15] + lt.v[4 * 15 + 1]) * 2ull, (unsigned long long)n_params);
	const uint32_t nblk = min(div_up(n, 256), 2048u);
	const dim3 grid(16 * nblk), block(256);
	hipStream_t s = (hipStream_t)stream;
	if (dtype == NGP_F32) NGP_LAUNCH((k_hash_bwd_input_bwd_grid<float>), grid, block, 0, s, n, pos, pos_stride, (const float *)dLdy, u, lt, grad, nblk);
	else NGP_LAUNCH((k_hash_bwd_input_bwd_grid<__half>), grid, block, 0, s, n, pos, pos_stride, (const __half *)dLdy, u, lt, grad, nblk);
	NGP_LAUNCH_CHECK("ngp_hash_encode_bwd_input_bwd_grid");
	return 0;
}

// method: 0 = owner-computes LDS scatter (default), 1 = one global atomic per corner (the reference's scheme).  NGP_HASH_BWD_ATOMICS=1 selects 1.
static int hash_bwd_method() {
	static int m = -1;
	if (m < 0) { const char *e = getenv("NGP_HASH_BWD_ATOMICS"); m = (e && e[0] == '1') ? 1 : 0; }
	return m;
}


// ---------------------------------------------------------------------------------------------------------------- binned scatter (every level of up to 2^19 entries)
// The owner-computes scan above redoes every sample's index arithmetic once per slice owner (32x per level, ~220 instructions each) — it is
// VALU-bound at ~0.45 ms per 2^18-sample batch.  With a workspace the levels take this two-phase path instead:
//   A  records: the eight (entry, weight*gradient) contributions of a (sample, level) are computed ONCE and appended to the record list of the bin the
//      entry lives in (64 bins per level).  Slots are handed out by an LDS histogram per workgroup plus ONE global integer atomic per
//      (workgroup, bin) — ~10^5 global atomics per batch instead of 3*10^7.
//        k_bin_records       (fine levels)   one thread per (sample, level);
//        k_bin_records_runs  (coarse levels, **r2b**) one thread per EIGHT CONSECUTIVE samples: the samples of a ray are consecutive in the batch and a cell of a
//                            level with res <= 300 is 3-40 marching steps long, so the thread sums the runs that share a cell in registers and emits one set of
//                            eight records per run - 2.2 instead of 10 levels' worth of records on the ngp_base.py batch, and the dense levels (whose whole
//                            table is a few thousand entries hit by 2 M contributions) fit the same machinery: no owner-computes scan, no partial slabs.
//   B  k_bin_accumulate: one workgroup per bin streams its records (coalesced reads) into 64-bit INTEGER accumulators in LDS
//      (ds_add_u64: 16.6 cycles per wave instruction vs 194 for ds_add_f32) and writes the bin's entries of the gradient with plain stores.
// A record is a 16-bit slot inside the bin plus the contribution, kept as two streams (structure of arrays: 2 + 4 bytes for fp16 gradients on the fine levels,
// 2 + 8 for fp32 and for every run record - the 8-byte {u32 index, half2} records of round 1 moved a third more bytes).
//   fp16 dL/dy, fine levels: the contribution is stored as fp16 after scaling by the power of two that maps the level's max |dL/dy| into [2^13, 2^14): every
//     fp16 value is a multiple of 2^-24, so value * 2^24 is an exact integer < 2^39 and the sum of up to 2^21 records cannot overflow 63 bits.  Each contribution
//     is rounded once (2^-11 relative, like the `(__half)(grad*weight)` of HashEncode.h:345).
//   fp32 records (fp32 dL/dy - ngp_base.py - and all run records): converted to fixed point at 2^38 / max|dL/dy| (a 64-bit sum per feature): fp32-exact for every
//     contribution within 2^-14 of the level's largest, and still 2^-10-relative 14 binades further down.
// In both cases the accumulation itself is EXACT and order-independent => bit-reproducible gradients (the reference's atomics round after every add, in
// random order; the run sums are fp32 sums in sample order inside one thread: deterministic too).  A bin that overflows its record capacity (pathological
// clustering) spills to one shared list that the bin's owner scans before it writes - no float atomics anywhere, still deterministic, just slow in that corner.
#define BIN_BITS 13u
#define BIN_ENTRIES (1u << BIN_BITS)
#define BINS_PER_LEVEL 64u
#define BIN_LEVEL_MAX (BIN_ENTRIES * BINS_PER_LEVEL)                  // 2^19 entries: the largest level the bins cover
#define RUN_RES_MAX 300u                                              // levels up to this resolution go through k_bin_records_runs
static_assert(RUN_RES_MAX == NGP_DP_COARSE_RES_MAX, "the data-parallel bucket boundary (ngp_dp_plan) is the boundary between the run-combined and the fine levels");
// One cursor per bin.  (r4 tried eight sub-lists with a cursor each, on the theory that same-address atomics queue behind each other: no change for the record kernels,
// +5 us for the accumulate's eight-way gather - the cost of these reservations is their NUMBER, see the edge records below.  CUR_SUBS is kept as the switch.)
#define CUR_SUBS 1u
#define N_CURSORS (16u * BINS_PER_LEVEL * CUR_SUBS)                       // u32 cursors of a workspace: [16][64][CUR_SUBS]
#define N_ZEROED (N_CURSORS + 16u)                                        // ... followed by the unit queue head of k_bin_accumulate2 (+ spare words): zeroed together with the cursors every step
struct BinPlan { uint32_t level[16]; uint32_t n_levels; uint32_t cap; uint32_t spill_cap; };   // binned levels, records per bin, entries of the spill list
struct LevelSel { uint32_t hl[16]; };                                  // the binned-level ordinals one launch works on (blockIdx.y, or blockIdx.x / 64)
struct SpillEntry { uint32_t key /* binned-level ordinal << 19 | entry */; float x, y; };       // value in record units (fp16 records: scaled)

// entry -> (bin, slot inside the bin).  A full 2^19-entry hashed level: bin = the entry's 8192-entry slice (pseudo-random entries: balanced; contiguous write-out).
// Any smaller level (the dense levels, small hashed tables): groups of 8 entries are dealt round-robin to the 64 bins, so the spatially coherent dense indices
// (x + y*res + z*res^2: a batch lives in a few z-slabs) spread evenly as well.
__device__ __forceinline__ uint32_t bin_of(uint32_t e, bool il) { return il ? (e >> 3) & 63u : e >> BIN_BITS; }
__device__ __forceinline__ uint32_t local_of(uint32_t e, bool il) { return il ? ((e >> 9) << 3) | (e & 7u) : e & (BIN_ENTRIES - 1u); }
__device__ __forceinline__ uint32_t entry_of(uint32_t bin, uint32_t local, bool il) { return il ? ((local >> 3) << 9) | (bin << 3) | (local & 7u) : (bin << BIN_BITS) | local; }
// record streams of (binned level hl, bin): every level owns 64 * cap * 8 bytes of the value area whatever its record type
// (cap = capacity of ONE sub-list; list = bin * CUR_SUBS + sub)
template <typename RV> __device__ __forceinline__ RV *rec_val_at(void *base, uint32_t hl, uint32_t list, uint32_t cap) {
	return reinterpret_cast<RV *>(reinterpret_cast<char *>(base) + (size_t)hl * BINS_PER_LEVEL * CUR_SUBS * cap * 8u) + (size_t)list * cap;
}
__device__ __forceinline__ uint16_t *rec_idx_at(uint16_t *base, uint32_t hl, uint32_t list, uint32_t cap) { return base + ((size_t)hl * BINS_PER_LEVEL * CUR_SUBS + list) * cap; }
// Reading a bin back: its eight sub-lists laid end to end in units of K records (`groups`) plus the < K leftover records of every sub-list (`tails`).
struct SubLists { uint32_t cnt[CUR_SUBS], gstart[CUR_SUBS + 1], tstart[CUR_SUBS + 1]; bool over; };
__device__ __forceinline__ SubLists sub_lists(const uint32_t *__restrict__ cur /* the bin's CUR_SUBS cursors */, uint32_t cap, uint32_t K) {
	SubLists m; m.over = false; m.gstart[0] = 0u; m.tstart[0] = 0u;
#pragma unroll
	for (uint32_t k = 0; k < CUR_SUBS; ++k) {
		const uint32_t raw = cur[k];
		m.over |= raw > cap;
		m.cnt[k] = min(raw, cap);
		m.gstart[k + 1] = m.gstart[k] + m.cnt[k] / K;
		m.tstart[k + 1] = m.tstart[k] + m.cnt[k] % K;
	}
	return m;
}
// flat group index r -> index into the bin's sub-list layout in units of K records (sub-list k begins k * gcap groups in); static indexing only (the tables stay in registers)
__device__ __forceinline__ uint32_t sub_group(const SubLists &m, uint32_t r, uint32_t gcap) {
	uint32_t k = 0, s0 = 0;
#pragma unroll
	for (uint32_t j = 1; j < CUR_SUBS; ++j) if (r >= m.gstart[j]) { k = j; s0 = m.gstart[j]; }
	return k * gcap + (r - s0);
}
// flat leftover index t -> record index in the bin's sub-list layout (sub-list k begins k * cap records in, its leftovers follow its cnt / K * K grouped records)
__device__ __forceinline__ uint32_t sub_tail(const SubLists &m, uint32_t t, uint32_t cap, uint32_t K) {
	uint32_t k = 0, s0 = 0, full = m.cnt[0] / K * K;
#pragma unroll
	for (uint32_t j = 1; j < CUR_SUBS; ++j) if (t >= m.tstart[j]) { k = j; s0 = m.tstart[j]; full = m.cnt[j] / K * K; }
	return k * cap + full + (t - s0);
}

// Largest |dL/dy| of every level (the scale of the fixed-point accumulation).  Every workgroup writes the maximum of its share of the samples to ABSMAX_PARTS
// partial slots per level - no atomics (same-address global atomics retire one at a time at the L2: 8192 of them on 16 addresses took ~90 us), nothing to zero
// beforehand; the consumers take the maximum of a level's partials with scalar loads (level_absmax).  The pass also zeroes the record cursors and the spill
// count for the kernels behind it in the stream (that was a separate 5 us memset launch).
#define ABSMAX_PARTS NGP_ABSMAX_PARTS
#define ABSMAX_OWN_PARTS 64u                                                    // partials the scatter's own pass writes (its grid); the remaining slots are zeroed by it
__device__ __forceinline__ uint32_t level_absmax(const uint32_t *__restrict__ parts, uint32_t level) {      // positive floats order like their bit patterns
	const uint4 q = reinterpret_cast<const uint4 *>(parts + level * ABSMAX_PARTS)[threadIdx.x & 63u];         // four partials per lane + a wavefront reduction (called by full wavefronts, at kernel entry)
	uint32_t m = max(max(q.x, q.y), max(q.z, q.w));
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
	return m;
}
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void k_level_absmax(uint32_t n, const T *__restrict__ dLdy, uint32_t *__restrict__ parts, const uint32_t *__restrict__ n_valid,
                                                      uint32_t *__restrict__ cursors, uint32_t *__restrict__ spill_count) {
	using P = typename Pair<T>::type;
	const uint32_t level = blockIdx.y;
	if (blockIdx.x == 0) {                                               // this level's sixteenth of the cursors, spill count
		for (uint32_t j = threadIdx.x; j < N_CURSORS / 16u; j += 256u) cursors[level * (N_CURSORS / 16u) + j] = 0u;
		if (level == 0 && threadIdx.x < N_ZEROED - N_CURSORS) cursors[N_CURSORS + threadIdx.x] = 0u;
		if (level == 0 && threadIdx.x == BINS_PER_LEVEL) *spill_count = 0u;
	}
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	const P *dy = reinterpret_cast<const P *>(dLdy);
	float m = 0.f;
	const uint32_t step = gridDim.x * 256u;
	uint32_t i = blockIdx.x * 256u + threadIdx.x;
	for (; i + 3 * step < lim; i += 4 * step) {                        // four independent loads in flight
		float2 g[4];
#pragma unroll
		for (int u = 0; u < 4; ++u) g[u] = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)level * n + i + u * step] : dy[(size_t)(i + u * step) * 16 + level]);
#pragma unroll
		for (int u = 0; u < 4; ++u) m = fmaxf(m, fmaxf(fabsf(g[u].x), fabsf(g[u].y)));
	}
	for (; i < lim; i += step) {
		const float2 g = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)level * n + i] : dy[(size_t)i * 16 + level]);
		m = fmaxf(m, fmaxf(fabsf(g.x), fabsf(g.y)));
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
	__shared__ float wave_max[4];
	if ((threadIdx.x & 63u) == 0) wave_max[threadIdx.x >> 6] = m;
	__syncthreads();
	if (threadIdx.x == 0) {
		m = fmaxf(fmaxf(wave_max[0], wave_max[1]), fmaxf(wave_max[2], wave_max[3]));
		parts[level * ABSMAX_PARTS + blockIdx.x] = (m > 0.f) ? __float_as_uint(m) : 0u;     // (NaN -> 0: a level without a usable gradient is skipped, as before)
	}
	if (threadIdx.x >= 1 && threadIdx.x < ABSMAX_PARTS / ABSMAX_OWN_PARTS) parts[level * ABSMAX_PARTS + blockIdx.x + threadIdx.x * ABSMAX_OWN_PARTS] = 0u;   // the slots of the (larger) fused producer's grid
}

// Records are staged in LDS grouped by bin and written out run by run: a wave then stores 64 consecutive records (full lines) instead of 64
// scattered words.  The scattered version was bound by the L2 request rate (2.4e7 partial-line writes ~ one per clock per channel), not by bytes.
template <typename T> struct RecVal;
template <> struct RecVal<__half> { using type = __half2; };
template <> struct RecVal<float> { using type = float2; };
// 512 samples per workgroup: 33 KiB (fp16) / 49 KiB (fp32) of LDS, so 3-4 workgroups share a CU and one workgroup's serial phases (loads -> LDS histogram -> the
// wave-0 reservation with its global atomics -> staging -> copy-out, five barriers) hide behind the others'.  With 1024 samples (99 KiB for fp32: one workgroup
// per CU) the fp32 pass took 185 us for 230 MB of records.
#define BIN_WG 512u
template <typename T> constexpr uint32_t bin_stage_bytes() { return BIN_WG * 8u * (uint32_t)(sizeof(typename RecVal<T>::type) + 4u) + 3u * BINS_PER_LEVEL * 4u; }

// the eight entries of the cell whose lowest corner is (gx, gy, gz): level-wide indices (HashEncode.h:68-94)
__device__ __forceinline__ void cell_entries(uint32_t size, uint32_t res, bool dense, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t idx[8]) {
	if (dense) {
		const uint32_t y0 = gy * res, z0 = gz * res * res;
#pragma unroll
		for (uint32_t q = 0; q < 8; ++q) {
			uint32_t e = (gx + (q & 1u)) + (y0 + ((q & 2u) ? res : 0u)) + (z0 + ((q & 4u) ? res * res : 0u));
			if (e >= size) { e -= size; if (e >= size) e %= size; }              // wraps only at the +1 boundary corner
			idx[q] = e;
		}
	} else {
		const uint32_t ty0 = gy * 19349663u, tz0 = gz * 83492791u;
#pragma unroll
		for (uint32_t q = 0; q < 8; ++q) idx[q] = ((gx + (q & 1u)) ^ (ty0 + ((q & 2u) ? 19349663u : 0u)) ^ (tz0 + ((q & 4u) ? 83492791u : 0u))) & (size - 1u);
	}
}

template <typename T, int LAYOUT>
__global__ __launch_bounds__(BIN_WG) void k_bin_records(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ dLdy, LevelTable lt, BinPlan bp, LevelSel sel,
                                                      const uint32_t *__restrict__ absmax_bits, uint32_t *__restrict__ cursors, void *__restrict__ rec_val,
                                                      uint16_t *__restrict__ rec_idx, uint32_t *__restrict__ spill_count, SpillEntry *__restrict__ spill,
                                                      const uint32_t *__restrict__ n_valid) {
	using P = typename Pair<T>::type;
	using RV = typename RecVal<T>::type;
	extern __shared__ __attribute__((aligned(16))) uint32_t bin_smem[];
	RV *stage_val = reinterpret_cast<RV *>(bin_smem);                                   // [4096] contributions, grouped by bin
	uint32_t *stage_idx = bin_smem + BIN_WG * 8u * (sizeof(RV) / 4u);                     // [4096] level-wide entry indices
	uint32_t *cnt = stage_idx + BIN_WG * 8u, *base = cnt + BINS_PER_LEVEL, *loff = base + BINS_PER_LEVEL;
	const uint32_t hl = sel.hl[blockIdx.y], level = bp.level[hl], sub = blockIdx.x % CUR_SUBS;
	const uint32_t size = lt.v[4 * level + 1], res = lt.v[4 * level + 2];
	const float scale = __uint_as_float(lt.v[4 * level + 3]);
	const bool dense = level_is_dense(size, res), il = size < BIN_LEVEL_MAX;
	const uint32_t amax = level_absmax(absmax_bits, level);
	const float vs = sizeof(T) == 2 ? bin_scale(amax) : (amax ? 1.0f : 0.f);       // fp32 records are stored unscaled
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	if (vs == 0.f || blockIdx.x * BIN_WG >= lim) return;                // uniform exit
	if (threadIdx.x < BINS_PER_LEVEL) cnt[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t i = blockIdx.x * BIN_WG + threadIdx.x;
	const P *dy = reinterpret_cast<const P *>(dLdy);
	uint32_t idx[8], rank[8]; RV val[8];
	bool live = false;
	if (i < lim) {
		const float2 g2 = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)level * n + i] : dy[(size_t)i * 16 + level]);
		live = (g2.x != 0.f || g2.y != 0.f);
		if (live) {
			const Corner c = locate(pos, stride, i, scale);
			cell_entries(size, res, dense, c.g[0], c.g[1], c.g[2], idx);
			const float gx = g2.x * vs, gy = g2.y * vs;
#pragma unroll
			for (uint32_t q = 0; q < 8; ++q) {
				const float w = ((q & 1u) ? c.w[0] : 1 - c.w[0]) * ((q & 2u) ? c.w[1] : 1 - c.w[1]) * ((q & 4u) ? c.w[2] : 1 - c.w[2]);
				from_f2(val[q], make_float2(gx * w, gy * w));
				rank[q] = atomicAdd(&cnt[bin_of(idx[q], il)], 1u);
			}
		}
	}
	__syncthreads();
	if (threadIdx.x < BINS_PER_LEVEL) {                                    // wave 0: global run reservation + exclusive prefix of the counts (LDS offsets of the runs)
		const uint32_t c = cnt[threadIdx.x];
		base[threadIdx.x] = c ? atomicAdd(&cursors[(hl * BINS_PER_LEVEL + threadIdx.x) * CUR_SUBS + sub], c) : 0u;
		uint32_t x = c;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if ((int)threadIdx.x >= o) x += y; }
		loff[threadIdx.x] = x - c;
	}
	__syncthreads();
	if (live) {
#pragma unroll
		for (uint32_t q = 0; q < 8; ++q) {
			const uint32_t slot = loff[bin_of(idx[q], il)] + rank[q];
			stage_val[slot] = val[q]; stage_idx[slot] = idx[q];
		}
	}
	__syncthreads();
	const uint32_t total = loff[BINS_PER_LEVEL - 1] + cnt[BINS_PER_LEVEL - 1];
	for (uint32_t p = threadIdx.x; p < total; p += BIN_WG) {
		const uint32_t e = stage_idx[p];
		const RV v = stage_val[p];
		const uint32_t bin = bin_of(e, il), slot = base[bin] + (p - loff[bin]);
		if (slot < bp.cap) {
			rec_val_at<RV>(rec_val, hl, bin * CUR_SUBS + sub, bp.cap)[slot] = v; rec_idx_at(rec_idx, hl, bin * CUR_SUBS + sub, bp.cap)[slot] = (uint16_t)local_of(e, il);
		} else {                                                        // bin full (pathological clustering): the shared spill list, scanned by the bin's owner
			const uint32_t k = atomicAdd(spill_count, 1u);
			if (k < bp.spill_cap) { const float2 f = to_f2(v); spill[k] = SpillEntry{(hl << 19) | e, f.x, f.y}; }
		}
	}
}

// Coarse levels: one thread per RUN_K consecutive samples, runs of samples in one cell summed in registers, one set of eight fp32 records per run.  The number
// of records a workgroup produces is data dependent (2048 samples: 2048 on the coarsest levels, 16384 for scattered positions), so the pass runs twice over
// the registers: COUNT (LDS histogram of the bins) - reservation - PLACE.  Up to RUN_STAGE records are staged in LDS and leave as full lines; whatever exceeds
// that (scattered positions only) is stored to its reserved slot directly.
#define RUN_K 8u
#define RUN_WG 256u
#define RUN_STAGE 3072u
static uint32_t run_stage_bytes(uint32_t stage) { return stage * 12u + 4u * BINS_PER_LEVEL * 4u; }

template <typename T, int LAYOUT, int OCC /* waves per SIMD the register budget is held to: 4 = natural (114 VGPRs), 5 = all 1280 workgroups of a 2^18-sample batch resident at once (probe) */>
__global__ __launch_bounds__(RUN_WG, OCC) void k_bin_records_runs(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ dLdy, LevelTable lt, BinPlan bp, LevelSel sel,
                                                           const uint32_t *__restrict__ absmax_bits, uint32_t *__restrict__ cursors, void *__restrict__ rec_val,
                                                           uint16_t *__restrict__ rec_idx, uint32_t *__restrict__ spill_count, SpillEntry *__restrict__ spill,
                                                           const uint32_t *__restrict__ n_valid, uint32_t stage /* records of LDS staging */) {
	using P = typename Pair<T>::type;
	extern __shared__ __attribute__((aligned(16))) uint32_t bin_smem[];
	float2 *stage_val = reinterpret_cast<float2 *>(bin_smem);                             // [stage]
	uint32_t *stage_idx = bin_smem + stage * 2u;                                          // [stage] level-wide entry indices
	uint32_t *cnt = stage_idx + stage, *base = cnt + BINS_PER_LEVEL, *loff = base + BINS_PER_LEVEL, *cnt2 = loff + BINS_PER_LEVEL;
	const uint32_t hl = sel.hl[blockIdx.y], level = bp.level[hl], sub = blockIdx.x % CUR_SUBS;
	const uint32_t size = lt.v[4 * level + 1], res = lt.v[4 * level + 2];
	const float scale = __uint_as_float(lt.v[4 * level + 3]);
	const bool dense = level_is_dense(size, res), il = size < BIN_LEVEL_MAX;
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	if (level_absmax(absmax_bits, level) == 0u || blockIdx.x * RUN_WG * RUN_K >= lim) return;          // uniform exit
	if (threadIdx.x < BINS_PER_LEVEL) { cnt[threadIdx.x] = 0; cnt2[threadIdx.x] = 0; }
	__syncthreads();
	const uint32_t first = (blockIdx.x * RUN_WG + threadIdx.x) * RUN_K;
	const P *dy = reinterpret_cast<const P *>(dLdy);
	uint32_t cell[RUN_K][3]; float frac[RUN_K][3]; float2 gk[RUN_K];
	{
		float px[RUN_K][3];
		if (first + RUN_K <= lim && stride == 3) {
			const float4 *p4 = reinterpret_cast<const float4 *>(pos + (size_t)first * 3);   // 24 floats, 16-byte aligned (first % 8 == 0)
			float4 v[6];
#pragma unroll
			for (int r = 0; r < 6; ++r) v[r] = p4[r];
			const float *f = reinterpret_cast<const float *>(v);
#pragma unroll
			for (uint32_t k = 0; k < RUN_K; ++k) { px[k][0] = f[3 * k]; px[k][1] = f[3 * k + 1]; px[k][2] = f[3 * k + 2]; }
#pragma unroll
			for (uint32_t k = 0; k < RUN_K; ++k) gk[k] = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)level * n + first + k] : dy[(size_t)(first + k) * 16 + level]);
		} else {
#pragma unroll
			for (uint32_t k = 0; k < RUN_K; ++k) {
				const uint32_t i = first + k;
				if (i < lim) {
					px[k][0] = pos[(size_t)i * stride]; px[k][1] = pos[(size_t)i * stride + 1]; px[k][2] = pos[(size_t)i * stride + 2];
					gk[k] = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)level * n + i] : dy[(size_t)i * 16 + level]);
				} else { px[k][0] = px[k][1] = px[k][2] = 0.f; gk[k] = make_float2(0.f, 0.f); }
			}
		}
#pragma unroll
		for (uint32_t k = 0; k < RUN_K; ++k)
#pragma unroll
			for (int d = 0; d < 3; ++d) { const float p = px[k][d] * scale + 0.5f; const float fl = floorf(p); cell[k][d] = (uint32_t)(int)fl; frac[k][d] = p - fl; }   // pos_fract, HashEncode.h:106-115
	}
	// one sweep over the thread's samples; emit(q, entry, x, y) is called for the eight corners of every finished run
	auto sweep = [&](auto emit) {
		bool open = false;
		uint32_t key[3] = {0u, 0u, 0u};
		float ax[8], ay[8];
		auto flush = [&]() {
			uint32_t idx[8];
			cell_entries(size, res, dense, key[0], key[1], key[2], idx);
#pragma unroll
			for (uint32_t q = 0; q < 8; ++q) emit(idx[q], ax[q], ay[q]);
		};
#pragma unroll
		for (uint32_t k = 0; k < RUN_K; ++k) {
			if (gk[k].x == 0.f && gk[k].y == 0.f) continue;        // zero rows (padding) add exact zeros in the reference: skipped, they do not end a run either
			if (open && !(cell[k][0] == key[0] && cell[k][1] == key[1] && cell[k][2] == key[2])) { flush(); open = false; }
			if (!open) {
				open = true; key[0] = cell[k][0]; key[1] = cell[k][1]; key[2] = cell[k][2];
#pragma unroll
				for (uint32_t q = 0; q < 8; ++q) { ax[q] = 0.f; ay[q] = 0.f; }
			}
			const float x1 = frac[k][0], x0 = 1 - x1, y1 = frac[k][1], y0 = 1 - y1, z1 = frac[k][2], z0 = 1 - z1;
#pragma unroll
			for (uint32_t q = 0; q < 8; ++q) {
				const float w = ((q & 1u) ? x1 : x0) * ((q & 2u) ? y1 : y0) * ((q & 4u) ? z1 : z0);                 // the reference's x, y, z multiplication order
				ax[q] += gk[k].x * w; ay[q] += gk[k].y * w;
			}
		}
		if (open) flush();
	};
	sweep([&](uint32_t e, float, float) { atomicAdd(&cnt[bin_of(e, il)], 1u); });
	__syncthreads();
	if (threadIdx.x < BINS_PER_LEVEL) {                                    // wave 0: global reservation + exclusive prefix of the counts
		const uint32_t c = cnt[threadIdx.x];
		base[threadIdx.x] = c ? atomicAdd(&cursors[(hl * BINS_PER_LEVEL + threadIdx.x) * CUR_SUBS + sub], c) : 0u;
		uint32_t x = c;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if ((int)threadIdx.x >= o) x += y; }
		loff[threadIdx.x] = x - c;
	}
	__syncthreads();
	const uint32_t cap = bp.cap;
	auto store = [&](uint32_t e, uint32_t bin, uint32_t slot, float2 v) {
		if (slot < cap) { rec_val_at<float2>(rec_val, hl, bin * CUR_SUBS + sub, cap)[slot] = v; rec_idx_at(rec_idx, hl, bin * CUR_SUBS + sub, cap)[slot] = (uint16_t)local_of(e, il); }
		else { const uint32_t k = atomicAdd(spill_count, 1u); if (k < bp.spill_cap) spill[k] = SpillEntry{(hl << 19) | e, v.x, v.y}; }
	};
	sweep([&](uint32_t e, float x, float y) {
		const uint32_t bin = bin_of(e, il), rank = atomicAdd(&cnt2[bin], 1u), p = loff[bin] + rank;
		if (p < stage) { stage_val[p] = make_float2(x, y); stage_idx[p] = e; }
		else store(e, bin, base[bin] + rank, make_float2(x, y));
	});
	__syncthreads();
	const uint32_t total = min(loff[BINS_PER_LEVEL - 1] + cnt[BINS_PER_LEVEL - 1], stage);
	for (uint32_t p = threadIdx.x; p < total; p += RUN_WG) {
		const uint32_t e = stage_idx[p];
		const uint32_t bin = bin_of(e, il);
		store(e, bin, base[bin] + (p - loff[bin]), stage_val[p]);
	}
}

// value of one record in the accumulator's integer unit.  fp16 records: multiples of 2^-24 (exact).  fp32 records: fixed point at `s32` = 2^38 / (binade of the level's largest |dL/dy|): a contribution of that size keeps all 24 bits of its
// fp32 significand, one 2^-16 of it still keeps 8, and a 64-bit sum of 2^21 run records of <= 8 samples cannot overflow.
__device__ __forceinline__ void rec_to_fixed(__half2 v, float, long long &ix, long long &iy) {
	const float2 f = __half22float2(v);
	ix = (long long)(f.x * 16777216.0f); iy = (long long)(f.y * 16777216.0f);
}
__device__ __forceinline__ void rec_to_fixed(float2 v, float s32, long long &ix, long long &iy) { ix = __float2ll_rn(v.x * s32); iy = __float2ll_rn(v.y * s32); }

template <typename G, typename RV>
__global__ __launch_bounds__(1024) void k_bin_accumulate(LevelTable lt, BinPlan bp, LevelSel sel, const uint32_t *__restrict__ absmax_bits, const uint32_t *__restrict__ cursors,
                                                         void *__restrict__ rec_val_base, uint16_t *__restrict__ rec_idx_base, const uint32_t *__restrict__ spill_count,
                                                         const SpillEntry *__restrict__ spill, G *__restrict__ grad, int overwrite) {
	extern __shared__ __attribute__((aligned(16))) unsigned long long iacc[];   // [BIN_ENTRIES][2] 64-bit fixed point
	using GP = typename Pair<G>::type;
	constexpr bool F32 = sizeof(RV) == 8;
	const uint32_t hl = sel.hl[blockIdx.x / BINS_PER_LEVEL], bin = blockIdx.x % BINS_PER_LEVEL, level = bp.level[hl];
	const uint32_t size = lt.v[4 * level + 1];
	const bool il = size < BIN_LEVEL_MAX;
	// slots of this bin that are entries of the level (interleaved: groups bin, bin + 64, ... of the level's ceil(size / 8) groups)
	const uint32_t groups_all = (size + 7u) >> 3;
	const uint32_t n_local = il ? (groups_all > bin ? ((groups_all - bin + 63u) >> 6) << 3 : 0u) : BIN_ENTRIES;
	const uint32_t amax = level_absmax(absmax_bits, level);
	float s32 = 0.f, inv;
	if (F32) {
		const float m = __uint_as_float(amax);
		if (m > 0.f && m < 3.0e38f) { int ex; frexpf(m, &ex); s32 = ldexpf(1.0f, 38 - ex); }
		inv = s32 > 0.f ? 1.0f / s32 : 0.f;
	} else {
		const float vs = bin_scale(amax);
		s32 = vs;                                                            // (only its zero-ness is used on this path)
		inv = vs > 0.f ? 1.0f / (vs * 16777216.0f) : 0.f;
	}
	constexpr uint32_t K = 8;
	const SubLists sl = sub_lists(cursors + (hl * BINS_PER_LEVEL + bin) * CUR_SUBS, bp.cap, K);
	const uint32_t count = sl.gstart[CUR_SUBS] * K + sl.tstart[CUR_SUBS];
	GP *dst = reinterpret_cast<GP *>(grad) + lt.v[4 * level];
	if (s32 == 0.f || count == 0) {                                      // nothing to add: an accumulating destination is left alone, an overwritten one gets its zeros
		if (overwrite) {
			GP zv; from_f2(zv, make_float2(0.f, 0.f));
			for (uint32_t e = threadIdx.x; e < n_local; e += 1024) { const uint32_t t = entry_of(bin, e, il); if (t < size) dst[t] = zv; }
		}
		return;
	}
	for (uint32_t e = threadIdx.x; e < n_local * 2; e += 1024) iacc[e] = 0ull;
	__syncthreads();
	const RV *rec_val = rec_val_at<RV>(rec_val_base, hl, bin * CUR_SUBS, bp.cap);  // cap % 8 == 0: both streams of every sub-list start 16-byte aligned; sub-list k begins k * cap records further
	const uint16_t *rec_idx = rec_idx_at(rec_idx_base, hl, bin * CUR_SUBS, bp.cap);
	auto add = [&](uint32_t local, RV v) {
		long long ix, iy; rec_to_fixed(v, s32, ix, iy);
		__hip_atomic_fetch_add(&iacc[2 * local], (unsigned long long)ix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		__hip_atomic_fetch_add(&iacc[2 * local + 1], (unsigned long long)iy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	};
	// Consecutive records of a bin come from neighbouring samples of a ray, which often still share a cell (a run that was split between two threads of the
	// record pass; a fine level's cell that is two steps long): a wavefront's 64 lanes would hit a handful of entries, and same-address ds_add_u64 serialise.  So
	// every thread takes K = 8 CONSECUTIVE records, sums runs of equal entries in registers (exact: the sums are integers) and issues one pair of LDS atomics per
	// run; neighbouring lanes are then 8 records apart.  Two trips (64 / 32 + 16 bytes per thread each) are in flight.
	struct alignas(16) VK { RV v[K]; };
	struct alignas(16) IK { uint16_t i[K]; };
	const VK *pv = reinterpret_cast<const VK *>(rec_val);
	const IK *pi = reinterpret_cast<const IK *>(rec_idx);
	const uint32_t groups = sl.gstart[CUR_SUBS], gcap = bp.cap / K;
	auto grp = [&](uint32_t r) { return sub_group(sl, r, gcap); };     // group r of the bin -> its place in the sub-list layout
	auto add_fixed = [&](uint32_t local, long long ix, long long iy) {
		__hip_atomic_fetch_add(&iacc[2 * local], (unsigned long long)ix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		__hip_atomic_fetch_add(&iacc[2 * local + 1], (unsigned long long)iy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	};
	auto run_add = [&](const VK &x, const IK &k) {
		uint32_t cur = k.i[0]; long long sx, sy; rec_to_fixed(x.v[0], s32, sx, sy);
#pragma unroll
		for (uint32_t q = 1; q < K; ++q) {
			long long ix, iy; rec_to_fixed(x.v[q], s32, ix, iy);
			if (k.i[q] == cur) { sx += ix; sy += iy; }
			else { add_fixed(cur, sx, sy); cur = k.i[q]; sx = ix; sy = iy; }
		}
		add_fixed(cur, sx, sy);
	};
	uint32_t r = threadIdx.x;
	for (; r + 1024 < groups; r += 2 * 1024) {
		const uint32_t a0 = grp(r), a1 = grp(r + 1024);
		const VK x0 = pv[a0], x1 = pv[a1]; const IK k0 = pi[a0], k1 = pi[a1];
		run_add(x0, k0); run_add(x1, k1);
	}
	for (; r < groups; r += 1024) { const uint32_t a0 = grp(r); const VK x = pv[a0]; const IK k = pi[a0]; run_add(x, k); }
	if (threadIdx.x < sl.tstart[CUR_SUBS]) {                             // the < K leftover records of every sub-list
		const uint32_t t = sub_tail(sl, threadIdx.x, bp.cap, K);
		add(rec_idx[t], rec_val[t]);
	}
	if (sl.over) {                                                        // a sub-list of this bin overflowed: its surplus records are somewhere in the shared spill list
		const uint32_t ns = min(*spill_count, bp.spill_cap);
		for (uint32_t t = threadIdx.x; t < ns; t += 1024) {
			const SpillEntry se = spill[t];
			const uint32_t e = se.key & (BIN_LEVEL_MAX - 1u);
			if ((se.key >> 19) == hl && bin_of(e, il) == bin) { RV v; from_f2(v, make_float2(se.x, se.y)); add(local_of(e, il), v); }
		}
	}
	__syncthreads();
	for (uint32_t e0 = 0; e0 < n_local; e0 += 8u * 1024u) {                // (a full bin: one trip, all eight read-modify-write loads in flight)
		GP oldv[8]; uint32_t tgt[8];
#pragma unroll
		for (uint32_t k = 0; k < 8; ++k) {
			const uint32_t e = e0 + threadIdx.x + k * 1024;
			tgt[k] = e < n_local ? entry_of(bin, e, il) : ~0u;
			if (tgt[k] >= size) tgt[k] = ~0u;
			if (!overwrite && tgt[k] != ~0u) oldv[k] = dst[tgt[k]];
		}
#pragma unroll
		for (uint32_t k = 0; k < 8; ++k) {
			if (tgt[k] == ~0u) continue;
			const uint32_t e = e0 + threadIdx.x + k * 1024;
			const long long sx = (long long)iacc[2 * e], sy = (long long)iacc[2 * e + 1];
			float2 v = make_float2((float)sx * inv, (float)sy * inv);
			if (!overwrite) { if (sx == 0 && sy == 0) continue; const float2 old = to_f2(oldv[k]); v.x += old.x; v.y += old.y; }
			GP o; from_f2(o, v);
			dst[tgt[k]] = o;
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------- edge records (r4): the fine hashed levels, fp32
// Round 3's fine levels wrote eight 10-byte records per (sample, level) - 124 MB out and back in for the six fine levels of the ngp_base.py table - through lists whose
// slots were handed out by one returning global atomic per (workgroup, bin).  Two things were wrong with that, both measured this round (profiles/r04_scatter_probes.md):
//   * the ATOMICS, not the bytes, set the record kernels' duration: returning device-scope atomics retire at ~8 G/s chip-wide whatever their addresses (eight cursors
//     per bin changed nothing), so the 196 k / 393 k reservations of a 2^18-sample batch cost 25 / 50 us of k_bin_pairs' 52 / 78 us (1024- / 512-sample workgroups);
//     the stores themselves cost 9 us, and the plain append pattern without atomics runs at 5-6 TB/s (tools/microbench_stream.py);
//   * the eight contributions of a cell share most of their bits.
// So: NO global atomics and half the bytes.
//   * The hash is x ^ y*P1 ^ z*P2 masked to 19 bits and x + 1 <= res <= 2048 touches bits 0..11 only: the two x-neighbours of a cell edge ALWAYS fall into the same
//     4096-entry slice of the level.  Bins are 4096 entries (128 per level) and ONE record carries the edge: {a = g.x*(wy*wz), b = g.y*(wy*wz), fx, slot0 | slot1 << 12},
//     16 bytes for two contributions (a*(1-fx), b*(1-fx) -> slot0; a*fx, b*fx -> slot1; the accumulate kernel multiplies: three roundings per contribution like the
//     reference's ((wx*wy)*wz)*g, in another order - each within 3 * 2^-24 of the exact product, up to 4 ulp apart (tests/test_host_cpu.py), far inside what the reference's float atomics scatter around the exact sum;
//     the accumulation itself stays exact, 64-bit integers).
//   * Every record workgroup owns a REGION of the record area (4 x its samples records): it sorts its edge records by bin in LDS (histogram, prefix, staging - as before)
//     and writes the staged block out as it is - one contiguous, fully coalesced 64 KiB store stream - plus the 129 bin offsets inside its region (u16).  Nothing is
//     reserved, nothing can overflow, the layout is deterministic.
//   * The accumulate workgroup of (level, bin) walks the W regions: eight lanes per region read that region's segment of the bin (offset table -> start, length;
//     32 records on average = one 64-byte quad of records per lane), neighbours that name the same edge are summed in registers, four LDS atomics per distinct edge.
//     64 KiB of accumulators: two workgroups per CU, one's write-out and start-up hide behind the other's record stream.
// Levels: hashed, 2^19 entries, run-combining limit < res <= 2048, fp32 dL/dy and gradient.  Everything else keeps the per-corner records above.
#define PAIR_BIN_BITS 12u
#define PAIR_BIN_ENTRIES (1u << PAIR_BIN_BITS)
#define PAIR_BINS 128u
#define PAIR_RES_MAX 2048u
#define PAIR_OFFS (PAIR_BINS + 2u)                                     // u16 offsets per region: 128 bin starts, the total, one pad (rows stay 4-byte aligned)
static_assert(PAIR_BIN_ENTRIES * PAIR_BINS == BIN_LEVEL_MAX, "the edge-record path covers the full 2^19-entry levels");
struct alignas(16) PairRec { float a, b, fx; uint32_t loc; };
#define PAIR_STAGE_RECORDS 4096u                                       // LDS staging of a record workgroup = its region: 1024 samples x 4 edge records | 512 samples x 8 single records
static uint32_t pair_stage_bytes() { return PAIR_STAGE_RECORDS * 16u + (2u * PAIR_BINS + 4u) * 4u; }
__host__ __device__ static inline uint32_t pair_region_records() { return PAIR_STAGE_RECORDS + 4u; }     // + 4: slack behind the last segment

// S samples per workgroup = one region.  Levels beyond res 2048 (aabb_scale > 1: ngp_fox.py's 2353 .. 8192) cannot pair their x-neighbours - x + 1 reaches into the
// bin bits - so they emit eight SINGLE records per sample in the same format (fx = 0, both slots the same entry: the second contribution is an exact zero and is
// skipped); S = 512 then.  T = type of dL/dy (fp16 configuration: the contributions are formed in fp32 from the fp16 gradient - at least as accurate as the fp16
// records of the per-corner path).
template <typename T, int LAYOUT, uint32_t S>
__global__ __launch_bounds__(S) void k_bin_pairs(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ dLdy, LevelTable lt, BinPlan bp, LevelSel sel,
                                                 const uint32_t *__restrict__ absmax_bits, PairRec *__restrict__ prec, uint16_t *__restrict__ poff,
                                                 uint32_t *__restrict__ spill_count, SpillEntry *__restrict__ spill, const uint32_t *__restrict__ n_valid,
                                                 uint32_t probe /* timing experiment (NGP_PAIR_PROBE; results wrong unless 0): 1 = no record stores */) {
	using P = typename Pair<T>::type;
	extern __shared__ __attribute__((aligned(16))) uint32_t bin_smem[];
	PairRec *stage = reinterpret_cast<PairRec *>(bin_smem);                                   // [PAIR_STAGE_RECORDS] records, grouped by bin
	uint32_t *cnt = bin_smem + PAIR_STAGE_RECORDS * 4u, *loff = cnt + PAIR_BINS;             // loff[PAIR_BINS] = total
	const uint32_t po = blockIdx.y, hl = sel.hl[po], level = bp.level[hl];
	const uint32_t mask = lt.v[4 * level + 1] - 1u;
	const bool split = lt.v[4 * level + 2] > PAIR_RES_MAX;                                    // (uniform; the host launches S = 512 when any level of the launch is split)
	const float scale = __uint_as_float(lt.v[4 * level + 3]);
	const size_t region = (size_t)po * gridDim.x + blockIdx.x;
	uint16_t *my_off = poff + region * PAIR_OFFS;
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	if (level_absmax(absmax_bits, level) == 0u || blockIdx.x * S >= lim || (split && S * 8u > PAIR_STAGE_RECORDS)) {   // uniform exit: an empty region
		if (threadIdx.x < PAIR_OFFS) my_off[threadIdx.x] = 0;
		return;
	}
	if (threadIdx.x < PAIR_BINS) cnt[threadIdx.x] = 0;
	__syncthreads();
	const uint32_t i = blockIdx.x * S + threadIdx.x;
	const P *dy = reinterpret_cast<const P *>(dLdy);
	float2 g2 = make_float2(0.f, 0.f);
	Corner c;
	uint32_t h[4], rank[8];
	bool live = false;
	if (i < lim) {
		g2 = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)level * n + i] : dy[(size_t)i * 16 + level]);
		live = (g2.x != 0.f || g2.y != 0.f);
		if (live) {
			c = locate(pos, stride, i, scale);
			const uint32_t ty0 = c.g[1] * 19349663u, tz0 = c.g[2] * 83492791u;
#pragma unroll
			for (uint32_t q = 0; q < 4; ++q) {                                // q = (y corner, z corner), HashEncode.h:68-94
				h[q] = (ty0 + ((q & 1u) ? 19349663u : 0u)) ^ (tz0 + ((q & 2u) ? 83492791u : 0u));
				const uint32_t i0 = (c.g[0] ^ h[q]) & mask, i1 = ((c.g[0] + 1u) ^ h[q]) & mask;
				if (split) { rank[2 * q] = atomicAdd(&cnt[i0 >> PAIR_BIN_BITS], 1u); rank[2 * q + 1] = atomicAdd(&cnt[i1 >> PAIR_BIN_BITS], 1u); }
				// (edge levels: positions outside the unit cube can carry x + 1 into the bin bits: that edge goes to the spill list as two contributions, rank = ~0)
				else rank[2 * q] = ((i0 ^ i1) >> PAIR_BIN_BITS) ? ~0u : atomicAdd(&cnt[i0 >> PAIR_BIN_BITS], 1u);
			}
		}
	}
	__syncthreads();
	if (threadIdx.x < 64u) {                                                    // wave 0, two bins per lane: exclusive prefix = the region's bin offsets
		const uint32_t b0 = 2u * threadIdx.x, c0 = cnt[b0], c1 = cnt[b0 + 1u];
		uint32_t x = c0 + c1;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if ((int)threadIdx.x >= o) x += y; }
		const uint32_t e0 = x - (c0 + c1);
		loff[b0] = e0; loff[b0 + 1u] = e0 + c0;
		reinterpret_cast<uint32_t *>(my_off)[threadIdx.x] = e0 | ((e0 + c0) << 16);          // (<= 4096 records per region < 2^16)
		if (threadIdx.x == 63u) { loff[PAIR_BINS] = x; reinterpret_cast<uint32_t *>(my_off)[64] = x; }
	}
	__syncthreads();
	if (live) {
#pragma unroll
		for (uint32_t q = 0; q < 4; ++q) {
			const float wy = (q & 1u) ? c.w[1] : 1 - c.w[1], wz = (q & 2u) ? c.w[2] : 1 - c.w[2];
			const uint32_t i0 = (c.g[0] ^ h[q]) & mask, i1 = ((c.g[0] + 1u) ^ h[q]) & mask;
			if (split) {
				const float w0 = ((1 - c.w[0]) * wy) * wz, w1 = (c.w[0] * wy) * wz;             // the reference's x, y, z multiplication order
				const uint32_t l0 = i0 & (PAIR_BIN_ENTRIES - 1u), l1 = i1 & (PAIR_BIN_ENTRIES - 1u);
				stage[loff[i0 >> PAIR_BIN_BITS] + rank[2 * q]] = PairRec{g2.x * w0, g2.y * w0, 0.f, l0 | (l0 << PAIR_BIN_BITS)};
				stage[loff[i1 >> PAIR_BIN_BITS] + rank[2 * q + 1]] = PairRec{g2.x * w1, g2.y * w1, 0.f, l1 | (l1 << PAIR_BIN_BITS)};
				continue;
			}
			const float wyz = wy * wz;
			const float a = g2.x * wyz, b = g2.y * wyz;
			if (rank[2 * q] != ~0u) {
				stage[loff[i0 >> PAIR_BIN_BITS] + rank[2 * q]] = PairRec{a, b, c.w[0], (i0 & (PAIR_BIN_ENTRIES - 1u)) | ((i1 & (PAIR_BIN_ENTRIES - 1u)) << PAIR_BIN_BITS)};
			} else {
				const uint32_t k = atomicAdd(spill_count, 2u);
				const float w0 = 1 - c.w[0];
				if (k + 1u < bp.spill_cap) { spill[k] = SpillEntry{(hl << 19) | i0, a * w0, b * w0}; spill[k + 1u] = SpillEntry{(hl << 19) | i1, a * c.w[0], b * c.w[0]}; }
			}
		}
	}
	__syncthreads();
	const uint32_t total = loff[PAIR_BINS];
	PairRec *out = prec + region * pair_region_records();
	if (probe & 1u) { if (total == 0x7fffffffu) out[0] = stage[0]; return; }
	for (uint32_t p = threadIdx.x; p < total; p += S) out[p] = stage[p];
}

// ---- run records without atomics (r4): the coarse levels of the fp32 path.  k_bin_records_runs above with the edge kernel's layout: a region per record workgroup
// (its run records sorted by bin, 12 bytes each {x, y, slot}; the bin offsets beside them), 128 bins of 4096 entries like the edge levels - so that ONE accumulate
// kernel serves every level.  Entries of a level smaller than 2^19 are dealt to the bins in interleaved groups of eight (the spatially coherent dense indices
// spread evenly), as before with 64 bins.
struct RunRec { float x, y; uint32_t loc; };
#define RUN2_OFFS (PAIR_BINS + 2u)
__host__ __device__ static inline uint32_t run2_region_records() { return RUN_WG * RUN_K * 8u + 4u; }
static uint32_t run2_stage_bytes(uint32_t stage) { return stage * 12u + (3u * PAIR_BINS + 4u) * 4u; }
__host__ __device__ __forceinline__ uint32_t bin2_of(uint32_t e, bool il) { return il ? (e >> 3) & (PAIR_BINS - 1u) : e >> PAIR_BIN_BITS; }
__host__ __device__ __forceinline__ uint32_t local2_of(uint32_t e, bool il) { return il ? ((e >> 10) << 3) | (e & 7u) : e & (PAIR_BIN_ENTRIES - 1u); }
__host__ __device__ __forceinline__ uint32_t entry2_of(uint32_t bin, uint32_t local, bool il) { return il ? ((local >> 3) << 10) | (bin << 3) | (local & 7u) : (bin << PAIR_BIN_BITS) | local; }

template <typename T, int LAYOUT, int OCC>
__global__ __launch_bounds__(RUN_WG, OCC) void k_bin_runs2(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ dLdy, LevelTable lt, BinPlan bp, LevelSel sel,
                                                         const uint32_t *__restrict__ absmax_bits, RunRec *__restrict__ rrec, uint16_t *__restrict__ roff,
                                                         const uint32_t *__restrict__ n_valid, uint32_t stage /* records of LDS staging */,
                                                         uint32_t probe /* timing experiments (results wrong): 1 = the histogram atomics spread over lane-distinct addresses, 2 = no global record stores */) {
	extern __shared__ __attribute__((aligned(16))) uint32_t bin_smem[];
	RunRec *stage_rec = reinterpret_cast<RunRec *>(bin_smem);                              // [stage]
	uint32_t *cnt = bin_smem + stage * 3u, *loff = cnt + PAIR_BINS, *cnt2 = loff + PAIR_BINS + 2u;   // loff[PAIR_BINS] = total
	const uint32_t ro = blockIdx.y, hl = sel.hl[ro], level = bp.level[hl];
	const uint32_t size = lt.v[4 * level + 1], res = lt.v[4 * level + 2];
	const float scale = __uint_as_float(lt.v[4 * level + 3]);
	const bool dense = level_is_dense(size, res), il = size < BIN_LEVEL_MAX;
	const size_t region = (size_t)ro * gridDim.x + blockIdx.x;
	uint16_t *my_off = roff + region * RUN2_OFFS;
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	if (level_absmax(absmax_bits, level) == 0u || blockIdx.x * RUN_WG * RUN_K >= lim) {        // uniform exit: an empty region
		if (threadIdx.x < RUN2_OFFS) my_off[threadIdx.x] = 0;
		return;
	}
	if (threadIdx.x < PAIR_BINS) { cnt[threadIdx.x] = 0; cnt2[threadIdx.x] = 0; }
	__syncthreads();
	const uint32_t first = (blockIdx.x * RUN_WG + threadIdx.x) * RUN_K;
	using P = typename Pair<T>::type;
	const P *dy = reinterpret_cast<const P *>(dLdy);
	uint32_t cell[RUN_K][3]; float frac[RUN_K][3]; float2 gk[RUN_K];
	{
		float px[RUN_K][3];
		if (first + RUN_K <= lim && stride == 3) {
			const float4 *p4 = reinterpret_cast<const float4 *>(pos + (size_t)first * 3);   // 24 floats, 16-byte aligned (first % 8 == 0)
			float4 v[6];
#pragma unroll
			for (int r = 0; r < 6; ++r) v[r] = p4[r];
			const float *f = reinterpret_cast<const float *>(v);
#pragma unroll
			for (uint32_t k = 0; k < RUN_K; ++k) { px[k][0] = f[3 * k]; px[k][1] = f[3 * k + 1]; px[k][2] = f[3 * k + 2]; }
#pragma unroll
			for (uint32_t k = 0; k < RUN_K; ++k) gk[k] = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)level * n + first + k] : dy[(size_t)(first + k) * 16 + level]);
		} else {
#pragma unroll
			for (uint32_t k = 0; k < RUN_K; ++k) {
				const uint32_t i = first + k;
				if (i < lim) {
					px[k][0] = pos[(size_t)i * stride]; px[k][1] = pos[(size_t)i * stride + 1]; px[k][2] = pos[(size_t)i * stride + 2];
					gk[k] = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)level * n + i] : dy[(size_t)i * 16 + level]);
				} else { px[k][0] = px[k][1] = px[k][2] = 0.f; gk[k] = make_float2(0.f, 0.f); }
			}
		}
#pragma unroll
		for (uint32_t k = 0; k < RUN_K; ++k)
#pragma unroll
			for (int d = 0; d < 3; ++d) { const float p = px[k][d] * scale + 0.5f; const float fl = floorf(p); cell[k][d] = (uint32_t)(int)fl; frac[k][d] = p - fl; }   // pos_fract, HashEncode.h:106-115
	}
	// one sweep over the thread's samples; emit(entry, x, y) is called for the eight corners of every finished run (k_bin_records_runs' sweep)
	auto sweep = [&](auto emit) {
		bool open = false;
		uint32_t key[3] = {0u, 0u, 0u};
		float ax[8], ay[8];
		auto flush = [&]() {
			uint32_t idx[8];
			cell_entries(size, res, dense, key[0], key[1], key[2], idx);
#pragma unroll
			for (uint32_t q = 0; q < 8; ++q) emit(idx[q], ax[q], ay[q]);
		};
#pragma unroll
		for (uint32_t k = 0; k < RUN_K; ++k) {
			if (gk[k].x == 0.f && gk[k].y == 0.f) continue;        // zero rows (padding) add exact zeros in the reference: skipped, they do not end a run either
			if (open && !(cell[k][0] == key[0] && cell[k][1] == key[1] && cell[k][2] == key[2])) { flush(); open = false; }
			if (!open) {
				open = true; key[0] = cell[k][0]; key[1] = cell[k][1]; key[2] = cell[k][2];
#pragma unroll
				for (uint32_t q = 0; q < 8; ++q) { ax[q] = 0.f; ay[q] = 0.f; }
			}
			const float x1 = frac[k][0], x0 = 1 - x1, y1 = frac[k][1], y0 = 1 - y1, z1 = frac[k][2], z0 = 1 - z1;
#pragma unroll
			for (uint32_t q = 0; q < 8; ++q) {
				const float w = ((q & 1u) ? x1 : x0) * ((q & 2u) ? y1 : y0) * ((q & 4u) ? z1 : z0);                 // the reference's x, y, z multiplication order
				ax[q] += gk[k].x * w; ay[q] += gk[k].y * w;
			}
		}
		if (open) flush();
	};
	const uint32_t spread = (probe & 1u) ? threadIdx.x : 0u;
	sweep([&](uint32_t e, float, float) { atomicAdd(&cnt[(bin2_of(e, il) + spread) & (PAIR_BINS - 1u)], 1u); });
	__syncthreads();
	if (threadIdx.x < 64u) {                                                    // wave 0, two bins per lane: exclusive prefix = the region's bin offsets
		const uint32_t b0 = 2u * threadIdx.x, c0 = cnt[b0], c1 = cnt[b0 + 1u];
		uint32_t x = c0 + c1;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(x, o); if ((int)threadIdx.x >= o) x += y; }
		const uint32_t e0 = x - (c0 + c1);
		loff[b0] = e0; loff[b0 + 1u] = e0 + c0;
		reinterpret_cast<uint32_t *>(my_off)[threadIdx.x] = e0 | ((e0 + c0) << 16);          // (<= 16384 records per region < 2^16)
		if (threadIdx.x == 63u) { loff[PAIR_BINS] = x; reinterpret_cast<uint32_t *>(my_off)[64] = x; }
	}
	__syncthreads();
	RunRec *out = rrec + region * run2_region_records();
	sweep([&](uint32_t e, float x, float y) {
		const uint32_t bin = (bin2_of(e, il) + spread) & (PAIR_BINS - 1u), p = loff[bin] + atomicAdd(&cnt2[bin], 1u);
		const RunRec r{x, y, local2_of(e, il)};
		if (p < stage) stage_rec[p] = r; else if (!(probe & 2u)) out[p] = r;     // beyond the staging area (scattered positions only): straight to its place in the region
	});
	__syncthreads();
	const uint32_t total = min(loff[PAIR_BINS], stage);
	if (probe & 2u) return;
	for (uint32_t p = threadIdx.x; p < total; p += RUN_WG) out[p] = stage_rec[p];
}

// ---- ONE accumulate kernel for every level of the fp32 path: a workgroup per (level, 4096-entry bin), 64 KiB of 64-bit accumulators (two workgroups per CU).  It walks
// the record regions of its level: LANES lanes per region read the region's segment of the bin (offset table -> start, length) with consecutive lanes on consecutive
// records (full lines), I records per lane in flight, U regions per thread.  Unit order: edge levels first (the heavier units), then the run levels.
#define ACC2_WG 512u
#define ACC2_RB 512u                                                     // regions per block of the gather (= threads: one region per thread when the segment table is built)
#define ACC2_MAPN 1024u
#define ACC2_LDS_EXTRA ((2u * (ACC2_RB + 1u) + 16u) * 4u + ACC2_MAPN * 8u)
struct Acc2Plan { uint32_t n_pair, n_run, pair_regions, pair_region_records, run_regions, run_region_records, probe; };
// c * s (s a power of two) rounded to the nearest integer (ties to even), as a 64-bit integer: __float2ll_rn without the generic expansion.  t = c * s is exact, rint(t) is
// an integer-valued float with <= 24 significant bits, so its split into hi * 2^32 + lo is exact too.  |t| < 2^62 by construction of the scale.
__host__ __device__ __forceinline__ long long fixed_rn(float c, float s) {
	const float r = rintf(c * s), m = fabsf(r);
	const float hi = floorf(m * 2.3283064365386963e-10f);               // floor(|r| / 2^32)
	const float lo = fmaf(hi, -4294967296.0f, m);                        // |r| - hi * 2^32, in [0, 2^32): exact (a multiple of ulp(|r|) below 2^32)
	const long long v = (long long)(((unsigned long long)(uint32_t)hi << 32) | (unsigned long long)(uint32_t)lo);
	return r < 0.f ? -v : v;
}
static_assert(ACC2_RB == ACC2_WG, "gather_flat builds one segment-table row per thread");
// test hook (tests/test_host_cpu.py): entry of a level with `size` entries -> (bin, slot inside the bin, entry rebuilt from them, slots the bin's accumulate workgroup owns) as the region
// kernels and k_bin_accumulate2 compute them
NGP_API void ngp_x_bin2_map(uint32_t size, uint32_t e, uint32_t *out4_host) {
	const bool il = size < BIN_LEVEL_MAX;
	const uint32_t bin = bin2_of(e, il), local = local2_of(e, il), groups_all = (size + 7u) >> 3;
	out4_host[0] = bin; out4_host[1] = local; out4_host[2] = entry2_of(bin, local, il);
	out4_host[3] = il ? (groups_all > bin ? ((groups_all - bin + PAIR_BINS - 1u) / PAIR_BINS) << 3 : 0u) : PAIR_BIN_ENTRIES;
}
// test hook (tests/test_host_cpu.py): the same function on the host, against round-half-even of the exact product
NGP_API long long ngp_x_fixed_rn(float c, float s) { return fixed_rn(c, s); }
// The records of one (level, bin) lie in n_regions segments (one per record workgroup).  Per block of ACC2_RB regions: segment table (start, length) -> exclusive prefix P in LDS ->
// the segments laid end to end as ONE flat list that the threads walk densely (thread t takes flat records t, t + 512, ...; eight loads in flight): consecutive lanes read
// consecutive records of a segment (full lines) and every lane has work - eight lanes per segment with a fixed number of slots left half of them idle, and the kernel was
// bound by exactly that (k_bin_accumulate2 90 us).  flat index -> segment: a map with one entry per 2^shift flat records (the segment holding the first of them), then a
// short forward walk over P.
// this thread's row of the segment table of (level, bin): region w0 + threadIdx.x -> {length, index of the segment's first record}
__device__ __forceinline__ uint2 segment_row(const uint16_t *__restrict__ offs, uint32_t level_ord, uint32_t n_regions, uint32_t region_records, uint32_t offs_per_region, uint32_t bin, uint32_t w0) {
	if (w0 + threadIdx.x >= n_regions) return make_uint2(0u, 0u);
	const size_t region = (size_t)level_ord * n_regions + w0 + threadIdx.x;
	const uint32_t o = *reinterpret_cast<const uint32_t *>(offs + region * offs_per_region + (bin & ~1u));        // offsets of bins 2k, 2k+1 in one word
	const uint32_t o2 = *reinterpret_cast<const uint32_t *>(offs + region * offs_per_region + (bin & ~1u) + 2u);  // ... of 2k+2 (or the total)
	const uint32_t q0 = (bin & 1u) ? o >> 16 : o & 0xffffu, q1 = (bin & 1u) ? o2 & 0xffffu : o >> 16;
	return make_uint2(q1 - q0, (uint32_t)(region * region_records) + q0);
}
template <typename Rec, typename F>
__device__ __forceinline__ void gather_flat(const Rec *__restrict__ recs, const uint16_t *__restrict__ offs, uint32_t level_ord, uint32_t n_regions, uint32_t region_records,
                                            uint32_t offs_per_region, uint32_t bin, uint32_t *__restrict__ lds, uint32_t probe /* timing experiments: 1 = records loaded, not processed; 2 = not loaded */,
                                            uint2 row0 /* segment_row(..., 0), loaded by the caller ahead of time */, F process) {
	uint32_t sink = 0;
	uint2 *seg = reinterpret_cast<uint2 *>(lds);                           // [RB + 1] per segment: {flat index of its END, start - flat index of its beginning}: one 8-byte read answers "is f mine" and "where is it"
	uint32_t *wsum = lds + 2u * (ACC2_RB + 1u);                            // per-wave totals[8 (+8 spare)]
	uint2 *map = reinterpret_cast<uint2 *>(wsum + 16u);                    // [ACC2_MAPN] per 2^shift flat records: the seg entry of the segment holding the first of them, its index in the top 9 bits of .x
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	for (uint32_t w0 = 0; w0 < n_regions; w0 += ACC2_RB) {
		const uint2 row = w0 == 0 ? row0 : segment_row(offs, level_ord, n_regions, region_records, offs_per_region, bin, w0);
		const uint32_t len = row.x, st = row.y;
		uint32_t inc = len;                                                 // inclusive prefix over the workgroup's 512 lengths
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if ((int)lane >= o) inc += y; }
		if (lane == 63u) wsum[wave] = inc;
		__syncthreads();
		uint32_t base = 0, T = 0;
#pragma unroll
		for (uint32_t k = 0; k < ACC2_WG / 64u; ++k) { const uint32_t v = wsum[k]; base += k < wave ? v : 0u; T += v; }
		const uint32_t p0 = base + inc - len;
		seg[threadIdx.x] = make_uint2(p0 + len, st - p0);
		if (threadIdx.x == ACC2_RB - 1u) seg[ACC2_RB] = make_uint2(0xffffffffu, 0u);     // sentinel: the forward walk stops here
		uint32_t shift = 3;
		while ((T >> shift) >= ACC2_MAPN) ++shift;
		if (T >> 23) { T = (1u << 23) - 1u; }                               // (cannot happen below 8 M records per bin; keeps the packed entries well-formed)
		if (len) { const uint32_t G = 1u << shift; for (uint32_t g = (p0 + G - 1u) >> shift; (g << shift) < p0 + len; ++g) map[g] = make_uint2((p0 + len) | (threadIdx.x << 23), st - p0); }   // (T < 2^23: 8 M records of one bin)
		__syncthreads();
		constexpr uint32_t B = 8;
		for (uint32_t f0 = threadIdx.x; f0 < T; f0 += ACC2_WG * B) {
			Rec x[B];
#pragma unroll
			for (uint32_t b = 0; b < B; ++b) {
				const uint32_t f = f0 + b * ACC2_WG;
				if (f < T) {
					uint2 e = map[f >> shift];
					uint32_t w = e.x >> 23; e.x &= 0x7fffffu;
					while (f >= e.x) e = seg[++w];                          // (rarely: f lies up to 2^shift - 1 records behind the mapped one; empty segments end where they begin: skipped)
					if (!(probe & 2u)) x[b] = recs[e.y + f];
				}
			}
#pragma unroll
			for (uint32_t b = 0; b < B; ++b) if (f0 + b * ACC2_WG < T) { if (probe & 1u) sink ^= reinterpret_cast<const uint32_t *>(&x[b])[0]; else process(x[b]); }
		}
		__syncthreads();                                                    // the tables are rebuilt for the next block of regions
	}
	if (sink == 0x9e3779b9u) lds[0] = sink;                                 // (keeps the probe's loads alive)
}

template <typename G, bool QUEUE /* units drawn from a queue by resident workgroups | one unit per workgroup */>
__global__ __launch_bounds__(ACC2_WG, 4) void k_bin_accumulate2(LevelTable lt, BinPlan bp, LevelSel sel_pair, LevelSel sel_run, Acc2Plan ap, const uint32_t *__restrict__ absmax_bits,
                                                               const PairRec *__restrict__ prec, const uint16_t *__restrict__ poff, const RunRec *__restrict__ rrec, const uint16_t *__restrict__ roff,
                                                               const uint32_t *__restrict__ spill_count, const SpillEntry *__restrict__ spill, G *__restrict__ grad, int overwrite,
                                                               uint32_t *__restrict__ queue_head /* zero at launch (hash_bwd_impl launches the kernel at most twice per step: two words) */) {
	extern __shared__ __attribute__((aligned(16))) unsigned long long iacc[];   // [PAIR_BIN_ENTRIES][2] 64-bit fixed point
	__shared__ uint32_t s_next;
	using GP = typename Pair<G>::type;
	uint32_t *tables = reinterpret_cast<uint32_t *>(iacc + 2u * PAIR_BIN_ENTRIES);        // gather_flat's segment tables, behind the accumulators
	// Persistent: two workgroups per CU take the units round-robin.  A unit's critical path held two memory round trips (its row of the segment table, then the records);
	// the row of the NEXT unit is now requested before this unit's records are, so only one of them is exposed (fixed cost of the 2048 units of a 2^18-sample batch: 30 of 74 us).
	const uint32_t n_units = (ap.n_pair + ap.n_run) * PAIR_BINS;
	auto unit_row = [&](uint32_t u) {
		const bool pr = u < ap.n_pair * PAIR_BINS;
		const uint32_t v = pr ? u : u - ap.n_pair * PAIR_BINS;
		return pr ? segment_row(poff, v / PAIR_BINS, ap.pair_regions, ap.pair_region_records, PAIR_OFFS, v % PAIR_BINS, 0u)
		          : segment_row(roff, v / PAIR_BINS, ap.run_regions, ap.run_region_records, RUN2_OFFS, v % PAIR_BINS, 0u);
	};
	// Units differ in weight (an edge unit carries ~3x the records of a run unit), so after its first unit - the launch order: the heaviest first - a workgroup draws the next
	// from a queue (one returning atomic per unit, issued a whole unit ahead of its use).
	uint2 row_next = blockIdx.x < n_units ? unit_row(blockIdx.x) : make_uint2(0u, 0u);
	if (QUEUE && threadIdx.x == 0) s_next = gridDim.x + atomicAdd(queue_head, 1u);
	uint32_t u = blockIdx.x;
	while (u < n_units) {
	const uint2 row0 = row_next;
	uint32_t u_next = ~0u;
	if (QUEUE) {
		__syncthreads();
		u_next = s_next;
		if (u_next < n_units) row_next = unit_row(u_next);
		__syncthreads();                                                  // everyone has read s_next
		if (threadIdx.x == 0) s_next = gridDim.x + atomicAdd(queue_head, 1u);
	}
	const uint32_t u_this = u;
	u = u_next;
	const bool is_pair = u_this < ap.n_pair * PAIR_BINS;
	const uint32_t unit = is_pair ? u_this : u_this - ap.n_pair * PAIR_BINS;
	const uint32_t ord = unit / PAIR_BINS, bin = unit % PAIR_BINS, hl = is_pair ? sel_pair.hl[ord] : sel_run.hl[ord], level = bp.level[hl];
	const uint32_t size = lt.v[4 * level + 1];
	const bool il = size < BIN_LEVEL_MAX;
	// slots of this bin that are entries of the level (interleaved: groups bin, bin + 128, ... of the level's ceil(size / 8) groups)
	const uint32_t groups_all = (size + 7u) >> 3;
	const uint32_t n_local = il ? (groups_all > bin ? ((groups_all - bin + PAIR_BINS - 1u) / PAIR_BINS) << 3 : 0u) : PAIR_BIN_ENTRIES;
	const uint32_t amax = level_absmax(absmax_bits, level);
	float s32 = 0.f;
	{ const float m = __uint_as_float(amax); if (m > 0.f && m < 3.0e38f) { int ex; frexpf(m, &ex); s32 = ldexpf(1.0f, 38 - ex); } }
	const float inv = s32 > 0.f ? 1.0f / s32 : 0.f;
	const uint32_t ns = is_pair ? min(*spill_count, bp.spill_cap) : 0u;
	GP *dst = reinterpret_cast<GP *>(grad) + lt.v[4 * level];
	if (s32 == 0.f) {                                                     // the level has no gradient: an accumulating destination is left alone, an overwritten one gets its zeros
		if (overwrite) {
			GP zv; from_f2(zv, make_float2(0.f, 0.f));
			for (uint32_t e = threadIdx.x; e < n_local; e += ACC2_WG) { const uint32_t t = entry2_of(bin, e, il); if (t < size) dst[t] = zv; }
		}
		continue;                                                           // (uniform)
	}
	for (uint32_t e = threadIdx.x; e < n_local; e += ACC2_WG) { iacc[2 * e] = 0ull; iacc[2 * e + 1] = 0ull; }
	__syncthreads();
	auto add_fixed = [&](uint32_t local, long long ix, long long iy) {
		if ((ix | iy) == 0) return;
		__hip_atomic_fetch_add(&iacc[2 * local], (unsigned long long)ix, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		__hip_atomic_fetch_add(&iacc[2 * local + 1], (unsigned long long)iy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	};
	if (is_pair) {
		gather_flat(prec, poff, ord, ap.pair_regions, ap.pair_region_records, PAIR_OFFS, bin, tables, ap.probe, row0, [&](const PairRec &r) {
			const float w0 = 1 - r.fx;
			add_fixed(r.loc & (PAIR_BIN_ENTRIES - 1u), fixed_rn(r.a * w0, s32), fixed_rn(r.b * w0, s32));
			add_fixed((r.loc >> PAIR_BIN_BITS) & (PAIR_BIN_ENTRIES - 1u), fixed_rn(r.a * r.fx, s32), fixed_rn(r.b * r.fx, s32));
		});
		for (uint32_t t = threadIdx.x; t < ns; t += ACC2_WG) {             // out-of-cube edges (normally ns == 0)
			const SpillEntry se = spill[t];
			const uint32_t e = se.key & (BIN_LEVEL_MAX - 1u);
			if ((se.key >> 19) == hl && (e >> PAIR_BIN_BITS) == bin) add_fixed(e & (PAIR_BIN_ENTRIES - 1u), __float2ll_rn(se.x * s32), __float2ll_rn(se.y * s32));
		}
	} else {
		gather_flat(rrec, roff, ord, ap.run_regions, ap.run_region_records, RUN2_OFFS, bin, tables, ap.probe, row0, [&](const RunRec &r) {
			add_fixed(r.loc, fixed_rn(r.x, s32), fixed_rn(r.y, s32));
		});
	}
	__syncthreads();
	if (!il) {                                                            // a full bin: contiguous, two entries (16 bytes of fp32 gradient) per thread and trip
		GP *d = dst + (bin << PAIR_BIN_BITS);
		for (uint32_t e = 2u * threadIdx.x; e < PAIR_BIN_ENTRIES; e += 2u * ACC2_WG) {
			const long long s0 = (long long)iacc[2 * e], s1 = (long long)iacc[2 * e + 1], s2 = (long long)iacc[2 * e + 2], s3 = (long long)iacc[2 * e + 3];
			float2 v0 = make_float2((float)s0 * inv, (float)s1 * inv), v1 = make_float2((float)s2 * inv, (float)s3 * inv);
			if (!overwrite) {
				if ((s0 | s1 | s2 | s3) == 0) continue;
				const float2 o0 = to_f2(d[e]), o1 = to_f2(d[e + 1]);
				v0.x += o0.x; v0.y += o0.y; v1.x += o1.x; v1.y += o1.y;
			}
			GP w0, w1; from_f2(w0, v0); from_f2(w1, v1);
			if (sizeof(GP) == 8) *reinterpret_cast<float4 *>(d + e) = make_float4(v0.x, v0.y, v1.x, v1.y);
			else { d[e] = w0; d[e + 1] = w1; }
		}
	} else {
		for (uint32_t e = threadIdx.x; e < n_local; e += ACC2_WG) {
			const uint32_t t = entry2_of(bin, e, il);
			if (t >= size) continue;
			const long long sx = (long long)iacc[2 * e], sy = (long long)iacc[2 * e + 1];
			float2 v = make_float2((float)sx * inv, (float)sy * inv);
			if (!overwrite) { if (sx == 0 && sy == 0) continue; const float2 old = to_f2(dst[t]); v.x += old.x; v.y += old.y; }
			GP o; from_f2(o, v);
			dst[t] = o;
		}
	}
	if (QUEUE) __syncthreads();                                           // the accumulators are cleared again for the next unit
	}
}

// Which levels can take the binned path: up to 2^19 entries, and indexed the way the record kernels index (dense, or the XOR hash masked by a power of two).
// (aabb_scale 23.4 has a DENSE level with res 80 = 512000 entries - round 1 binned it with the XOR hash by looking at the size alone.)
static bool level_dense_host(uint32_t size, uint32_t res) { uint32_t stride = 1; for (int d = 0; d < 3; ++d) if (stride <= size) stride *= res; return !(size < stride); }
static bool level_binned(const LevelTable &lt, int l) {
	const uint32_t size = lt.v[4 * l + 1], res = lt.v[4 * l + 2];
	return size <= BIN_LEVEL_MAX && (level_dense_host(size, res) || (size & (size - 1)) == 0);
}
// the hashed 2^19-entry levels: the only ones that do not need partial slabs when the owner-computes scan runs (no bins: fixed-point request, small workspace)
static bool level_exclusive(const LevelTable &lt, int l) {
	const uint32_t size = lt.v[4 * l + 1], res = lt.v[4 * l + 2];
	return div_up(size, OWN_SLICE) >= 32 && (size & (size - 1)) == 0 && !level_dense_host(size, res);
}
static uint64_t hash_bwd_workspace_bytes(const LevelTable &lt) {           // partial slabs of the owner-computes scan
	uint64_t entries = 0;
	for (int l = 0; l < 16; ++l) if (!level_exclusive(lt, l)) entries += (uint64_t)32u * lt.v[4 * l + 1];
	return (entries * sizeof(float2) + 255) & ~(uint64_t)255;
}
// capacity of one record list: 4x the expected n*8/64 records per bin; % 8: 16-byte aligned streams
static uint32_t bin_capacity(uint32_t n) { uint32_t c = ((n / 2) / CUR_SUBS + 7u) & ~7u; const uint32_t lo = 4096u / CUR_SUBS; return c < lo ? lo : c; }
// the levels whose cell edges never leave a 4096-entry bin (k_bin_pairs): full 2^19-entry hashed tables up to res 2048
static bool level_pair_capable(const LevelTable &lt, int l) {            // (beyond res 2048: as eight single records per sample)
	const uint32_t size = lt.v[4 * l + 1], res = lt.v[4 * l + 2];
	return size == BIN_LEVEL_MAX && !level_dense_host(size, res);
}
// workspace = slabs | cursors u32[N_CURSORS] | absmax partials u32[16*NGP_ABSMAX_PARTS], spill count u32 | record values | record indices | spill list | edge records
struct WsLayout { uint64_t cursors, absmax, rec_val, rec_idx, spill, pair_rec, pair_off, run_rec, run_off, total; uint32_t cap, spill_cap, n_binned, n_pair; };
static WsLayout ws_layout(const LevelTable &lt, uint32_t n) {
	WsLayout w;
	w.n_binned = 0;
	for (int l = 0; l < 16; ++l) if (level_binned(lt, l)) ++w.n_binned;
	w.cap = bin_capacity(n);
	w.spill_cap = w.n_binned * 8u * (n < (1u << 25) / (w.n_binned ? w.n_binned : 1u) ? n : (1u << 25) / (w.n_binned ? w.n_binned : 1u));   // worst case: every record of every binned level overflows (12 B each)
	w.cursors = hash_bwd_workspace_bytes(lt);
	w.absmax = w.cursors + N_ZEROED * 4u;
	w.rec_val = w.absmax + 16u * ABSMAX_PARTS * 4u + 256;
	w.rec_idx = w.rec_val + (uint64_t)w.n_binned * BINS_PER_LEVEL * CUR_SUBS * w.cap * sizeof(float2);       // every level owns 64 * 8 * cap * 8 bytes (rec_val_at)
	w.spill = (w.rec_idx + (uint64_t)w.n_binned * BINS_PER_LEVEL * CUR_SUBS * w.cap * sizeof(uint16_t) + 255) & ~(uint64_t)255;
	w.pair_rec = (w.spill + (uint64_t)w.spill_cap * sizeof(SpillEntry) + 255) & ~(uint64_t)255;
	w.n_pair = 0;
	for (int l = 0; l < 16; ++l) if (level_binned(lt, l) && level_pair_capable(lt, l)) ++w.n_pair;
	// edge records: one region per record workgroup (sized for the smaller workgroup choice: more regions, more slack records), then the regions' bin offsets
	const uint64_t regions512 = div_up(n, 512u);
	w.pair_off = (w.pair_rec + (uint64_t)w.n_pair * regions512 * pair_region_records() * sizeof(PairRec) + 255) & ~(uint64_t)255;
	w.run_rec = w.pair_off + (((uint64_t)w.n_pair * regions512 * PAIR_OFFS * sizeof(uint16_t) + 255) & ~(uint64_t)255);
	// run records of the fp32 path: a region per 2048-sample workgroup, worst case eight records per sample (scattered positions); any binned level can be a run level
	const uint64_t regions_run = div_up(n, RUN_WG * RUN_K);
	w.run_off = (w.run_rec + (uint64_t)w.n_binned * regions_run * run2_region_records() * sizeof(RunRec) + 255) & ~(uint64_t)255;
	w.total = w.run_off + (((uint64_t)w.n_binned * regions_run * RUN2_OFFS * sizeof(uint16_t) + 255) & ~(uint64_t)255);
	return w;
}
static uint64_t hash_bwd_workspace_bytes_binned(const LevelTable &lt, uint32_t n) { return ws_layout(lt, n).total; }
NGP_API uint64_t ngp_hash_bwd_workspace_bytes(const uint32_t *level_table_host, uint32_t n) { return hash_bwd_workspace_bytes_binned(load_table(level_table_host), n); }

// helper stream for the owner-computes kernels when some levels are binned and others are not (tables beyond 2^19 entries per level); created once per process
struct SideStream {
	hipStream_t stream = nullptr; hipEvent_t fork = nullptr, join = nullptr; bool ok = false;
	SideStream() {
		if (getenv("NGP_HASH_BWD_NO_SIDE_STREAM")) return;
		ok = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess &&
		     hipEventCreateWithFlags(&join, hipEventDisableTiming) == hipSuccess;
	}
};

static int hash_bwd_impl(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *dLdy, const uint32_t *level_table_host,
                         void *grad, uint64_t n_params, int dtype, int grad_dtype, int in_layout, int zero_first, const uint32_t *n_valid, float *level_scratch, void *workspace, uint64_t workspace_bytes,
                         hipEvent_t after_coarse = nullptr /* data parallel, overlapped exchange: recorded behind the accumulate launch of the run-combined (coarse) levels, which then is a launch of its own */,
                         bool absmax_done = false /* the abs-max partials, zeroed cursors and spill count are already in the workspace (written by the field backward kernel, ngp_hash_bwd_absmax_slots) */) {
	NGP_REQUIRE(grad && level_table_host && (n == 0 || (pos && dLdy)), NGP_E_ARG, "ngp_hash_encode_bwd: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_hash_encode_bwd: bad dtype %d", dtype);
	NGP_REQUIRE(grad_dtype == NGP_F32 || (grad_dtype == NGP_F16 && dtype == NGP_F16), NGP_E_DTYPE, "ngp_hash_encode_bwd: bad grad dtype %d for dtype %d", grad_dtype, dtype);
	hipStream_t s = (hipStream_t)stream;
	static SideStream side;
	const size_t gsz = grad_dtype == NGP_F16 ? 2 : 4;
	const LevelTable lt = load_table(level_table_host);
	bool owner_ok = true;
	for (int l = 0; l < 16; ++l) {
		const uint32_t size = lt.v[4 * l + 1], res = lt.v[4 * l + 2];
		uint32_t stride_ = 1; for (int d = 0; d < 3; ++d) if (stride_ <= size) stride_ *= res;
		if (size < stride_ && (size & (size - 1)) != 0) owner_ok = false;
	}
	if (hash_bwd_method() == 1 || !owner_ok) {
		if (zero_first) {
			hipError_t e = hipMemsetAsync(grad, 0, n_params * gsz, s);
			if (e != hipSuccess) { ngp_set_error("ngp_hash_encode_bwd memset: %s", hipGetErrorString(e)); return (int)e; }
		}
		if (n == 0) { if (after_coarse) hipEventRecord(after_coarse, s); return 0; }
		const uint32_t nblk = div_up(n, 256);
		const dim3 grid(16 * nblk), block(256);
#define GO(T, G, L) NGP_LAUNCH((k_hash_bwd<T, G, L>), grid, block, 0, s, n, pos, pos_stride, (const T *)dLdy, lt, (G *)grad, nblk, n_valid)
		if (dtype == NGP_F32) { if (in_layout == NGP_LAYOUT_SOA) GO(float, float, NGP_LAYOUT_SOA); else GO(float, float, NGP_LAYOUT_AOS); }
		else if (grad_dtype == NGP_F32) { if (in_layout == NGP_LAYOUT_SOA) GO(__half, float, NGP_LAYOUT_SOA); else GO(__half, float, NGP_LAYOUT_AOS); }
		else { if (in_layout == NGP_LAYOUT_SOA) GO(__half, __half, NGP_LAYOUT_SOA); else GO(__half, __half, NGP_LAYOUT_AOS); }
#undef GO
		NGP_LAUNCH_CHECK("ngp_hash_encode_bwd");
		if (after_coarse) hipEventRecord(after_coarse, s);
		return 0;
	}
	// ---- which levels go where.  With the full workspace and no fixed-point request: every level of up to 2^19 entries through the bins (coarse ones with run
	// combining).  Whatever is left - everything, without a workspace - takes the owner-computes scan: slices of OWN_SLICE entries, the 2^19-entry hashed levels
	// with one exclusive owner per slice, the others split into sample chunks (partial slabs if the workspace holds them, an atomic flush otherwise).
	OwnerPlan plan;
	uint32_t slices[16], units = 0, k = 0;
	const bool use_slabs = workspace && workspace_bytes >= hash_bwd_workspace_bytes(lt);
	const WsLayout wl = ws_layout(lt, n);
	const bool use_bins = use_slabs && !level_scratch && (dtype == NGP_F16 || grad_dtype == NGP_F32) && workspace_bytes >= wl.total && getenv("NGP_HASH_BWD_NO_BINS") == nullptr;
	bool in_bins[16];
	BinPlan bp; bp.n_levels = 0; bp.cap = wl.cap; bp.spill_cap = wl.spill_cap;
	LevelSel sel_fine, sel_runs, sel_all, sel_pair;                      // sel_all: every level with per-corner records (runs + fine); sel_pair: the edge-record levels
	uint32_t n_fine = 0, n_runs = 0, n_pair = 0, n_all = 0;
	// edge records (k_bin_pairs): fp32 in, fp32 out, 16-byte aligned gradient.  NGP_HASH_BWD_PAIRS=0 keeps round 3's per-corner records (A/B), _PAIR_WG is a probe hook
	const bool pairs_on = [] { const char *e = getenv("NGP_HASH_BWD_PAIRS"); return !(e && e[0] == '0'); }();
	const uint32_t pair_wg = [] { const char *e = getenv("NGP_HASH_BWD_PAIR_WG"); const uint32_t v = e ? (uint32_t)strtoul(e, nullptr, 0) : 1024u; return v == 512u ? 512u : 1024u; }();
	// fp32 dL/dy only by default.  The kernels take fp16 dL/dy as well (NGP_HASH_BWD_PAIRS=2), but the fp16 configuration is better off with its 6-byte per-corner records:
	// measured on the ngp_fox.py shape, stage alone 103 vs 99 us, in the step (beside the cone-stepping marcher on the side streams) 1777 vs 1818 it/s
	const bool pairs_f16 = [] { const char *e = getenv("NGP_HASH_BWD_PAIRS"); return e && e[0] == '2'; }();
	const bool use_pairs = use_bins && pairs_on && (dtype == NGP_F32 || pairs_f16) && grad_dtype == NGP_F32 && ((uintptr_t)grad & 15u) == 0;
	const uint32_t run_res_max = [] { const char *e = getenv("NGP_HASH_BWD_RUN_RES"); return e ? (uint32_t)strtoul(e, nullptr, 0) : RUN_RES_MAX; }();   // probe hook
	// (probe hooks) staging records / register budget of the run kernels.  fp32 path: 2560 records and five waves per SIMD - all 1280 workgroups of a 2^18-sample batch resident
	// at once (measured: 54 -> 43 us); per-corner path: round 2's 3072 / natural register count
	const int run_occ = [&] { const char *e = getenv("NGP_HASH_BWD_RUN_OCC"); return e ? (e[0] == '5' ? 5 : 4) : 0; }();
	const uint32_t run_stage_env = [] { const char *e = getenv("NGP_HASH_BWD_RUN_STAGE"); const uint32_t v = e ? (uint32_t)strtoul(e, nullptr, 0) : 0u; return v > 8192u ? 8192u : v; }();
	const int run_occ_v = run_occ ? run_occ : (use_pairs ? 5 : 4);
	const uint32_t run_stage = run_stage_env ? run_stage_env : (use_pairs ? 2048u : RUN_STAGE);   // (2048 x 12 B + tables = 26 KiB: five workgroups per CU with room to spare)
	for (int l = 0; l < 16; ++l) {                                       // (coarsest level first measured 1 % faster than finest first on both samplings)
		in_bins[l] = use_bins && level_binned(lt, l);
		if (!in_bins[l]) continue;
		const uint32_t hl = bp.n_levels++;
		bp.level[hl] = (uint32_t)l;
		if (lt.v[4 * l + 2] <= run_res_max) { sel_runs.hl[n_runs++] = hl; if (!use_pairs) sel_all.hl[n_all++] = hl; }
		else if (use_pairs && level_pair_capable(lt, l)) sel_pair.hl[n_pair++] = hl;
		else { sel_fine.hl[n_fine++] = hl; sel_all.hl[n_all++] = hl; }
	}
	uint64_t slab_cursor = 0;
	for (int l = 0; l < 16; ++l) {
		slices[l] = div_up(lt.v[4 * l + 1], OWN_SLICE);
		plan.slab_off[l] = ~0u;
		plan.chunks[l] = 1u;
		if (in_bins[l] || level_exclusive(lt, l)) continue;
		// Sample chunks per slice of a level without an exclusive owner.  A unit (slice, chunk) pays a fixed price - clear and flush 128 KiB of LDS, one partial
		// slab to write and later re-read - and holds a whole CU while it runs.  Small, heavily contended coarse levels want many short units, large levels few
		// long ones: with slabs 144 / slices clamped to [8, 32] (swept on fox- and lego-like batches when this scan still carried the dense levels of the training path).
		uint32_t c = use_slabs ? 144u / slices[l] : (32u / slices[l] ? 32u / slices[l] : 1u);
		if (use_slabs && c < 8u) c = 8u;
		plan.chunks[l] = c > 32u ? 32u : (use_slabs && c < 2u ? 2u : c);          // with slabs >= 2: the exclusive-owner (chunks == 1) branch does not write slabs
		if (use_slabs) { plan.slab_off[l] = (uint32_t)slab_cursor; slab_cursor += (uint64_t)plan.chunks[l] * lt.v[4 * l + 1]; }
	}
	for (int pass = 0; pass < 2; ++pass)                                  // chunked levels first (their hot slices are the long poles), largest level first
		for (int l = 15; l >= 0; --l)
			if ((plan.chunks[l] > 1) == (pass == 0)) {
				plan.order[k] = (uint32_t)l; plan.first_unit[k] = units;
				if (!in_bins[l]) units += slices[l] * plan.chunks[l];              // binned levels get no scan units
				++k;
			}
	plan.first_unit[16] = units;
	char *ws = (char *)workspace;
	uint32_t *cursors = use_bins ? (uint32_t *)(ws + wl.cursors) : nullptr;
	uint32_t *absmax = use_bins ? (uint32_t *)(ws + wl.absmax) : nullptr;
	uint32_t *spill_count = use_bins ? absmax + 16u * ABSMAX_PARTS : nullptr;
	void *rec_val = use_bins ? (void *)(ws + wl.rec_val) : nullptr;
	uint16_t *rec_idx = use_bins ? (uint16_t *)(ws + wl.rec_idx) : nullptr;
	SpillEntry *spill = use_bins ? (SpillEntry *)(ws + wl.spill) : nullptr;
	PairRec *pair_rec = use_bins ? (PairRec *)(ws + wl.pair_rec) : nullptr;
	uint16_t *pair_off = use_bins ? (uint16_t *)(ws + wl.pair_off) : nullptr;
	RunRec *run_rec = use_bins ? (RunRec *)(ws + wl.run_rec) : nullptr;
	uint16_t *run_off = use_bins ? (uint16_t *)(ws + wl.run_off) : nullptr;
	{ const char *e = getenv("NGP_PROBE_LEVEL_MASK"); plan.level_mask = e ? (uint32_t)strtoul(e, nullptr, 0) : 0xffffu; }
	{ const char *e = getenv("NGP_PROBE_COARSE_RES"); plan.coarse_res = e ? (uint32_t)strtoul(e, nullptr, 0) : 0u; }
	for (int l = 0; l < 16; ++l) {                                       // chunked levels without slabs are flushed with atomics -> need a zeroed destination
		if (!in_bins[l] && plan.chunks[l] > 1 && zero_first && !use_slabs) {
			hipError_t e = hipMemsetAsync((char *)grad + (size_t)lt.v[4 * l] * 2 * gsz, 0, (size_t)lt.v[4 * l + 1] * 2 * gsz, s);
			if (e != hipSuccess) { ngp_set_error("ngp_hash_encode_bwd memset: %s", hipGetErrorString(e)); return (int)e; }
		}
	}
	if (level_scratch) { hipError_t e = hipMemsetAsync(level_scratch, 0, 16 * sizeof(float), s); if (e != hipSuccess) { ngp_set_error("ngp_hash_encode_bwd memset: %s", hipGetErrorString(e)); return (int)e; } }
	const int accumulate = (zero_first ? 0 : 1) | ((getenv("NGP_PROBE_NO_LDS_ATOMICS") != nullptr) ? 2 : 0);
	const size_t shmem = (size_t)OWN_SLICE * 2 * sizeof(float);
	const dim3 grid(units), block(1024);
	const bool probe_skip_bins = getenv("NGP_PROBE_SKIP_BINS") != nullptr;      // tools/probe_scatter.py: time the scan kernel alone
	const int ow = zero_first ? 1 : 0;
	bool coarse_marked = false;
	int pair_err = 0;
	auto pair_set_lds = [&](const void *k, size_t bytes) { hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); if (e != hipSuccess) { ngp_set_error("ngp_hash_encode_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); pair_err = (int)e; } };
	// ---- the fp32 path (use_pairs): run records + edge records in regions, no global atomics, one accumulate kernel.  (Levels it cannot take - hashed tables that are
	// neither run levels nor edge-capable - stay on the per-corner kernels below, with their own accumulate launch.)
	bool any_split = false;                                              // an edge level beyond res 2048 in this call: 512-sample record workgroups (eight single records per sample)
	for (uint32_t k = 0; k < n_pair; ++k) any_split |= lt.v[4 * bp.level[sel_pair.hl[k]] + 2] > PAIR_RES_MAX;
	const uint32_t pair_s = any_split ? 512u : pair_wg;
	auto v2_set_lds = [&]() {
		static bool once = false;
		if (once) return;
#define PSET(T) pair_set_lds((const void *)k_bin_pairs<T, NGP_LAYOUT_SOA, 512u>, pair_stage_bytes()); pair_set_lds((const void *)k_bin_pairs<T, NGP_LAYOUT_AOS, 512u>, pair_stage_bytes()); \
		pair_set_lds((const void *)k_bin_pairs<T, NGP_LAYOUT_SOA, 1024u>, pair_stage_bytes()); pair_set_lds((const void *)k_bin_pairs<T, NGP_LAYOUT_AOS, 1024u>, pair_stage_bytes()); \
		pair_set_lds((const void *)k_bin_runs2<T, NGP_LAYOUT_SOA, 4>, run2_stage_bytes(8192u)); pair_set_lds((const void *)k_bin_runs2<T, NGP_LAYOUT_AOS, 4>, run2_stage_bytes(8192u)); \
		pair_set_lds((const void *)k_bin_runs2<T, NGP_LAYOUT_SOA, 5>, run2_stage_bytes(8192u)); pair_set_lds((const void *)k_bin_runs2<T, NGP_LAYOUT_AOS, 5>, run2_stage_bytes(8192u));
		PSET(float) PSET(__half)
#undef PSET
		pair_set_lds((const void *)k_bin_accumulate2<float, true>, PAIR_BIN_ENTRIES * 16u + ACC2_LDS_EXTRA); pair_set_lds((const void *)k_bin_accumulate2<float, false>, PAIR_BIN_ENTRIES * 16u + ACC2_LDS_EXTRA);
		once = true;
	};
	const uint32_t run_probe = [] { const char *e = getenv("NGP_RUN_PROBE"); return e ? (uint32_t)atoi(e) : 0u; }();
	const uint32_t pair_probe = [] { const char *e = getenv("NGP_PAIR_PROBE"); return e ? (uint32_t)atoi(e) : 0u; }();
	// ---- the region path (use_pairs): run records + edge records in regions, no global atomics, one accumulate kernel.  (Levels it cannot take - small hashed tables
	// that are not run levels - stay on the per-corner kernels below, with their own accumulate launch.)
#define V2_RECORDS(T) do { v2_set_lds(); \
	if (n_runs) { \
		if (in_layout == NGP_LAYOUT_SOA) { if (run_occ_v == 5) V2_RGO(T, NGP_LAYOUT_SOA, 5); else V2_RGO(T, NGP_LAYOUT_SOA, 4); } \
		else { if (run_occ_v == 5) V2_RGO(T, NGP_LAYOUT_AOS, 5); else V2_RGO(T, NGP_LAYOUT_AOS, 4); } } \
	if (n_pair) { \
		if (in_layout == NGP_LAYOUT_SOA) { if (pair_s == 512u) V2_PGO(T, NGP_LAYOUT_SOA, 512u); else V2_PGO(T, NGP_LAYOUT_SOA, 1024u); } \
		else { if (pair_s == 512u) V2_PGO(T, NGP_LAYOUT_AOS, 512u); else V2_PGO(T, NGP_LAYOUT_AOS, 1024u); } } } while (0)
#define V2_RGO(T, L, O) NGP_LAUNCH((k_bin_runs2<T, L, O>), dim3(div_up(n, RUN_WG * RUN_K), n_runs), dim3(RUN_WG), run2_stage_bytes(run_stage), s, n, pos, pos_stride, (const T *)dLdy, lt, bp, sel_runs, (const uint32_t *)absmax, run_rec, run_off, n_valid, run_stage, run_probe)
#define V2_PGO(T, L, S) NGP_LAUNCH_INDEPENDENT((k_bin_pairs<T, L, S>), dim3(div_up(n, S), n_pair), dim3(S), pair_stage_bytes(), s, n, pos, pos_stride, (const T *)dLdy, lt, bp, sel_pair, (const uint32_t *)absmax, pair_rec, pair_off, spill_count, spill, n_valid, pair_probe)
	uint32_t acc2_launches = 0;
	auto v2_accumulate = [&](bool runs, bool pairs) {
		Acc2Plan ap;
		ap.n_pair = pairs ? n_pair : 0u; ap.n_run = runs ? n_runs : 0u;
		ap.pair_regions = div_up(n, pair_s); ap.pair_region_records = pair_region_records();
		ap.run_regions = div_up(n, RUN_WG * RUN_K); ap.run_region_records = run2_region_records();
		ap.probe = [] { const char *e = getenv("NGP_ACC_PROBE"); return e ? (uint32_t)atoi(e) : 0u; }();
		if (ap.n_pair + ap.n_run == 0) return;
		const uint32_t acc2_grid = [] { const char *e = getenv("NGP_ACC_GRID"); return e ? (uint32_t)strtoul(e, nullptr, 0) : 0u; }();   // probe hook: > 0 = that many resident workgroups drawing units from a queue
		const uint32_t n_units = (ap.n_pair + ap.n_run) * PAIR_BINS;
		uint32_t *qh = cursors + N_CURSORS + (acc2_launches++);
#define AGO(Q, GRID) NGP_LAUNCH((k_bin_accumulate2<float, Q>), dim3(GRID), dim3(ACC2_WG), PAIR_BIN_ENTRIES * 16u + ACC2_LDS_EXTRA, s, lt, bp, sel_pair, sel_runs, ap, (const uint32_t *)absmax, (const PairRec *)pair_rec, \
		           (const uint16_t *)pair_off, (const RunRec *)run_rec, (const uint16_t *)run_off, (const uint32_t *)spill_count, (const SpillEntry *)spill, (float *)grad, ow, qh)
		if (acc2_grid) AGO(true, min(n_units, acc2_grid)); else AGO(false, n_units);
#undef AGO
	};
#define SET_LDS(K, BYTES) do { static bool done_ = false; if (!done_) { hipError_t e = hipFuncSetAttribute((const void *)K, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES)); \
	if (e != hipSuccess) { ngp_set_error("ngp_hash_encode_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; } done_ = true; } } while (0)
#define GO(T, G, L) do { \
	using RV_ = typename RecVal<T>::type; \
	if (units) SET_LDS((k_hash_bwd_owner<T, G, L>), shmem); \
	if (level_scratch) NGP_LAUNCH((k_level_l1<T, L>), dim3(64, 16), dim3(256), 0, s, n, (const T *)dLdy, level_scratch, n_valid); \
	hipStream_t sd = s; \
	if (use_bins && bp.n_levels) { \
		if (!absmax_done) NGP_LAUNCH((k_level_absmax<T, L>), dim3(ABSMAX_OWN_PARTS, 16), dim3(256), 0, s, n, (const T *)dLdy, absmax, n_valid, cursors, spill_count);   /* also zeroes the cursors and the spill count */ \
		if (units && side.ok) { hipEventRecord(side.fork, s); sd = side.stream; hipStreamWaitEvent(sd, side.fork, 0); }   /* the scan of the remaining levels runs beside the binning kernels */ \
		if (!probe_skip_bins) { \
		if (use_pairs) V2_RECORDS(T); \
		else if (n_runs && run_occ_v == 5) { SET_LDS((k_bin_records_runs<T, L, 5>), run_stage_bytes(8192u)); \
			NGP_LAUNCH((k_bin_records_runs<T, L, 5>), dim3(div_up(n, RUN_WG * RUN_K), n_runs), dim3(RUN_WG), run_stage_bytes(run_stage), s, n, pos, pos_stride, (const T *)dLdy, lt, bp, sel_runs, (const uint32_t *)absmax, cursors, rec_val, rec_idx, spill_count, spill, n_valid, run_stage); } \
		else if (n_runs) { SET_LDS((k_bin_records_runs<T, L, 4>), run_stage_bytes(8192u)); \
			NGP_LAUNCH((k_bin_records_runs<T, L, 4>), dim3(div_up(n, RUN_WG * RUN_K), n_runs), dim3(RUN_WG), run_stage_bytes(run_stage), s, n, pos, pos_stride, (const T *)dLdy, lt, bp, sel_runs, (const uint32_t *)absmax, cursors, rec_val, rec_idx, spill_count, spill, n_valid, run_stage); } \
		if (n_fine) { SET_LDS((k_bin_records<T, L>), bin_stage_bytes<T>()); \
			NGP_LAUNCH((k_bin_records<T, L>), dim3(div_up(n, BIN_WG), n_fine), dim3(BIN_WG), bin_stage_bytes<T>(), s, n, pos, pos_stride, (const T *)dLdy, lt, bp, sel_fine, (const uint32_t *)absmax, cursors, rec_val, rec_idx, spill_count, spill, n_valid); } \
		if (use_pairs) {   /* fp32 path: ONE accumulate launch over run + edge levels - two when the data-parallel exchange wants the coarse levels first */ \
			if (after_coarse && n_runs && (n_pair || n_fine)) { v2_accumulate(true, false); hipEventRecord(after_coarse, s); coarse_marked = true; v2_accumulate(false, true); } \
			else v2_accumulate(true, true); \
			if (n_fine) { SET_LDS((k_bin_accumulate<G, RV_>), BIN_ENTRIES * 16); \
				NGP_LAUNCH((k_bin_accumulate<G, RV_>), dim3(n_fine * BINS_PER_LEVEL), dim3(1024), BIN_ENTRIES * 16, s, lt, bp, sel_fine, (const uint32_t *)absmax, (const uint32_t *)cursors, rec_val, rec_idx, (const uint32_t *)spill_count, (const SpillEntry *)spill, (G *)grad, ow); } \
		} else if (sizeof(RV_) == 8 && !(after_coarse && n_runs && n_fine)) {   /* fp32 records, one type: one accumulate launch over all their levels */ \
			if (n_all) { SET_LDS((k_bin_accumulate<G, float2>), BIN_ENTRIES * 16); \
				NGP_LAUNCH((k_bin_accumulate<G, float2>), dim3(n_all * BINS_PER_LEVEL), dim3(1024), BIN_ENTRIES * 16, s, lt, bp, sel_all, (const uint32_t *)absmax, (const uint32_t *)cursors, rec_val, rec_idx, (const uint32_t *)spill_count, (const SpillEntry *)spill, (G *)grad, ow); } \
		} else { \
			if (n_runs) { SET_LDS((k_bin_accumulate<G, float2>), BIN_ENTRIES * 16); \
				NGP_LAUNCH((k_bin_accumulate<G, float2>), dim3(n_runs * BINS_PER_LEVEL), dim3(1024), BIN_ENTRIES * 16, s, lt, bp, sel_runs, (const uint32_t *)absmax, (const uint32_t *)cursors, rec_val, rec_idx, (const uint32_t *)spill_count, (const SpillEntry *)spill, (G *)grad, ow); \
				if (after_coarse && n_fine) { hipEventRecord(after_coarse, s); coarse_marked = true; } } \
			if (n_fine) { SET_LDS((k_bin_accumulate<G, RV_>), BIN_ENTRIES * 16); \
				NGP_LAUNCH((k_bin_accumulate<G, RV_>), dim3(n_fine * BINS_PER_LEVEL), dim3(1024), BIN_ENTRIES * 16, s, lt, bp, sel_fine, (const uint32_t *)absmax, (const uint32_t *)cursors, rec_val, rec_idx, (const uint32_t *)spill_count, (const SpillEntry *)spill, (G *)grad, ow); } \
		} } \
	} \
	if (units) NGP_LAUNCH((k_hash_bwd_owner<T, G, L>), grid, block, shmem, sd, n, pos, pos_stride, (const T *)dLdy, lt, plan, (G *)grad, accumulate, n_valid, (const float *)level_scratch, use_slabs ? (float2 *)workspace : (float2 *)nullptr); \
	if (units && use_slabs && slab_cursor) NGP_LAUNCH((k_reduce_dense<G>), dim3(1024, 16), dim3(256), 0, sd, lt, plan, (const float2 *)workspace, (G *)grad, accumulate & 1); \
	if (sd != s) { hipEventRecord(side.join, sd); hipStreamWaitEvent(s, side.join, 0); } } while (0)
	if (dtype == NGP_F32) { if (in_layout == NGP_LAYOUT_SOA) GO(float, float, NGP_LAYOUT_SOA); else GO(float, float, NGP_LAYOUT_AOS); }
	else if (grad_dtype == NGP_F32) { if (in_layout == NGP_LAYOUT_SOA) GO(__half, float, NGP_LAYOUT_SOA); else GO(__half, float, NGP_LAYOUT_AOS); }
	else { if (in_layout == NGP_LAYOUT_SOA) GO(__half, __half, NGP_LAYOUT_SOA); else GO(__half, __half, NGP_LAYOUT_AOS); }
#undef GO
#undef SET_LDS
	if (pair_err) return pair_err;
	NGP_LAUNCH_CHECK("ngp_hash_encode_bwd");
	if (after_coarse && !coarse_marked) hipEventRecord(after_coarse, s);      // no separate coarse launch on this path: the marker follows the whole scatter
	return 0;
}

// the workspace path with the data-parallel marker and the fused abs-max (csrc/train_step.hip); not part of the public ABI
int ngp_hash_encode_bwd_ws_marked(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *dLdy, const uint32_t *level_table_host, void *grad, uint64_t n_params, int dtype,
                                  int grad_dtype, int in_layout, int zero_first, const uint32_t *n_valid, void *workspace, uint64_t workspace_bytes, hipEvent_t after_coarse, int absmax_done) {
	return hash_bwd_impl(stream, n, pos, pos_stride, dLdy, level_table_host, grad, n_params, dtype, grad_dtype, in_layout, zero_first, n_valid, nullptr, workspace, workspace_bytes, after_coarse, absmax_done != 0);
}
// mirrors the routing decisions of hash_bwd_impl: the slots are handed out only when that call will read them
AbsmaxOut ngp_hash_bwd_absmax_slots(const uint32_t *level_table_host, uint32_t n, int dtype, int grad_dtype, void *workspace, uint64_t workspace_bytes) {
	AbsmaxOut am{nullptr, nullptr, 0u, nullptr};
	if (!workspace || !level_table_host || n == 0 || hash_bwd_method() == 1 || getenv("NGP_HASH_BWD_NO_BINS") || getenv("NGP_NO_FUSED_ABSMAX")) return am;
	const LevelTable lt = load_table(level_table_host);
	for (int l = 0; l < 16; ++l) {
		const uint32_t size = lt.v[4 * l + 1], res = lt.v[4 * l + 2];
		if (!level_dense_host(size, res) && (size & (size - 1)) != 0) return am;            // (non-power-of-two hashed table: the atomic path)
		if (!level_binned(lt, l)) return am;                                                 // a level outside the bins would still need the owner-computes scan: keep the plain sequence
	}
	const WsLayout wl = ws_layout(lt, n);
	if (!(dtype == NGP_F16 || grad_dtype == NGP_F32) || workspace_bytes < wl.total) return am;
	char *ws = (char *)workspace;
	am.parts = (uint32_t *)(ws + wl.absmax); am.cursors = (uint32_t *)(ws + wl.cursors); am.n_cursors = N_ZEROED; am.spill_count = am.parts + 16u * ABSMAX_PARTS;
	return am;
}

NGP_API int ngp_hash_encode_bwd(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *dLdy, const uint32_t *level_table_host,
                                void *grad, uint64_t n_params, int dtype, int grad_dtype, int in_layout, int zero_first, const uint32_t *n_valid) {
	return hash_bwd_impl(stream, n, pos, pos_stride, dLdy, level_table_host, grad, n_params, dtype, grad_dtype, in_layout, zero_first, n_valid, nullptr, nullptr, 0);
}
NGP_API int ngp_hash_encode_bwd_fx(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *dLdy, const uint32_t *level_table_host,
                                   void *grad, uint64_t n_params, int dtype, int grad_dtype, int in_layout, int zero_first, const uint32_t *n_valid, float *level_scratch) {
	NGP_REQUIRE(level_scratch, NGP_E_ARG, "ngp_hash_encode_bwd_fx: level_scratch (device f32[16]) is required");
	return hash_bwd_impl(stream, n, pos, pos_stride, dLdy, level_table_host, grad, n_params, dtype, grad_dtype, in_layout, zero_first, n_valid, level_scratch, nullptr, 0);
}

NGP_API int ngp_hash_encode_bwd_ws(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *dLdy, const uint32_t *level_table_host,
                                   void *grad, uint64_t n_params, int dtype, int grad_dtype, int in_layout, int zero_first, const uint32_t *n_valid,
                                   float *level_scratch, void *workspace, uint64_t workspace_bytes) {
	return hash_bwd_impl(stream, n, pos, pos_stride, dLdy, level_table_host, grad, n_params, dtype, grad_dtype, in_layout, zero_first, n_valid, level_scratch, workspace, workspace_bytes);
}

// test hook (tests/test_host_cpu.py): the balanced forward map as host arrays, u32[8][FWD_MAP_SEGS][3] = (level, first chunk, chunks); returns blocks per XCD
NGP_API uint32_t ngp_x_fwd_map(const uint32_t *level_table_host, uint32_t nblk, float light, uint32_t *out_host) {
	const FwdMap m = light == -3.0f ? fwd_map_light_aside(load_table(level_table_host), nblk) : light < 0.f ? fwd_map_helpers(load_table(level_table_host), nblk, -light) : fwd_map_balanced(load_table(level_table_host), nblk, light);
	for (int x = 0; x < 8; ++x) for (int g = 0; g < FWD_MAP_SEGS; ++g) { out_host[(x * FWD_MAP_SEGS + g) * 3] = m.level[x][g]; out_host[(x * FWD_MAP_SEGS + g) * 3 + 1] = m.begin[x][g]; out_host[(x * FWD_MAP_SEGS + g) * 3 + 2] = m.count[x][g]; }
	return m.slots;
}

// ---------------------------------------------------------------------------------------------------------------- probes (tools/microbench_hash.py only)
// Not part of the public ABI: lets the micro-benchmark time one level at a time and compare atomic scopes.  scope 0 = agent, 1 = workgroup
// (an L2-local atomic: only valid when every accessor of an address sits on one XCD — used here for TIMING the idea, not for results).
template <int SCOPE, bool PK16>
__global__ __launch_bounds__(256) void k_probe_bwd(uint32_t n, const float *__restrict__ pos, const __half2 *__restrict__ dy, LevelTable lt, void *__restrict__ grad, uint32_t level_fixed) {
	const uint32_t level = level_fixed;
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	const uint32_t off = lt.v[4 * level], size = lt.v[4 * level + 1], res = lt.v[4 * level + 2];
	const float scale = __uint_as_float(lt.v[4 * level + 3]);
	const bool dense = level_is_dense(size, res);
	const float2 g2 = __half22float2(dy[(size_t)level * n + i]);
	const Corner c = locate(pos, 3, i, scale);
#pragma unroll
	for (uint32_t k = 0; k < 8; ++k) {
		float weight = 1; uint32_t g[3];
#pragma unroll
		for (int d = 0; d < 3; ++d) { if ((k & (1u << d)) == 0) { weight *= 1 - c.w[d]; g[d] = c.g[d]; } else { weight *= c.w[d]; g[d] = c.g[d] + 1; } }
		const uint32_t idx = grid_index(size, res, dense, g[0], g[1], g[2]);
		if (PK16) {
			typedef _Float16 __attribute__((ext_vector_type(2))) h2;
			h2 x; x[0] = (_Float16)(g2.x * weight); x[1] = (_Float16)(g2.y * weight);
			h2 *p = reinterpret_cast<h2 *>(grad) + off + idx;
			if (SCOPE == 0) (void)__builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2 *)p, x);
			else asm volatile("global_atomic_pk_add_f16 %0, %1, off" :: "v"(p), "v"(x) : "memory");   // no sc bits: performed in the issuing XCD's L2
		} else {
			float *p = reinterpret_cast<float *>(grad) + ((size_t)off + idx) * 2;
			if (SCOPE == 0) { __hip_atomic_fetch_add(p, g2.x * weight, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_fetch_add(p + 1, g2.y * weight, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
			else { __hip_atomic_fetch_add(p, g2.x * weight, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(p + 1, g2.y * weight, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
		}
	}
}
NGP_API int ngp_x_probe_hash_bwd(void *stream, uint32_t n, const float *pos, const void *dy, const uint32_t *level_table_host, void *grad, uint32_t level, int scope, int pk16) {
	const LevelTable lt = load_table(level_table_host);
	const dim3 grid(div_up(n, 256)), block(256);
	hipStream_t s = (hipStream_t)stream;
#define GO(S, P) NGP_LAUNCH((k_probe_bwd<S, P>), grid, block, 0, s, n, pos, (const __half2 *)dy, lt, grad, level)
	if (scope == 0) { if (pk16) GO(0, true); else GO(0, false); } else { if (pk16) GO(1, true); else GO(1, false); }
#undef GO
	NGP_LAUNCH_CHECK("ngp_x_probe_hash_bwd");
	return 0;
}
