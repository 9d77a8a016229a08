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
//    reservations (k_bin_records_runs / k_bin_records / k_bin_accumulate).  Without a workspace, with NGP_HASH_BWD_ATOMICS=1, or for a table the bins cannot take (a level
//    beyond 2^19 entries, a hashed table that is not a power of two): the reference's scheme, one global float atomic per corner (k_hash_bwd) - the ONE fallback since the
//    owner-computes scan of rounds 1-2 was deleted in round 5.  (r6) On the single-GPU training path the accumulate kernels of both workspace designs also apply the
//    table's Adam + EMA sweep in place of the gradient store (AdamRide: k_bin_accumulate2<float, true>, k_bin_accumulate<float, RV, true>).
#include "ngp_common.h"
#include <stdlib.h>
#include <string.h>
#include <mutex>
#pragma clang fp contract(off)

#include "hash_common.h"
#include "mlp_tail.h"

template <typename T, int LAYOUT>
__device__ __forceinline__ void hash_fwd_body(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ table, const LevelTable &lt,
                                              T *__restrict__ out, uint32_t nblk, const uint32_t *__restrict__ n_valid, uint32_t level, uint32_t chunk, float *__restrict__ dy_dx = nullptr);
// forward with d(encoding)/d(position) - the dy_dx branch of the reference's kernel_grid (HashEncode.h:205-251): same gathers, three more outputs per (sample, level)
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void k_hash_fwd_dydx(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ table, LevelTable lt,
                                                       T *__restrict__ out, uint32_t nblk, const uint32_t *__restrict__ n_valid, float *__restrict__ dy_dx) {
	uint32_t level, chunk; block_to_level_chunk(nblk, level, chunk);
	hash_fwd_body<T, LAYOUT>(n, pos, stride, table, lt, out, nblk, n_valid, level, chunk, dy_dx);
}
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void k_hash_fwd(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ table, LevelTable lt,
                                                  T *__restrict__ out, uint32_t nblk, const uint32_t *__restrict__ n_valid) {
	uint32_t level, chunk; block_to_level_chunk(nblk, level, chunk);
	hash_fwd_body<T, LAYOUT>(n, pos, stride, table, lt, out, nblk, n_valid, level, chunk);
}
template <typename T, int LAYOUT>
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
	// The two x-neighbours of a cell edge are adjacent in memory on dense levels (index x + ...; not across the wrap) and on hashed levels when x is even
	// ((x+1) ^ h == (x ^ h) ^ 1 then).  fp16 table: they are fetched with ONE 8-byte load then - 6 requests per (sample, level) on average instead of 8; the second single load
	// is issued only by the other lanes.  fp32 table (r5): NOT - rounds 1-4 used 16-byte loads there and measured them against nothing; an A/B in one call
	// (profiles/r05e_wide_loads_ab.txt) has the plain eight 8-byte loads FASTER (k_hash_fwd 101 -> 86 us per iteration incl. the refresh share, +2.4 % it/s: the three-way
	// branch serialises a wavefront's loads), and the 16-byte gathers were the one thing that made this kernel return wrong values for a quarter wavefront while ANOTHER PROCESS
	// trained with this package on the same GPU (profiles/r05d_shared_gpu.txt, DESIGN.md 6: 506 of 2909 launches with them, 0 of 3369 without; alone, or with two streams in one
	// process, never).  -DNGP_PROBE_WIDE_LOADS_F32 rebuilds the old kernel for that diagnosis (tools/probe_shared_gpu.sh).
	struct alignas(sizeof(P)) PP { P a, b; };
#ifdef NGP_PROBE_WIDE_LOADS_F32
	constexpr bool WIDE = true;
#else
	constexpr bool WIDE = sizeof(P) == 4;
#endif
	const bool pow2 = (size & (size - 1)) == 0;
#pragma unroll
	for (uint32_t j = 0; j < 4; ++j) {        // j = (y corner, z corner); all gathers are issued before the first use
		const uint32_t gy = c.g[1] + (j & 1u), gz = c.g[2] + (j >> 1);
		const float wy = (j & 1u) ? c.w[1] : 1 - c.w[1], wz = (j >> 1) ? c.w[2] : 1 - c.w[2];
		w[2 * j] = ((1 - c.w[0]) * wy) * wz; w[2 * j + 1] = (c.w[0] * wy) * wz;          // the reference's x, y, z multiplication order
		const uint32_t i0 = grid_index(size, res, dense, c.g[0], gy, gz), i1 = grid_index(size, res, dense, c.g[0] + 1, gy, gz);
		const bool adjacent_up = WIDE && i1 == i0 + 1u && (dense || pow2), adjacent_dn = WIDE && i0 == i1 + 1u && !dense && pow2;     // (x^h)^1 is either one above or one below
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

// (r6) fp32 table, two LANES per (sample, level): lane pair (2p, 2p + 1) = the cell's x = g and x = g + 1 faces, four 8-byte gathers each.  The forward gather is bound by
// the number of distinct cache lines a wavefront's load instruction names (~one line per clock through the CU's texture path: 33.5 M lines of a 2^18-sample batch = 54 us of
// the kernel's 65), and the two x-neighbours of a cell edge are adjacent 8-byte entries - always on a dense level, and on a hashed one whenever x is even
// ((x + 1) ^ h == (x ^ h) ^ 1) - i.e. ONE line.  One lane loading both (the 16-byte pair loads of rounds 1-4) needed a three-way branch that serialised the loads; two
// adjacent lanes loading one each in the SAME instruction are merged by the address coalescer with no branch at all: 6 lines per (sample, hashed level) on average instead
// of 8.  Bit-exact: the lanes swap halves (lane 2p ends up with every corner's first component, lane 2p + 1 with the second) and each sums its component's eight terms in the
// reference's order, k = x + 2 y + 4 z.
template <typename T, int LAYOUT>
__global__ __launch_bounds__(256) void k_hash_fwd_x2(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ table, LevelTable lt,
                                                     T *__restrict__ out, uint32_t nblk, const uint32_t *__restrict__ n_valid) {
	using P = typename Pair<T>::type;
	uint32_t level, chunk; block_to_level_chunk(nblk, level, chunk);
	uint32_t lim = n; if (n_valid) { uint32_t nv = *n_valid; lim = nv < n ? nv : n; }
	const uint32_t off = lt.v[4 * level], size = lt.v[4 * level + 1], res = lt.v[4 * level + 2];
	const float scale = __uint_as_float(lt.v[4 * level + 3]);
	const bool dense = level_is_dense(size, res);
	const P *tab = reinterpret_cast<const P *>(table) + off;
	const uint32_t xc = threadIdx.x & 1u;
	// (both lanes of a pair take the same trips, so the swap below always has its partner)
	for (uint32_t i = chunk * 128u + (threadIdx.x >> 1); i < lim; i += nblk * 128u) {
		const Corner c = locate(pos, stride, i, scale);
		const uint32_t gx = c.g[0] + xc;
		P v[4];
#pragma unroll
		for (uint32_t j = 0; j < 4; ++j) v[j] = tab[grid_index(size, res, dense, gx, c.g[1] + (j & 1u), c.g[2] + (j >> 1))];      // j = (y corner, z corner); all four gathers issued before the first use
		float acc = 0.f;
#pragma unroll
		for (uint32_t j = 0; j < 4; ++j) {
			const float wy = (j & 1u) ? c.w[1] : 1 - c.w[1], wz = (j >> 1) ? c.w[2] : 1 - c.w[2];
			const float w0 = ((1 - c.w[0]) * wy) * wz, w1 = (c.w[0] * wy) * wz;          // the reference's x, y, z multiplication order
			const float2 f = to_f2(v[j]);
			const float own = xc ? f.y : f.x, give = xc ? f.x : f.y;
			const float got = __shfl_xor(give, 1);                                      // the partner's value of MY component
			acc += w0 * (xc ? got : own);                                                // corner k = 2 j     (x = g)
			acc += w1 * (xc ? own : got);                                                // corner k = 2 j + 1 (x = g + 1)
		}
		const size_t o = (LAYOUT == NGP_LAYOUT_SOA ? (size_t)level * n + i : (size_t)i * 16 + level) * 2u + xc;
		if (sizeof(T) == 4) reinterpret_cast<float *>(out)[o] = acc;
		else reinterpret_cast<__half *>(out)[o] = __float2half_rn(acc);
	}
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


__device__ __forceinline__ float bin_scale(uint32_t absmax_bits) {     // power of two s with 2^13 <= max*s < 2^14 (0 if the level has no gradient)
	const float m = __uint_as_float(absmax_bits);
	if (!(m > 0.f) || !(m < 3.0e38f)) return 0.f;
	int ex; frexpf(m, &ex);                                           // m = f * 2^ex, f in [0.5, 1)
	return ldexpf(1.0f, 14 - ex);
}

NGP_API int ngp_hash_encode_fwd(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *table, const uint32_t *level_table_host,
                                void *out, int dtype, int out_layout, const uint32_t *n_valid) {
	NGP_REQUIRE(n == 0 || (pos && table && level_table_host && out), NGP_E_ARG, "ngp_hash_encode_fwd: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_hash_encode_fwd: bad dtype %d", dtype);
	if (n == 0) return 0;
	NGP_REQUIRE(pos_stride >= 3, NGP_E_ARG, "ngp_hash_encode_fwd: pos stride %u < 3", pos_stride);
	const dim3 block(256);
	const LevelTable lt = load_table(level_table_host);
	hipStream_t s = (hipStream_t)stream;
	// A/B hook NGP_HASH_FWD_X2: bit 0 = two lanes per (sample, level) for the fp32 table (default on), bit 1 = for the fp16 table (default off: its 8-byte pair loads stay)
	static const int x2 = [] { const char *e = getenv("NGP_HASH_FWD_X2"); return e ? atoi(e) : 1; }();
	if ((dtype == NGP_F32 && (x2 & 1)) || (dtype == NGP_F16 && (x2 & 2))) {            // two lanes per (sample, level): 128 samples per workgroup
		const uint32_t nblk2 = min(div_up(n, 128), 4096u);
#define GO2(T, L) NGP_LAUNCH((k_hash_fwd_x2<T, L>), dim3(16 * nblk2), block, 0, s, n, pos, pos_stride, (const T *)table, lt, (T *)out, nblk2, n_valid)
		if (dtype == NGP_F32) { if (out_layout == NGP_LAYOUT_SOA) GO2(float, NGP_LAYOUT_SOA); else GO2(float, NGP_LAYOUT_AOS); }
		else { if (out_layout == NGP_LAYOUT_SOA) GO2(__half, NGP_LAYOUT_SOA); else GO2(__half, NGP_LAYOUT_AOS); }
#undef GO2
		NGP_LAUNCH_CHECK("ngp_hash_encode_fwd");
		return 0;
	}
	const uint32_t nblk = min(div_up(n, 256), 2048u);         // chunks per level in flight (k_hash_fwd strides over the rest)
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
// method: 0 = by what the caller hands over (a workspace: the binned scatter; none: global atomics), 1 = one global atomic per corner whatever is handed over
// (the reference's scheme; NGP_HASH_BWD_ATOMICS=1)
static int hash_bwd_method() {
	static int m = -1;
	if (m < 0) { const char *e = getenv("NGP_HASH_BWD_ATOMICS"); m = (e && e[0] == '1') ? 1 : 0; }
	return m;
}


// ---------------------------------------------------------------------------------------------------------------- binned scatter (every level of up to 2^19 entries)
// (Rounds 1-2 scattered through an owner-computes scan - every slice owner re-deriving every sample's indices, VALU-bound at ~0.45 ms per 2^18-sample batch; deleted in
// round 5.)  With a workspace the levels take this two-phase path:
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
#define N_ZEROED (N_CURSORS + 16u)                                        // (+ spare words) zeroed together with the cursors every step
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
                                                           const uint32_t *__restrict__ n_valid, uint32_t stage /* records of LDS staging */, TailJobs tj) {
	using P = typename Pair<T>::type;
	extern __shared__ __attribute__((aligned(16))) uint32_t bin_smem[];
	const uint32_t by = blockIdx.y;
	if (tj.do_reduce) tail_reduce_share(tj, reinterpret_cast<float *>(bin_smem), by * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);      // (r6, mlp_tail.h) this workgroup's 8 columns of the MLP weight-gradient slabs (+ the two packs' sweep)
	float2 *stage_val = reinterpret_cast<float2 *>(bin_smem);                             // [stage]
	uint32_t *stage_idx = bin_smem + stage * 2u;                                          // [stage] level-wide entry indices
	uint32_t *cnt = stage_idx + stage, *base = cnt + BINS_PER_LEVEL, *loff = base + BINS_PER_LEVEL, *cnt2 = loff + BINS_PER_LEVEL;
	const uint32_t hl = sel.hl[by], level = bp.level[hl], sub = blockIdx.x % CUR_SUBS;
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
			if (LAYOUT == NGP_LAYOUT_SOA && sizeof(P) == 8 && (n & 1u) == 0u) {       // (r6) level-major fp32 gradients: the thread's eight pairs are 64 contiguous, 16-byte aligned bytes - four loads instead of eight
				const float4 *g4 = reinterpret_cast<const float4 *>(dy + (size_t)level * n + first);
				float4 u[4];
#pragma unroll
				for (int r = 0; r < 4; ++r) u[r] = g4[r];
#pragma unroll
				for (int r = 0; r < 4; ++r) { gk[2 * r] = make_float2(u[r].x, u[r].y); gk[2 * r + 1] = make_float2(u[r].z, u[r].w); }
			} else {
#pragma unroll
				for (uint32_t k = 0; k < RUN_K; ++k) gk[k] = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)level * n + first + k] : dy[(size_t)(first + k) * 16 + level]);
			}
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

// (r6) ADAM: the table's Adam + EMA sweep rides here as it does in k_bin_accumulate2 below (see there): the eight entries a thread would store the gradient of get
// optim.hip's update instead - fp32 master, both moments and, when the table has one, the fp16 shadow the gathers read.
__device__ __forceinline__ void adam_ride_update(float &p, float &m, float &v, float g, const AdamRide &ar) {
	float e = p;
	if (ar.ema) adam_ema_update<true>(p, m, v, e, g, ar.c); else adam_ema_update<false>(p, m, v, e, g, ar.c);
}
template <typename G, typename RV, bool ADAM>
__global__ __launch_bounds__(1024) void k_bin_accumulate(LevelTable lt, BinPlan bp, LevelSel sel, const uint32_t *__restrict__ absmax_bits, const uint32_t *__restrict__ cursors,
                                                         void *__restrict__ rec_val_base, uint16_t *__restrict__ rec_idx_base, const uint32_t *__restrict__ spill_count,
                                                         const SpillEntry *__restrict__ spill, G *__restrict__ grad, int overwrite, AdamRide ar) {
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
	float2 *P2 = nullptr, *M2 = nullptr, *V2 = nullptr; __half2 *H2 = nullptr;     // ADAM: the level's parameters, moments and fp16 shadow as pairs
	if (ADAM) {
		P2 = reinterpret_cast<float2 *>(ar.p) + lt.v[4 * level]; M2 = reinterpret_cast<float2 *>(ar.m) + lt.v[4 * level]; V2 = reinterpret_cast<float2 *>(ar.v) + lt.v[4 * level];
		if (ar.p_half) H2 = reinterpret_cast<__half2 *>(ar.p_half) + lt.v[4 * level];
	}
	auto sweep_store = [&](uint32_t t, float2 p, float2 m, float2 v, float gx, float gy) {
		adam_ride_update(p.x, m.x, v.x, gx, ar); adam_ride_update(p.y, m.y, v.y, gy, ar);
		P2[t] = p; M2[t] = m; V2[t] = v;
		if (H2) H2[t] = __floats2half2_rn(p.x, p.y);
	};
	if (s32 == 0.f || count == 0) {                                      // nothing to add: an accumulating destination is left alone, an overwritten one gets its zeros
		if (ADAM) {                                                         // ... and the sweep sees a zero gradient
			for (uint32_t e = threadIdx.x; e < n_local; e += 1024) { const uint32_t t = entry_of(bin, e, il); if (t < size) sweep_store(t, P2[t], M2[t], V2[t], 0.f, 0.f); }
		} else if (overwrite) {
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
	if (ADAM) {
		static_assert(BIN_ENTRIES == 8u * 1024u, "eight entries per thread");
		float2 rp[8], rm[8], rv[8];                                         // this thread's eight entries (the write-out's own assignment): all loads first
#pragma unroll
		for (uint32_t k = 0; k < 8; ++k) {
			const uint32_t e = threadIdx.x + k * 1024, t = e < n_local ? entry_of(bin, e, il) : ~0u;
			rp[k] = rm[k] = rv[k] = make_float2(0.f, 0.f);
			if (t < size) { rp[k] = P2[t]; rm[k] = M2[t]; rv[k] = V2[t]; }
		}
#pragma unroll
		for (uint32_t k = 0; k < 8; ++k) {
			const uint32_t e = threadIdx.x + k * 1024, t = e < n_local ? entry_of(bin, e, il) : ~0u;
			if (!(t < size)) continue;
			const long long sx = (long long)iacc[2 * e], sy = (long long)iacc[2 * e + 1];
			sweep_store(t, rp[k], rm[k], rv[k], (float)sx * inv, (float)sy * inv);
		}
		return;
	}
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
                                                 uint32_t *__restrict__ spill_count, SpillEntry *__restrict__ spill, const uint32_t *__restrict__ n_valid, TailJobs tj) {
	using P = typename Pair<T>::type;
	extern __shared__ __attribute__((aligned(16))) uint32_t bin_smem[];
	const uint32_t po = blockIdx.y;
	if (tj.do_sweep) tail_pack_share(tj, po * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);      // (r6, mlp_tail.h) this workgroup's 32 slots of the MLP fragment buffer, from the pack the run kernel's launch has swept
	PairRec *stage = reinterpret_cast<PairRec *>(bin_smem);                                   // [PAIR_STAGE_RECORDS] records, grouped by bin
	uint32_t *cnt = bin_smem + PAIR_STAGE_RECORDS * 4u, *loff = cnt + PAIR_BINS;             // loff[PAIR_BINS] = total
	const uint32_t hl = sel.hl[po], level = bp.level[hl];
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
#ifdef NGP_PROBE_SCATTER                                                                      // timing experiment (tools/probe_scatter.py builds it): no record stores - results WRONG
	if (total != 0x7fffffffu) return;
#endif
	for (uint32_t p = threadIdx.x; p < total; p += S) out[p] = stage[p];
}

// ---- run records without atomics (r4): the coarse levels of the fp32 path.  k_bin_records_runs above with the edge kernel's layout: a region per record workgroup
// (its run records sorted by bin, 12 bytes each {x, y, slot}; the bin offsets beside them), 128 bins of 4096 entries like the edge levels - so that ONE accumulate
// kernel serves every level.  Entries of a level smaller than 2^19 are dealt to the bins in interleaved groups of eight (the spatially coherent dense indices
// spread evenly), as before with 64 bins.
struct RunRec { float x, y; uint32_t loc; };
#define RUN2_OFFS (PAIR_BINS + 2u)
__host__ __device__ static inline uint32_t run2_region_records() { return RUN_WG * RUN_K * 8u + 4u; }
#define RUN2_STAGE 2048u                                                 // staged run records per workgroup of the region path (x 12 B + tables = 26 KiB: five workgroups per CU with room to spare; round 4: 54 -> 43 us)
static uint32_t run2_stage_bytes(uint32_t stage) { return stage * 12u + (3u * PAIR_BINS + 4u) * 4u; }
__host__ __device__ __forceinline__ uint32_t bin2_of(uint32_t e, bool il) { return il ? (e >> 3) & (PAIR_BINS - 1u) : e >> PAIR_BIN_BITS; }
__host__ __device__ __forceinline__ uint32_t local2_of(uint32_t e, bool il) { return il ? ((e >> 10) << 3) | (e & 7u) : e & (PAIR_BIN_ENTRIES - 1u); }
__host__ __device__ __forceinline__ uint32_t entry2_of(uint32_t bin, uint32_t local, bool il) { return il ? ((local >> 3) << 10) | (bin << 3) | (local & 7u) : (bin << PAIR_BIN_BITS) | local; }

template <typename T, int LAYOUT, int OCC>
__global__ __launch_bounds__(RUN_WG, OCC) void k_bin_runs2(uint32_t n, const float *__restrict__ pos, uint32_t stride, const T *__restrict__ dLdy, LevelTable lt, BinPlan bp, LevelSel sel,
                                                         const uint32_t *__restrict__ absmax_bits, RunRec *__restrict__ rrec, uint16_t *__restrict__ roff,
                                                         const uint32_t *__restrict__ n_valid, uint32_t stage /* records of LDS staging */, TailJobs tj) {
	extern __shared__ __attribute__((aligned(16))) uint32_t bin_smem[];
	const uint32_t ro = blockIdx.y;
	if (tj.do_reduce) tail_reduce_share(tj, reinterpret_cast<float *>(bin_smem), ro * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);      // (r6, mlp_tail.h) this workgroup's 8 columns of the MLP weight-gradient slabs (+ their parameters' sweep)
	RunRec *stage_rec = reinterpret_cast<RunRec *>(bin_smem);                              // [stage]
	uint32_t *cnt = bin_smem + stage * 3u, *loff = cnt + PAIR_BINS, *cnt2 = loff + PAIR_BINS + 2u;   // loff[PAIR_BINS] = total
	const uint32_t hl = sel.hl[ro], level = bp.level[hl];
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
			if (LAYOUT == NGP_LAYOUT_SOA && sizeof(P) == 8 && (n & 1u) == 0u) {       // (r6) level-major fp32 gradients: the thread's eight pairs are 64 contiguous, 16-byte aligned bytes - four loads instead of eight
				const float4 *g4 = reinterpret_cast<const float4 *>(dy + (size_t)level * n + first);
				float4 u[4];
#pragma unroll
				for (int r = 0; r < 4; ++r) u[r] = g4[r];
#pragma unroll
				for (int r = 0; r < 4; ++r) { gk[2 * r] = make_float2(u[r].x, u[r].y); gk[2 * r + 1] = make_float2(u[r].z, u[r].w); }
			} else {
#pragma unroll
				for (uint32_t k = 0; k < RUN_K; ++k) gk[k] = to_f2(LAYOUT == NGP_LAYOUT_SOA ? dy[(size_t)level * n + first + k] : dy[(size_t)(first + k) * 16 + level]);
			}
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
	sweep([&](uint32_t e, float, float) { atomicAdd(&cnt[bin2_of(e, il)], 1u); });
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
		const uint32_t bin = bin2_of(e, il), p = loff[bin] + atomicAdd(&cnt2[bin], 1u);
		const RunRec r{x, y, local2_of(e, il)};
		if (p < stage) stage_rec[p] = r; else out[p] = r;                      // beyond the staging area (scattered positions only): straight to its place in the region
	});
	__syncthreads();
	const uint32_t total = min(loff[PAIR_BINS], stage);
#ifdef NGP_PROBE_SCATTER                                                                  // timing experiment: no record stores from the staging area - results WRONG
	if (total != 0x7fffffffu) return;
#endif
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
template <typename Rec> __device__ __forceinline__ Rec nt_ld_rec(const Rec *p) {
	uint32_t w[sizeof(Rec) / 4];
#pragma unroll
	for (uint32_t k = 0; k < sizeof(Rec) / 4; ++k) w[k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p) + k);
	Rec r; __builtin_memcpy(&r, w, sizeof(Rec)); return r;
}
template <uint32_t B /* records in flight per thread */, typename Rec, typename F>
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
		for (uint32_t f0 = threadIdx.x; f0 < T; f0 += ACC2_WG * B) {
			Rec x[B];
#pragma unroll
			for (uint32_t b = 0; b < B; ++b) {
				const uint32_t f = f0 + b * ACC2_WG;
				if (f < T) {
					uint2 e = map[f >> shift];
					uint32_t w = e.x >> 23; e.x &= 0x7fffffu;
					while (f >= e.x) e = seg[++w];                          // (rarely: f lies up to 2^shift - 1 records behind the mapped one; empty segments end where they begin: skipped)
#ifndef ACC2_NO_NT
					if (!(probe & 2u)) x[b] = nt_ld_rec(recs + e.y + f);
#else
					if (!(probe & 2u)) x[b] = recs[e.y + f];
#endif
				}
			}
#pragma unroll
			for (uint32_t b = 0; b < B; ++b) if (f0 + b * ACC2_WG < T) { if (probe & 1u) sink ^= reinterpret_cast<const uint32_t *>(&x[b])[0]; else process(x[b]); }
		}
		__syncthreads();                                                    // the tables are rebuilt for the next block of regions
	}
	if (sink == 0x9e3779b9u) lds[0] = sink;                                 // (keeps the probe's loads alive)
}

// (r6) The table's Adam + EMA sweep can RIDE in this kernel (ADAM = true, fp32 table on one GPU with backward and sweep in one ngp_train_step call): a unit owns its
// 4096 entries exclusively and holds their finished gradient sums in LDS, so the thread that would store a gradient pair applies optim.hip's adam_ema_update to the
// parameter instead - the same arithmetic on the same float, hence the same bits as the separate k_adam_ema launch (test_fused_launches_of_the_native_step_change_no_bit).
// The gradient itself is never written (48.8 MB out + 48.8 MB back in per iteration saved, one launch less), and the sweep's HBM streams run beside the other resident
// workgroups' LDS-atomic phases: 103 us for both jobs against 70 + 49 us as two launches (+3.7 % it/s, profiles/r06l_ab_variants.txt).  Requesting the unit's p, m, v
// BEFORE the record gather (48 registers per thread held across it, six to eight records in flight) was measured too and is slower than loading them here: 110 - 114 us.
// the moments and the records are touched once per iteration: as non-temporal accesses they leave the caches to the table the next forward gathers from
// (profiles/r06m_ab_variants.txt: every kernel of the step 0-2 us shorter, +0.2 % it/s - inside the noise of a pair, consistent over both; -DACC2_NO_NT builds the plain accesses)
typedef float f4v_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_ld4(const float4 *p) { const f4v_nt v = __builtin_nontemporal_load(reinterpret_cast<const f4v_nt *>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void nt_st4(float4 *p, float4 v) { const f4v_nt t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<f4v_nt *>(p)); }
#ifndef ACC2_NO_NT
#define MV_LD(ptr) nt_ld4(ptr)
#define MV_ST(ptr, val) nt_st4((ptr), (val))
#else
#define MV_LD(ptr) (*(ptr))
#define MV_ST(ptr, val) (*(ptr) = (val))
#endif
template <typename G, bool ADAM>
__global__ __launch_bounds__(ACC2_WG, 4) void k_bin_accumulate2(LevelTable lt, BinPlan bp, LevelSel sel_pair, LevelSel sel_run, Acc2Plan ap, const uint32_t *__restrict__ absmax_bits,
                                                               const PairRec *__restrict__ prec, const uint16_t *__restrict__ poff, const RunRec *__restrict__ rrec, const uint16_t *__restrict__ roff,
                                                               const uint32_t *__restrict__ spill_count, const SpillEntry *__restrict__ spill, G *__restrict__ grad, int overwrite, AdamRide ar) {
	extern __shared__ __attribute__((aligned(16))) unsigned long long iacc[];   // [PAIR_BIN_ENTRIES][2] 64-bit fixed point
	using GP = typename Pair<G>::type;
	uint32_t *tables = reinterpret_cast<uint32_t *>(iacc + 2u * PAIR_BIN_ENTRIES);        // gather_flat's segment tables, behind the accumulators
	// one unit = one (level, bin) per workgroup, two workgroups per CU.  (A persistent variant - resident workgroups drawing units from a queue - was measured in round 4
	// and dropped: profiles/r04_scatter_probes.md.)
	const uint32_t n_units = (ap.n_pair + ap.n_run) * PAIR_BINS;
	const uint32_t u_this = blockIdx.x;
	if (u_this >= n_units) return;
	{
		const bool pr0 = u_this < ap.n_pair * PAIR_BINS;
		const uint32_t v0 = pr0 ? u_this : u_this - ap.n_pair * PAIR_BINS;
		const uint2 row0 = pr0 ? segment_row(poff, v0 / PAIR_BINS, ap.pair_regions, ap.pair_region_records, PAIR_OFFS, v0 % PAIR_BINS, 0u)
		                       : segment_row(roff, v0 / PAIR_BINS, ap.run_regions, ap.run_region_records, RUN2_OFFS, v0 % PAIR_BINS, 0u);
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
#ifndef ACC2_RECORDS_IN_FLIGHT
#define ACC2_RECORDS_IN_FLIGHT 8u
#endif
	constexpr uint32_t ACC2_B = ACC2_RECORDS_IN_FLIGHT;                    // records in flight per thread
	float2 *P2 = nullptr, *M2 = nullptr, *V2 = nullptr;                  // ADAM: the level's parameters and moments as pairs
	if (ADAM) { P2 = reinterpret_cast<float2 *>(ar.p) + lt.v[4 * level]; M2 = reinterpret_cast<float2 *>(ar.m) + lt.v[4 * level]; V2 = reinterpret_cast<float2 *>(ar.v) + lt.v[4 * level]; }
	if (s32 == 0.f) {                                                     // the level has no gradient: an accumulating destination is left alone, an overwritten one gets its zeros
		if (ADAM) {                                                         // ... and the sweep sees a zero gradient (the moments decay, the parameter follows them)
			for (uint32_t e = threadIdx.x; e < n_local; e += ACC2_WG) {
				const uint32_t t = entry2_of(bin, e, il);
				if (t >= size) continue;
				float2 p = P2[t], m = M2[t], v = V2[t];
				adam_ride_update(p.x, m.x, v.x, 0.f, ar); adam_ride_update(p.y, m.y, v.y, 0.f, ar);
				P2[t] = p; M2[t] = m; V2[t] = v;
			}
		} else if (overwrite) {
			GP zv; from_f2(zv, make_float2(0.f, 0.f));
			for (uint32_t e = threadIdx.x; e < n_local; e += ACC2_WG) { const uint32_t t = entry2_of(bin, e, il); if (t < size) dst[t] = zv; }
		}
		return;                                                             // (uniform)
	}
	for (uint32_t e = threadIdx.x; e < n_local; e += ACC2_WG) { iacc[2 * e] = 0ull; iacc[2 * e + 1] = 0ull; }
	__syncthreads();
#ifdef NGP_PROBE_ACC_U32_CARRY
	// diagnosis build (tools/probe_acc_carry.sh): the 64-bit sum as two 32-bit LDS atomics with carry instead of ds_add_u64 - the low word's returning add
	// tells the one thread whose add wrapped it, which adds the carry to the high word; adds commute, so the final sum is the same integer.  Same results, slower; never the product.
	auto add64 = [&](unsigned long long *p, unsigned long long v) {
		uint32_t *w = reinterpret_cast<uint32_t *>(p);
		const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
		const uint32_t old = __hip_atomic_fetch_add(w, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		const uint32_t up = hi + (uint32_t)((uint32_t)(old + lo) < old);
		if (up) __hip_atomic_fetch_add(w + 1, up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	};
#else
	auto add64 = [&](unsigned long long *p, unsigned long long v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
#endif
	auto add_fixed = [&](uint32_t local, long long ix, long long iy) {
		if ((ix | iy) == 0) return;
		add64(&iacc[2 * local], (unsigned long long)ix);
		add64(&iacc[2 * local + 1], (unsigned long long)iy);
	};
	if (is_pair) {
		gather_flat<ACC2_B>(prec, poff, ord, ap.pair_regions, ap.pair_region_records, PAIR_OFFS, bin, tables, ap.probe, row0, [&](const PairRec &r) {
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
		gather_flat<ACC2_B>(rrec, roff, ord, ap.run_regions, ap.run_region_records, RUN2_OFFS, bin, tables, ap.probe, row0, [&](const RunRec &r) {
			add_fixed(r.loc, fixed_rn(r.x, s32), fixed_rn(r.y, s32));
		});
	}
#ifdef ACC2_PMV_BEFORE_BARRIER
	// (A/B build) a full bin's p, m, v requested by each thread as soon as IT has processed its records - the loads then travel while the workgroup waits at the barrier for its
	// slowest wavefront - instead of after the barrier
	float4 rp[4], rm[4], rv[4];
	if (ADAM && !il) {
		const float4 *Pq = reinterpret_cast<const float4 *>(P2 + (bin << PAIR_BIN_BITS)), *Mq = reinterpret_cast<const float4 *>(M2 + (bin << PAIR_BIN_BITS)), *Vq = reinterpret_cast<const float4 *>(V2 + (bin << PAIR_BIN_BITS));
#pragma unroll
		for (uint32_t k = 0; k < 4u; ++k) { const uint32_t q = threadIdx.x + k * ACC2_WG; rp[k] = Pq[q]; rm[k] = MV_LD(Mq + q); rv[k] = MV_LD(Vq + q); }
	}
#endif
	__syncthreads();
	if (ADAM) {                                                           // the sweep instead of the gradient store: same entries per thread as below; all of a thread's loads first
		static_assert(PAIR_BIN_ENTRIES == 8u * ACC2_WG, "eight entries per thread");
		if (!il) {
			float4 *Pq = reinterpret_cast<float4 *>(P2 + (bin << PAIR_BIN_BITS)), *Mq = reinterpret_cast<float4 *>(M2 + (bin << PAIR_BIN_BITS)), *Vq = reinterpret_cast<float4 *>(V2 + (bin << PAIR_BIN_BITS));
#ifndef ACC2_PMV_BEFORE_BARRIER
			float4 rp[4], rm[4], rv[4];
#pragma unroll
			for (uint32_t k = 0; k < 4u; ++k) { const uint32_t q = threadIdx.x + k * ACC2_WG; rp[k] = Pq[q]; rm[k] = MV_LD(Mq + q); rv[k] = MV_LD(Vq + q); }
#endif
#pragma unroll
			for (uint32_t k = 0; k < 4u; ++k) {
				const uint32_t q = threadIdx.x + k * ACC2_WG, e = 2u * q;
				const long long s0 = (long long)iacc[2 * e], s1 = (long long)iacc[2 * e + 1], s2 = (long long)iacc[2 * e + 2], s3 = (long long)iacc[2 * e + 3];
				float4 p = rp[k], m = rm[k], v = rv[k];
				adam_ride_update(p.x, m.x, v.x, (float)s0 * inv, ar); adam_ride_update(p.y, m.y, v.y, (float)s1 * inv, ar);
				adam_ride_update(p.z, m.z, v.z, (float)s2 * inv, ar); adam_ride_update(p.w, m.w, v.w, (float)s3 * inv, ar);
				Pq[q] = p; MV_ST(Mq + q, m); MV_ST(Vq + q, v);
			}
		} else {
			float2 rp[8], rm[8], rv[8];
#pragma unroll
			for (uint32_t k = 0; k < 8u; ++k) {
				const uint32_t e = threadIdx.x + k * ACC2_WG, t = entry2_of(bin, e, il);
				rp[k] = rm[k] = rv[k] = make_float2(0.f, 0.f);
				if (e < n_local && t < size) { rp[k] = P2[t]; rm[k] = M2[t]; rv[k] = V2[t]; }
			}
#pragma unroll
			for (uint32_t k = 0; k < 8u; ++k) {
				const uint32_t e = threadIdx.x + k * ACC2_WG, t = entry2_of(bin, e, il);
				if (!(e < n_local && t < size)) continue;
				const long long sx = (long long)iacc[2 * e], sy = (long long)iacc[2 * e + 1];
				float2 p = rp[k], m = rm[k], v = rv[k];
				adam_ride_update(p.x, m.x, v.x, (float)sx * inv, ar); adam_ride_update(p.y, m.y, v.y, (float)sy * inv, ar);
				P2[t] = p; M2[t] = m; V2[t] = v;
			}
		}
	} else if (!il) {                                                     // a full bin: contiguous, two entries (16 bytes of fp32 gradient) per thread and trip
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
	}
}
// the accumulate with the table's sweep riding, under its own name for NGP_LAUNCH's brackets (bench.py tells the two apart; rocprof shows the template arguments)
#define k_bin_accumulate2_adam (k_bin_accumulate2<float, true>)
#define k_bin_accumulate_adam_f32rec (k_bin_accumulate<float, float2, true>)
#define k_bin_accumulate_adam_f16rec (k_bin_accumulate<float, __half2, true>)

// Which levels can take the binned path: up to 2^19 entries, and indexed the way the record kernels index (dense, or the XOR hash masked by a power of two).
// (aabb_scale 23.4 has a DENSE level with res 80 = 512000 entries - round 1 binned it with the XOR hash by looking at the size alone.)
static bool level_dense_host(uint32_t size, uint32_t res) { uint32_t stride = 1; for (int d = 0; d < 3; ++d) if (stride <= size) stride *= res; return !(size < stride); }
static bool level_binned(const LevelTable &lt, int l) {
	const uint32_t size = lt.v[4 * l + 1], res = lt.v[4 * l + 2];
	return size <= BIN_LEVEL_MAX && (level_dense_host(size, res) || (size & (size - 1)) == 0);
}
// capacity of one record list of the per-corner path: 4x the expected n*8/64 records per bin; % 8: 16-byte aligned streams
static uint32_t bin_capacity(uint32_t n) { uint32_t c = ((n / 2) / CUR_SUBS + 7u) & ~7u; const uint32_t lo = 4096u / CUR_SUBS; return c < lo ? lo : c; }
// the levels whose cell edges never leave a 4096-entry bin (k_bin_pairs): full 2^19-entry hashed tables (beyond res 2048: as eight single records per sample)
static bool level_pair_capable(const LevelTable &lt, int l) {
	const uint32_t size = lt.v[4 * l + 1], res = lt.v[4 * l + 2];
	return size == BIN_LEVEL_MAX && !level_dense_host(size, res);
}
static bool level_run(const LevelTable &lt, int l) { return lt.v[4 * l + 2] <= RUN_RES_MAX; }
// How a call with a workspace is routed - decided by the level table and the dtypes alone, so that the workspace can be SIZED for the path that will run (r5; rounds 2-4
// reserved the sum of both designs, ~2.3 GB at 2^18 samples):
//   HB_REGIONS    fp32 dL/dy -> fp32 gradient, every level a run level or edge-capable (every table GridEncode builds): record regions, no global atomic (r4)
//   HB_PERCORNER  everything else the bins can take (fp16 dL/dy - ngp_fox.py): per-corner record lists with cursor reservations (r2 / r3)
//   HB_ATOMICS    no workspace, a level beyond 2^19 entries, a hashed table that is not a power of two, NGP_HASH_BWD_ATOMICS=1: the reference's scheme, one global float
//                 atomic per corner (HashEncode.h:299-396) - correct, ~25x slower (7.8 ms per 2^18-sample batch), the ONE fallback of this stage
enum HashBwdPath { HB_ATOMICS = 0, HB_PERCORNER = 1, HB_REGIONS = 2 };
static HashBwdPath hash_bwd_path(const LevelTable &lt, int dtype, int grad_dtype, const void *grad) {
	if (hash_bwd_method() == 1) return HB_ATOMICS;
	bool regions = dtype == NGP_F32 && grad_dtype == NGP_F32 && ((uintptr_t)grad & 15u) == 0;
	for (int l = 0; l < 16; ++l) {
		if (!level_binned(lt, l)) return HB_ATOMICS;
		if (!level_run(lt, l) && !level_pair_capable(lt, l)) regions = false;
	}
	return regions ? HB_REGIONS : HB_PERCORNER;
}
// test hook (tests/test_host_cpu.py): the routing decision for a 16-byte aligned gradient buffer - 0 = global float atomics, 1 = per-corner lists, 2 = record regions
NGP_API int ngp_x_hash_bwd_path(const uint32_t *level_table_host, int dtype, int grad_dtype) {
	static float aligned_dummy[4] __attribute__((aligned(16)));
	return (int)hash_bwd_path(load_table(level_table_host), dtype, grad_dtype, aligned_dummy);
}
// workspace = cursors u32[N_ZEROED] | abs-max partials u32[16 * NGP_ABSMAX_PARTS], spill count | { per-corner: record values | record indices | spill list }
//                                                                                                 { regions: spill list | edge records | their bin offsets | run records | offsets }
struct WsLayout { uint64_t cursors, absmax, rec_val, rec_idx, spill, pair_rec, pair_off, run_rec, run_off, total; uint32_t cap, spill_cap, n_binned, n_pair, n_run, pair_s; };
static WsLayout ws_layout(const LevelTable &lt, uint32_t n, HashBwdPath path) {
	WsLayout w; memset(&w, 0, sizeof(w));
	bool any_split = false;                                              // an edge level beyond res 2048: 512-sample record workgroups (eight single records per sample)
	for (int l = 0; l < 16; ++l) {
		if (!level_binned(lt, l)) continue;
		++w.n_binned;
		if (level_run(lt, l)) ++w.n_run;
		else if (level_pair_capable(lt, l)) { ++w.n_pair; any_split |= lt.v[4 * l + 2] > PAIR_RES_MAX; }
	}
	w.pair_s = any_split ? 512u : 1024u;
	w.cap = bin_capacity(n);
	w.cursors = 0;
	w.absmax = w.cursors + N_ZEROED * 4u;
	const uint64_t body = w.absmax + 16u * ABSMAX_PARTS * 4u + 256;
	if (path == HB_REGIONS) {
		// spill list of the edge kernel: edges of positions OUTSIDE the unit cube (x + 1 can carry into the bin bits) - none in a marched batch; sized for the worst case,
		// every edge of every sample on every edge level (2 entries each), because a dropped entry would be a silently wrong gradient
		w.spill_cap = (uint32_t)(((uint64_t)w.n_pair * 8u * n < (1ull << 27)) ? (uint64_t)w.n_pair * 8u * n : (1ull << 27));
		w.spill = body;
		w.pair_rec = (w.spill + (uint64_t)w.spill_cap * sizeof(SpillEntry) + 255) & ~(uint64_t)255;
		const uint64_t regions_pair = div_up(n, w.pair_s);                  // one region per record workgroup
		w.pair_off = (w.pair_rec + (uint64_t)w.n_pair * regions_pair * pair_region_records() * sizeof(PairRec) + 255) & ~(uint64_t)255;
		w.run_rec = w.pair_off + (((uint64_t)w.n_pair * regions_pair * PAIR_OFFS * sizeof(uint16_t) + 255) & ~(uint64_t)255);
		const uint64_t regions_run = div_up(n, RUN_WG * RUN_K);             // a region per 2048-sample workgroup, worst case eight records per sample (scattered positions)
		w.run_off = (w.run_rec + (uint64_t)w.n_run * regions_run * run2_region_records() * sizeof(RunRec) + 255) & ~(uint64_t)255;
		w.total = w.run_off + (((uint64_t)w.n_run * regions_run * RUN2_OFFS * sizeof(uint16_t) + 255) & ~(uint64_t)255);
	} else if (path == HB_PERCORNER) {
		const uint32_t per = (1u << 25) / (w.n_binned ? w.n_binned : 1u);
		w.spill_cap = w.n_binned * 8u * (n < per ? n : per);              // worst case: every record of every binned level overflows (12 B each)
		w.rec_val = body;
		w.rec_idx = w.rec_val + (uint64_t)w.n_binned * BINS_PER_LEVEL * CUR_SUBS * w.cap * sizeof(float2);       // every level owns 64 * cap * 8 bytes (rec_val_at)
		w.spill = (w.rec_idx + (uint64_t)w.n_binned * BINS_PER_LEVEL * CUR_SUBS * w.cap * sizeof(uint16_t) + 255) & ~(uint64_t)255;
		w.total = (w.spill + (uint64_t)w.spill_cap * sizeof(SpillEntry) + 255) & ~(uint64_t)255;
	} else w.total = 0;
	return w;
}
// bytes ngp_hash_encode_bwd_ws needs for n samples: for the path the given dtypes take (ngp_base.py / fp32: ~0.6 GB at 2^18 samples, ngp_fox.py / fp16: ~1.7 GB) ...
NGP_API uint64_t ngp_hash_bwd_workspace_bytes_for(const uint32_t *level_table_host, uint32_t n, int dtype, int grad_dtype) {
	const LevelTable lt = load_table(level_table_host);
	static float aligned_dummy[4] __attribute__((aligned(16)));
	return ws_layout(lt, n, hash_bwd_path(lt, dtype, grad_dtype, aligned_dummy)).total;
}
// ... and for a caller that does not say: enough for either
NGP_API uint64_t ngp_hash_bwd_workspace_bytes(const uint32_t *level_table_host, uint32_t n) {
	const uint64_t a = ngp_hash_bwd_workspace_bytes_for(level_table_host, n, NGP_F32, NGP_F32), b = ngp_hash_bwd_workspace_bytes_for(level_table_host, n, NGP_F16, NGP_F32);
	return a > b ? a : b;
}

static int set_dyn_lds(const void *k, size_t bytes) {
	hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
	if (e != hipSuccess) { ngp_set_error("ngp_hash_encode_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
	return 0;
}
// the dynamic-LDS limits of the scatter kernels, raised once per DEVICE and only marked done when every call succeeded
static int hash_bwd_set_lds() {
	static std::mutex mu;
	static bool done[64] = {false};
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
	std::lock_guard<std::mutex> lk(mu);
	if (done[dev]) return 0;
	int rc = 0;
#define SET(K, BYTES) do { if (!rc) rc = set_dyn_lds((const void *)K, (BYTES)); } while (0)
#define SET_T(T) \
	SET((k_bin_records_runs<T, NGP_LAYOUT_SOA, 4>), run_stage_bytes(RUN_STAGE)); SET((k_bin_records_runs<T, NGP_LAYOUT_AOS, 4>), run_stage_bytes(RUN_STAGE)); \
	SET((k_bin_records<T, NGP_LAYOUT_SOA>), bin_stage_bytes<T>()); SET((k_bin_records<T, NGP_LAYOUT_AOS>), bin_stage_bytes<T>());
	SET_T(float) SET_T(__half)
	SET((k_bin_pairs<float, NGP_LAYOUT_SOA, 512u>), pair_stage_bytes()); SET((k_bin_pairs<float, NGP_LAYOUT_AOS, 512u>), pair_stage_bytes());
	SET((k_bin_pairs<float, NGP_LAYOUT_SOA, 1024u>), pair_stage_bytes()); SET((k_bin_pairs<float, NGP_LAYOUT_AOS, 1024u>), pair_stage_bytes());
	SET((k_bin_runs2<float, NGP_LAYOUT_SOA, 5>), run2_stage_bytes(RUN2_STAGE)); SET((k_bin_runs2<float, NGP_LAYOUT_AOS, 5>), run2_stage_bytes(RUN2_STAGE));
#undef SET_T
	SET((k_bin_accumulate2<float, false>), PAIR_BIN_ENTRIES * 16u + ACC2_LDS_EXTRA);
	SET((k_bin_accumulate2<float, true>), PAIR_BIN_ENTRIES * 16u + ACC2_LDS_EXTRA);
	SET((k_bin_accumulate<float, float2, false>), BIN_ENTRIES * 16); SET((k_bin_accumulate<float, __half2, false>), BIN_ENTRIES * 16); SET((k_bin_accumulate<__half, float2, false>), BIN_ENTRIES * 16); SET((k_bin_accumulate<__half, __half2, false>), BIN_ENTRIES * 16);
	SET((k_bin_accumulate<float, float2, true>), BIN_ENTRIES * 16); SET((k_bin_accumulate<float, __half2, true>), BIN_ENTRIES * 16);
#undef SET
	if (!rc) done[dev] = true;
	return rc;
}

static int hash_bwd_impl(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *dLdy, const uint32_t *level_table_host,
                         void *grad, uint64_t n_params, int dtype, int grad_dtype, int in_layout, int zero_first, const uint32_t *n_valid, void *workspace, uint64_t workspace_bytes,
                         hipEvent_t after_coarse = nullptr /* data parallel, overlapped exchange: recorded behind the accumulate launch of the run-combined (coarse) levels, which then is a launch of its own */,
                         bool absmax_done = false /* the abs-max partials, zeroed cursors and spill count are already in the workspace (written by the field backward kernel, ngp_hash_bwd_absmax_slots) */,
                         const TailJobs *tail = nullptr /* (r6) jobs that may ride in the record launches (mlp_tail.h) */, int *tail_taken = nullptr /* set to 1 when they did */,
                         const AdamRide *adam = nullptr /* (r6) the table's Adam + EMA sweep, applied by the accumulate kernel INSTEAD of storing the gradient */, int *adam_taken = nullptr /* set to 1 when it was */) {
	if (tail_taken) *tail_taken = 0;
	if (adam_taken) *adam_taken = 0;
	NGP_REQUIRE(grad && level_table_host && (n == 0 || (pos && dLdy)), NGP_E_ARG, "ngp_hash_encode_bwd: null pointer");
	NGP_REQUIRE(dtype == NGP_F32 || dtype == NGP_F16, NGP_E_DTYPE, "ngp_hash_encode_bwd: bad dtype %d", dtype);
	NGP_REQUIRE(grad_dtype == NGP_F32 || (grad_dtype == NGP_F16 && dtype == NGP_F16), NGP_E_DTYPE, "ngp_hash_encode_bwd: bad grad dtype %d for dtype %d", grad_dtype, dtype);
	hipStream_t s = (hipStream_t)stream;
	const size_t gsz = grad_dtype == NGP_F16 ? 2 : 4;
	const LevelTable lt = load_table(level_table_host);
	const HashBwdPath path = workspace ? hash_bwd_path(lt, dtype, grad_dtype, grad) : HB_ATOMICS;
	const WsLayout wl = ws_layout(lt, n, path);
	// (r6) the routing looks at the gradient pointer's alignment, the sizing function cannot: an fp32 call with an unaligned gradient used to land on the per-corner path,
	// which needs three times the bytes ngp_hash_bwd_workspace_bytes_for returned - and the capacity error below then named the wrong size.  The workspace path takes aligned
	// gradients only (every buffer torch or hipMalloc hands out is; a view into a flat parameter buffer at an odd offset is the caller's to avoid).
	NGP_REQUIRE(!workspace || ((uintptr_t)grad & 15u) == 0, NGP_E_ALIGN, "ngp_hash_encode_bwd: with a workspace the gradient buffer must be 16-byte aligned");
	// (r6) the spill lists are sized for the worst case up to a cap; beyond it entries would be DROPPED - a silently wrong gradient.  Refuse such a batch up front.
	if (path == HB_REGIONS) NGP_REQUIRE((uint64_t)wl.n_pair * 8u * n <= (1ull << 27), NGP_E_CAPACITY, "ngp_hash_encode_bwd: %u samples exceed the workspace path's spill capacity (%llu per call for this level table): split the batch", n, (unsigned long long)((1ull << 27) / (8u * (wl.n_pair ? wl.n_pair : 1u))));
	if (path == HB_PERCORNER) NGP_REQUIRE(n <= (1u << 25) / (wl.n_binned ? wl.n_binned : 1u), NGP_E_CAPACITY, "ngp_hash_encode_bwd: %u samples exceed the workspace path's spill capacity (%u per call for this level table): split the batch", n, (1u << 25) / (wl.n_binned ? wl.n_binned : 1u));
	// (r5) a workspace that is too small for the path its dtypes take is an ERROR - rounds 2-4 dropped silently to a slower path
	NGP_REQUIRE(path == HB_ATOMICS || workspace_bytes >= wl.total, NGP_E_CAPACITY, "ngp_hash_encode_bwd: workspace of %llu bytes, ngp_hash_bwd_workspace_bytes_for(n = %u) = %llu",
	            (unsigned long long)workspace_bytes, n, (unsigned long long)wl.total);
	if (path == HB_ATOMICS) {
		if (zero_first) {
			hipError_t e = hipMemsetAsync(grad, 0, n_params * gsz, s);
			if (e != hipSuccess) { ngp_set_error("ngp_hash_encode_bwd memset: %s", hipGetErrorString(e)); return (int)e; }
		}
		if (n == 0) { if (after_coarse) (void)hipEventRecord(after_coarse, s); return 0; }
		const uint32_t nblk = div_up(n, 256);
		const dim3 grid(16 * nblk), block(256);
#define GO(T, G, L) NGP_LAUNCH((k_hash_bwd<T, G, L>), grid, block, 0, s, n, pos, pos_stride, (const T *)dLdy, lt, (G *)grad, nblk, n_valid)
		if (dtype == NGP_F32) { if (in_layout == NGP_LAYOUT_SOA) GO(float, float, NGP_LAYOUT_SOA); else GO(float, float, NGP_LAYOUT_AOS); }
		else if (grad_dtype == NGP_F32) { if (in_layout == NGP_LAYOUT_SOA) GO(__half, float, NGP_LAYOUT_SOA); else GO(__half, float, NGP_LAYOUT_AOS); }
		else { if (in_layout == NGP_LAYOUT_SOA) GO(__half, __half, NGP_LAYOUT_SOA); else GO(__half, __half, NGP_LAYOUT_AOS); }
#undef GO
		NGP_LAUNCH_CHECK("ngp_hash_encode_bwd");
		if (after_coarse) (void)hipEventRecord(after_coarse, s);
		return 0;
	}
	int rc = hash_bwd_set_lds(); if (rc) return rc;
	// ---- the binned scatter: every level through records - run levels (res <= 300) with run combining, the others as edge records (regions) or per-corner records
	BinPlan bp; bp.n_levels = 0; bp.cap = wl.cap; bp.spill_cap = wl.spill_cap;
	LevelSel sel_fine, sel_runs, sel_all, sel_pair;                      // sel_all: every level with per-corner records (runs + fine); sel_pair: the edge-record levels
	uint32_t n_fine = 0, n_runs = 0, n_pair = 0, n_all = 0;
	const bool regions = path == HB_REGIONS;
	for (int l = 0; l < 16; ++l) {                                       // (coarsest level first measured 1 % faster than finest first on both samplings)
		const uint32_t hl = bp.n_levels++;
		bp.level[hl] = (uint32_t)l;
		if (level_run(lt, l)) { sel_runs.hl[n_runs++] = hl; if (!regions) sel_all.hl[n_all++] = hl; }
		else if (regions) sel_pair.hl[n_pair++] = hl;
		else { sel_fine.hl[n_fine++] = hl; sel_all.hl[n_all++] = hl; }
	}
	char *ws = (char *)workspace;
	uint32_t *cursors = (uint32_t *)(ws + wl.cursors);
	uint32_t *absmax = (uint32_t *)(ws + wl.absmax);
	uint32_t *spill_count = absmax + 16u * ABSMAX_PARTS;
	SpillEntry *spill = (SpillEntry *)(ws + wl.spill);
	const int ow = zero_first ? 1 : 0;
	bool coarse_marked = false;
	// (r6) the timing probes of tools/probe_scatter.py - parts of a kernel skipped, results WRONG - are compile-time builds now (-DNGP_PROBE_SCATTER: no record stores;
	// -DNGP_PROBE_ACC=1|2: accumulate's records loaded but not processed | not loaded), no longer environment variables a user could set on the product binary
#ifdef NGP_PROBE_ACC
	const uint32_t acc_probe = NGP_PROBE_ACC;
#else
	const uint32_t acc_probe = 0u;
#endif
	// (r6) the table's sweep rides in the accumulate launches when together they OVERWRITE the whole table: every level is a unit level on the workspace paths, so that is
	// "the levels tile [0, n_params) without gaps" (every table GridEncode builds), an fp32 gradient (the sweep's input) and no data-parallel exchange between the two
	// (after_coarse); anything else leaves the sweep to the caller.  The fp32 region kernel has no shadow to write.
	bool ride = adam && adam->p && adam->m && adam->v && ow && !after_coarse && grad_dtype == NGP_F32 && (((uintptr_t)adam->p | (uintptr_t)adam->m | (uintptr_t)adam->v) & 15u) == 0 &&
	            (((uintptr_t)adam->p_half) & 7u) == 0 && !(regions && adam->p_half);
	if (ride) {
		uint64_t covered = 0;
		for (int l = 0; l < 16; ++l) { if ((uint64_t)lt.v[4 * l] * 2u != covered || (lt.v[4 * l] & 1u)) ride = false; covered += (uint64_t)lt.v[4 * l + 1] * 2u; }
		if (covered != n_params) ride = false;
	}
	const AdamRide no_ride{nullptr, nullptr, nullptr, nullptr, AdamConsts{}, 0};
#define ABSMAX(T, L) do { if (!absmax_done) NGP_LAUNCH((k_level_absmax<T, L>), dim3(ABSMAX_OWN_PARTS, 16), dim3(256), 0, s, n, (const T *)dLdy, absmax, n_valid, cursors, spill_count); } while (0)   /* also zeroes the cursors and the spill count */
	if (regions) {
		// fp32 -> fp32: run records + edge records in regions, no global atomics, ONE accumulate kernel - two launches of it when the data-parallel exchange wants the coarse levels first
		PairRec *pair_rec = (PairRec *)(ws + wl.pair_rec);
		uint16_t *pair_off = (uint16_t *)(ws + wl.pair_off);
		RunRec *run_rec = (RunRec *)(ws + wl.run_rec);
		uint16_t *run_off = (uint16_t *)(ws + wl.run_off);
		const uint32_t pair_s = wl.pair_s;
		// (r6) the MLP tail rides along when both record kernels run: every workgroup of k_bin_runs2 reduces its share of the slabs and sweeps those parameters, every workgroup
		// of k_bin_pairs - the next launch in the stream, so it reads the swept pack - gathers its share of the fragment buffer (mlp_tail.h)
		TailJobs tj_run = no_tail_jobs(), tj_pair = no_tail_jobs();
		if (tail && tail->do_reduce && tail->do_sweep && tail->pack_table && n_runs && n_pair && n > 0) {
			tj_run = *tail; tj_pair = *tail; tj_pair.do_reduce = 0;
			if (tail_taken) *tail_taken = 1;
		}
#define RGO(L) NGP_LAUNCH((k_bin_runs2<float, L, 5>), dim3(div_up(n, RUN_WG * RUN_K), n_runs), dim3(RUN_WG), run2_stage_bytes(RUN2_STAGE), s, n, pos, pos_stride, (const float *)dLdy, lt, bp, sel_runs, (const uint32_t *)absmax, run_rec, run_off, n_valid, RUN2_STAGE, tj_run)
// (k_bin_pairs depends on the abs-max pass like k_bin_runs2 does, not on k_bin_runs2: both are plain in-order launches - the any-order launch of round 4 lost its A/B and raced with the abs-max pass when there was no run level)
#define PGO(L, S) NGP_LAUNCH((k_bin_pairs<float, L, S>), dim3(div_up(n, S), n_pair), dim3(S), pair_stage_bytes(), s, n, pos, pos_stride, (const float *)dLdy, lt, bp, sel_pair, (const uint32_t *)absmax, pair_rec, pair_off, spill_count, spill, n_valid, tj_pair)
#define RECORDS(L) do { ABSMAX(float, L); if (n_runs) RGO(L); if (n_pair) { if (pair_s == 512u) PGO(L, 512u); else PGO(L, 1024u); } } while (0)
		if (in_layout == NGP_LAYOUT_SOA) RECORDS(NGP_LAYOUT_SOA); else RECORDS(NGP_LAYOUT_AOS);
#undef RECORDS
#undef PGO
#undef RGO
		auto accumulate = [&](bool runs, bool pairs) {
			Acc2Plan ap;
			ap.n_pair = pairs ? n_pair : 0u; ap.n_run = runs ? n_runs : 0u;
			ap.pair_regions = div_up(n, pair_s); ap.pair_region_records = pair_region_records();
			ap.run_regions = div_up(n, RUN_WG * RUN_K); ap.run_region_records = run2_region_records();
			ap.probe = acc_probe;
			const uint32_t n_units = (ap.n_pair + ap.n_run) * PAIR_BINS;
			if (!n_units) return;
			if (ride) {
				NGP_LAUNCH(k_bin_accumulate2_adam, dim3(n_units), dim3(ACC2_WG), PAIR_BIN_ENTRIES * 16u + ACC2_LDS_EXTRA, s, lt, bp, sel_pair, sel_runs, ap, (const uint32_t *)absmax, (const PairRec *)pair_rec,
				           (const uint16_t *)pair_off, (const RunRec *)run_rec, (const uint16_t *)run_off, (const uint32_t *)spill_count, (const SpillEntry *)spill, (float *)nullptr, 1, *adam);
				return;
			}
			NGP_LAUNCH((k_bin_accumulate2<float, false>), dim3(n_units), dim3(ACC2_WG), PAIR_BIN_ENTRIES * 16u + ACC2_LDS_EXTRA, s, lt, bp, sel_pair, sel_runs, ap, (const uint32_t *)absmax, (const PairRec *)pair_rec,
			           (const uint16_t *)pair_off, (const RunRec *)run_rec, (const uint16_t *)run_off, (const uint32_t *)spill_count, (const SpillEntry *)spill, (float *)grad, ow, no_ride);
		};
		if (after_coarse && n_runs && n_pair) { accumulate(true, false); (void)hipEventRecord(after_coarse, s); coarse_marked = true; accumulate(false, true); }
		else accumulate(true, true);
		if (ride && adam_taken) *adam_taken = 1;
	} else {
		// per-corner record lists (fp16 dL/dy): 6-byte fp16 records for the fine levels, fp32 run records for the coarse ones
		void *rec_val = (void *)(ws + wl.rec_val);
		uint16_t *rec_idx = (uint16_t *)(ws + wl.rec_idx);
		// (r6) fp16 configuration: the slab reduction that also sweeps the two MLP packs (k_reduce_slabs_sweep's job) rides as row 0 of the run-record launch
		TailJobs tj_pc = no_tail_jobs();
		if (tail && tail->do_reduce && tail->do_sweep16 && !tail->do_sweep && n_runs && n > 0) { tj_pc = *tail; if (tail_taken) *tail_taken = 1; }
#define GO(T, G, L) do { \
	using RV_ = typename RecVal<T>::type; \
	ABSMAX(T, L); \
	if (n_runs) NGP_LAUNCH((k_bin_records_runs<T, L, 4>), dim3(div_up(n, RUN_WG * RUN_K), n_runs), dim3(RUN_WG), run_stage_bytes(RUN_STAGE), s, n, pos, pos_stride, (const T *)dLdy, lt, bp, sel_runs, (const uint32_t *)absmax, cursors, rec_val, rec_idx, spill_count, spill, n_valid, RUN_STAGE, tj_pc); \
	if (n_fine) NGP_LAUNCH((k_bin_records<T, L>), dim3(div_up(n, BIN_WG), n_fine), dim3(BIN_WG), bin_stage_bytes<T>(), s, n, pos, pos_stride, (const T *)dLdy, lt, bp, sel_fine, (const uint32_t *)absmax, cursors, rec_val, rec_idx, spill_count, spill, n_valid); \
	if (ride && sizeof(G) == 4) {                                    /* (r6) the table's sweep rides: same launches, the update in place of the gradient store */ \
		if (sizeof(RV_) == 8) { \
			if (n_all) NGP_LAUNCH(k_bin_accumulate_adam_f32rec, dim3(n_all * BINS_PER_LEVEL), dim3(1024), BIN_ENTRIES * 16, s, lt, bp, sel_all, (const uint32_t *)absmax, (const uint32_t *)cursors, rec_val, rec_idx, (const uint32_t *)spill_count, (const SpillEntry *)spill, (float *)nullptr, 1, *adam); \
		} else { \
			if (n_runs) NGP_LAUNCH(k_bin_accumulate_adam_f32rec, dim3(n_runs * BINS_PER_LEVEL), dim3(1024), BIN_ENTRIES * 16, s, lt, bp, sel_runs, (const uint32_t *)absmax, (const uint32_t *)cursors, rec_val, rec_idx, (const uint32_t *)spill_count, (const SpillEntry *)spill, (float *)nullptr, 1, *adam); \
			if (n_fine) NGP_LAUNCH(k_bin_accumulate_adam_f16rec, dim3(n_fine * BINS_PER_LEVEL), dim3(1024), BIN_ENTRIES * 16, s, lt, bp, sel_fine, (const uint32_t *)absmax, (const uint32_t *)cursors, rec_val, rec_idx, (const uint32_t *)spill_count, (const SpillEntry *)spill, (float *)nullptr, 1, *adam); \
		} \
		if (adam_taken) *adam_taken = 1; \
	} else if (sizeof(RV_) == 8 && !(after_coarse && n_runs && n_fine)) {   /* fp32 records, one type: one accumulate launch over all their levels */ \
		if (n_all) NGP_LAUNCH((k_bin_accumulate<G, float2, false>), dim3(n_all * BINS_PER_LEVEL), dim3(1024), BIN_ENTRIES * 16, s, lt, bp, sel_all, (const uint32_t *)absmax, (const uint32_t *)cursors, rec_val, rec_idx, (const uint32_t *)spill_count, (const SpillEntry *)spill, (G *)grad, ow, no_ride); \
	} else { \
		if (n_runs) { NGP_LAUNCH((k_bin_accumulate<G, float2, false>), dim3(n_runs * BINS_PER_LEVEL), dim3(1024), BIN_ENTRIES * 16, s, lt, bp, sel_runs, (const uint32_t *)absmax, (const uint32_t *)cursors, rec_val, rec_idx, (const uint32_t *)spill_count, (const SpillEntry *)spill, (G *)grad, ow, no_ride); \
			if (after_coarse && n_fine) { (void)hipEventRecord(after_coarse, s); coarse_marked = true; } } \
		if (n_fine) NGP_LAUNCH((k_bin_accumulate<G, RV_, false>), dim3(n_fine * BINS_PER_LEVEL), dim3(1024), BIN_ENTRIES * 16, s, lt, bp, sel_fine, (const uint32_t *)absmax, (const uint32_t *)cursors, rec_val, rec_idx, (const uint32_t *)spill_count, (const SpillEntry *)spill, (G *)grad, ow, no_ride); \
	} } while (0)
		if (dtype == NGP_F32) { if (in_layout == NGP_LAYOUT_SOA) GO(float, float, NGP_LAYOUT_SOA); else GO(float, float, NGP_LAYOUT_AOS); }
		else if (grad_dtype == NGP_F32) { if (in_layout == NGP_LAYOUT_SOA) GO(__half, float, NGP_LAYOUT_SOA); else GO(__half, float, NGP_LAYOUT_AOS); }
		else { if (in_layout == NGP_LAYOUT_SOA) GO(__half, __half, NGP_LAYOUT_SOA); else GO(__half, __half, NGP_LAYOUT_AOS); }
#undef GO
	}
#undef ABSMAX
	NGP_LAUNCH_CHECK("ngp_hash_encode_bwd");
	if (after_coarse && !coarse_marked) (void)hipEventRecord(after_coarse, s);      // no separate coarse launch on this path: the marker follows the whole scatter
	return 0;
}

// the workspace path with the data-parallel marker and the fused abs-max (csrc/train_step.hip); not part of the public ABI
int ngp_hash_encode_bwd_ws_marked(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *dLdy, const uint32_t *level_table_host, void *grad, uint64_t n_params, int dtype,
                                  int grad_dtype, int in_layout, int zero_first, const uint32_t *n_valid, void *workspace, uint64_t workspace_bytes, hipEvent_t after_coarse, int absmax_done,
                                  const TailJobs *tail, int *tail_taken, const AdamRide *adam, int *adam_taken) {
	return hash_bwd_impl(stream, n, pos, pos_stride, dLdy, level_table_host, grad, n_params, dtype, grad_dtype, in_layout, zero_first, n_valid, workspace, workspace_bytes, after_coarse, absmax_done != 0,
	                     tail, tail_taken, adam, adam_taken);
}
// mirrors the routing of hash_bwd_impl: the slots are handed out only when that call will read them (grad: the gradient buffer the call will be given)
AbsmaxOut ngp_hash_bwd_absmax_slots(const uint32_t *level_table_host, uint32_t n, int dtype, int grad_dtype, void *workspace, uint64_t workspace_bytes, const void *grad) {
	AbsmaxOut am{nullptr, nullptr, 0u, nullptr};
	if (!workspace || !level_table_host || n == 0 || getenv("NGP_NO_FUSED_ABSMAX")) return am;
	const LevelTable lt = load_table(level_table_host);
	const HashBwdPath path = hash_bwd_path(lt, dtype, grad_dtype, grad);
	if (path == HB_ATOMICS) return am;
	const WsLayout wl = ws_layout(lt, n, path);
	if (workspace_bytes < wl.total) return am;                                                   // (the scatter call will report it)
	char *ws = (char *)workspace;
	am.parts = (uint32_t *)(ws + wl.absmax); am.cursors = (uint32_t *)(ws + wl.cursors); am.n_cursors = N_ZEROED; am.spill_count = am.parts + 16u * ABSMAX_PARTS;
	return am;
}

NGP_API int ngp_hash_encode_bwd(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *dLdy, const uint32_t *level_table_host,
                                void *grad, uint64_t n_params, int dtype, int grad_dtype, int in_layout, int zero_first, const uint32_t *n_valid) {
	return hash_bwd_impl(stream, n, pos, pos_stride, dLdy, level_table_host, grad, n_params, dtype, grad_dtype, in_layout, zero_first, n_valid, nullptr, 0);
}
NGP_API int ngp_hash_encode_bwd_ws(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *dLdy, const uint32_t *level_table_host,
                                   void *grad, uint64_t n_params, int dtype, int grad_dtype, int in_layout, int zero_first, const uint32_t *n_valid,
                                   void *workspace, uint64_t workspace_bytes) {
	return hash_bwd_impl(stream, n, pos, pos_stride, dLdy, level_table_host, grad, n_params, dtype, grad_dtype, in_layout, zero_first, n_valid, workspace, workspace_bytes);
}
