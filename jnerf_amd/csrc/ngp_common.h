// Shared device/host helpers for libngp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ngp_hip.h"

#define NGP_API extern "C" __attribute__((visibility("default")))

void ngp_set_error(const char *fmt, ...);
#define NGP_REQUIRE(cond, code, ...) do { if (!(cond)) { ngp_set_error(__VA_ARGS__); return (code); } } while (0)
#define NGP_LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { ngp_set_error("%s: %s", name, hipGetErrorString(e_)); return (int)e_; } } while (0)

static inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// Every kernel launch goes through NGP_LAUNCH: identical to hipLaunchKernelGGL unless the kernel was enabled with ngp_prof_enable, in which case the launch is
// bracketed by a HIP event pair on its own stream (csrc/prof.hip).
extern int g_ngp_prof_on;
int ngp_prof_register(const char *kernel_expr);
struct NgpProfScope { int id; hipStream_t s; hipEvent_t a, b; NgpProfScope(int id, hipStream_t s); ~NgpProfScope(); };
#define NGP_LAUNCH(kernel, grid, block, shmem, stream, ...) do { \
	static const int ngp_kid_ = ngp_prof_register(#kernel); \
	NgpProfScope ngp_ps_(ngp_kid_, (hipStream_t)(stream)); \
	hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__); } while (0)

// internal cross-file entry points (not exported)
struct TailJobs;
struct AdamRide;
// tail (may be null) / tail_taken: jobs the call may carry in its record launches; *tail_taken says whether it did (else the caller launches them itself)
// adam (may be null) / adam_taken: the table's sweep, applied by the accumulate kernel in place of the gradient store; *adam_taken says whether it was (the gradient buffer is NOT written then)
int ngp_hash_encode_bwd_ws_marked(void *stream, uint32_t n, const float *pos, uint32_t pos_stride, const void *dLdy, const uint32_t *level_table_host, void *grad, uint64_t n_params, int dtype,
                                  int grad_dtype, int in_layout, int zero_first, const uint32_t *n_valid, void *workspace, uint64_t workspace_bytes, hipEvent_t after_coarse, int absmax_done,
                                  const TailJobs *tail, int *tail_taken, const AdamRide *adam, int *adam_taken);
// Largest |dL/dfeature| per level, the scale of the hash scatter's fixed-point accumulation.  NGP_ABSMAX_PARTS partial maxima per level (bit patterns of
// non-negative floats, 0 = none); the consumers take the maximum.  Written either by the scatter's own abs-max pass or - training path, r3 - by the field backward
// kernel's epilogue (one partial per workgroup: no extra pass over the 33 MB of feature gradients, one launch less); that kernel then also zeroes the scatter's
// record cursors and spill count.
#define NGP_ABSMAX_PARTS 256u
struct AbsmaxOut { uint32_t *parts; uint32_t *cursors; uint32_t n_cursors; uint32_t *spill_count; };
// where the abs-max partials / cursors of a hash-backward workspace live, or parts == nullptr when that call would not take the binned path
AbsmaxOut ngp_hash_bwd_absmax_slots(const uint32_t *level_table_host, uint32_t n, int dtype, int grad_dtype, void *workspace, uint64_t workspace_bytes, const void *grad);
int ngp_reduce_slabs_sweep(void *stream, const float *slabs, uint32_t n_slabs, uint32_t width, float *out, const float *const pk[2][5], const uint32_t begin[2], const uint32_t count[2],
                           float lr, float beta0, float beta1, float eps, uint32_t step, float ema_decay);
int ngp_field32_bwd_am(void *stream, uint32_t n, const float *feat, int layout, const float *dir, uint32_t dir_stride, const float *wd, const float *wc,
                       const float *dLdout, float *dLdfeat, float *wgrad_slabs, uint32_t n_slabs, const uint32_t *n_valid, const AbsmaxOut *am);
int ngp_field_bwd_am(void *stream, uint32_t n, const void *feat, int layout, const float *dir, uint32_t dir_stride, const void *wd, const void *wc,
                     const void *dLdout, int out_dtype, void *dLdfeat, float *wgrad_slabs, uint32_t n_slabs, const uint32_t *n_valid, const AbsmaxOut *am);
// fused tail of the fp32 step: Adam+EMA of the flat 10240-float weight pack (EMA aliasing the parameter) AND the MFMA fragments of the updated weights, one launch
int ngp_mlp32_sweep_pack(void *stream, float *pack, const float *grad, float *m, float *v, float lr, float beta0, float beta1, float eps, uint32_t step, float ema_decay, float *packed_out);
// (r6) device table u16[NF32_ALL * 256 + NSPLIT_FRAGS * 512]: for every slot of the fp32 fragments, then of the split fp16 fragments (one entry for a slot's h AND m half), the
// index + 1 of the weight of the flat 10240-float pack it holds (0: a constant zero).  Built once per device from the layout functions themselves (frag_value32 /
// split_frag_weight evaluated on a ramp), so the sweep's packing is a table-driven gather instead of ~60 instructions of index arithmetic per slot.  nullptr on failure.
const uint16_t *ngp_mlp32_pack_table(void *stream);
// fp32 field forward on split fp16 operands (field_split.hip): the split fragments live behind the NF32_ALL fp32 fragments of the packed weight buffer
int ngp_field32_pack_split(void *stream, const float *wd, const float *wc, void *out_halves, int n_frags);
int ngp_field32_bwd_split(void *stream, uint32_t n, const float *feat, int layout, const float *dir, uint32_t dir_stride, const void *split_frags, const float *dout, float *dfeat,
                          float *slabs, uint32_t n_slabs, const uint32_t *n_valid, const AbsmaxOut *am);
int ngp_field32_fwd_split(void *stream, uint32_t n, const float *feat, int layout, const float *dir, uint32_t dir_stride, const void *split_frags, float *out, const uint32_t *n_valid, int density_only);
int ngp_dp_reduce(void *comm, hipStream_t s, const NgpDpPlan *plan, void *grad, int dtype, uint32_t first_bucket, uint32_t last_bucket, float *tail_f32, float *extra_f32, uint64_t extra_count);

struct LevelTable { uint32_t v[64]; };   // [16][4] = offset, size, res, scale bits — passed by value (256 B of kernarg)

// ------------------------------------------------------------------ pcg32 (ops/op_include/pcg32/pcg32.h semantics)
struct Pcg32 {
	uint64_t state, inc;
	__host__ __device__ uint32_t next_uint() {
		uint64_t old = state;
		state = old * 0x5851f42d4c957f2dULL + inc;
		uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
		uint32_t rot = (uint32_t)(old >> 59u);
		return (xs >> rot) | (xs << ((~rot + 1u) & 31));
	}
	__host__ __device__ float next_float() {
		uint32_t u = (next_uint() >> 9) | 0x3f800000u;
		float f;
#if defined(__HIP_DEVICE_COMPILE__)
		f = __uint_as_float(u);
#else
		__builtin_memcpy(&f, &u, 4);
#endif
		return f - 1.0f;
	}
	__host__ __device__ void advance(uint64_t delta) {
		uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
		while (delta > 0) {
			if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
			cur_plus = (cur_mult + 1) * cur_plus;
			cur_mult *= cur_mult;
			delta >>= 1;
		}
		state = acc_mult * state + acc_plus;
	}
};

// ------------------------------------------------------------------ morton (ray_sampler_header.h:642-667)
__host__ __device__ static inline uint32_t expand_bits(uint32_t v) {
	v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu;
	v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u; return v;
}
__host__ __device__ static inline uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) { return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2); }
__host__ __device__ static inline uint32_t morton3D_invert(uint32_t x) {
	x = x & 0x49249249; x = (x | (x >> 2)) & 0xc30c30c3; x = (x | (x >> 4)) & 0x0f00f00f;
	x = (x | (x >> 8)) & 0xff0000ff; x = (x | (x >> 16)) & 0x0000ffff; return x;
}

// Epilogue of the field backward kernels: per-lane running maxima lmax[t][pr] (level 8t + 2g + pr of the samples the lane handled; lane = 16g + s) -> one partial per
// level per workgroup.  `scratch`: >= n_waves * 16 floats of LDS nobody uses any more; all threads of the workgroup call it.
__device__ __forceinline__ void absmax_epilogue(const AbsmaxOut &am, float lmax[2][2], float *scratch, int n_waves) {
	const int lane = threadIdx.x & 63, g = lane >> 4, w = threadIdx.x >> 6;
#pragma unroll
	for (int t = 0; t < 2; ++t)
#pragma unroll
		for (int pr = 0; pr < 2; ++pr) {
			float m = lmax[t][pr];
#pragma unroll
			for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
			if ((lane & 15) == 0) scratch[w * 16 + 8 * t + 2 * g + pr] = m;
		}
	__syncthreads();
	if (threadIdx.x < 16) {
		float m = 0.f;
		for (int k = 0; k < n_waves; ++k) m = fmaxf(m, scratch[k * 16 + threadIdx.x]);
		am.parts[threadIdx.x * NGP_ABSMAX_PARTS + blockIdx.x] = (m > 0.f) ? __float_as_uint(m) : 0u;      // (NaN -> 0: a level without a usable gradient is skipped)
	}
	if (blockIdx.x == 0) {                                     // partial slots no workgroup owns, and the scatter's cursors
		for (uint32_t j = gridDim.x * 16u + threadIdx.x; j < NGP_ABSMAX_PARTS * 16u; j += blockDim.x) am.parts[(j & 15u) * NGP_ABSMAX_PARTS + (j >> 4)] = 0u;
		for (uint32_t j = threadIdx.x; j < am.n_cursors; j += blockDim.x) am.cursors[j] = 0u;
		if (threadIdx.x == 0) *am.spill_count = 0u;
	}
}

// ------------------------------------------------------------------ Adam (Jittor nn.Adam, bias-corrected) + EMA.ema_step (optims/ema.py:26-37): one element
struct AdamConsts { float step_size, b0, b1, eps, ema_decay, debias_old, debias_new, g_mul /* gradient multiplier: undoes the scale a data-parallel fp16 gradient travelled with */; };
#include <math.h>
static inline AdamConsts adam_consts(float lr, float beta0, float beta1, float eps, uint32_t step, float ema_decay, float grad_mul) {
	const double bc0 = 1.0 - pow((double)beta0, (double)step), bc1 = 1.0 - pow((double)beta1, (double)step);
	AdamConsts c;
	c.step_size = (float)((double)lr * sqrt(bc1) / bc0); c.b0 = beta0; c.b1 = beta1; c.eps = eps; c.ema_decay = ema_decay;
	c.debias_old = (float)(1.0 - pow((double)ema_decay, (double)step - 1.0));
	c.debias_new = (float)(1.0 / (1.0 - pow((double)ema_decay, (double)step)));
	c.g_mul = grad_mul;
	return c;
}
template <bool EMA>
__device__ __forceinline__ void adam_ema_update(float &p, float &m, float &v, float &e, float g, const AdamConsts &c) {
	const float mi = c.b0 * m + (1 - c.b0) * g;
	const float vi = c.b1 * v + (1 - c.b1) * g * g;
	m = mi; v = vi;
	float pi = p - mi * c.step_size / (sqrtf(vi) + c.eps);
	if (EMA) { pi = ((1 - c.ema_decay) * pi + c.ema_decay * e * c.debias_old) * c.debias_new; e = pi; }
	p = pi;
}

// (r6) The hash table's Adam + EMA sweep as a rider of the hash backward's accumulate kernel (hash_encode.hip: k_bin_accumulate2_adam).
struct AdamRide { float *p, *m, *v; __half *p_half /* the fp16 shadow the gathers read (fp16 configuration), or null */; AdamConsts c; int ema /* 0: none, else the EMA state IS the parameter (k_adam_ema's EMA == 2) */; };

// (r6) Small jobs that RIDE in the grid of the hash backward's record kernels instead of being launches of their own (mlp_tail.h; all null / zero: nothing rides).
// reduce: reduce_out[col] = sum over the MLP weight-gradient slabs, k_reduce_slabs' order, overwrite.  sweep: k_mlp32_sweep_pack's job on (pack, grad = reduce_out, m, v).
//         sweep16 (fp16 configuration): k_reduce_slabs_sweep's job - the thread that holds a column's sum also applies Adam + EMA to the parameter it is the gradient of
//         (two packs a / b whose gradients tile reduce_out).
struct PackSweep { float *p, *m, *v, *ema; __half *p_half; uint32_t begin, count; };     // columns [begin, begin + count) of the flat gradient
struct TailJobs {
	const float *slabs; uint32_t n_slabs, width; float *reduce_out;
	float *pack, *m, *v, *packed_out; AdamConsts c;
	const uint16_t *pack_table;                                          // ngp_mlp32_pack_table(): source weight (+ 1; 0 = a constant zero) of every fragment slot
	PackSweep a16, b16;
	int do_reduce, do_sweep, do_sweep16;
};

// sampler constants (density_grid_sampler.py:35-39, 96-116)
#define NGP_GRIDSIZE 128u
#define NGP_STEPS 1024u
#define NGP_SQRT3 1.73205080757f
__host__ __device__ static inline float min_cone_stepsize() { return NGP_SQRT3 / NGP_STEPS; }
