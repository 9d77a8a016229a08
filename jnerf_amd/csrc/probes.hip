// Micro-probes (not part of the public ABI; tools/microbench_*.py only): LDS float-atomic cost versus same-address conflict degree.
#include "hash_common.h"
#include <stdlib.h>

__global__ __launch_bounds__(1024) void k_probe_lds_atomic(uint32_t iters, uint32_t distinct, float *out, int use_int) {
	__shared__ float acc[32768];
	for (uint32_t e = threadIdx.x; e < 32768; e += 1024) acc[e] = 0.f;
	__syncthreads();
	// lane l of every wave hits address ((l % distinct) * 67 + wave * 3) : `distinct` different addresses per wave instruction => 64/distinct-way conflict
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t a = ((lane % distinct) * 67u + wave * 3u) & 32767u;
	for (uint32_t i = 0; i < iters; ++i) {
		if (use_int == 1) atomicAdd(reinterpret_cast<uint32_t *>(acc) + a, 1u);
		else if (use_int == 2) __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(acc) + (a >> 1), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		else if (use_int == 3) {
			typedef _Float16 __attribute__((ext_vector_type(2))) h2;
			h2 one = {(_Float16)1.0f, (_Float16)0.5f};
			(void)__builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) h2 *)(reinterpret_cast<h2 *>(acc) + a), one);
		}
		else __hip_atomic_fetch_add(&acc[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		a = (a + 64u * 67u) & 32767u;
	}
	__syncthreads();
	if (threadIdx.x == 0) out[blockIdx.x] = acc[0] + acc[67];
}
NGP_API int ngp_x_probe_lds_atomic(void *stream, uint32_t blocks, uint32_t iters, uint32_t distinct, float *out, int use_int) {
	hipLaunchKernelGGL(k_probe_lds_atomic, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, iters, distinct, out, use_int);
	NGP_LAUNCH_CHECK("ngp_x_probe_lds_atomic");
	return 0;
}

// ---- memory-stream probes (tools/microbench_stream.py): what the chip sustains for the access patterns of the hash-backward record lists
//   mode 0  read    every thread sums 16-byte loads, grid-stride              (read roofline)
//   mode 1  write   every thread stores 16-byte values, grid-stride            (write roofline)
//   mode 2  copy    b[i] = a[i]
//   mode 3  append  the record kernels' pattern without their arithmetic: `streams` lists `spacing` bytes apart, workgroup w writes a fragment of `frag` bytes
//                   at offset w * frag of EVERY list (consecutive threads -> consecutive 16 bytes of a fragment, then the next list)
__global__ __launch_bounds__(1024) void k_probe_stream(int mode, uint64_t n16, const uint4 *__restrict__ a, uint4 *__restrict__ b, uint32_t streams, uint32_t frag, uint64_t spacing, float *__restrict__ sink) {
	const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
	if (mode == 0) {
		uint32_t acc = 0;
		for (uint64_t i = tid; i < n16; i += nth) { const uint4 v = a[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
		if (acc == 0x12345u) sink[0] = 1.f;
	} else if (mode == 1) {
		const uint4 v = make_uint4((uint32_t)tid, 1u, 2u, 3u);
		for (uint64_t i = tid; i < n16; i += nth) b[i] = v;
	} else if (mode == 2) {
		for (uint64_t i = tid; i < n16; i += nth) b[i] = a[i];
	} else {
		const uint32_t per = frag / 16u, total = streams * per;                 // 16-byte units this workgroup writes
		const uint4 v = make_uint4(blockIdx.x, threadIdx.x, 2u, 3u);
		for (uint32_t p = threadIdx.x; p < total; p += blockDim.x) {
			const uint32_t s = p / per, j = p - s * per;
			b[((uint64_t)s * spacing + (uint64_t)blockIdx.x * frag) / 16u + j] = v;
		}
	}
}
NGP_API int ngp_x_probe_stream(void *stream, int mode, uint32_t blocks, uint32_t threads, uint64_t n16, const void *a, void *b, uint32_t streams, uint32_t frag, uint64_t spacing, float *sink) {
	hipLaunchKernelGGL(k_probe_stream, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, mode, n16, (const uint4 *)a, (uint4 *)b, streams, frag, spacing, sink);
	NGP_LAUNCH_CHECK("ngp_x_probe_stream");
	return 0;
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
