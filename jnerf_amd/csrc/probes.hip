// Micro-probes (not part of the public ABI; tools/microbench_*.py only): LDS float-atomic cost versus same-address conflict degree.
#include "ngp_common.h"

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
