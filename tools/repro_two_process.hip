// Minimal reproducer for VERDICT r3 "weak #3" (profiles/r03_two_process_probe.md): does ngp_hash_encode_fwd return different outputs for identical inputs when a
// SECOND PROCESS shares the GPU?  No torch, no gloo, no Python: plain HIP + the C ABI of libngp_hip.so.  Build + run (tools/repro_two_process.sh):
//   hipcc --offload-arch=gfx950 -O2 tools/repro_two_process.hip -Iinclude -Ljnerf_amd/csrc -lngp_hip -o /tmp/repro2p
//   /tmp/repro2p <seconds> <mode> [tag]         mode: 0 = back-to-back launches on one stream
//                                                      1 = hipDeviceSynchronize before every launch
//                                                      2 = every launch reads PRIVATE copies of positions and table (copied right before it, same stream)
//                                                      3 = a second stream runs an unrelated streaming kernel all the time (the marcher's role in training)
//                                                      4 = (r5) NOT this library at all: a five-line random gather out[i] = table[hash(i) % m] over the same 48 MB table - if even
//                                                          that differs between identical launches beside another process, the effect is the platform's
//                                                      6 = (r5) ONE process, two streams: a second stream runs ngp_hash_encode_fwd on OTHER positions all the time (what the three
//                                                          chunk streams of Runner.render_img do) - is concurrency inside one process enough?
//                                                      5 = control for 4: a coalesced streaming read of the same table (out[i] = 2 table[i % m])
// Every repetition writes the 16 x n x 2 fp32 features into the same output buffer and a device-side comparison counts the values that differ from the first
// repetition's.  Run one copy alone and two copies concurrently; a non-zero count in either says the kernel (or the platform under it) is not a function of its inputs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <time.h>
#include "ngp_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void k_fill(float *p, size_t n, uint32_t seed, float lo, float hi) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		uint32_t x = (uint32_t)i * 747796405u + seed; x ^= x >> 16; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
		p[i] = lo + (hi - lo) * (float)(x >> 8) * (1.0f / 16777216.0f);
	}
}
__global__ void k_diff(const uint32_t *a, const uint32_t *b, size_t n, unsigned long long *count) {
	unsigned long long c = 0;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
	if (c) atomicAdd(count, c);
}
__global__ void k_plain_gather(const float *__restrict__ tab, uint32_t m, float *__restrict__ out, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
		uint32_t x = (uint32_t)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
		out[i] = tab[x % m] + tab[(x ^ 0x9e3779b9u) % m];
	}
}
__global__ void k_plain_stream(const float *__restrict__ tab, uint32_t m, float *__restrict__ out, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = 2.0f * tab[i % m];
}
__global__ void k_noise(float *p, size_t n) {
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.0f;
}

int main(int argc, char **argv) {
	const double seconds = argc > 1 ? atof(argv[1]) : 5.0;
	const int mode = argc > 2 ? atoi(argv[2]) : 0;
	const char *tag = argc > 3 ? argv[3] : "solo";
	const uint32_t n = 1u << 21;                                   // the occupancy refresh's query size
	uint32_t table_host[64];
	const uint32_t n_params = ngp_level_table(1.0, table_host);
	float *pos, *pos2, *table, *table2, *out, *ref, *noise;
	unsigned long long *count;
	CK(hipMalloc(&pos, (size_t)n * 3 * 4)); CK(hipMalloc(&pos2, (size_t)n * 3 * 4));
	CK(hipMalloc(&table, (size_t)n_params * 4)); CK(hipMalloc(&table2, (size_t)n_params * 4));
	CK(hipMalloc(&out, (size_t)n * 32 * 4)); CK(hipMalloc(&ref, (size_t)n * 32 * 4));
	CK(hipMalloc(&noise, (size_t)64 << 20)); CK(hipMalloc(&count, 8));
	float *out2 = nullptr;
	if (mode == 6) CK(hipMalloc(&out2, (size_t)n * 32 * 4));
	hipStream_t s, s2;
	CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
	hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, pos, (size_t)n * 3, 1u, 0.0f, 1.0f);
	hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, table, (size_t)n_params, 2u, -1e-4f, 1e-4f);
	CK(hipMemsetAsync(noise, 0, (size_t)64 << 20, s));
	if (mode == 6) hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, s, pos2, (size_t)n * 3, 77u, 0.0f, 1.0f);
	CK(hipStreamSynchronize(s));
	auto run = [&](float *dst) -> int {
		const float *p = pos, *t = table;
		if (mode == 2) { if (hipMemcpyAsync(pos2, pos, (size_t)n * 3 * 4, hipMemcpyDeviceToDevice, s) != hipSuccess || hipMemcpyAsync(table2, table, (size_t)n_params * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return 1; p = pos2; t = table2; }
		if (mode == 1 && hipDeviceSynchronize() != hipSuccess) return 1;
		if (mode == 5) { hipLaunchKernelGGL(k_plain_stream, dim3(8192), dim3(256), 0, s, t, n_params, dst, (size_t)n * 32); return hipGetLastError() == hipSuccess ? 0 : 1; }
		if (mode == 4) { hipLaunchKernelGGL(k_plain_gather, dim3(8192), dim3(256), 0, s, t, n_params, dst, (size_t)n * 32); return hipGetLastError() == hipSuccess ? 0 : 1; }
		return ngp_hash_encode_fwd(s, n, p, 3, t, table_host, dst, NGP_F32, NGP_LAYOUT_SOA, nullptr);
	};
	// (r5) REPRO_REF=<file>: the reference result is computed by a run that had the GPU to itself and stored; later runs beside another process compare with THAT
	const char *ref_file = getenv("REPRO_REF");
	FILE *rf = ref_file ? fopen(ref_file, "rb") : nullptr;
	if (rf) {
		float *h = (float *)malloc((size_t)n * 32 * 4);
		if (fread(h, 4, (size_t)n * 32, rf) != (size_t)n * 32) { fprintf(stderr, "short reference file\n"); return 2; }
		fclose(rf);
		CK(hipMemcpy(ref, h, (size_t)n * 32 * 4, hipMemcpyHostToDevice)); free(h);
	} else {
		if (run(ref)) { fprintf(stderr, "hash fwd: %s\n", ngp_last_error()); return 2; }
		CK(hipStreamSynchronize(s));
		if (ref_file) {
			float *h = (float *)malloc((size_t)n * 32 * 4);
			CK(hipMemcpy(h, ref, (size_t)n * 32 * 4, hipMemcpyDeviceToHost));
			FILE *wf = fopen(ref_file, "wb"); if (wf) { fwrite(h, 4, (size_t)n * 32, wf); fclose(wf); } free(h);
		}
	}
	float *bad_host = nullptr, *ref_host = nullptr;
	unsigned long long bad_reps = 0, worst = 0, reps = 0;
	const auto t0 = std::chrono::steady_clock::now();
	while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
		if (mode == 3) for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_noise, dim3(512), dim3(256), 0, s2, noise, (size_t)16 << 20);
		if (mode == 6) for (int k = 0; k < 2; ++k) if (ngp_hash_encode_fwd(s2, n, pos2, 3, table, table_host, out2, NGP_F32, NGP_LAYOUT_SOA, nullptr)) return 2;
		CK(hipMemsetAsync(count, 0, 8, s));
		if (run(out)) { fprintf(stderr, "hash fwd: %s\n", ngp_last_error()); return 2; }
		hipLaunchKernelGGL(k_diff, dim3(2048), dim3(256), 0, s, (const uint32_t *)out, (const uint32_t *)ref, (size_t)n * 32, count);
		unsigned long long c = 0;
		CK(hipMemcpyAsync(&c, count, 8, hipMemcpyDeviceToHost, s));
		CK(hipStreamSynchronize(s));
		++reps;
		if (c) {
			++bad_reps; if (c > worst) worst = c;
			if (!bad_host) {          // keep the first differing result for the report below
				bad_host = (float *)malloc((size_t)n * 32 * 4); ref_host = (float *)malloc((size_t)n * 32 * 4);
				CK(hipMemcpy(bad_host, out, (size_t)n * 32 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ref_host, ref, (size_t)n * 32 * 4, hipMemcpyDeviceToHost));
			}
		}
	}
	CK(hipDeviceSynchronize());
	if (bad_host) {                // where the first differing repetition differs: flat index = level * 2 n + 2 sample + component (level-major pairs)
		int shown = 0; unsigned long long per_level[16] = {0};
		for (size_t i = 0; i < (size_t)n * 32; ++i) if (memcmp(&bad_host[i], &ref_host[i], 4)) {
			const unsigned level = (unsigned)(i / ((size_t)n * 2)); ++per_level[level & 15u];
			if (shown < 12) { const size_t r = i % ((size_t)n * 2); printf("    [%s] level %u sample %zu (block %zu, thread %zu) comp %zu: %.9g  vs reference %.9g\n", tag, level, r / 2, r / 2 / 256, (r / 2) % 256, r % 2, bad_host[i], ref_host[i]); ++shown; }
		}
		printf("    [%s] differing values per level:", tag); for (int l = 0; l < 16; ++l) printf(" %llu", per_level[l]); printf("\n");
	}
	printf("[%s] mode %d: %llu of %llu repetitions differ from the first result (worst: %llu of %llu values)\n", tag, mode, bad_reps, reps, worst, (unsigned long long)n * 32ull);
	return 0;
}
