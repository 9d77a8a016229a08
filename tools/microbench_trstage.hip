// (r6) Staging of the split field backward's weight-gradient operands: is a [sample][neuron] LDS image written with 8-byte stores and read back with the hardware
// transpose read (ds_read_b64_tr_b16) (a) what the ISA text suggests it is, (b) free of bank conflicts under the swizzle below, (c) faster than today's [neuron][sample]
// image (sixty-four 2-byte stores per lane and phase, 16-byte reads)?  Plain HIP, no torch: hipcc --offload-arch=gfx950 -O3 tools/microbench_trstage.hip -o /tmp/trstage && /tmp/trstage
// Emulates phase A of k_field32_bwd_split (dG1 rows 0..63 | G0 rows 64..127, two planes, 128 samples per trip, eight waves; two 16 x 16 tiles per wave over all 128 samples):
// LDS traffic only (no MFMAs), one 512-thread workgroup per CU, `iters` trips.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef short s4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef short s8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- new image: plane p, sample row s (0..127), neuron chunk c (0..31, four neurons each) -> offset in halves.  Row = 256 B, chunk index XORed with a 5-bit function of the row:
//   stores (16-lane group = 16 consecutive samples, one chunk):       (c ^ f(s)) mod 16 distinct  <=  f mod 16 is a bijection of s mod 16
//   transpose reads (16-lane group = 4 samples x 4 chunks; 32 lanes = two groups 8 samples apart): (c ^ f(s)) distinct over the 32  <=  bits 0-1 from s >> 2, bits 2-3 from s & 3, bit 4 from s >> 3
#define TPLANE (128 * 128)
__device__ __forceinline__ int f_of(int s) { return ((s & 3) << 2) | ((s >> 2) & 3) | (((s >> 3) & 1) << 4); }
__device__ __forceinline__ int soff(int p, int s, int c) { return p * TPLANE + s * 128 + ((c ^ f_of(s)) << 2); }
__device__ __forceinline__ s4 ld_tr(const short *lds, int off) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)(lds + off)); }

// (a) semantics: S[s][n] = s * 128 + n in both planes (+ 0x4000 in plane 1); every lane checks the eight values of an A operand read
__global__ __launch_bounds__(512) void k_check(int *bad, int swz) {
	extern __shared__ __attribute__((aligned(16))) short lds[];
	for (int i = threadIdx.x; i < 2 * TPLANE; i += 512) lds[i] = 0;
	__syncthreads();
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, s16 = lane & 15, g = lane >> 4;
	// the store pattern of the kernel: lane (sample 16 w + s16, group g) owns neuron chunks 8 kb + g and 8 kb + 4 + g of every 32-neuron block kb (k64 order)
	const int s = 16 * w + s16;
	for (int kb = 0; kb < 4; ++kb)
		for (int h = 0; h < 2; ++h) {
			const int c = 8 * kb + 4 * h + g;
			for (int p = 0; p < 2; ++p) {
				s4 v;
				for (int j = 0; j < 4; ++j) v[j] = (short)((s * 128 + 4 * c + j) & 0x3fff) | (short)(p ? 0x4000 : 0);
				*reinterpret_cast<s4 *>(lds + (swz ? soff(p, s, c) : p * TPLANE + s * 128 + 4 * c)) = v;
			}
		}
	__syncthreads();
	int nbad = 0;
	const int o = lane & 15;
	for (int R = 0; R < 128; R += 16)                 // neuron tile
		for (int c0 = 0; c0 < 128; c0 += 32)          // k-step of 32 samples
			for (int p = 0; p < 2; ++p)
				for (int h = 0; h < 2; ++h) {
					const int s0 = c0 + 8 * g + 4 * h;
					const int row = s0 + (o >> 2), ch = R / 4 + (o & 3);
					const s4 v = ld_tr(lds, swz ? soff(p, row, ch) : p * TPLANE + row * 128 + 4 * ch);
					for (int j = 0; j < 4; ++j) {
						const short want = (short)(((s0 + j) * 128 + R + o) & 0x3fff) | (short)(p ? 0x4000 : 0);
						nbad += v[j] != want;
					}
				}
	if (nbad) atomicAdd(bad, nbad);
}

// (c) timing.  MODE 0: today's image ([neuron][sample], row stride 136 halves, 2-byte stores, 16-byte reads); 1: new image, linear; 2: new image, swizzled
template <int MODE>
__global__ __launch_bounds__(512) void k_time(int iters, unsigned long long *cyc, int *sink) {
	extern __shared__ __attribute__((aligned(16))) short lds[];
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, s16 = lane & 15, g = lane >> 4, o = lane & 15;
	const int col = 16 * w + s16;
	short val = (short)threadIdx.x;
	int acc = 0;
	__syncthreads();
	const unsigned long long t0 = clock64();
	for (int it = 0; it < iters; ++it) {
		// ---- stores: two B2 pairs (64 neurons each: rows 0..63 and 64..127), h and m planes
		if (MODE == 0) {
			constexpr int SRS = 136, SPLANE = 128 * SRS;
#pragma unroll
			for (int blk = 0; blk < 2; ++blk)
#pragma unroll
				for (int kb = 0; kb < 2; ++kb)
#pragma unroll
					for (int j = 0; j < 8; ++j) {
						const int r = (64 * blk + 32 * kb + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4))) * SRS + col;
						lds[r] = val; lds[SPLANE + r] = (short)(val + j);
					}
		} else {
#pragma unroll
			for (int blk = 0; blk < 2; ++blk)
#pragma unroll
				for (int kb = 0; kb < 2; ++kb)
#pragma unroll
					for (int h = 0; h < 2; ++h) {
						const int c = 16 * blk + 8 * kb + 4 * h + g;
						s4 v; v[0] = val; v[1] = (short)(val + 1); v[2] = (short)(val + h); v[3] = (short)(val + kb);
#pragma unroll
						for (int p = 0; p < 2; ++p) *reinterpret_cast<s4 *>(lds + (MODE == 2 ? soff(p, col, c) : p * TPLANE + col * 128 + 4 * c)) = v;
					}
		}
		__syncthreads();
		// ---- reads: two 16 x 16 tiles (gradient rows 16 to .., activation rows 64 + 16 ti ..), four k-steps of 32 samples, operands ah, am, bh, bm
		const int to = w >> 1, ti0 = 2 * (w & 1);
#pragma unroll
		for (int q = 0; q < 2; ++q)
#pragma unroll 2
			for (int c0 = 0; c0 < 128; c0 += 32) {
				if (MODE == 0) {
					constexpr int SRS = 136, SPLANE = 128 * SRS;
					const int cs = c0 + 8 * g;
					const s8 ah = *reinterpret_cast<const s8 *>(lds + (16 * to + o) * SRS + cs), am = *reinterpret_cast<const s8 *>(lds + SPLANE + (16 * to + o) * SRS + cs);
					const s8 bh = *reinterpret_cast<const s8 *>(lds + (64 + 16 * (ti0 + q) + o) * SRS + cs), bm = *reinterpret_cast<const s8 *>(lds + SPLANE + (64 + 16 * (ti0 + q) + o) * SRS + cs);
					acc += ah[0] + am[1] + bh[2] + bm[3] + ah[7] + bm[6];
				} else {
#pragma unroll
					for (int p = 0; p < 2; ++p)
#pragma unroll
						for (int h = 0; h < 2; ++h) {
							const int row = c0 + 8 * g + 4 * h + (o >> 2);
							const int ca = 4 * to + (o & 3), cb = 16 + 4 * (ti0 + q) + (o & 3);
							const s4 a = ld_tr(lds, MODE == 2 ? soff(p, row, ca) : p * TPLANE + row * 128 + 4 * ca);
							const s4 b = ld_tr(lds, MODE == 2 ? soff(p, row, cb) : p * TPLANE + row * 128 + 4 * cb);
							acc += a[0] + a[3] + b[1] + b[2];
						}
				}
			}
		__syncthreads();
		val = (short)(val + (short)acc);
	}
	const unsigned long long t1 = clock64();
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
	if (acc == 0x12345678) *sink = acc;
}

int main() {
	int *bad; unsigned long long *cyc; int *sink;
	CHECK(hipMalloc(&bad, 4)); CHECK(hipMalloc(&cyc, 256 * 8)); CHECK(hipMalloc(&sink, 4));
	const size_t lds_new = 2 * TPLANE * 2, lds_old = 2 * 128 * 136 * 2;
	CHECK(hipFuncSetAttribute((const void *)k_check, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_new));
	for (int swz = 0; swz < 2; ++swz) {
		CHECK(hipMemset(bad, 0, 4));
		hipLaunchKernelGGL(k_check, dim3(1), dim3(512), lds_new, 0, bad, swz);
		int h = -1; CHECK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
		printf("transpose-read semantics, %s image: %d wrong values of %d\n", swz ? "swizzled" : "linear", h, 512 * 8 * 4 * 2 * 2 * 4);
	}
	const int iters = 2000;
	auto run = [&](auto kern, size_t lds, const char *name) {
		CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
		hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, 10, cyc, sink);
		CHECK(hipEventRecord(a));
		hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, iters, cyc, sink);
		CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
		float ms; CHECK(hipEventElapsedTime(&ms, a, b));
		std::vector<unsigned long long> c(256); CHECK(hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
		double avg = 0; for (auto v : c) avg += (double)v; avg /= 256;
		printf("%-58s %8.3f us per trip (one workgroup per CU, 256 workgroups), %8.0f cycles per trip\n", name, ms * 1e3 / iters, avg / iters);
	};
	run(k_time<0>, lds_old, "today: [neuron][sample], 2-byte stores, 16-byte reads");
	run(k_time<1>, lds_new, "new: [sample][neuron] linear, 8-byte stores, tr reads");
	run(k_time<2>, lds_new, "new: [sample][neuron] swizzled, 8-byte stores, tr reads");
	return 0;
}
