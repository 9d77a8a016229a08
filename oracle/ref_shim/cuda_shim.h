// TEST INFRASTRUCTURE ONLY (oracle/_ref build).
// Minimal host shim that lets the reference's CUDA kernel headers under
// /root/reference/python/jnerf/**/op_header/*.h compile as plain serial C++:
// every __global__ kernel becomes an ordinary function that reads the
// thread_local threadIdx/blockIdx/blockDim set by cpu_launch() below.
// Nothing here is shipped in the product path.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <cmath>
#include <cassert>
#include <cstdio>
#include <algorithm>
#include <type_traits>

#define __global__
#define __device__
#define __host__
#define __shared__
#define __restrict__
#define __forceinline__ inline

struct dim3 {
	unsigned x, y, z;
	dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
static thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
typedef void *cudaStream_t;
static const int warpSize = 32;

// ---- fp16 ------------------------------------------------------------------
struct __half {
	_Float16 v;
	__half() = default;
	__half(float f) : v((_Float16)f) {}
	__half(double f) : v((_Float16)f) {}
	__half(int f) : v((_Float16)f) {}
	operator float() const { return (float)v; }
	__half &operator+=(const __half &o) { v = (_Float16)(v + o.v); return *this; }
};
struct __half2 { __half x, y; };

// ---- atomics (serial launcher => plain read-modify-write) --------------------
inline float atomicAdd(float *p, float v) { float o = *p; *p = o + v; return o; }
inline uint32_t atomicAdd(uint32_t *p, uint32_t v) { uint32_t o = *p; *p = o + v; return o; }
inline int atomicAdd(int *p, int v) { int o = *p; *p = o + v; return o; }
inline __half atomicAdd(__half *p, __half v) { __half o = *p; *p += v; return o; }
inline __half2 atomicAdd(__half2 *p, __half2 v) { __half2 o = *p; p->x += v.x; p->y += v.y; return o; }
inline uint32_t atomicMax(uint32_t *p, uint32_t v) { uint32_t o = *p; if (v > o) *p = v; return o; }

inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float __expf(float x) { return expf(x); }
template <typename T> inline T __shfl_xor_sync(unsigned, T v, int) { return v; }  // block_reduce is never called
inline void __syncthreads() {}
struct int4 { int x, y, z, w; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };

// mixed-signedness min/max used by the reference (CUDA provides these overloads)
inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
inline uint32_t max(uint32_t a, uint32_t b) { return a > b ? a : b; }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline float min(float a, float b) { return a < b ? a : b; }
inline float max(float a, float b) { return a > b ? a : b; }
inline uint32_t min(uint32_t a, int b) { return a < (uint32_t)b ? a : (uint32_t)b; }
inline int min(int a, uint32_t b) { return a < (int)b ? a : (int)b; }
inline int max(int a, uint32_t b) { return a > (int)b ? a : (int)b; }

// Serial launcher: runs kernel(args...) once per (block, thread) in CUDA order.
template <typename K, typename... A>
inline void cpu_launch(dim3 grid, dim3 block, K kernel, A... args) {
	blockDim = block; gridDim = grid;
	for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
		blockIdx = dim3(bx, by, bz);
		for (unsigned tz = 0; tz < block.z; ++tz) for (unsigned ty = 0; ty < block.y; ++ty) for (unsigned tx = 0; tx < block.x; ++tx) {
			threadIdx = dim3(tx, ty, tz);
			kernel(args...);
		}
	}
}
// "lin128" launch used by every linear_kernel() call site in the reference.
template <typename K, typename... A>
inline void cpu_linear(K kernel, uint32_t n, A... args) {
	if (n == 0) return;
	cpu_launch(dim3((n + 127) / 128), dim3(128), kernel, n, args...);
}
