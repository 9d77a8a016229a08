// TEST INFRASTRUCTURE ONLY. Builds the reference's own hash-grid kernels
// (/root/reference/python/jnerf/models/position_encoders/hash_encoder/op_header/HashEncode.h)
// for the host. The launch sequence restates the cuda_src glue of
// hash_encoder/grid_encode.py:71-125 (forward) and :137-184 (backward).
#include <vector>
#include <cmath>
// hash_func from projects/ngp/configs/ngp_base.py:69, injected exactly like hash_encoder.py:14-16
#define get_index(p0,p1,p2) p0 ^ p1 * 19349663 ^ p2 * 83492791
#include "HashEncode.h"

template <typename T>
static void fwd(uint32_t n, const float *x, const T *grid, const uint32_t *offsets, double per_level_scale, T *out) {
	if (n == 0) return;
	std::vector<float> positions((size_t)n * 3);
	std::vector<vector_t<T, 2>> enc((size_t)n * 16);
	cpu_launch(dim3(div_round_up(n, 64u)), dim3(64, 3, 1), extract_position<float, 3>, n, PitchedPtr<const float>(x, 3), positions.data());
	cpu_launch(dim3(div_round_up(n, 512u), 16, 1), dim3(512), kernel_grid<T, 3, 2>, n, 32u, offsets, 16u,
	           (float)std::log2(per_level_scale), 0.0f, 1000.0f, 1u, 0u, grid, (const float *)positions.data(), enc.data(), (float *)nullptr);
	cpu_launch(dim3(div_round_up(n, 8u)), dim3(16, 8, 1), transpose_encoded_position<vector_t<T, 2>>, n,
	           (const vector_t<T, 2> *)enc.data(), PitchedPtr<vector_t<T, 2>>(PitchedPtr<T>(out, 32)));
}

// the same launch with the dy_dx output enabled (grid_encode.py:96 passes nullptr; the branch HashEncode.h:205-251 is compiled either way)
template <typename T>
static void fwd_dydx(uint32_t n, const float *x, const T *grid, const uint32_t *offsets, double per_level_scale, T *out, float *dydx) {
	if (n == 0) return;
	std::vector<float> positions((size_t)n * 3);
	std::vector<vector_t<T, 2>> enc((size_t)n * 16);
	cpu_launch(dim3(div_round_up(n, 64u)), dim3(64, 3, 1), extract_position<float, 3>, n, PitchedPtr<const float>(x, 3), positions.data());
	cpu_launch(dim3(div_round_up(n, 512u), 16, 1), dim3(512), kernel_grid<T, 3, 2>, n, 32u, offsets, 16u,
	           (float)std::log2(per_level_scale), 0.0f, 1000.0f, 1u, 0u, grid, (const float *)positions.data(), enc.data(), dydx);
	cpu_launch(dim3(div_round_up(n, 8u)), dim3(16, 8, 1), transpose_encoded_position<vector_t<T, 2>>, n,
	           (const vector_t<T, 2> *)enc.data(), PitchedPtr<vector_t<T, 2>>(PitchedPtr<T>(out, 32)));
}

template <typename T>
static void bwd(uint32_t n, const float *x, const T *dy, const uint32_t *offsets, double per_level_scale, T *grad, size_t n_params) {
	std::memset(grad, 0, n_params * sizeof(T));
	if (n == 0) return;
	std::vector<float> positions((size_t)n * 3);
	std::vector<vector_t<T, 2>> tr((size_t)n * 16);
	cpu_launch(dim3(div_round_up(n, 64u)), dim3(64, 3, 1), extract_position<float, 3>, n, PitchedPtr<const float>(x, 3), positions.data());
	cpu_launch(dim3(div_round_up(n, 8u)), dim3(16, 8, 1), transpose_gradients<vector_t<T, 2>>, n, tr.data(),
	           PitchedPtr<const vector_t<T, 2>>(PitchedPtr<const T>(dy, 32)));
	cpu_launch(dim3(div_round_up(n * 2 / 2, 256u), 16, 1), dim3(256), kernel_grid_backward<T, T, 3, 2, 2>, n, 32u, offsets, 16u,
	           (float)std::log2(per_level_scale), 1000.0f, false, 1u, 0u, grad, (const float *)positions.data(),
	           (const vector_t<T, 2> *)tr.data());
}

extern "C" {
__attribute__((visibility("default"))) void ref_hash_fwd_f32(uint32_t n, const float *x, const float *grid, const uint32_t *offsets, double s, float *out) { fwd<float>(n, x, grid, offsets, s, out); }
__attribute__((visibility("default"))) void ref_hash_fwd_f16(uint32_t n, const float *x, const void *grid, const uint32_t *offsets, double s, void *out) { fwd<__half>(n, x, (const __half *)grid, offsets, s, (__half *)out); }
__attribute__((visibility("default"))) void ref_hash_fwd_dydx_f32(uint32_t n, const float *x, const float *grid, const uint32_t *offsets, double s, float *out, float *dydx) { fwd_dydx<float>(n, x, grid, offsets, s, out, dydx); }
__attribute__((visibility("default"))) void ref_hash_fwd_dydx_f16(uint32_t n, const float *x, const void *grid, const uint32_t *offsets, double s, void *out, float *dydx) { fwd_dydx<__half>(n, x, (const __half *)grid, offsets, s, (__half *)out, dydx); }
__attribute__((visibility("default"))) void ref_hash_bwd_f32(uint32_t n, const float *x, const float *dy, const uint32_t *offsets, double s, float *grad, uint64_t n_params) { bwd<float>(n, x, dy, offsets, s, grad, n_params); }
__attribute__((visibility("default"))) void ref_hash_bwd_f16(uint32_t n, const float *x, const void *dy, const uint32_t *offsets, double s, void *grad, uint64_t n_params) { bwd<__half>(n, x, (const __half *)dy, offsets, s, (__half *)grad, n_params); }
}
