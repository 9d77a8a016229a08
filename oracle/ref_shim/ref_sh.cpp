// TEST INFRASTRUCTURE ONLY. Host build of the reference's kernel_sh
// (sh_encoder/op_header/SphericalEncode.h:45-…), launched as sh_encoder.py:29-51 does (degree 4, no padding).
#include "SphericalEncode.h"
extern "C" {
__attribute__((visibility("default"))) void ref_sh_f32(uint32_t n, const float *d, float *out) {
	cpu_linear(kernel_sh<float>, n, 4u, 0u, PitchedPtr<const float>(d, 3), PitchedPtr<float>(out, 16), (float *)nullptr);
}
__attribute__((visibility("default"))) void ref_sh_f16(uint32_t n, const float *d, void *out) {
	cpu_linear(kernel_sh<__half>, n, 4u, 0u, PitchedPtr<const float>(d, 3), PitchedPtr<__half>((__half *)out, 16), (float *)nullptr);
}
}
