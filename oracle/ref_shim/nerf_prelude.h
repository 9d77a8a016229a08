// TEST INFRASTRUCTURE ONLY. The constant prelude that density_grid_sampler.py:96-116 generates as a
// string ("density_grad_header") before including any sampler header. CONST_DT selects the calc_dt variant.
#pragma once
#ifndef REF_NERF_CASCADES
#define REF_NERF_CASCADES 5
#endif
inline constexpr uint32_t NERF_GRIDSIZE() { return 128; }
inline constexpr float NERF_RENDERING_NEAR_DISTANCE() { return 0.05f; }
inline constexpr uint32_t NERF_STEPS() { return 1024; }
inline constexpr uint32_t NERF_CASCADES() { return REF_NERF_CASCADES; }
inline float NERF_MIN_OPTICAL_THICKNESS() { return 0.01f; }
inline constexpr float SQRT3() { return 1.73205080757f; }
inline constexpr float STEPSIZE() { return (SQRT3() / NERF_STEPS()); }
inline constexpr float MIN_CONE_STEPSIZE() { return STEPSIZE(); }
inline constexpr float MAX_CONE_STEPSIZE() { return STEPSIZE() * (1 << (NERF_CASCADES() - 1)) * NERF_STEPS() / NERF_GRIDSIZE(); }
#if CONST_DT
inline float calc_dt(float t, float cone_angle) { return MIN_CONE_STEPSIZE() * 0.5; }
#else
inline float clamp_(float val, float lower, float upper) { return val < lower ? lower : (upper < val ? upper : val); }
inline float calc_dt(float t, float cone_angle) { return clamp_(t * cone_angle, MIN_CONE_STEPSIZE(), MAX_CONE_STEPSIZE()); }
#endif
