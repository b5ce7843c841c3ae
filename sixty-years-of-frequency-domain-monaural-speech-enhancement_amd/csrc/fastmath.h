// Transcendentals of the activations, cells and compression stages in ONE place.
//
// Default build: the gfx950 hardware approximations (v_exp_f32 / v_log_f32 / v_rcp_f32 / v_sqrt_f32 / v_rsq_f32, 1 ulp each) -
// round 4 moved every sigmoid / tanh / ELU / softplus / swish / softmax onto them.
// -DSE_EXACT_MATH (make exact -> libse_engine_exact.so): libm's correctly rounded expf / logf / sqrtf and IEEE divisions in
// the same places.  Not a product path: `SE_ENGINE_LIB=.../libse_engine_exact.so python tools/parity_record.py` measures what the
// approximations cost against the reference fixtures (profiles/r05_parity.json), i.e. how much of the 1e-4 bar they use.
#pragma once
#include <hip/hip_runtime.h>

namespace se {

#ifdef SE_EXACT_MATH
__device__ __forceinline__ float fm_exp(float x) { return expf(x); }
__device__ __forceinline__ float fm_exp2(float x) { return exp2f(x); }
__device__ __forceinline__ float fm_log(float x) { return logf(x); }
__device__ __forceinline__ float fm_rcp(float x) { return 1.f / x; }
__device__ __forceinline__ float fm_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ float fm_rsq(float x) { return 1.f / sqrtf(x); }
__device__ __forceinline__ float fm_expm1(float x) { return expm1f(x); }
__device__ __forceinline__ float fm_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float fm_tanh(float x) { return tanhf(x); }
constexpr bool kExactMath = true;
#else
__device__ __forceinline__ float fm_exp(float x) { return __expf(x); }
__device__ __forceinline__ float fm_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fm_log(float x) { return __logf(x); }
__device__ __forceinline__ float fm_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fm_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fm_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
// (near zero, where exp(v) - 1 cancels, the series v + v^2 / 2 is exact to 2e-10)
__device__ __forceinline__ float fm_expm1(float x) { return x > -1e-3f ? fmaf(0.5f * x, x, x) : __expf(x) - 1.f; }
__device__ __forceinline__ float fm_softplus(float x) { return x > 20.f ? x : __logf(1.f + __expf(x)); }      // absolute error < 1e-7
__device__ __forceinline__ float fm_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }
constexpr bool kExactMath = false;
#endif
__device__ __forceinline__ float fm_sigmoid(float x) { return fm_rcp(1.f + fm_exp(-x)); }

}  // namespace se
