// Does VALU work of one wave hide under the MFMAs of another wave of the same SIMD?  (tuning tool)
// 512-thread workgroups = 2 waves per SIMD.  mode 0: all waves MFMA; 1: all waves VALU; 2: waves 0-3 MFMA, 4-7 VALU (same
// per-wave instruction counts as in 0 / 1); 3: every wave alternates 1 MFMA : 1 VALU in its own stream.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int SHAPE>
__device__ __forceinline__ void mfma_block(floatx4 (&acc)[8], const float (&a)[8], const float (&b)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[(r + i) & 7], b[r], acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(r + i) & 7], b[r], acc[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
__device__ __forceinline__ void valu_block(float (&x)[16], float m) {      // 64 dependent-free-ish FMAs
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], m, 1.0f);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int SHAPE, int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    floatx4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    float a[8], b[8], x[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + i); b[i] = seed * (i + 3); }
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = seed * i;
    const int wave = threadIdx.x >> 6;
    const bool do_m = MODE == 0 || (MODE == 2 && wave < 4), do_v = MODE == 1 || (MODE == 2 && wave >= 4);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[(r + i) & 7], b[r], acc[i], 0, 0, 0);
                    else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(r + i) & 7], b[r], acc[i], 0, 0, 0);
                    x[(r * 8 + i) & 15] = __builtin_fmaf(x[(r * 8 + i) & 15], seed, 1.0f);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            if (do_m) mfma_block<SHAPE>(acc, a, b);
            if (do_v) valu_block(x, seed);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int SHAPE, int MODE>
void run(float* d, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, MODE>), dim3(blocks), dim3(512), 0, 0, d, iters, 1e-3f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%s mode %d: %.3f ms  (per iteration of 64 MFMA and/or 64 VALU per wave: %.1f ns)\n", SHAPE == 0 ? "4x4x1 " : "16x16x4", MODE, ms, ms * 1e6 / iters);
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    run<0, 0>(d, 256); run<0, 1>(d, 256); run<0, 2>(d, 256); run<0, 3>(d, 256);
    run<1, 0>(d, 256); run<1, 1>(d, 256); run<1, 2>(d, 256); run<1, 3>(d, 256);
    return 0;
}
