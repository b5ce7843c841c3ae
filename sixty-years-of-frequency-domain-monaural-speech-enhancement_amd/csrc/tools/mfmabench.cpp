// f32 MFMA ceiling probe (tuning tool): N independent accumulators per wave, W waves per SIMD, optional LDS operand reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float seed) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = seed * (i & 15);
    __syncthreads();
    floatx16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a0 = seed + threadIdx.x, a1 = seed * 2, b0 = seed * 3, b1 = seed * 5;
    const float* pa = lds + (threadIdx.x & 63);
    for (int it = 0; it < (MODE >= 2 ? 0 : iters); ++it) {
        if (MODE == 1) {
            const int o = (it & 31) * 128;
            a0 = pa[o]; a1 = pa[o + 32]; b0 = pa[o + 4096]; b1 = pa[o + 4096 + 32];
        }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
    }
    if (MODE >= 2) {
        int kreg = (threadIdx.x & 31) * 128 + (threadIdx.x & 32 ? 64 : 0);
        const int hi = (threadIdx.x >> 5) & 1;
        float x0 = pa[0], x1 = pa[32], y0 = pa[4096], y1 = pa[4096 + 32];
        for (int it = 0; it < iters; it += 2) {
            int o;
            if (MODE == 3) { const int lo_ = __builtin_amdgcn_readlane(kreg, (it + 1) & 31), hi_ = __builtin_amdgcn_readlane(kreg, ((it + 1) & 31) + 32); o = (hi ? hi_ : lo_) & 4095; }
            else o = ((it + 1) & 31) * 128;
            const float nx0 = pa[o], nx1 = pa[o + 32], ny0 = pa[o + 4096], ny1 = pa[o + 4096 + 32];
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            int o2;
            if (MODE == 3) { const int lo_ = __builtin_amdgcn_readlane(kreg, (it + 2) & 31), hi_ = __builtin_amdgcn_readlane(kreg, ((it + 2) & 31) + 32); o2 = (hi ? hi_ : lo_) & 4095; }
            else o2 = ((it + 2) & 31) * 128;
            x0 = pa[o2]; x1 = pa[o2 + 32]; y0 = pa[o2 + 4096]; y1 = pa[o2 + 4096 + 32];
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(nx0, ny0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(nx0, ny1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(nx1, ny0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(nx1, ny1, acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 512, iters = 200000;
    float* d; hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
            else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
            else hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fl = (double)blocks * 4 * iters * 4 * 4096.0;
            if (rep) printf("mode %d blocks %d: %.3f ms %.1f TFLOP/s\n", mode, blocks, ms, fl / ms / 1e9);
        }
    }
    return 0;
}
