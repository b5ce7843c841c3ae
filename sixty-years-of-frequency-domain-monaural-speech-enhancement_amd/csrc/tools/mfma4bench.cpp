// Issue-rate probe for the small f32 MFMA shapes (tuning tool): NACC independent accumulator chains per wave, one wave per SIMD.
//   mfma4bench            prints TFLOP/s and cycles per instruction (at the measured shader clock) per shape and chain count
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int SHAPE, int NACC>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters, float seed, unsigned long long* clk) {
    floatx4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed * (threadIdx.x + i); b[i] = seed * (i + 3); }
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (SHAPE == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[(r + i) & 7], b[r], acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(r + i) & 7], b[r], acc[i], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) *clk = c1 - c0;
}
template <int SHAPE, int NACC>
void run(float* d, unsigned long long* dc, int blocks) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, NACC>), dim3(blocks), dim3(256), 0, 0, d, iters, 1e-3f, dc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    unsigned long long cyc; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 8 * NACC;            // instructions per wave
    const double fl = (SHAPE == 0 ? 512.0 : 2048.0) * n * blocks * 4;
    printf("%s  chains %2d: %.3f ms  %.1f TFLOP/s  %.2f counter ticks per instruction (s_memtime @100 MHz -> %.1f ns)\n",
           SHAPE == 0 ? "4x4x1_16b " : "16x16x4   ", NACC, ms, fl / ms / 1e9, (double)cyc / n, ms * 1e6 / n);
}
int main(int argc, char** argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 256;
    float* d; hipMalloc(&d, blocks * 256 * 4);
    unsigned long long* dc; hipMalloc(&dc, 8);
    run<0, 1>(d, dc, blocks); run<0, 2>(d, dc, blocks); run<0, 3>(d, dc, blocks); run<0, 4>(d, dc, blocks);
    run<0, 6>(d, dc, blocks); run<0, 8>(d, dc, blocks); run<0, 16>(d, dc, blocks);
    run<1, 1>(d, dc, blocks); run<1, 2>(d, dc, blocks); run<1, 4>(d, dc, blocks); run<1, 8>(d, dc, blocks);
    return 0;
}
