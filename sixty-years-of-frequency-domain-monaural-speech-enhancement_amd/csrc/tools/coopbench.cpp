// Micro-benchmark of the cooperative LSTM recurrence (tuning tool, not product path).
//   coopbench <H> <S> [T=401] [Z=1]        SE_COOP_DBG bits: 4 = no exchange barrier (wrong results), 8 = no store-acknowledge wait
#include "../kernels.h"
#include "../common.h"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
using namespace se;

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 1024, S = argc > 2 ? atoi(argv[2]) : 256;
    const int T = argc > 3 ? atoi(argv[3]) : 401, Z = argc > 4 ? atoi(argv[4]) : 1;
    std::mt19937 rng(3);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<float> w((size_t)Z * 4 * H * H), g((size_t)Z * T * 4 * H * S);
    for (auto& v : w) v = U(rng) * 0.03f;
    for (auto& v : g) v = U(rng);
    float *dw, *dg, *dout, *dcell;
    SE_HIP(hipMalloc(&dw, w.size() * 4));
    SE_HIP(hipMalloc(&dg, g.size() * 4));
    SE_HIP(hipMalloc(&dout, (size_t)Z * T * H * S * 4));
    SE_HIP(hipMalloc(&dcell, (size_t)Z * H * S * 4));
    SE_HIP(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    SE_HIP(hipMemcpy(dg, g.data(), g.size() * 4, hipMemcpyHostToDevice));
    LstmCoopArgs a{};
    a.gx = dg; a.whh = dw; a.out = dout; a.cell = dcell;
    a.gx_z = (long)T * 4 * H * S; a.gx_t = (long)4 * H * S; a.gx_row = S;
    a.whh_z = (long)4 * H * H;
    a.out_z = (long)T * H * S; a.out_t = (long)H * S; a.out_row = S;
    a.H = H; a.T = T; a.S = S; a.Z = Z; a.reverse = 0;
    for (int it = 0; it < 2; ++it) launch_lstm_coop(a, 0);
    SE_HIP(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 5;
    hipEventRecord(e0, 0);
    for (int it = 0; it < reps; ++it) launch_lstm_coop(a, 0);
    hipEventRecord(e1, 0);
    SE_HIP(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<float> o(64);
    SE_HIP(hipMemcpy(o.data(), dout + (size_t)(T - 1) * H * S, 64 * 4, hipMemcpyDeviceToHost));
    double cs = 0; for (float v : o) cs += v;
    printf("coop LSTM H=%d S=%d T=%d Z=%d: %.3f ms  %.2f us/step  %.2f us/tile-step  %.1f TFLOP/s  (checksum %.6f)\n", H, S, T, Z, ms,
           ms * 1e3 / T, ms * 1e3 / T / (((S + 15) / 16 + (256 / (H / 16) / Z) - 1) / (256 / (H / 16) / Z)), 2.0 * Z * 4 * H * H * S * T / ms / 1e9, cs);
    return 0;
}
