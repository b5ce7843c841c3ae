// Micro-benchmark of the cooperative LSTM recurrence (tuning tool, not product path).
//   coopbench <H> <S> [T=401] [Z=1]        SE_COOP_DBG bits: 4 = no exchange barrier (wrong results), 8 = no store-acknowledge wait
#include "../kernels.h"
#include "../common.h"
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
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
    // float64 recurrence of a few sequences on the host (sequences are independent): the kernel under test against the definition
    {
        const int Tc = std::min(T, getenv("COOPBENCH_TC") ? atoi(getenv("COOPBENCH_TC")) : 24);
        std::vector<float> og((size_t)Tc * H * S);
        SE_HIP(hipMemcpy(og.data(), dout, og.size() * 4, hipMemcpyDeviceToHost));      // z = 0
        std::vector<int> seqs = {0, 1, 3, S / 2, S - 2 < 0 ? 0 : S - 2, S - 1};
        double md = 0;
        for (int n : seqs) {
            if (n < 0 || n >= S) continue;
            std::vector<double> h(H, 0.0), c(H, 0.0), hn(H);
            for (int t = 0; t < Tc; ++t) {
                for (int u = 0; u < H; ++u) {
                    double gt[4];
                    for (int q = 0; q < 4; ++q) {
                        const float* wr = &w[(size_t)(4 * u + q) * H];
                        double acc = g[((size_t)t * 4 * H + 4 * u + q) * S + n];
                        for (int k = 0; k < H; ++k) acc += (double)wr[k] * h[k];
                        gt[q] = acc;
                    }
                    auto sg = [](double x) { return 1.0 / (1.0 + exp(-x)); };
                    c[u] = sg(gt[1]) * c[u] + sg(gt[0]) * tanh(gt[2]);
                    hn[u] = sg(gt[3]) * tanh(c[u]);
                }
                h = hn;
                for (int u = 0; u < H; ++u) md = std::max(md, std::fabs(h[u] - (double)og[((size_t)t * H + u) * S + n]));
            }
        }
        printf("  max |gpu - float64 host| over %zu sequences x %d steps: %.3e\n", seqs.size(), Tc, md);
    }
    printf("coop LSTM H=%d S=%d T=%d Z=%d: %.3f ms  %.2f us/step  %.2f us/tile-step  %.1f TFLOP/s  (checksum %.6f)\n", H, S, T, Z, ms,
           ms * 1e3 / T, ms * 1e3 / T / (((S + 15) / 16 + (256 / (H / 16) / Z) - 1) / (256 / (H / 16) / Z)), 2.0 * Z * 4 * H * H * S * T / ms / 1e9, cs);
    return 0;
}
