// Micro-benchmark of the tap-table implicit-GEMM conv on one DCCRN layer shape (tuning tool, not product path).
//   gcbench <Cin> <Cout> <Fin> <B> [T=501] [deconv=0]
#include "../layers.h"
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
using namespace se;
#include "../rnn.h"
// gcbench step <H> <S>: one fused LSTM step GEMM (M = 4H, K = H, N = S sequences) with the cell epilogue
static int bench_step(int H, int S) {
    std::mt19937 rng(2);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<float> w((size_t)4 * H * H);
    for (auto& v : w) v = U(rng) * 0.05f;
    GCPlan pl = gc_make_plan(4 * H, H, one_tap(), w, {}, {}, ACT_NONE, EPI_LSTM, 1, 1, 0, S);
    float *h0, *h1, *gx, *cell;
    SE_HIP(hipMalloc(&h0, (size_t)H * S * 4));
    SE_HIP(hipMalloc(&h1, (size_t)H * S * 4));
    SE_HIP(hipMalloc(&gx, (size_t)4 * H * S * 4));
    SE_HIP(hipMalloc(&cell, (size_t)H * S * 4));
    SE_HIP(hipMemset(h0, 0, (size_t)H * S * 4));
    SE_HIP(hipMemset(gx, 0, (size_t)4 * H * S * 4));
    SE_HIP(hipMemset(cell, 0, (size_t)H * S * 4));
#ifdef GC_TIMING
    unsigned long long* dt;
    SE_HIP(hipMalloc(&dt, 64));
    SE_HIP(hipMemset(dt, 0, 64));
    pl.p.timing = dt;
#endif
    auto launch = [&]() {
        GCParams p = pl.p;
        p.first_step = 0;
        p.src0 = h0; p.s0_b = 0; p.s0_c = S; p.s0_f = 0; p.src1 = nullptr;
        p.Fin = 1; p.Tin = S; p.B = 1; p.Q = 1; p.Tout = S;
        p.aux = gx; p.x_b = 0; p.x_c = S; p.x_f = 0;
        p.dst = h1; p.d_b = 0; p.d_c = S; p.d_f = 0;
        p.cell = cell;
        gc_launch(pl, p, 0);
    };
    for (int it = 0; it < 3; ++it) launch();
    SE_HIP(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0, 0);
    for (int it = 0; it < reps; ++it) launch();
    hipEventRecord(e1, 0);
    SE_HIP(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
#ifdef GC_TIMING
    {
        SE_HIP(hipMemset(dt, 0, 64));
        launch();
        SE_HIP(hipDeviceSynchronize());
        unsigned long long h[8];
        SE_HIP(hipMemcpy(h, dt, 64, hipMemcpyDeviceToHost));
        const char* nm[6] = {"prologue/desc", "load issue", "mfma", "vmcnt wait", "barrier", "epilogue"};
        double tot = 0;
        for (int i = 0; i < 6; ++i) tot += (double)h[i];
        printf("blocks=%llu  per-block ticks:", h[6]);
        for (int i = 0; i < 6; ++i) printf("  %s %.0f (%.1f%%)", nm[i], (double)h[i] / h[6], 100.0 * h[i] / tot);
        printf("  total %.0f\n", tot / h[6]);
    }
#endif
    printf("LSTM step H=%d S=%d BM=%d BN=%d KCp=%d chunks=%d: %.1f us  %.1f TFLOP/s\n", H, S, pl.BM, pl.BN, pl.p.KCp, pl.p.nchunks,
           ms * 1e3, 2.0 * 4 * H * H * S / ms / 1e9);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 3 && std::string(argv[1]) == "step") return bench_step(atoi(argv[2]), atoi(argv[3]));
    int Cin = argc > 1 ? atoi(argv[1]) : 128, Cout = argc > 2 ? atoi(argv[2]) : 256, Fin = argc > 3 ? atoi(argv[3]) : 32;
    int B = argc > 4 ? atoi(argv[4]) : 64, T = argc > 5 ? atoi(argv[5]) : 501;
    int Fout = Fin / 2;
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    DenseW d;
    d.M = Cout; d.Cin = Cin; d.nkf = 5; d.nkt = 2;
    d.w.resize((size_t)Cout * Cin * 10);
    for (auto& v : d.w) v = U(rng) * 0.05f;
    d.bias.assign(Cout, 0.1f);
    GCPlan pl = make_conv_plan(d, 2, 2, 1, 1, 1, ACT_PRELU, std::vector<float>(Cout, 0.25f), EPI_ACT, T);
    size_t nin = (size_t)B * Cin * Fin * T, nout = (size_t)B * Cout * Fout * T;
    std::vector<float> hin(nin);
    for (auto& v : hin) v = U(rng);
    float *din, *dout;
    SE_HIP(hipMalloc(&din, nin * 4 + 4096));
    SE_HIP(hipMalloc(&dout, nout * 4));
    SE_HIP(hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice));
    gc_register_overread_range(din, nin * 4 + 4096);      // like an engine arena: 16 B group staging allowed
#ifdef GC_TIMING
    unsigned long long* dt;
    SE_HIP(hipMalloc(&dt, 128));
    SE_HIP(hipMemset(dt, 0, 128));
    pl.p.timing = dt;
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    Act4 a = act4(din, Cin, Fin, T);
    for (int it = 0; it < 2; ++it) run_conv(pl, a, nullptr, dout, Cout, Fout, B, T, T, 0);
    SE_HIP(hipDeviceSynchronize());
    const int reps = 5;
    hipEventRecord(e0, 0);
    for (int it = 0; it < reps; ++it) run_conv(pl, a, nullptr, dout, Cout, Fout, B, T, T, 0);
    hipEventRecord(e1, 0);
    SE_HIP(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    double fl = 2.0 * Cout * Cin * 10.0 * B * Fout * T;
#ifdef GC_TIMING
    {
        SE_HIP(hipMemset(dt, 0, 128));
        run_conv(pl, a, nullptr, dout, Cout, Fout, B, T, T, 0);
        SE_HIP(hipDeviceSynchronize());
        unsigned long long h[16];
        SE_HIP(hipMemcpy(h, dt, 128, hipMemcpyDeviceToHost));
        printf("prologue split: entry/decode %.0f  tab/koff/aoff %.0f  Bs zero %.0f  descriptors %.0f\n", (double)h[7] / h[6],
               (double)h[8] / h[6], (double)h[9] / h[6], (double)h[10] / h[6]);
        const char* nm[6] = {"prologue/desc", "load issue", "mfma", "vmcnt wait", "barrier", "epilogue"};
        double tot = 0;
        for (int i = 0; i < 6; ++i) tot += (double)h[i];
        printf("blocks=%llu  per-block s_memtime ticks (wave 0):", h[6]);
        for (int i = 0; i < 6; ++i) printf("  %s %.0f (%.1f%%)", nm[i], (double)h[i] / h[6], 100.0 * h[i] / tot);
        printf("  total %.0f\n", tot / h[6]);
    }
#endif
    printf("Cin=%d Cout=%d Fin=%d B=%d T=%d BM=%d BN=%d CI_C=%d KCp=%d: %.3f ms  %.1f TFLOP/s\n", Cin, Cout, Fin, B, T, pl.BM, pl.BN,
           pl.p.CI_C, pl.p.KCp, ms, fl / ms / 1e9);
    return 0;
}
