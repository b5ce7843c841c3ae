// LSTM layer helpers shared by the models.
//   * weights come from torch.nn.LSTM state-dict entries (gate order i,f,g,o; b_ih + b_hh folded into the input
//     projection); rows are gate-interleaved (4u+g) for the fused cell epilogues.
//   * time-major activations: x [T][I][S], gates G [T][4H][S], h [T][H][S]  (S sequences contiguous).
#pragma once
#include "model.h"
#include <algorithm>
#include <map>
#include <vector>

namespace se {

struct LstmW {
    DenseW wih;   // [4H][I], rows interleaved, bias = b_ih + b_hh
    DenseW whh;   // [4H][H], rows interleaved
    int I = 0, H = 0;
};

inline LstmW load_lstm(const TrackedSD& sd, const std::string& prefix, int layer, const std::string& suffix, int I,
                       int H) {
    const std::string l = "_l" + std::to_string(layer) + suffix;
    LstmW w;
    w.I = I;
    w.H = H;
    w.wih = linear_weights(sd.get(prefix + "weight_ih" + l, {4 * H, I}), nullptr);
    const HostTensor& bi = sd.get(prefix + "bias_ih" + l, {4 * H});
    const HostTensor& bh = sd.get(prefix + "bias_hh" + l, {4 * H});
    for (int i = 0; i < 4 * H; ++i) w.wih.bias[i] = bi.data[i] + bh.data[i];
    w.whh = linear_weights(sd.get(prefix + "weight_hh" + l, {4 * H, H}), nullptr);
    const auto perm = lstm_gate_perm(H);
    permute_rows(w.wih, perm);
    permute_rows(w.whh, perm);
    return w;
}

// torch.nn.GRU layer (FullSubNet `sequence_model="GRU"`, sequence_model.py:36-43) on the LSTM step kernel's 4-rows-per-unit
// layout: unit u owns rows 4u + {r, z, n, -}.  Input projection rows: W_ir / W_iz / W_in with biases b_ir + b_hr, b_iz + b_hz,
// b_in, and a zero row whose bias is b_hn (the cell epilogue reads it as the constant inside r * (W_hn h + b_hn));
// recurrent rows: W_hr / W_hz / W_hn and a zero row.  A quarter of the matrix work is padding - the decode scripts never
// select the GRU (fullsubnet_sa_decode_vb.py:16), it is built for completeness of the north star's "LSTM/GRU time step".
inline LstmW load_gru(const TrackedSD& sd, const std::string& prefix, int layer, const std::string& suffix, int I, int H) {
    const std::string l = "_l" + std::to_string(layer) + suffix;
    const HostTensor& wi = sd.get(prefix + "weight_ih" + l, {3 * H, I});
    const HostTensor& wh = sd.get(prefix + "weight_hh" + l, {3 * H, H});
    const HostTensor& bi = sd.get(prefix + "bias_ih" + l, {3 * H});
    const HostTensor& bh = sd.get(prefix + "bias_hh" + l, {3 * H});
    auto pad4 = [&](const HostTensor& w, int K) {
        HostTensor o;
        o.shape = {4 * H, K};
        o.data.assign((size_t)4 * H * K, 0.f);
        for (int u = 0; u < H; ++u)
            for (int g = 0; g < 3; ++g)
                std::copy(w.data.begin() + ((size_t)g * H + u) * K, w.data.begin() + ((size_t)g * H + u + 1) * K,
                          o.data.begin() + ((size_t)4 * u + g) * K);
        return o;
    };
    LstmW w;
    w.I = I;
    w.H = H;
    w.wih = linear_weights(pad4(wi, I), nullptr);
    w.whh = linear_weights(pad4(wh, H), nullptr);
    for (int u = 0; u < H; ++u) {
        w.wih.bias[4 * u + 0] = bi.data[u] + bh.data[u];
        w.wih.bias[4 * u + 1] = bi.data[H + u] + bh.data[H + u];
        w.wih.bias[4 * u + 2] = bi.data[2 * H + u];
        w.wih.bias[4 * u + 3] = bh.data[2 * H + u];
    }
    return w;
}

// per-stream state of the frame-online mode: history columns of every chunk tensor, LSTM (h, c).  A new stream on the same
// engine re-uses the buffers of the last one (release() parks them by size, zeros() takes a parked buffer of the wanted size
// and clears it on the stream): starting a stream costs memsets, not a device-synchronising hipFree + hipMalloc per buffer
// (ADVICE r2 - per-utterance streams would serialise with every other stream on the device).
struct StreamState {
    int B = 0;
    bool first = true;
    std::vector<float*> hist;
    float *h[4] = {}, *c[4] = {};
    std::multimap<size_t, float*> parked;
    std::map<float*, size_t> size_of;
    void release() {
        auto park = [&](float*& p) {
            if (p) parked.emplace(size_of[p], p);
            p = nullptr;
        };
        for (float*& p : hist) park(p);
        hist.clear();
        for (int l = 0; l < 4; ++l) {
            park(h[l]);
            park(c[l]);
        }
        B = 0;
    }
    float* zeros(size_t n, hipStream_t st) {
        n = std::max<size_t>(n, 1);
        float* p = nullptr;
        auto it = parked.find(n);
        if (it != parked.end()) {
            p = it->second;
            parked.erase(it);
        } else {
            // (+ 16 B and registered with gemmconv: a step GEMM stages its h_{t-1} rows in 16 B groups, and the last group of a
            // row whose sequence count is no multiple of 4 reaches up to 12 B past the tensor)
            SE_HIP(hipMalloc(&p, (n + 4) * sizeof(float)));
            gc_register_overread_range(p, (n + 4) * sizeof(float));
            size_of[p] = n;
        }
        SE_HIP(hipMemsetAsync(p, 0, n * sizeof(float), st));
        return p;
    }
    ~StreamState() {
        release();
        for (auto& kv : parked) {
            gc_unregister_overread_range(kv.second);
            (void)hipFree(kv.second);
        }
    }
};

inline TapSpec one_tap() {
    TapSpec t;
    t.ntaps = 1;
    t.df[0] = 0;
    t.dt[0] = 0;
    return t;
}

// Pointwise (1x1 / Linear) plan over channels.
inline GCPlan make_pointwise_plan(const DenseW& d, int act, const std::vector<float>& slope, int tout_hint,
                                  int epi = EPI_ACT) {
    return gc_make_plan(d.M, d.Cin, one_tap(), d.w, d.bias, slope, act, epi, 1, 1, 0, tout_hint);
}

// Launch a pointwise plan on a time-major / generic 3-level tensor: element (o, c, n) at o*s_o + c*s_c + n.
inline void run_pointwise(const GCPlan& pl, const float* src, long s_o, long s_c, float* dst, long d_o, long d_c, int O,
                          int N, hipStream_t st, Profiler* prof) {
    GCParams p = pl.p;
    p.src0 = src;
    p.s0_b = s_o;
    p.s0_c = s_c;
    p.s0_f = 0;
    p.src1 = nullptr;
    p.Fin = 1;
    p.Tin = N;
    p.B = O;
    p.Q = 1;
    p.Tout = N;
    p.dst = dst;
    p.d_b = d_o;
    p.d_c = d_c;
    p.d_f = 0;
    gc_launch_prof(pl, p, st, prof);
}

// the same with the row length N decoupled from the row pitches (a column range of wider rows)
inline void run_pointwise_cols(const GCPlan& pl, const float* src, long s_o, long s_c, float* dst, long d_o, long d_c, int O,
                               int N, hipStream_t st, Profiler* prof) {
    run_pointwise(pl, src, s_o, s_c, dst, d_o, d_c, O, N, st, prof);
}

// LSTM layer with a hidden size too large for register-resident weights (H = 1024 in LSTM/CRN): input projection
// as one GEMM over all steps, then one fused GEMM + LSTM-cell launch per step (weights stream from L2 / Infinity Cache).
struct LstmBig {
    GCPlan gin, step;
    GCPlan step_x;                // recurrent step + input projection in one GEMM (K = H + I): layers with a narrow input
    bool has_x = false;
    GCPlan gin_fm;                // the input projection planned for feature-major activations (rows of T * S frames)
    float* whh_dev = nullptr;     // row-major [4H][H] (gate-interleaved rows) for the weight-stationary cooperative kernel
    float *wih_dev = nullptr, *bih_dev = nullptr;      // row-major [4H][I] and [4H] (b_ih + b_hh) for the one-sequence stack kernel
    int I = 0, H = 0;
    // gru: w comes from load_gru (GRU cell in the step epilogue; one launch per step - the weight-stationary cooperative
    // kernel only knows the LSTM cell)
    // fuse_x: also pack the [W_hh | W_ih] plan of a step that projects its own input (run_cols_x) - opt-in, only FullSubNet's
    // sub-band layers run it (for LSTM / CRN's 1024-wide layers it would be 19-33 MB of dead device memory per layer)
    void build(const LstmW& w, int s_hint, bool gru = false, bool fuse_x = false) {
        I = w.I;
        H = w.H;
        gin = make_pointwise_plan(w.wih, ACT_NONE, {}, s_hint);
        step = gc_make_plan(4 * H, H, one_tap(), w.whh.w, {}, {}, ACT_NONE, EPI_LSTM, 1, 1, 0, s_hint);
        step.p.gru = gru ? 1 : 0;
        static const int fuse_i = getenv("SE_LSTM_FUSE_I") ? atoi(getenv("SE_LSTM_FUSE_I")) : 384;
        if (fuse_x && !gru && I <= fuse_i) {
            // Layers that run one GEMM launch per step over many sequences (FullSubNet's sub-band LSTMs: 4 x 384 gate rows over
            // 257 * B sequences): the batched input projection writes [T][4H][S] gate pre-activations - 51 GB at 128 clips, read
            // back one step at a time - and for the 32-feature first layer that write IS its cost.  As a second source of the
            // step GEMM's K loop the projection costs its matrix work (8 % of a step for the first layer, as much again as
            // the step for the 384-feature second one) and no traffic: 414 -> 449 (first layer) -> 463 utt/s (both).
            std::vector<float> wcat((size_t)4 * H * (H + I));
            for (int m = 0; m < 4 * H; ++m) {
                for (int k = 0; k < H; ++k) wcat[(size_t)m * (H + I) + k] = w.whh.w[(size_t)m * H + k];
                for (int k = 0; k < I; ++k) wcat[(size_t)m * (H + I) + H + k] = w.wih.w[(size_t)m * I + k];
            }
            step_x = gc_make_plan(4 * H, H + I, one_tap(), wcat, w.wih.bias, {}, ACT_NONE, EPI_LSTM, 1, 1, 0, s_hint, 1, H);
            has_x = true;
        }
        if (!gru && (H == 512 || H == 1024)) {
            whh_dev = to_device(w.whh.w);
            if (I == H) {
                wih_dev = to_device(w.wih.w);
                bih_dev = to_device(w.wih.bias);
            }
            gin_fm = make_pointwise_plan(w.wih, ACT_NONE, {}, 4096);
            has_fm = true;
        }
    }
    bool has_fm = false;
    // Feature-major form: x [I][T][S] -> out [H][T][S] (G scratch [4H][T][S]).  In the time-major layout a row of the input
    // projection is the S sequences of ONE step, so its GEMM tiles are at most S columns wide: 128 x 64 tiles at batch 64
    // (97 TFLOP/s on CRN's 4096 x 1024 projections), one useful column of 64 at batch 1 (LSTM model: 0.9 ms per layer for
    // 3.4 GFLOP).  With the features outermost the T * S (frame, sequence) pairs of a feature are ONE contiguous row and
    // the projection is a full-width GEMM whatever the batch; the cooperative recurrence reads / writes through strides.
    bool fm_ok(int S) const { return has_fm && coop_enabled() && lstm_coop_supported(H, S, 1); }
    void run_fm(const float* x, float* G, float* cell, float* out, int T, int S, hipStream_t st, Profiler* prof) const {
        const long N = (long)T * S;
        run_pointwise(gin_fm, x, 0, N, G, 0, N, 1, (int)N, st, prof);
        LstmCoopArgs a{};
        a.gx = G; a.whh = whh_dev; a.out = out; a.cell = cell;
        a.gx_z = 0; a.gx_t = S; a.gx_row = N;
        a.whh_z = 0;
        a.out_z = 0; a.out_t = S; a.out_row = N;
        a.H = H; a.T = T; a.S = S; a.Z = 1; a.reverse = 0;
        const bool timed = prof && prof->on;
        if (timed) prof->begin(st);
        launch_lstm_coop(a, st);
        if (timed) prof->end(st, 2.0 * 4 * H * (double)H * S * (T - 1));
    }
    void free() {
        gc_free_plan(gin);
        gc_free_plan(step);
        if (has_x) gc_free_plan(step_x);
        has_x = false;
        if (has_fm) gc_free_plan(gin_fm);
        has_fm = false;
        if (whh_dev) (void)hipFree(whh_dev);
        if (wih_dev) (void)hipFree(wih_dev);
        if (bih_dev) (void)hipFree(bih_dev);
        whh_dev = wih_dev = bih_dev = nullptr;
    }
    static bool coop_enabled() {
        static const int on = [] {
            const char* e = getenv("SE_LSTM_COOP");
            return e ? atoi(e) : 1;
        }();
        return on != 0;
    }
    // x [T][I][S] -> out [T][H][S];  G scratch [T][4H][S];  cell scratch [H][S]
    void run(const float* x, float* G, float* cell, float* out, int T, int S, hipStream_t st, Profiler* prof) const {
        run_strided(x, (long)I * S, G, cell, out, (long)H * S, 1, T, S, st, prof);
    }
    // general form: x rows are a slice of a wider tensor (per-step stride x_t), h_t unit j is written to row
    // j * out_rs of a tensor with per-step stride out_t (out_rs = 2 interleaves two groups, GCRN_noncprs.py:28-29)
    void run_strided(const float* x, long x_t, float* G, float* cell, float* out, long out_t, int out_rs, int T, int S,
                     hipStream_t st, Profiler* prof) const {
        run_cols(x, x_t, G, cell, out, out_t, out_rs, T, S, 0, S, st, prof);
    }
    // Streaming: T more steps of S sequences continuing from (h_state [H][S], cell [H][S]) - one fused GEMM + cell launch
    // per step (the weight-stationary cooperative kernel starts every launch from a zero state); `first` = the stream's very
    // first step (state is zero).  h_state / cell are left holding the state after the last step.
    void run_stream(const float* x, float* G, float* cell, float* h_state, float* out, int T, int S, bool first, hipStream_t st,
                    Profiler* prof) const {
        run_stream_strided(x, (long)I * S, G, cell, h_state, out, (long)H * S, 1, T, S, first, st, prof);
    }
    // general form (as run_strided): per-step input stride x_t, h_t unit j written to row j * out_rs of a tensor with
    // per-step stride out_t; h_state stays a dense [H][S] tensor
    void run_stream_strided(const float* x, long x_t, float* G, float* cell, float* h_state, float* out, long out_t, int out_rs,
                            int T, int S, bool first, hipStream_t st, Profiler* prof) const {
        run_pointwise_cols(gin, x, x_t, S, G, 4L * H * S, S, T, S, st, prof);
        for (int t = 0; t < T; ++t) {
            GCParams p = step.p;
            p.first_step = (first && t == 0);
            p.src0 = t > 0 ? out + (size_t)(t - 1) * out_t : h_state;
            p.s0_b = 0;
            p.s0_c = t > 0 ? (long)out_rs * S : S;
            p.s0_f = 0;
            p.src1 = nullptr;
            p.Fin = 1;
            p.Tin = S;
            p.B = 1;
            p.Q = 1;
            p.Tout = S;
            p.aux = G + (size_t)t * 4 * H * S;
            p.x_b = 0;
            p.x_c = S;
            p.x_f = 0;
            p.dst = out + (size_t)t * out_t;
            p.d_b = 0;
            p.d_c = (long)out_rs * S;
            p.d_f = 0;
            p.cell = cell;
            gc_launch_prof(step, p, st, prof);
        }
        SE_HIP(hipMemcpy2DAsync(h_state, (size_t)S * sizeof(float), out + (size_t)(T - 1) * out_t,
                                (size_t)out_rs * S * sizeof(float), (size_t)S * sizeof(float), H, hipMemcpyDeviceToDevice, st));
    }
    // the same on the sequence columns [c0, c0 + Sn) of tensors whose rows hold S sequences: sequences are independent,
    // so disjoint column ranges can run concurrently on different streams (FullSubNet's 257 * B sub-band sequences)
    void run_cols(const float* x, long x_t, float* G, float* cell, float* out, long out_t, int out_rs, int T, int S, int c0,
                  int Sn, hipStream_t st, Profiler* prof) const {
        run_pointwise_cols(gin, x + c0, x_t, S, G + c0, 4L * H * S, S, T, Sn, st, prof);
        if (c0 == 0 && Sn == S && whh_dev && coop_enabled() && lstm_coop_supported(H, S, 1)) {
            LstmCoopArgs a{};
            a.gx = G; a.whh = whh_dev; a.out = out; a.cell = cell;
            a.gx_z = 0; a.gx_t = 4L * H * S; a.gx_row = S;
            a.whh_z = 0;
            a.out_z = 0; a.out_t = out_t; a.out_row = (long)out_rs * S;
            a.H = H; a.T = T; a.S = S; a.Z = 1; a.reverse = 0;
            const bool timed = prof && prof->on;
            if (timed) prof->begin(st);
            launch_lstm_coop(a, st);
            if (timed) prof->end(st, 2.0 * 4 * H * (double)H * S * (T - 1));
            return;
        }
        for (int t = 0; t < T; ++t) {
            GCParams p = step.p;
            p.first_step = (t == 0);
            p.src0 = (t > 0 ? out + (size_t)(t - 1) * out_t : out) + c0;
            p.s0_b = 0;
            p.s0_c = (long)out_rs * S;
            p.s0_f = 0;
            p.src1 = nullptr;
            p.Fin = 1;
            p.Tin = Sn;
            p.B = 1;
            p.Q = 1;
            p.Tout = Sn;
            p.aux = G + (size_t)t * 4 * H * S + c0;
            p.x_b = 0;
            p.x_c = S;
            p.x_f = 0;
            p.dst = out + (size_t)t * out_t + c0;
            p.d_b = 0;
            p.d_c = (long)out_rs * S;
            p.d_f = 0;
            p.cell = cell + c0;
            gc_launch_prof(step, p, st, prof);
        }
    }
    // the same with the input projection inside the step GEMM (step_x): no gate tensor; hz = zeros [H][S] standing in for h_{-1}
    void run_cols_x(const float* x, long x_t, float* cell, const float* hz, float* out, long out_t, int out_rs, int T, int S, int c0,
                    int Sn, hipStream_t st, Profiler* prof) const {
        SE_CHECK(has_x, "run_cols_x: layer was not built with a fused input projection");
        for (int t = 0; t < T; ++t) {
            GCParams p = step_x.p;
            p.first_step = (t == 0);
            p.src0 = (t > 0 ? out + (size_t)(t - 1) * out_t : hz) + c0;
            p.s0_b = 0;
            p.s0_c = t > 0 ? (long)out_rs * S : S;
            p.s0_f = 0;
            p.src1 = x + (size_t)t * x_t + c0;
            p.s1_b = 0;
            p.s1_c = S;
            p.s1_f = 0;
            p.Fin = 1;
            p.Tin = Sn;
            p.B = 1;
            p.Q = 1;
            p.Tout = Sn;
            p.aux = nullptr;
            p.dst = out + (size_t)t * out_t + c0;
            p.d_b = 0;
            p.d_c = (long)out_rs * S;
            p.d_f = 0;
            p.cell = cell + c0;
            gc_launch_prof(step_x, p, st, prof);
        }
    }
};

// A stack of 2 / 3 equal-width layers on ONE sequence, feature-major (x [I][T] -> out [H][T] of the last layer): the first
// layer's input projection as one GEMM, then every layer's recurrence AND the upper layers' input projections in one
// cooperative launch, layer l one frame behind layer l - 1 (k_lstm_coop.hip: lstm_stack_kernel).  false: not applicable.
inline bool lstm_stack_fm(const LstmBig* const* ly, int L, const float* x, float* G, float* out, int T, hipStream_t st,
                          Profiler* prof) {
    if (L < 2 || L > 3 || !LstmBig::coop_enabled() || !lstm_stack_supported(ly[0]->H, L)) return false;
    for (int l = 0; l < L; ++l)
        if (!ly[l]->has_fm || !ly[l]->whh_dev || ly[l]->H != ly[0]->H || (l > 0 && (!ly[l]->wih_dev || ly[l]->I != ly[0]->H))) return false;
    const int H = ly[0]->H;
    run_pointwise(ly[0]->gin_fm, x, 0, T, G, 0, T, 1, T, st, prof);
    LstmStackArgs a{};
    a.gx0 = G; a.gx_t = 1; a.gx_row = T;
    for (int l = 0; l < L; ++l) {
        a.whh[l] = ly[l]->whh_dev;
        a.wih[l] = ly[l]->wih_dev;
        a.bias[l] = ly[l]->bih_dev;
    }
    a.out = out; a.out_t = 1; a.out_row = T;
    a.H = H; a.T = T; a.L = L;
    const bool timed = prof && prof->on;
    if (timed) prof->begin(st);
    launch_lstm_stack(a, st);
    if (timed) prof->end(st, 2.0 * 4 * H * (double)H * ((double)L * (T - 1) + (double)(L - 1) * T));
    return true;
}

// A stack of L equal-width layers over S sequences, feature-major (x [I][T][S], outs[l] [H][T][S], G [4H][T][S], cells
// [L][H][S]), as a PIPELINE OF LAYERS over chunks of steps: launch s runs layer l on chunk s - l, all in one cooperative launch
// (k_lstm_coop.hip: lstm_coop16_kernel, a.pz) - where one layer alone leaves a workgroup a single 16-sequence tile per step (batch
// <= 64 at H = 1024) and nothing to overlap its h exchange with, L layers give it L tiles: CRN's two layers at batch 64 take
// 10.2 us per step TOGETHER instead of 2 x 8.3.  The input projection of layer l >= 1 on a chunk (a GEMM over out[l - 1]) runs
// between the launches and writes its gates over the chunk of G the layer below has just consumed - one G serves the stack.
// outs[l] may alias outs[l - 2] (the projection that read that chunk has run).  false: not applicable (the caller runs the
// layers one after the other).
inline bool lstm_stack_chunked_fm(const LstmBig* const* ly, int L, const float* x, float* G, float* cells, float* const* outs, int T, int S,
                                  hipStream_t st, Profiler* prof) {
    if (L < 2 || L > 4 || !LstmBig::coop_enabled()) return false;
    const int H = ly[0]->H;
    for (int l = 0; l < L; ++l)
        if (!ly[l]->has_fm || !ly[l]->whh_dev || ly[l]->H != H || (l > 0 && ly[l]->I != H)) return false;
    if (!lstm_coop_chunk_supported(H, S, L)) return false;
    // the chunk kernel addresses G / out with 32-bit lane offsets over rows of T * S elements (launch_lstm_coop_chunk checks the same
    // bound): batches of long clips (H = 1024: T * S >= ~244 k, e.g. 64 clips of 3 800 frames) go back to the per-layer path, whose
    // launchers fall back by themselves
    if ((double)T * S * 4 * H * 4 >= 4.0e9 || (double)T * S * H >= 4.0e9) return false;
    static const int chunk_env = getenv("SE_LSTM_CHUNK_T") ? atoi(getenv("SE_LSTM_CHUNK_T")) : 0;
    // chunk length: the pipeline runs (L - 1) chunks longer than the sequence, every launch costs ~3 steps' worth of launch + fill
    int Tc = chunk_env > 0 ? chunk_env : 48;                 // (measured at batch 64, T = 401: 24 / 36 / 48 / 64 -> CRN 4 907 / 5 022 / 5 281 / 5 204 utt/s)
    Tc = std::max(2, std::min(Tc, T)) & ~1;                  // even: a chunk keeps the parity of the exchange slabs
    if (T < 2 * Tc) return false;
    const long N = (long)T * S;
    const int nc = (T + Tc - 1) / Tc;
    run_pointwise(ly[0]->gin_fm, x, 0, N, G, 0, N, 1, (int)N, st, prof);
    for (int s = 0; s < nc + L - 1; ++s) {
        LstmCoopArgs a{};
        a.cell = cells;
        a.gx_t = S; a.gx_row = N; a.out_t = S; a.out_row = N;
        a.H = H; a.S = S; a.reverse = 0;
        double flops = 0;
        for (int l = 0; l < L; ++l) {
            const int c = s - l;
            if (c < 0 || c >= nc) continue;
            const int t0 = c * Tc, len = std::min(Tc, T - t0);
            if (l > 0) {       // gates of layer l on this chunk, over the chunk of G layer l - 1 consumed one launch ago
                const long off = (long)t0 * S;
                run_pointwise(ly[l]->gin_fm, outs[l - 1] + off, 0, N, G + off, 0, N, 1, len * S, st, prof);
            }
            const int z = a.Z++;
            a.gxp[z] = G + (long)t0 * S;
            a.whhp[z] = ly[l]->whh_dev;
            a.outp[z] = outs[l] + (long)t0 * S;
            a.lz[z] = l;
            a.t0[z] = t0;
            a.Tz[z] = len;
            a.T = std::max(a.T, len);
            flops += 2.0 * 4 * H * (double)H * S * (len - (t0 == 0 ? 1 : 0));
        }
        const bool timed = prof && prof->on;
        if (timed) prof->begin(st);
        launch_lstm_coop_chunk(a, L, st);
        if (timed) prof->end(st, flops);
    }
    return true;
}

// Two independent LSTM layers of equal shape (GCRN's grouped LSTM, GCRN/GCRN_noncprs.py:5-39) as ONE cooperative launch
// (Z = 2): each alone covers H/16 x SS workgroups of the chip, together they fill it.  whh2 = [2][4H][H] device copy of
// both recurrent matrices; G / cell hold both groups back to back.  Falls back to two sequential layers.
inline void run_lstm_pair(const LstmBig& l0, const LstmBig& l1, const float* whh2, const float* x0, const float* x1, long x_t,
                          float* G, float* cell, float* out0, long out_z, long out_t, int out_rs, int T, int S,
                          hipStream_t st, Profiler* prof) {
    const int H = l0.H;
    const long gz = (long)T * 4 * H * S;
    if (!(whh2 && LstmBig::coop_enabled() && lstm_coop_supported(H, S, 2))) {
        l0.run_strided(x0, x_t, G, cell, out0, out_t, out_rs, T, S, st, prof);
        l1.run_strided(x1, x_t, G, cell, out0 + out_z, out_t, out_rs, T, S, st, prof);
        return;
    }
    run_pointwise(l0.gin, x0, x_t, S, G, 4L * H * S, S, T, S, st, prof);
    run_pointwise(l1.gin, x1, x_t, S, G + gz, 4L * H * S, S, T, S, st, prof);
    LstmCoopArgs a{};
    a.gx = G; a.whh = whh2; a.out = out0; a.cell = cell;
    a.gx_z = gz; a.gx_t = 4L * H * S; a.gx_row = S;
    a.whh_z = 4L * H * H;
    a.out_z = out_z; a.out_t = out_t; a.out_row = (long)out_rs * S;
    a.H = H; a.T = T; a.S = S; a.Z = 2; a.reverse = 0;
    const bool timed = prof && prof->on;
    if (timed) prof->begin(st);
    launch_lstm_coop(a, st);
    if (timed) prof->end(st, 2.0 * 2 * 4 * H * (double)H * S * (T - 1));
}

}  // namespace se
