// DPCRN on the MI355X engine (the only model whose real checkpoints ship with the reference).
//
// Reference: DPCRN/DPCRN.py:16-174 (dpcrn: conv encoder x5 -> DPRNN applied twice with shared weights -> deconv
// decoder x5 -> complex ratio mask on the input), decode loop DPCRN/dpcrn_decode_vb.py:33-60.
//
// Engine mapping ([B][C][F][T] activations, T contiguous):
//   * convs / deconvs: tap-table implicit GEMMs, BatchNorm folded, scalar PReLU in the epilogue, two-source skips;
//   * intra-frame BiLSTM (over the 4 frequency positions, DPCRN.py:66-70): 1x1-conv input projections for both
//     directions at once, then the persistent register-resident LSTM kernel with steps = F and sequences = frames
//     (contiguous t), both directions in one launch (blockIdx.y);
//   * inter-frame LSTM (over T, :77-82): transposed to time-major [T][128][B*4], same persistent kernel;
//   * LayerNorm([4,128]) + residual fused in one kernel; complex mask + decode-script decompress fused.
#include "rnn.h"
#include "k_lstm_short.h"

namespace se {

namespace {

constexpr int NFFT = 320, HOP = 160, NBIN = 161, CH = 128, NF = 4;

class Dpcrn final : public Model {
  public:
    explicit Dpcrn(EngineCtx& c) : Model(c) {}
    ~Dpcrn() override {
        for (auto& p : enc) gc_free_plan(p);
        for (auto& p : dec) free_deconv_plan(p);
        for (auto& p : intra_in) gc_free_plan(p);
        for (auto& p : inter_in) gc_free_plan(p);
        gc_free_plan(intra_fc);
        gc_free_plan(inter_fc);
        for (float* d : {intra_whh[0], intra_whh[1], inter_whh[0], inter_whh[1], ln_w[0], ln_b[0], ln_w[1], ln_b[1], intra_wih[0],
                         intra_wih[1], intra_bias[0], intra_bias[1]})
            if (d) (void)hipFree(d);
    }
    StftGeom default_geom() const override { return StftGeom{NFFT, HOP, NFFT}; }

    void finalize(const TrackedSD& sd) override {
        const int EC[6] = {2, 32, 32, 32, 64, 128};
        for (int i = 0; i < 5; ++i) {   // DPCRN.py:94-130
            const std::string p = "en.en_module." + std::to_string(i) + ".";
            DenseW w = conv_weights(sd.get(p + "1.weight", {EC[i + 1], EC[i], 2, 3}), &sd.get(p + "1.bias", {EC[i + 1]}), true);
            fold_bn(w, sd.get(p + "2.weight"), sd.get(p + "2.bias"), sd.get(p + "2.running_mean"), sd.get(p + "2.running_var"));
            enc[i] = make_conv_plan(w, 2, 0, 1, 1, 1, ACT_PRELU, prelu_slopes(sd.get(p + "3.weight"), EC[i + 1]), EPI_ACT, 401);
        }
        // ---- DPRNN (DPCRN.py:44-57)
        for (int l = 0; l < 2; ++l) {
            LstmW f = load_lstm(sd, "dprnn.intra_rnn.", l, "", CH, 64);
            LstmW r = load_lstm(sd, "dprnn.intra_rnn.", l, "_reverse", CH, 64);
            DenseW both = concat_rows(f.wih, r.wih);                       // [2*256][128]: fwd gates, bwd gates
            intra_in[l] = make_pointwise_plan(both, ACT_NONE, {}, 401);
            std::vector<float> w = f.whh.w;
            w.insert(w.end(), r.whh.w.begin(), r.whh.w.end());             // [2][256][64]
            intra_whh[l] = to_device(w);
            intra_wih[l] = to_device(both.w);                              // [2][256][128] and [2][256]: the fused short-sequence kernel
            intra_bias[l] = to_device(both.bias);
            LstmW t = load_lstm(sd, "dprnn.inter_rnn.", l, "", CH, CH);
            inter_in[l] = make_pointwise_plan(t.wih, ACT_NONE, {}, ctx.max_batch * NF);
            inter_whh[l] = to_device(t.whh.w);
        }
        intra_fc = make_pointwise_plan(linear_weights(sd.get("dprnn.intra_fc.weight", {CH, CH}), &sd.get("dprnn.intra_fc.bias", {CH})),
                                       ACT_NONE, {}, 401);
        inter_fc = make_pointwise_plan(linear_weights(sd.get("dprnn.inter_fc.weight", {CH, CH}), &sd.get("dprnn.inter_fc.bias", {CH})),
                                       ACT_NONE, {}, ctx.max_batch * NF);
        for (int k = 0; k < 2; ++k) {
            const std::string n = k == 0 ? "dprnn.ln1." : "dprnn.ln2.";
            ln_w[k] = to_device(sd.get(n + "weight", {NF, CH}).data);
            ln_b[k] = to_device(sd.get(n + "bias", {NF, CH}).data);
        }
        // ---- decoder (DPCRN.py:132-166)
        const int DC[5][2] = {{256, 64}, {128, 32}, {64, 32}, {64, 32}, {64, 2}};
        for (int i = 0; i < 5; ++i) {
            const std::string p = "de.de_module." + std::to_string(i) + ".";
            DenseW w = deconv_weights(sd.get(p + "0.weight", {DC[i][0], DC[i][1], 2, 3}), &sd.get(p + "0.bias", {DC[i][1]}), true);
            DenseW wz = w;
            wz.bias.assign(wz.M, 0.f);
            std::vector<float> slope;
            int act = ACT_NONE;
            if (i < 4) {
                const int o = (i == 3) ? 3 : 2;
                const std::string bn = p + std::to_string(o) + ".";
                fold_bn(w, sd.get(bn + "weight"), sd.get(bn + "bias"), sd.get(bn + "running_mean"), sd.get(bn + "running_var"));
                fold_bn(wz, sd.get(bn + "weight"), sd.get(bn + "bias"), sd.get(bn + "running_mean"), sd.get(bn + "running_var"));
                slope = prelu_slopes(sd.get(p + std::to_string(o + 1) + ".weight"), DC[i][1]);
                act = ACT_PRELU;
            }
            dec[i] = make_deconv_plan(w, 2, i == 3 ? -1 : 0, 0, act, slope, 401, DC[i][0] / 2, &wz.bias);
        }
    }

    void plan_buffers(int B, int T) override {
        cur.B = 0;
        bufs(B, T);
    }

    void forward(const float* in, const int64_t* shape, int ndim, float* out, hipStream_t st) override {
        SE_CHECK(ndim == 4 && shape[1] == 2 && shape[3] == NBIN, "DPCRN forward expects [B,2,T,161]");
        const int B = (int)shape[0], T = (int)shape[2];
        Bufs& b = bufs(B, T);
        // [B*2][T][F] -> [B*2][F][T]
        launch_transpose_akt(in, b.spec, T, 2 * B, NBIN, NBIN, (long)T * NBIN, T, (long)NBIN * T, st);
        network(b, st);
        launch_cmask_apply(b.D[5], b.spec, b.est, B, NBIN, T, 1.f, st);                            // DPCRN.py:33-42
        launch_transpose_akt(b.est, out, NBIN, 2 * B, T, T, (long)NBIN * T, NBIN, (long)T * NBIN, st);
    }

    // (causal end to end - eval BatchNorm folded, the intra-frame BiLSTM and both LayerNorms work inside one frame - so an
    // equal-length batch runs with its rows zero-extended to whole 128 B lines, model.h causal_work_frames; the inter-frame LSTMs
    // walk the clip's own frames: Bufs::Tl)
    int frame_multiple() const override { return causal_frame_multiple(true); }
    void enhance(const float* wav, long pitch, int B, int L, float* out, long out_pitch, hipStream_t st) override {
        const int T = 1 + L / HOP;
        const int Tw = causal_work_frames(T, true);
        const bool rag = ragged_ctx() != nullptr;
        const int Ts = rag ? Tw : T;          // frames the STFT / iSTFT walk (ragged rows: zeros behind a row's own last frame)
        Bufs& b = bufs(B, Tw);
        b.Tl = T;
        launch_rms_scale(wav, B, L, pitch, b.c, st);                                               // dpcrn_decode_vb.py:34-35
        if (Tw != T && !rag) SE_HIP(hipMemsetAsync(b.spec, 0, (size_t)B * 2 * NBIN * Tw * sizeof(float), st));
        launch_stft(ctx.geom, wav, pitch, B, L, L, b.c, ctx.p_in, b.spec, nullptr, Ts, Tw, st);     // :37-45
        network(b, st);                                                                            // :47
        launch_cmask_apply(b.D[5], b.spec, b.est, B, NBIN, Tw, ctx.p_out, st);                     // model :33-42 + :48-57
        launch_istft(ctx.geom, b.est, B, Ts, Tw, b.frames, b.c, out, out_pitch, L, st);             // :58-60
        b.Tl = 0;
    }

    // ---- frame-online mode (model.h): (de)convs look back one frame (DPCRN.py:94-166, same pad / chomp scheme as CRN), the
    // intra-frame BiLSTM runs over frequency inside one frame, LayerNorm is per frame; the inter-frame LSTM (2 layers,
    // applied in both DPRNN passes with shared weights, :28-29) carries (h, c) - four states
    bool stream_supported() const override { return true; }
    void stream_begin(int B, int max_chunk, hipStream_t st) override {
        ss.release();
        ss.B = B;
        ss.first = true;
        for (long rows : stream_rows()) ss.hist.push_back(ss.zeros((size_t)B * rows * STREAM_HC, st));
        for (int l = 0; l < 4; ++l) {
            ss.h[l] = ss.zeros((size_t)CH * NF * B, st);
            ss.c[l] = ss.zeros((size_t)CH * NF * B, st);
        }
        (void)max_chunk;
    }
    void stream_bufs(int B, int n, float** spec, float** mag, float** est) override {
        Bufs& b = bufs(B, STREAM_HC + n);
        *spec = b.spec;
        *mag = nullptr;
        *est = b.est;
    }
    void stream_chunk(int B, int t0, int n, hipStream_t st, bool last) override {
        (void)last;
        SE_CHECK(ss.B == B && !ss.hist.empty(), "stream_chunk without stream_begin");
        const int HC = STREAM_HC, Tw = HC + n;
        Bufs& b = bufs(B, Tw);
        Profiler* pf = &ctx.prof;
        const std::vector<long> rows = stream_rows();
        float* tens[13] = {b.spec, b.E[0], b.E[1], b.E[2], b.E[3], b.E[4], b.P1, b.D[0], b.D[1], b.D[2], b.D[3], b.D[4], b.D[5]};
        auto restore = [&](int k) { launch_hist_restore(tens[k], ss.hist[k], B, rows[k], Tw, HC, st); };
        // the (de)convs only produce the new frames: the history columns of their outputs come from the state in one launch
        HistBatch hb, hb_all;
        for (int k = 0; k < 13; ++k) {
            hb_all.add(tens[k], ss.hist[k], rows[k]);
            if (k != 6 && k != 7) hb.add(tens[k], ss.hist[k], rows[k]);
        }
        launch_hist_batch(hb, B, Tw, HC, false, st);
        const int EC[5] = {32, 32, 32, 64, 128}, EF[5] = {80, 39, 19, 9, 4};
        Act4 x = act4(b.spec, 2, NBIN, Tw);
        for (int i = 0; i < 5; ++i) {
            run_conv(enc[i], x, nullptr, b.E[i], EC[i], EF[i], B, Tw, Tw, st, pf, nullptr, HC);
            x = act4(b.E[i], EC[i], EF[i], Tw);
        }
        dprnn(b, b.E[4], b.P1, st, n, 0);
        restore(6);                              // the history columns of a DPRNN output saw no inter-frame LSTM
        dprnn(b, b.P1, b.D[0], st, n, 1);
        restore(7);
        const int DCo[5] = {64, 32, 32, 32, 2}, DF[5] = {9, 19, 39, 80, 161};
        int cin = CH, fin = NF;
        for (int i = 0; i < 5; ++i) {
            Act4 a0 = act4(b.D[i], cin, fin, Tw);
            Act4 a1 = act4(b.E[4 - i], cin, fin, Tw);
            run_deconv(dec[i], a0, &a1, b.D[i + 1], DCo[i], DF[i], B, Tw, Tw, st, pf, nullptr, HC);
            cin = DCo[i];
            fin = DF[i];
        }
        launch_cmask_apply(b.D[5], b.spec, b.est, B, NBIN, Tw, ctx.p_out, st);
        launch_hist_batch(hb_all, B, Tw, HC, true, st);
        ss.first = false;
        (void)t0;
    }

  private:
    StreamState ss;
    static std::vector<long> stream_rows() {      // rows (C * F) of spec, E[0..4], P1, D[0..5]
        return {2L * NBIN, 32L * 80, 32L * 39, 32L * 19, 64L * 9, 128L * 4, (long)CH * NF, (long)CH * NF, 64L * 9, 32L * 19, 32L * 39,
                32L * 80, 2L * 161};
    }
    struct Bufs {
        int B = 0, T = 0;
        int Tl = 0;      // > 0: the clip's own frame count when the rows are zero-extended to T (offline equal-length batches)
        float *c, *spec, *est, *frames, *E[5], *D[6];
        float *Gi, *Hi[2], *Y, *R1, *Xt, *Gt, *Ht[2], *Yt, *R2, *P1;
    } cur;
    GCPlan enc[5], intra_in[2], inter_in[2], intra_fc, inter_fc;
    DeconvPlan dec[5];
    float *intra_whh[2] = {nullptr, nullptr}, *inter_whh[2] = {nullptr, nullptr};
    float *intra_wih[2] = {nullptr, nullptr}, *intra_bias[2] = {nullptr, nullptr};
    float *ln_w[2] = {nullptr, nullptr}, *ln_b[2] = {nullptr, nullptr};

    Bufs& bufs(int B, int T) {
        if (cur.B == B && cur.T == T) return cur;
        Arena& a = ctx.arena;
        a.reset();
        Bufs b;
        b.B = B;
        b.T = T;
        const size_t BT = (size_t)B * T;
        b.c = a.alloc_f(B);
        b.spec = a.alloc_f(BT * 2 * NBIN);
        b.est = a.alloc_f(BT * 2 * NBIN);
        b.frames = nullptr;      // the fused iSTFT keeps its frames in LDS (k_stft.hip); kept in the struct for the launcher signature
        const int EC[5] = {32, 32, 32, 64, 128}, EF[5] = {80, 39, 19, 9, 4};
        for (int i = 0; i < 5; ++i) b.E[i] = a.alloc_f(BT * EC[i] * EF[i]);
        const int DCo[5] = {64, 32, 32, 32, 2}, DF[5] = {9, 19, 39, 80, 161};
        b.D[0] = a.alloc_f(BT * CH * NF);
        for (int i = 0; i < 5; ++i) b.D[i + 1] = a.alloc_f(BT * DCo[i] * DF[i]);
        const size_t act = BT * CH * NF;
        b.Gi = a.alloc_f(act * 4);       // [B][512][4][T]
        b.Hi[0] = a.alloc_f(act);
        b.Hi[1] = a.alloc_f(act);
        b.Y = a.alloc_f(act);
        b.R1 = a.alloc_f(act);
        b.Xt = a.alloc_f(act);           // [T][128][4B]
        b.Gt = a.alloc_f(act * 4);       // [T][512][4B]
        b.Ht[0] = a.alloc_f(act);
        b.Ht[1] = a.alloc_f(act);
        b.Yt = a.alloc_f(act);
        b.R2 = a.alloc_f(act);
        b.P1 = a.alloc_f(act);
        cur = b;
        return cur;
    }

    // DPRNN.forward (DPCRN.py:59-92): x [B][128][4][T] -> out [B][128][4][T]
    // n_stream > 0: frame-online chunk - only the last n_stream columns are new frames, the inter-frame LSTMs of DPRNN pass
    // `pass` continue from their carried state
    void dprnn(Bufs& b, const float* x, float* out, hipStream_t st, int n_stream = 0, int pass = 0) {
        const int B = b.B, T = b.T;
        // inter-frame steps and their first column (frame-online: the last n_stream columns; zero-extended rows: the clip's own frames)
        const int nT = n_stream > 0 ? n_stream : (b.Tl > 0 ? b.Tl : T), c0 = n_stream > 0 ? T - nT : 0;
        Profiler* pf = &ctx.prof;
        const long plane = (long)NF * T;            // one channel
        // ---- intra: BiLSTM(128 -> 64 x2, 2 layers) over F for every (b, t)
        const float* lin = x;
        // many short sequences: input projection and recurrence in one kernel, no gate tensor (k_lstm_short.hip; SE_LSTM_SHORT=0:
        // the projection GEMM + the persistent kernel below).  Frame-online windows of a few frames keep the two-launch form
        // (measured at 1 / 16 streams x 1 / 8 frames: 2-3 % faster per push there; one whole clip: the same either way).
        static const long short_min = getenv("SE_LSTM_SHORT_MIN") ? atol(getenv("SE_LSTM_SHORT_MIN")) : 256;      // (utterance, frame) pairs
        const bool fused_intra = lstm_short_supported(64, CH, NF) && (long)B * T >= short_min;
        for (int l = 0; l < 2 && fused_intra; ++l) {
            LstmShortArgs a{};
            a.x = lin; a.x_o = (long)CH * plane; a.x_c = plane; a.x_t = T;
            a.wih = intra_wih[l]; a.whh = intra_whh[l]; a.bias = intra_bias[l];
            a.wih_z = 256L * CH; a.whh_z = 256L * 64; a.bias_z = 256;
            a.out = b.Hi[l]; a.out_o = (long)CH * plane; a.out_z = 64L * plane; a.out_t = T; a.out_row = plane;
            a.T = NF; a.S = T; a.Z = 2; a.O = B; a.reverse = 2;
            const bool timed = pf && pf->on;
            if (timed) pf->begin(st);
            launch_lstm_short(a, st);
            if (timed) pf->end(st, 2.0 * 2 * 256 * (double)(CH + 64) * NF * B * T);
            lin = b.Hi[l];
        }
        for (int l = 0; l < 2 && !fused_intra; ++l) {
            // Gi[b][dir*256 + row][f][t]
            run_pointwise(intra_in[l], lin, (long)CH * plane, plane, b.Gi, 512L * plane, plane, B, (int)plane, st, pf);
            LstmPersistArgs a{};
            a.gx = b.Gi; a.whh = intra_whh[l]; a.out = b.Hi[l];
            a.gx_o = 512L * plane; a.gx_z = 256L * plane; a.gx_t = T; a.gx_row = plane;
            a.whh_z = 256L * 64;
            a.out_o = (long)CH * plane; a.out_z = 64L * plane; a.out_t = T; a.out_row = plane;
            a.H = 64; a.T = NF; a.S = T; a.Z = 2; a.O = B; a.reverse = 2;      // z = 1 is the reverse direction
            launch_lstm_persist(a, st);
            lin = b.Hi[l];
        }
        run_pointwise(intra_fc, b.Hi[1], (long)CH * plane, plane, b.Y, (long)CH * plane, plane, B, (int)plane, st, pf);
        launch_layernorm_cf(b.Y, x, ln_w[0], ln_b[0], b.R1, B, CH, NF, T, 1e-5f, st);              // :73-74
        // ---- inter: LSTM(128 -> 128, 2 layers) over T for every (b, f);  time-major, s = f*B + b
        const int S = NF * B;
        for (int f = 0; f < NF; ++f)
            launch_transpose_akt(b.R1 + (size_t)f * T + c0, b.Xt + (size_t)f * B, B, CH, nT, (long)CH * plane, plane,
                                 (long)CH * S, S, st);
        const float* tin = b.Xt;
        for (int l = 0; l < 2; ++l) {
            run_pointwise(inter_in[l], tin, (long)CH * S, S, b.Gt, 512L * S, S, nT, S, st, pf);
            LstmPersistArgs a{};
            a.gx = b.Gt; a.whh = inter_whh[l]; a.out = b.Ht[l];
            a.gx_o = 0; a.gx_z = 0; a.gx_t = 512L * S; a.gx_row = S;
            a.whh_z = 0;
            a.out_o = 0; a.out_z = 0; a.out_t = (long)CH * S; a.out_row = S;
            a.H = CH; a.T = nT; a.S = S; a.Z = 1; a.O = 1; a.reverse = 0;
            if (n_stream > 0) {
                a.st_h = ss.h[2 * pass + l];
                a.st_c = ss.c[2 * pass + l];
            }
            launch_lstm_persist(a, st);
            tin = b.Ht[l];
        }
        run_pointwise(inter_fc, b.Ht[1], (long)CH * S, S, b.Yt, (long)CH * S, S, nT, S, st, pf);
        for (int f = 0; f < NF; ++f)
            launch_transpose_akt(b.Yt + (size_t)f * B, b.R2 + (size_t)f * T + c0, nT, CH, B, (long)CH * S, S, (long)CH * plane,
                                 plane, st);
        launch_layernorm_cf(b.R2, b.R1, ln_w[1], ln_b[1], out, B, CH, NF, T, 1e-5f, st);           // :87-88
    }

    // b.spec [B][2][161][T] -> mask b.D[5] [B][2][161][T]
    void network(Bufs& b, hipStream_t st) {
        const int B = b.B, T = b.T;
        Profiler* pf = &ctx.prof;
        const int EC[5] = {32, 32, 32, 64, 128}, EF[5] = {80, 39, 19, 9, 4};
        Act4 x = act4(b.spec, 2, NBIN, T);
        for (int i = 0; i < 5; ++i) {
            run_conv(enc[i], x, nullptr, b.E[i], EC[i], EF[i], B, T, T, st, pf);
            x = act4(b.E[i], EC[i], EF[i], T);
        }
        dprnn(b, b.E[4], b.P1, st);
        dprnn(b, b.P1, b.D[0], st);            // second application with the same weights (DPCRN.py:28-29)
        const int DCo[5] = {64, 32, 32, 32, 2}, DF[5] = {9, 19, 39, 80, 161};
        int cin = CH, fin = NF;
        for (int i = 0; i < 5; ++i) {
            Act4 a0 = act4(b.D[i], cin, fin, T);
            Act4 a1 = act4(b.E[4 - i], cin, fin, T);
            run_deconv(dec[i], a0, &a1, b.D[i + 1], DCo[i], DF[i], B, T, T, st, pf);
            cin = DCo[i];
            fin = DF[i];
        }
    }
};

}  // namespace

std::unique_ptr<Model> make_dpcrn(EngineCtx& ctx) { return std::unique_ptr<Model>(new Dpcrn(ctx)); }

}  // namespace se
