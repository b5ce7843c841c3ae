// GCRN on the MI355X engine.
//
// Reference: GCRN/GCRN_noncprs.py:86-165 (Net: GLU conv encoder x5 -> grouped LSTM (2 x 512, two layers, LayerNorm)
// -> two GLU deconv decoders (real, imag) -> Linear(161,161) per branch), decode loop GCRN/gcrn_decode_vb.py:34-58
// (checked in with the compressed exponents 0.5 / 2.0).  Complex spectral MAPPING: the output is the estimate.
//
// Engine mapping: a GLU layer is ONE tap-table GEMM whose rows are (value, gate) pairs, the gate product, the
// eval-BatchNorm that follows it and the ELU run in the epilogue; the reference's `elu(cat(y, skip))` re-applies
// ELU to the skip tensors, so elu(e_k) is materialised once and shared by both decoders; the grouped LSTM runs
// time-major with one fused GEMM + cell launch per step and group, group outputs interleaved by row stride.
#include "rnn.h"

namespace se {

namespace {

constexpr int NFFT = 320, HOP = 160, NBIN = 161;
constexpr int EC[6] = {2, 16, 32, 64, 128, 256}, EF[5] = {80, 39, 19, 9, 4};

class Gcrn final : public Model {
  public:
    explicit Gcrn(EngineCtx& c) : Model(c) {}
    ~Gcrn() override {
        for (auto& p : enc) gc_free_plan(p);
        for (auto& br : dec)
            for (auto& p : br) free_deconv_plan(p);
        for (auto& l : l1) l.free();
        for (auto& l : l2) l.free();
        for (float*& w : whh_pair) {
            if (w) (void)hipFree(w);
            w = nullptr;
        }
        gc_free_plan(fc[0]);
        gc_free_plan(fc[1]);
        for (float* d : {ln_w[0], ln_b[0], ln_w[1], ln_b[1]})
            if (d) (void)hipFree(d);
    }
    StftGeom default_geom() const override { return StftGeom{NFFT, HOP, NFFT}; }

    void finalize(const TrackedSD& sd) override {
        auto glu_w = [&](const std::string& p, bool deconv, std::vector<int64_t> shape) {
            DenseW a = deconv ? deconv_weights(sd.get(p + "conv1.weight", shape), &sd.get(p + "conv1.bias"), true)
                              : conv_weights(sd.get(p + "conv1.weight", shape), &sd.get(p + "conv1.bias"), true);
            DenseW g = deconv ? deconv_weights(sd.get(p + "conv2.weight", shape), &sd.get(p + "conv2.bias"), true)
                              : conv_weights(sd.get(p + "conv2.weight", shape), &sd.get(p + "conv2.bias"), true);
            return interleave_rows(a, g);
        };
        auto post = [&](GCPlan& pl, const std::string& bn) {
            set_post_bn(pl, sd.get(bn + "weight"), sd.get(bn + "bias"), sd.get(bn + "running_mean"), sd.get(bn + "running_var"));
        };
        for (int k = 0; k < 5; ++k) {   // GCRN_noncprs.py:90-94,137-141: GluConv2d((1,3),(1,2)) -> BN -> ELU
            DenseW w = glu_w("conv" + std::to_string(k + 1) + ".", false, {EC[k + 1], EC[k], 1, 3});
            enc[k] = make_conv_plan(w, 2, 0, 0, 1, 1, ACT_ELU, {}, EPI_GLU, 401);
            post(enc[k], "bn" + std::to_string(k + 1) + ".");
        }
        for (int i = 0; i < 2; ++i) {   // GLSTM :5-39
            l1[i].build(load_lstm(sd, "glstm.lstm_list1." + std::to_string(i) + ".", 0, "", 512, 512), ctx.max_batch);
            l2[i].build(load_lstm(sd, "glstm.lstm_list2." + std::to_string(i) + ".", 0, "", 512, 512), ctx.max_batch);
            const std::string n = i == 0 ? "glstm.ln1." : "glstm.ln2.";
            ln_w[i] = to_device(sd.get(n + "weight", {1024}).data);
            ln_b[i] = to_device(sd.get(n + "bias", {1024}).data);
        }
        for (int L = 0; L < 2; ++L) {   // both groups' recurrent matrices back to back for the paired cooperative launch
            const LstmBig* pr = L == 0 ? l1 : l2;
            if (!pr[0].whh_dev || !pr[1].whh_dev) continue;
            const size_t n = (size_t)4 * 512 * 512;
            SE_HIP(hipMalloc(&whh_pair[L], 2 * n * sizeof(float)));
            SE_HIP(hipMemcpy(whh_pair[L], pr[0].whh_dev, n * sizeof(float), hipMemcpyDeviceToDevice));
            SE_HIP(hipMemcpy(whh_pair[L] + n, pr[1].whh_dev, n * sizeof(float), hipMemcpyDeviceToDevice));
        }
        const int DCI[5] = {512, 256, 128, 64, 32}, DCO[5] = {128, 64, 32, 16, 1};
        for (int br = 0; br < 2; ++br) {
            for (int i = 0; i < 5; ++i) {   // :98-112,149-159  GluConvTranspose2d((1,3),(1,2)) -> BN -> (cat) -> ELU
                const std::string name = "conv" + std::to_string(5 - i) + "_t_" + std::to_string(br + 1) + ".";
                DenseW w = glu_w(name, true, {DCI[i], DCO[i], 1, 3});
                DeconvPlan d;
                d.sf = 2;
                // parity classes built from the interleaved (value, gate) rows
                d = make_glu_deconv(w, DCI[i] / 2);
                for (auto& g : d.par) post(g, "bn" + std::to_string(5 - i) + "_t_" + std::to_string(br + 1) + ".");
                dec[br][i] = d;
            }
            fc[br] = make_pointwise_plan(linear_weights(sd.get("fc" + std::to_string(br + 1) + ".weight", {NBIN, NBIN}),
                                                        &sd.get("fc" + std::to_string(br + 1) + ".bias", {NBIN})),
                                         ACT_NONE, {}, 401);
        }
    }

    void plan_buffers(int B, int T) override {
        cur.B = 0;
        bufs(B, T);
    }

    void forward(const float* in, const int64_t* shape, int ndim, float* out, hipStream_t st) override {
        SE_CHECK(ndim == 4 && shape[1] == 2 && shape[3] == NBIN, "GCRN forward expects [B,2,T,161]");
        const int B = (int)shape[0], T = (int)shape[2];
        Bufs& b = bufs(B, T);
        launch_transpose_akt(in, b.spec, T, 2 * B, NBIN, NBIN, (long)T * NBIN, T, (long)NBIN * T, st);
        network(b, st);
        launch_transpose_akt(b.est, out, NBIN, 2 * B, T, T, (long)NBIN * T, NBIN, (long)T * NBIN, st);
    }

    // (causal end to end - the convs have no extent in time, eval BatchNorm is folded, LayerNorm is per frame - so an equal-length
    // batch runs with its rows zero-extended to whole 128 B lines, model.h causal_work_frames; the LSTMs walk the clip's own frames)
    int frame_multiple() const override { return causal_frame_multiple(true); }
    void enhance(const float* wav, long pitch, int B, int L, float* out, long out_pitch, hipStream_t st) override {
        const int T = 1 + L / HOP;
        const int Tw = causal_work_frames(T, true);
        const bool rag = ragged_ctx() != nullptr;
        const int Ts = rag ? Tw : T;          // frames the STFT / iSTFT walk (ragged rows: zeros behind a row's own last frame)
        Bufs& b = bufs(B, Tw);
        b.Tl = T;
        launch_rms_scale(wav, B, L, pitch, b.c, st);                                               // gcrn_decode_vb.py:35-36
        if (Tw != T && !rag) SE_HIP(hipMemsetAsync(b.spec, 0, (size_t)B * 2 * NBIN * Tw * sizeof(float), st));
        launch_stft(ctx.geom, wav, pitch, B, L, L, b.c, ctx.p_in, b.spec, nullptr, Ts, Tw, st);     // :37-44
        network(b, st);                                                                            // :46
        launch_polar_pow(b.est, b.est, B, NBIN, Tw, ctx.p_out, st);                                // :47-55
        launch_istft(ctx.geom, b.est, B, Ts, Tw, b.frames, b.c, out, out_pitch, L, st);             // :56-58
        b.Tl = 0;
    }

    // ---- frame-online mode (model.h): the (1,3) convs have no extent in time (GCRN_noncprs.py:42-83), LayerNorm is per
    // frame, so the only state is (h, c) of the four grouped LSTMs (:5-39) - and the last estimate frames for the iSTFT
    bool stream_supported() const override { return true; }
    void stream_begin(int B, int max_chunk, hipStream_t st) override {
        ss.release();
        ss.B = B;
        ss.first = true;
        ss.hist.push_back(ss.zeros((size_t)B * 2 * NBIN * STREAM_HC, st));      // est
        for (int l = 0; l < 4; ++l) {
            ss.h[l] = ss.zeros((size_t)512 * B, st);
            ss.c[l] = ss.zeros((size_t)1024 * B, st);          // layer 1 writes its units to every other row (stride 2 S)
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
        network(b, st, n);
        launch_polar_pow(b.est, b.est, B, NBIN, Tw, ctx.p_out, st);
        launch_hist_restore(b.est, ss.hist[0], B, 2L * NBIN, Tw, HC, st);     // the history columns saw no LSTM output
        launch_hist_save(b.est, ss.hist[0], B, 2L * NBIN, Tw, HC, st);
        ss.first = false;
        (void)t0;
    }

  private:
    struct Bufs {
        int B = 0, T = 0;
        int Tl = 0;      // > 0: the clip's own frame count when the rows are zero-extended to T (offline equal-length batches)
        float *c, *spec, *est, *frames, *E[5], *EE[4], *D[2][5], *X, *Y, *Z, *G, *cell, *L0;
    } cur;
    StreamState ss;
    GCPlan enc[5], fc[2];
    DeconvPlan dec[2][5];
    LstmBig l1[2], l2[2];
    float* whh_pair[2] = {nullptr, nullptr};      // [2][2048][512]: both groups' W_hh of a layer, for the paired launch
    float *ln_w[2] = {nullptr, nullptr}, *ln_b[2] = {nullptr, nullptr};

    // kernel (1,3), stride 2 in F, no padding; conv2_t has output_padding 1 (an extra bias-only top row, handled by Fout)
    DeconvPlan make_glu_deconv(const DenseW& w, int c0) {
        DeconvPlan out;
        out.sf = 2;
        for (int par = 0; par < 2; ++par) {
            TapSpec ts;
            std::vector<int> sel;
            for (int kf = 0; kf < 3; ++kf) {
                const int num = par - kf;
                if (((num % 2) + 2) % 2 != 0) continue;
                ts.df[ts.ntaps] = num / 2;
                ts.dt[ts.ntaps] = 0;
                ts.ntaps++;
                sel.push_back(kf);
            }
            std::vector<float> ww((size_t)w.M * w.Cin * ts.ntaps);
            for (int m = 0; m < w.M; ++m)
                for (int c = 0; c < w.Cin; ++c)
                    for (int j = 0; j < ts.ntaps; ++j)
                        ww[((size_t)m * w.Cin + c) * ts.ntaps + j] = w.w[((size_t)m * w.Cin + c) * 3 + sel[j]];
            out.par.push_back(gc_make_plan(w.M, w.Cin, ts, ww, w.bias, {}, ACT_ELU, EPI_GLU, 1, 2, par, 401, 1, c0));
        }
        return out;
    }

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
        for (int i = 0; i < 5; ++i) b.E[i] = a.alloc_f(BT * EC[i + 1] * EF[i]);
        for (int i = 0; i < 4; ++i) b.EE[i] = a.alloc_f(BT * EC[i + 1] * EF[i]);
        const int DCO[5] = {128, 64, 32, 16, 1}, DF[5] = {9, 19, 39, 80, 161};
        for (int br = 0; br < 2; ++br)
            for (int i = 0; i < 5; ++i) b.D[br][i] = a.alloc_f(BT * DCO[i] * DF[i]);
        b.X = a.alloc_f(BT * 1024);
        b.Y = a.alloc_f(BT * 1024);
        b.Z = a.alloc_f(BT * 1024);
        b.L0 = a.alloc_f(BT * 1024);
        b.G = a.alloc_f(BT * 2048 * 2);             // gate pre-activations of both groups
        b.cell = a.alloc_f((size_t)1024 * B);
        cur = b;
        return cur;
    }

    // b.spec [B][2][161][T] -> b.est [B][2][161][T];  n_stream > 0: frame-online chunk - only the last n_stream columns are
    // new frames, the LSTMs continue from the carried state
    void network(Bufs& b, hipStream_t st, int n_stream = 0) {
        const int B = b.B, T = b.T;
        Profiler* pf = &ctx.prof;
        // (frame-online chunk: nothing here has an extent in time, so only the n new columns are produced)
        const int tb = n_stream > 0 ? T - n_stream : 0;
        Act4 x = act4(b.spec, 2, NBIN, T);
        // elu(e_k), the skip tensors as the decoders read them (:147-157), leaves with e_k from the encoder layer's own epilogue
        // (GCParams::dst_elu; SE_GCRN_ELU_FOLD=0 and frame-online chunks - thin kernels - : a pass of its own)
        static const bool elu_fold_env = !(getenv("SE_GCRN_ELU_FOLD") && atoi(getenv("SE_GCRN_ELU_FOLD")) == 0);
        const bool elu_fold = elu_fold_env && n_stream == 0 && !enc[0].p.Ws;
        for (int k = 0; k < 5; ++k) {
            run_conv(enc[k], x, nullptr, b.E[k], EC[k + 1], EF[k], B, T, T, st, pf, nullptr, tb, nullptr, 2, false,
                     (elu_fold && k < 4) ? b.EE[k] : nullptr);
            x = act4(b.E[k], EC[k + 1], EF[k], T);
        }
        for (int k = 0; k < 4 && !elu_fold; ++k) launch_elu(b.E[k], b.EE[k], (long)B * EC[k + 1] * EF[k] * T, st);
        // ---- GLSTM, time-major [T][1024][B]
        const long S = B;
        if (n_stream > 0) {
            const int n = n_stream, c0 = T - n;
            const long gh = (long)n * 2048 * S;
            launch_transpose_akt(b.E[4] + c0, b.X, B, 1024, n, 1024L * T, T, 1024L * S, S, st);
            l1[0].run_stream_strided(b.X, 1024L * S, b.G, ss.c[0], ss.h[0], b.Y, 1024L * S, 2, n, (int)S, ss.first, st, pf);
            l1[1].run_stream_strided(b.X + 512L * S, 1024L * S, b.G + gh, ss.c[1], ss.h[1], b.Y + S, 1024L * S, 2, n, (int)S,
                                     ss.first, st, pf);
            launch_layernorm_cf(b.Y, nullptr, ln_w[0], ln_b[0], b.Z, n, 1024, 1, (int)S, 1e-5f, st);
            l2[0].run_stream_strided(b.Z, 1024L * S, b.G, ss.c[2], ss.h[2], b.Y, 1024L * S, 1, n, (int)S, ss.first, st, pf);
            l2[1].run_stream_strided(b.Z + 512L * S, 1024L * S, b.G + gh, ss.c[3], ss.h[3], b.Y + 512L * S, 1024L * S, 1, n,
                                     (int)S, ss.first, st, pf);
            launch_layernorm_cf(b.Y, nullptr, ln_w[1], ln_b[1], b.Z, n, 1024, 1, (int)S, 1e-5f, st);
            launch_transpose_akt(b.Z, b.L0 + c0, n, 1024, B, 1024L * S, S, 1024L * T, T, st);
        } else {
        const int Tl = b.Tl > 0 ? b.Tl : T;      // frames the recurrent section walks (rows may be zero-extended: enhance())
        launch_transpose_akt(b.E[4], b.X, B, 1024, Tl, 1024L * T, T, 1024L * S, S, st);
        // group i reads features [512i, 512i+512); outputs interleaved (row 2j+i) :26-29; both groups in one launch
        run_lstm_pair(l1[0], l1[1], whh_pair[0], b.X, b.X + 512L * S, 1024L * S, b.G, b.cell, b.Y, S, 1024L * S, 2, Tl, (int)S,
                      st, pf);
        launch_layernorm_cf(b.Y, nullptr, ln_w[0], ln_b[0], b.Z, Tl, 1024, 1, (int)S, 1e-5f, st);
        run_lstm_pair(l2[0], l2[1], whh_pair[1], b.Z, b.Z + 512L * S, 1024L * S, b.G, b.cell, b.Y, 512L * S, 1024L * S, 1, Tl,
                      (int)S, st, pf);                                                             // :32-33 (cat)
        launch_layernorm_cf(b.Y, nullptr, ln_w[1], ln_b[1], b.Z, Tl, 1024, 1, (int)S, 1e-5f, st);
        launch_transpose_akt(b.Z, b.L0, Tl, 1024, B, 1024L * S, S, 1024L * T, T, st);
        }
        // ---- two decoders (real, imaginary: :147-163) - independent chains over the same inputs with their own tensors.  Offline
        // the imaginary one runs on a second stream (fork / join through events, as TaylorSENet's separate encoder): the deep levels'
        // launches fill a fraction of the chip each and the two chains fill each other's tails.  SE_GCRN_FORK=0: one stream; under
        // hipGraph replay the fork is captured with the rest (SE_GRAPH_FORK=0: one stream then)
        const int DCO[5] = {128, 64, 32, 16, 1}, DF[5] = {9, 19, 39, 80, 161};
        static const bool fork_env = !(getenv("SE_GCRN_FORK") && atoi(getenv("SE_GCRN_FORK")) == 0);
        const bool fork = fork_env && n_stream == 0 && !stream_ctx() && (!ctx.graphs_wanted() || graph_fork_enabled());
        auto decoder = [&](int br, hipStream_t sd, Profiler* pd) {
            Act4 a0 = act4(b.L0, 256, 4, T);
            Act4 a1 = act4(b.E[4], 256, 4, T);          // cat((out, e5)) without ELU :147
            for (int i = 0; i < 5; ++i) {
                run_deconv(dec[br][i], a0, &a1, b.D[br][i], DCO[i], DF[i], B, T, T, sd, pd, nullptr, tb);
                a0 = act4(b.D[br][i], DCO[i], DF[i], T);
                if (i < 4) a1 = act4(b.EE[3 - i], EC[4 - i], EF[3 - i], T);
            }
            // Linear(161,161) over F (:161-162): the [B][1][161][T] map is a 161-channel pointwise layer
            GCParams p = fc[br].p;
            p.src0 = b.D[br][4]; p.s0_b = (long)NBIN * T; p.s0_c = T; p.s0_f = 0; p.src1 = nullptr;
            p.Fin = 1; p.Tin = T; p.B = B; p.Q = 1; p.Tout = T; p.t_base = tb;
            p.dst = b.est + (long)br * NBIN * T; p.d_b = 2L * NBIN * T; p.d_c = T; p.d_f = 0;
            gc_launch_prof(fc[br], p, sd, pd);
        };
        if (fork) {
            hipStream_t s2 = ctx.aux_stream(0);
            SE_HIP(hipEventRecord(ctx.ev_fork, st));
            SE_HIP(hipStreamWaitEvent(s2, ctx.ev_fork, 0));
            decoder(1, s2, &ctx.aux_prof[0]);
            SE_HIP(hipEventRecord(ctx.ev_join[0], s2));
            decoder(0, st, pf);
            SE_HIP(hipStreamWaitEvent(st, ctx.ev_join[0], 0));
        } else {
            decoder(0, st, pf);
            decoder(1, st, pf);
        }
    }
};

}  // namespace

std::unique_ptr<Model> make_gcrn(EngineCtx& ctx) { return std::unique_ptr<Model>(new Gcrn(ctx)); }

}  // namespace se
