// CTSNet (two-stage: magnitude mapping, then complex residual) on the MI355X engine.
//
// Reference: CTSNet/Step1_network.py:12-211 (Step1_net), CTSNet/Step2_network.py:13-210 (Step2_net(X=6, R=3)),
// glue and decode loop CTSNet/two_stage_com_decode_vb.py:62-96.
//
// Engine mapping: a Gate_Conv (conv * sigmoid(gate_conv)) is one tap-table GEMM over (value, gate) row pairs with
// the product in the epilogue; InstanceNorm2d(affine) + PReLU(C) is one plane-wise kernel (per-utterance statistics,
// so batched results equal batch-1 results); the TCM blocks run on [B][256][1][T]: 1x1 convs are pointwise GEMMs,
// PReLU + InstanceNorm1d + ShareSepConv is one row-resident kernel, the dilated k=5 convs are tap-table GEMMs with
// the sigmoid / gate product / residual in their epilogues.
#include "blocks.h"

namespace se {

namespace {

constexpr int NFFT = 320, HOP = 160, NBIN = 161;
constexpr int EF[5] = {79, 39, 19, 9, 4}, DF[5] = {9, 19, 39, 79, 161};

struct GatedEncoder {
    GCPlan conv[5];
    NormAct na[5];
    void load(const TrackedSD& sd, const std::string& p, int cin) {
        for (int i = 0; i < 5; ++i) {
            const std::string q = p + std::to_string(i) + ".";
            const int ci = i == 0 ? cin : 64, kf = i == 0 ? 5 : 3;
            DenseW a = conv_weights(sd.get(q + "0.conv.1.weight", {64, ci, 2, kf}), &sd.get(q + "0.conv.1.bias", {64}), true);
            DenseW g = conv_weights(sd.get(q + "0.gate_conv.1.weight", {64, ci, 2, kf}), &sd.get(q + "0.gate_conv.1.bias", {64}), true);
            conv[i] = make_conv_plan(interleave_rows(a, g), 2, 0, 1, 1, 1, ACT_NONE, {}, EPI_GLU, 401, i == 0 && cin == 4 ? 2 : -1);
            na[i].load(sd, q + "1.", q + "2.");
        }
    }
    void free() {
        for (auto& c : conv) gc_free_plan(c);
        for (auto& n : na) n.free();
    }
    void run(const Act4& in0, const Act4* in1, float* const E[5], int B, int T, hipStream_t st, Profiler* pf) const {
        Act4 x = in0;
        for (int i = 0; i < 5; ++i) {
            conv_norm2d_prelu(conv[i], na[i], x, i == 0 ? in1 : nullptr, E[i], E[i], 64, EF[i], B, T, st, pf);
            x = act4(E[i], 64, EF[i], T);
        }
    }
};

struct GatedDecoder {
    DeconvPlan dc[5];
    NormAct na[5];
    GCPlan fc;
    void load(const TrackedSD& sd, const std::string& p, const std::string& p6, bool softplus) {
        for (int i = 0; i < 5; ++i) {
            const std::string q = p + std::to_string(i) + ".";
            const int co = i == 4 ? 1 : 64, kf = i == 4 ? 5 : 3;
            DenseW a = deconv_weights(sd.get(q + "0.conv.0.weight", {128, co, 2, kf}), &sd.get(q + "0.conv.0.bias", {co}), true);
            DenseW g = deconv_weights(sd.get(q + "0.gate_conv.0.weight", {128, co, 2, kf}), &sd.get(q + "0.gate_conv.0.bias", {co}), true);
            dc[i] = make_deconv_plan(interleave_rows(a, g), 2, 0, 0, ACT_NONE, {}, 401, 64, nullptr, EPI_GLU);
            na[i].load(sd, q + "1.", q + "2.");
        }
        fc = make_pointwise_plan(linear_weights(sd.get(p6 + "0.weight", {NBIN, NBIN}), &sd.get(p6 + "0.bias", {NBIN})),
                                 softplus ? ACT_SOFTPLUS : ACT_NONE, {}, 401);
    }
    void free() {
        for (auto& d : dc) free_deconv_plan(d);
        for (auto& n : na) n.free();
        gc_free_plan(fc);
    }
    // x [B][64][4][T] + skips E -> out (one 161-bin plane per utterance, at out + b*out_b)
    void run(const float* x, float* const E[5], float* const D[5], float* out, long out_b, int B, int T, hipStream_t st,
             Profiler* pf) const {
        Act4 a0 = act4(x, 64, 4, T);
        for (int i = 0; i < 5; ++i) {
            const int co = i == 4 ? 1 : 64;
            Act4 a1 = act4(E[4 - i], 64, a0.F, T);
            deconv_norm2d_prelu(dc[i], na[i], a0, &a1, D[i], D[i], co, DF[i], B, T, st, pf);
            a0 = act4(D[i], co, DF[i], T);
        }
        GCParams p = fc.p;    // Linear(161,161) over F
        p.src0 = D[4]; p.s0_b = (long)NBIN * T; p.s0_c = T; p.s0_f = 0; p.src1 = nullptr;
        p.Fin = 1; p.Tin = T; p.B = B; p.Q = 1; p.Tout = T;
        p.dst = out; p.d_b = out_b; p.d_c = T; p.d_f = 0;
        gc_launch_prof(fc, p, st, pf);
    }
};

class CtsNet final : public Model {
  public:
    explicit CtsNet(EngineCtx& c) : Model(c) {}
    ~CtsNet() override {
        en1.free(); en2.free(); de1.free(); de2r.free(); de2i.free();
        for (auto& t : tcm1) t.free();
        for (auto& t : tcm2) t.free();
    }
    StftGeom default_geom() const override { return StftGeom{NFFT, HOP, NFFT}; }
    int padded_samples(int L) const override { return ((L + HOP - 1) / HOP) * HOP; }   // two_stage_com_decode_vb.py:66-69

    void finalize(const TrackedSD& sd) override {
        has1 = sd.has("step1.de.de6.0.weight");
        has2 = sd.has("step2.de_r.de6.0.weight");
        SE_CHECK(has1 || has2, "CTSNet needs the step1.* and/or step2.* state dicts");
        if (has1) {
            en1.load(sd, "step1.en.en.", 1);
            de1.load(sd, "step1.de.de.", "step1.de.de6.", true);
            tcm1.resize(18);
            for (int k = 0; k < 3; ++k)
                for (int i = 0; i < 6; ++i)
                    tcm1[k * 6 + i].load(sd, "step1.tcm" + std::to_string(k + 1) + ".tcm_list." + std::to_string(i) + ".", 1 << i,
                                         "left_conv", "right_conv", 4, 2 * (1 << i) - 1, 5);
        }
        if (has2) {
            en2.load(sd, "step2.en.en_module.", 4);
            de2r.load(sd, "step2.de_r.de_list.", "step2.de_r.de6.", false);
            de2i.load(sd, "step2.de_i.de_list.", "step2.de_i.de6.", false);
            // Step2_net(X, R) (Step2_network.py:13-21): 6 / 3 in the decode script, others through SE_CFG_REPEATS2 / SE_CFG_REPEATS
            R2 = ctx.repeats(3);
            X2 = ((ctx.flags >> 12) & 15) ? ((ctx.flags >> 12) & 15) - 1 : 6;      // SE_CFG_REPEATS2(X)
            SE_CHECK(R2 >= 1 && R2 <= 8 && X2 >= 1 && X2 <= 6, "Step2_net: R outside [1, 8] or X outside [1, 6]");
            tcm2.resize((size_t)R2 * X2);
            for (int r = 0; r < R2; ++r)
                for (int i = 0; i < X2; ++i)
                    tcm2[r * X2 + i].load(sd, "step2.tcm_list." + std::to_string(r) + ".glu_list." + std::to_string(i) + ".", 1 << i,
                                         "ori_conv", "att_ori", 4, 2 * (1 << i) - 1, 5);
        }
        cum = (has1 ? en1.na[0].cum : true) && (has2 ? en2.na[0].cum : true);
    }

    void plan_buffers(int B, int T) override {
        cur.B = 0;
        bufs(B, T);
    }

    void forward(const float* in, const int64_t* shape, int ndim, float* out, hipStream_t st) override {
        if (ndim == 3) {            // Step1_net: [B,T,161] -> [B,T,161]
            SE_CHECK(has1 && shape[2] == NBIN, "Step1_net forward expects [B,T,161] and loaded step1 weights");
            const int B = (int)shape[0], T = (int)shape[1];
            Bufs& b = bufs(B, T);
            launch_transpose_akt(in, b.mag, T, B, NBIN, NBIN, (long)T * NBIN, T, (long)NBIN * T, st);
            step1(b, b.mag, b.est1, st);
            launch_transpose_akt(b.est1, out, NBIN, B, T, T, (long)NBIN * T, NBIN, (long)T * NBIN, st);
        } else {                    // Step2_net: [B,4,T,161] -> [B,2,T,161]
            SE_CHECK(has2 && ndim == 4 && shape[1] == 4 && shape[3] == NBIN, "Step2_net forward expects [B,4,T,161]");
            const int B = (int)shape[0], T = (int)shape[2];
            Bufs& b = bufs(B, T);
            // [B][4][T][F] -> spec (ch 0,1) and s1 (ch 2,3) in [B][2][F][T]
            for (int half = 0; half < 2; ++half)
                for (int bb = 0; bb < B; ++bb)
                    launch_transpose_akt(in + ((long)bb * 4 + 2 * half) * T * NBIN, (half ? b.s1 : b.spec) + (long)bb * 2 * NBIN * T, T, 2,
                                         NBIN, NBIN, (long)T * NBIN, T, (long)NBIN * T, st);
            step2(b, b.spec, b.s1, b.est, st);
            launch_transpose_akt(b.est, out, NBIN, 2 * B, T, T, (long)NBIN * T, NBIN, (long)T * NBIN, st);
        }
    }

    int frame_multiple() const override { return causal_frame_multiple(cum && has1 && has2); }
    void enhance(const float* wav, long pitch, int B, int L, float* out, long out_pitch, hipStream_t st) override {
        SE_CHECK(has1 && has2, "CTSNet decode needs both stages' weights");
        const int Lpad = padded_samples(L), T = 1 + Lpad / HOP;
        // InstanceNorm weights: rows of whole 128 B lines as ragged rows of one length; cLN weights: zero-extended (model.h)
        PadFrames pad(ctx, B, L, Lpad, T, L, st, cum ? 1 : in_pad_multiple());
        const int Tw = cum ? causal_work_frames(T, true) : pad.T;
        const bool rag = ragged_ctx() != nullptr;
        const int Ts = (cum && !rag) ? T : Tw;          // frames the STFT / iSTFT walk (ragged rows: zeros behind a row's own last frame)
        Bufs& b = bufs(B, Tw);
        launch_rms_scale(wav, B, L, pitch, b.c, st);                                               // :63-64
        if (Tw != T && cum && !rag) {
            SE_HIP(hipMemsetAsync(b.spec, 0, (size_t)B * 2 * NBIN * Tw * sizeof(float), st));
            SE_HIP(hipMemsetAsync(b.mag, 0, (size_t)B * NBIN * Tw * sizeof(float), st));
        }
        launch_stft(ctx.geom, wav, pitch, B, L, Lpad, b.c, ctx.p_in, b.spec, b.mag, Ts, Tw, st);    // :65-76
        step1(b, b.mag, b.est1, st);                                                               // :79
        launch_mag_phase(b.est1, b.spec, b.s1, B, NBIN, Tw, 1.f, st);                              // :80-81
        step2(b, b.spec, b.s1, b.est, st);                                                         // :82-83
        launch_add(b.est, b.s1, b.est, (long)B * 2 * NBIN * Tw, st);                               // :84
        launch_polar_pow(b.est, b.est, B, NBIN, Tw, ctx.p_out, st);                                // :87-90
        launch_istft(ctx.geom, b.est, B, Ts, Tw, b.frames, b.c, out, out_pitch, L, st);             // :93-96 ([:wav_len])
    }

    // ---- frame-online mode (CTSNet_new: every norm is a cumulative LayerNorm, so the whole network is causal).  The chunk
    // runs the same launch sequence as enhance() on windows of SH history columns + n new frames; the shared helpers keep
    // the per-layer history and the cLN sums (kernels.h: StreamCtx).  SH = the deepest look-back: (5 - 1) * 32 frames of
    // the last dilated conv of a TCM group (its ShareSepConv reaches 62 back).
    // (with one kernel per TCM block, k_tcm_stream.hip, the dilated convs and FIRs keep their own ring state and the windows
    // only serve the U-Net's one-frame look-back and the iSTFT overlap: 4 columns - rows of 5 floats instead of 129, and a
    // one-frame access touches a fraction of the cache lines)
    const int SH = tcm_stream_enabled() ? 4 : 128;
    bool stream_supported() const override { return has1 && has2 && cum; }
    int stream_hc() const override { return SH; }
    void stream_begin(int B, int max_chunk, hipStream_t st) override {
        SE_CHECK(stream_supported(), "frame-online CTSNet needs the cumulative-LayerNorm (`_new`) weights of both stages");
        slots.begin(B, st);
    }
    void stream_bufs(int B, int n, float** spec, float** mag, float** est) override {
        Bufs& b = bufs(B, SH + n);
        *spec = b.spec;
        *mag = b.mag;
        *est = b.est;
    }
    void stream_chunk(int B, int t0, int n, hipStream_t st, bool last) override {
        (void)last;
        Bufs& b = bufs(B, SH + n);
        const int T = b.T;
        StreamScope sc(slots, SH, n, t0, B);
        step1(b, b.mag, b.est1, st);
        launch_mag_phase(b.est1, b.spec, b.s1, B, NBIN, T, 1.f, st);
        step2(b, b.spec, b.s1, b.est, st);
        launch_add(b.est, b.s1, b.est, (long)B * 2 * NBIN * T, st);
        launch_polar_pow(b.est, b.est, B, NBIN, T, ctx.p_out, st);
        stream_exchange(b.est, 2L * NBIN * T, (long)NBIN * T, T, B, 2, NBIN, 2, st);      // the iSTFT overlaps one frame back
    }

  private:
    StreamSlots slots;
    bool cum = false;
    struct Bufs {
        int B = 0, T = 0;
        float *c, *spec, *mag, *est1, *s1, *est, *frames, *E[5], *D[5], *X[2], *acc;
        float* D2[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};      // frame-online windows only: the imaginary decoder's own tensors
        TcmScratch ts;
    } cur;
    bool has1 = false, has2 = false;
    GatedEncoder en1, en2;
    GatedDecoder de1, de2r, de2i;
    std::vector<TcmBlock> tcm1, tcm2;
    int R2 = 3, X2 = 6;

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
        b.mag = a.alloc_f(BT * NBIN);
        b.est1 = a.alloc_f(BT * NBIN);
        b.s1 = a.alloc_f(BT * 2 * NBIN);
        b.est = a.alloc_f(BT * 2 * NBIN);
        b.frames = nullptr;      // the fused iSTFT keeps its frames in LDS (k_stft.hip); kept in the struct for the launcher signature
        for (int i = 0; i < 5; ++i) b.E[i] = a.alloc_f(BT * 64 * EF[i]);
        for (int i = 0; i < 5; ++i) b.D[i] = a.alloc_f(BT * 64 * DF[i]);
        if (T <= 64)
            for (int i = 0; i < 5; ++i) b.D2[i] = a.alloc_f(BT * 64 * DF[i]);
        b.X[0] = a.alloc_f(BT * 256);
        b.X[1] = a.alloc_f(BT * 256);
        b.acc = a.alloc_f(BT * 256);
        b.ts.h = a.alloc_f(BT * 64);
        b.ts.a = a.alloc_f(BT * 64);
        b.ts.r = a.alloc_f(BT * 64);
        b.ts.m = a.alloc_f(BT * 64);
        cur = b;
        return cur;
    }

    // x = E[4] viewed as [B][256][T]; returns the accumulated TCM output in b.acc
    void tcm_stack(Bufs& b, const TcmBlock* blocks, int groups, int per, hipStream_t st) {
        const int B = b.B, T = b.T;
        const long n = (long)B * 256 * T;
        const float* x = b.E[4];
        for (int g = 0; g < groups; ++g) {
            x = run_tcm_chain(blocks + g * per, per, x, b.X, b.ts, B, T, st, &ctx.prof);
            if (g == 0) SE_HIP(hipMemcpyAsync(b.acc, x, n * sizeof(float), hipMemcpyDeviceToDevice, st));
            else launch_add(b.acc, x, b.acc, n, st);
        }
    }

    void step1(Bufs& b, const float* mag, float* est, hipStream_t st) {      // Step1_network.py:21-40
        en1.run(act4(mag, 1, NBIN, b.T), nullptr, b.E, b.B, b.T, st, &ctx.prof);
        tcm_stack(b, tcm1.data(), 3, 6, st);
        de1.run(b.acc, b.E, b.D, est, (long)NBIN * b.T, b.B, b.T, st, &ctx.prof);
    }
    void step2(Bufs& b, const float* spec, const float* s1, float* est, hipStream_t st) {   // Step2_network.py:23-38
        Act4 a1 = act4(s1, 2, NBIN, b.T);
        en2.run(act4(spec, 2, NBIN, b.T), &a1, b.E, b.B, b.T, st, &ctx.prof);
        tcm_stack(b, tcm2.data(), R2, X2, st);
        // Frame-online (round 6): the real and the imaginary decoder are two chains of 16 launches that read the same inputs;
        // in a one- / two-frame push they run side by side, the imaginary one on an auxiliary stream with its own tensors.  It is
        // ENQUEUED first in every frame-online chunk, forked or not (the host is ~0.1 ms ahead of the device per decoder: the
        // chain enqueued second starts that much later), so the state slots are taken in one order whatever the chunk length.
        // Offline their 128-row gated layers fill the chip by themselves (profiles/r06_experiments.md).  SE_CTSNET_STREAM_FORK=0.
        static const bool sfork_env = !(getenv("SE_CTSNET_STREAM_FORK") && atoi(getenv("SE_CTSNET_STREAM_FORK")) == 0);
        if (const StreamCtx* scx = stream_ctx()) {
            const bool fork = sfork_env && scx->n <= 2 && b.D2[0] && !ctx.graphs_wanted();
            if (fork) {
                hipStream_t s2 = ctx.aux_stream(0);
                SE_HIP(hipEventRecord(ctx.ev_fork, st));
                SE_HIP(hipStreamWaitEvent(s2, ctx.ev_fork, 0));
                de2i.run(b.acc, b.E, b.D2, est + (long)NBIN * b.T, 2L * NBIN * b.T, b.B, b.T, s2, &ctx.aux_prof[0]);
                SE_HIP(hipEventRecord(ctx.ev_join[0], s2));
            } else {
                de2i.run(b.acc, b.E, b.D, est + (long)NBIN * b.T, 2L * NBIN * b.T, b.B, b.T, st, &ctx.prof);
            }
            de2r.run(b.acc, b.E, b.D, est, 2L * NBIN * b.T, b.B, b.T, st, &ctx.prof);
            if (fork) SE_HIP(hipStreamWaitEvent(st, ctx.ev_join[0], 0));
            return;
        }
        de2r.run(b.acc, b.E, b.D, est, 2L * NBIN * b.T, b.B, b.T, st, &ctx.prof);
        de2i.run(b.acc, b.E, b.D, est + (long)NBIN * b.T, 2L * NBIN * b.T, b.B, b.T, st, &ctx.prof);
    }
};

}  // namespace

std::unique_ptr<Model> make_ctsnet(EngineCtx& ctx) { return std::unique_ptr<Model>(new CtsNet(ctx)); }

}  // namespace se
