// G2Net (glance-and-gaze) on the MI355X engine.
//
// Reference: G2Net_VB/gaf_net_320.py:10-526 for the decode script's constructor (G2Net_VB/com_decode.py:23:
// gaf_base(3, 64, 2, 4, 4, [1,2,5,9], 256+161*2, 256, 256, (2,3), (1,3), 64, 'cat', 3, is_aux=False,
// encoder_type='U2Net', tcm_type='full-band')); decode loop com_decode.py:39-88 (x / c with c = RMS, y * c).
//
// U^2-Net gated encoder -> 3 GAF stages; a stage fuses [feature(256) ; previous estimate (2x161)] with a gated 1x1
// conv (one two-source GEMM over (value, gate) row pairs) in a glance branch (-> sigmoid gain per bin) and a focus
// branch (-> complex residual), each followed by two stacks of un-gated dilated TCMs:
//   x_{s+1} = gain * x_s + residual                                  (gaf_net_320.py:104-115, |x| e^{j angle x} = x)
#include "unet.h"

namespace se {

namespace {

constexpr int NFFT = 320, HOP = 160, NBIN = 161, NDIL = 4;
constexpr int DIL[NDIL] = {1, 2, 5, 9};

// x_next = gain * pre + resi ; gain [B][161][T], pre / resi / out [B][2][161][T]
__global__ __launch_bounds__(256) void gaf_combine_kernel(const float* __restrict__ gain, const float* __restrict__ pre,
                                                          const float* __restrict__ resi, float* __restrict__ out, long plane,
                                                          long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / plane, r = i - b * plane;
    const long o = b * 2 * plane + r;
    const float g = gain[i];
    out[o] = g * pre[o] + resi[o];
    out[o + plane] = g * pre[o + plane] + resi[o + plane];
}

struct G2TcmSeq {      // nn.Sequential(Tcm_list, Tcm_list, Conv1d(256, 161, 1)[, Sigmoid])
    TcmBlock blk[2 * NDIL];
    GCPlan out;
    void load(const TrackedSD& sd, const std::string& p, int act, DenseW* stack_with = nullptr) {
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < NDIL; ++j)
                blk[i * NDIL + j].load(sd, p + std::to_string(i) + ".tcm_list." + std::to_string(j) + ".", DIL[j], "left_conv",
                                       "left_conv", 3, 0, 3, false);
        HostTensor w4 = sd.get(p + "2.weight", {NBIN, 256, 1});
        w4.shape = {NBIN, 256, 1, 1};
        out = make_pointwise_plan(conv_weights(w4, &sd.get(p + "2.bias", {NBIN}), false), act, {}, 401);
        (void)stack_with;
    }
    void free() {
        for (auto& b : blk) b.free();
        gc_free_plan(out);
    }
    // x [B][256][T] -> dst ([B] planes of 161 x T at stride dst_b)
    void run(const float* x, float* const X[2], const TcmScratch& ts, float* dst, long dst_b, int B, int T, hipStream_t st,
             Profiler* pf) const {
        x = run_tcm_chain(blk, 2 * NDIL, x, X, ts, B, T, st, pf);
        run_pointwise(out, x, 256L * T, T, dst, dst_b, T, B, T, st, pf);
    }
};

struct GafStage {
    GCPlan gin, fin;          // gated input convs of the glance / focus branch (two-source, (value, gate) pairs)
    G2TcmSeq glance, fr, fi;
    void load(const TrackedSD& sd, const std::string& p) {
        auto gate_in = [&](const std::string& q) {
            auto c1 = [&](const std::string& key) {
                HostTensor w4 = sd.get(key + "weight", {256, 256 + 2 * NBIN, 1});
                w4.shape = {256, 256 + 2 * NBIN, 1, 1};
                return conv_weights(w4, &sd.get(key + "bias", {256}), false);
            };
            DenseW w = interleave_rows(c1(q + "in_conv_main."), c1(q + "in_conv_gate.0."));
            return gc_make_plan(512, 256 + 2 * NBIN, one_tap(), w.w, w.bias, {}, ACT_NONE, EPI_GLU, 1, 1, 0, 401, 1, 256);
        };
        gin = gate_in(p + "glance_branch.");
        fin = gate_in(p + "focus_branch.");
        glance.load(sd, p + "glance_branch.mstcm_filter.", ACT_SIGMOID);
        fr.load(sd, p + "focus_branch.mstcm_r.", ACT_NONE);
        fi.load(sd, p + "focus_branch.mstcm_i.", ACT_NONE);
    }
    void free() {
        gc_free_plan(gin);
        gc_free_plan(fin);
        glance.free();
        fr.free();
        fi.free();
    }
};

class G2Net final : public Model {
  public:
    explicit G2Net(EngineCtx& c) : Model(c) {}
    ~G2Net() override {
        en.free();
        for (auto& s : st_) s.free();
    }
    StftGeom default_geom() const override { return StftGeom{NFFT, HOP, NFFT}; }

    void finalize(const TrackedSD& sd) override {
        en.load(sd, "en.", 2, UNET_G2NET, 2);
        // stage_num (gaf_net_320.py:27,55-58): 3 in the decode script (com_decode.py:23), others through SE_CFG_REPEATS
        const int nstage = ctx.repeats(3);
        SE_CHECK(nstage >= 1 && nstage <= 8, "G2Net: stage_num outside [1, 8]");
        st_.resize(nstage);
        for (int s = 0; s < nstage; ++s) st_[s].load(sd, "gafs." + std::to_string(s) + ".");
        cum = en.last.na.cum;
    }

    void plan_buffers(int B, int T) override {
        cur.B = 0;
        bufs(B, T);
    }

    // forward returns the LAST stage output, [B,2,161,T] (the decode script takes esti_x_list[-1], com_decode.py:69)
    void forward(const float* in, const int64_t* shape, int ndim, float* out, hipStream_t st) override {
        SE_CHECK(ndim == 4 && shape[1] == 2 && shape[3] == NBIN, "G2Net forward expects [B,2,T,161]");
        const int B = (int)shape[0], T = (int)shape[2];
        Bufs& b = bufs(B, T);
        launch_transpose_akt(in, b.spec, T, 2 * B, NBIN, NBIN, (long)T * NBIN, T, (long)NBIN * T, st);
        const float* y = network(b, st);
        SE_HIP(hipMemcpyAsync(out, y, (size_t)B * 2 * NBIN * T * sizeof(float), hipMemcpyDeviceToDevice, st));
    }

    int frame_multiple() const override { return causal_frame_multiple(cum); }
    void enhance(const float* wav, long pitch, int B, int L, float* out, long out_pitch, hipStream_t st) override {
        const int T = 1 + L / HOP;
        // InstanceNorm weights: rows of whole 128 B lines as ragged rows of one length; cLN weights: zero-extended (model.h)
        PadFrames pad(ctx, B, L, L, T, L, st, cum ? 1 : in_pad_multiple());
        const int Tw = cum ? causal_work_frames(T, true) : pad.T;
        const bool rag = ragged_ctx() != nullptr;
        const int Ts = (cum && !rag) ? T : Tw;          // frames the STFT / iSTFT walk (ragged rows: zeros behind a row's own last frame)
        Bufs& b = bufs(B, Tw);
        launch_rms_scale(wav, B, L, pitch, b.c, st);        // c_engine = 1 / RMS: x * c_engine == x / RMS (:43-44), y / c_engine == y * RMS (:88)
        if (Tw != T && cum && !rag) SE_HIP(hipMemsetAsync(b.spec, 0, (size_t)B * 2 * NBIN * Tw * sizeof(float), st));
        launch_stft(ctx.geom, wav, pitch, B, L, L, b.c, ctx.p_in, b.spec, nullptr, Ts, Tw, st);     // :49-61
        const float* y = network(b, st);                                                           // :66-69
        launch_polar_pow(y, b.est, B, NBIN, Tw, ctx.p_out, st);                                    // :76-82
        launch_istft(ctx.geom, b.est, B, Ts, Tw, b.frames, b.c, out, out_pitch, L, st);             // :86-88
    }

    // ---- frame-online mode (G2Net_new: cumulative LayerNorms only).  Windows of SH history columns + n new frames through
    // the same launch sequence; history / cLN sums are kept by the shared helpers (kernels.h: StreamCtx).  SH covers the
    // deepest look-back, (3 - 1) * 9 frames of the widest dilated conv.
    // (with one kernel per TCM block, k_tcm_stream.hip, the dilated convs and FIRs keep their own ring state and the windows
    // only serve the U-Net's one-frame look-back and the iSTFT overlap: 4 columns - rows of 5 floats instead of 21, and a
    // one-frame access touches a fraction of the cache lines)
    const int SH = tcm_stream_enabled() ? 4 : 20;
    bool stream_supported() const override { return cum; }
    int stream_hc() const override { return SH; }
    void stream_begin(int B, int max_chunk, hipStream_t st) override {
        SE_CHECK(cum, "frame-online G2Net needs the cumulative-LayerNorm (`_new`) weights");
        slots.begin(B, st);
    }
    void stream_bufs(int B, int n, float** spec, float** mag, float** est) override {
        Bufs& b = bufs(B, SH + n);
        *spec = b.spec;
        *mag = nullptr;
        *est = b.est;
    }
    void stream_chunk(int B, int t0, int n, hipStream_t st, bool last) override {
        (void)last;
        Bufs& b = bufs(B, SH + n);
        const int T = b.T;
        StreamScope sc(slots, SH, n, t0, B);
        const float* y = network(b, st);
        launch_polar_pow(y, b.est, B, NBIN, T, ctx.p_out, st);
        stream_exchange(b.est, 2L * NBIN * T, (long)NBIN * T, T, B, 2, NBIN, 2, st);      // the iSTFT overlaps one frame back
    }

  private:
    StreamSlots slots;
    bool cum = false;
    struct Bufs {
        int B = 0, T = 0;
        float *c, *spec, *est, *frames, *ens[5], *pre[2], *gain, *resi, *hx, *X[2];
        UnetScratch us;
        TcmScratch ts;
        // frame-online windows only (T small): the glance branch's own input, and a second / third set of TCM scratch, so that the
        // three TCM sequences of a stage can run side by side (network())
        float *hxg = nullptr, *Xg[2] = {nullptr, nullptr}, *Xi[2] = {nullptr, nullptr};
        TcmScratch tsg{}, tsi{};
    } cur;
    U2Encoder en;
    std::vector<GafStage> st_;

    Bufs& bufs(int B, int T) {
        if (cur.B == B && cur.T == T) return cur;
        Arena& a = ctx.arena;
        a.reset();
        Bufs b;
        b.B = B;
        b.T = T;
        const size_t BT = (size_t)B * T;
        const int F[5] = {79, 39, 19, 9, 4};
        b.c = a.alloc_f(B);
        b.spec = a.alloc_f(BT * 2 * NBIN);
        b.est = a.alloc_f(BT * 2 * NBIN);
        b.pre[0] = a.alloc_f(BT * 2 * NBIN);
        b.pre[1] = a.alloc_f(BT * 2 * NBIN);
        b.resi = a.alloc_f(BT * 2 * NBIN);
        b.gain = a.alloc_f(BT * NBIN);
        b.frames = nullptr;      // the fused iSTFT keeps its frames in LDS (k_stft.hip); kept in the struct for the launcher signature
        for (int i = 0; i < 5; ++i) b.ens[i] = a.alloc_f(BT * 64 * F[i]);
        b.hx = a.alloc_f(BT * 256);
        b.X[0] = a.alloc_f(BT * 256);
        b.X[1] = a.alloc_f(BT * 256);
        b.us.alloc(a, BT, B);
        b.ts.h = a.alloc_f(BT * 64);
        b.ts.a = a.alloc_f(BT * 64);
        b.ts.r = a.alloc_f(BT * 64);
        b.ts.m = a.alloc_f(BT * 64);
        static const bool ofork_env = !(getenv("SE_G2NET_FORK") && atoi(getenv("SE_G2NET_FORK")) == 0);
        if (T <= 64 || ofork_env) {
            b.hxg = a.alloc_f(BT * 256);
            for (float** X : {b.Xg, b.Xi}) {
                X[0] = a.alloc_f(BT * 256);
                X[1] = a.alloc_f(BT * 256);
            }
            for (TcmScratch* t : {&b.tsg, &b.tsi}) {
                t->h = a.alloc_f(BT * 64);
                t->a = a.alloc_f(BT * 64);
                t->r = a.alloc_f(BT * 64);
                t->m = a.alloc_f(BT * 64);
            }
        }
        cur = b;
        return cur;
    }

    void gate_in(const GCPlan& pl, const float* feat, const float* pre, float* dst, int B, int T, hipStream_t st,
                 Profiler* pf = nullptr) {
        GCParams p = pl.p;
        p.src0 = feat; p.s0_b = 256L * T; p.s0_c = T; p.s0_f = 0; p.C0 = 256;
        p.src1 = pre; p.s1_b = 2L * NBIN * T; p.s1_c = T; p.s1_f = 0; p.C1 = 2 * NBIN;
        p.Fin = 1; p.Tin = T; p.B = B; p.Q = 1; p.Tout = T;
        p.dst = dst; p.d_b = 256L * T; p.d_c = T; p.d_f = 0;
        gc_launch_prof(pl, p, st, pf ? pf : &ctx.prof);
    }

    // b.spec [B][2][161][T] -> pointer to the last stage output [B][2][161][T]
    const float* network(Bufs& b, hipStream_t st) {
        const int B = b.B, T = b.T;
        Profiler* pf = &ctx.prof;
        en.run(act4(b.spec, 2, NBIN, T), b.ens, b.us, B, T, st, pf);
        const float* feat = b.ens[4];            // [B][256][T]
        const float* pre = b.spec;               // inpt.transpose(-2,-1) is the engine layout already (:78)
        const long plane = (long)NBIN * T, tot = plane * B;
        // Frame-online (round 6): a stage's three TCM sequences - glance, focus real, focus imaginary - are three chains of
        // ~50 us that depend only on the stage's input; offline at batch 256 each of them fills the chip by itself, in a push
        // each is ONE workgroup per stream.  They run side by side on the caller's stream and two auxiliary ones (fork / join
        // through events; the calls are enqueued in the order of the one-stream form, so the state slots are taken in the same
        // order either way): one-frame push 0.96 -> 0.7 ms.  SE_G2NET_STREAM_FORK=0: one stream.
        static const bool sfork_env = !(getenv("SE_G2NET_STREAM_FORK") && atoi(getenv("SE_G2NET_STREAM_FORK")) == 0);
        // Offline, from the batch on at which every TCM block runs as the one-workgroup-per-utterance kernel (which owns no
        // scratch): the same fork.  Two such kernels do not fit a CU together (104 KB of LDS each) and their matrix pipe is busy
        // half of the time (3.3); the neighbours' launches and the 1 x 1 layers in between fill the gaps: G2Net 6 376 / 6 384 ->
        // 6 579 / 6 577 utt/s at batch 256, G2Net_new 6 387 -> 6 597 (+ 3.1 ... 3.3 %; 0.76 GB more arena).  SE_G2NET_FORK=0: one
        // stream (also under the profiler: its kernel summaries are single-stream durations).
        static const bool ofork_env = !(getenv("SE_G2NET_FORK") && atoi(getenv("SE_G2NET_FORK")) == 0);
        // With three sequences in flight that kernel wins from ONE clip on (batch 1 ... 4: even; 8: + 4.5 %, 32: + 12 %, 64: + 25 %,
        // 96: + 40 %, 128: + 29 %; G2Net_new one clip 5.77 -> 4.20 ms, batch 8 + 38 %, 64 + 36 %), so the model lowers the batch
        // threshold of blocks.h: run_tcm to 1 for its own calls (CTSNet / TaylorSENet, whose TCM groups feed each other, keep 96).
        struct MinBatch {
            const int old = tcm_fused_min_override();
            explicit MinBatch(bool on) { if (on) tcm_fused_min_override() = 1; }
            ~MinBatch() { tcm_fused_min_override() = old; }
        } min_batch(ofork_env && !stream_ctx() && b.hxg && tcm_fused_supported(T));      // (the KERNELS do not depend on the profiler
                                                                                       // or on graph replay - only the fork does:
                                                                                       // a replayed decode stays bit-identical)
        const bool ofork = ofork_env && !stream_ctx() && B >= tcm_fused_min_batch() && tcm_fused_supported(T) &&
                           st_[0].glance.blk[0].fused.w1 && !(pf && pf->on);
        // (offline the fork is captured into a replayed decode's hipGraph like any other work: the auxiliary streams join the
        // capture through the fork event and leave it through the join events)
        const bool fork = ((sfork_env && stream_ctx() && tcm_chain_enabled() && !ctx.graphs_wanted()) ||
                           (ofork && (!ctx.graphs_wanted() || graph_fork_enabled()))) && b.hxg;
        for (int s = 0; s < (int)st_.size(); ++s) {
            if (fork) {
                hipStream_t sg = ctx.aux_stream(0), si = ctx.aux_stream(1);
                (void)ctx.aux_stream(2);                                   // (its join event marks "focus input ready")
                SE_HIP(hipEventRecord(ctx.ev_fork, st));
                SE_HIP(hipStreamWaitEvent(sg, ctx.ev_fork, 0));
                gate_in(st_[s].gin, feat, pre, b.hxg, B, T, sg, &ctx.aux_prof[0]);
                st_[s].glance.run(b.hxg, b.Xg, b.tsg, b.gain, plane, B, T, sg, &ctx.aux_prof[0]);
                SE_HIP(hipEventRecord(ctx.ev_join[0], sg));
                gate_in(st_[s].fin, feat, pre, b.hx, B, T, st);
                SE_HIP(hipEventRecord(ctx.ev_join[2], st));
                st_[s].fr.run(b.hx, b.X, b.ts, b.resi, 2 * plane, B, T, st, pf);
                SE_HIP(hipStreamWaitEvent(si, ctx.ev_join[2], 0));
                st_[s].fi.run(b.hx, b.Xi, b.tsi, b.resi + plane, 2 * plane, B, T, si, &ctx.aux_prof[1]);
                SE_HIP(hipEventRecord(ctx.ev_join[1], si));
                SE_HIP(hipStreamWaitEvent(st, ctx.ev_join[0], 0));
                SE_HIP(hipStreamWaitEvent(st, ctx.ev_join[1], 0));
            } else {
                gate_in(st_[s].gin, feat, pre, b.hx, B, T, st);
                st_[s].glance.run(b.hx, b.X, b.ts, b.gain, plane, B, T, st, pf);
                gate_in(st_[s].fin, feat, pre, b.hx, B, T, st);
                st_[s].fr.run(b.hx, b.X, b.ts, b.resi, 2 * plane, B, T, st, pf);
                st_[s].fi.run(b.hx, b.X, b.ts, b.resi + plane, 2 * plane, B, T, st, pf);
            }
            float* nxt = b.pre[s & 1];
            hipLaunchKernelGGL(gaf_combine_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, b.gain, pre, b.resi, nxt,
                               plane, tot);
            pre = nxt;
        }
        SE_HIP(hipGetLastError());
        return pre;
    }
};

}  // namespace

std::unique_ptr<Model> make_g2net(EngineCtx& ctx) { return std::unique_ptr<Model>(new G2Net(ctx)); }

}  // namespace se
