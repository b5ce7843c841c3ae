// TaylorSENet on the MI355X engine.
//
// Reference: TaylorSENet/TaylorSENet.py:8-693 for the decode script's constructor
// (TaylorSENet/taylorsenet_decode_vb.py:11-13: k1=(1,3), k2=(2,3), c=64, kd1=5, cd1=64, d_feat=256,
// dilations [1,2,5,9], p=2, order_num=3, intra/inter 'cat', causal, U2 encoder, no sharing); decode loop :26-51.
//
// Taylor unfolding: zero-order term = sigmoid gain (U^2-Net encoder -> 2 TCM stacks -> U^2-Net decoder) times the
// noisy spectrum; three high-order blocks (1x1 fuse conv over [feature(256) ; previous term (2x161)] -> 2 TCM stacks
// -> two 1x1 convs to 161 bins) follow the recurrence  update = block(feat, pre) + k*pre,  out += update / (k+1)!.
#include "unet.h"

namespace se {

namespace {

constexpr int NFFT = 320, HOP = 160, NBIN = 161, NDIL = 4;
constexpr int DIL[NDIL] = {1, 2, 5, 9};

// zero[b][c][f][t] = gain[b][f][t] * spec[b][c][f][t]     (TaylorSENet.py:73-76: gain*|X| e^{j angle X} = gain * X)
__global__ __launch_bounds__(256) void taylor_zero_kernel(const float* __restrict__ gain, const float* __restrict__ spec,
                                                          float* __restrict__ zero, float* __restrict__ out, long plane,
                                                          long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / plane, r = i - b * plane;
    const long o = b * 2 * plane + r;
    const float g = gain[i];
    const float zr = g * spec[o], zi = g * spec[o + plane];
    zero[o] = zr;
    zero[o + plane] = zi;
    out[o] = zr;
    out[o + plane] = zi;
}
// update = hob + k * pre;  pre <- update;  out += update / (k+1)!      (:85-93)
__global__ __launch_bounds__(256) void taylor_update_kernel(const float* __restrict__ hob, float* __restrict__ pre,
                                                            float* __restrict__ out, long n, float k, float inv_fact) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float u = hob[i] + k * pre[i];
    pre[i] = u;
    out[i] += u * inv_fact;
}

struct TcmStack {       // p x TCM_list(dilations) of SqueezedTCM (:617-685)
    TcmBlock blk[2 * NDIL];
    void load(const TrackedSD& sd, const std::string& p) {
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < NDIL; ++j)
                blk[i * NDIL + j].load(sd, p + "tcms." + std::to_string(i) + ".tcm_list." + std::to_string(j) + ".", DIL[j],
                                       "left_conv", "right_conv", 3, 0, 5);
    }
    void free() {
        for (auto& b : blk) b.free();
    }
    // x [B][256][T] -> result pointer (one of the ping-pong buffers)
    const float* run(const float* x, float* const X[2], const TcmScratch& ts, int B, int T, hipStream_t st, Profiler* pf) const {
        return run_tcm_chain(blk, 2 * NDIL, x, X, ts, B, T, st, pf);
    }
};

class TaylorSENet final : public Model {
  public:
    explicit TaylorSENet(EngineCtx& c) : Model(c) {}
    ~TaylorSENet() override {
        zen.free(); sen.free();
        for (auto& m : zde) m.free();
        zlast.free();
        gc_free_plan(zgain);
        ztcm.free();
        for (size_t k = 0; k < htcm.size(); ++k) {
            gc_free_plan(h_in[k]);
            gc_free_plan(h_out[k]);
            htcm[k].free();
        }
    }
    StftGeom default_geom() const override { return StftGeom{NFFT, HOP, NFFT}; }
    int padded_samples(int L) const override { return ((L + HOP - 1) / HOP) * HOP; }   // taylorsenet_decode_vb.py:31-35
    // network() forks the separate encoder onto a second stream; under hipGraph replay the fork is captured with the rest (the
    // auxiliary stream joins the capture through the fork event and leaves it through the join event - round 5 kept the two
    // mutually exclusive; SE_GRAPH_FORK=0 restores that)

    void finalize(const TrackedSD& sd) override {
        zen.load(sd, "zeroorderblock.en.", 2);
        sen.load(sd, "separate_en.", 2);
        ztcm.load(sd, "zeroorderblock.");
        for (int i = 0; i < 4; ++i)      // U2Net_Decoder 'cat' (:398-424): En_unet_module(128, 64, k1, k2, scale i+1, de)
            zde[i].load(sd, "zeroorderblock.de.meta_unet_list." + std::to_string(i) + ".", 128, 1, 3, i + 1, true, 64);
        zlast = load_gate_deconv_in(sd, "zeroorderblock.de.last_conv.", 128, 16, 2, 5, 64);
        {
            const HostTensor& w = sd.get("zeroorderblock.de.last_conv.3.weight", {1, 16, 1, 1});
            DenseW d = conv_weights(w, &sd.get("zeroorderblock.de.last_conv.3.bias", {1}), true);
            zgain = make_conv_plan(d, 1, 0, 0, 1, 1, ACT_SIGMOID, {}, EPI_ACT, 401);
        }
        // order_num (TaylorSENet.py:27,66-70): 3 in the decode script (taylorsenet_decode_vb.py:11-13), others through SE_CFG_REPEATS
        const int norder = ctx.repeats(3);
        SE_CHECK(norder >= 0 && norder <= 8, "TaylorSENet: order_num outside [0, 8]");
        h_in.resize(norder);
        h_out.resize(norder);
        htcm.resize(norder);
        for (int k = 0; k < norder; ++k) {    // HighOrderBlock (:155-214)
            const std::string p = "highorderblock_list." + std::to_string(k) + ".";
            auto c1 = [&](const std::string& key, int co, int ci) {
                HostTensor w4 = sd.get(key + "weight", {co, ci, 1});
                w4.shape = {co, ci, 1, 1};
                return conv_weights(w4, &sd.get(key + "bias", {co}), false);
            };
            DenseW in = c1(p + "in_conv.", 256, 256 + 2 * NBIN);
            h_in[k] = gc_make_plan(256, 256 + 2 * NBIN, one_tap(), in.w, in.bias, {}, ACT_NONE, EPI_ACT, 1, 1, 0, 401, 1, 256);
            h_out[k] = make_pointwise_plan(concat_rows(c1(p + "real_resi.", NBIN, 256), c1(p + "imag_resi.", NBIN, 256)),
                                           ACT_NONE, {}, 401);
            htcm[k].load(sd, p);
        }
        cum = zen.last.na.cum;
    }

    void plan_buffers(int B, int T) override {
        cur.B = 0;
        bufs(B, T);
    }

    void forward(const float* in, const int64_t* shape, int ndim, float* out, hipStream_t st) override {
        SE_CHECK(ndim == 4 && shape[1] == 2 && shape[3] == NBIN, "TaylorSENet forward expects [B,2,T,161]");
        const int B = (int)shape[0], T = (int)shape[2];
        Bufs& b = bufs(B, T);
        launch_transpose_akt(in, b.spec, T, 2 * B, NBIN, NBIN, (long)T * NBIN, T, (long)NBIN * T, st);
        network(b, st);
        launch_transpose_akt(b.est, out, NBIN, 2 * B, T, T, (long)NBIN * T, NBIN, (long)T * NBIN, st);
    }

    int frame_multiple() const override { return causal_frame_multiple(cum); }
    void enhance(const float* wav, long pitch, int B, int L, float* out, long out_pitch, hipStream_t st) override {
        const int Lpad = padded_samples(L), T = 1 + Lpad / HOP;
        // InstanceNorm weights: rows of whole 128 B lines as ragged rows of one length; cLN weights: zero-extended (model.h)
        PadFrames pad(ctx, B, L, Lpad, T, L, st, cum ? 1 : in_pad_multiple());
        const int Tw = cum ? causal_work_frames(T, true) : pad.T;
        const bool rag = ragged_ctx() != nullptr;
        const int Ts = (cum && !rag) ? T : Tw;          // frames the STFT / iSTFT walk (ragged rows: zeros behind a row's own last frame)
        Bufs& b = bufs(B, Tw);
        launch_rms_scale(wav, B, L, pitch, b.c, st);                                               // :27-28
        if (Tw != T && cum && !rag) SE_HIP(hipMemsetAsync(b.spec, 0, (size_t)B * 2 * NBIN * Tw * sizeof(float), st));
        launch_stft(ctx.geom, wav, pitch, B, L, Lpad, b.c, ctx.p_in, b.spec, nullptr, Ts, Tw, st);  // :30-41
        network(b, st);                                                                            // :42
        launch_polar_pow(b.est, b.est, B, NBIN, Tw, ctx.p_out, st);                                // :44-45
        launch_istft(ctx.geom, b.est, B, Ts, Tw, b.frames, b.c, out, out_pitch, L, st);             // :48-51
    }

    // ---- frame-online mode (TaylorSENet_new: cumulative LayerNorms only).  Windows of SH history columns + n new frames
    // through the same launch sequence; history / cLN sums are kept by the shared helpers (kernels.h: StreamCtx).  SH is the
    // deepest look-back, (5 - 1) * 9 frames of the widest dilated conv.
    // (with one kernel per TCM block, k_tcm_stream.hip, the dilated convs and FIRs keep their own ring state and the windows
    // only serve the U-Net's one-frame look-back and the iSTFT overlap: 4 columns - rows of 5 floats instead of 37, and a
    // one-frame access touches a fraction of the cache lines)
    const int SH = tcm_stream_enabled() ? 4 : 36;
    bool stream_supported() const override { return cum; }
    int stream_hc() const override { return SH; }
    void stream_begin(int B, int max_chunk, hipStream_t st) override {
        SE_CHECK(cum, "frame-online TaylorSENet needs the cumulative-LayerNorm (`_new`) weights");
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
        StreamScope sc(slots, SH, n, t0, B);
        network(b, st);
        launch_polar_pow(b.est, b.est, B, NBIN, b.T, ctx.p_out, st);
        stream_exchange(b.est, 2L * NBIN * b.T, (long)NBIN * b.T, b.T, B, 2, NBIN, 2, st);   // the iSTFT overlaps one frame back
    }

  private:
    StreamSlots slots;
    bool cum = false;
    struct Bufs {
        int B = 0, T = 0;
        float *c, *spec, *est, *frames, *ens[5], *sens[5], *dx[4], *dlast, *gain, *zero, *hx, *hob, *X[2];
        UnetScratch us, us2;      // us2: the separate encoder's own scratch (it runs on a second stream next to the zero-order block)
        TcmScratch ts;
    } cur;
    U2Encoder zen, sen;
    UnetModule zde[4];
    DeconvIN zlast;
    GCPlan zgain;
    std::vector<GCPlan> h_in, h_out;
    TcmStack ztcm;
    std::vector<TcmStack> htcm;

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
        b.zero = a.alloc_f(BT * 2 * NBIN);
        b.hob = a.alloc_f(BT * 2 * NBIN);
        b.frames = nullptr;      // the fused iSTFT keeps its frames in LDS (k_stft.hip); kept in the struct for the launcher signature
        for (int i = 0; i < 5; ++i) {
            b.ens[i] = a.alloc_f(BT * 64 * F[i]);
            b.sens[i] = a.alloc_f(BT * 64 * F[i]);
        }
        const int DFo[4] = {9, 19, 39, 79};
        for (int i = 0; i < 4; ++i) b.dx[i] = a.alloc_f(BT * 64 * DFo[i]);
        b.dlast = a.alloc_f(BT * 16 * NBIN);
        b.gain = a.alloc_f(BT * NBIN);
        b.hx = a.alloc_f(BT * 256);
        b.X[0] = a.alloc_f(BT * 256);
        b.X[1] = a.alloc_f(BT * 256);
        b.us.alloc(a, BT, B);
        b.us2.alloc(a, BT, B);
        b.ts.h = a.alloc_f(BT * 64);
        b.ts.a = a.alloc_f(BT * 64);
        b.ts.r = a.alloc_f(BT * 64);
        b.ts.m = a.alloc_f(BT * 64);
        cur = b;
        return cur;
    }

    // b.spec [B][2][161][T] -> b.est [B][2][161][T]
    void network(Bufs& b, hipStream_t st) {
        const int B = b.B, T = b.T;
        Profiler* pf = &ctx.prof;
        const long n2 = (long)B * 2 * NBIN * T;
        // The separate encoder (:78-82) reads only the spectrum: offline it runs on a second stream NEXT TO the zero-order block
        // (fork / join through events, as FullSubNet's column ranges do).  Both are long chains of launches of which the deep
        // U^2-Net levels fill a fraction of the chip each (F = 4 ... 19: one to three rounds of workgroups, a tail per launch) -
        // two independent chains fill each other's tails.  SE_TAYLOR_FORK=0: one stream.
        static const bool fork_env = !(getenv("SE_TAYLOR_FORK") && atoi(getenv("SE_TAYLOR_FORK")) == 0);
        // Frame-online (round 6): a push is a chain of ~235 dependent launches of a few microseconds each, a third of them the
        // separate encoder's - the two chains side by side shorten the push by that third.  One- / two-frame chunks only (longer
        // ones take the cLN path that owns a device-wide scratch buffer); the encoder is ENQUEUED first in every frame-online
        // chunk, forked or not, so that the state slots are taken in one order whatever the chunk length.  SE_TAYLOR_STREAM_FORK=0.
        static const bool sfork_env = !(getenv("SE_TAYLOR_STREAM_FORK") && atoi(getenv("SE_TAYLOR_STREAM_FORK")) == 0);
        const StreamCtx* scx = stream_ctx();
        const bool fork = fork_env && !batch_split_active() && (!ctx.graphs_wanted() || (!scx && graph_fork_enabled())) &&
                          (!scx || (sfork_env && scx->n <= 2));
        const bool sen_first = fork || scx;
        const bool turns = scx || fork;
        if (turns) {
            // (the two encoders are enqueued module by module in turn: the host is ~3.5 us per launch ahead of nothing - a chain
            // whose ~80 launches are enqueued behind the other's starts 0.27 ms late; offline that is 5 % of a single clip's decode)
            hipStream_t s2 = fork ? ctx.aux_stream(0) : st;
            if (fork) {
                SE_HIP(hipEventRecord(ctx.ev_fork, st));
                SE_HIP(hipStreamWaitEvent(s2, ctx.ev_fork, 0));
            }
            const int EF[5] = {79, 39, 19, 9, 4};
            for (int i = 0; i < 5; ++i)
                for (int which = 0; which < 2; ++which) {
                    const U2Encoder& e = which ? zen : sen;
                    float* const* ens = which ? b.ens : b.sens;
                    const UnetScratch& us = (which || !fork) ? b.us : b.us2;
                    hipStream_t s = which ? st : s2;
                    Profiler* p = (which || !fork) ? pf : &ctx.aux_prof[0];
                    const Act4 x = i == 0 ? act4(b.spec, 2, NBIN, T) : act4(ens[i - 1], 64, EF[i - 1], T);
                    if (i < 4) e.m[i].run(x, nullptr, ens[i], us, B, T, s, p);
                    else conv_norm2d_prelu(e.last.plan, e.last.na, x, nullptr, ens[4], ens[4], 64, 4, B, T, s, p);
                }
            if (fork) SE_HIP(hipEventRecord(ctx.ev_join[0], s2));
        }
        // ---- zero-order block (:139-153)
        if (!turns) zen.run(act4(b.spec, 2, NBIN, T), b.ens, b.us, B, T, st, pf);
        const float* x = ztcm.run(b.ens[4], b.X, b.ts, B, T, st, pf);      // [B][64*4][T] view of the bottleneck
        Act4 a0 = act4(x, 64, 4, T);
        int F = 4;
        for (int i = 0; i < 4; ++i) {                                       // U2Net_Decoder.forward 'cat' (:432-438)
            Act4 a1 = act4(b.ens[4 - i], 64, F, T);
            zde[i].run(a0, &a1, b.dx[i], b.us, B, T, st, pf);
            F = zde[i].out_F(F);
            a0 = act4(b.dx[i], 64, F, T);
        }
        {
            Act4 a1 = act4(b.ens[0], 64, 79, T);
            run_deconv(zlast.plan, a0, &a1, b.dlast, 16, NBIN, B, T, T, st, pf);
            norm2d_prelu(zlast.na, b.dlast, b.dlast, B, 16, NBIN, T, st);
            run_conv(zgain, act4(b.dlast, 16, NBIN, T), nullptr, b.gain, 1, NBIN, B, T, T, st, pf);
        }
        const long plane = (long)NBIN * T, tot = plane * B;
        hipLaunchKernelGGL(taylor_zero_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, b.gain, b.spec, b.zero, b.est,
                           plane, tot);
        // ---- separate encoder (:78-82) and the high-order recurrence (:84-93); `zero` doubles as pre_term
        if (fork) SE_HIP(hipStreamWaitEvent(st, ctx.ev_join[0], 0));
        else if (!sen_first) sen.run(act4(b.spec, 2, NBIN, T), b.sens, b.us, B, T, st, pf);
        float fact = 1.f;
        for (int k = 0; k < (int)htcm.size(); ++k) {
            GCParams p = h_in[k].p;      // in_conv over cat(feature_head [B][256][T], pre [B][322][T])
            p.src0 = b.sens[4]; p.s0_b = 256L * T; p.s0_c = T; p.s0_f = 0; p.C0 = 256;
            p.src1 = b.zero; p.s1_b = 2L * NBIN * T; p.s1_c = T; p.s1_f = 0; p.C1 = 2 * NBIN;
            p.Fin = 1; p.Tin = T; p.B = B; p.Q = 1; p.Tout = T;
            p.dst = b.hx; p.d_b = 256L * T; p.d_c = T; p.d_f = 0;
            gc_launch_prof(h_in[k], p, st, pf);
            const float* y = htcm[k].run(b.hx, b.X, b.ts, B, T, st, pf);
            run_pointwise(h_out[k], y, 256L * T, T, b.hob, 2L * NBIN * T, T, B, T, st, pf);
            fact *= (float)(k + 1);
            hipLaunchKernelGGL(taylor_update_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, b.hob, b.zero, b.est, n2,
                               (float)k, 1.f / fact);
        }
        SE_HIP(hipGetLastError());
    }
};

}  // namespace

std::unique_ptr<Model> make_taylorsenet(EngineCtx& ctx) { return std::unique_ptr<Model>(new TaylorSENet(ctx)); }

}  // namespace se
