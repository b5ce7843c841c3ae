// Building blocks shared by the TCM / gated-U-Net models (CTSNet, TaylorSENet, G2Net).
#pragma once
#include "rnn.h"
#include <cmath>

namespace se {

struct NormAct {
    float *g = nullptr, *b = nullptr, *s = nullptr;
    bool cum = false;      // CumulativeLayerNorm (`_new` variants: parameters `gain` / `bias`) instead of InstanceNorm
    bool gain_nonzero = true;      // every channel's gain != 0: the norm can be folded into its consumers (a left-pad frame is
                                   // staged as the raw value that normalises to zero, which a zero gain does not have)
    void load(const TrackedSD& sd, const std::string& in_key, const std::string& prelu_key) {
        cum = sd.has(in_key + "gain");
        // (a tiny gain makes that raw value -shift / scale overflow: below 1e-6 the layer keeps the stand-alone pass)
        for (float v : sd.get(in_key + (cum ? "gain" : "weight")).data) gain_nonzero = gain_nonzero && std::fabs(v) >= 1e-6f;
        g = to_device(sd.get(in_key + (cum ? "gain" : "weight")).data);
        b = to_device(sd.get(in_key + "bias").data);
        s = to_device(sd.get(prelu_key + "weight").data);
    }
    void free() {
        for (float* d : {g, b, s})
            if (d) (void)hipFree(d);
        g = b = s = nullptr;
    }
};

// norm -> PReLU on x [B][C][F][T] (in place allowed)
// res (optional, may alias y): y = PReLU(norm(x)) + res
inline void norm2d_prelu(const NormAct& n, const float* x, float* y, int B, int C, int F, int T, hipStream_t st,
                         const float* res = nullptr) {
    if (n.cum) {
        // the residual rides on the apply pass (offline, k_misc.hip: cln_apply_plane_kernel) or on the one- / two-frame register
        // kernel of a frame-online push
        if (res && (!stream_ctx() || cln_stream_takes_res(C, F))) {
            launch_cln(x, y, n.g, n.b, nullptr, n.s, nullptr, 0, B, C, F, T, st, res);
            return;
        }
        launch_cln(x, y == res ? const_cast<float*>(x) : y, n.g, n.b, nullptr, n.s, nullptr, 0, B, C, F, T, st);
        if (res) launch_add(res, y == res ? x : y, y, (long)B * C * F * T, st);
    } else {
        launch_instnorm_prelu(x, y, n.g, n.b, n.s, B, C, F * T, st, res, T);
    }
}
// conv / deconv -> InstanceNorm -> PReLU (+ res): the conv's epilogue hands the norm its statistics as per-tile partial
// sums, so the norm pass reads the plane once instead of twice (SE_IN_STATS=0: separate statistics pass).  Falls back to
// the two-kernel sequence for the cumulative-LayerNorm variants and for tile configurations without the epilogue.
inline bool in_stats_enabled() {
    static const bool on = !(getenv("SE_IN_STATS") && atoi(getenv("SE_IN_STATS")) == 0);
    // (round 6: the epilogue cuts its partial sums at the row's own frame count - gc_kernel `tstat` - and the norm kernels divide
    // by it, so ragged batches and the padded equal-length batches of model.h PadFrames keep the epilogue statistics and the fold)
    return on;
}
inline float* in_stats_scratch(int B, int C, int F, int T, hipStream_t st) {
    return reinterpret_cast<float*>(device_scratch(2, (size_t)B * C * F * ((T + 31) / 32) * 2 * sizeof(float), st));
}
inline bool cln_stats_enabled() {
    static const bool on = !(getenv("SE_CLN_STATS") && atoi(getenv("SE_CLN_STATS")) == 0);
    return on && !ragged_ctx() && !stream_ctx();
}
inline float* cln_parts_scratch(int B, int Fout, int T, hipStream_t st) {
    return reinterpret_cast<float*>(device_scratch(7, (size_t)B * Fout * T * 2 * sizeof(float), st));
}
inline void conv_norm2d_prelu(const GCPlan& pl, const NormAct& n, const Act4& s0, const Act4* s1, float* y, float* out, int C,
                              int Fout, int B, int T, hipStream_t st, Profiler* pf, const float* res = nullptr) {
    if (n.cum && cln_stats_enabled() && conv_stats_supported(pl) && pl.p.n_mtiles == 1) {      // cLN: per-frame sums from the epilogue
        float* parts = cln_parts_scratch(B, Fout, T, st);
        run_conv(pl, s0, s1, y, C, Fout, B, T, T, st, pf, parts, 0, nullptr, 2, true);
        launch_cln_parts(y, out, n.g, n.b, n.s, parts, B, C, Fout, T, st, res);
        return;
    }
    if (!n.cum && in_stats_enabled() && conv_stats_supported(pl)) {
        float* stats = in_stats_scratch(B, C, Fout, T, st);
        run_conv(pl, s0, s1, y, C, Fout, B, T, T, st, pf, stats);
        launch_instnorm_prelu_stats(y, out, n.g, n.b, n.s, stats, Fout * ((T + 31) / 32), B, C, Fout * T, st, res, T);
        return;
    }
    run_conv(pl, s0, s1, y, C, Fout, B, T, T, st, pf);
    norm2d_prelu(n, y, out, B, C, Fout, T, st, res);
}
inline void deconv_norm2d_prelu(const DeconvPlan& pl, const NormAct& n, const Act4& s0, const Act4* s1, float* y, float* out,
                                int C, int Fout, int B, int T, hipStream_t st, Profiler* pf, const float* res = nullptr) {
    bool one_mtile = !pl.has_pair && !pl.par.empty();
    for (const auto& g : pl.par) one_mtile = one_mtile && g.p.n_mtiles == 1;       // (a one-tap parity class of <= 128 input channels runs on 64-row tiles)
    if (n.cum && cln_stats_enabled() && deconv_stats_supported(pl) && one_mtile) {
        float* parts = cln_parts_scratch(B, Fout, T, st);
        run_deconv(pl, s0, s1, y, C, Fout, B, T, T, st, pf, parts, 0, -1, false, nullptr, 2, true);
        launch_cln_parts(y, out, n.g, n.b, n.s, parts, B, C, Fout, T, st, res);
        return;
    }
    if (!n.cum && in_stats_enabled() && deconv_stats_supported(pl)) {
        float* stats = in_stats_scratch(B, C, Fout, T, st);
        run_deconv(pl, s0, s1, y, C, Fout, B, T, T, st, pf, stats);
        launch_instnorm_prelu_stats(y, out, n.g, n.b, n.s, stats, Fout * ((T + 31) / 32), B, C, Fout * T, st, res, T);
        return;
    }
    run_deconv(pl, s0, s1, y, C, Fout, B, T, T, st, pf);
    norm2d_prelu(n, y, out, B, C, Fout, T, st, res);
}
// The same two layers with the normalisation left to the CONSUMERS (round 5): the conv stores its raw output, its epilogue
// statistics become the per-(b, c) parameters `nrm` ([B][C] float4, kernels.h: launch_instnorm_finalize) that consumers pass as
// Act4::nrm (gc_kernel NRM applies InstanceNorm + PReLU to its B-operand fragments) - the plane is neither read nor written again.
inline void conv_stats_nrm(const GCPlan& pl, const NormAct& n, const Act4& s0, const Act4* s1, float* y, float* nrm, int C, int Fout,
                           int B, int T, hipStream_t st, Profiler* pf) {
    float* stats = in_stats_scratch(B, C, Fout, T, st);
    run_conv(pl, s0, s1, y, C, Fout, B, T, T, st, pf, stats);
    launch_instnorm_finalize(stats, Fout * ((T + 31) / 32), n.g, n.b, n.s, nrm, B, C, Fout * T, st, T);
}
inline void deconv_stats_nrm(const DeconvPlan& pl, const NormAct& n, const Act4& s0, const Act4* s1, float* y, float* nrm, int C,
                             int Fout, int B, int T, hipStream_t st, Profiler* pf) {
    float* stats = in_stats_scratch(B, C, Fout, T, st);
    run_deconv(pl, s0, s1, y, C, Fout, B, T, T, st, pf, stats);
    launch_instnorm_finalize(stats, Fout * ((T + 31) / 32), n.g, n.b, n.s, nrm, B, C, Fout * T, st, T);
}
// PReLU -> norm -> shared FIR on x [B][C][T]
inline void tcm_head(const NormAct& n, const float* fir, int K, const float* x, float* y, int B, int C, int T, hipStream_t st) {
    if (n.cum) launch_cln(x, y, n.g, n.b, n.s, nullptr, fir, K, B, C, 1, T, st);
    else launch_tcm_head(x, y, n.s, n.g, n.b, fir, K, B, C, T, st);
}

struct TcmBlock {      // Glu / glu (Step1_network.py:158-188, Step2_network.py:126-158)
    GCPlan in_conv, convL, convR, out_conv;
    TcmFusedW fused;       // the same block as one kernel per utterance (k_tcm.hip), built when every norm is an InstanceNorm
    TcmStreamW sfused;     // frame-online chunks of the cLN variants as one kernel per block (k_tcm_stream.hip)
    NormAct nL, nR, nO;
    float *firL = nullptr, *firR = nullptr;
    int K = 0, d = 1;
    bool gated = true;     // false: single branch (G2Net_VB/gaf_net_320.py:245-274 Glu has no gate)
    // conv_idx: index of the dilated Conv1d inside the branch Sequential; fir_k: ShareSepConv length (0 = none);
    // ks: kernel size of the dilated conv (causal pad (ks-1)*dil)
    void load(const TrackedSD& sd, const std::string& p, int dil, const std::string& left, const std::string& right,
              int conv_idx, int fir_k, int ks, bool gated_ = true) {
        gated = gated_;
        d = dil;
        K = fir_k;
        const std::string ci = "." + std::to_string(conv_idx) + ".weight";
        auto c1 = [&](const std::string& key, int co, int ci, int k) {
            const HostTensor& w = sd.get(key, {co, ci, k});
            HostTensor w4 = w;
            w4.shape = {co, ci, 1, k};
            return conv_weights(w4, nullptr, false);
        };
        const DenseW w_in = c1(p + "in_conv.weight", 64, 256, 1), w_l = c1(p + left + ci, 64, 64, ks),
                     w_out = c1(p + "out_conv.2.weight", 256, 64, 1);
        DenseW w_r;
        if (gated) w_r = c1(p + right + ci, 64, 64, ks);
        in_conv = make_pointwise_plan(w_in, ACT_NONE, {}, 401);
        if (gated) convR = make_conv_plan(w_r, 1, 0, (ks - 1) * d, 1, d, ACT_SIGMOID, {}, EPI_ACT, 401);
        convL = make_conv_plan(w_l, 1, 0, (ks - 1) * d, 1, d, ACT_NONE, {}, gated ? EPI_MUL : EPI_ACT, 401);
        out_conv = make_pointwise_plan(w_out, ACT_NONE, {}, 401, EPI_ADD);
        if (ks == 3 || ks == 5)        // InstanceNorm and (round 3) cumulative-LayerNorm heads
            fused = tcm_fused_build(w_in.w, w_l.w, gated ? &w_r.w : nullptr, w_out.w, ks);
        if (sd.has(p + left + ".1.gain"))                                 // ... and stream through one kernel per block
            sfused = tcm_stream_build(w_in.w, w_l.w, gated ? &w_r.w : nullptr, w_out.w, ks);
        nL.load(sd, p + left + ".1.", p + left + ".0.");
        if (gated) nR.load(sd, p + right + ".1.", p + right + ".0.");
        nO.load(sd, p + "out_conv.1.", p + "out_conv.0.");
        if (K > 0) {
            firL = to_device(sd.get(p + left + ".2.weight", {1, 1, K}).data);
            if (gated) firR = to_device(sd.get(p + right + ".2.weight", {1, 1, K}).data);
        }
    }
    void free() {
        for (GCPlan* g : {&in_conv, &convL, &convR, &out_conv}) gc_free_plan(*g);
        tcm_fused_free(fused);
        tcm_stream_free(sfused);
        nL.free();
        nR.free();
        nO.free();
        if (firL) (void)hipFree(firL);
        if (firR) (void)hipFree(firR);
    }
};

struct TcmScratch {
    float *h, *a, *r, *m;     // [B][64][T] each
};

// x [B][256][T] -> y [B][256][T]
// batch from which one workgroup per utterance beats the multi-launch path (SE_TCM_FUSED_MINB; 0 = never fuse)
// (a model that runs several TCM sequences side by side lowers it for its own calls - G2Net, round 6: with three sequences in
// flight the one-workgroup-per-utterance kernel wins from one clip on; the environment variable overrides both)
inline int& tcm_fused_min_override() {
    static thread_local int v = 0;
    return v;
}
inline int tcm_fused_min_batch() {
    static const int env = getenv("SE_TCM_FUSED_MINB") ? atoi(getenv("SE_TCM_FUSED_MINB")) : -1;
    if (env >= 0) return env;
    return tcm_fused_min_override() > 0 ? tcm_fused_min_override() : 96;
}
inline void run_tcm(const TcmBlock& k, const float* x, float* y, const TcmScratch& s, int B, int T, hipStream_t st, Profiler* pf) {
    if (stream_ctx() && k.sfused.w_in && k.nL.cum && k.nO.cum && tcm_stream_enabled()) {
        const TcmFusedHeads hd{k.nL.s, k.nL.g, k.nL.b, k.firL, k.nR.s, k.nR.g, k.nR.b, k.firR, k.nO.s, k.nO.g, k.nO.b};
        launch_tcm_stream(k.sfused, hd, x, y, k.d, k.K, st);
        return;
    }
    static const bool cum_fused = !(getenv("SE_TCM_FUSED_CLN") && atoi(getenv("SE_TCM_FUSED_CLN")) == 0);
    if (k.fused.w1 && !stream_ctx() && (!k.nL.cum || cum_fused) && tcm_fused_min_batch() > 0 && B >= tcm_fused_min_batch() &&
        tcm_fused_supported(T)) {
        const TcmFusedHeads hd{k.nL.s, k.nL.g, k.nL.b, k.firL, k.nR.s, k.nR.g, k.nR.b, k.firR, k.nO.s, k.nO.g, k.nO.b};
        const bool timed = pf && pf->on;
        if (timed) pf->begin(st);
        launch_tcm_fused(k.fused, hd, x, y, B, T, k.d, k.K, st, k.nL.cum);
        if (timed) pf->end(st, 2.0 * B * T * (64.0 * 256 * 2 + 64.0 * 64 * k.fused.ks * (k.gated ? 2 : 1)));
        return;
    }
    run_pointwise(k.in_conv, x, 256L * T, T, s.h, 64L * T, T, B, T, st, pf);
    if (k.gated) {
        tcm_head(k.nR, k.firR, k.K, s.h, s.a, B, 64, T, st);
        run_conv(k.convR, act4(s.a, 64, 1, T), nullptr, s.r, 64, 1, B, T, T, st, pf);
    }
    tcm_head(k.nL, k.firL, k.K, s.h, s.a, B, 64, T, st);
    {
        GCParams p = k.convL.p;
        p.src0 = s.a; p.s0_b = 64L * T; p.s0_c = T; p.s0_f = T; p.src1 = nullptr;
        p.Fin = 1; p.Tin = T; p.B = B; p.Q = 1; p.Tout = T;
        p.dst = s.m; p.d_b = 64L * T; p.d_c = T; p.d_f = T;
        p.aux = s.r; p.x_b = 64L * T; p.x_c = T; p.x_f = T;
        gc_launch_prof(k.convL, p, st, pf);
    }
    tcm_head(k.nO, nullptr, 0, s.m, s.a, B, 64, T, st);
    {
        GCParams p = k.out_conv.p;
        p.src0 = s.a; p.s0_b = 64L * T; p.s0_c = T; p.s0_f = 0; p.src1 = nullptr;
        p.Fin = 1; p.Tin = T; p.B = B; p.Q = 1; p.Tout = T;
        p.dst = y; p.d_b = 256L * T; p.d_c = T; p.d_f = 0;
        p.aux = x; p.x_b = 256L * T; p.x_c = T; p.x_f = 0;
        gc_launch_prof(k.out_conv, p, st, pf);
    }
}

// blocks blk[0..n) feed each other through the ping-pong buffers X[0] / X[1]; returns the buffer that holds the result.
// Frame-online chunks of the cLN variants run up to 8 of them per launch (k_tcm_stream.hip: tcm_chain_kernel).
inline const float* run_tcm_chain(const TcmBlock* blk, int n, const float* x, float* const X[2], const TcmScratch& s, int B, int T,
                                  hipStream_t st, Profiler* pf) {
    int i = 0;
    while (i < n) {
        int m = 1;
        auto chainable = [&](const TcmBlock& k) { return k.sfused.w_in && k.nL.cum && k.nO.cum; };
        if (stream_ctx() && tcm_chain_enabled() && chainable(blk[i])) {
            while (m < 8 && i + m < n && chainable(blk[i + m]) && blk[i + m].sfused.ks == blk[i].sfused.ks) ++m;
            const TcmStreamW* f[8];
            TcmFusedHeads hd[8];
            int dil[8], K[8];
            for (int j = 0; j < m; ++j) {
                const TcmBlock& k = blk[i + j];
                f[j] = &k.sfused;
                hd[j] = TcmFusedHeads{k.nL.s, k.nL.g, k.nL.b, k.firL, k.nR.s, k.nR.g, k.nR.b, k.firR, k.nO.s, k.nO.g, k.nO.b};
                dil[j] = k.d;
                K[j] = k.K;
            }
            float* y = X[(i + m - 1) & 1];
            launch_tcm_chain(f, hd, dil, K, m, x, y, st);
            x = y;
        } else {
            float* y = X[i & 1];
            run_tcm(blk[i], x, y, s, B, T, st, pf);
            x = y;
        }
        i += m;
    }
    return x;
}

}  // namespace se
