// FullSubNet on the MI355X engine.
//
// Reference: FullSubNet/fullsubnet_net_sa/model.py:68-118 (Model.forward), base_model.py:13-42 (unfold),
// :197-209 (offline_laplace_norm), sequence_model.py:66-84; constructor and decode loop
// FullSubNet/fullsubnet_sa_decode_vb.py:11-24, :37-72 (complex mask applied in the script, :56-61).
//
// Batch semantics: the reference only ever decodes B = 1 and its forward runs the training-time `drop_band`
// (model.py:101-104) whenever batch_size > 1; the engine therefore computes B INDEPENDENT batch-1 results (per-
// utterance normalisation statistics, no band dropping), see SURVEY.md 0.8.
//
// Engine mapping (time-major, sequences contiguous):
//   full-band LSTM  : [T+2][257][B]      (B sequences, hidden 512)
//   sub-band LSTM   : [T+2][32][257*B]   (sequence s = n*B + b: sub-band n of utterance b, hidden 384) - the one
//                     place a recurrent step is a large GEMM (M = 1536, K = 384, N = 257*B), run on f32 MFMA with the
//                     LSTM cell fused in the epilogue.
#include "rnn.h"
#include "../../include/se_engine.h"

namespace se {

namespace {

constexpr int NFFT = 512, HOP = 256, NBIN = 257, LA = 2, SBN = 15, SBW = 2 * SBN + 2;   // 31 noisy + 1 full-band

// sum over (f, t) of mag[b][f][t]  ->  mu[b] = sum / (F * (T + LA))   (the look-ahead pad frames are zeros)
// ragged batch (tlen != null): frames >= tlen[b] of mag are zeros (the STFT wrote them), so only the denominator changes
__global__ __launch_bounds__(256) void fsn_mean_kernel(const float* __restrict__ mag, int n, float denom,
                                                       float* __restrict__ mu, const int* __restrict__ tlen) {
    const int b = blockIdx.x;
    if (tlen) denom = (float)NBIN * (tlen[b] + LA);
    const float* x = mag + (long)b * n;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) mu[b] = (float)((part[0] + part[1] + part[2] + part[3]) / denom);
}

// y[i] = x[i] / (mu[i % B] + 1e-5)   for time-major tensors whose innermost index is the utterance
__global__ __launch_bounds__(256) void fsn_scale_kernel(const float* __restrict__ x, float* __restrict__ y, long n,
                                                        int B, const float* __restrict__ mu) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = x[i] / (mu[i % B] + 1e-5f);
}

// sub-band input before normalisation: sb[t][k][n*B + b];  k < 31: noisy mag of bin reflect(n - 15 + k)
// (base_model.py:29-42, reflect pad), k = 31: full-band output of bin n (model.py:88-96)
__global__ __launch_bounds__(256) void fsn_build_sb_kernel(const float* __restrict__ magT, const float* __restrict__ fbo,
                                                           float* __restrict__ sb, int B) {
    const int S = NBIN * B;
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y, t = blockIdx.z;
    if (s >= S) return;
    const int n = s / B, b = s - n * B;
    float v;
    if (k == SBW - 1) {
        v = fbo[((long)t * NBIN + n) * B + b];
    } else {
        int f = n - SBN + k;
        if (f < 0) f = -f;
        if (f >= NBIN) f = 2 * (NBIN - 1) - f;
        v = magT[((long)t * NBIN + f) * B + b];
    }
    sb[((long)t * SBW + k) * S + s] = v;
}

// column sums of a [rows][B] matrix (utterance = innermost index) -> mu[b] = sum / rows; two deterministic stages
// (per-block partials in a fixed order, then a fixed-order fp64 sum) - no float atomics, so repeated decodes are
// bit-identical
constexpr int FSN_SUM_BLOCKS = 1024;
// ragged batch: rows are (t, k, n) t-major; utterance b only counts its own tlen[b] + LA frames
__global__ __launch_bounds__(256) void fsn_colsum_kernel(const float* __restrict__ x, long rows, int B,
                                                         float* __restrict__ part, const int* __restrict__ tlen) {
    __shared__ float sh[256];
    // thread handles column (tid % B) of rows tid / B + k * (256 / B)
    const int per = 256 / B;
    const int b = threadIdx.x % B, r0 = threadIdx.x / B;
    float s = 0.f;
    if (per > 0 && r0 < per) {
        const long rmax = tlen ? (long)(tlen[b] + LA) * SBW * NBIN : rows;
        for (long r = (long)blockIdx.x * per + r0; r < rmax; r += (long)gridDim.x * per) s += x[r * B + b];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (per > 0 && r0 == 0) {
        float t = 0.f;
        for (int k = 0; k < per; ++k) t += sh[k * B + b];
        part[(long)blockIdx.x * B + b] = t;
    }
}
__global__ void fsn_finish_mean_kernel(const float* __restrict__ part, float* __restrict__ mu, float denom, int B,
                                       const int* __restrict__ tlen) {
    const int b = threadIdx.x;
    if (b >= B) return;
    if (tlen) denom = (float)(tlen[b] + LA) * SBW * NBIN;
    double s = 0.0;
    for (int k = 0; k < FSN_SUM_BLOCKS; ++k) s += part[(long)k * B + b];
    mu[b] = (float)(s / denom);
}

// out[b][c][n][t] = mask[(n*B + b)][c][t + LA]                       (forward hook), or
// est = mask (x) spec, decompressed (fullsubnet_sa_decode_vb.py:56-66)  (decode path)
__global__ __launch_bounds__(256) void fsn_mask_kernel(const float* __restrict__ maskBT, const float* __restrict__ spec,
                                                       float* __restrict__ out, int B, int T, float p_out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int n = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const int Tp = T + LA;
    const float* m = maskBT + ((long)(n * B + b) * 2) * Tp + t + LA;
    const float mr = m[0], mi = m[Tp];
    const long o = (((long)b * 2) * NBIN + n) * T + t;
    const long plane = (long)NBIN * T;
    if (!spec) {
        out[o] = mr;
        out[o + plane] = mi;
        return;
    }
    const float xr = spec[o], xi = spec[o + plane];
    float er = mr * xr - mi * xi, ei = mr * xi + mi * xr;
    if (p_out != 1.f) {
        const float mg = sqrtf(er * er + ei * ei);
        const float sc = mg > 0.f ? ((p_out == 2.f) ? mg : powf(mg, p_out - 1.f)) : 0.f;
        er *= sc;
        ei *= sc;
    }
    out[o] = er;
    out[o + plane] = ei;
}

// ---- cumulative_laplace_norm (base_model.py:212-240): x / (mean over (features, frames <= t) + EPSILON), float32 sums in the
// reference's order - the features of a frame first, then a running sum over the frames.  `csum` (optional) carries the running
// sum across the chunks of a frame-online stream; step 0 of the call is frame t_first of the utterance.
constexpr float FSN_EPS = 1.1920928955078125e-07f;        // np.finfo(np.float32).eps (constant.py:8)
// full band: x / y [n][257][B] time-major, one workgroup per utterance
__global__ __launch_bounds__(256) void fsn_cum_fb_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int B,
                                                         float* __restrict__ csum, int t_first) {
    __shared__ float part[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    float run = (csum && t_first > 0) ? csum[b] : 0.f;
    for (int t = 0; t < n; ++t) {
        const float* xr = x + (long)t * NBIN * B + b;
        const float v0 = xr[(long)tid * B], v1 = tid + 256 < NBIN ? xr[(long)(tid + 256) * B] : 0.f;
        float s = v0 + v1;
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        __syncthreads();
        if ((tid & 63) == 0) part[tid >> 6] = s;
        __syncthreads();
        run += (part[0] + part[1]) + (part[2] + part[3]);
        const float d = run / ((float)NBIN * (float)(t_first + t + 1)) + FSN_EPS;
        float* yr = y + (long)t * NBIN * B + b;
        yr[(long)tid * B] = v0 / d;
        if (tid + 256 < NBIN) yr[(long)(tid + 256) * B] = v1 / d;
    }
    if (csum && tid == 0) csum[b] = run;
}
// sub bands: sb [n][32][S] in place, one thread per sequence s (sub-band of an utterance)
__global__ __launch_bounds__(256) void fsn_cum_sb_kernel(float* __restrict__ sb, int n, int S, float* __restrict__ csum, int t_first) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    float run = (csum && t_first > 0) ? csum[s] : 0.f;
    for (int t = 0; t < n; ++t) {
        float* xr = sb + (long)t * SBW * S + s;
        float v[SBW], sum = 0.f;
#pragma unroll
        for (int k = 0; k < SBW; ++k) {
            v[k] = xr[(long)k * S];
            sum += v[k];
        }
        run += sum;
        const float d = run / ((float)SBW * (float)(t_first + t + 1)) + FSN_EPS;
#pragma unroll
        for (int k = 0; k < SBW; ++k) xr[(long)k * S] = v[k] / d;
    }
    if (csum) csum[s] = run;
}
// frame-online: mask of network step tau (maskT [ns][2][S], s = n B + b) applied to the spectrum of frame tau - LA, which sits in
// column col0 + tau of the chunk window ([B][2][257][Tw]); frames before the start of the stream (gt0 + tau < LA) do not exist
__global__ __launch_bounds__(256) void fsn_stream_apply_kernel(const float* __restrict__ maskT, const float* __restrict__ spec,
                                                               float* __restrict__ est, int B, int Tw, int ns, int col0, int gt0,
                                                               float p_out) {
    const int i = blockIdx.x * 256 + threadIdx.x, tau = blockIdx.y;
    const int S = NBIN * B;
    if (i >= S || gt0 + tau < LA) return;
    const int n = i / B, b = i - n * B, col = col0 + tau;
    const float mr = maskT[((long)tau * 2) * S + i], mi = maskT[((long)tau * 2 + 1) * S + i];
    const long o = (((long)b * 2) * NBIN + n) * Tw + col, plane = (long)NBIN * Tw;
    const float xr = spec[o], xi = spec[o + plane];
    float er = mr * xr - mi * xi, ei = mr * xi + mi * xr;
    if (p_out != 1.f) {
        const float mg = sqrtf(er * er + ei * ei);
        const float sc = mg > 0.f ? ((p_out == 2.f) ? mg : powf(mg, p_out - 1.f)) : 0.f;
        er *= sc;
        ei *= sc;
    }
    est[o] = er;
    est[o + plane] = ei;
    (void)ns;
}

class FullSubNet final : public Model {
  public:
    explicit FullSubNet(EngineCtx& c) : Model(c) {}
    ~FullSubNet() override {
        for (auto& l : fb) l.free();
        for (auto& l : sbl) l.free();
        gc_free_plan(fb_fc);
        gc_free_plan(sb_fc);
    }
    StftGeom default_geom() const override { return StftGeom{NFFT, HOP, NFFT}; }

    void finalize(const TrackedSD& sd) override {
        const int S = NBIN * ctx.max_batch;
        // sequence_model = "LSTM" (the decode script's choice, fullsubnet_sa_decode_vb.py:16) or "GRU" (sequence_model.py:36-43)
        const bool gru = (ctx.flags & SE_CFG_FSN_GRU) != 0;
        cum = (ctx.flags & SE_CFG_FSN_CUMULATIVE) != 0;      // norm_type = "cumulative_laplace_norm" (base_model.py:212-240)
        is_gru = gru;
        auto load = [&](const std::string& p, int layer, int I, int H) {
            return gru ? load_gru(sd, p, layer, "", I, H) : load_lstm(sd, p, layer, "", I, H);
        };
        fb[0].build(load("fb_model.sequence_model.", 0, NBIN, 512), ctx.max_batch, gru);
        fb[1].build(load("fb_model.sequence_model.", 1, 512, 512), ctx.max_batch, gru);
        fb_fc = make_pointwise_plan(linear_weights(sd.get("fb_model.fc_output_layer.weight", {NBIN, 512}),
                                                   &sd.get("fb_model.fc_output_layer.bias", {NBIN})),
                                    ACT_RELU, {}, ctx.max_batch);
        sbl[0].build(load("sb_model.sequence_model.", 0, SBW, 384), S, gru, true);
        sbl[1].build(load("sb_model.sequence_model.", 1, 384, 384), S, gru, true);
        sb_fc = make_pointwise_plan(linear_weights(sd.get("sb_model.fc_output_layer.weight", {2, 384}),
                                                   &sd.get("sb_model.fc_output_layer.bias", {2})),
                                    ACT_NONE, {}, S);
    }

    void plan_buffers(int B, int T) override {
        cur.B = 0;
        bufs(B, T);
    }

    void forward(const float* in, const int64_t* shape, int ndim, float* out, hipStream_t st) override {
        SE_CHECK(ndim == 4 && shape[1] == 1 && shape[2] == NBIN, "FullSubNet forward expects [B,1,257,T]");
        const int B = (int)shape[0], T = (int)shape[3];
        Bufs& b = bufs(B, T);
        network(b, in, st);
        hipLaunchKernelGGL(fsn_mask_kernel, dim3((T + 255) / 256, NBIN, B), dim3(256), 0, st, b.maskBT, nullptr, out, B, T, 1.f);
        SE_HIP(hipGetLastError());
    }

    void enhance(const float* wav, long pitch, int B, int L, float* out, long out_pitch, hipStream_t st) override {
        const int T = 1 + L / HOP;
        Bufs& b = bufs(B, T);
        launch_rms_scale(wav, B, L, pitch, b.c, st);                                               // :38-39
        launch_stft(ctx.geom, wav, pitch, B, L, L, b.c, ctx.p_in, b.spec, b.mag, T, T, st);        // :46-54
        network(b, b.mag, st);                                                                     // :56
        hipLaunchKernelGGL(fsn_mask_kernel, dim3((T + 255) / 256, NBIN, B), dim3(256), 0, st, b.maskBT, b.spec, b.est, B, T,
                           ctx.p_out);                                                             // :57-66
        SE_HIP(hipGetLastError());
        launch_istft(ctx.geom, b.est, B, T, T, b.frames, b.c, out, out_pitch, L, st);              // :69-72
    }

    // ---- frame-online mode (model.h), with the cumulative norm only: every operator is then causal and the network looks
    // look_ahead = 2 frames ahead (model.py:79, :117 - the mask of frame t is the output of step t + 2).  A chunk of n new
    // frames runs n network steps and finalises the estimate of frames [t0 - 2, t0 + n - 2) (stream_lag); the chunk that ends
    // the stream runs two more steps on the zero frames the offline forward pads.  State: the two running sums of the norms
    // (per utterance / per sub-band sequence), (h, c) of the four LSTM layers, the last columns of spectrum and estimate.
    static constexpr int STREAM_TP_MAX = 64;      // longest window (frames incl. look-ahead) whose sub-band gate tensor is always kept
    bool stream_supported() const override { return cum && !is_gru; }
    int stream_lag() const override { return LA; }
    void stream_begin(int B, int max_chunk, hipStream_t st) override {
        SE_CHECK(STREAM_HC + max_chunk + LA <= STREAM_TP_MAX, "FullSubNet frame-online mode: chunks of at most 58 frames");
        ss.release();
        ss.B = B;
        ss.first = true;
        const size_t S = (size_t)NBIN * B;
        ss.hist.push_back(ss.zeros((size_t)B * 2 * NBIN * STREAM_HC, st));      // spectrum
        ss.hist.push_back(ss.zeros((size_t)B * 2 * NBIN * STREAM_HC, st));      // estimate
        ss.hist.push_back(ss.zeros((size_t)B, st));                             // running sum of the full-band norm
        ss.hist.push_back(ss.zeros(S, st));                                     // ... of the sub-band norm
        for (int l = 0; l < 2; ++l) {
            ss.h[l] = ss.zeros((size_t)512 * B, st);
            ss.c[l] = ss.zeros((size_t)512 * B, st);
            ss.h[2 + l] = ss.zeros(384 * S, st);
            ss.c[2 + l] = ss.zeros(384 * S, st);
        }
    }
    void stream_bufs(int B, int n, float** spec, float** mag, float** est) override {
        Bufs& b = bufs(B, STREAM_HC + n);
        *spec = b.spec;
        *mag = b.mag;
        *est = b.est;
    }
    void stream_chunk(int B, int t0, int n, hipStream_t st, bool last) override {
        SE_CHECK(ss.B == B && ss.hist.size() == 4, "stream_chunk without stream_begin");
        const int HC = STREAM_HC, Tw = HC + n, S = NBIN * B;
        const int ns = n + (last ? LA : 0);                 // network steps of this chunk
        SE_CHECK(Tw + LA <= STREAM_TP_MAX, "FullSubNet frame-online mode: chunk too long");
        Bufs& b = bufs(B, Tw);
        Profiler* pf = &ctx.prof;
        launch_hist_restore(b.spec, ss.hist[0], B, 2L * NBIN, Tw, HC, st);
        launch_hist_restore(b.est, ss.hist[1], B, 2L * NBIN, Tw, HC, st);
        // magnitudes of the new frames, time-major [ns][257][B]; at the end of the stream two zero frames (model.py:79)
        launch_transpose_akt(b.mag + HC, b.magT, B, NBIN, n, (long)NBIN * Tw, Tw, (long)NBIN * B, B, st);
        if (last) launch_fill(b.magT + (size_t)n * S, (long)LA * S, 0.f, st);
        hipLaunchKernelGGL(fsn_cum_fb_kernel, dim3(B), dim3(256), 0, st, b.magT, b.xfb, ns, B, ss.hist[2], t0);
        fb[0].run_stream(b.xfb, b.G, ss.c[0], ss.h[0], b.h[0], ns, B, ss.first, st, pf);
        fb[1].run_stream(b.h[0], b.G, ss.c[1], ss.h[1], b.h[1], ns, B, ss.first, st, pf);
        run_pointwise(fb_fc, b.h[1], 512L * B, B, b.fbo, (long)NBIN * B, B, ns, B, st, pf);
        hipLaunchKernelGGL(fsn_build_sb_kernel, dim3((S + 255) / 256, SBW, ns), dim3(256), 0, st, b.magT, b.fbo, b.sb, B);
        hipLaunchKernelGGL(fsn_cum_sb_kernel, dim3((S + 255) / 256), dim3(256), 0, st, b.sb, ns, S, ss.hist[3], t0);
        sbl[0].run_stream(b.sb, b.G, ss.c[2], ss.h[2], b.h[0], ns, S, ss.first, st, pf);
        sbl[1].run_stream(b.h[0], b.G, ss.c[3], ss.h[3], b.h[1], ns, S, ss.first, st, pf);
        run_pointwise(sb_fc, b.h[1], 384L * S, S, b.maskT, 2L * S, S, ns, S, st, pf);
        hipLaunchKernelGGL(fsn_stream_apply_kernel, dim3((S + 255) / 256, ns), dim3(256), 0, st, b.maskT, b.spec, b.est, B, Tw, ns,
                           HC - LA, t0, ctx.p_out);
        SE_HIP(hipGetLastError());
        launch_hist_save(b.spec, ss.hist[0], B, 2L * NBIN, Tw, HC, st);
        launch_hist_save(b.est, ss.hist[1], B, 2L * NBIN, Tw, HC, st);
        ss.first = false;
    }

  private:
    StreamState ss;
    bool cum = false, is_gru = false;
    struct Bufs {
        int B = 0, T = 0;
        float *c, *spec, *mag, *est, *frames, *mu, *mu2, *part;
        float *magT, *xfb, *G, *h[2], *cell, *fbo, *sb, *maskT, *maskBT, *hz;
    } cur;
    LstmBig fb[2], sbl[2];
    GCPlan fb_fc, sb_fc;

    Bufs& bufs(int B, int T) {
        if (cur.B == B && cur.T == T) return cur;
        Arena& a = ctx.arena;
        a.reset();
        Bufs b;
        b.B = B;
        b.T = T;
        const size_t BT = (size_t)B * T, Tp = T + LA, S = (size_t)NBIN * B;
        b.c = a.alloc_f(B);
        b.mu = a.alloc_f(B);
        b.mu2 = a.alloc_f(B);
        b.part = a.alloc_f((size_t)FSN_SUM_BLOCKS * B);
        b.spec = a.alloc_f(BT * 2 * NBIN);
        b.mag = a.alloc_f(BT * NBIN);
        b.est = a.alloc_f(BT * 2 * NBIN);
        b.frames = nullptr;      // the fused iSTFT keeps its frames in LDS (k_stft.hip); kept in the struct for the launcher signature
        b.magT = a.alloc_f(Tp * S);
        b.xfb = a.alloc_f(Tp * S);
        b.fbo = a.alloc_f(Tp * S);
        b.sb = a.alloc_f(Tp * SBW * S);
        // gate pre-activations: the full-band layers' [Tp][2048][B]; the sub-band layers' [Tp][1536][S] only when they do not
        // project their inputs inside the step GEMM (51 GB at 128 clips)
        // (a frame-online window - a few frames - always keeps them: the streamed steps go through LstmBig::run_stream)
        const bool sb_gates = !(fuse_x_on(B) && sbl[0].has_x && sbl[1].has_x) || (cum && Tp <= STREAM_TP_MAX);
        b.G = a.alloc_f(sb_gates ? Tp * 1536 * S : Tp * 2048 * (size_t)B);
        b.h[0] = a.alloc_f(Tp * 384 * S);               // also the full-band hidden [Tp][512][B]
        b.h[1] = a.alloc_f(Tp * 384 * S);
        b.cell = a.alloc_f(384 * S + 512 * (size_t)B);
        b.hz = a.alloc_f(384 * S);                      // zeros: h_{-1} of the sub-band layer that projects its own input
        b.maskT = a.alloc_f(Tp * 2 * S);
        b.maskBT = a.alloc_f(Tp * 2 * S);
        cur = b;
        return cur;
    }

    bool graph_capturable() const override { return false; }     // sub-band halves run on two streams
    // SE_FSN_FUSE_X=0: the sub-band layers' input projections as batched GEMMs into a [T][4H][S] gate tensor (rnn.h step_x)
    // (from 16 clips on: below, a step launch is a latency-bound K loop of a few workgroups and a K of 768 instead of 384 costs
    // more than the batched projection it replaces - one clip: 38 vs 30 utt/s)
    static bool fuse_x_on(int B) {
        static const bool on = !(getenv("SE_FSN_FUSE_X") && atoi(getenv("SE_FSN_FUSE_X")) == 0);
        return on && B >= 16;
    }

    // mag [B][257][T] -> maskBT [n*B+b][2][T+2]
    void network(Bufs& b, const float* mag, hipStream_t st) {
        const int B = b.B, T = b.T, Tp = T + LA, S = NBIN * B;
        Profiler* pf = &ctx.prof;
        // ---- full-band model (model.py:84-85): utterance-mean normalisation, LSTM(257->512)x2, Linear + ReLU
        const Ragged* rg = ragged_ctx();
        const int* tlen = rg ? rg->tlen : nullptr;
        if (!cum) hipLaunchKernelGGL(fsn_mean_kernel, dim3(B), dim3(256), 0, st, mag, NBIN * T, (float)NBIN * Tp, b.mu, tlen);
        launch_fill(b.magT + (size_t)T * S, (long)LA * S, 0.f, st);                                // look-ahead pad :79 (a kernel, not a memset node: graph-replay safe)
        launch_transpose_akt(mag, b.magT, B, NBIN, T, (long)NBIN * T, T, (long)NBIN * B, B, st);
        const long nfb = (long)Tp * S;
        // (cumulative norm, ragged rows: frames >= tlen[b] are zeros - the STFT wrote them - and everything is causal, so a
        // row's own frames and its two look-ahead frames see exactly what the clip decoded alone sees)
        if (cum) hipLaunchKernelGGL(fsn_cum_fb_kernel, dim3(B), dim3(256), 0, st, b.magT, b.xfb, Tp, B, (float*)nullptr, 0);
        else hipLaunchKernelGGL(fsn_scale_kernel, dim3((unsigned)((nfb + 255) / 256)), dim3(256), 0, st, b.magT, b.xfb, nfb, B, b.mu);
        fb[0].run(b.xfb, b.G, b.cell, b.h[0], Tp, B, st, pf);
        fb[1].run(b.h[0], b.G, b.cell, b.h[1], Tp, B, st, pf);
        run_pointwise(fb_fc, b.h[1], 512L * B, B, b.fbo, (long)NBIN * B, B, Tp, B, st, pf);
        // ---- sub-band input (:88-97): unfold(noisy, 15) ++ unfold(fb_out, 0), normalised by its utterance mean
        hipLaunchKernelGGL(fsn_build_sb_kernel, dim3((S + 255) / 256, SBW, Tp), dim3(256), 0, st, b.magT, b.fbo, b.sb, B);
        const long rows = (long)Tp * SBW * NBIN;
        SE_CHECK(B <= 256, "FullSubNet batch per call is limited to 256 utterances");
        if (cum) {
            hipLaunchKernelGGL(fsn_cum_sb_kernel, dim3((S + 255) / 256), dim3(256), 0, st, b.sb, Tp, S, (float*)nullptr, 0);
        } else {
            hipLaunchKernelGGL(fsn_colsum_kernel, dim3(FSN_SUM_BLOCKS), dim3(256), 0, st, b.sb, rows, B, b.part, tlen);
            hipLaunchKernelGGL(fsn_finish_mean_kernel, dim3(1), dim3(256), 0, st, b.part, b.mu2, (float)rows, B, tlen);
            const long nsb = rows * B;
            hipLaunchKernelGGL(fsn_scale_kernel, dim3((unsigned)((nsb + 255) / 256)), dim3(256), 0, st, b.sb, b.sb, nsb, B, b.mu2);
        }
        SE_HIP(hipGetLastError());
        // ---- sub-band model (:106-114): LSTM(32->384)x2 over 257*B sequences, Linear(384->2)
        // The 257 * B sequences are independent: two column halves run on two streams, so that the short per-step
        // launches of one half (2 block rounds, every block in the same phase) overlap the other half's instead of
        // leaving the matrix pipes idle during everybody's prologue / epilogue.
        float* cell = b.cell + 512 * (size_t)B;
        static const int parts_env = getenv("SE_FSN_SPLIT") ? atoi(getenv("SE_FSN_SPLIT")) : 2;
        const int parts = std::max(1, std::min({parts_env, 1 + EngineCtx::MAX_AUX, S / 256}));
        const int Sp = ((S + parts - 1) / parts + 127) / 128 * 128;           // columns per part (tile aligned)
        const bool fuse_x = fuse_x_on(B);
        const bool l0x = fuse_x && sbl[0].has_x;
        if (l0x) launch_fill(b.hz, 384L * S, 0.f, st);
        auto part = [&](int c0, int Sn, hipStream_t s, Profiler* p) {
            if (l0x) sbl[0].run_cols_x(b.sb, (long)SBW * S, cell, b.hz, b.h[0], 384L * S, 1, Tp, S, c0, Sn, s, p);
            else sbl[0].run_cols(b.sb, (long)SBW * S, b.G, cell, b.h[0], 384L * S, 1, Tp, S, c0, Sn, s, p);
            if (fuse_x && sbl[1].has_x) sbl[1].run_cols_x(b.h[0], 384L * S, cell, b.hz, b.h[1], 384L * S, 1, Tp, S, c0, Sn, s, p);
            else sbl[1].run_cols(b.h[0], 384L * S, b.G, cell, b.h[1], 384L * S, 1, Tp, S, c0, Sn, s, p);
            run_pointwise(sb_fc, b.h[1] + c0, 384L * S, S, b.maskT + c0, 2L * S, S, Tp, Sn, s, p);
        };
        if (parts > 1) {
            for (int i = 1; i < parts; ++i) (void)ctx.aux_stream(i - 1);
            SE_HIP(hipEventRecord(ctx.ev_fork, st));
            for (int i = 1; i < parts; ++i) {
                const int c0 = i * Sp, Sn = std::min(Sp, S - c0);
                if (Sn <= 0) break;
                SE_HIP(hipStreamWaitEvent(ctx.aux[i - 1], ctx.ev_fork, 0));
                part(c0, Sn, ctx.aux[i - 1], &ctx.aux_prof[i - 1]);
                SE_HIP(hipEventRecord(ctx.ev_join[i - 1], ctx.aux[i - 1]));
            }
            part(0, std::min(Sp, S), st, pf);
            for (int i = 1; i < parts; ++i)
                if (i * Sp < S) SE_HIP(hipStreamWaitEvent(st, ctx.ev_join[i - 1], 0));
        } else {
            part(0, S, st, pf);
        }
        // [Tp][2][S] -> [S][2][Tp]
        launch_transpose_akt(b.maskT, b.maskBT, Tp, 2, S, 2L * S, S, 2L * Tp, Tp, st);
    }
};

}  // namespace

std::unique_ptr<Model> make_fullsubnet(EngineCtx& ctx) { return std::unique_ptr<Model>(new FullSubNet(ctx)); }

}  // namespace se
