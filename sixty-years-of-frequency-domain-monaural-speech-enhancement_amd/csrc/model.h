// Model interface of the engine: one subclass per reference model class (include/se_engine.h se_model_id).
#pragma once
#include "common.h"
#include "kernels.h"
#include "layers.h"
#include <memory>
#include <set>

namespace se {

struct EngineCtx {
    int max_batch = 1, max_samples = 64000;
    float p_in = 1.f, p_out = 1.f;
    int flags = 0;          // se_config.flags
    // hipGraph replay asked for (SE_CFG_GRAPHS, or SE_GRAPH=1 / 0 in the environment): models that would fork work onto an
    // auxiliary stream stay on the caller's stream then, so that the decode can be captured (graphs were silently
    // dropped for TaylorSENet, the most launch-heavy model)
    bool graphs_wanted() const {
        static const int graphs_env = getenv("SE_GRAPH") ? atoi(getenv("SE_GRAPH")) : -1;
        return graphs_env >= 0 ? graphs_env != 0 : (flags & 1 /* SE_CFG_GRAPHS */) != 0;
    }
    // SE_CFG_REPEATS(n) (include/se_engine.h): gaf_base's stage_num / TaylorSENet's order_num; `dflt` = the decode script's
    int repeats(int dflt) const {
        const int v = (flags >> 8) & 15;
        return v ? v - 1 : dflt;
    }
    StftGeom geom{0, 0, 0};
    Arena arena;
    Profiler prof;
    Profiler aux_prof[3];   // one per auxiliary stream (events are recorded on the stream they time); se_get_profile sums all
    StageProf stage_prof;   // HBM-bound front / back-end kernels (se_get_stage_profile)
    void prof_reset() {
        prof.reset();
        for (auto& p : aux_prof) p.reset();
        stage_prof.reset();
    }
    void prof_set(bool on) {
        prof.on = on;
        for (auto& p : aux_prof) p.on = on;
    }
    // second stream for independent sub-problems of one call (fork / join through events around the caller's stream)
    static constexpr int MAX_AUX = 3;
    hipStream_t aux[MAX_AUX] = {};
    hipEvent_t ev_fork = nullptr, ev_join[MAX_AUX] = {};
    // The auxiliary streams are PROCESS-wide (one set per device, created on first use, never destroyed); the events are the
    // engine's own.  Round 6: with a set per engine, a process that builds one engine after the other (bench.py's zoo, tools/sweep.py)
    // saw every engine that forks after G2Net's three-stream fork slow down - TaylorSENet 2 377 -> 2 275 utt/s at batch 256, one clip
    // 4.6 -> 13 ms - streams created behind destroyed ones did not run side by side with the caller's any more.  Work of two
    // engines that share a stream is ordered by each engine's own fork / join events; the rest is false sharing at worst.
    static hipStream_t shared_aux(int dev, int i) {
        static std::mutex mu;
        static hipStream_t pool[16][MAX_AUX] = {};
        SE_CHECK(dev >= 0 && dev < 16, "device index");
        std::lock_guard<std::mutex> lk(mu);
        if (!pool[dev][i]) SE_HIP(hipStreamCreateWithFlags(&pool[dev][i], hipStreamNonBlocking));
        return pool[dev][i];
    }
    hipStream_t aux_stream(int i) {
        SE_CHECK(i >= 0 && i < MAX_AUX, "aux stream index");
        if (!ev_fork) SE_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        if (!aux[i]) {
            int dev = 0;
            SE_HIP(hipGetDevice(&dev));
            aux[i] = shared_aux(dev, i);
            SE_HIP(hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming));
        }
        return aux[i];
    }
    int* eq_rows = nullptr;     // device [4][max_batch]: per-row sizes of an equal-length batch decoded as ragged rows (PadFrames)
    ~EngineCtx() {
        if (eq_rows) (void)hipFree(eq_rows);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        for (int i = 0; i < MAX_AUX; ++i)
            if (aux[i]) {
                (void)hipStreamSynchronize(aux[i]);      // (shared with other engines: only this engine's work is waited for in effect)
                (void)hipEventDestroy(ev_join[i]);
            }
    }
};

// State dict with use tracking, so finalize can reject unexpected keys like a strict load_state_dict.
struct TrackedSD {
    const StateDict& sd;
    mutable std::set<std::string> used;
    explicit TrackedSD(const StateDict& s) : sd(s) {}
    const HostTensor& get(const std::string& k, std::vector<int64_t> shape = {}) const {
        used.insert(k);
        return sd_get(sd, k, std::move(shape));
    }
    bool has(const std::string& k) const { return sd.count(k) != 0; }
    void check_all_used() const {
        for (const auto& kv : sd) {
            const std::string& k = kv.first;
            const bool nbt = k.size() >= 19 && k.compare(k.size() - 19, 19, "num_batches_tracked") == 0;
            SE_CHECK(nbt || used.count(k), "unexpected key in state dict: '" + k + "'");
        }
    }
};

// Models with operators that look ahead in time stage 16 B groups only over rows of whole groups (gemmconv: a group that
// straddles the end of a row would feed the next row's head into stored frames).  Their equal-length batches therefore run
// with the frame count rounded up to a multiple of 4 and the added frames treated like the tail of a shorter clip in a
// ragged batch - zeroed in front of every look-ahead operator, masked in attention - by publishing per-row sizes that
// are all equal (measured against trimming the straddling groups in LDS: DCCRN + 2 %, Uformer + 1.7 % at batch 256).
// The multiple itself: 4 (whole 16 B groups).  Uformer rounds to 16 - 401 -> 416 frames are rows of 1 664 B = 13 whole 128 B
// lines: 3.7 % more frames and still + 1.1 % at batch 256 (its many low-channel layers and elementwise passes move whole
// lines); DCCRN (501 -> 512 against 504) gained nothing over 4 in rounds 3-5; with round 6's epilogue it does (2 669 / 2 672 -> 2 697 utt/s
// at batch 256: rows of 2 048 B, whole tiles) and rounds to 16 too.  SE_PAD_FRAMES_TO=n overrides both.
inline int pad_frames_mult(int model_default = 4) {
    static const int m = getenv("SE_PAD_FRAMES_TO") ? std::max(4, atoi(getenv("SE_PAD_FRAMES_TO")) & ~3) : 0;
    return m ? m : model_default;
}
// The cLN (`_new`) variants are causal end to end - every InstanceNorm of the base directories is a CumulativeLayerNorm
// (CTSNet_new/Step1_network.py:213-286) - so frames behind a clip's last one never reach a frame the iSTFT reads: an
// equal-length offline batch runs with its rows zero-extended to whole 128 B lines (T = 401 -> 416).  Rows of 401 floats put
// three of four rows off 16 B alignment: every 16 B staging group and every 16 B store of the conv family straddles its
// segment (round 6, `gcbench ... 401` vs `416`: the store-bound 2 -> 64 layer 1.11 -> 0.84 ms, the memory pipeline of a
// 64 -> 64 layer alone - SE_GC_DBG=4 - 1.17 -> 0.81 ms).  SE_CLN_PAD=n: multiple (default 32, 1 = off).
inline int causal_work_frames(int T, bool causal_all) {
    static const int m = getenv("SE_CLN_PAD") ? std::max(1, atoi(getenv("SE_CLN_PAD"))) : 32;
    // (ragged calls too: the rows' own lengths travel in the ragged context, the STFT writes zeros behind each row's last frame)
    if (!causal_all || m <= 1 || stream_ctx()) return T;
    return (T + m - 1) / m * m;
}
// The InstanceNorm flavours of the same networks cannot be zero-extended (the statistics run over the whole utterance): their
// equal-length batches run as ragged rows of ONE length (PadFrames below, multiple SE_IN_PAD, default 32, 1 = off) - every
// reduction over time stops at the row's own frame count (the norm kernels, the fused TCM block and, since round 6, the conv
// epilogue's statistics riders), everything else is causal.
inline int in_pad_multiple() {
    static const int m = getenv("SE_IN_PAD") ? std::max(1, atoi(getenv("SE_IN_PAD"))) : 32;
    return m <= 1 ? 1 : std::max(4, (m + 3) & ~3);
}
inline int causal_frame_multiple(bool causal_all) {
    static const int m = getenv("SE_CLN_PAD") ? std::max(1, atoi(getenv("SE_CLN_PAD"))) : 32;
    return std::max(4, causal_all ? ((m + 3) & ~3) : in_pad_multiple());
}
struct PadFrames {
    Ragged rg;
    bool on = false;
    int T;                  // frame count (row pitch) to run with
    PadFrames(EngineCtx& ctx, int B, int L, int Lpad, int T_true, int olen, hipStream_t st, int mult = 4) : T(T_true) {
        static const bool env = !(getenv("SE_PAD_FRAMES") && atoi(getenv("SE_PAD_FRAMES")) == 0);
        const int m = mult == 1 ? 1 : pad_frames_mult(mult);      // (mult 1: never pad - the caller's switch is off)
        if (!env || T_true % m == 0) return;
        T = (T_true + m - 1) / m * m;
        if (ragged_ctx()) return;          // rows of different lengths already carry their sizes
        const int MB = ctx.max_batch;
        if (!ctx.eq_rows) SE_HIP(hipMalloc(reinterpret_cast<void**>(&ctx.eq_rows), sizeof(int) * 4 * MB));
        launch_fill_rows(ctx.eq_rows, MB, B, L, Lpad, T_true, olen, st);
        rg = Ragged{ctx.eq_rows, ctx.eq_rows + MB, ctx.eq_rows + 2 * MB, ctx.eq_rows + 3 * MB};
        set_ragged_ctx(&rg);
        on = true;
        pctx = &ctx;
        set_fscale((double)T_true / T);        // every launch of these networks is linear in the frame count
    }
    ~PadFrames() {
        if (on) {
            set_ragged_ctx(nullptr);
            set_fscale(1.0);
        }
    }
    EngineCtx* pctx = nullptr;
    void set_fscale(double v) {
        pctx->prof.fscale = v;
        for (auto& p : pctx->aux_prof) p.fscale = v;
    }
    PadFrames(const PadFrames&) = delete;
    PadFrames& operator=(const PadFrames&) = delete;
};

class Model {
  public:
    explicit Model(EngineCtx& c) : ctx(c) {}
    virtual ~Model() {}
    virtual StftGeom default_geom() const = 0;
    virtual void finalize(const TrackedSD& sd) = 0;
    // y = model(x) in the reference's tensor layout
    virtual void forward(const float* in, const int64_t* shape, int ndim, float* out, hipStream_t st) = 0;
    // Uformer only: the full 4-tuple of `model(inputs, src)` (uformer.py:287); any of src / the three extra outputs may be null
    virtual void forward_uformer(const float* inputs, const float* src, int B, int L, float* output, float* src_out,
                                 float* output_cplx, float* src_cplx, hipStream_t st) {
        SE_CHECK(false, "se_uformer_forward: the engine was not created with SE_MODEL_UFORMER");
    }
    // body of enhance(args) for B equal-length clips
    virtual void enhance(const float* wav, long pitch, int B, int L, float* out, long out_pitch, hipStream_t st) = 0;
    // carve the activation workspace for (B clips, T frames) out of ctx.arena (also used to size it)
    virtual void plan_buffers(int B, int T) = 0;
    // frame counts (row pitches) this model's workspace is planned for: multiples of
    virtual int frame_multiple() const { return 4; }
    virtual int64_t output_samples(int L) const { return L; }
    // samples the STFT sees (decode scripts that tail-pad to a hop multiple)
    virtual int padded_samples(int L) const { return L; }
    int num_frames(int L) const { return 1 + padded_samples(L) / ctx.geom.hop; }
    // false: the model has operators that look ahead in time and does not zero / mask what lies past a row's own last frame,
    // so rows of different lengths cannot share a call (every model of the zoo does: DCCRN and Uformer through
    // launch_zero_tail / key masks)
    virtual bool ragged_supported() const { return true; }
    // ---- frame-online decoding (se_stream_*): the causal models carry their state (one history frame per conv layer,
    // LSTM (h, c), the iSTFT's overlap) across calls instead of seeing the whole utterance.  stream_chunk() handles frames
    // [t0, t0 + n) of B parallel streams: the engine has written their STFT into columns [STREAM_HC, STREAM_HC + n) of
    // stream_spec() / stream_mag() (row pitch STREAM_HC + n) and reads the estimate from stream_est() in the same layout.
    static constexpr int STREAM_HC = 4;     // history columns in front of every chunk tensor (convs look back 1 frame, the iSTFT 1;
                                            // a multiple of 4: the convs start at frame STREAM_HC, gemmconv.hip gc_launch)
    // models whose front end overlaps more frames, or whose network looks a bounded number of frames AHEAD (DCCRN's decoder:
    // one frame per transposed conv), keep more history and finalise their estimate `stream_lag()` frames late: the chunk
    // tensors then start stream_hc() frames before the new ones and the last stream_lag() estimate frames of a chunk are
    // provisional (recomputed by the next chunk; final at the end of the stream, where "no future" is the truth)
    virtual int stream_hc() const { return STREAM_HC; }
    virtual int stream_lag() const { return 0; }
    virtual bool stream_supported() const { return false; }
    virtual void stream_begin(int B, int max_chunk, hipStream_t st) { SE_CHECK(false, "this model has no streaming mode"); }
    virtual void stream_bufs(int B, int n, float** spec, float** mag, float** est) { SE_CHECK(false, "no streaming mode"); }
    // (`last`: no frame follows this chunk - a model that looks ahead finalises its provisional frames)
    virtual void stream_chunk(int B, int t0, int n, hipStream_t st, bool last) { SE_CHECK(false, "no streaming mode"); }
    // false: enhance() forks onto auxiliary streams and is not replayed from a captured hipGraph (SE_CFG_GRAPHS)
    virtual bool graph_capturable() const { return true; }

  protected:
    EngineCtx& ctx;
};

std::unique_ptr<Model> make_dccrn(EngineCtx& ctx);
std::unique_ptr<Model> make_crn(EngineCtx& ctx);
std::unique_ptr<Model> make_lstm(EngineCtx& ctx);
std::unique_ptr<Model> make_dpcrn(EngineCtx& ctx);
std::unique_ptr<Model> make_fullsubnet(EngineCtx& ctx);
std::unique_ptr<Model> make_gcrn(EngineCtx& ctx);
std::unique_ptr<Model> make_ctsnet(EngineCtx& ctx);
std::unique_ptr<Model> make_taylorsenet(EngineCtx& ctx);
std::unique_ptr<Model> make_g2net(EngineCtx& ctx);
std::unique_ptr<Model> make_uformer(EngineCtx& ctx);

}  // namespace se
