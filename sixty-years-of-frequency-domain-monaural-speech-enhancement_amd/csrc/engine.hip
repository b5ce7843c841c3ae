// C ABI of the engine (include/se_engine.h): handle, strict state-dict load, stage hooks.
#include "../../include/se_engine.h"
#include "model.h"
#include <algorithm>
#include <cstring>
#include <mutex>

using namespace se;

struct se_engine {
    se_config cfg{};
    EngineCtx ctx;
    StateDict sd;
    std::unique_ptr<Model> model;
    bool finalized = false;
    int plan_frames = 0;     // frames the workspace was sized for (>= frames of max_samples; frame-online windows may be longer)
    std::string err;
    // hipGraph replay of se_enhance_batch (SE_CFG_GRAPHS): one instantiated graph per (batch, samples) shape, captured
    // on the second call of a shape (the first, eager call sets kernel attributes and grows the lazy scratch buffers);
    // caller buffers are decoupled from the graph by engine-owned staging rows
    struct GraphEntry {
        int batch, samples;
        hipGraphExec_t exec;     // nullptr: this shape could not be captured, stay eager
    };
    std::vector<GraphEntry> graphs;
    std::vector<std::pair<int, int>> warmed;
    float *stage_in = nullptr, *stage_out = nullptr;
    float* hook_buf = nullptr;     // se_backend: the decompressed spectrum between the mask stage and the iSTFT (grows, never shrinks)
    size_t hook_cap = 0;
    // se_enhance_ragged: per-row sizes (len | lpad | tlen | olen, max_batch ints each) go host -> device through a small
    // ring of pinned slots, so back-to-back calls never wait for each other's copy
    static constexpr int RAG_SLOTS = 8;
    int* rag_host = nullptr;       // pinned [RAG_SLOTS][4 * max_batch]
    int* rag_dev = nullptr;        // device [RAG_SLOTS][4 * max_batch]
    hipEvent_t rag_ev[RAG_SLOTS] = {};
    int rag_next = 0;
    hipStream_t cap_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // Two half-batches side by side (round 6: Uformer, DPCRN, CTSNet, TaylorSENet): a second instance of the model with its own context and
    // workspace decodes rows [B / 2, B) of a se_enhance_batch call on a process-wide auxiliary stream while the first decodes
    // rows [0, B / 2) on the caller's - a decode is ~400 launches of very different shapes, two of them in flight fill each
    // other's tails and launch gaps (Uformer + 3 %, DPCRN + 2 %, CTSNet + 2.9 % at batch 256; rows are independent, the results are the
    // rows' own).  Not under the profiler (per-launch durations of concurrent launches are not what a roofline prices), not
    // for graph replay, ragged or frame-online calls.  SE_BATCH_SPLIT=0 / 1: off / on for every model.
    // (Uformer and DPCRN: THREE parts - + 1.5 % over two; CTSNet and TaylorSENet: two - a third loses for TaylorSENet)
    static constexpr int MAX_TWINS = 2;
    int ntwins = 0;
    std::unique_ptr<EngineCtx> ctx2[MAX_TWINS];
    std::unique_ptr<Model> twin[MAX_TWINS];
    hipEvent_t ev_tfork = nullptr, ev_tjoin[MAX_TWINS] = {};
    // se_stream_*: the samples received so far ([batch][max_samples]), frames transformed, samples emitted
    struct Stream {
        bool active = false;
        int batch = 0, max_chunk = 0, n_total = 0, t_done = 0, o_done = 0;
        int carve_B = -1, carve_n = -1;      // (batch, chunk frames) the arena was last carved and zero-filled for
        // running unit-RMS scale (se_stream_begin_running): sum of squares so far per stream, 1 / c per frame in a ring
        bool running = false;
        double* sumsq = nullptr;
        float* frame_inv = nullptr;
        int ring = 0;
        float* wav = nullptr;
        float* c = nullptr;
        // the hipStream the last push / flush / begin of this handle ran on: a new stream that starts on ANOTHER hipStream
        // re-uses (and clears) the parked state buffers, so it must be ordered behind the work still in flight there
        bool has_last = false;
        hipStream_t last_st = nullptr;
        hipEvent_t ev_order = nullptr;
    } strm;
};

static std::string g_create_err;

template <typename F>
static int guard(se_engine* e, F&& f) {
    // the engine's device is current while the call runs; the caller's current device is restored on the way out
    int prev = -1;
    int rc = 0;
    try {
        if (e) {
            SE_HIP(hipGetDevice(&prev));
            if (prev != e->cfg.device) SE_HIP(hipSetDevice(e->cfg.device));
            else prev = -1;
        }
        f();
    } catch (const std::exception& ex) {
        if (e) e->err = ex.what();
        else g_create_err = ex.what();
        rc = 1;
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}

// the stage profiler is visible to the launchers only while a profiled call enqueues work
struct StageProfScope {
    explicit StageProfScope(se_engine* e) { set_stage_prof(e->ctx.prof.on ? &e->ctx.stage_prof : nullptr); }
    ~StageProfScope() { set_stage_prof(nullptr); }
};

extern "C" {

int64_t se_resample_samples(int32_t n_in, int32_t sr_in, int32_t sr_out) {
    if (n_in <= 0 || sr_in <= 0 || sr_out <= 0) return -1;
    return se::resample_out_samples(n_in, sr_in, sr_out);
}

int se_resample(const float* in_dev, int64_t in_pitch, int32_t batch, int32_t n_in, int32_t sr_in, int32_t sr_out,
                float* out_dev, int64_t out_pitch, void* stream) {
    return guard(nullptr, [&] {         // errors are reported through se_last_error(NULL)
        SE_CHECK(in_dev && out_dev, "null argument");
        SE_CHECK(in_pitch >= n_in && out_pitch >= se::resample_out_samples(n_in, sr_in, sr_out), "row pitch too small");
        se::launch_resample(in_dev, in_pitch, batch, n_in, sr_in, sr_out, out_dev, out_pitch, static_cast<hipStream_t>(stream));
    });
}

int se_pcm16_decode(const int16_t* in_dev, int64_t in_pitch, int32_t batch, int32_t n, float* out_dev, int64_t out_pitch,
                    void* stream) {
    return guard(nullptr, [&] {
        SE_CHECK(in_dev && out_dev, "null argument");
        SE_CHECK(batch == 1 || (in_pitch >= n && out_pitch >= n), "row pitch too small");
        se::launch_pcm16_decode(in_dev, in_pitch, batch, n, out_dev, out_pitch, static_cast<hipStream_t>(stream));
    });
}

int se_pcm16_encode(const float* in_dev, int64_t in_pitch, int32_t batch, int32_t n, int16_t* out_dev, int64_t out_pitch,
                    void* stream) {
    return guard(nullptr, [&] {
        SE_CHECK(in_dev && out_dev, "null argument");
        SE_CHECK(batch == 1 || (in_pitch >= n && out_pitch >= n), "row pitch too small");
        se::launch_pcm16_encode(in_dev, in_pitch, batch, n, out_dev, out_pitch, static_cast<hipStream_t>(stream));
    });
}

int32_t se_abi_version(void) { return 5; }

const char* se_last_error(const se_engine* e) { return e ? e->err.c_str() : g_create_err.c_str(); }

static std::unique_ptr<Model> make_model(int id, EngineCtx& c) {
    switch (id) {
        case SE_MODEL_DCCRN: return make_dccrn(c);
        case SE_MODEL_CRN: return make_crn(c);
        case SE_MODEL_LSTM: return make_lstm(c);
        case SE_MODEL_DPCRN: return make_dpcrn(c);
        case SE_MODEL_GCRN: return make_gcrn(c);
        case SE_MODEL_CTSNET: return make_ctsnet(c);
        case SE_MODEL_TAYLORSENET: return make_taylorsenet(c);
        case SE_MODEL_G2NET: return make_g2net(c);
        case SE_MODEL_UFORMER: return make_uformer(c);
        case SE_MODEL_FULLSUBNET: return make_fullsubnet(c);
        default: SE_CHECK(false, "model id " + std::to_string(id) + " is not built into this engine yet");
    }
    return nullptr;
}
// The stream the second half-batch runs on: the first of the process-wide auxiliary streams (model.h) - the models that split
// do not fork, so it is free during their decodes.  NOT a stream of its own: the runtime maps streams onto four hardware queues
// by default; with a fifth stream in the process (caller's + three auxiliary + this one) two of them shared a queue and G2Net's
// three-stream fork, built later in the same process, ran slower than on one stream (0.448 -> 0.417 in bench.py's zoo).
static hipStream_t twin_stream(int dev, int i = 0) { return EngineCtx::shared_aux(dev, i); }

int se_engine_create(const se_config* cfg, se_engine** out) {
    se_engine* e = nullptr;
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    int rc = guard(nullptr, [&] {
        SE_CHECK(cfg && out, "null argument");
        int ndev = 0;
        SE_HIP(hipGetDeviceCount(&ndev));
        SE_CHECK(ndev > 0, "no HIP device visible: the engine has no CPU fallback");
        SE_CHECK(cfg->device >= 0 && cfg->device < ndev, "device ordinal out of range");
        SE_HIP(hipSetDevice(cfg->device));
        hipDeviceProp_t prop;
        SE_HIP(hipGetDeviceProperties(&prop, cfg->device));
        SE_CHECK(std::string(prop.gcnArchName).rfind("gfx950", 0) == 0,
                 std::string("built for gfx950 (MI355X) only, device is ") + prop.gcnArchName);
        e = new se_engine();
        e->cfg = *cfg;
        e->ctx.max_batch = cfg->max_batch > 0 ? cfg->max_batch : 1;
        e->ctx.max_samples = cfg->max_samples > 0 ? cfg->max_samples : 64000;
        e->ctx.p_in = cfg->p_in > 0.f ? cfg->p_in : 1.f;
        e->ctx.p_out = cfg->p_out > 0.f ? cfg->p_out : 1.f;
        e->ctx.flags = cfg->flags;
        e->model = make_model(cfg->model, e->ctx);
        e->ctx.geom = e->model->default_geom();
        if (cfg->n_fft > 0) {
            SE_CHECK(cfg->n_fft == e->ctx.geom.n_fft, "n_fft override must match the model's front end");
            e->ctx.geom = StftGeom{cfg->n_fft, cfg->hop > 0 ? cfg->hop : e->ctx.geom.hop, cfg->win > 0 ? cfg->win : cfg->n_fft};
        }
        static const int split_env = getenv("SE_BATCH_SPLIT") ? atoi(getenv("SE_BATCH_SPLIT")) : -1;
        const bool split = split_env >= 0 ? split_env != 0
                                          : (cfg->model == SE_MODEL_UFORMER || cfg->model == SE_MODEL_DPCRN || cfg->model == SE_MODEL_CTSNET ||
                                             cfg->model == SE_MODEL_TAYLORSENET);
        if (split && e->ctx.max_batch >= 64) {
            static const int parts_env = getenv("SE_BATCH_PARTS") ? atoi(getenv("SE_BATCH_PARTS")) : 0;
            const int parts = parts_env >= 2 && parts_env <= 1 + se_engine::MAX_TWINS
                                  ? parts_env
                                  : ((cfg->model == SE_MODEL_UFORMER || cfg->model == SE_MODEL_DPCRN) ? 3 : 2);
            e->ntwins = parts - 1;
            SE_HIP(hipEventCreateWithFlags(&e->ev_tfork, hipEventDisableTiming));
            for (int i = 0; i < e->ntwins; ++i) {
                e->ctx2[i].reset(new EngineCtx());
                e->ctx2[i]->max_batch = (e->ctx.max_batch + i + 1) / (i + 2);      // the first: a half (two-part calls), the second: a third
                e->ctx2[i]->max_samples = e->ctx.max_samples;
                e->ctx2[i]->p_in = e->ctx.p_in;
                e->ctx2[i]->p_out = e->ctx.p_out;
                e->ctx2[i]->flags = e->ctx.flags;
                e->twin[i] = make_model(cfg->model, *e->ctx2[i]);
                e->ctx2[i]->geom = e->ctx.geom;
                SE_HIP(hipEventCreateWithFlags(&e->ev_tjoin[i], hipEventDisableTiming));
            }
        }
        *out = e;
        scratch_engine_created(cfg->device);
    });
    if (rc && e) {
        delete e;
    }
    if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
    return rc;
}

int se_engine_destroy(se_engine* e) {
    if (!e) return 0;
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    for (auto& g : e->graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (e->cap_stream) {
        (void)hipStreamDestroy(e->cap_stream);
        (void)hipEventDestroy(e->ev_fork);
        (void)hipEventDestroy(e->ev_join);
    }
    if (e->strm.wav) (void)hipFree(e->strm.wav);
    if (e->strm.c) (void)hipFree(e->strm.c);
    if (e->strm.sumsq) (void)hipFree(e->strm.sumsq);
    if (e->strm.frame_inv) (void)hipFree(e->strm.frame_inv);
    if (e->strm.ev_order) (void)hipEventDestroy(e->strm.ev_order);
    if (e->rag_host) (void)hipHostFree(e->rag_host);
    if (e->rag_dev) (void)hipFree(e->rag_dev);
    for (auto& ev : e->rag_ev)
        if (ev) (void)hipEventDestroy(ev);
    if (e->hook_buf) (void)hipFree(e->hook_buf);
    if (e->stage_in) (void)hipFree(e->stage_in);
    if (e->stage_out) (void)hipFree(e->stage_out);
    if (e->ctx.arena.base()) gc_unregister_overread_range(e->ctx.arena.base());
    if (e->ev_tfork) (void)hipEventDestroy(e->ev_tfork);
    for (int i = 0; i < se_engine::MAX_TWINS; ++i) {
        if (e->ctx2[i] && e->ctx2[i]->arena.base()) gc_unregister_overread_range(e->ctx2[i]->arena.base());
        if (e->ev_tjoin[i]) (void)hipEventDestroy(e->ev_tjoin[i]);
        e->twin[i].reset();          // (the model before its context)
        e->ctx2[i].reset();
    }
    const int dev = e->cfg.device;
    delete e;
    scratch_engine_destroyed(dev);      // the device's last engine takes the engine-lifetime scratch slots with it
    if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
    return 0;
}

int se_engine_set_tensor(se_engine* e, const char* key, const void* data, const int64_t* shape, int32_t ndim,
                         int32_t dtype) {
    if (!e) return 1;
    return guard(e, [&] {
        SE_CHECK(!e->finalized, "set_tensor after finalize");
        SE_CHECK(key && (data || ndim == 0) && ndim >= 0 && ndim <= 8, "bad argument");
        HostTensor t;
        int64_t n = 1;
        for (int i = 0; i < ndim; ++i) {
            SE_CHECK(shape[i] >= 0, "negative dim");
            t.shape.push_back(shape[i]);
            n *= shape[i];
        }
        if (dtype == 0) {
            t.data.assign(static_cast<const float*>(data), static_cast<const float*>(data) + n);
        } else if (dtype == 1) {
            t.data.resize(n);
            for (int64_t i = 0; i < n; ++i) t.data[i] = (float)static_cast<const int64_t*>(data)[i];
        } else {
            SE_CHECK(false, "dtype must be 0 (float32) or 1 (int64)");
        }
        e->sd[key] = std::move(t);
    });
}

int se_engine_finalize(se_engine* e) {
    if (!e) return 1;
    return guard(e, [&] {
        SE_CHECK(!e->finalized, "already finalized");
        TrackedSD tsd(e->sd);
        e->model->finalize(tsd);
        tsd.check_all_used();
        // size the workspace for (max_batch, frames of max_samples)
        const int T = e->model->num_frames(e->ctx.max_samples);
        // (a frame-online window is the model's history columns + the chunk: models that keep a long history - CTSNet_new's
        // dilated convs reach 128 frames back - must be able to stream through an engine created for short clips)
        // (models that look ahead run with the frame count rounded up to whole 16 B groups: model.h PadFrames)
        const int fm = pad_frames_mult(e->model->frame_multiple());
        const int Tr = (T + fm - 1) / fm * fm;
        e->plan_frames = e->model->stream_supported() ? std::max(Tr, e->model->stream_hc() + 16) : Tr;
        // a model's layout may depend on the batch (FullSubNet keeps a [T][4H][S] gate tensor below 16 clips and none from 16
        // on, so 15 clips need more than 16...42): the arena covers every batch a call may bring, not only the largest
        // ... and a layout may depend on the window length: FullSubNet's frame-online windows (<= 64 frames) always keep the
        // sub-band gate tensor, at every batch - a stream of 32 rows on an engine made for short clips needs more for a
        // 58-frame chunk than the offline plan at plan_frames holds: the short-window plan is measured too
        size_t need = 0;
        for (int bq : {e->ctx.max_batch, std::min(e->ctx.max_batch, 15)})
            for (int tq : {e->plan_frames, std::min(e->plan_frames, 64)}) {
                if (tq != e->plan_frames && !e->model->stream_supported()) continue;
                e->ctx.arena.measure_begin();
                e->model->plan_buffers(bq, tq);
                need = std::max(need, e->ctx.arena.measure_end());
            }
        e->ctx.arena.reserve(need + (1 << 20));
        gc_register_overread_range(e->ctx.arena.base(), e->ctx.arena.capacity());
        e->model->plan_buffers(e->ctx.max_batch, T);
        for (int i = 0; i < e->ntwins; ++i) {          // the other instances: the same tensors, a workspace for their part of the batch (offline decodes only)
            TrackedSD tsd2(e->sd);
            e->twin[i]->finalize(tsd2);
            size_t need2 = 0;
            for (int bq : {e->ctx2[i]->max_batch, std::min(e->ctx2[i]->max_batch, 15)}) {
                e->ctx2[i]->arena.measure_begin();
                e->twin[i]->plan_buffers(bq, Tr);
                need2 = std::max(need2, e->ctx2[i]->arena.measure_end());
            }
            e->ctx2[i]->arena.reserve(need2 + (1 << 20));
            gc_register_overread_range(e->ctx2[i]->arena.base(), e->ctx2[i]->arena.capacity());
            e->twin[i]->plan_buffers(e->ctx2[i]->max_batch, T);
        }
        e->sd.clear();
        SE_HIP(hipDeviceSynchronize());
        e->finalized = true;
    });
}

// an offline decode re-carves the arena: it is ordered behind the last frame-online call made on ANOTHER hipStream (the event
// stream_mark() recorded there), so that it cannot overwrite windows that call is still reading
static void stream_order_wait(se_engine* e, hipStream_t st) {
    se_engine::Stream& S = e->strm;
    if (S.has_last && S.last_st != st && S.ev_order) SE_HIP(hipStreamWaitEvent(st, S.ev_order, 0));
}
// ... and the other direction: EVERY call that enqueues work on the arena - offline decodes as well - leaves the
// ordering event behind its last launch, also when it throws after enqueueing, so that a se_stream_begin issued on another
// hipStream right after an offline decode cannot re-carve or zero-fill the arena under it
static void stream_mark(se_engine::Stream& S, hipStream_t st);
struct StreamMarkScope {
    se_engine* e;
    hipStream_t st;
    StreamMarkScope(se_engine* e_, hipStream_t st_) : e(e_), st(st_) {}
    ~StreamMarkScope() {
        try {
            stream_mark(e->strm, st);
        } catch (const std::exception&) {      // (a failed record leaves the previous event in place)
        }
    }
};

int se_forward(se_engine* e, const float* in_dev, const int64_t* in_shape, int32_t in_ndim, float* out_dev,
               void* stream) {
    if (!e) return 1;
    return guard(e, [&] {
        SE_CHECK(e->finalized, "engine not finalized");
        e->strm.carve_B = -1;          // (any decode re-carves the arena: a stream running on this handle zero-fills its next windows)
        stream_order_wait(e, static_cast<hipStream_t>(stream));
        StreamMarkScope sms(e, static_cast<hipStream_t>(stream));
        SE_CHECK(in_dev && out_dev && in_shape, "null argument");
        e->ctx.prof_reset();
        e->model->forward(in_dev, in_shape, in_ndim, out_dev, static_cast<hipStream_t>(stream));
    });
}

int se_uformer_forward(se_engine* e, const float* inputs_dev, const float* src_dev, int32_t batch, int32_t n_samples,
                       float* output_dev, float* src_out_dev, float* output_cplx_dev, float* src_cplx_dev, void* stream) {
    if (!e) return 1;
    return guard(e, [&] {
        SE_CHECK(e->finalized, "engine not finalized");
        e->strm.carve_B = -1;          // (any decode re-carves the arena: a stream running on this handle zero-fills its next windows)
        stream_order_wait(e, static_cast<hipStream_t>(stream));
        StreamMarkScope sms(e, static_cast<hipStream_t>(stream));
        SE_CHECK(inputs_dev && output_dev, "null argument");
        SE_CHECK(batch >= 1 && batch <= e->ctx.max_batch, "batch exceeds max_batch given at create");
        SE_CHECK(n_samples >= e->ctx.geom.n_fft && n_samples <= e->ctx.max_samples, "n_samples outside [n_fft, max_samples]");
        e->ctx.prof_reset();
        e->model->forward_uformer(inputs_dev, src_dev, batch, n_samples, output_dev, src_out_dev, output_cplx_dev, src_cplx_dev,
                                  static_cast<hipStream_t>(stream));
    });
}

int se_enhance_batch(se_engine* e, const float* wav_in_dev, int64_t in_pitch, int32_t batch, int32_t n_samples,
                     float* wav_out_dev, int64_t out_pitch, void* stream) {
    if (!e) return 1;
    return guard(e, [&] {
        SE_CHECK(e->finalized, "engine not finalized");
        e->strm.carve_B = -1;          // (any decode re-carves the arena: a stream running on this handle zero-fills its next windows)
        stream_order_wait(e, static_cast<hipStream_t>(stream));
        StreamMarkScope sms(e, static_cast<hipStream_t>(stream));
        SE_CHECK(wav_in_dev && wav_out_dev, "null argument");
        SE_CHECK(batch >= 1 && batch <= e->ctx.max_batch, "batch exceeds max_batch given at create");
        SE_CHECK(n_samples >= e->ctx.geom.n_fft && n_samples <= e->ctx.max_samples,
                 "n_samples outside [n_fft, max_samples]");
        SE_CHECK(in_pitch >= n_samples && out_pitch >= e->model->output_samples(n_samples), "row pitch too small");
        e->ctx.prof_reset();
        StageProfScope sps(e);
        hipStream_t st = static_cast<hipStream_t>(stream);
        static const int graphs_env = getenv("SE_GRAPH") ? atoi(getenv("SE_GRAPH")) : -1;
        const bool want_graph = (graphs_env >= 0 ? graphs_env != 0 : (e->cfg.flags & SE_CFG_GRAPHS) != 0) && !e->ctx.prof.on &&
                                e->model->graph_capturable();
        if (!want_graph) {
            if (e->ntwins > 0 && !e->ctx.prof.on && batch >= 64) {
                // three parts from 192 clips on where the model has them (at 64 clips a third part loses: DPCRN - 3 %), else two
                const int parts = (e->ntwins >= 2 && batch >= 192) ? 3 : 2;
                const int Bp = (batch + parts - 1) / parts;           // rows per part (the last part may be shorter)
                struct Active {
                    Active() { batch_split_active() = true; }
                    ~Active() { batch_split_active() = false; }
                } active;
                SE_HIP(hipEventRecord(e->ev_tfork, st));
                int njoin = 0;
                for (int i = 0; i < parts - 1; ++i) {
                    const int r0 = (i + 1) * Bp, nr = std::min(Bp, batch - r0);
                    if (nr <= 0) break;
                    hipStream_t s2 = twin_stream(e->cfg.device, i);
                    SE_HIP(hipStreamWaitEvent(s2, e->ev_tfork, 0));
                    e->twin[i]->enhance(wav_in_dev + (size_t)r0 * in_pitch, in_pitch, nr, n_samples, wav_out_dev + (size_t)r0 * out_pitch,
                                        out_pitch, s2);
                    SE_HIP(hipEventRecord(e->ev_tjoin[i], s2));
                    njoin = i + 1;
                }
                e->model->enhance(wav_in_dev, in_pitch, std::min(Bp, batch), n_samples, wav_out_dev, out_pitch, st);
                for (int i = 0; i < njoin; ++i) SE_HIP(hipStreamWaitEvent(st, e->ev_tjoin[i], 0));
                return;
            }
            e->model->enhance(wav_in_dev, in_pitch, batch, n_samples, wav_out_dev, out_pitch, st);
            return;
        }
        const int64_t n_out = e->model->output_samples(n_samples);
        se_engine::GraphEntry* ge = nullptr;
        for (auto& g : e->graphs)
            if (g.batch == batch && g.samples == n_samples) ge = &g;
        if (!ge) {
            const std::pair<int, int> key(batch, n_samples);
            if (!e->cap_stream) {
                SE_HIP(hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking));
                SE_HIP(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
                SE_HIP(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
            }
            if (std::find(e->warmed.begin(), e->warmed.end(), key) == e->warmed.end()) {
                // first call of a shape: eager, but on the capture stream (fork / join around the caller's stream) so that
                // the lazily grown scratch buffers - keyed by stream - already exist when the shape is captured
                e->warmed.push_back(key);
                SE_HIP(hipEventRecord(e->ev_fork, st));
                SE_HIP(hipStreamWaitEvent(e->cap_stream, e->ev_fork, 0));
                e->model->enhance(wav_in_dev, in_pitch, batch, n_samples, wav_out_dev, out_pitch, e->cap_stream);
                SE_HIP(hipEventRecord(e->ev_join, e->cap_stream));
                SE_HIP(hipStreamWaitEvent(st, e->ev_join, 0));
                return;
            }
            if (!e->stage_in) {
                const int64_t max_out = e->model->output_samples(e->ctx.max_samples);
                SE_HIP(hipMalloc(&e->stage_in, (size_t)e->ctx.max_batch * e->ctx.max_samples * sizeof(float)));
                SE_HIP(hipMalloc(&e->stage_out, (size_t)e->ctx.max_batch * max_out * sizeof(float)));
            }
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            // capture on the engine-owned stream: the caller's stream may be the legacy default stream, which cannot capture
            hipStream_t cs = e->cap_stream;
            SE_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
            bool ok = true;
            try {
                e->model->enhance(e->stage_in, n_samples, batch, n_samples, e->stage_out, n_out, cs);
            } catch (const std::exception&) {
                ok = false;
            }
            if (hipStreamEndCapture(cs, &graph) != hipSuccess || !graph) ok = false;
            if (ok && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) ok = false;
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();
            if (getenv("SE_GRAPH_DEBUG"))
                fprintf(stderr, "se_graph: %s (batch %d, samples %d)\n", ok ? "captured" : "capture failed", batch, n_samples);
            e->graphs.push_back({batch, n_samples, ok ? exec : nullptr});
            ge = &e->graphs.back();
        }
        if (!ge->exec) {
            e->model->enhance(wav_in_dev, in_pitch, batch, n_samples, wav_out_dev, out_pitch, st);
            return;
        }
        SE_HIP(hipMemcpy2DAsync(e->stage_in, (size_t)n_samples * sizeof(float), wav_in_dev, (size_t)in_pitch * sizeof(float),
                                (size_t)n_samples * sizeof(float), batch, hipMemcpyDeviceToDevice, st));
        SE_HIP(hipGraphLaunch(ge->exec, st));
        SE_HIP(hipMemcpy2DAsync(wav_out_dev, (size_t)out_pitch * sizeof(float), e->stage_out, (size_t)n_out * sizeof(float),
                                (size_t)n_out * sizeof(float), batch, hipMemcpyDeviceToDevice, st));
    });
}

int se_enhance_ragged(se_engine* e, const float* wav_in_dev, int64_t in_pitch, int32_t batch, const int32_t* lengths,
                      float* wav_out_dev, int64_t out_pitch, void* stream) {
    if (!e) return 1;
    return guard(e, [&] {
        SE_CHECK(e->finalized, "engine not finalized");
        e->strm.carve_B = -1;          // (any decode re-carves the arena: a stream running on this handle zero-fills its next windows)
        stream_order_wait(e, static_cast<hipStream_t>(stream));
        StreamMarkScope sms(e, static_cast<hipStream_t>(stream));
        SE_CHECK(wav_in_dev && wav_out_dev && lengths, "null argument");
        SE_CHECK(batch >= 1 && batch <= e->ctx.max_batch, "batch exceeds max_batch given at create");
        SE_CHECK(e->model->ragged_supported(),
                 "this model looks ahead in time (non-causal convolutions / attention): batch only clips of equal length");
        int Lmax = 0;
        for (int b = 0; b < batch; ++b) {
            SE_CHECK(lengths[b] >= e->ctx.geom.n_fft && lengths[b] <= e->ctx.max_samples,
                     "lengths[" + std::to_string(b) + "] outside [n_fft, max_samples]");
            Lmax = std::max(Lmax, (int)lengths[b]);
        }
        SE_CHECK(in_pitch >= Lmax && out_pitch >= e->model->output_samples(Lmax), "row pitch too small");
        hipStream_t st = static_cast<hipStream_t>(stream);
        const int MB = e->ctx.max_batch;
        if (!e->rag_host) {
            SE_HIP(hipHostMalloc(reinterpret_cast<void**>(&e->rag_host), sizeof(int) * 4 * MB * se_engine::RAG_SLOTS, hipHostMallocDefault));
            SE_HIP(hipMalloc(reinterpret_cast<void**>(&e->rag_dev), sizeof(int) * 4 * MB * se_engine::RAG_SLOTS));
            for (auto& ev : e->rag_ev) SE_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
        const int slot = e->rag_next;
        e->rag_next = (slot + 1) % se_engine::RAG_SLOTS;
        SE_HIP(hipEventSynchronize(e->rag_ev[slot]));      // the copy that last used this slot has run (no-op when unused)
        int* h = e->rag_host + (size_t)slot * 4 * MB;
        int* d = e->rag_dev + (size_t)slot * 4 * MB;
        const int hop = e->ctx.geom.hop;
        for (int b = 0; b < batch; ++b) {
            const int L = lengths[b], Lp = e->model->padded_samples(L);
            h[b] = L;
            h[MB + b] = Lp;
            h[2 * MB + b] = 1 + Lp / hop;
            h[3 * MB + b] = (int)e->model->output_samples(L);
        }
        SE_HIP(hipMemcpyAsync(d, h, sizeof(int) * 4 * MB, hipMemcpyHostToDevice, st));
        SE_HIP(hipEventRecord(e->rag_ev[slot], st));
        e->ctx.prof_reset();
        StageProfScope sps(e);
        Ragged rg{d, d + MB, d + 2 * MB, d + 3 * MB};
        struct Scope {          // the per-row sizes are visible to the launchers only while this call enqueues work
            explicit Scope(const Ragged* r) { set_ragged_ctx(r); }
            ~Scope() { set_ragged_ctx(nullptr); }
        } scope(&rg);
        e->model->enhance(wav_in_dev, in_pitch, batch, Lmax, wav_out_dev, out_pitch, st);
    });
}

// frames [e->strm.t_done, t_end) of the stream through the network, then every output sample they complete (all of them up to
// the end when `last`); the samples are appended to out row b at out_dev[b * out_pitch + *written ...]
static void stream_process(se_engine* e, int t_end, bool last, float* out_dev, int64_t out_pitch, int* written, hipStream_t st) {
    se_engine::Stream& S = e->strm;
    const StftGeom& g = e->ctx.geom;
    const int HC = e->model->stream_hc(), LAG = e->model->stream_lag(), B = S.batch;
    // a frame that is transformed before the end of the stream never touches the end reflection / the tail padding
    // (se_stream_push only releases frames whose last sample has arrived); at the end the decode script's padded length
    const int Lpad = last ? e->model->padded_samples(S.n_total) : S.n_total;
    const int n_final = last ? (int)e->model->output_samples(S.n_total) : S.n_total;
    while (S.t_done < t_end) {
        const int t0 = S.t_done, n = std::min(S.max_chunk, t_end - t0), Tw = HC + n;
        float *spec = nullptr, *mag = nullptr, *est = nullptr;
        e->model->stream_bufs(B, n, &spec, &mag, &est);
        // A chunk size that differs from the last one re-carves the arena: the history columns of the new windows would hold
        // whatever the old carve left there.  Layers only look at the columns their taps reach (restored from the state),
        // but zero-weight padding of the matrix operands may touch a neighbouring column - finite values are harmless there,
        // stale bit patterns need not be finite.
        if (S.carve_B != B || S.carve_n != n) {
            SE_HIP(hipMemsetAsync(const_cast<void*>(e->ctx.arena.base()), 0, e->ctx.arena.used(), st));
            S.carve_B = B;
            S.carve_n = n;
        }
        launch_stft(g, S.wav, e->ctx.max_samples, B, S.n_total, Lpad, S.c, e->ctx.p_in, spec, mag, t0 + n, Tw, st, t0, HC);
        e->model->stream_chunk(B, t0, n, st, last && t0 + n == t_end);
        S.t_done = t0 + n;
        const bool end = last && S.t_done == t_end;
        // estimate frames below t_fin are final (a model that looks ahead finalises LAG frames late); samples whose every
        // covering frame is final: positions below t_fin * hop - and everything once the stream has ended
        const int t_fin = end ? S.t_done : S.t_done - LAG;
        const int o_hi = end ? n_final : std::min(S.n_total, t_fin * g.hop - g.n_fft / 2);
        if (o_hi > S.o_done) {
            launch_istft(g, est, B, t_fin, Tw, nullptr, S.running ? nullptr : S.c, out_dev + *written, out_pitch, o_hi, st, t0 - HC,
                         std::max(0, t0 - HC), S.o_done, S.running ? S.frame_inv : nullptr, S.ring);
            *written += o_hi - S.o_done;
            S.o_done = o_hi;
        }
    }
}

// the last thing a stream call does on its hipStream: record the ordering event the NEXT se_stream_begin (possibly on another
// hipStream) waits for, while the handle is certainly alive
static void stream_mark(se_engine::Stream& S, hipStream_t st) {
    if (!S.ev_order) SE_HIP(hipEventCreateWithFlags(&S.ev_order, hipEventDisableTiming));
    SE_HIP(hipEventRecord(S.ev_order, st));
    S.has_last = true;
    S.last_st = st;
}

static int stream_begin_impl(se_engine* e, int32_t batch, int32_t max_chunk_frames, const float* c_dev, bool running, void* stream) {
    if (!e) return 1;
    return guard(e, [&] {
        SE_CHECK(e->finalized, "engine not finalized");
        e->strm.carve_B = -1;          // (any decode re-carves the arena: a stream running on this handle zero-fills its next windows)
        SE_CHECK(e->model->stream_supported(), "this model has no frame-online mode (CRN, LSTM, GCRN, DPCRN, DCCRN, FullSubNet with the cumulative norm and the cLN `_new` weights of CTSNet / TaylorSENet / G2Net have)");
        SE_CHECK(batch >= 1 && batch <= e->ctx.max_batch, "batch exceeds max_batch given at create");
        const StftGeom& g = e->ctx.geom;
        SE_CHECK((g.n_fft + g.hop - 1) / g.hop - 1 + e->model->stream_lag() <= e->model->stream_hc(),
                 "front end overlap + look-ahead exceed the history the model keeps");
        hipStream_t st = static_cast<hipStream_t>(stream);
        se_engine::Stream& S = e->strm;
        // work of the previous stream on another hipStream: wait for the event that stream_mark() recorded behind ITS last
        // call (the handle itself may have been destroyed since - it is never touched again)
        if (S.has_last && S.last_st != st && S.ev_order) SE_HIP(hipStreamWaitEvent(st, S.ev_order, 0));
        S.max_chunk = std::max(1, std::min(max_chunk_frames > 0 ? max_chunk_frames : 16, e->plan_frames - e->model->stream_hc()));
        if (!S.wav) {
            SE_HIP(hipMalloc(&S.wav, (size_t)e->ctx.max_batch * e->ctx.max_samples * sizeof(float)));
            SE_HIP(hipMalloc(&S.c, (size_t)e->ctx.max_batch * sizeof(float)));
        }
        if (c_dev) SE_HIP(hipMemcpyAsync(S.c, c_dev, (size_t)batch * sizeof(float), hipMemcpyDeviceToDevice, st));
        else launch_fill(S.c, batch, 1.f, st);
        S.running = running;
        if (running) {
            if (!S.sumsq) {
                S.ring = 64;
                while (S.ring < e->plan_frames + 64) S.ring <<= 1;
                SE_HIP(hipMalloc(&S.sumsq, (size_t)e->ctx.max_batch * sizeof(double)));
                SE_HIP(hipMalloc(&S.frame_inv, (size_t)e->ctx.max_batch * S.ring * sizeof(float)));
            }
            SE_HIP(hipMemsetAsync(S.sumsq, 0, (size_t)batch * sizeof(double), st));
            launch_fill(S.frame_inv, (long)batch * S.ring, 1.f, st);
        }
        S.batch = batch;
        S.n_total = S.t_done = S.o_done = 0;
        S.carve_B = S.carve_n = -1;
        e->model->stream_begin(batch, S.max_chunk, st);
        S.active = true;
        stream_mark(S, st);
    });
}

int se_stream_begin(se_engine* e, int32_t batch, int32_t max_chunk_frames, const float* c_dev, void* stream) {
    return stream_begin_impl(e, batch, max_chunk_frames, c_dev, false, stream);
}
int se_stream_begin_running(se_engine* e, int32_t batch, int32_t max_chunk_frames, void* stream) {
    return stream_begin_impl(e, batch, max_chunk_frames, nullptr, true, stream);
}

int se_stream_push(se_engine* e, const float* wav_dev, int64_t pitch, int32_t n_new, float* out_dev, int64_t out_pitch,
                   int32_t* n_out, void* stream) {
    if (!e) return 1;
    return guard(e, [&] {
        se_engine::Stream& S = e->strm;
        SE_CHECK(S.active, "se_stream_push without se_stream_begin");
        SE_CHECK(wav_dev && out_dev && n_out && n_new >= 0, "bad argument");
        SE_CHECK(S.batch == 1 || pitch >= n_new, "se_stream_push: input row pitch smaller than n_new");
        SE_CHECK(S.n_total + n_new <= e->ctx.max_samples, "stream longer than max_samples given at create");
        hipStream_t st = static_cast<hipStream_t>(stream);
        StreamMarkScope sms(e, st);
        const StftGeom& g = e->ctx.geom;
        if (n_new > 0)
            SE_HIP(hipMemcpy2DAsync(S.wav + S.n_total, (size_t)e->ctx.max_samples * sizeof(float), wav_dev,
                                    (size_t)pitch * sizeof(float), (size_t)n_new * sizeof(float), S.batch,
                                    hipMemcpyDeviceToDevice, st));
        S.n_total += n_new;
        // frame t is final once sample t * hop + n_fft / 2 has arrived (its right half is real signal, and frame 0's reflected
        // left half needs sample n_fft / 2 as well)
        const int t_avail = S.n_total > g.n_fft / 2 ? (S.n_total - g.n_fft / 2 - 1) / g.hop + 1 : 0;
        if (S.running)      // c of the frames this push releases: sqrt(samples so far / their sum of squares)
            launch_stream_rms(S.wav, e->ctx.max_samples, S.batch, S.n_total, n_new, S.sumsq, S.c, S.frame_inv, S.ring, S.t_done,
                              std::max(t_avail, S.t_done), st);
        int written = 0;
        const int will = std::max(0, std::min(S.n_total, (t_avail - e->model->stream_lag()) * g.hop - g.n_fft / 2) - S.o_done);
        SE_CHECK(out_pitch >= will, "output row pitch too small for the samples this push completes");
        e->ctx.prof_reset();
        stream_process(e, std::max(t_avail, S.t_done), false, out_dev, out_pitch, &written, st);
        *n_out = written;
    });
}

int se_stream_flush(se_engine* e, float* out_dev, int64_t out_pitch, int32_t* n_out, void* stream) {
    if (!e) return 1;
    return guard(e, [&] {
        se_engine::Stream& S = e->strm;
        SE_CHECK(S.active, "se_stream_flush without se_stream_begin");
        SE_CHECK(out_dev && n_out, "bad argument");
        SE_CHECK(S.n_total >= e->ctx.geom.n_fft, "stream shorter than one FFT frame");
        SE_CHECK(out_pitch >= e->model->output_samples(S.n_total) - S.o_done, "output row pitch too small for the rest of the stream");
        int written = 0;
        StreamMarkScope sms(e, static_cast<hipStream_t>(stream));
        e->ctx.prof_reset();
        if (S.running)
            launch_stream_rms(S.wav, e->ctx.max_samples, S.batch, S.n_total, 0, S.sumsq, S.c, S.frame_inv, S.ring, S.t_done,
                              e->model->num_frames(S.n_total), static_cast<hipStream_t>(stream));
        stream_process(e, e->model->num_frames(S.n_total), true, out_dev, out_pitch, &written, static_cast<hipStream_t>(stream));
        *n_out = written;
        S.active = false;
    });
}

int64_t se_output_samples(const se_engine* e, int32_t n_samples) { return e ? e->model->output_samples(n_samples) : -1; }
int32_t se_num_frames(const se_engine* e, int32_t n_samples) { return e ? e->model->num_frames(n_samples) : -1; }
int32_t se_num_bins(const se_engine* e) { return e ? e->ctx.geom.F() : -1; }

int se_rms_scale(se_engine* e, const float* wav_dev, int64_t pitch, int32_t batch, int32_t n_samples, float* c_dev,
                 void* stream) {
    if (!e) return 1;
    return guard(e, [&] { launch_rms_scale(wav_dev, batch, n_samples, pitch, c_dev, static_cast<hipStream_t>(stream)); });
}

int se_stft(se_engine* e, const float* wav_dev, int64_t pitch, int32_t batch, int32_t n_samples, const float* c_dev,
            float p_in, float* spec_dev, void* stream) {
    if (!e) return 1;
    return guard(e, [&] {
        const int Lpad = e->model->padded_samples(n_samples);
        const int T = 1 + Lpad / e->ctx.geom.hop;
        launch_stft(e->ctx.geom, wav_dev, pitch, batch, n_samples, Lpad, c_dev, p_in, spec_dev, nullptr, T, T,
                    static_cast<hipStream_t>(stream));
    });
}

int se_istft(se_engine* e, const float* spec_dev, int32_t batch, int32_t n_frames, const float* c_dev, float* wav_dev,
             int64_t pitch, int32_t n_out, void* stream) {
    if (!e) return 1;
    return guard(e, [&] {
        launch_istft(e->ctx.geom, spec_dev, batch, n_frames, n_frames, nullptr, c_dev, wav_dev, pitch, n_out,
                     static_cast<hipStream_t>(stream));
    });
}

int se_frontend(se_engine* e, const float* wav_dev, int64_t pitch, int32_t batch, int32_t n_samples, float* c_dev,
                float* spec_dev, void* stream) {
    if (!e) return 1;
    return guard(e, [&] {
        SE_CHECK(wav_dev && c_dev && spec_dev, "null argument");
        SE_CHECK(batch >= 1 && n_samples >= e->ctx.geom.n_fft && pitch >= n_samples, "se_frontend: bad shape");
        hipStream_t st = static_cast<hipStream_t>(stream);
        launch_rms_scale(wav_dev, batch, n_samples, pitch, c_dev, st);
        const int Lpad = e->model->padded_samples(n_samples);
        const int T = 1 + Lpad / e->ctx.geom.hop;
        launch_stft(e->ctx.geom, wav_dev, pitch, batch, n_samples, Lpad, c_dev, e->ctx.p_in, spec_dev, nullptr, T, T, st);
    });
}

int se_backend(se_engine* e, int32_t kind, const float* est_dev, const float* spec_dev, int32_t batch, int32_t n_frames,
               const float* c_dev, float* wav_dev, int64_t pitch, int32_t n_out, void* stream) {
    if (!e) return 1;
    return guard(e, [&] {
        SE_CHECK(est_dev && wav_dev, "null argument");
        SE_CHECK(kind == SE_BACKEND_RI || kind == SE_BACKEND_MAG || kind == SE_BACKEND_CMASK, "se_backend: unknown kind");
        SE_CHECK(kind == SE_BACKEND_RI || spec_dev, "se_backend: this kind needs the front end's spectrum");
        SE_CHECK(batch >= 1 && n_frames >= 1 && n_out >= 1 && pitch >= n_out, "se_backend: bad shape");
        hipStream_t st = static_cast<hipStream_t>(stream);
        const int F = e->ctx.geom.F();
        const size_t need = (size_t)batch * 2 * F * n_frames * sizeof(float);
        if (need > e->hook_cap) {
            SE_HIP(hipStreamSynchronize(st));       // (an earlier call on this stream may still read the old buffer)
            if (e->hook_buf) SE_HIP(hipFree(e->hook_buf));
            e->hook_buf = nullptr;
            e->hook_cap = 0;
            SE_HIP(hipMalloc(&e->hook_buf, need));
            e->hook_cap = need;
        }
        if (kind == SE_BACKEND_RI) launch_polar_pow(est_dev, e->hook_buf, batch, F, n_frames, e->ctx.p_out, st);
        else if (kind == SE_BACKEND_MAG) launch_mag_phase(est_dev, spec_dev, e->hook_buf, batch, F, n_frames, e->ctx.p_out, st);
        else launch_cmask_apply(est_dev, spec_dev, e->hook_buf, batch, F, n_frames, e->ctx.p_out, st);
        launch_istft(e->ctx.geom, e->hook_buf, batch, n_frames, n_frames, nullptr, c_dev, wav_dev, pitch, n_out, st);
    });
}

int se_set_profiling(se_engine* e, int32_t on) {
    if (!e) return 1;
    e->ctx.prof_set(on != 0);
    return 0;
}

int se_get_profile(se_engine* e, double* gemm_ms, int64_t* gemm_launches, double* gemm_flops) {
    if (!e) return 1;
    return guard(e, [&] {
        // launches on the auxiliary streams (FullSubNet's sub-band halves) run concurrently with the main stream's: the sum
        // of their durations is kernel time, not wall time
        double ms = e->ctx.prof.total_ms(), fl = e->ctx.prof.flops;
        int64_t n = e->ctx.prof.launches;
        for (auto& p : e->ctx.aux_prof) {
            ms += p.total_ms();
            fl += p.flops;
            n += p.launches;
        }
        if (gemm_ms) *gemm_ms = ms;
        if (gemm_launches) *gemm_launches = n;
        if (gemm_flops) *gemm_flops = fl;
    });
}

int se_get_stage_profile(se_engine* e, int32_t stage, double* ms, int64_t* launches, double* bytes) {
    if (!e) return 1;
    return guard(e, [&] {
        SE_CHECK(stage >= 0 && stage < STAGE_COUNT, "stage id out of range");
        if (ms) *ms = e->ctx.stage_prof.ms(stage);
        if (launches) *launches = e->ctx.stage_prof.slot[stage].launches;
        if (bytes) *bytes = e->ctx.stage_prof.slot[stage].bytes;
    });
}

}  // extern "C"
