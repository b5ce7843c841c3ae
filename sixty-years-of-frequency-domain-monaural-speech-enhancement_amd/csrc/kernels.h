// Front/back-end and auxiliary kernels (launchers).  All tensors are fp32 device pointers.
// Engine-internal spectrogram layout: [B][C][F][Tp] with T contiguous, Tp = row pitch (>= T).
#pragma once
#include <hip/hip_runtime.h>
#include <utility>
#include <vector>
#include "common.h"

namespace se {

// Ragged batches (se_enhance_ragged): rows of one call have different lengths.  The engine publishes the per-row sizes
// for the duration of Model::enhance(); every launcher whose arithmetic depends on the utterance length reads them:
// unit-RMS scale, STFT (reflect pad at the row's own end, frames >= tlen[b] written as zeros), iSTFT / overlap-add,
// and every statistic taken over the whole utterance (InstanceNorm, the TCM head, FullSubNet's utterance means).
// Everything else on the path is causal in time, so frames >= tlen[b] never reach a valid frame.
struct Ragged {
    const int* len = nullptr;    // device [B]: samples of row b
    const int* lpad = nullptr;   // device [B]: samples the STFT sees (decode scripts that tail-pad to a hop multiple)
    const int* tlen = nullptr;   // device [B]: frames of row b
    const int* olen = nullptr;   // device [B]: output samples of row b
};
const Ragged* ragged_ctx();              // nullptr: equal-length batch
// d [4][MB] <- (len, lpad, tlen, olen) for rows 0..B-1 (an equal-length batch as ragged rows: model.h PadFrames)
void launch_fill_rows(int* d, int MB, int B, int len, int lpad, int tlen, int olen, hipStream_t s);
void set_ragged_ctx(const Ragged* r);    // thread-local (a handle is driven by one thread at a time)

// HIP-event timing of the HBM-bound front / back-end kernels (se_get_stage_profile): every launcher below brackets its
// launches with events on the launch stream and books the ALGORITHMIC bytes of the stage (what it must read + write once,
// SURVEY 8(d)), so that bench.py can put bytes / time next to the 8 TB/s HBM peak.  Published thread-locally by the
// engine while profiling is on; null otherwise (no events, no overhead).
enum Stage : int { STAGE_RMS = 0, STAGE_STFT = 1, STAGE_MASK = 2, STAGE_ISTFT = 3, STAGE_COUNT = 4 };
struct StageProf {
    struct Slot {
        std::vector<hipEvent_t> ev;
        size_t used = 0;
        double bytes = 0.0;
        long launches = 0;
    } slot[STAGE_COUNT];
    void begin(int stage, hipStream_t st);
    void end(int stage, hipStream_t st, double bytes);
    void reset();
    double ms(int stage);
    ~StageProf();
};
StageProf* stage_prof();
void set_stage_prof(StageProf* p);
struct StageScope {       // RAII bracket used by the launchers
    int stage; hipStream_t st; double bytes; StageProf* p;
    StageScope(int stage_, hipStream_t st_, double bytes_) : stage(stage_), st(st_), bytes(bytes_), p(stage_prof()) {
        if (p) p->begin(stage, st);
    }
    ~StageScope() {
        if (p) p->end(stage, st, bytes);
    }
};

struct StftGeom {
    int n_fft, hop, win;   // win <= n_fft (window centred in n_fft, torch.stft convention)
    int F() const { return n_fft / 2 + 1; }
};

// c[b] = sqrt(L / sum_t x[b][t]^2)  (reference: `c = np.sqrt(len(x) / np.sum(x ** 2.0))`, every *_decode_vb.py);
// recip=1 stores 1/c instead (G2Net_VB/com_decode.py:43-44 normalises with x / c, c = RMS).
void launch_rms_scale(const float* wav, int B, int L, long pitch, float* c_out, hipStream_t s);

// Framed, centred (reflect-padded) STFT with periodic Hann window, fused with the unit-RMS scaling and the
// magnitude power-compression |X|^p * e^{j angle X}.
//   wav [B][pitch] (first L samples valid; samples in [L, Lpad) are the decode scripts' zero tail pad)
//   spec_ri [B][2][F][Tp]   (may be null)     mag [B][F][Tp] = |X|^p (may be null)
// Streaming: only frames [t_first, T) are transformed, frame t lands in column t - t_first + col0 of the Tp-pitch rows.
void launch_stft(const StftGeom& g, const float* wav, long pitch, int B, int L, int Lpad, const float* c_scale,
                 float p_in, float* spec_ri, float* mag, int T, int Tp, hipStream_t s, int t_first = 0, int col0 = 0);

// Inverse: spec_ri [B][2][F][Tp] -> windowed frames [B][T][n_fft] (scratch) -> overlap-add, divide by the
// overlap-added squared window, drop n_fft/2 head, write Lout samples, divide by c.
// Streaming window: frame t sits in spec column t - t_off, frames [t_lo, T) exist; output samples [o_lo, Lout) are written
// to wav_out[o - o_lo] (every frame covering them must be in the window).
void launch_istft(const StftGeom& g, const float* spec_ri, int B, int T, int Tp, float* frames_scratch,
                  const float* c_scale, float* wav_out, long out_pitch, int Lout, hipStream_t s, int t_off = 0, int t_lo = 0,
                  int o_lo = 0, const float* frame_inv = nullptr, int ring = 0);
// frame_inv (optional, [B][ring], ring a power of two): frame t is multiplied by frame_inv[b][t % ring] before the
// overlap-add - streams that run on a RUNNING unit-RMS scale transform every frame under the c known when it was
// released (se_stream_begin_running); c_scale is then null

// running unit-RMS scale of frame-online streams (k_misc.hip): sumsq[b] += the n_new newest samples squared, c[b] =
// sqrt(n_total / sumsq[b]), frames [t0, t1) get 1 / c[b] in the ring frame_inv [B][ring]
void launch_stream_rms(const float* wav, long pitch, int B, int n_total, int n_new, double* sumsq, float* c, float* frame_inv,
                       int ring, int t0, int t1, hipStream_t s);

// round-3 kernels behind the two launchers above (k_stft2.hip): FFT points in registers, two LDS exchanges, 32-frame tiles
// moved through LDS so that the [F][T]-major spectrogram is touched in 128 B runs; SE_STFT_V1=1 selects the old kernels
bool stft2_enabled();
void launch_stft2(const StftGeom& g, const float* wav, long pitch, int B, int L, int Lpad, const float* c_scale, float p_in,
                  float* spec_ri, float* mag, int T, int Tp, hipStream_t s, int t_first, int col0);
void launch_istft2(const StftGeom& g, const float* spec_ri, int B, int T, int Tp, const float* c_scale, float* wav_out,
                   long out_pitch, int Lout, hipStream_t s, int t_off, int t_lo, int o_lo, const float* frame_inv = nullptr,
                   int ring = 0);

// Frame-online context of the models that are built from shared blocks (blocks.h / unet.h: the cLN `_new` variants).
// While a chunk is decoded the model publishes it thread-locally; every activation of the chunk is a window of H history
// columns + n new frames (row pitch H + n), and the shared launch helpers then
//   * produce only the new frames (gc_launch's first output frame = H),
//   * bring the history columns their taps / FIRs reach back to into the source tensor before they read it, from a state
//     slot that belongs to the call site (slots are taken in call order, which is the same for every chunk), and leave the
//     last columns of the window there for the next chunk,
//   * carry the cumulative LayerNorm sums across chunks.
// Zero-initialised slots are the zero padding of the causal convs; frames before the start of the stream never count.
struct StreamCtx {
    int H = 0, n = 0;        // window = H history columns + n new frames
    long t0 = 0;             // index of the first new frame in the stream
    int B = 0;
    std::vector<std::pair<void*, size_t>>* slots = nullptr;     // device state, owned by the model
    size_t cursor = 0;
    const void* memo_src = nullptr;     // source whose history the previous launch restored (parity classes of one deconv)
    int memo_need = 0;
    void* slot(size_t bytes, hipStream_t st);      // next state slot (allocated zeroed on the first chunk)
};
StreamCtx* stream_ctx();                 // nullptr outside frame-online chunks
void set_stream_ctx(StreamCtx* c);
// x [B][C][F][H + n] (strides in floats, frames contiguous): columns [H - need, H) <- slot, then slot <- the last `need`
// columns of the window
void stream_exchange(float* x, long sb, long sc, long sf, int B, int C, int F, int need, hipStream_t st);
// the two sources of a concatenating layer in one launch (same state slots, in the same order, as two calls)
void stream_exchange_pair(float* x0, long sb0, long sc0, long sf0, int C0, int F0, float* x1, long sb1, long sc1, long sf1, int C1,
                          int F1, int B, int need, hipStream_t st);

// Model-side owner of the state slots and RAII publisher of one chunk's context
struct StreamSlots {
    std::vector<std::pair<void*, size_t>> v;
    int B = 0;
    void begin(int B_, hipStream_t st) {       // new stream: zero state (= zero padding before the first frame)
        if (B_ != B) clear();
        for (auto& s : v) SE_HIP(hipMemsetAsync(s.first, 0, s.second, st));
        B = B_;
    }
    void clear() {
        for (auto& s : v) (void)hipFree(s.first);
        v.clear();
    }
    ~StreamSlots() { clear(); }
};
struct StreamScope {
    StreamCtx cx;
    StreamScope(StreamSlots& sl, int H, int n, long t0, int B) {
        cx.H = H; cx.n = n; cx.t0 = t0; cx.B = B; cx.slots = &sl.v;
        set_stream_ctx(&cx);
    }
    ~StreamScope() { set_stream_ctx(nullptr); }
    StreamScope(const StreamScope&) = delete;
    StreamScope& operator=(const StreamScope&) = delete;
};

// Streaming history columns of a [B][rows][Tw] activation: restore the first `hc` columns from state [B][rows][hc] (the
// producing kernel recomputed them without their own history), or save the last `hc` columns into it.
void launch_hist_restore(float* buf, const float* state, int B, long rows, int Tw, int hc, hipStream_t s);
void launch_hist_save(const float* buf, float* state, int B, long rows, int Tw, int hc, hipStream_t s);

// the same for up to 16 tensors of one chunk in one launch (a one-frame push is bound by its launch count)
struct HistBatch {
    static constexpr int MAX = 16;
    float* buf[MAX];
    float* state[MAX];
    long rows[MAX];
    int n = 0;
    void add(float* b, float* s, long r) {
        buf[n] = b;
        state[n] = s;
        rows[n] = r;
        ++n;
    }
};
void launch_hist_batch(const HistBatch& hb, int B, int Tw, int hc, bool save, hipStream_t s);

// DCCRN 'E' mask (DCCRN_cprs.py:201-225) + the decode script's mag/phase/decompress (dccrn_decode_vb.py:45-58):
//   mask [B][2][F-1][Tp] (bins 1..F-1), spec [B][2][F][Tp] -> est [B][2][F][Tp], DC bin = 0.
void launch_dccrn_mask(const float* mask, const float* spec, float* est, int B, int F, int T, int Tp, float p_out,
                       hipStream_t s, int mode = 0);

// Generic tiled transpose of the outer and inner dims:  out[t][k][a] = in[a][k][t]
//   in element (a,k,t) at a*in_sa + k*in_sk + t ; out element (t,k,a) at t*out_st + k*out_sk + a
void launch_transpose_akt(const float* in, float* out, int A, int K, int T, long in_sa, long in_sk, long out_st,
                          long out_sk, hipStream_t s);

// y = x  (strided 4-D copy / layout change):  out[b][c][i][j] at strides so_*, in at si_*; inner j contiguous in out.
void launch_copy4(const float* in, float* out, int B, int C, int I, int J, long si_b, long si_c, long si_i, long si_j,
                  long so_b, long so_c, long so_i, hipStream_t s);

void launch_fill(float* p, long n, float v, hipStream_t s);
// ragged batches only (no-op otherwise): x [B][rows][T], frames t >= tlen[b] of every row set to zero
void launch_zero_tail(float* x, int B, long rows, int T, hipStream_t s);

// nn.InstanceNorm2d / InstanceNorm1d (affine, per-utterance statistics, biased variance, eps 1e-5) over the contiguous
// plane of P values of every (b, c), optionally followed by a per-channel PReLU; in place allowed.
//   x [B][C][P];  gamma/beta [C];  slope [C] or null
//   res (optional, [B][C][P], may alias y): y = PReLU(norm(x)) + res
// T (ragged batches only): frames per line of the plane (P = lines * T), so that the statistics can stop at the row's
// own frame count
void launch_instnorm_prelu(const float* x, float* y, const float* gamma, const float* beta, const float* slope, int B,
                           int C, int P, hipStream_t s, const float* res = nullptr, int T = 0);
// InstanceNorm folded into the consumers: statistics -> per-(b, c) {scale, shift, slope - 1, x0} (GCParams::nrm0 / nrm1), and the
// elementwise pass y = f_a(xa) (+ f_b(xb)) for tensors that still have to exist normalised
void launch_instnorm_finalize(const float* stats, int nslot, const float* gamma, const float* beta, const float* slope, float* nrm,
                              int B, int C, int P, hipStream_t s, int T = 0);      // (T: as launch_instnorm_prelu - ragged batches)
void launch_instnorm_apply2(const float* xa, const float* na, const float* xb, const float* nb, float* y, int B, int C, int P,
                            hipStream_t s);
// statistics from the producing conv (nslot (sum, sum of squares) pairs per (b, c) plane, GCParams::stats)
void launch_instnorm_prelu_stats(const float* x, float* y, const float* gamma, const float* beta, const float* slope,
                                 const float* stats, int nslot, int B, int C, int P, hipStream_t s, const float* res = nullptr,
                                 int T = 0);

// TCM branch head (CTSNet/Step1_network.py:161-176): y = ShareSepConv( InstanceNorm1d( PReLU(x) ) ) per (b, c) row of
// T frames; fir [K] is the single FIR shared by all channels (causal, left pad K-1), K = 0 -> no FIR.
void launch_tcm_head(const float* x, float* y, const float* slope, const float* gamma, const float* beta,
                     const float* fir, int K, int B, int C, int T, hipStream_t s);

// One whole TCM / GLU block per utterance in a single kernel (k_tcm.hip): x [B][256][T] -> y [B][256][T].
struct TcmFusedW {          // device weights in MFMA fragment order, owned by the block
    float *w1 = nullptr, *w2L = nullptr, *w2R = nullptr, *w3 = nullptr;     // w2R == nullptr: single branch (no gate)
    int ks = 0;
};
struct TcmFusedHeads {      // per-channel [64] PReLU slopes / InstanceNorm affine of the three heads, branch FIRs [K]
    const float *sL, *gL, *bL, *firL, *sR, *gR, *bR, *firR, *sO, *gO, *bO;
};
TcmFusedW tcm_fused_build(const std::vector<float>& w_in, const std::vector<float>& w_left, const std::vector<float>* w_right,
                          const std::vector<float>& w_out, int ks);
void tcm_fused_free(TcmFusedW& f);
bool tcm_fused_supported(int T);
void launch_tcm_fused(const TcmFusedW& f, const TcmFusedHeads& hd, const float* x, float* y, int B, int T, int dil, int K,
                      hipStream_t s, bool cum = false);     // cum: CumulativeLayerNorm heads (the `_new` variants)

// The same block in the frame-online mode of the cumulative-LayerNorm variants: one launch per block and chunk, one workgroup
// per stream, state (cLN sums, FIR / dilated-conv history) in one slot of the stream context (k_tcm_stream.hip).
struct TcmStreamW {         // device weights, transposed (output row contiguous), owned by the block
    float *w_in = nullptr, *w_l = nullptr, *w_r = nullptr, *w_out = nullptr;     // w_r == nullptr: single branch (no gate)
    int ks = 0;
};
TcmStreamW tcm_stream_build(const std::vector<float>& w_in, const std::vector<float>& w_left, const std::vector<float>* w_right,
                            const std::vector<float>& w_out, int ks);
void tcm_stream_free(TcmStreamW& f);
bool tcm_stream_enabled();      // SE_TCM_STREAM=0: the multi-launch path of round 2
void launch_tcm_stream(const TcmStreamW& f, const TcmFusedHeads& hd, const float* x, float* y, int dil, int K, hipStream_t s);
// up to 8 blocks that feed each other as one launch (SE_TCM_CHAIN=0: one launch per block)
bool tcm_chain_enabled();
void launch_tcm_chain(const TcmStreamW* const* f, const TcmFusedHeads* hd, const int* dil, const int* K, int nblk, const float* x,
                      float* y, hipStream_t s);

// CumulativeLayerNorm2d / 1d of the `_new` variants (CTSNet_new/Step1_network.py:213-286): frame t is normalised by the
// statistics of all C*F values of frames 0..t;  x [B][C][F][T] (F = 1 for 1-D), gain / bias [C].
//   y = FIR_K( cLN( PReLU_pre(x) ) )   (TCM branch head, K > 0, not in place)   or   y = PReLU_post( cLN(x) )
// res (optional, offline, plain 2-D form only; may alias y): y = PReLU_post( cLN(x) ) + res
// frame-online chunk: true when launch_cln(..., res) adds the residual in its own launch (one- / two-frame register form)
bool cln_stream_takes_res(int C, int F);
void launch_cln(const float* x, float* y, const float* gain, const float* bias, const float* pre_slope,
                const float* post_slope, const float* fir, int K, int B, int C, int F, int T, hipStream_t s,
                const float* res = nullptr);

// the offline 2-D form behind a conv whose epilogue emitted the per-frame sums (parts [B][F][T][2], GCParams::cstats)
void launch_cln_parts(const float* x, float* y, const float* gain, const float* bias, const float* post_slope, const float* parts,
                      int B, int C, int F, int T, hipStream_t s, const float* res = nullptr);

// y = a + b (n elements);  y may alias a
void launch_add(const float* a, const float* b, float* y, long n, hipStream_t s);

// y = elu(x)  (GCRN applies ELU to the skip tensors again inside every concat, GCRN_noncprs.py:149-158)
void launch_elu(const float* x, float* y, long n, hipStream_t s);

// Decode-script decompress of a complex-mapping output (GCRN/gcrn_decode_vb.py:47-55): out = |e|^p * e / |e|.
void launch_polar_pow(const float* x, float* out, int B, int F, int T, float p_out, hipStream_t s);

// nn.LayerNorm([F, C]) over the (C, F) plane of every (b, t) column of a [B][C][F][T] tensor, affine weight/bias
// indexed [f][c] (DPCRN/DPCRN.py:56-57 ln1/ln2), fused with the residual add that follows it (:74, :88):
//   out = LN(x) * w + b + res
// post: 0 none, 1 swish y*sigmoid(y) (Uformer/dsconv2d_cplx.py:56); prelu_slope: device scalar or null, applied after
// the norm (Uformer attention branches, t_att_cplx.py:93); order: norm -> post -> prelu -> + res
void launch_layernorm_cf(const float* x, const float* res, const float* w, const float* b, float* out, int B, int C,
                         int F, int T, float eps, hipStream_t s, int post = 0, const float* prelu_slope = nullptr);

// Complex ratio mask applied to the network input (DPCRN/DPCRN.py:33-42) + decode-script decompress
// (dpcrn_decode_vb.py:48-57): est = (X * M); out = |est|^p_out * est / |est|.   All [B][2][F][T].
void launch_cmask_apply(const float* mask, const float* spec, float* out, int B, int F, int T, float p_out,
                        hipStream_t s);

// Magnitude mapping back end (LSTM/lstm_decode_vb.py:47-49, CRN/crn_decode_vb.py:46-49):
//   out = mag^p_out * exp(j * angle(X_noisy));  mag [B][F][T], spec/out [B][2][F][T].
void launch_mag_phase(const float* mag, const float* spec, float* out, int B, int F, int T, float p_out,
                      hipStream_t s);

// Persistent LSTM recurrence (k_lstm.hip): all T steps of Z x O x ceil(S/16) independent sequence tiles in one launch.
//   gx  : gate pre-activations W_ih x + b_ih + b_hh, rows gate-interleaved (4u+g); element (o, z, t, row, n) at
//         gx + o*gx_o + z*gx_z + t*gx_t + row*gx_row + n          (n = sequence index, contiguous)
//   whh : [Z][4H][H] row-major, rows gate-interleaved
//   out : h_t, element (o, z, t, u, n) at out + o*out_o + z*out_z + t*out_t + u*out_row + n
struct LstmPersistArgs {
    const float* gx; const float* whh; float* out;
    long gx_o, gx_z, gx_t, gx_row;
    long whh_z;
    long out_o, out_z, out_t, out_row;
    int H, T, S, Z, O, reverse;   // reverse: bit z set -> LSTM z walks the steps backwards (BiLSTM)
    // streaming (optional, O = 1): start from (st_h, st_c) instead of zeros and leave the state after the last step there;
    // element (z, u, n) at z * st_z + u * S + n
    float* st_h = nullptr;
    float* st_c = nullptr;
    long st_z = 0;
};
void launch_lstm_persist(const LstmPersistArgs& a, hipStream_t s);


// librosa.resample(y, sr_in, sr_out, fix=True, scale=False) (resampy 'kaiser_best'), k_resample.hip; x [batch][n_in] ->
// y [batch][resample_out_samples(n_in, sr_in, sr_out)]
long resample_out_samples(int n_in, int sr_in, int sr_out);
void launch_resample(const float* x, long in_pitch, int batch, int n_in, int sr_in, int sr_out, float* y, long out_pitch,
                     hipStream_t s);
// PCM_16 <-> float32 rows at the two ends of the decode driver (sf.read / sf.write PCM_16 of every decode script)
void launch_pcm16_decode(const short* in, long in_pitch, int batch, int n, float* out, long out_pitch, hipStream_t s);
void launch_pcm16_encode(const float* in, long in_pitch, int batch, int n, short* out, long out_pitch, hipStream_t s);

// Weight-stationary cooperative LSTM recurrence for H = 512 / 1024 (k_lstm_coop.hip): W_hh spread over the register
// files of all CUs, one launch for all T steps.  Same tensor conventions as LstmPersistArgs (O = 1):
//   gx   : element (z, t, row, n) at gx + z*gx_z + t*gx_t + row*gx_row + n   (rows gate-interleaved, bias included)
//   out  : h_t element (z, t, u, n) at out + z*out_z + t*out_t + u*out_row + n
//   cell : scratch [Z][H][S];  hx / bar / SS are filled in by the launcher
struct LstmCoopArgs {
    const float* gx; const float* whh; float* out; float* cell;
    long gx_z, gx_t, gx_row;
    long whh_z;
    long out_z, out_t, out_row;
    int H, T, S, Z, reverse;
    float* hx; unsigned* bar; int SS, dbg;
    // chunked layer pipeline (launch_lstm_coop_chunk, pz = 1): the Z LSTMs of a launch are LAYERS of one stack, each on its own
    // range of steps - layer lz[z] runs steps t0[z] .. t0[z] + Tz[z] of its recurrence, continuing from the state the launch of
    // its previous range left (h in the launcher's exchange slabs, c in `cell`, both indexed by layer); gxp / outp point at the
    // first step of the range, whhp at the layer's matrix
    int pz;
    const float* gxp[4]; const float* whhp[4]; float* outp[4];
    int lz[4], t0[4], Tz[4];
};
bool lstm_coop_supported(int H, int S, int Z);
// layers of a stack on consecutive chunks of steps in one cooperative launch (k_lstm_coop.hip: lstm_coop16_kernel); false: the
// shape has no such kernel.  n_layers = layers of the whole stack (sizes the exchange slabs / `cell` = [n_layers][H][S])
bool lstm_coop_chunk_supported(int H, int S, int n_layers);
void launch_lstm_coop_chunk(const LstmCoopArgs& a, int n_layers, hipStream_t s);
// A stack of L LSTM layers (equal width) on ONE sequence as one cooperative launch (k_lstm_coop.hip: lstm_stack_kernel):
// layer l runs l frames behind layer l - 1, one exchange latency per frame serves all layers.  gx0: the first layer's gate
// pre-activations (input projection + bias, rows 4u + gate); whh / wih / bias: row-major [4H][H] / [4H][H] / [4H] per layer
// (wih, bias unused for layer 0); out: h of the LAST layer, unit u of frame t at out[t * out_t + u * out_row].
struct LstmStackArgs {
    const float* gx0; long gx_t, gx_row;
    const float* whh[3]; const float* wih[3]; const float* bias[3];
    float* out; long out_t, out_row;
    int H, T, L;
    float* hx;          // internal
};
bool lstm_stack_supported(int H, int L);
void launch_lstm_stack(const LstmStackArgs& a, hipStream_t s);
void launch_lstm_coop(const LstmCoopArgs& a, hipStream_t s);

}  // namespace se
