/* se_engine.h - C ABI of the MI355X (gfx950) speech-enhancement decode engine (libse_engine.so).
 *
 * The reference (cszheng-ioa/Sixty-years-of-frequency-domain-monaural-speech-enhancement) has no FFI: its only
 * seams are Python ones.  Each entry point below states the reference interface it stands in for
 * (paths relative to the reference root).  Plain C types only; no torch types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; the message is read with se_last_error().
 *   - `*_dev` pointers are DEVICE pointers owned by the caller (e.g. torch-ROCm `tensor.data_ptr()`);
 *     the engine never frees them.  Scratch and weights are engine-owned, sized at create / finalize.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Work is enqueued, not synchronised;
 *     only se_engine_finalize() and se_engine_destroy() synchronise.
 *   - one handle per GPU / rank; a handle is not thread-safe.
 *   - all floating-point data is fp32 (the reference feeds `torch.FloatTensor`).
 */
#ifndef SE_ENGINE_H
#define SE_ENGINE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct se_engine se_engine;

/* One id per reference model class on the decode path (constructor call sites cited). */
enum se_model_id {
    SE_MODEL_LSTM = 1,        /* lstm_net()   LSTM/lstm_decode_vb.py:18   (LSTM/LSTM.py:14-28)            */
    SE_MODEL_CRN = 2,         /* crn_net()    CRN/crn_decode_vb.py:18     (CRN/CRN.py:16-117)             */
    SE_MODEL_GCRN = 3,        /* Net()        GCRN/gcrn_decode_vb.py:19   (GCRN/GCRN_noncprs.py:86-165)   */
    SE_MODEL_DPCRN = 4,       /* dpcrn()      DPCRN/dpcrn_decode_vb.py:19 (DPCRN/DPCRN.py:16-174)         */
    SE_MODEL_DCCRN = 5,       /* DCCRN(rnn_units=256, masking_mode='E', use_clstm=True,
                                 kernel_num=[32,64,128,256,256,256])  DCCRN/dccrn_decode_vb.py:11         */
    SE_MODEL_FULLSUBNET = 6,  /* Model(...)   FullSubNet/fullsubnet_sa_decode_vb.py:11-24                 */
    SE_MODEL_CTSNET = 7,      /* Step1_net(), Step2_net(X=6,R=3)  CTSNet/two_stage_com_decode_vb.py:13-14 */
    SE_MODEL_G2NET = 8,       /* gaf_base(...)  G2Net_VB/com_decode.py:23                                 */
    SE_MODEL_TAYLORSENET = 9, /* TaylorSENet(...) TaylorSENet/taylorsenet_decode_vb.py:11-13              */
    SE_MODEL_UFORMER = 10     /* Uformer()    Uformer/uformer_decode_vb.py:19                             */
};

/* Replaces the module constants + hand-edited exponents of each decode script
 * (e.g. DCCRN/config.py:5-8 win_size/fft_num/win_shift; `** 1.0` vs `** 0.5 / ** 2.0` at
 * DCCRN/dccrn_decode_vb.py:40,48 and DCCRN/dccrn_decode.py:44,52).  n_fft/hop/win = 0 selects the
 * model's own front end. */
/* se_config.flags: replay se_enhance_batch as a hipGraph per (batch, n_samples) shape - captured on the second call of
 * a shape, caller buffers staged through engine-owned rows.  Results are bit-identical to the eager path.  Off by default:
 * measured on MI355X the replay does not shorten a batch-1 decode (the path is bound by its chain of dependent small
 * kernels, not by launch submission); models that fork onto auxiliary streams (FullSubNet) always run eagerly. */
#define SE_CFG_GRAPHS 1
/* DCCRN only.  `DCCRN/DCCRN_cprs.py:6` imports its operators from a third-party `complexnn.py` that is absent from the
 * reference and unversioned.  The engine follows the published upstream (huyanxin/DeepComplexCRN) as restated in
 * oracle/_complexnn_recall.py; the two conventions DCCRN_cprs.py itself does not determine can be
 * flipped here - they are weight-preparation switches at se_engine_finalize, so adopting the real file, should it differ,
 * is a flag and a fixture regeneration, not a kernel change.
 *   SE_CFG_DCCRN_BIAS_PER_PART: each part adds its own conv's bias once (real += b_real, imag += b_imag) instead of the
 *                               two-real-conv combination real += b_real - b_imag, imag += b_real + b_imag
 *   SE_CFG_DCCRN_PLAIN_CAT    : complex_cat([out, skip], 1) is a plain channel concat ([out_r, out_i, skip_r, skip_i])
 *                               instead of real halves together, imaginary halves together */
#define SE_CFG_DCCRN_BIAS_PER_PART 2
#define SE_CFG_DCCRN_PLAIN_CAT 4
/* FullSubNet only: `sequence_model="GRU"` of Model(...) - both sequence models are torch.nn.GRU stacks instead of nn.LSTM
 * (FullSubNet/fullsubnet_net_sa/sequence_model.py:36-43; the decode script passes "LSTM", fullsubnet_sa_decode_vb.py:16).
 * The state dict then carries the GRU's [3H, .] weight_ih / weight_hh / bias_ih / bias_hh entries. */
#define SE_CFG_FSN_GRU 8
/* FullSubNet with norm_type = "cumulative_laplace_norm" (FullSubNet/fullsubnet_net_sa/base_model.py:212-240, selected at
 * :296-303) instead of the decode script's "offline_laplace_norm": every input is divided by its running mean over the frames
 * seen so far.  The network is then causal up to its look_ahead = 2 frames, and se_stream_* accepts the engine. */
#define SE_CFG_FSN_CUMULATIVE 16
/* DCCRN(masking_mode=...) (DCCRN/DCCRN_cprs.py:205-223): 'E' (default, the decode script's: tanh-bounded magnitude mask and
 * phase rotation), 'C' (complex ratio mask: est = spec x mask), 'R' (one real mask per part: est_r = spec_r mask_r, est_i =
 * spec_i mask_i).  At most one of the two bits. */
#define SE_CFG_DCCRN_MASK_C 32
#define SE_CFG_DCCRN_MASK_R 64
/* G2Net / TaylorSENet: the constructor's repeat count, bits 8-11 of se_config.flags.
 *   gaf_base(..., stage_num = n)      G2Net_VB/gaf_net_320.py:27,55-58  (n GAF stages `gafs.<s>.`, 1 <= n <= 8)
 *   TaylorSENet(..., order_num = n)   TaylorSENet/TaylorSENet.py:27,66-70 (n high-order blocks `highorderblock_list.<k>.`, 0 <= n <= 8)
 * 0 in the field = the decode scripts' value (3 for both: G2Net_VB/com_decode.py:23, TaylorSENet/taylorsenet_decode_vb.py:11-13). */
#define SE_CFG_REPEATS(n) ((((n) + 1) & 15) << 8)
/* CTSNet: Step2_net(X, R) (CTSNet/Step2_network.py:13-21: R groups `tcm_list.<r>.` of X gated blocks `glu_list.<i>.`, dilation 2^i):
 * R in SE_CFG_REPEATS (1 <= R <= 8), X in bits 12-15 (1 <= X <= 6); 0 in a field = the decode script's 3 / 6
 * (two_stage_com_decode_vb.py:14). */
#define SE_CFG_REPEATS2(n) ((((n) + 1) & 15) << 12)

typedef struct se_config {
    int32_t model;        /* enum se_model_id */
    int32_t device;       /* HIP device ordinal */
    int32_t max_batch;    /* utterances per se_enhance_batch call (scratch is sized for this) */
    int32_t max_samples;  /* longest utterance, samples */
    float p_in;           /* magnitude exponent applied before the network (1.0 noncprs, 0.5 cprs) */
    float p_out;          /* magnitude exponent applied after the network  (1.0 noncprs, 2.0 cprs) */
    int32_t n_fft, hop, win;
    int32_t flags;        /* SE_CFG_* bits, 0 = defaults */
} se_config;

/* Model construction: `model = <Class>(...)` + `.cuda()`. */
int se_engine_create(const se_config* cfg, se_engine** out);
int se_engine_destroy(se_engine* e);

/* Error text of the last failing call on this handle (e == NULL: last se_engine_create failure). */
const char* se_last_error(const se_engine* e);

/* Weight load: one call per entry of `model.load_state_dict(torch.load('./BEST_MODEL/<name>.pth'))`
 * (e.g. CRN/crn_decode_vb.py:19).  `data` is HOST memory, copied; dtype 0 = float32, 1 = int64
 * (`num_batches_tracked` buffers: accepted and ignored).  Key names are the reference state-dict keys
 * (SURVEY.md Appendix D).  Unknown keys are rejected at finalize, like a strict load. */
int se_engine_set_tensor(se_engine* e, const char* key, const void* data, const int64_t* shape, int32_t ndim,
                         int32_t dtype);
/* End of load_state_dict (strict): checks every expected key is present with the right shape, folds
 * eval-mode BatchNorm into the adjacent convolution, packs weights for the MFMA kernels, uploads. Synchronises. */
int se_engine_finalize(se_engine* e);

/* `y = model(x)` under torch.no_grad()/eval(): model-only parity hook.  Shapes are the reference's
 * (SURVEY.md 8(a)), e.g. DCCRN [B,2,257,T] -> [B,2,257,T]; CRN [B,T,161] -> [B,T,161]. */
int se_forward(se_engine* e, const float* in_dev, const int64_t* in_shape, int32_t in_ndim, float* out_dev,
               void* stream);

/* Uformer's full return: `output, src, output_cplx, src_cplx = model(inputs, src)` (Uformer/uformer.py:172-287; the decode
 * script calls `model(x, x)[0]`, uformer_decode_vb.py:40).  inputs_dev / src_dev: waveforms [batch][n_samples] (dense rows).
 *   output_dev      [batch][se_output_samples(n)]   enhanced waveform (uformer.py:276) - what se_forward returns
 *   src_out_dev     [batch][se_output_samples(n)]   istft(stft(src)) (uformer.py:186), NULL = not wanted
 *   output_cplx_dev [batch][2][257][T]              the RI estimate the waveform is synthesised from (uformer.py:264-286)
 *   src_cplx_dev    [batch][2][257][T]              |S| e^{j angle S} of the source's STFT (uformer.py:187-194)
 * T = se_num_frames(e, n_samples); rows of T frames, T contiguous.  src_dev == NULL skips the two source outputs.
 * Only for engines created with SE_MODEL_UFORMER. */
int se_uformer_forward(se_engine* e, const float* inputs_dev, const float* src_dev, int32_t batch, int32_t n_samples,
                       float* output_dev, float* src_out_dev, float* output_cplx_dev, float* src_cplx_dev, void* stream);

/* The per-utterance body of `enhance(args)` for a batch of equal-length clips, device to device:
 * unit-RMS normalise -> (tail pad) -> STFT -> compress -> network (+mask) -> decompress -> iSTFT -> /c.
 * wav_in_dev [B][in_pitch] (first n_samples of each row valid), wav_out_dev [B][out_pitch]; the number of
 * output samples per utterance is se_output_samples(e, n_samples) (DCCRN returns the hop-padded length,
 * dccrn_decode_vb.py:59-64; Uformer hop*floor(L/hop); others L). */
int se_enhance_batch(se_engine* e, const float* wav_in_dev, int64_t in_pitch, int32_t batch, int32_t n_samples,
                     float* wav_out_dev, int64_t out_pitch, void* stream);
int64_t se_output_samples(const se_engine* e, int32_t n_samples);

/* The same loop body for a batch of clips of DIFFERENT lengths (the reference decodes one clip at a time, so every clip
 * has its own length: ~824 distinct lengths on VoiceBank+DEMAND, `for file_id in file_list`, DCCRN/dccrn_decode_vb.py:24).
 * `lengths` is a HOST array of `batch` sample counts; row b of wav_in_dev holds lengths[b] valid samples (the rest of the
 * row is ignored), row b of wav_out_dev receives se_output_samples(e, lengths[b]) samples followed by zeros up to
 * se_output_samples(e, max lengths).  Each row gets exactly the result of decoding it alone: its own unit-RMS scale, its
 * own reflect padding and frame count in the STFT / iSTFT, and utterance-wide statistics (InstanceNorm - CTSNet, G2Net,
 * TaylorSENet, e.g. CTSNet/Step1_network.py:121-145; FullSubNet's offline_laplace_norm, base_model.py:197-209) taken over
 * its own frames only; operators that look ahead in time (DCCRN's decoder, Uformer's symmetric dilated convs and its
 * attention over time) see zeros / masked keys past a clip's own last frame, as they do when the clip is decoded alone. */
int se_enhance_ragged(se_engine* e, const float* wav_in_dev, int64_t in_pitch, int32_t batch, const int32_t* lengths,
                      float* wav_out_dev, int64_t out_pitch, void* stream);

/* Frame-online ("streaming") decoding for the causal models - SURVEY.md 8(f) rank 4.  The reference only ever runs its
 * causal architectures offline (`for file_id in file_list`, whole utterance per forward, e.g. CRN/crn_decode_vb.py:33-52;
 * CRN.py:38 / :112-117 pad-top-1 + Chomp_T make every (de)conv look back exactly one frame, the LSTMs are unidirectional).
 * Here the same decode runs incrementally over `batch` parallel streams: se_stream_push() appends n_new samples per row,
 * transforms every STFT frame whose samples have all arrived, advances the network by those frames (one history frame per
 * conv layer and the LSTM (h, c) are carried in the engine) and returns the output samples that are now final - those
 * covered by no future frame - at the start of each row of out_dev; *n_out (host) = their count (same for all rows, 0 is
 * normal for short pushes: the algorithmic latency is n_fft / 2 + 1 samples plus up to one hop).  se_stream_flush() ends
 * the stream: the remaining frames (reflected right edge, as the offline STFT) and samples; the concatenated outputs equal
 * se_enhance_batch() of the whole signal sample for sample (up to fp32 rounding of the differently tiled recurrence).
*   c_dev: the utterance scale c (se_rms_scale) cannot be known before the utterance ends; the caller provides one value
 *          per stream (e.g. from a calibration run), NULL = 1.0; se_stream_begin_running() below estimates it on the fly.  With the offline c the two paths
 *          agree exactly - that is what the tests check.
 *   max_chunk_frames: frames advanced per internal step (latency / efficiency trade-off, default 16).
 * Supported: SE_MODEL_CRN, SE_MODEL_LSTM, SE_MODEL_GCRN, SE_MODEL_DPCRN, SE_MODEL_DCCRN (whose decoder looks six frames ahead:
 * its output is final six frames later than the others'), and SE_MODEL_CTSNET / SE_MODEL_TAYLORSENET / SE_MODEL_G2NET when
 * loaded with the cumulative-LayerNorm weights of the `_new` directories (CTSNet_new/Step1_network.py:213-286 - with the
 * InstanceNorm weights of the base directories the network needs the whole utterance and se_stream_begin fails; the engine
 * then carries up to 128 history frames per dilated conv and the running cLN sums).  Streams are limited to max_samples
 * of se_config. */
int se_stream_begin(se_engine* e, int32_t batch, int32_t max_chunk_frames, const float* c_dev, void* stream);
/* The same, for a caller that has no scale to give: the stream runs on a RUNNING unit-RMS scale.  After every push
 * c = sqrt(samples so far / their sum of squares) - the decode scripts' `c = np.sqrt(len(x) / np.sum(x ** 2.0))`
 * (e.g. CRN/crn_decode_vb.py:34) over what has been heard; the frames that push releases are transformed under that c and
 * taken back by it in the iSTFT (the overlap-add mixes frames of different c, each divided by its own).  A single push of a
 * whole utterance followed by se_stream_flush() therefore equals se_enhance_batch(); piecewise, the first frames see the
 * scale of a short prefix - the price of not knowing the future - and the output tracks the offline decode as the
 * estimate settles (tests/test_gpu_streaming.py).  Scaling the input by k scales the output by k exactly as offline. */
int se_stream_begin_running(se_engine* e, int32_t batch, int32_t max_chunk_frames, void* stream);
int se_stream_push(se_engine* e, const float* wav_dev, int64_t pitch, int32_t n_new, float* out_dev, int64_t out_pitch,
                   int32_t* n_out, void* stream);
int se_stream_flush(se_engine* e, float* out_dev, int64_t out_pitch, int32_t* n_out, void* stream);

/* Stage hooks, so each oracle-pinned stage can be diffed alone (engine-internal spectrogram layout
 * [B][2][F][T] re/im planes, T contiguous, row pitch = T).
 *   se_stft     : torch.stft / librosa.stft call of the model's decode script, fused with x*c and |X|^p_in.
 *                 c_dev (may be NULL -> no scaling) holds one scale per utterance.
 *   se_istft    : torch.istft / librosa.istft (+ division by c when c_dev != NULL), n_out samples per row.
 *   se_rms_scale: c = sqrt(L / sum x^2).
 *   se_num_frames / se_num_bins: T and F for n_samples. */
int se_rms_scale(se_engine* e, const float* wav_dev, int64_t pitch, int32_t batch, int32_t n_samples, float* c_dev,
                 void* stream);
int se_stft(se_engine* e, const float* wav_dev, int64_t pitch, int32_t batch, int32_t n_samples,
            const float* c_dev, float p_in, float* spec_dev, void* stream);
int se_istft(se_engine* e, const float* spec_dev, int32_t batch, int32_t n_frames, const float* c_dev,
             float* wav_dev, int64_t pitch, int32_t n_out, void* stream);
int32_t se_num_frames(const se_engine* e, int32_t n_samples);
int32_t se_num_bins(const se_engine* e);

/* The two halves of a decode loop's body around `model(feat)` as stage hooks of their own (se_frontend /
 * se_backend), so that the front end and the mask / decompress stage can be diffed against the oracle alone.
 *   se_frontend: c = sqrt(L / sum x^2); x * c (+ the script's tail pad); STFT; |X|^p_in e^{j angle X} with the engine's p_in
 *                (e.g. DCCRN/dccrn_decode_vb.py:26-42, LSTM/lstm_decode_vb.py:33-38, GCRN/gcrn_decode_vb.py:35-46).
 *                wav_dev [batch][pitch] -> c_dev [batch], spec_dev [batch][2][F][T]  (T = se_num_frames(n_samples)).
 *   se_backend : network output -> estimated spectrum -> |S|^p_out e^{j angle S} (the engine's p_out) -> iSTFT -> / c.
 *                kind SE_BACKEND_RI   : est_dev [batch][2][F][T] IS the estimated spectrum (complex-mapping scripts:
 *                                       GCRN/gcrn_decode_vb.py:47-58, DCCRN/dccrn_decode_vb.py:45-62); spec_dev unused
 *                     SE_BACKEND_MAG  : est_dev [batch][F][T] is a magnitude, the phase is the noisy spectrum's
 *                                       (LSTM/lstm_decode_vb.py:47-52, CRN/crn_decode_vb.py:46-52)
 *                     SE_BACKEND_CMASK: est_dev [batch][2][F][T] is a complex ratio mask applied to spec_dev
 *                                       (FullSubNet/fullsubnet_sa_decode_vb.py:56-72, DPCRN/DPCRN.py:33-42)
 *                spec_dev: the front end's output (kinds MAG / CMASK); c_dev may be NULL (no division); n_out samples per row. */
#define SE_BACKEND_RI 0
#define SE_BACKEND_MAG 1
#define SE_BACKEND_CMASK 2
int se_frontend(se_engine* e, const float* wav_dev, int64_t pitch, int32_t batch, int32_t n_samples, float* c_dev,
                float* spec_dev, void* stream);
int se_backend(se_engine* e, int32_t kind, const float* est_dev, const float* spec_dev, int32_t batch, int32_t n_frames,
               const float* c_dev, float* wav_dev, int64_t pitch, int32_t n_out, void* stream);

/* Kernel time of the dominant kernel family (f32-MFMA implicit-GEMM convolution) inside the last
 * se_enhance_batch / se_forward, measured with HIP events on the stream each launch runs on (the caller's `stream` and the
 * engine's auxiliary streams: launches that overlap on two streams both count); enabled by se_set_profiling(e, 1).
 * Returns accumulated milliseconds and the launch count through the out-params. */
int se_set_profiling(se_engine* e, int32_t on);
int se_get_profile(se_engine* e, double* gemm_ms, int64_t* gemm_launches, double* gemm_flops);

/* The HBM-bound front / back-end kernels of the last profiled se_enhance_batch / se_enhance_ragged, per stage: accumulated
 * HIP-event milliseconds on `stream`, launch count, and the stage's ALGORITHMIC bytes (what it has to read and write once:
 * RMS 4L; STFT 4L + 8FT (+4FT magnitudes); mask apply / decompress 16-24 FT; iSTFT + overlap-add 8FT + 4L - per utterance,
 * SURVEY.md 8(d)).  bytes / ms against the HBM peak is the stage's roofline fraction (bench.py `roofline_stages`). */
enum se_stage { SE_STAGE_RMS = 0, SE_STAGE_STFT = 1, SE_STAGE_MASK = 2, SE_STAGE_ISTFT = 3 };
int se_get_stage_profile(se_engine* e, int32_t stage, double* ms, int64_t* launches, double* bytes);

/* Sample-rate conversion in front of the path: librosa.resample(y, sr_in, sr_out, fix=True, scale=False) as called at
 * DCCRN/dccrn_decode_vb.py:26 and LSTM/lstm_decode_vb.py:34 (VoiceBank+DEMAND ships at 48 kHz; the models run at 16 kHz).
 * Stateless (no engine handle; errors through se_last_error(NULL)).  Device pointers; `n_in` samples per row in,
 * se_resample_samples(n_in, sr_in, sr_out) = ceil(n_in * sr_out / sr_in) samples per row out. */
int64_t se_resample_samples(int32_t n_in, int32_t sr_in, int32_t sr_out);
int se_resample(const float* in_dev, int64_t in_pitch, int32_t batch, int32_t n_in, int32_t sr_in, int32_t sr_out,
                float* out_dev, int64_t out_pitch, void* stream);

/* The two ends of `enhance(args)`: `feat_wav, orig_fs = sf.read(path)` hands out int16 / 32768 as floats, and
 * `sf.write(path, y, fs)` stores PCM_16 (soundfile's default subtype for .wav: round to nearest, clipped) - e.g.
 * DCCRN/dccrn_decode_vb.py:25,64.  Both conversions are exact in fp32, so they run on the device and the host moves raw
 * 2-byte samples only.  Stateless; rows of `n` samples, pitches in elements. */
int se_pcm16_decode(const int16_t* in_dev, int64_t in_pitch, int32_t batch, int32_t n, float* out_dev, int64_t out_pitch,
                    void* stream);
int se_pcm16_encode(const float* in_dev, int64_t in_pitch, int32_t batch, int32_t n, int16_t* out_dev, int64_t out_pitch,
                    void* stream);

/* ABI version of this header. */
int32_t se_abi_version(void);   /* 2: se_enhance_ragged, se_get_stage_profile, se_stream_*; 3: se_uformer_forward, se_pcm16_*; 4: se_stream_begin_running; 5: se_frontend, se_backend */

#ifdef __cplusplus
}
#endif
#endif /* SE_ENGINE_H */
