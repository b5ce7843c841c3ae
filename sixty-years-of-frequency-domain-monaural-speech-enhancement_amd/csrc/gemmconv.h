// Tap-table implicit-GEMM convolution on f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// One kernel family carries every dense layer of the zoo: Conv2d, the parity
// classes of a strided ConvTranspose2d, 1x1 convs / Linear layers, the LSTM
// input projection and the LSTM recurrent step (with the cell update fused in
// the epilogue).  Activations live as [B][C][F][T] with T contiguous ("features
// x samples"); a weight matrix is packed K-major so that both MFMA operands are
// read from LDS with 32 consecutive lanes on 32 consecutive floats.
//
//   acc[m](z,b,q,t) = sum_{ci<C0+C1} sum_{j<ntaps} A_z[(ci,j)][m] * X_z(b, ci, q*si + df[j], t + dt[j])
//   dst_z[b][m][q*so+po][t] = epilogue(acc[m] + bias[m])
//
// X is the virtual channel-concat of src0 (C0 channels) and src1 (C1 channels),
// zero outside [0,Fin) x [0,Tin).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

namespace se {

constexpr int GC_MAX_TAPS = 16;
constexpr int GC_MAX_ROWS = 8;
constexpr int GC_MAX_KCP = 48;
constexpr int GC_TAB_KOFF = GC_MAX_ROWS + 2 * GC_MAX_TAPS;   // start of the per-K-row patch offsets in GCParams::tab
// K rows of one staged chunk and patch elements staged per thread, by output-channel tile
constexpr int gc_kcp_max(int BM) { return 32; }
constexpr int gc_bld_max(int BM) { return (BM >= 128 || BM <= 32) ? 9 : 13; }
// resident blocks per CU the kernel is register-budgeted for: small-M tiles do little matrix work per staged K row and
// hide the global-load latency with occupancy instead
// (the 32-row tile keeps one accumulator tile per wave: with 9 patch slots it fits 78 VGPRs, i.e. 6 workgroups per CU - its
// chunks are short, 10 MFMAs per wave and barrier, and more resident workgroups fill the gaps: DPCRN / Uformer + 1.5 %)
constexpr int gc_blocks_per_cu(int BM) { return BM <= 32 ? 6 : 3; }

enum Act : int { ACT_NONE = 0, ACT_PRELU = 1, ACT_ELU = 2, ACT_SOFTPLUS = 3, ACT_SIGMOID = 4, ACT_TANH = 5, ACT_RELU = 6 };
enum Epi : int {
    EPI_ACT = 0,     // dst = act(acc + bias)
    EPI_LSTM = 1,    // rows are gate-interleaved (4j+{i,f,g,o}); dst = h_t, c updated in place (GCParams::gru: the GRU cell on the same 4-row layout)
    EPI_GLU = 2,     // rows are pair-interleaved (2j, 2j+1): dst[j] = (a+bias) * sigmoid(g+bias)
    EPI_ADD = 3,     // dst = act(acc + bias) + res   (res laid out like dst)
    EPI_MUL = 4,     // dst = act(acc + bias) * aux   (gated TCM branches, CTSNet/Step1_network.py:184)
    EPI_CMB = 5,     // dst = prelu((aux +- acc) * post_scale + post_shift): the last two of Gauss' three products finish the complex
                     // layer in their own store (aux = k1; GCParams::cmb_neg bit z: minus; per-z rows of post_scale / post_shift /
                     // slope at stride ps_z) - gauss.h
};

// launches that produce at most this many frames per row go to the thin kernel (frame-online chunks)
constexpr int GC_THIN_NT = 8;

struct GCParams {
    const float* A;          // packed weights [nchunks][KCp][Mp]
    const float* Ws;         // direct small-M path (M <= 4): plain weights [z][ci][tap][MM]; nullptr -> MFMA path
    const float* bias;       // [M] or nullptr
    const float* slope;      // [M] PReLU slopes or nullptr
    const float* post_scale; // EPI_GLU: per output channel scale / shift applied after the gate product (eval BatchNorm)
    const float* post_shift;
    const float* bias_pad;   // bias for output rows fo < pad_lo (a zero-padded frequency row: no conv bias, only the folded BN shift)
    int pad_lo;
    long ps_z;               // EPI_CMB: per-z element stride of post_scale / post_shift / slope
    int cmb_neg;             // EPI_CMB: bit z set -> dst = f(aux - acc), else f(aux + acc)
    const float* cmb_i;      // EPI_CMB, optional (with cmb_s): the OTHER finished plane (I, laid out like dst) ...
    float* cmb_s;            // ... and where their sum goes: S = dst + I, the third plane of a three-plane tensor (gauss.h)
    const float* src0;
    const float* src1;
    float* dst;
    float* dst_elu;          // EPI_GLU, optional: a second tensor laid out like dst that receives ELU(stored value) - GCRN's decoders
                             // read elu(e_k) of every encoder output beside e_k itself (GCRN_noncprs.py:147-157); gc_kernel only
    const float* aux;        // EPI_LSTM: gate pre-activations gx; EPI_ADD: residual
    float* cell;             // EPI_LSTM: cell state, laid out like dst (channels = M/4), updated in place
    long A_z, bias_z, src0_z, src1_z, dst_z, aux_z, cell_z;   // per-z (blockIdx.z) element strides
    long s0_b, s0_c, s0_f;   // src0 element strides (t stride is 1)
    long s1_b, s1_c, s1_f;
    long d_b, d_c, d_f;      // dst strides (also cell strides)
    long x_b, x_c, x_f;      // aux strides
    int C0, C1;
    int Fin, Tin;
    int B, Q, Tout, M, Mp;
    int si, so, po;
    int ntaps, nrows, dtmin, Wp;
    int CI_C, KC, KCp, nchunks;
    int act, epi;
    int n_ttiles, n_mtiles, Z;
    int dbg;                 // ablation switches for tuning (0 in production): 1 no global loads, 4 no MFMA, 8 no epilogue
    int first_step;          // EPI_LSTM: 1 -> h_{-1} = c_{-1} = 0 (nchunks forced to 0 by the host)
    int gru;                 // EPI_LSTM with the GRU cell: rows 4u + {r, z, n, 0}; aux rows 4u + {gx_r, gx_z, gx_n, b_hn}; `cell` holds h_{t-1}
    unsigned long long* timing;   // tuning builds (-DGC_TIMING): per-phase s_memtime accumulators, else unused
    const unsigned* desc4;   // descriptors of the patch seen as 16 B groups (same packing as desc, w = first frame)
    int t_base;              // first frame of time tile 0 of this launch (tail launches start at the last tile)
    int tb_soft;             // 1: the frames just below t_base may be produced again (their inputs are there): the MFMA path
                             // then starts at the multiple of 4 below t_base
    int pw4;                 // per launch: 1 -> stage the patch in 16 B groups
    int causal;              // no tap looks ahead in time (dt <= 0 for every tap)
    const int* tlen;         // per launch (optional, device [B]): output frames >= tlen[b] are stored as zeros (MFMA path)
    int qt2, qq_off, Qt;     // per launch: two-row tiles (LDS offset of the second row's patch rows; row tiles per plane)
    int pair, po2, fo_lim;   // direct path: both parity classes of a transposed conv as 2 * pair virtual output channels (0 = off)
    short tdf[GC_MAX_TAPS], tdt[GC_MAX_TAPS];     // tap offsets (frequency rows, frames) by value, for the thin kernel
    int nbuf;                // per launch (resident-K form of the kernel): staging buffers = chunks of the longer source
    int trim;                // per launch: 1 -> 16 B groups that straddle the end of a row are cut back to Tin in LDS
    const unsigned* desc;    // host-built patch-slot descriptors [NB][256]: w | r << 12 | cil << 16 | staged << 31
    const int* tab;          // device table: row_df[GC_MAX_ROWS], tap_row[GC_MAX_TAPS], tap_dt[GC_MAX_TAPS], koff[GC_MAX_KCP + 8]    // optional (EPI_ACT on 64-row tiles, EPI_GLU): per (b, output channel, output frequency row, group of 32 frames)
    // partial (sum, sum of squares) of the values this launch stores - [B][Mo][Fstat][ceil(Tout / 32)][2] floats - so that
    // the InstanceNorm that follows does not have to read the plane for its statistics (blocks.h: conv_norm2d_prelu)
    float* stats;
    long st_b, st_c, st_f;   // float strides of the statistics tensor
    // optional (same tile configurations, one m-tile): per (b, output frequency row, frame) the (sum, sum of squares) over ALL
    // output channels of the values this launch stores - [B][Fout][Tout][2] floats - for the CumulativeLayerNorm behind the layer
    // (k_misc.hip: cln_scan_parts_kernel sums the rows and scans the frames; the norm's own statistics pass over the tensor -
    // 6 % of a G2Net_new / TaylorSENet_new step - is not launched)
    float* cstats;
    long cs_b, cs_f;
    // optional (EPI_ACT / EPI_ADD on the MFMA path; Uformer's interaction of the two branches, fusion.py:13-19, folded into
    // the magnitude branch's last launch): fz = the complex branch's tensor, real plane at fz, imaginary plane fz_im floats
    // behind it, laid out like dst with its own strides.  The stored value v and the complex pair (re, im) at the same
    // (b, channel, row, frame) become  re + sig(v), im + sig(v), v + sig(|re + i im|)  - the pair is rewritten in place
    float* fz;
    long fz_b, fz_c, fz_f, fz_im;
    long fz_s;               // != 0 (three-plane tensors, gauss.h): the rewritten pair's sum re + im is stored fz_s floats from the real plane
    // optional (gc_nrm_supported(): EPI_ACT on 64-row tiles, causal taps, <= GC_NRM_MAXC input channels): the sources are RAW conv
    // outputs whose InstanceNorm + PReLU has not been applied - per (b, channel) float4 {scale = rstd * gamma, shift = beta - mean *
    // scale, slope - 1, x0 = -shift / scale} (k_misc.hip: instnorm_finalize_kernel).  The kernel applies
    //   y = fma(x, scale, shift);  y + (slope - 1) * min(y, 0)
    // to every B-operand fragment on its way from LDS to the matrix instruction, so the normalised tensor never exists in HBM
    // (round 5: the stand-alone InstanceNorm passes of the U^2-Net levels were 13-15 % of a G2Net / TaylorSENet step).  Zero
    // padding stays zero: K rows whose frequency row lies outside the plane get scale = shift = 0, left-pad frames are staged as x0.
    // nullptr for a source: that source is consumed as it is.
    const float* nrm0;
    const float* nrm1;
    // per launch (gc_launch; equal-length offline batches, causal taps with <= 4 frames of look-back, 16 B staging groups): the
    // column tiles run over the batch rows' 32-frame UNITS flattened - unit u = b * flat_upr + (t >> 5) - instead of over one
    // row's frames, so that only the very last tile of a (z, q) plane is partly filled: a T = 401 row is 13 units, which the
    // 64 x 256 tile (8 units) covered with 2 tiles = 16 slots (81 %), the 64 x 128 / 32 x 128 tiles (4 units) with 4 tiles.
    // A unit is staged with its own halo (32 + 4 columns of LDS per patch row), its own batch row and first frame.
    int flat_upr;            // units per row (0: off)
    int flat_units;          // B * flat_upr
};
constexpr int GC_NRM_MAXC = 128;

// Device tables of one patch geometry (owned by the plan)
struct GCGeom {
    int* tab = nullptr;
    unsigned* desc = nullptr;
    unsigned* desc4 = nullptr;
};
// Narrow geometry for the last time tile of a row (BN = 0: unused)
struct GCTail {
    int BN = 0, Wp = 0;
    GCGeom g;
};

// Tap extents of a layer on the direct (<= 4 output channels) path
struct GCSmallGeom {
    int dfmin = 0, dfmax = 0, dtmin = 0, dtmax = 0;
};

// Host-side description of one dense layer, built once at finalize.
struct GCPlan {
    GCParams p{};            // static part (taps, chunking, weights); pointers for activations filled per launch
    int BM = 128, BN = 128;  // tile config
    int lookback = 0;        // frames of history the taps reach back (max -dt)
    double flop_scale = 1.0; // algorithmic / executed multiply-adds (fused parity pairs carry zero-weight taps)
    float* dA = nullptr;     // device copies owned by the plan
    float* dWs = nullptr;
    unsigned* dDesc = nullptr;
    unsigned* dDesc4 = nullptr;
    GCSmallGeom small;       // direct path only
    bool tail_split = false; // tail[0] may be used as a separate launch for the last time tile
    GCTail tail[3];          // [0]: 32-column geometry for a mostly empty last time tile; [1]: 64-column geometry of the whole layer for small launches; [2]: 256-column geometry of a 64-row layer for big launches
    GCTail qt2;              // two-row geometry: 2 output rows x 64 frames per tile (BN = 0: unused)
    GCTail flat[2];          // unit-flattened geometries (GCParams::flat_upr): [0] 128 columns = 4 units, [1] 256 columns = 8 units of a
                             // 64-row layer; Wp = units x flat_uw (BN = 0: unused)
    int flat_uw = 0;         // LDS columns per unit and patch row: 32 (pointwise) or 36 (causal taps, <= 4 frames of look-back)
    int qt2_nrows = 0, qt2_qoff = 0;
    float* dBias = nullptr;
    float* dSlope = nullptr;
    int* dTab = nullptr;
    float* dBiasPad = nullptr;
    float* dPostScale = nullptr;
    float* dPostShift = nullptr;
};

struct TapSpec {
    int ntaps = 0;
    int df[GC_MAX_TAPS];
    int dt[GC_MAX_TAPS];
};

// Build a plan.  w_logical[m][ci][j] (row-major, M x Cin x ntaps) are the effective real weights
// (BatchNorm folded, complex structure expanded) for this tap set.  bias/slope may be empty.
GCPlan gc_make_plan(int M, int Cin, const TapSpec& taps, const std::vector<float>& w_logical,
                    const std::vector<float>& bias, const std::vector<float>& slope, int act, int epi,
                    int si, int so, int po, int tout_hint, int z = 1, int C0split = -1);
void gc_free_plan(GCPlan& pl);

// Device ranges (the engine arenas) whose tensors may be over-read by <= 12 B past their end (pointwise 16 B staging).
void gc_register_overread_range(const void* lo, size_t bytes);
void gc_unregister_overread_range(const void* lo);

// Launch: p must have src/dst pointers, strides, B/Q/Tout/Fin/Tin/C0/C1 filled in.
// true when launches of this plan can emit the per-tile statistics of GCParams::stats
bool gc_stats_supported(const GCPlan& pl);
// true when launches of this plan can normalise their sources on the fly (GCParams::nrm0 / nrm1)
bool gc_nrm_supported(const GCPlan& pl);
void gc_launch(const GCPlan& pl, GCParams p, hipStream_t stream);
// two launches that would both take the thin path (the frequency-parity classes of a transposed conv on a few frames) as one;
// false: nothing was launched
bool gc_launch_thin_pair(const GCParams& p0, const GCParams& p1, hipStream_t stream);

}  // namespace se
