// Host-side weight preparation: turns reference state-dict tensors (torch layouts) into the effective real
// weight matrices the tap-table implicit-GEMM kernel consumes (eval-mode BatchNorm folded, complex convs
// expanded to real 2x2 block form, transposed convs split into output-parity classes, LSTM gates interleaved).
#pragma once
#include "common.h"
#include "gemmconv.h"

namespace se {

// Effective real weights of one dense layer: w[m][ci][j], j indexes taps (kf-major: j = kf_idx * nkt + kt_idx).
struct DenseW {
    int M = 0, Cin = 0, nkf = 1, nkt = 1;
    std::vector<float> w;      // [M][Cin][nkf*nkt]
    std::vector<float> bias;   // [M] (always present; zeros if the layer has none)
    int ntaps() const { return nkf * nkt; }
    float& at(int m, int ci, int kf, int kt) { return w[((size_t)m * Cin + ci) * ntaps() + kf * nkt + kt]; }
    float at(int m, int ci, int kf, int kt) const { return w[((size_t)m * Cin + ci) * ntaps() + kf * nkt + kt]; }
};

const HostTensor& sd_get(const StateDict& sd, const std::string& key, std::vector<int64_t> shape = {});

// nn.Conv2d weight [Co][Ci][k0][k1]; tf_order: (k0,k1) = (time,freq) as in CRN/DPCRN; else (freq,time) as in DCCRN.
DenseW conv_weights(const HostTensor& w, const HostTensor* b, bool tf_order);
// nn.ConvTranspose2d weight [Ci][Co][k0][k1] -> M = Co, Cin = Ci (taps still indexed by the kernel position).
DenseW deconv_weights(const HostTensor& w, const HostTensor* b, bool tf_order);
// nn.Linear / 1x1: weight [M][Cin]
DenseW linear_weights(const HostTensor& w, const HostTensor* b);
// complexnn-style complex layer from its real_conv / imag_conv halves (channels = [real half ; imag half]):
//   out_r = Wr*x_r - Wi*x_i + (br - bi);  out_i = Wi*x_r + Wr*x_i + (br + bi)
DenseW complex_expand(const DenseW& wr, const DenseW& wi);
// eval-mode BatchNorm on the output channels, folded into w / bias.
void fold_bn(DenseW& d, const HostTensor& gamma, const HostTensor& beta, const HostTensor& mean,
             const HostTensor& var, float eps = 1e-5f);
// eval-mode BatchNorm on the INPUT channels of a pointwise layer (BatchNorm1d before an LSTM, LSTM/LSTM.py:25).
void fold_bn_input(DenseW& d, const HostTensor& gamma, const HostTensor& beta, const HostTensor& mean,
                   const HostTensor& var, float eps = 1e-5f);
// new input channel c reads old channel perm[c]
void permute_cin(DenseW& d, const std::vector<int>& perm);
// rows: new row r = old row perm[r]
void permute_rows(DenseW& d, const std::vector<int>& perm);
// LSTM gate interleave: torch rows [i;f;g;o] (each H) -> row 4j+g
std::vector<int> lstm_gate_perm(int H);
// concatenate along K (input channels) / along M (rows)
DenseW concat_cin(const DenseW& a, const DenseW& b, float scale_b = 1.f);
DenseW concat_rows(const DenseW& a, const DenseW& b);
// rows a0,b0,a1,b1,... (GLU pairs: value row 2j, gate row 2j+1)
DenseW interleave_rows(const DenseW& a, const DenseW& b);
// attach an eval-BatchNorm (applied AFTER the gate product) to a GLU plan
void set_post_bn(GCPlan& pl, const HostTensor& gamma, const HostTensor& beta, const HostTensor& mean, const HostTensor& var,
                 float eps = 1e-5f);
std::vector<float> prelu_slopes(const HostTensor& w, int M);

// Regular conv: out[f][t] = sum w[kf][kt] x[f*sf - pf + kf*dil_f][t - pt_left + kt*dil_t]
GCPlan make_conv_plan(const DenseW& d, int sf, int pf, int pt_left, int dil_f, int dil_t, int act,
                      const std::vector<float>& slope, int epi, int tout_hint, int C0split = -1);

// Transposed conv with frequency stride sf (time stride 1):
//   out[fo][to] = sum_{kf,kt: (fo+pf-kf) % sf == 0} x[(fo+pf-kf)/sf][to + toff - kt] w[kf][kt]
// split in sf output-parity classes, each a dense tap-table conv.  pf < 0 expresses a left frequency pad.
struct DeconvPlan {
    std::vector<GCPlan> par;   // one per parity class (classes with no taps are dropped -> bias-only rows unsupported)
    int sf = 1;
    // layers back to <= 2 channels (DCCRN's last transposed conv, 64 -> 2): both parity classes as ONE launch of the direct
    // path - 2 x M virtual output channels over the union of the classes' taps (a tap a class does not have gets zero
    // weights), so the input plane is read once instead of once per class
    GCPlan pair;
    bool has_pair = false;
};
DeconvPlan make_deconv_plan(const DenseW& d, int sf, int pf, int toff, int act, const std::vector<float>& slope,
                            int tout_hint, int C0split = -1, const std::vector<float>* bias_pad = nullptr,
                            int epi = EPI_ACT);
void free_deconv_plan(DeconvPlan& p);
// run_conv / run_deconv of these plans store zeros past a ragged row's own last frame themselves (MFMA path), so the
// models that need zero tails in front of look-ahead operators can skip their launch_zero_tail
inline bool conv_zeroes_tail(const GCPlan& pl) { return pl.p.Ws == nullptr; }
inline bool conv_zeroes_tail(const DeconvPlan& pl) { return !pl.has_pair && !pl.par.empty() && pl.par[0].p.Ws == nullptr; }

// Convenience launcher for [B][C][F][T]-layout tensors (row pitch Tp).
struct Act4 {          // a view of an activation tensor
    const float* p = nullptr;
    int C = 0, F = 0;
    long sb = 0, sc = 0, sf = 0;   // element strides; t stride 1
    // optional: the tensor is a RAW conv output whose InstanceNorm + PReLU the consumer applies on the fly - [B][C] float4
    // parameters (GCParams::nrm0 / nrm1; only for plans with conv_nrm_supported())
    const float* nrm = nullptr;
    Act4 with_nrm(const float* n) const {
        Act4 a = *this;
        a.nrm = n;
        return a;
    }
};
inline Act4 act4(const float* p, int C, int F, int Tp) { return Act4{p, C, F, (long)C * F * Tp, (long)F * Tp, (long)Tp}; }

struct Profiler;
// stats (optional): [B][dstC][Fout][ceil(T / 32)][2] partial sums of the stored output (GCParams::stats)
// fz (optional, plans for which conv_folds_interaction() holds): the complex branch's tensor [B][2 * dstC][Fout][Tp] whose
// interaction with this launch's output is folded into the store (GCParams::fz); fz_planes = 3: fz is the REAL plane of a
// three-plane tensor [B][3 * dstC][Fout][Tp] = [S | R | I] (gauss.h; the sum plane is the caller's to refresh)
// colstats: `stats` is [B][Fout][T][2] - per (b, output row, frame) sums over all output channels (GCParams::cstats, for a
// CumulativeLayerNorm behind the layer) - instead of the per-channel partials of an InstanceNorm
void run_conv(const GCPlan& pl, const Act4& s0, const Act4* s1, float* dst, int dstC, int Fout, int B, int T, int Tp,
              hipStream_t st, Profiler* prof = nullptr, float* stats = nullptr, int t_base = 0, float* fz = nullptr, int fz_planes = 2,
              bool colstats = false, float* dst_elu = nullptr);      // dst_elu: GCParams::dst_elu (gated layers on the matrix path)
// true: a folded interaction into a three-plane tensor (fz_planes = 3) also stores its sum plane S = R + I
bool conv_fold_writes_sum();
inline bool conv_folds_interaction(const GCPlan& pl) { return pl.p.Ws == nullptr && (pl.p.epi == EPI_ACT || pl.p.epi == EPI_ADD); }
inline bool conv_folds_interaction(const DeconvPlan& pl) {
    if (pl.has_pair || pl.par.empty()) return false;
    for (const auto& g : pl.par)
        if (!conv_folds_interaction(g)) return false;
    return true;
}
// frame-online chunks: only output frames [t_base, t_out) of the T-frame window are produced (t_out < 0: up to T); tb_soft:
// GCParams::tb_soft
void run_deconv(const DeconvPlan& pl, const Act4& s0, const Act4* s1, float* dst, int dstC, int Fout, int B, int T,
                int Tp, hipStream_t st, Profiler* prof = nullptr, float* stats = nullptr, int t_base = 0, int t_out = -1,
                bool tb_soft = false, float* fz = nullptr, int fz_planes = 2, bool colstats = false);
bool conv_stats_supported(const GCPlan& pl);
bool conv_nrm_supported(const GCPlan& pl);
bool deconv_nrm_supported(const DeconvPlan& pl);
bool deconv_stats_supported(const DeconvPlan& pl);

// HIP-event timing of the dominant kernel family (gemmconv launches) on the launch stream.
struct Profiler {
    bool on = false;
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    double flops = 0.0;
    double fscale = 1.0;    // algorithmic / executed frames: PadFrames runs a batch with the frame count rounded up to whole 16 B
                            // groups; the FLOPs reported are those of the clip's own frame count, not of the padding
    long launches = 0;
    std::vector<double> fls;      // FLOPs per timed launch (SE_PROF_DUMP=1: total_ms() prints one line per launch, tools/profl.py)
    void begin(hipStream_t st);
    void end(hipStream_t st, double fl);
    void reset() { used = 0; flops = 0.0; launches = 0; fls.clear(); }
    double total_ms();
    ~Profiler();
};
void gc_launch_prof(const GCPlan& pl, const GCParams& p, hipStream_t st, Profiler* prof);

}  // namespace se
