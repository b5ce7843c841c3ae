#include "layers.h"
#include "kernels.h"
#include <algorithm>
#include <cmath>

namespace se {

const HostTensor& sd_get(const StateDict& sd, const std::string& key, std::vector<int64_t> shape) {
    auto it = sd.find(key);
    SE_CHECK(it != sd.end(), "missing state-dict key '" + key + "'");
    if (!shape.empty()) {
        bool ok = it->second.shape.size() == shape.size();
        for (size_t i = 0; ok && i < shape.size(); ++i) ok = it->second.shape[i] == shape[i];
        if (!ok) {
            std::string got, want;
            for (auto s : it->second.shape) got += std::to_string(s) + ",";
            for (auto s : shape) want += std::to_string(s) + ",";
            SE_CHECK(false, "shape mismatch for '" + key + "': got [" + got + "] want [" + want + "]");
        }
    }
    return it->second;
}

static void fill_bias(DenseW& d, const HostTensor* b) {
    d.bias.assign(d.M, 0.f);
    if (b) {
        SE_CHECK((int)b->numel() == d.M, "bias length");
        std::copy(b->data.begin(), b->data.end(), d.bias.begin());
    }
}

DenseW conv_weights(const HostTensor& w, const HostTensor* b, bool tf_order) {
    SE_CHECK(w.shape.size() == 4, "conv weight must be 4-D");
    DenseW d;
    d.M = (int)w.shape[0];
    d.Cin = (int)w.shape[1];
    const int k0 = (int)w.shape[2], k1 = (int)w.shape[3];
    d.nkf = tf_order ? k1 : k0;
    d.nkt = tf_order ? k0 : k1;
    d.w.resize((size_t)d.M * d.Cin * k0 * k1);
    for (int m = 0; m < d.M; ++m)
        for (int c = 0; c < d.Cin; ++c)
            for (int a = 0; a < k0; ++a)
                for (int e = 0; e < k1; ++e) {
                    const int kf = tf_order ? e : a, kt = tf_order ? a : e;
                    d.at(m, c, kf, kt) = w.data[(((size_t)m * d.Cin + c) * k0 + a) * k1 + e];
                }
    fill_bias(d, b);
    return d;
}

DenseW deconv_weights(const HostTensor& w, const HostTensor* b, bool tf_order) {
    SE_CHECK(w.shape.size() == 4, "deconv weight must be 4-D");
    DenseW d;
    d.Cin = (int)w.shape[0];
    d.M = (int)w.shape[1];
    const int k0 = (int)w.shape[2], k1 = (int)w.shape[3];
    d.nkf = tf_order ? k1 : k0;
    d.nkt = tf_order ? k0 : k1;
    d.w.resize((size_t)d.M * d.Cin * k0 * k1);
    for (int c = 0; c < d.Cin; ++c)
        for (int m = 0; m < d.M; ++m)
            for (int a = 0; a < k0; ++a)
                for (int e = 0; e < k1; ++e) {
                    const int kf = tf_order ? e : a, kt = tf_order ? a : e;
                    d.at(m, c, kf, kt) = w.data[(((size_t)c * d.M + m) * k0 + a) * k1 + e];
                }
    fill_bias(d, b);
    return d;
}

DenseW linear_weights(const HostTensor& w, const HostTensor* b) {
    SE_CHECK(w.shape.size() == 2, "linear weight must be 2-D");
    DenseW d;
    d.M = (int)w.shape[0];
    d.Cin = (int)w.shape[1];
    d.w = w.data;
    fill_bias(d, b);
    return d;
}

DenseW complex_expand(const DenseW& wr, const DenseW& wi) {
    SE_CHECK(wr.M == wi.M && wr.Cin == wi.Cin && wr.nkf == wi.nkf && wr.nkt == wi.nkt, "complex halves differ");
    DenseW d;
    d.M = 2 * wr.M;
    d.Cin = 2 * wr.Cin;
    d.nkf = wr.nkf;
    d.nkt = wr.nkt;
    const int nt = d.ntaps();
    d.w.assign((size_t)d.M * d.Cin * nt, 0.f);
    d.bias.assign(d.M, 0.f);
    for (int m = 0; m < wr.M; ++m) {
        for (int c = 0; c < wr.Cin; ++c)
            for (int j = 0; j < nt; ++j) {
                const float r = wr.w[((size_t)m * wr.Cin + c) * nt + j], i = wi.w[((size_t)m * wr.Cin + c) * nt + j];
                d.w[((size_t)m * d.Cin + c) * nt + j] = r;                                // real out <- real in
                d.w[((size_t)m * d.Cin + wr.Cin + c) * nt + j] = -i;                       // real out <- imag in
                d.w[((size_t)(wr.M + m) * d.Cin + c) * nt + j] = i;                       // imag out <- real in
                d.w[((size_t)(wr.M + m) * d.Cin + wr.Cin + c) * nt + j] = r;              // imag out <- imag in
            }
        d.bias[m] = wr.bias[m] - wi.bias[m];
        d.bias[wr.M + m] = wr.bias[m] + wi.bias[m];
    }
    return d;
}

void fold_bn(DenseW& d, const HostTensor& gamma, const HostTensor& beta, const HostTensor& mean, const HostTensor& var,
             float eps) {
    SE_CHECK((int)gamma.numel() == d.M && (int)var.numel() == d.M, "BatchNorm channel count");
    const size_t per = (size_t)d.Cin * d.ntaps();
    for (int m = 0; m < d.M; ++m) {
        const double s = (double)gamma.data[m] / std::sqrt((double)var.data[m] + (double)eps);
        for (size_t i = 0; i < per; ++i) d.w[m * per + i] = (float)(d.w[m * per + i] * s);
        d.bias[m] = (float)(((double)d.bias[m] - mean.data[m]) * s + beta.data[m]);
    }
}

void fold_bn_input(DenseW& d, const HostTensor& gamma, const HostTensor& beta, const HostTensor& mean,
                   const HostTensor& var, float eps) {
    SE_CHECK((int)gamma.numel() == d.Cin && d.ntaps() == 1, "input BatchNorm fold needs a pointwise layer");
    for (int m = 0; m < d.M; ++m) {
        double badd = 0.0;
        for (int c = 0; c < d.Cin; ++c) {
            const double s = (double)gamma.data[c] / std::sqrt((double)var.data[c] + (double)eps);
            const double sh = (double)beta.data[c] - (double)mean.data[c] * s;
            const double w = d.w[(size_t)m * d.Cin + c];
            badd += w * sh;
            d.w[(size_t)m * d.Cin + c] = (float)(w * s);
        }
        d.bias[m] = (float)(d.bias[m] + badd);
    }
}

void permute_cin(DenseW& d, const std::vector<int>& perm) {
    SE_CHECK((int)perm.size() == d.Cin, "cin perm size");
    const int nt = d.ntaps();
    std::vector<float> nw(d.w.size());
    for (int m = 0; m < d.M; ++m)
        for (int c = 0; c < d.Cin; ++c)
            for (int j = 0; j < nt; ++j) nw[((size_t)m * d.Cin + c) * nt + j] = d.w[((size_t)m * d.Cin + perm[c]) * nt + j];
    d.w.swap(nw);
}

void permute_rows(DenseW& d, const std::vector<int>& perm) {
    SE_CHECK((int)perm.size() == d.M, "row perm size");
    const size_t per = (size_t)d.Cin * d.ntaps();
    std::vector<float> nw(d.w.size()), nb(d.M);
    for (int m = 0; m < d.M; ++m) {
        std::copy(d.w.begin() + perm[m] * per, d.w.begin() + (perm[m] + 1) * per, nw.begin() + m * per);
        nb[m] = d.bias[perm[m]];
    }
    d.w.swap(nw);
    d.bias.swap(nb);
}

std::vector<int> lstm_gate_perm(int H) {
    std::vector<int> p(4 * H);
    for (int j = 0; j < H; ++j)
        for (int g = 0; g < 4; ++g) p[4 * j + g] = g * H + j;
    return p;
}

DenseW concat_cin(const DenseW& a, const DenseW& b, float scale_b) {
    SE_CHECK(a.M == b.M && a.ntaps() == b.ntaps(), "concat_cin shape");
    DenseW d = a;
    d.Cin = a.Cin + b.Cin;
    const int nt = a.ntaps();
    d.w.assign((size_t)d.M * d.Cin * nt, 0.f);
    for (int m = 0; m < d.M; ++m) {
        std::copy(a.w.begin() + (size_t)m * a.Cin * nt, a.w.begin() + (size_t)(m + 1) * a.Cin * nt,
                  d.w.begin() + (size_t)m * d.Cin * nt);
        for (size_t i = 0; i < (size_t)b.Cin * nt; ++i)
            d.w[(size_t)m * d.Cin * nt + (size_t)a.Cin * nt + i] = scale_b * b.w[(size_t)m * b.Cin * nt + i];
    }
    return d;
}

DenseW concat_rows(const DenseW& a, const DenseW& b) {
    SE_CHECK(a.Cin == b.Cin && a.ntaps() == b.ntaps(), "concat_rows shape");
    DenseW d = a;
    d.M = a.M + b.M;
    d.w.insert(d.w.end(), b.w.begin(), b.w.end());
    d.bias.insert(d.bias.end(), b.bias.begin(), b.bias.end());
    return d;
}

DenseW interleave_rows(const DenseW& a, const DenseW& b) {
    SE_CHECK(a.M == b.M && a.Cin == b.Cin && a.ntaps() == b.ntaps(), "interleave_rows shape");
    DenseW d = a;
    d.M = 2 * a.M;
    const size_t per = (size_t)a.Cin * a.ntaps();
    d.w.resize(2 * a.w.size());
    d.bias.resize(2 * a.M);
    for (int m = 0; m < a.M; ++m) {
        std::copy(a.w.begin() + m * per, a.w.begin() + (m + 1) * per, d.w.begin() + (2 * m) * per);
        std::copy(b.w.begin() + m * per, b.w.begin() + (m + 1) * per, d.w.begin() + (2 * m + 1) * per);
        d.bias[2 * m] = a.bias[m];
        d.bias[2 * m + 1] = b.bias[m];
    }
    return d;
}

void set_post_bn(GCPlan& pl, const HostTensor& gamma, const HostTensor& beta, const HostTensor& mean, const HostTensor& var,
                 float eps) {
    const int C = (int)gamma.numel();
    std::vector<float> sc(C), sh(C);
    for (int c = 0; c < C; ++c) {
        const double s = (double)gamma.data[c] / std::sqrt((double)var.data[c] + (double)eps);
        sc[c] = (float)s;
        sh[c] = (float)((double)beta.data[c] - (double)mean.data[c] * s);
    }
    pl.dPostScale = to_device(sc);
    pl.dPostShift = to_device(sh);
    pl.p.post_scale = pl.dPostScale;
    pl.p.post_shift = pl.dPostShift;
}

std::vector<float> prelu_slopes(const HostTensor& w, int M) {
    std::vector<float> s(M);
    SE_CHECK(w.numel() == 1 || w.numel() == M, "PReLU parameter count");
    for (int m = 0; m < M; ++m) s[m] = w.numel() == 1 ? w.data[0] : w.data[m];
    return s;
}

GCPlan make_conv_plan(const DenseW& d, int sf, int pf, int pt_left, int dil_f, int dil_t, int act,
                      const std::vector<float>& slope, int epi, int tout_hint, int C0split) {
    TapSpec ts;
    ts.ntaps = d.ntaps();
    for (int kf = 0; kf < d.nkf; ++kf)
        for (int kt = 0; kt < d.nkt; ++kt) {
            ts.df[kf * d.nkt + kt] = kf * dil_f - pf;
            ts.dt[kf * d.nkt + kt] = kt * dil_t - pt_left;
        }
    return gc_make_plan(d.M, d.Cin, ts, d.w, d.bias, slope, act, epi, sf, 1, 0, tout_hint, 1, C0split);
}

DeconvPlan make_deconv_plan(const DenseW& d, int sf, int pf, int toff, int act, const std::vector<float>& slope,
                            int tout_hint, int C0split, const std::vector<float>* bias_pad, int epi) {
    SE_CHECK(pf >= 0 || bias_pad, "a left frequency pad needs the bias of the padded rows");
    DeconvPlan out;
    out.sf = sf;
    std::vector<TapSpec> cls_taps;
    std::vector<std::vector<float>> cls_w;
    for (int par = 0; par < sf; ++par) {
        TapSpec ts;
        std::vector<int> sel;
        for (int kf = 0; kf < d.nkf; ++kf) {
            int num = par + pf - kf;
            if (((num % sf) + sf) % sf != 0) continue;
            const int df = num / sf;   // exact
            for (int kt = 0; kt < d.nkt; ++kt) {
                ts.df[ts.ntaps] = df;
                ts.dt[ts.ntaps] = toff - kt;
                ts.ntaps++;
                sel.push_back(kf * d.nkt + kt);
            }
        }
        SE_CHECK(ts.ntaps > 0, "transposed-conv parity class without taps");
        std::vector<float> w((size_t)d.M * d.Cin * ts.ntaps);
        for (int m = 0; m < d.M; ++m)
            for (int c = 0; c < d.Cin; ++c)
                for (int j = 0; j < ts.ntaps; ++j)
                    w[((size_t)m * d.Cin + c) * ts.ntaps + j] = d.w[((size_t)m * d.Cin + c) * d.ntaps() + sel[j]];
        out.par.push_back(gc_make_plan(d.M, d.Cin, ts, w, d.bias, slope, act, epi, 1, sf, par, tout_hint, 1, C0split));
        cls_taps.push_back(ts);
        cls_w.push_back(w);
        if (pf < 0) {
            GCPlan& g = out.par.back();
            g.dBiasPad = to_device(*bias_pad);
            g.p.bias_pad = g.dBiasPad;
            g.p.pad_lo = -pf;
        }
    }
    static const bool pair_env = !(getenv("SE_GC_PAIR") && atoi(getenv("SE_GC_PAIR")) == 0);
    const bool pair_epi = epi == EPI_ACT || (epi == EPI_GLU && d.M == 2 && slope.empty());     // one gated output channel: rows (value, gate)
    if (pair_env && sf == 2 && d.M <= 2 && pair_epi && pf >= 0 && out.par.size() == 2 && out.par[0].p.Ws && out.par[1].p.Ws) {
        TapSpec un;
        for (int c = 0; c < 2; ++c)
            for (int j = 0; j < cls_taps[c].ntaps; ++j) {
                bool have = false;
                for (int u = 0; u < un.ntaps; ++u) have = have || (un.df[u] == cls_taps[c].df[j] && un.dt[u] == cls_taps[c].dt[j]);
                if (!have && un.ntaps < GC_MAX_TAPS) {
                    un.df[un.ntaps] = cls_taps[c].df[j];
                    un.dt[un.ntaps] = cls_taps[c].dt[j];
                    un.ntaps++;
                }
            }
        const int M2 = 2 * d.M;
        std::vector<float> w((size_t)M2 * d.Cin * un.ntaps, 0.f), bias, sl;
        for (int c = 0; c < 2; ++c)
            for (int m = 0; m < d.M; ++m)
                for (int ci = 0; ci < d.Cin; ++ci)
                    for (int j = 0; j < cls_taps[c].ntaps; ++j)
                        for (int u = 0; u < un.ntaps; ++u)
                            if (un.df[u] == cls_taps[c].df[j] && un.dt[u] == cls_taps[c].dt[j])
                                w[((size_t)(c * d.M + m) * d.Cin + ci) * un.ntaps + u] =
                                    cls_w[c][((size_t)m * d.Cin + ci) * cls_taps[c].ntaps + j];
        for (int c = 0; c < 2; ++c) {
            bias.insert(bias.end(), d.bias.begin(), d.bias.end());
            sl.insert(sl.end(), slope.begin(), slope.end());
        }
        out.pair = gc_make_plan(M2, d.Cin, un, w, bias, sl, act, epi, 1, sf, 0, tout_hint, 1, C0split);
        out.has_pair = out.pair.p.Ws != nullptr;
        if (out.has_pair) {
            out.pair.p.pair = d.M;
            out.pair.p.po2 = 1;
            out.pair.flop_scale = (double)(cls_taps[0].ntaps + cls_taps[1].ntaps) / (2.0 * un.ntaps);
        } else {
            gc_free_plan(out.pair);
        }
    }
    return out;
}

void free_deconv_plan(DeconvPlan& p) {
    for (auto& g : p.par) gc_free_plan(g);
    p.par.clear();
    if (p.has_pair) gc_free_plan(p.pair);
    p.has_pair = false;
}

// ------------------------------------------------------------------------------------------------ profiler
void Profiler::begin(hipStream_t st) {
    if (!on) return;
    if (used + 2 > ev.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            SE_HIP(hipEventCreate(&e));
            ev.push_back(e);
        }
    }
    SE_HIP(hipEventRecord(ev[used], st));
}
void Profiler::end(hipStream_t st, double fl) {
    if (!on) return;
    SE_HIP(hipEventRecord(ev[used + 1], st));
    used += 2;
    flops += fl * fscale;
    launches += 1;
    fls.push_back(fl * fscale);
}
double Profiler::total_ms() {
    double tot = 0.0;
    for (size_t i = 0; i + 1 < used; i += 2) {
        SE_HIP(hipEventSynchronize(ev[i + 1]));
        float ms = 0.f;
        SE_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        tot += ms;
        static const bool dump = getenv("SE_PROF_DUMP") != nullptr;      // per-launch lines for tools/profl.py
        if (dump && i / 2 < fls.size())
            fprintf(stderr, "PROFL %zu %.4f ms %.2f GFLOP %.1f TF/s\n", i / 2, ms, fls[i / 2] * 1e-9, fls[i / 2] / (ms * 1e9));
    }
    return tot;
}
Profiler::~Profiler() {
    for (auto e : ev) (void)hipEventDestroy(e);
}

// Frame-online chunk (kernels.h: StreamCtx): bring the history columns the taps reach back to into the sources, then
// produce only the new frames.
void gc_launch_prof(const GCPlan& pl, const GCParams& p, hipStream_t st, Profiler* prof);
static void gc_stream_prepare(StreamCtx& cx, const GCPlan& pl, GCParams& p, hipStream_t st) {
    SE_CHECK(pl.p.causal && p.Z <= 1 && p.epi != EPI_LSTM && !p.stats, "frame-online mode: layer is not a causal feed-forward conv");
    SE_CHECK(p.Tin == cx.H + cx.n && p.Tout == p.Tin && p.B == cx.B, "frame-online mode: tensor is not a window of the current chunk");
    const int need = pl.lookback;
    if (need > 0 && !(cx.memo_src == p.src0 && cx.memo_need == need)) {
        // (the parity classes of one transposed conv read the same sources: one exchange serves both launches)
        // (a concatenating layer's two sources in one launch; an exchange per consumer: remembering which tensors already have
        // their history in place for the chunk - 21 fewer launches per push of TaylorSENet_new - bought 0.5 % and was removed)
        if (p.src1 && p.C1 > 0)
            stream_exchange_pair(const_cast<float*>(p.src0), p.s0_b, p.s0_c, p.s0_f, p.C0, p.s0_f ? p.Fin : 1,
                                 const_cast<float*>(p.src1), p.s1_b, p.s1_c, p.s1_f, p.C1, p.s1_f ? p.Fin : 1, p.B, need, st);
        else
            stream_exchange(const_cast<float*>(p.src0), p.s0_b, p.s0_c, p.s0_f, p.B, p.C0, p.s0_f ? p.Fin : 1, need, st);
    }
    cx.memo_src = need > 0 ? p.src0 : nullptr;
    cx.memo_need = need;
    p.t_base = cx.H;
}
static void gc_launch_stream(StreamCtx& cx, const GCPlan& pl, GCParams p, hipStream_t st) {
    gc_stream_prepare(cx, pl, p, st);
    gc_launch(pl, p, st);
}
// the two frequency-parity classes of a transposed conv: one thin launch when both would take that path
static void gc_launch_classes(const GCPlan& pa, GCParams a, const GCPlan& pb, GCParams b, hipStream_t st, Profiler* prof) {
    if (StreamCtx* cx = stream_ctx()) {
        gc_stream_prepare(*cx, pa, a, st);
        gc_stream_prepare(*cx, pb, b, st);
        if (gc_launch_thin_pair(a, b, st)) return;
        gc_launch(pa, a, st);
        gc_launch(pb, b, st);
        return;
    }
    if (!(prof && prof->on) && gc_launch_thin_pair(a, b, st)) return;
    gc_launch_prof(pa, a, st, prof);
    gc_launch_prof(pb, b, st, prof);
}

void gc_launch_prof(const GCPlan& pl, const GCParams& p, hipStream_t st, Profiler* prof) {
    if (StreamCtx* cx = stream_ctx()) {
        gc_launch_stream(*cx, pl, p, st);
        return;
    }
    if (prof && prof->on) {
        prof->begin(st);
        gc_launch(pl, p, st);
        const double nch = (p.epi == EPI_LSTM && p.first_step) ? 0.0 : (double)(p.C0 + p.C1);
        prof->end(st, pl.flop_scale * 2.0 * p.M * nch * p.ntaps * (double)p.Z * p.B * p.Q * p.Tout);
    } else {
        gc_launch(pl, p, st);
    }
}

// ------------------------------------------------------------------------------------------------ launch helpers
static void fill_src(GCParams& p, const Act4& s0, const Act4* s1) {
    p.src0 = s0.p;
    p.C0 = s0.C;
    p.s0_b = s0.sb;
    p.s0_c = s0.sc;
    p.s0_f = s0.sf;
    p.nrm0 = s0.nrm;
    p.nrm1 = s1 ? s1->nrm : nullptr;
    if (s1) {
        p.src1 = s1->p;
        p.C1 = s1->C;
        p.s1_b = s1->sb;
        p.s1_c = s1->sc;
        p.s1_f = s1->sf;
    } else {
        p.src1 = nullptr;
        p.C1 = 0;
    }
}

static void set_stats(GCParams& p, float* stats, int dstC, int Fout, int T) {
    const long ns = (T + 31) / 32;
    p.stats = stats;
    p.st_f = ns * 2;
    p.st_c = (long)Fout * p.st_f;
    p.st_b = (long)dstC * p.st_c;
}
// GCParams::cstats: [B][Fout][T][2]
static void set_cstats(GCParams& p, float* cstats, int Fout, int T) {
    p.cstats = cstats;
    p.cs_f = 2L * T;
    p.cs_b = (long)Fout * p.cs_f;
}
bool conv_stats_supported(const GCPlan& pl) { return gc_stats_supported(pl); }
bool deconv_stats_supported(const DeconvPlan& pl) {
    for (const auto& g : pl.par)
        if (!gc_stats_supported(g)) return false;
    return true;
}

bool conv_nrm_supported(const GCPlan& pl) { return gc_nrm_supported(pl); }
bool deconv_nrm_supported(const DeconvPlan& pl) {
    if (pl.has_pair || pl.par.empty()) return false;
    for (const auto& g : pl.par)
        if (!gc_nrm_supported(g)) return false;
    return true;
}

static void set_fz(GCParams& p, float* fz, int dstC, int Fout, int Tp, int planes) {
    p.fz = fz;
    p.fz_im = (long)dstC * Fout * Tp;
    p.fz_b = planes * p.fz_im;
    p.fz_c = (long)Fout * Tp;
    p.fz_f = Tp;
    // three planes [S | R | I] (fz = the R plane): the epilogue that rewrites R and I stores S = R + I as well - no gauss_sum pass
    static const bool fzs = !(getenv("SE_UF_FOLD_SUM") && atoi(getenv("SE_UF_FOLD_SUM")) == 0);
    p.fz_s = (planes == 3 && fzs) ? -p.fz_im : 0;
}
bool conv_fold_writes_sum() {
    static const bool fzs = !(getenv("SE_UF_FOLD_SUM") && atoi(getenv("SE_UF_FOLD_SUM")) == 0);
    return fzs;
}

void run_conv(const GCPlan& pl, const Act4& s0, const Act4* s1, float* dst, int dstC, int Fout, int B, int T, int Tp,
              hipStream_t st, Profiler* prof, float* stats, int t_base, float* fz, int fz_planes, bool colstats, float* dst_elu) {
    GCParams p = pl.p;
    p.t_base = t_base;
    p.dst_elu = dst_elu;
    if (fz) set_fz(p, fz, dstC, Fout, Tp, fz_planes);
    if (stats && colstats) set_cstats(p, stats, Fout, T);
    else if (stats) set_stats(p, stats, dstC, Fout, T);
    fill_src(p, s0, s1);
    p.Fin = s0.F;
    p.Tin = T;
    p.B = B;
    p.Q = Fout;
    p.Tout = T;
    p.dst = dst;
    p.d_b = (long)dstC * Fout * Tp;
    p.d_c = (long)Fout * Tp;
    p.d_f = Tp;
    if (const Ragged* rg = ragged_ctx()) p.tlen = rg->tlen;       // MFMA path: tails of shorter rows leave as zeros
    gc_launch_prof(pl, p, st, prof);
}

void run_deconv(const DeconvPlan& pl, const Act4& s0, const Act4* s1, float* dst, int dstC, int Fout, int B, int T,
                int Tp, hipStream_t st, Profiler* prof, float* stats, int t_base, int t_out, bool tb_soft, float* fz, int fz_planes,
                bool colstats) {
    if (t_out < 0) t_out = T;
    SE_CHECK(!fz || conv_folds_interaction(pl), "run_deconv: this plan cannot fold the branch interaction into its store");
    // (a BatchNorm attached to the class plans later is not in the pair; the few frames of a frame-online chunk go through
    // the thin kernel, one launch per class)
    const StreamCtx* cx = stream_ctx();
    const bool few = (cx ? cx->n : t_out - t_base) <= GC_THIN_NT;
    if (pl.has_pair && !stats && !pl.par[0].p.post_scale && !few) {
        GCParams p = pl.pair.p;
        fill_src(p, s0, s1);
        p.Fin = s0.F;
        p.Tin = T;
        p.B = B;
        p.Q = (Fout + p.so - 1) / p.so;
        p.fo_lim = Fout;
        p.Tout = t_out;
        p.t_base = t_base;
        p.tb_soft = tb_soft;
        p.dst = dst;
        p.d_b = (long)dstC * Fout * Tp;
        p.d_c = (long)Fout * Tp;
        p.d_f = Tp;
        gc_launch_prof(pl.pair, p, st, prof);
        return;
    }
    GCParams ps[2];
    int np = 0;
    const GCPlan* gp[2] = {nullptr, nullptr};
    for (const auto& g : pl.par) {
        GCParams p = g.p;
        if (stats && colstats) set_cstats(p, stats, Fout, T);
        else if (stats) set_stats(p, stats, dstC, Fout, T);
        fill_src(p, s0, s1);
        p.Fin = s0.F;
        p.Tin = T;
        p.B = B;
        p.Q = (Fout - p.po + p.so - 1) / p.so;
        p.Tout = t_out;
        p.t_base = t_base;
        p.tb_soft = tb_soft;
        p.dst = dst;
        p.d_b = (long)dstC * Fout * Tp;
        p.d_c = (long)Fout * Tp;
        p.d_f = Tp;
        if (const Ragged* rg = ragged_ctx()) p.tlen = rg->tlen;
        if (fz) set_fz(p, fz, dstC, Fout, Tp, fz_planes);
        if (p.Q <= 0) continue;
        if (few && !stats && pl.par.size() == 2 && np < 2) {      // (a few frames: both classes in one thin launch if they qualify)
            ps[np] = p;
            gp[np++] = &g;
            continue;
        }
        gc_launch_prof(g, p, st, prof);
    }
    if (np == 2) gc_launch_classes(*gp[0], ps[0], *gp[1], ps[1], st, prof);
    else if (np == 1) gc_launch_prof(*gp[0], ps[0], st, prof);
}

}  // namespace se
