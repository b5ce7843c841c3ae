// dccrn_cpu.cpp - C++ / OpenMP restatement of the DCCRN decode loop for the host CPU.
//
// TEST INFRASTRUCTURE / CPU BASELINE (checker side).  Nothing on the product path loads this library: only tests/
// (tests/test_dccrn_cpu.py pins it to the same reference-generated fixtures as the numpy oracle) and bench.py's
// `cpu_baseline` leg, which times it on the GPU box's host cores at 1 thread and at all physical cores (SURVEY.md 8(d):
// "the on-box CPU comparator is the build's C++ CPU restatement").
//
// What it restates (paths into the reference, cszheng-ioa/Sixty-years-of-frequency-domain-monaural-speech-enhancement):
//   DCCRN/dccrn_decode_vb.py:25-62   the loop body: c = sqrt(L / sum x^2), tail pad to a hop multiple (:32-35),
//                                    torch.stft(512, 128, 512, hann) (:37-38), |X|^p e^{j angle X} (:40-42), model (:44),
//                                    |S|^p' e^{j angle S} (:45-58), librosa.istft(length = padded length) (:59-60), / c (:62)
//   DCCRN/DCCRN_cprs.py:142-226      DCCRN.forward for the decode script's constructor (dccrn_decode_vb.py:11): six complex
//                                    conv + BatchNorm + PReLU encoder layers (:66-75, :170-173), two complex LSTM layers
//                                    (:82-92, :175-185), six complex transposed-conv decoder layers with complex_cat skips
//                                    and drop-first-frame (:108-134, :196-199), the 'E' mask (:201-225)
//   complexnn.{ComplexConv2d, ComplexConvTranspose2d, NavieComplexLSTM, complex_cat}: ABSENT from the reference (imported at
//                                    DCCRN_cprs.py:6) - follows oracle/_complexnn_recall.py, the same restatement the numpy
//                                    oracle and the engine follow: parity unpinned AT that boundary, pinned above it.
//
// Arithmetic: fp32 network (the reference feeds torch.FloatTensor), float64 STFT / iSTFT like oracle/stft.py.  A complex
// (de)conv is evaluated as one real conv over the 2 x 2 block matrix [[Wr, -Wi], [Wi, Wr]] - the same four real
// multiply-adds per complex one that the reference's two nn.Conv2d applied to both halves perform.
//
// Layout: activations [C][F][row] with frames contiguous; a row holds LEAD zero floats, T frames, then zeros up to the pitch,
// so taps that look one frame back / ahead read zeros without a branch.  The conv micro-kernel keeps an 8-channel x 32-frame
// tile of accumulators in vector registers (GCC vector extensions, cloned for AVX-512 / AVX2 / baseline x86-64).
//
// Build: g++ -O3 -fopenmp -shared -fPIC (oracle/Makefile).  C ABI at the bottom (ctypes: oracle/dccrn_cpu.py).
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include <omp.h>

namespace {

typedef float vf __attribute__((vector_size(64), aligned(4)));   // 16 floats, unaligned access allowed
constexpr int VL = 16, TB = 2 * VL, CB = 8, LEAD = 16;
constexpr int NFFT = 512, HOP = 128, NBIN = 257;

inline int pitch_of(int T) { return LEAD + (T + TB - 1) / TB * TB + TB; }

struct Tensor {                 // [C][F][pitch]
    int C = 0, F = 0, T = 0, P = 0;
    std::vector<float> d;
    void shape(int c, int f, int t) {
        C = c; F = f; T = t; P = pitch_of(t);
        d.assign((size_t)c * f * P, 0.f);
    }
    float* row(int c, int f) { return d.data() + ((size_t)c * F + f) * P + LEAD; }
    const float* row(int c, int f) const { return d.data() + ((size_t)c * F + f) * P + LEAD; }
};

// ---------------------------------------------------------------------------------------------- tap convolution
// out[co][fo][t] = post(bias[co] + sum_taps sum_ci W[tap][ci][co] * in[ci][fi(tap, fo)][t + dt(tap)])
struct TapW {
    int dt;                       // frame offset of the tap
    std::vector<float> w;         // [Cin][CoP], output channel contiguous, zero padded to a multiple of CB
};
struct Layer {
    int Cin = 0, Cout = 0, CoP = 0;
    std::vector<float> scale, shift;   // per output channel: y = (acc + bias) * scale + shift  (conv bias and eval BatchNorm)
    float slope = 1.f;                 // scalar PReLU (1 = identity)
    // frequency geometry: conv: fi = 2 fo + kf - 2; transposed conv: fi = (fo + 2 - kf) / 2 where that is an integer
    bool transposed = false;
    std::vector<TapW> taps;            // index kf * 2 + kt
};

struct RowTap { const float* w; const float* in; long in_cstride; };     // one (kf, kt) tap resolved for an output row

__attribute__((target_clones("avx512f", "avx2", "default")))
void conv_tile(const RowTap* rt, int ntap, int Cin, int CoP, int co0, int t0, float* const* orow, const float* scale,
               const float* shift, float slope, int ncv) {
    vf acc[CB][2];
    for (int c = 0; c < CB; ++c) acc[c][0] = acc[c][1] = vf{};
    for (int j = 0; j < ntap; ++j) {
        const float* w = rt[j].w + co0;
        const float* x = rt[j].in + t0;
        const long cs = rt[j].in_cstride;
        for (int ci = 0; ci < Cin; ++ci) {
            vf x0, x1;
            __builtin_memcpy(&x0, x, sizeof(vf));
            __builtin_memcpy(&x1, x + VL, sizeof(vf));
#pragma GCC unroll 8
            for (int c = 0; c < CB; ++c) {
                const float wv = w[c];
                acc[c][0] += wv * x0;
                acc[c][1] += wv * x1;
            }
            w += CoP;
            x += cs;
        }
    }
    for (int c = 0; c < ncv; ++c) {
        const float s = scale[co0 + c], h = shift[co0 + c];
        for (int v = 0; v < 2; ++v) {
            vf y = acc[c][v] * s + h;
            vf neg = y * slope;
            y = y > 0.f ? y : neg;                        // PReLU with a scalar slope (slope 1: identity)
            __builtin_memcpy(orow[c] + t0 + v * VL, &y, sizeof(vf));
        }
    }
}

void run_layer(const Layer& L, const Tensor& in, Tensor& out, int Fout, int threads, int drop_first) {
    // drop_first: a transposed conv emits T + 1 frames and the reference keeps frames 1..T (DCCRN_cprs.py:199): frame t of
    // the kept tensor reads input frame t + 1 - kt
    const int T = in.T;
    out.shape(L.Cout, Fout, T);
    const int ncb = L.CoP / CB, ntt = (T + TB - 1) / TB;
    const long tasks = (long)Fout * ntt;
    // one task = one output row x one 32-frame tile, all output channels: the input patch of the tile (<= 5 rows x Cin x 32
    // frames) stays in the core's L2 while the layer's weights stream through it from the shared L3
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads) if (threads > 1)
    for (long task = 0; task < tasks; ++task) {
        const int fo = (int)(task / ntt), tt = (int)(task % ntt);
        RowTap rt[10];
        int n = 0;
        for (int kf = 0; kf < 5; ++kf) {
            int fi;
            if (!L.transposed) {
                fi = 2 * fo + kf - 2;
            } else {
                const int num = fo + 2 - kf;
                if (num & 1) continue;
                fi = num / 2;
            }
            if (fi < 0 || fi >= in.F) continue;
            for (int kt = 0; kt < 2; ++kt) {
                const TapW& tw = L.taps[kf * 2 + kt];
                rt[n].w = tw.w.data();
                rt[n].in = in.row(0, fi) + tw.dt + (L.transposed ? drop_first : 0);
                rt[n].in_cstride = (long)in.F * in.P;
                ++n;
            }
        }
        for (int cb = 0; cb < ncb; ++cb) {
            float* orow[CB];
            const int co0 = cb * CB, ncv = std::min(CB, L.Cout - co0);
            for (int c = 0; c < ncv; ++c) orow[c] = out.row(co0 + c, fo);
            conv_tile(rt, n, L.Cin, L.CoP, co0, tt * TB, orow, L.scale.data(), L.shift.data(), L.slope, ncv);
            if (tt == ntt - 1)                               // frames past T must read as zeros for the next layer's look-ahead
                for (int c = 0; c < ncv; ++c) std::fill(orow[c] + T, orow[c] + (out.P - LEAD), 0.f);
        }
    }
}

// ---------------------------------------------------------------------------------------------- weights
typedef std::map<std::string, std::vector<float>> SD;
const std::vector<float>& get(const SD& sd, const std::string& k, size_t n) {
    auto it = sd.find(k);
    if (it == sd.end()) throw std::runtime_error("missing key " + k);
    if (it->second.size() != n) throw std::runtime_error("wrong size for " + k);
    return it->second;
}

// complex (de)conv -> real block layer.  conv weights [Co][Ci][5][2]; transposed conv weights [Ci][Co][5][2].
Layer make_cplx(const SD& sd, const std::string& p, int Ci, int Co, bool transposed, const std::string& bn, const std::string& act) {
    const auto& wr = get(sd, p + "real_conv.weight", (size_t)Co * Ci * 10);
    const auto& wi = get(sd, p + "imag_conv.weight", (size_t)Co * Ci * 10);
    const auto& br = get(sd, p + "real_conv.bias", Co);
    const auto& bi = get(sd, p + "imag_conv.bias", Co);
    Layer L;
    L.Cin = 2 * Ci;
    L.Cout = 2 * Co;
    L.CoP = (L.Cout + CB - 1) / CB * CB;
    L.transposed = transposed;
    L.taps.resize(10);
    for (int kf = 0; kf < 5; ++kf)
        for (int kt = 0; kt < 2; ++kt) {
            TapW& t = L.taps[kf * 2 + kt];
            // conv: time padded by one frame on the LEFT (causal), kernel 2: tap kt reads frame t + kt - 1
            // transposed conv (gather form, after dropping frame 0): tap kt reads frame t + 1 - kt  -> dt = -kt (+1 at launch)
            t.dt = transposed ? -kt : kt - 1;
            t.w.assign((size_t)L.Cin * L.CoP, 0.f);
            for (int ci = 0; ci < Ci; ++ci)
                for (int co = 0; co < Co; ++co) {
                    const size_t idx = transposed ? (((size_t)ci * Co + co) * 5 + kf) * 2 + kt : (((size_t)co * Ci + ci) * 5 + kf) * 2 + kt;
                    const float r = wr[idx], i = wi[idx];
                    // real out = Wr x_r - Wi x_i ; imag out = Wi x_r + Wr x_i   (_complexnn_recall.py ComplexConv2d.forward)
                    t.w[(size_t)ci * L.CoP + co] = r;
                    t.w[(size_t)(Ci + ci) * L.CoP + co] = -i;
                    t.w[(size_t)ci * L.CoP + Co + co] = i;
                    t.w[(size_t)(Ci + ci) * L.CoP + Co + co] = r;
                }
        }
    L.scale.assign(L.CoP, 1.f);
    L.shift.assign(L.CoP, 0.f);
    for (int co = 0; co < Co; ++co) {          // biases of the two real convs combine with the signs of their outputs
        L.shift[co] = br[co] - bi[co];
        L.shift[Co + co] = bi[co] + br[co];
    }
    if (!bn.empty()) {                          // eval-mode BatchNorm2d over the 2 Co channels (use_cbn = False)
        const auto& g = get(sd, bn + "weight", L.Cout);
        const auto& b = get(sd, bn + "bias", L.Cout);
        const auto& m = get(sd, bn + "running_mean", L.Cout);
        const auto& v = get(sd, bn + "running_var", L.Cout);
        for (int c = 0; c < L.Cout; ++c) {
            const float s = g[c] / std::sqrt(v[c] + 1e-5f);
            L.scale[c] = s;
            L.shift[c] = (L.shift[c] - m[c]) * s + b[c];
        }
    }
    L.slope = act.empty() ? 1.f : get(sd, act + "weight", 1)[0];
    return L;
}

// Linear / LSTM input projection as a one-tap layer over [I][1][T]
Layer make_linear(const std::vector<float>& W, const std::vector<float>& b, int out, int in) {
    Layer L;
    L.Cin = in;
    L.Cout = out;
    L.CoP = (out + CB - 1) / CB * CB;
    L.taps.resize(1);
    L.taps[0].dt = 0;
    L.taps[0].w.assign((size_t)in * L.CoP, 0.f);
    for (int o = 0; o < out; ++o)
        for (int i = 0; i < in; ++i) L.taps[0].w[(size_t)i * L.CoP + o] = W[(size_t)o * in + i];
    L.scale.assign(L.CoP, 1.f);
    L.shift.assign(L.CoP, 0.f);
    for (int o = 0; o < out; ++o) L.shift[o] = b[o];
    return L;
}
void run_linear(const Layer& L, const Tensor& in, Tensor& out, int threads) {
    const int T = in.T;
    out.shape(L.Cout, 1, T);
    const int ncb = L.CoP / CB, ntt = (T + TB - 1) / TB;
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
    for (int cb = 0; cb < ncb; ++cb) {
        RowTap rt{L.taps[0].w.data(), in.row(0, 0), (long)in.P};
        float* orow[CB];
        const int co0 = cb * CB, ncv = std::min(CB, L.Cout - co0);
        for (int c = 0; c < ncv; ++c) orow[c] = out.row(co0 + c, 0);
        for (int tt = 0; tt < ntt; ++tt) conv_tile(&rt, 1, L.Cin, L.CoP, co0, tt * TB, orow, L.scale.data(), L.shift.data(), 1.f, ncv);
        for (int c = 0; c < ncv; ++c) std::fill(orow[c] + T, orow[c] + (out.P - LEAD), 0.f);
    }
}

struct Lstm {                      // one torch.nn.LSTM(I, H) layer, gate order i, f, g, o, zero initial state
    int I = 0, H = 0;
    Layer gin;                     // x -> W_ih x + b_ih + b_hh over all frames
    std::vector<float> whhT;       // [H][4H]
    void load(const SD& sd, const std::string& p, int i, int h) {
        I = i; H = h;
        const auto& wih = get(sd, p + "weight_ih_l0", (size_t)4 * h * i);
        const auto& whh = get(sd, p + "weight_hh_l0", (size_t)4 * h * h);
        const auto& bi = get(sd, p + "bias_ih_l0", 4 * h);
        const auto& bh = get(sd, p + "bias_hh_l0", 4 * h);
        std::vector<float> b(4 * h);
        for (int k = 0; k < 4 * h; ++k) b[k] = bi[k] + bh[k];
        gin = make_linear(wih, b, 4 * h, i);
        whhT.assign((size_t)h * 4 * h, 0.f);
        for (int r = 0; r < 4 * h; ++r)
            for (int k = 0; k < h; ++k) whhT[(size_t)k * 4 * h + r] = whh[(size_t)r * h + k];
    }
    // x [I][1][T] -> y [H][1][T]
    void run(const Tensor& x, Tensor& y, Tensor& gx, int threads) const {
        run_linear(gin, x, gx, threads);
        const int T = x.T, G = 4 * H;
        y.shape(H, 1, T);
        std::vector<float> h(H, 0.f), c(H, 0.f), g(G);
        for (int t = 0; t < T; ++t) {
            for (int r = 0; r < G; ++r) g[r] = gx.row(r, 0)[t];
            for (int k = 0; k < H; ++k) {
                const float hk = h[k];
                const float* w = whhT.data() + (size_t)k * G;
                for (int r = 0; r < G; ++r) g[r] += w[r] * hk;
            }
            for (int u = 0; u < H; ++u) {
                const float ig = 1.f / (1.f + std::exp(-g[u])), fg = 1.f / (1.f + std::exp(-g[H + u]));
                const float gg = std::tanh(g[2 * H + u]), og = 1.f / (1.f + std::exp(-g[3 * H + u]));
                c[u] = fg * c[u] + ig * gg;
                h[u] = og * std::tanh(c[u]);
                y.row(u, 0)[t] = h[u];
            }
        }
    }
};

struct Model {
    SD sd;
    bool ready = false;
    Layer enc[6], dec[6];
    Lstm l0r, l0i, l1r, l1i;
    Layer rtrans, itrans;
    void finalize() {
        const int kn[7] = {2, 32, 64, 128, 256, 256, 256};                           // DCCRN_cprs.py:47 with the script's list
        for (int k = 0; k < 6; ++k) {
            const std::string p = "encoder." + std::to_string(k) + ".";
            enc[k] = make_cplx(sd, p + "0.", kn[k] / 2, kn[k + 1] / 2, false, p + "1.", p + "2.");
        }
        for (int k = 0; k < 6; ++k) {
            const std::string p = "decoder." + std::to_string(k) + ".";
            const int ci = kn[6 - k], co = kn[5 - k];                                  // input = cat(out, skip): 2 * ci channels
            dec[k] = make_cplx(sd, p + "0.", ci, co / 2, true, k < 5 ? p + "1." : "", k < 5 ? p + "2." : "");
        }
        l0r.load(sd, "enhance.0.real_lstm.", 512, 128);
        l0i.load(sd, "enhance.0.imag_lstm.", 512, 128);
        l1r.load(sd, "enhance.1.real_lstm.", 128, 128);
        l1i.load(sd, "enhance.1.imag_lstm.", 128, 128);
        rtrans = make_linear(get(sd, "enhance.1.r_trans.weight", 512 * 128), get(sd, "enhance.1.r_trans.bias", 512), 512, 128);
        itrans = make_linear(get(sd, "enhance.1.i_trans.weight", 512 * 128), get(sd, "enhance.1.i_trans.bias", 512), 512, 128);
        ready = true;
    }
};

struct Work {                       // per-thread buffers, reused from clip to clip (no allocation after the first clip)
    Tensor x0, e[6], d, tmp, cat, r, i, a, b, c2, d2, gx, ro, io, r2, i2, pr, pi;
    std::vector<float> x, feat, est;
    std::vector<double> y, env, frames;
};

// NavieComplexLSTM.forward([r, i]): real = real_lstm(r) - imag_lstm(i); imag = real_lstm(i) + imag_lstm(r)
void complex_lstm(const Lstm& lr, const Lstm& li, const Tensor& r, const Tensor& i, Tensor& ro, Tensor& io, Work& w, int threads) {
    lr.run(r, w.a, w.gx, threads);      // r2r
    li.run(i, w.b, w.gx, threads);      // i2i
    lr.run(i, w.c2, w.gx, threads);     // i2r
    li.run(r, w.d2, w.gx, threads);     // r2i
    const int H = lr.H, T = r.T;
    ro.shape(H, 1, T);
    io.shape(H, 1, T);
    for (int u = 0; u < H; ++u)
        for (int t = 0; t < T; ++t) {
            ro.row(u, 0)[t] = w.a.row(u, 0)[t] - w.b.row(u, 0)[t];
            io.row(u, 0)[t] = w.c2.row(u, 0)[t] + w.d2.row(u, 0)[t];
        }
}

// DCCRN.forward: in [2][257][T] (dense, T contiguous) -> out [2][257][T]
void forward(const Model& m, const float* in, int T, float* out, Work& w, int threads) {
    const size_t plane = (size_t)NBIN * T;
    w.x0.shape(2, 256, T);                                                             // drop the DC bin, :166
    for (int c = 0; c < 2; ++c)
        for (int f = 0; f < 256; ++f) std::memcpy(w.x0.row(c, f), in + c * plane + (size_t)(f + 1) * T, sizeof(float) * T);
    const Tensor* cur = &w.x0;
    int F = 256;
    for (int k = 0; k < 6; ++k) {                                                      // :170-173
        F /= 2;
        run_layer(m.enc[k], *cur, w.e[k], F, threads, 0);
        cur = &w.e[k];
    }
    // :175-185  [C=256][D=4][T] -> r, i = [T][C/2 * D] each (feature index c * D + d)
    const int Ch = 128, D = 4;
    w.r.shape(Ch * D, 1, T);
    w.i.shape(Ch * D, 1, T);
    for (int c = 0; c < Ch; ++c)
        for (int d = 0; d < D; ++d) {
            std::memcpy(w.r.row(c * D + d, 0), cur->row(c, d), sizeof(float) * T);
            std::memcpy(w.i.row(c * D + d, 0), cur->row(Ch + c, d), sizeof(float) * T);
        }
    complex_lstm(m.l0r, m.l0i, w.r, w.i, w.ro, w.io, w, threads);
    complex_lstm(m.l1r, m.l1i, w.ro, w.io, w.r2, w.i2, w, threads);
    run_linear(m.rtrans, w.r2, w.pr, threads);                                         // projection_dim on the last layer
    run_linear(m.itrans, w.i2, w.pi, threads);
    w.d.shape(2 * Ch, D, T);
    for (int c = 0; c < Ch; ++c)
        for (int d = 0; d < D; ++d) {
            std::memcpy(w.d.row(c, d), w.pr.row(c * D + d, 0), sizeof(float) * T);
            std::memcpy(w.d.row(Ch + c, d), w.pi.row(c * D + d, 0), sizeof(float) * T);
        }
    // :196-199 decoder: complex_cat([out, skip]) = [out_r, skip_r, out_i, skip_i]
    Tensor* dcur = &w.d;
    Tensor& tmp = w.tmp;
    F = 4;
    for (int k = 0; k < 6; ++k) {
        const Tensor& sk = w.e[5 - k];
        const int hc = dcur->C / 2;
        w.cat.shape(4 * hc, F, T);
        for (int c = 0; c < hc; ++c)
            for (int f = 0; f < F; ++f) {
                std::memcpy(w.cat.row(c, f), dcur->row(c, f), sizeof(float) * T);
                std::memcpy(w.cat.row(hc + c, f), sk.row(c, f), sizeof(float) * T);
                std::memcpy(w.cat.row(2 * hc + c, f), dcur->row(hc + c, f), sizeof(float) * T);
                std::memcpy(w.cat.row(3 * hc + c, f), sk.row(hc + c, f), sizeof(float) * T);
            }
        run_layer(m.dec[k], w.cat, tmp, 2 * F, threads, 1);
        std::swap(w.d, tmp);
        dcur = &w.d;
        F *= 2;
    }
    // :201-225 'E' mask; DC row of the mask is zero
    for (int f = 0; f < NBIN; ++f)
        for (int t = 0; t < T; ++t) {
            const float xr = in[(size_t)f * T + t], xi = in[plane + (size_t)f * T + t];
            const float mr = f ? dcur->row(0, f - 1)[t] : 0.f, mi = f ? dcur->row(1, f - 1)[t] : 0.f;
            const float smag = std::sqrt(xr * xr + xi * xi), sph = std::atan2(xi, xr);
            const float mm = std::sqrt(mr * mr + mi * mi);
            const float rp = mr / (mm + 1e-8f), ip = mi / (mm + 1e-8f);
            const float mph = std::atan2(ip, rp);
            const float em = std::tanh(mm) * smag, ep = sph + mph;
            out[(size_t)f * T + t] = em * std::cos(ep);
            out[plane + (size_t)f * T + t] = em * std::sin(ep);
        }
}

// ---------------------------------------------------------------------------------------------- STFT / iSTFT (float64)
struct Fft {
    std::vector<std::complex<double>> tw;
    std::vector<int> rev;
    std::vector<double> win;
    Fft() {
        tw.resize(NFFT / 2);
        for (int k = 0; k < NFFT / 2; ++k) tw[k] = std::polar(1.0, -2.0 * M_PI * k / NFFT);
        rev.resize(NFFT);
        for (int i = 0; i < NFFT; ++i) {
            int r = 0;
            for (int b = 0; b < 9; ++b) r |= ((i >> b) & 1) << (8 - b);
            rev[i] = r;
        }
        win.resize(NFFT);
        for (int n = 0; n < NFFT; ++n) win[n] = 0.5 - 0.5 * std::cos(2.0 * M_PI * n / NFFT);     // periodic Hann
    }
    void run(std::complex<double>* a, bool inverse) const {
        for (int i = 0; i < NFFT; ++i)
            if (rev[i] > i) std::swap(a[i], a[rev[i]]);
        for (int len = 2; len <= NFFT; len <<= 1) {
            const int step = NFFT / len;
            for (int s = 0; s < NFFT; s += len)
                for (int k = 0; k < len / 2; ++k) {
                    std::complex<double> w = tw[k * step];
                    if (inverse) w = std::conj(w);
                    const std::complex<double> u = a[s + k], v = a[s + k + len / 2] * w;
                    a[s + k] = u + v;
                    a[s + k + len / 2] = u - v;
                }
        }
    }
};
const Fft& fft() {
    static Fft f;
    return f;
}

long padded_len(long n) {                                   // dccrn_decode_vb.py:32-35
    const long frame_num = (long)std::ceil((double)n / HOP + 1.0);
    return (frame_num - 1) * HOP;
}

// the loop body of enhance(args) for one clip; out receives padded_len(n) samples
void enhance_one(const Model& m, const float* wav, long n, float p_in, float p_out, float* out, Work& w, int threads) {
    double ss = 0.0;
    for (long k = 0; k < n; ++k) ss += (double)wav[k] * wav[k];
    const double c = std::sqrt((double)n / ss);                                        // :27
    const long L = padded_len(n);
    std::vector<float>& x = w.x;
    x.assign(L, 0.f);
    for (long k = 0; k < n; ++k) x[k] = (float)((double)wav[k] * c);                  // :28, FloatTensor :36
    const int T = 1 + (int)(L / HOP), pad = NFFT / 2;
    std::vector<float>&feat = w.feat, &est = w.est;
    feat.resize((size_t)2 * NBIN * T);
    est.resize((size_t)2 * NBIN * T);
    const Fft& F = fft();
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
    for (int t = 0; t < T; ++t) {                                                      // torch.stft, centre = True, reflect pad
        std::complex<double> a[NFFT];
        for (int k = 0; k < NFFT; ++k) {
            long s = (long)t * HOP + k - pad;
            if (s < 0) s = -s;
            if (s >= L) s = 2 * (L - 1) - s;
            a[k] = (double)x[s] * F.win[k];
        }
        F.run(a, false);
        for (int f = 0; f < NBIN; ++f) {
            const float re = (float)a[f].real(), im = (float)a[f].imag();
            float mag = std::sqrt(re * re + im * im);                                  // :40
            if (p_in != 1.f) mag = std::pow(mag, p_in);
            const float ph = std::atan2(im, re);
            feat[(size_t)f * T + t] = mag * std::cos(ph);                               // :42
            feat[(size_t)(NBIN + f) * T + t] = mag * std::sin(ph);
        }
    }
    forward(m, feat.data(), T, est.data(), w, threads);                                // :44
    const long full = NFFT + (long)HOP * (T - 1);
    std::vector<double>&y = w.y, &env = w.env, &frames = w.frames;
    y.assign(full, 0.0);
    env.assign(full, 0.0);
    frames.resize((size_t)T * NFFT);
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
    for (int t = 0; t < T; ++t) {                                                      // :45-58 + irfft per frame
        std::complex<double> a[NFFT];
        for (int f = 0; f < NBIN; ++f) {
            const float er = est[(size_t)f * T + t], ei = est[(size_t)(NBIN + f) * T + t];
            float mag = std::sqrt(er * er + ei * ei);
            if (p_out != 1.f) mag = std::pow(mag, p_out);
            const double ph = (double)std::atan2(ei, er);
            a[f] = std::polar((double)mag, ph);
        }
        a[0] = a[0].real();                                                             // irfft ignores the imaginary parts of DC / Nyquist
        a[NFFT / 2] = a[NFFT / 2].real();
        for (int f = 1; f < NFFT / 2; ++f) a[NFFT - f] = std::conj(a[f]);
        F.run(a, true);
        for (int k = 0; k < NFFT; ++k) frames[(size_t)t * NFFT + k] = a[k].real() / NFFT * F.win[k];
    }
    for (int t = 0; t < T; ++t)
        for (int k = 0; k < NFFT; ++k) {
            y[(long)t * HOP + k] += frames[(size_t)t * NFFT + k];
            env[(long)t * HOP + k] += F.win[k] * F.win[k];
        }
    for (long k = 0; k < L; ++k) {                                                     // :59-62 length = padded length, / c
        const double e = env[pad + k];
        const double v = e > 1e-11 ? y[pad + k] / e : y[pad + k];
        out[k] = (float)(v / c);
    }
}

thread_local std::string g_err;

}  // namespace

extern "C" {

void* dccrn_cpu_create() { return new Model(); }
void dccrn_cpu_destroy(void* h) { delete static_cast<Model*>(h); }
const char* dccrn_cpu_last_error() { return g_err.c_str(); }

// one call per float32 state-dict entry (the int64 num_batches_tracked counters are not passed)
int dccrn_cpu_set(void* h, const char* key, const float* data, long n) {
    static_cast<Model*>(h)->sd[key] = std::vector<float>(data, data + n);
    return 0;
}
int dccrn_cpu_finalize(void* h) {
    try {
        static_cast<Model*>(h)->finalize();
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return 1;
    }
}
long dccrn_cpu_output_samples(long n) { return padded_len(n); }

// y = model(x): x, y [2][257][T] dense
int dccrn_cpu_forward(void* h, const float* in, int T, float* out, int threads) {
    const Model& m = *static_cast<Model*>(h);
    if (!m.ready) return 1;
    Work w;
    forward(m, in, T, out, w, std::max(1, threads));
    return 0;
}

// B clips of n samples each.  mode 0: the reference's batch-1 loop, one clip after the other, `threads` OpenMP threads
// inside each layer; mode 1: `threads` clips in flight, one thread each (utterance-parallel, how a host would be filled).
int dccrn_cpu_enhance(void* h, const float* wav, long pitch, int B, long n, float p_in, float p_out, float* out, long out_pitch,
                      int threads, int mode) {
    const Model& m = *static_cast<Model*>(h);
    if (!m.ready || n < NFFT) return 1;
    threads = std::max(1, threads);
    if (mode == 0) {
        Work w;
        for (int b = 0; b < B; ++b) enhance_one(m, wav + (size_t)b * pitch, n, p_in, p_out, out + (size_t)b * out_pitch, w, threads);
        return 0;
    }
#pragma omp parallel num_threads(threads)
    {
        Work w;
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; ++b) enhance_one(m, wav + (size_t)b * pitch, n, p_in, p_out, out + (size_t)b * out_pitch, w, 1);
    }
    return 0;
}

int dccrn_cpu_max_threads() { return omp_get_max_threads(); }

}  // extern "C"
