// Gauss' three-product complex (de)conv: the kernels and the per-layer record shared by the models that run their wide complex
// layers this way (model_dccrn.hip, model_uformer.hip).  DESIGN.md 3.6.
#pragma once
#include "common.h"
#include "gemmconv.h"
#include "layers.h"
#include <vector>

namespace se {
namespace gauss {

// ---- Gauss' three-product complex (de)conv (measure it) -------------------------------------------------
// The reference computes a complex conv as four real ones (complexnn: r2r - i2i, r2i + i2r); the engine runs them as ONE real
// conv over the 2 x 2 block matrix.  Gauss: k1 = Wr (xr + xi), k2 = (Wi - Wr) xr, k3 = (Wr + Wi) xi, yr = k1 - k3, yi = k1 + k2 -
// three real convs of half the rows and half the K: 3/4 of the matrix instructions.  Here as a GROUPED launch of the same
// gc_kernel (blockIdx.z = product, sources = the planes [xr + xi | xr | xi] of a three-plane tensor, outputs k1..k3 in a scratch
// tensor) and one elementwise pass that combines them, applies BatchNorm / bias / PReLU and writes the next layer's three planes.
// tools/gcbench.cpp `gauss` (profiles/r04_gauss_gcbench.log): 128 -> 128 complex channels 0.81x the block GEMM's time, 64 -> 128
// 0.87x, 32 -> 64 0.97x, 16 -> 32 1.28x (the combine pass is as big as the GEMM there) - so the layers with >= 128 complex output
// channels take this path: encoder 3 - 5, decoder 0 - 1 (50 % of the step).  Rounding: the products are formed on sums of
// weights / inputs - 4e-7 ... 1e-6 relative per layer against the four-product form (bar 1e-4 on the waveform).
// x3 [B][3 C][P] planes (S | R | I): S = R + I
static __global__ __launch_bounds__(256) void gauss_sum_kernel(float* __restrict__ x3, long CP) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4, b = blockIdx.y;
    if (i >= CP) return;
    float* xb = x3 + b * 3 * CP;
    if (i + 3 < CP && (CP & 3) == 0) {
        const float4 r = *reinterpret_cast<const float4*>(xb + CP + i), m = *reinterpret_cast<const float4*>(xb + 2 * CP + i);
        *reinterpret_cast<float4*>(xb + i) = make_float4(r.x + m.x, r.y + m.y, r.z + m.z, r.w + m.w);
    } else {
        for (long j = i; j < CP && j < i + 4; ++j) xb[j] = xb[CP + j] + xb[2 * CP + j];
    }
}
// k [3][B][Co][F][T] -> y: yr = act((k1 - k3) sc[c] + sh[c]), yi = act((k1 + k2) sc[Co + c] + sh[Co + c]); planes of batch item b at
// y + b ob + {oS, oR, oI} (oS < 0: no sum plane); frames >= tlen[b] are stored as zeros (ragged rows: the decoder looks ahead)
static __global__ __launch_bounds__(128) void gauss_combine_kernel(const float* __restrict__ k, float* __restrict__ y, int Co, int F, int T,
                                                            long kz, long ob, long oS, long oR, long oI,
                                                            const float* __restrict__ sc, const float* __restrict__ sh,
                                                            const float* __restrict__ slope, const int* __restrict__ tlen) {
    const int f = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
    const long row = (((long)b * Co + c) * F + f) * T, orow = (long)b * ob + ((long)c * F + f) * T;
    const float s_r = sc[c], h_r = sh[c], s_i = sc[Co + c], h_i = sh[Co + c], a_r = slope[c], a_i = slope[Co + c];
    const int tv = tlen ? tlen[b] : T;
    const bool v4 = (T & 3) == 0;
    for (int t = threadIdx.x * 4; t < T; t += 512) {
        float k1[4], k2[4], k3[4], yr[4], yi[4];
        if (v4) {
            *reinterpret_cast<float4*>(k1) = *reinterpret_cast<const float4*>(k + row + t);
            *reinterpret_cast<float4*>(k2) = *reinterpret_cast<const float4*>(k + kz + row + t);
            *reinterpret_cast<float4*>(k3) = *reinterpret_cast<const float4*>(k + 2 * kz + row + t);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tt = min(t + j, T - 1);
                k1[j] = k[row + tt]; k2[j] = k[kz + row + tt]; k3[j] = k[2 * kz + row + tt];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float r = (k1[j] - k3[j]) * s_r + h_r, m = (k1[j] + k2[j]) * s_i + h_i;
            r = r >= 0.f ? r : a_r * r;
            m = m >= 0.f ? m : a_i * m;
            const bool live = t + j < tv;
            yr[j] = live ? r : 0.f;
            yi[j] = live ? m : 0.f;
        }
        if (v4) {
            *reinterpret_cast<float4*>(y + orow + oR + t) = *reinterpret_cast<const float4*>(yr);
            *reinterpret_cast<float4*>(y + orow + oI + t) = *reinterpret_cast<const float4*>(yi);
            if (oS >= 0) *reinterpret_cast<float4*>(y + orow + oS + t) = make_float4(yr[0] + yi[0], yr[1] + yi[1], yr[2] + yi[2], yr[3] + yi[3]);
        } else {
            for (int j = 0; j < 4 && t + j < T; ++j) {
                y[orow + oR + t + j] = yr[j];
                y[orow + oI + t + j] = yi[j];
                if (oS >= 0) y[orow + oS + t + j] = yr[j] + yi[j];
            }
        }
    }
}
struct GaussLayer {
    std::vector<GCPlan> pl;      // encoder: one grouped plan (Z = 3); decoder: one per output-parity class
    float *sc = nullptr, *sh = nullptr, *slope = nullptr;      // [2 co] rows [real; imag]
    int co = 0;
    void free() {
        for (auto& g : pl) gc_free_plan(g);
        pl.clear();
        for (float** p : {&sc, &sh, &slope})
            if (*p) { (void)hipFree(*p); *p = nullptr; }
    }
};


// a two-plane tensor [B][2 C][P] = [R | I] -> three planes [B][3 C][P] = [R + I | R | I] (a tensor produced by block-form layers
// that a three-product layer reads)
static __global__ __launch_bounds__(256) void gauss_planes23_kernel(const float* __restrict__ x2, float* __restrict__ x3, long CP) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4, b = blockIdx.y;
    if (i >= CP) return;
    const float* xb = x2 + b * 2 * CP;
    float* yb = x3 + b * 3 * CP;
    if (i + 3 < CP && (CP & 3) == 0) {
        const float4 r = *reinterpret_cast<const float4*>(xb + i), m = *reinterpret_cast<const float4*>(xb + CP + i);
        *reinterpret_cast<float4*>(yb + i) = make_float4(r.x + m.x, r.y + m.y, r.z + m.z, r.w + m.w);
        *reinterpret_cast<float4*>(yb + CP + i) = r;
        *reinterpret_cast<float4*>(yb + 2 * CP + i) = m;
    } else {
        for (long j = i; j < CP && j < i + 4; ++j) {
            const float r = xb[j], m = xb[CP + j];
            yb[j] = r + m;
            yb[CP + j] = r;
            yb[2 * CP + j] = m;
        }
    }
}

// the three weight matrices of a layer, stacked along z: Wr | Wi - Wr | Wr + Wi
inline std::vector<float> three_products(const std::vector<float>& r, const std::vector<float>& i) {
    std::vector<float> w(3 * r.size());
    for (size_t k = 0; k < r.size(); ++k) { w[k] = r[k]; w[r.size() + k] = i[k] - r[k]; w[2 * r.size() + k] = r[k] + i[k]; }
    return w;
}
// tap sets as make_conv_plan(w, 2, 2, 1, 1, 1, ..) / one output-parity class of make_deconv_plan(w, 2, 2, toff, ..) build them
inline TapSpec conv52_taps() {
    TapSpec ts;
    ts.ntaps = 10;
    for (int kf = 0; kf < 5; ++kf)
        for (int kt = 0; kt < 2; ++kt) { ts.df[kf * 2 + kt] = kf - 2; ts.dt[kf * 2 + kt] = kt - 1; }
    return ts;
}
inline TapSpec deconv52_taps(int par, int toff, std::vector<int>& sel) {
    TapSpec ts;
    sel.clear();
    for (int kf = 0; kf < 5; ++kf) {
        const int num = par + 2 - kf;
        if (((num % 2) + 2) % 2 != 0) continue;
        for (int kt = 0; kt < 2; ++kt) {
            ts.df[ts.ntaps] = num / 2;
            ts.dt[ts.ntaps] = toff - kt;
            ts.ntaps++;
            sel.push_back(kf * 2 + kt);
        }
    }
    return ts;
}
// grouped plans of a (5, 2) conv, stride 2 along frequency (encoder) / of both parity classes of its transposed form (decoder;
// wr / wi: [co][ci][10] real matrices, ci = all complex input channels, the first c0split of them from the first source)
inline void make_conv_plans(GaussLayer& g, const DenseW& wr, const DenseW& wi, int tout) {
    g.co = wr.M;
    g.pl.push_back(gc_make_plan(wr.M, wr.Cin, conv52_taps(), three_products(wr.w, wi.w), {}, {}, ACT_NONE, EPI_ACT, 2, 1, 0, tout, 3));
    g.pl.back().flop_scale = 4.0 / 3.0;          // the profiler books the reference's four products
}
inline void make_deconv_plans(GaussLayer& g, const DenseW& wr, const DenseW& wi, int toff, int c0split, int tout) {
    const int co = wr.M, ci = wr.Cin;
    g.co = co;
    for (int par = 0; par < 2; ++par) {
        std::vector<int> sel;
        TapSpec ts = deconv52_taps(par, toff, sel);
        std::vector<float> r((size_t)co * ci * ts.ntaps), i(r.size());
        for (int m = 0; m < co; ++m)
            for (int c = 0; c < ci; ++c)
                for (int j = 0; j < ts.ntaps; ++j) {
                    r[((size_t)m * ci + c) * ts.ntaps + j] = wr.w[((size_t)m * ci + c) * 10 + sel[j]];
                    i[((size_t)m * ci + c) * ts.ntaps + j] = wi.w[((size_t)m * ci + c) * 10 + sel[j]];
                }
        g.pl.push_back(gc_make_plan(co, ci, ts, three_products(r, i), {}, {}, ACT_NONE, EPI_ACT, 1, 2, par, tout, 3, c0split));
        g.pl.back().flop_scale = 4.0 / 3.0;
    }
}

}  // namespace gauss
}  // namespace se
