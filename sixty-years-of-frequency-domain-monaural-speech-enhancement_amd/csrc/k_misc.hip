// Elementwise / layout kernels of the decode path (HBM-bound; T-contiguous rows, one pass each).
#include "kernels.h"
#include "common.h"
#include <algorithm>

namespace se {

// ---- DCCRN 'E' mask + decode-script decompress --------------------------------------------------------------
// DCCRN/DCCRN_cprs.py:201-225  mask_mags = |M|, mask_phase = atan2(Mi/(|M|+1e-8), Mr/(|M|+1e-8)),
//   est_mags = tanh(|M|) * |X|, est_phase = angle(X) + mask_phase, DC row of the mask is zero-padded.
// DCCRN/dccrn_decode_vb.py:45-58  |est|**p_out * exp(j*angle(est)).
// cos/sin(angle X + angle M) is evaluated as the product of the two unit phasors (no atan2/sincos round trip).
// mode (DCCRN(masking_mode=...), DCCRN_cprs.py:205-223): 0 'E' (above), 1 'C' est = spec x mask (complex), 2 'R' est_r = spec_r mask_r,
// est_i = spec_i mask_i; the decode script's |est|**p_out follows in every mode
__global__ __launch_bounds__(256) void dccrn_mask_kernel(const float* __restrict__ mask, const float* __restrict__ spec,
                                                         float* __restrict__ est, int F, int T, int Tp, float p_out, int mode) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const long so = (((long)b * 2) * F + k) * Tp + t;
    const long plane = (long)F * Tp;
    float outr = 0.f, outi = 0.f;
    if (k > 0) {
        const long mo = (((long)b * 2) * (F - 1) + (k - 1)) * Tp + t;
        const float mr = mask[mo], mi = mask[mo + (long)(F - 1) * Tp];
        const float xr = spec[so], xi = spec[so + plane];
        if (mode != 0) {
            float er = mode == 1 ? xr * mr - xi * mi : xr * mr, ei = mode == 1 ? xr * mi + xi * mr : xi * mi;
            if (p_out != 1.f) {
                const float mg = sqrtf(er * er + ei * ei);
                const float sc = mg > 0.f ? ((p_out == 2.f) ? mg : powf(mg, p_out - 1.f)) : 0.f;
                er *= sc;
                ei *= sc;
            }
            est[so] = er;
            est[so + plane] = ei;
            return;
        }
        const float mm = sqrtf(mr * mr + mi * mi);
        const float xm = sqrtf(xr * xr + xi * xi);
        float pr = 1.f, pi = 0.f, qr = 1.f, qi = 0.f;
        if (mm > 0.f) { pr = mr / mm; pi = mi / mm; }
        if (xm > 0.f) { qr = xr / xm; qi = xi / xm; }
        float em = tanhf(mm) * xm;
        if (p_out == 2.f) em = em * em;
        else if (p_out != 1.f) em = powf(em, p_out);
        outr = em * (pr * qr - pi * qi);
        outi = em * (pr * qi + pi * qr);
    }
    est[so] = outr;
    est[so + plane] = outi;
}

void launch_dccrn_mask(const float* mask, const float* spec, float* est, int B, int F, int T, int Tp, float p_out,
                       hipStream_t s, int mode) {
    StageScope prof(STAGE_MASK, s, (8.0 * (F - 1) + 16.0 * F) * T * B);
    hipLaunchKernelGGL(dccrn_mask_kernel, dim3((T + 255) / 256, F, B), dim3(256), 0, s, mask, spec, est, F, T, Tp, p_out, mode);
    SE_HIP(hipGetLastError());
}

// ---- out[t][k][a] = in[a][k][t] : 32x32 LDS tile transpose ------------------------------------------------
__global__ __launch_bounds__(256) void transpose_akt_kernel(const float* __restrict__ in, float* __restrict__ out, int A,
                                                            int T, long in_sa, long in_sk, long out_st, long out_sk) {
    __shared__ float tile[32][33];
    const int k = blockIdx.z;
    const int a0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int a = a0 + ty + 8 * i, t = t0 + tx;
        tile[ty + 8 * i][tx] = (a < A && t < T) ? in[(long)a * in_sa + (long)k * in_sk + t] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int t = t0 + ty + 8 * i, a = a0 + tx;
        if (a < A && t < T) out[(long)t * out_st + (long)k * out_sk + a] = tile[tx][ty + 8 * i];
    }
}

void launch_transpose_akt(const float* in, float* out, int A, int K, int T, long in_sa, long in_sk, long out_st,
                          long out_sk, hipStream_t s) {
    hipLaunchKernelGGL(transpose_akt_kernel, dim3((T + 31) / 32, (A + 31) / 32, K), dim3(256), 0, s, in, out, A, T,
                       in_sa, in_sk, out_st, out_sk);
    SE_HIP(hipGetLastError());
}

// ---- strided 4-D copy ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy4_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int I,
                                                    int J, long si_b, long si_c, long si_i, long si_j, long so_b,
                                                    long so_c, long so_i) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= J) return;
    const int i = blockIdx.y % I, c = blockIdx.y / I, b = blockIdx.z;
    out[(long)b * so_b + (long)c * so_c + (long)i * so_i + j] = in[(long)b * si_b + (long)c * si_c + (long)i * si_i + (long)j * si_j];
}

void launch_copy4(const float* in, float* out, int B, int C, int I, int J, long si_b, long si_c, long si_i, long si_j,
                  long so_b, long so_c, long so_i, hipStream_t s) {
    hipLaunchKernelGGL(copy4_kernel, dim3((J + 255) / 256, C * I, B), dim3(256), 0, s, in, out, C, I, J, si_b, si_c,
                       si_i, si_j, so_b, so_c, so_i);
    SE_HIP(hipGetLastError());
}

constexpr int NU = 8;      // independent loads per thread in the streaming passes of the norm kernels

// ---- LayerNorm over (C, F) per (b, t) + residual ------------------------------------------------------------
// block = 64 frames x 4 row groups: the C*F rows of a frame are split over 4 waves (partial sums meet in LDS), so a
// launch has 4x the blocks and a quarter of the serial row walk of a thread-per-frame layout
__global__ __launch_bounds__(256) void layernorm_cf_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                           const float* __restrict__ w, const float* __restrict__ bb,
                                                           float* __restrict__ out, int C, int F, int T, float eps, int post,
                                                           const float* __restrict__ prelu_slope) {
    __shared__ double red[4][64], red2[4][64];
    const int tl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tl, b = blockIdx.y;
    const bool ok = t < T;
    const long base = (long)b * C * F * T + (ok ? t : T - 1);
    const int n = C * F;
    // one statistics pass (sum and sum of squares in fp64), then the normalise pass
    double s = 0.0, v = 0.0;
    {
        int i = rg;
        for (; i + 4 * (NU - 1) < n; i += 4 * NU) {         // NU rows in flight per thread, summed in row order
            float xr[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) xr[u] = x[base + (long)(i + 4 * u) * T];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const double xv = xr[u];
                s += xv;
                v += xv * xv;
            }
        }
        for (; i < n; i += 4) {
            const double xv = x[base + (long)i * T];
            s += xv;
            v += xv * xv;
        }
    }
    red[rg][tl] = s;
    red2[rg][tl] = v;
    __syncthreads();
    const double mud = (red[0][tl] + red[1][tl] + red[2][tl] + red[3][tl]) / n;
    const double vard = fmax((red2[0][tl] + red2[1][tl] + red2[2][tl] + red2[3][tl]) / n - mud * mud, 0.0);
    const float mu = (float)mud;
    const float rs = (float)(1.0 / sqrt(vard + (double)eps));
    if (!ok) return;
    const float slope = prelu_slope ? prelu_slope[0] : 1.f;
    auto finish = [&](int i, float xv, float rv) {
        const int c = i / F, f = i - c * F;
        float y = (xv - mu) * rs * w[f * C + c] + bb[f * C + c];
        if (post == 1) y = y * fm_sigmoid(y);      // swish on the hardware exp2 / reciprocal
        if (prelu_slope) y = y >= 0.f ? y : slope * y;
        if (res) y += rv;
        out[base + (long)i * T] = y;
    };
    int i = rg;
    for (; i + 4 * (NU - 1) < n; i += 4 * NU) {
        float xr[NU], rr[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const long o = base + (long)(i + 4 * u) * T;
            xr[u] = x[o];
            rr[u] = res ? res[o] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) finish(i + 4 * u, xr[u], rr[u]);
    }
    for (; i < n; i += 4) {
        const long o = base + (long)i * T;
        finish(i, x[o], res ? res[o] : 0.f);
    }
}
void launch_layernorm_cf(const float* x, const float* res, const float* w, const float* b, float* out, int B, int C,
                         int F, int T, float eps, hipStream_t s, int post, const float* prelu_slope) {
    hipLaunchKernelGGL(layernorm_cf_kernel, dim3((T + 63) / 64, B), dim3(256), 0, s, x, res, w, b, out, C, F, T, eps, post,
                       prelu_slope);
    SE_HIP(hipGetLastError());
}

__device__ __forceinline__ float pow_scale(float m, float p) {   // m^p / m
    if (p == 1.f) return 1.f;
    if (m <= 0.f) return 0.f;
    return (p == 2.f) ? m : powf(m, p - 1.f);
}

__global__ __launch_bounds__(256) void cmask_apply_kernel(const float* __restrict__ mask, const float* __restrict__ spec,
                                                          float* __restrict__ out, long plane, long total, float p_out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / plane, r = i - b * plane;
    const long o = b * 2 * plane + r;
    const float mr = mask[o], mi = mask[o + plane], xr = spec[o], xi = spec[o + plane];
    float er = xr * mr - xi * mi, ei = xr * mi + xi * mr;
    const float sc = pow_scale(sqrtf(er * er + ei * ei), p_out);
    out[o] = er * sc;
    out[o + plane] = ei * sc;
}
void launch_cmask_apply(const float* mask, const float* spec, float* out, int B, int F, int T, float p_out,
                        hipStream_t s) {
    const long plane = (long)F * T, total = plane * B;
    StageScope prof(STAGE_MASK, s, 24.0 * total);
    hipLaunchKernelGGL(cmask_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, mask, spec, out, plane,
                       total, p_out);
    SE_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void mag_phase_kernel(const float* __restrict__ mag, const float* __restrict__ spec,
                                                        float* __restrict__ out, long plane, long total, float p_out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / plane, r = i - b * plane;
    const long o = b * 2 * plane + r;
    float m = mag[i];
    if (p_out == 2.f) m = m * m;
    else if (p_out != 1.f) m = powf(m, p_out);
    const float xr = spec[o], xi = spec[o + plane];
    const float xm = sqrtf(xr * xr + xi * xi);
    float pr = 1.f, pi = 0.f;                 // np.angle(0) = 0
    if (xm > 0.f) { pr = xr / xm; pi = xi / xm; }
    out[o] = m * pr;
    out[o + plane] = m * pi;
}
void launch_mag_phase(const float* mag, const float* spec, float* out, int B, int F, int T, float p_out,
                      hipStream_t s) {
    const long plane = (long)F * T, total = plane * B;
    StageScope prof(STAGE_MASK, s, 20.0 * total);
    hipLaunchKernelGGL(mag_phase_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, mag, spec, out, plane,
                       total, p_out);
    SE_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void elu_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float v = x[i];
        y[i] = v > 0.f ? v : fm_expm1(v);      // (as gemmconv.hip: act_apply)
    }
}
void launch_elu(const float* x, float* y, long n, hipStream_t s) {
    hipLaunchKernelGGL(elu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n);
    SE_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void polar_pow_kernel(const float* __restrict__ x, float* __restrict__ out, long plane,
                                                        long total, float p_out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long b = i / plane, r = i - b * plane;
    const long o = b * 2 * plane + r;
    const float er = x[o], ei = x[o + plane];
    const float sc = pow_scale(sqrtf(er * er + ei * ei), p_out);
    out[o] = er * sc;
    out[o + plane] = ei * sc;
}
void launch_polar_pow(const float* x, float* out, int B, int F, int T, float p_out, hipStream_t s) {
    const long plane = (long)F * T, total = plane * B;
    StageScope prof(STAGE_MASK, s, 16.0 * total);
    hipLaunchKernelGGL(polar_pow_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, out, plane, total, p_out);
    SE_HIP(hipGetLastError());
}

// ---- InstanceNorm over a contiguous plane + PReLU ------------------------------------------------------------
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// y = PReLU((x - mu) * rs * g + bt) (+ r) over one plane, NU elements per thread in flight (r may alias y)
__device__ __forceinline__ void norm_apply_pass(const float* __restrict__ xp, float* yp, const float* rp, int P, float muf,
                                                float rs, float g, float bt, float sl) {
    // 16 B accesses where the plane allows them: input, output (and residual) share their misalignment - a plane starts at an
    // odd multiple of 4 B when P is odd - so a head of <= 3 values brings all of them to a 16 B boundary
    if (((((size_t)xp ^ (size_t)yp) & 15) == 0) && (!rp || ((((size_t)xp ^ (size_t)rp) & 15) == 0))) {
        const int head = min(P, (int)((4 - (((size_t)xp >> 2) & 3)) & 3));
        auto one = [&](int k) {
            float o = (xp[k] - muf) * rs * g + bt;
            o = o >= 0.f ? o : sl * o;
            yp[k] = rp ? o + rp[k] : o;
        };
        if ((int)threadIdx.x < head) one(threadIdx.x);
        const int n4 = (P - head) >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(xp + head);
        float4* y4 = reinterpret_cast<float4*>(yp + head);
        const float4* r4 = rp ? reinterpret_cast<const float4*>(rp + head) : nullptr;
        constexpr int NV = 4;
        auto f4 = [&](float4 v, float4 r) {
            float o[4] = {v.x, v.y, v.z, v.w}, q[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t = (o[k] - muf) * rs * g + bt;
                t = t >= 0.f ? t : sl * t;
                o[k] = rp ? t + q[k] : t;
            }
            return make_float4(o[0], o[1], o[2], o[3]);
        };
        int i = threadIdx.x;
        for (; i + (NV - 1) * 256 < n4; i += NV * 256) {
            float4 v[NV], r[NV];
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                v[u] = x4[i + u * 256];
                r[u] = r4 ? r4[i + u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < NV; ++u) y4[i + u * 256] = f4(v[u], r[u]);
        }
        for (; i < n4; i += 256) y4[i] = f4(x4[i], r4 ? r4[i] : make_float4(0.f, 0.f, 0.f, 0.f));
        const int done = head + 4 * n4;
        if (done + (int)threadIdx.x < P) one(done + threadIdx.x);
        return;
    }
    int i = threadIdx.x;
    for (; i + (NU - 1) * 256 < P; i += NU * 256) {
        float v[NU], r[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            v[u] = xp[i + u * 256];
            r[u] = rp ? rp[i + u * 256] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            float o = (v[u] - muf) * rs * g + bt;
            o = o >= 0.f ? o : sl * o;
            yp[i + u * 256] = rp ? o + r[u] : o;
        }
    }
    for (; i < P; i += 256) {
        float o = (xp[i] - muf) * rs * g + bt;
        o = o >= 0.f ? o : sl * o;
        yp[i] = rp ? o + rp[i] : o;
    }
}

// Ragged batch: the statistics of row b cover only its own tlen[b] frames of every T-frame line of the plane (the
// reference decodes each clip alone: InstanceNorm never sees another clip's padding); the normalise pass still covers the
// whole plane - the tail frames are dead values no valid frame ever reads (every conv on these models is causal in time).
__global__ __launch_bounds__(256) void instnorm_prelu_ragged_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta,
                                                                    const float* __restrict__ slope, const float* res, int C,
                                                                    int P, int T, const int* __restrict__ tlen) {
    __shared__ double sh[4];
    const int c = blockIdx.x % C, b = blockIdx.x / C;
    const int Tb = tlen[b];
    const float* xp = x + (long)blockIdx.x * P;
    float* yp = y + (long)blockIdx.x * P;
    const float* rp = res ? res + (long)blockIdx.x * P : nullptr;
    double s = 0.0, q = 0.0;
    {
        int i = threadIdx.x;
        for (; i + (NU - 1) * 256 < P; i += NU * 256) {
            float v[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) v[u] = xp[i + u * 256];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const double d = ((i + u * 256) % T < Tb) ? v[u] : 0.f;
                s += d;
                q += d * d;
            }
        }
        for (; i < P; i += 256) {
            const double d = (i % T < Tb) ? xp[i] : 0.f;
            s += d;
            q += d * d;
        }
    }
    const double cnt = (double)(P / T) * Tb;
    const double mu = block_sum_d(s, sh) / cnt;
    const double var = fmax(block_sum_d(q, sh) / cnt - mu * mu, 0.0);
    const float rs = (float)(1.0 / sqrt(var + 1e-5)), muf = (float)mu;
    const float g = gamma[c], bt = beta[c], sl = slope ? slope[c] : 1.f;
    norm_apply_pass(xp, yp, rp, P, muf, rs, g, bt, sl);
}

__global__ __launch_bounds__(256) void instnorm_prelu_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ slope, const float* res, int C,
                                                             int P) {
    __shared__ double sh[4];
    const int c = blockIdx.x % C;
    const float* xp = x + (long)blockIdx.x * P;
    float* yp = y + (long)blockIdx.x * P;
    const float* rp = res ? res + (long)blockIdx.x * P : nullptr;     // optional residual, may alias y
    // one statistics pass: sum and sum of squares in fp64 (the cancellation in E[x^2] - mu^2 costs ~1e-16 * mu^2 / var,
    // far below fp32 resolution), then one normalise pass: 2 reads + 1 write of the plane instead of 3 + 1
    // NU loads in flight per thread (the compiler does not unroll these loops: one load, one vmcnt(0) per element left the
    // pass latency bound); sums keep their element order
    double s = 0.0, q = 0.0;
    {
        int i = threadIdx.x;
        for (; i + (NU - 1) * 256 < P; i += NU * 256) {
            float v[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) v[u] = xp[i + u * 256];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const double d = v[u];
                s += d;
                q += d * d;
            }
        }
        for (; i < P; i += 256) {
            const double d = xp[i];
            s += d;
            q += d * d;
        }
    }
    const double mu = block_sum_d(s, sh) / P;
    const double var = fmax(block_sum_d(q, sh) / P - mu * mu, 0.0);
    const float rs = (float)(1.0 / sqrt(var + 1e-5)), muf = (float)mu;
    const float g = gamma[c], bt = beta[c], sl = slope ? slope[c] : 1.f;
    norm_apply_pass(xp, yp, rp, P, muf, rs, g, bt, sl);
}
// the same with the statistics handed over by the producing conv's epilogue (GCParams::stats): nslot (sum, sum of squares)
// pairs per (b, c) plane, combined in fp64 in a fixed order; the plane itself is read once
__global__ __launch_bounds__(256) void instnorm_prelu_stats_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta,
                                                                   const float* __restrict__ slope, const float* res,
                                                                   const float* __restrict__ stats, int nslot, int C, int P,
                                                                   const int* __restrict__ tlen, int T) {
    __shared__ double sh[4];
    const int c = blockIdx.x % C;
    // ragged batch: the epilogue's partial sums were cut at the row's own frame count (gc_kernel: tstat), so is the divisor
    const double Pn = tlen ? (double)(P / T) * min(tlen[blockIdx.x / C], T) : (double)P;
    const float* xp = x + (long)blockIdx.x * P;
    float* yp = y + (long)blockIdx.x * P;
    const float* rp = res ? res + (long)blockIdx.x * P : nullptr;
    const float2* sp = reinterpret_cast<const float2*>(stats) + (long)blockIdx.x * nslot;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nslot; i += 256) {
        const float2 v = sp[i];
        s += v.x;
        q += v.y;
    }
    const double mu = block_sum_d(s, sh) / Pn;
    const double var = fmax(block_sum_d(q, sh) / Pn - mu * mu, 0.0);
    const float rs = (float)(1.0 / sqrt(var + 1e-5)), muf = (float)mu;
    const float g = gamma[c], bt = beta[c], sl = slope ? slope[c] : 1.f;
    norm_apply_pass(xp, yp, rp, P, muf, rs, g, bt, sl);
}
void launch_instnorm_prelu_stats(const float* x, float* y, const float* gamma, const float* beta, const float* slope,
                                 const float* stats, int nslot, int B, int C, int P, hipStream_t s, const float* res, int T) {
    const Ragged* rg = ragged_ctx();
    SE_CHECK(!rg || (T > 0 && P % T == 0), "ragged InstanceNorm from epilogue statistics needs the frame count of the plane's lines");
    hipLaunchKernelGGL(instnorm_prelu_stats_kernel, dim3(B * C), dim3(256), 0, s, x, y, gamma, beta, slope, res, stats, nslot,
                       C, P, rg ? rg->tlen : nullptr, T);
    SE_HIP(hipGetLastError());
}
// ---- InstanceNorm folded into the consumers (round 5) ---------------------------------------------------------------------------
// The epilogue statistics of a conv (GCParams::stats) -> the per-(b, c) parameters its CONSUMERS apply on the fly
// (GCParams::nrm0 / nrm1): float4 {scale = rstd * gamma, shift = beta - mean * scale, slope - 1, x0 = -shift / scale}.  One wave
// per (b, c) plane combines the nslot (sum, sum of squares) pairs in fp64 in a fixed order, like instnorm_prelu_stats_kernel.
__global__ __launch_bounds__(256) void instnorm_finalize_kernel(const float* __restrict__ stats, int nslot, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ slope,
                                                                float* __restrict__ nrm, int C, int P, int planes,
                                                                const int* __restrict__ tlen, int T) {
    const int pl = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (pl >= planes) return;
    const float2* sp = reinterpret_cast<const float2*>(stats) + (long)pl * nslot;
    double s = 0.0, q = 0.0;
    for (int i = lane; i < nslot; i += 64) {
        const float2 v = sp[i];
        s += v.x;
        q += v.y;
    }
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_down(s, o, 64);
        q += __shfl_down(q, o, 64);
    }
    if (lane == 0) {
        const int c = pl % C;
        const double Pn = tlen ? (double)(P / T) * min(tlen[pl / C], T) : (double)P;      // (ragged batch: the row's own frames)
        const double mu = s / Pn, var = fmax(q / Pn - mu * mu, 0.0);
        const double sc = (1.0 / sqrt(var + 1e-5)) * (double)gamma[c], sh = (double)beta[c] - mu * sc;
        // x0: the raw value that normalises to (numerically) zero - what a left-pad frame is staged as.  A zero gain has no such
        // value; the fold is not used for such a layer (blocks.h checks gamma at load time), the field is then unused
        reinterpret_cast<float4*>(nrm)[pl] = make_float4((float)sc, (float)sh, (slope ? slope[c] : 1.f) - 1.f, sc != 0.0 ? (float)(-sh / sc) : 0.f);
    }
}
void launch_instnorm_finalize(const float* stats, int nslot, const float* gamma, const float* beta, const float* slope, float* nrm,
                              int B, int C, int P, hipStream_t s, int T) {
    const Ragged* rg = ragged_ctx();
    SE_CHECK(!rg || (T > 0 && P % T == 0), "ragged InstanceNorm from epilogue statistics needs the frame count of the plane's lines");
    const int planes = B * C;
    hipLaunchKernelGGL(instnorm_finalize_kernel, dim3((planes + 3) / 4), dim3(256), 0, s, stats, nslot, gamma, beta, slope, nrm, C, P,
                       planes, rg ? rg->tlen : nullptr, T);
    SE_HIP(hipGetLastError());
}
// y = f_a(xa) + f_b(xb) over one (b, c) plane per block, f = the on-the-fly normalisation of gc_kernel NRM (same two fused
// multiply-adds, so a value is the same whether a conv consumes it from the raw tensor or from this pass's output): the last
// decoder level of a U^2-Net module + the module's residual (TaylorSENet.py:489-494), whose sum is what the NEXT module reads.
// nb = nullptr: y = f_a(xa).  y may alias xa or xb.
__global__ __launch_bounds__(256) void instnorm_apply2_kernel(const float* xa, const float* __restrict__ na, const float* xb,
                                                              const float* __restrict__ nb, float* y, int P) {
    const long base = (long)blockIdx.x * P;
    const float4 pa = reinterpret_cast<const float4*>(na)[blockIdx.x];
    const float4 pb = nb ? reinterpret_cast<const float4*>(nb)[blockIdx.x] : make_float4(0.f, 0.f, 0.f, 0.f);
    auto f = [](float x, const float4& p) {
        const float t = fmaf(x, p.x, p.y);
        return fmaf(fminf(t, 0.f), p.z, t);
    };
    const float* ap = xa + base;
    const bool two = nb != nullptr;
    const float* bp = two ? xb + base : ap;
    float* yp = y + base;
    auto one = [&](int k) { yp[k] = f(ap[k], pa) + (two ? f(bp[k], pb) : 0.f); };
    // 16 B accesses: the tensors share their misalignment (a plane starts at an odd multiple of 4 B when P is odd), so a head of
    // <= 3 values brings all of them to a 16 B boundary (as norm_apply_pass)
    if (((((size_t)ap ^ (size_t)yp) | ((size_t)ap ^ (size_t)bp)) & 15) == 0) {
        const int head = min(P, (int)((4 - (((size_t)ap >> 2) & 3)) & 3));
        if ((int)threadIdx.x < head) one(threadIdx.x);
        const int n4 = (P - head) >> 2;
        const float4* a4 = reinterpret_cast<const float4*>(ap + head);
        const float4* b4 = reinterpret_cast<const float4*>(bp + head);
        float4* y4 = reinterpret_cast<float4*>(yp + head);
        auto f4 = [&](const float4& va, const float4& vb) {
            float4 o;
            o.x = f(va.x, pa) + (two ? f(vb.x, pb) : 0.f);
            o.y = f(va.y, pa) + (two ? f(vb.y, pb) : 0.f);
            o.z = f(va.z, pa) + (two ? f(vb.z, pb) : 0.f);
            o.w = f(va.w, pa) + (two ? f(vb.w, pb) : 0.f);
            return o;
        };
        constexpr int NV = 4;
        int i = threadIdx.x;
        for (; i + (NV - 1) * 256 < n4; i += NV * 256) {
            float4 va[NV], vb[NV];
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                va[u] = a4[i + u * 256];
                vb[u] = two ? b4[i + u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < NV; ++u) y4[i + u * 256] = f4(va[u], vb[u]);
        }
        for (; i < n4; i += 256) y4[i] = f4(a4[i], two ? b4[i] : make_float4(0.f, 0.f, 0.f, 0.f));
        const int done = head + 4 * n4;
        if (done + (int)threadIdx.x < P) one(done + threadIdx.x);
        return;
    }
    for (int i = threadIdx.x; i < P; i += 256) one(i);
}
void launch_instnorm_apply2(const float* xa, const float* na, const float* xb, const float* nb, float* y, int B, int C, int P,
                            hipStream_t s) {
    hipLaunchKernelGGL(instnorm_apply2_kernel, dim3(B * C), dim3(256), 0, s, xa, na, xb, nb, y, P);
    SE_HIP(hipGetLastError());
}
void launch_instnorm_prelu(const float* x, float* y, const float* gamma, const float* beta, const float* slope, int B,
                           int C, int P, hipStream_t s, const float* res, int T) {
    if (const Ragged* rg = ragged_ctx()) {
        SE_CHECK(T > 0 && P % T == 0, "ragged InstanceNorm needs the frame count of the plane's lines");
        hipLaunchKernelGGL(instnorm_prelu_ragged_kernel, dim3(B * C), dim3(256), 0, s, x, y, gamma, beta, slope, res, C, P, T,
                           rg->tlen);
        SE_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(instnorm_prelu_kernel, dim3(B * C), dim3(256), 0, s, x, y, gamma, beta, slope, res, C, P);
    SE_HIP(hipGetLastError());
}

// ---- PReLU -> InstanceNorm1d -> shared causal FIR, one block per (b, c) row -----------------------------------
__global__ __launch_bounds__(256) void tcm_head_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       const float* __restrict__ slope, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ fir, int K,
                                                       int C, int T, const int* __restrict__ tlen) {
    extern __shared__ float row[];      // [T] normalised row, then FIR input
    __shared__ double sh[4];
    const int c = blockIdx.x % C;
    const int Ts = tlen ? tlen[blockIdx.x / C] : T;      // frames the statistics cover (ragged batch: the row's own)
    const float* xp = x + (long)blockIdx.x * T;
    float* yp = y + (long)blockIdx.x * T;
    const float sl = slope[c];
    double s = 0.0;
    {
        int i = threadIdx.x;
        for (; i + 256 < T; i += 512) {       // two loads in flight: a 4 s clip (T = 401) is one round
            const float v0 = xp[i], v1 = xp[i + 256];
            const float a0 = v0 >= 0.f ? v0 : sl * v0, a1 = v1 >= 0.f ? v1 : sl * v1;
            row[i] = a0;
            row[i + 256] = a1;
            if (i < Ts) s += a0;
            if (i + 256 < Ts) s += a1;
        }
        for (; i < T; i += 256) {
            float a = xp[i];
            a = a >= 0.f ? a : sl * a;
            row[i] = a;
            if (i < Ts) s += a;
        }
    }
    const double mu = block_sum_d(s, sh) / Ts;
    double vv = 0.0;
    for (int i = threadIdx.x; i < Ts; i += 256) {
        const double d = row[i] - mu;
        vv += d * d;
    }
    const double var = block_sum_d(vv, sh) / Ts;
    const float rs = (float)(1.0 / sqrt(var + 1e-5)), muf = (float)mu, g = gamma[c], bt = beta[c];
    for (int i = threadIdx.x; i < T; i += 256) row[i] = (row[i] - muf) * rs * g + bt;
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) {
        float o;
        if (K <= 0) {
            o = row[t];
        } else {
            o = 0.f;
            for (int k = 0; k < K; ++k) {      // y[t] = sum_k w[k] * x[t - (K-1) + k]
                const int ti = t - (K - 1) + k;
                if (ti >= 0) o += fir[k] * row[ti];
            }
        }
        yp[t] = o;
    }
}
void launch_tcm_head(const float* x, float* y, const float* slope, const float* gamma, const float* beta,
                     const float* fir, int K, int B, int C, int T, hipStream_t s) {
    SE_CHECK((size_t)T * 4 <= 60000, "TCM row too long for the LDS-resident head kernel");
    const Ragged* rg = ragged_ctx();
    hipLaunchKernelGGL(tcm_head_kernel, dim3(B * C), dim3(256), (size_t)T * 4, s, x, y, slope, gamma, beta, fir, K, C, T,
                       rg ? rg->tlen : nullptr);
    SE_HIP(hipGetLastError());
}

// ---- CumulativeLayerNorm (the `_new` variants) ----------------------------------------------------------------
// Frame t is normalised with the mean / variance of ALL rows of frames 0..t (CTSNet_new/Step1_network.py:213-286).
// x [B][R][T] with R = C * F rows (F = 1 for the 1-D flavour); three passes: per-frame sums over the rows, an
// in-LDS prefix scan per utterance, then normalise + affine (+ PReLU before / after, + the TCM branch FIR).
// Frame-online chunks (c0 > 0 / carry): only columns [c0, T) of the window are summed and scanned, the scan starts from
// the carried totals of the frames before column c0 (`tg0` = stream index of column 0; columns before the start of the
// stream do not count) and leaves the totals up to column c0 + n - 1 - the frames before the next window's column c0.
__global__ __launch_bounds__(256) void cln_stats_kernel(const float* __restrict__ x, const float* __restrict__ pre_slope,
                                                        double* __restrict__ sum, double* __restrict__ sq, int R, int F,
                                                        int T, int c0, int WP) {
    // WP columns per workgroup, 256 / WP threads share a column's rows (whole utterances: 64 x 4; narrow frame-online
    // windows put more threads on each column)
    __shared__ double sh[2][256];
    const int NL = 256 / WP, tl = threadIdx.x % WP, rg = threadIdx.x / WP, t = c0 + blockIdx.x * WP + tl, b = blockIdx.y;
    double s = 0.0, q = 0.0;
    if (t < T) {
        const float* xp = x + (long)b * R * T + t;
        auto take = [&](int r, float v) {
            if (pre_slope) v = v >= 0.f ? v : pre_slope[r / F] * v;
            s += v;
            q += (double)v * v;
        };
        int r = rg;
        for (; r + NL * (NU - 1) < R; r += NL * NU) {         // NU rows in flight per thread, summed in row order
            float xr[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) xr[u] = xp[(long)(r + NL * u) * T];
#pragma unroll
            for (int u = 0; u < NU; ++u) take(r + NL * u, xr[u]);
        }
        for (; r < R; r += NL) take(r, xp[(long)r * T]);
    }
    sh[0][rg * WP + tl] = s;
    sh[1][rg * WP + tl] = q;
    __syncthreads();
    if (rg == 0 && t < T) {
        double a = sh[0][tl], c = sh[1][tl];
        for (int l = 1; l < NL; ++l) {
            a += sh[0][l * WP + tl];
            c += sh[1][l * WP + tl];
        }
        sum[(long)b * T + t] = a;
        sq[(long)b * T + t] = c;
    }
}

// In-LDS inclusive prefix sums of sc[c0 .. T) and sc[T + c0 .. 2T), starting from (a0, q0), by ONE wave: a lane sums its contiguous
// segment, the 64 segment totals are scanned on the shuffle network, the lane walks its segment again with its offset.  The serial
// loop of rounds 2-5 (thread 0 over all frames, two dependent LDS round trips per frame) was 29 us for T = 416 - hidden behind the
// other utterances' blocks at batch 256, but 144 launches x 29 us = 4.2 of the 10 ms of a single clip's decode (round 6).
// Sums in double precision; only the association order differs from the serial loop (~1e-16 relative).
__device__ __forceinline__ void cln_wave_scan(double* sc, int T, int c0, double a0, double q0) {
    const int lane = threadIdx.x;            // (called by threads 0..63)
    const int n = T - c0, L = (n + 63) >> 6;
    const int lo = c0 + lane * L, hi = min(lo + L, T);
    double a = 0.0, q = 0.0;
    for (int t = lo; t < hi; ++t) {
        a += sc[t];
        q += sc[T + t];
    }
    double ia = a, iq = q;                   // inclusive scan of the segment totals
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double ua = __shfl_up(ia, o, 64), uq = __shfl_up(iq, o, 64);
        if (lane >= o) {
            ia += ua;
            iq += uq;
        }
    }
    a = a0 + (ia - a);                       // exclusive offset of this lane's segment
    q = q0 + (iq - q);
    for (int t = lo; t < hi; ++t) {
        a += sc[t];
        q += sc[T + t];
        sc[t] = a;
        sc[T + t] = q;
    }
}

__global__ __launch_bounds__(256) void cln_scan_kernel(const double* __restrict__ sum, const double* __restrict__ sq,
                                                       float* __restrict__ mean, float* __restrict__ rstd, int R, int T,
                                                       int c0, long tg0, int n_new, double* __restrict__ carry) {
    extern __shared__ double sc[];       // [2][T]
    const int b = blockIdx.x;
    for (int t = c0 + threadIdx.x; t < T; t += 256) {
        const bool live = tg0 + t >= 0;
        sc[t] = live ? sum[(long)b * T + t] : 0.0;
        sc[T + t] = live ? sq[(long)b * T + t] : 0.0;
    }
    __syncthreads();
    if (threadIdx.x < 64 && c0 < T) {
        cln_wave_scan(sc, T, c0, carry ? carry[2 * b] : 0.0, carry ? carry[2 * b + 1] : 0.0);
        // (the carry was read by every lane before any lane writes it: the wave runs in lockstep through the scan)
        const int tc = c0 + n_new - 1;
        const int L = (T - c0 + 63) >> 6;
        if (carry && tc >= c0 && tc < T && (int)threadIdx.x == (tc - c0) / L) {
            carry[2 * b] = sc[tc];
            carry[2 * b + 1] = sc[T + tc];
        }
    }
    __syncthreads();
    for (int t = c0 + threadIdx.x; t < T; t += 256) {
        const double cnt = (double)R * (double)(tg0 + t >= 0 ? tg0 + t + 1 : 1), mu = sc[t] / cnt;
        const double var = (sc[T + t] - 2.0 * mu * sc[t]) / cnt + mu * mu;
        mean[(long)b * T + t] = (float)mu;
        rstd[(long)b * T + t] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

// The scan with the per-frame sums handed over by the producing conv's epilogue (GCParams::cstats: parts [B][F][T][2] floats - per
// output row and frame the sums over all channels): the rows are added in double precision in row order, then the serial scan as above
__global__ __launch_bounds__(256) void cln_scan_parts_kernel(const float* __restrict__ parts, int F, float* __restrict__ mean,
                                                             float* __restrict__ rstd, int R, int T) {
    extern __shared__ double sc[];       // [2][T]
    const int b = blockIdx.x;
    const float2* pb = reinterpret_cast<const float2*>(parts) + (long)b * F * T;
    for (int t = threadIdx.x; t < T; t += 256) {
        double a = 0.0, q = 0.0;
        int f = 0;
        for (; f + 7 < F; f += 8) {           // eight rows in flight (one workgroup per utterance: the loads' latency is the pass)
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = pb[(long)(f + u) * T + t];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a += v[u].x;
                q += v[u].y;
            }
        }
        for (; f < F; ++f) {
            const float2 v = pb[(long)f * T + t];
            a += v.x;
            q += v.y;
        }
        sc[t] = a;
        sc[T + t] = q;
    }
    __syncthreads();
    if (threadIdx.x < 64) cln_wave_scan(sc, T, 0, 0.0, 0.0);
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) {
        const double cnt = (double)R * (double)(t + 1), mu = sc[t] / cnt;
        const double var = (sc[T + t] - 2.0 * mu * sc[t]) / cnt + mu * mu;
        mean[(long)b * T + t] = (float)mu;
        rstd[(long)b * T + t] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

__global__ __launch_bounds__(256) void cln_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gain, const float* __restrict__ bias,
                                                        const float* __restrict__ pre_slope,
                                                        const float* __restrict__ post_slope, const float* __restrict__ fir,
                                                        int K, int R, int F, int T, int t_first, long tg0) {
    const int t = t_first + blockIdx.x * 256 + threadIdx.x, b = blockIdx.z;
    if (t >= T) return;
    const float* mu = mean + (long)b * T;
    const float* rs = rstd + (long)b * T;
    for (int r = blockIdx.y * 8; r < min(R, blockIdx.y * 8 + 8); ++r) {
        const int c = r / F;
        const float* xp = x + ((long)b * R + r) * T;
        const float g = gain[c], bt = bias[c], ps = pre_slope ? pre_slope[c] : 1.f;
        auto nrm = [&](int ti) {
            float v = xp[ti];
            v = v >= 0.f ? v : ps * v;
            return (v - mu[ti]) * rs[ti] * g + bt;
        };
        float o;
        if (K <= 0) {
            o = nrm(t);
            if (post_slope) o = o >= 0.f ? o : post_slope[c] * o;
        } else {
            o = 0.f;
            for (int k = 0; k < K; ++k) {
                const int ti = t - (K - 1) + k;
                if (tg0 + ti >= 0) o += fir[k] * nrm(ti);
            }
        }
        y[((long)b * R + r) * T + t] = o;
    }
}

// Offline form of the apply pass for the 2-D norms of the U-Net levels (no pre-activation, no FIR): one workgroup per (b, c) plane
// of F * T contiguous values, the utterance's per-frame (mean, rstd) in LDS, 16 B accesses with four of them in flight per thread
// (cln_apply_kernel above walks eight rows of one column with one 4 B load in flight: 8.7 ms per G2Net_new step for the traffic
// the InstanceNorm pass of the base model moved in 6.3), and the residual of the module's last level added here instead of by
// an add_kernel pass behind it (4 % / 7 % of a G2Net_new / TaylorSENet_new step).  res may alias y.
__global__ __launch_bounds__(256) void cln_apply_plane_kernel(const float* x, float* y, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ gain,
                                                              const float* __restrict__ bias, const float* __restrict__ post_slope,
                                                              const float* res, int C, int F, int T) {
    extern __shared__ float cl_sm[];      // mu [T], rs [T]
    const int c = blockIdx.x % C, b = blockIdx.x / C, tid = threadIdx.x;
    float* mu = cl_sm;
    float* rs = cl_sm + T;
    for (int t = tid; t < T; t += 256) {
        mu[t] = mean[(long)b * T + t];
        rs[t] = rstd[(long)b * T + t];
    }
    __syncthreads();
    const int P = F * T;
    const float* xp = x + (long)blockIdx.x * P;
    float* yp = y + (long)blockIdx.x * P;
    const float* rp = res ? res + (long)blockIdx.x * P : nullptr;
    const float g = gain[c], bt = bias[c], sl = post_slope ? post_slope[c] : 1.f;
    auto one = [&](float v, int t, float r) {
        float o = (v - mu[t]) * rs[t] * g + bt;
        o = o >= 0.f ? o : sl * o;
        return rp ? o + r : o;
    };
    if (((((size_t)xp ^ (size_t)yp) & 15) == 0) && (!rp || ((((size_t)xp ^ (size_t)rp) & 15) == 0))) {
        const int head = min(P, (int)((4 - (((size_t)xp >> 2) & 3)) & 3));
        if (tid < head) yp[tid] = one(xp[tid], tid % T, rp ? rp[tid] : 0.f);
        const int n4 = (P - head) >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(xp + head);
        float4* y4 = reinterpret_cast<float4*>(yp + head);
        const float4* r4 = rp ? reinterpret_cast<const float4*>(rp + head) : nullptr;
        auto f4 = [&](float4 v, float4 r, int t) {        // t: frame of the first element (< T); the group may wrap into the next row
            float in[4] = {v.x, v.y, v.z, v.w}, rr[4] = {r.x, r.y, r.z, r.w}, o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int tk = t + k;
                if (tk >= T) {      // (T <= 2 - a one- or two-frame spectrogram through se_forward: the group wraps more than once)
                    tk -= T;
                    if (tk >= T) tk %= T;
                }
                o[k] = one(in[k], tk, rr[k]);
            }
            return make_float4(o[0], o[1], o[2], o[3]);
        };
        constexpr int NV = 4;
        const int step1 = 1024 % T;                    // frames one 256-thread sweep of 16 B groups advances, mod T
        int i = tid;
        int t = (head + 4 * tid) % T;
        for (; i + (NV - 1) * 256 < n4; i += NV * 256) {
            float4 v[NV], r[NV];
            int tu[NV];
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                v[u] = x4[i + u * 256];
                r[u] = r4 ? r4[i + u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
                tu[u] = t;
                t += step1;
                t = t >= T ? t - T : t;
            }
#pragma unroll
            for (int u = 0; u < NV; ++u) y4[i + u * 256] = f4(v[u], r[u], tu[u]);
        }
        for (; i < n4; i += 256) {
            y4[i] = f4(x4[i], r4 ? r4[i] : make_float4(0.f, 0.f, 0.f, 0.f), t);
            t += step1;
            t = t >= T ? t - T : t;
        }
        const int done = head + 4 * n4;
        if (done + tid < P) yp[done + tid] = one(xp[done + tid], (done + tid) % T, rp ? rp[done + tid] : 0.f);
        return;
    }
    for (int i = tid; i < P; i += 256) yp[i] = one(xp[i], i % T, rp ? rp[i] : 0.f);
}

// Frame-online windows are a few columns wide: one workgroup per utterance does the three passes in one launch (sums in
// double precision like the kernels above, the same serial scan; the order inside a column's row sum differs).
__global__ __launch_bounds__(256) void cln_window_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         const float* __restrict__ gain, const float* __restrict__ bias,
                                                         const float* __restrict__ pre_slope,
                                                         const float* __restrict__ post_slope, const float* __restrict__ fir,
                                                         int K, int R, int F, int T, int c0, int t_first, long tg0, int n_new,
                                                         double* __restrict__ carry) {
    extern __shared__ double sc[];       // [2][T] totals, then [2][T] floats mean / rstd behind them
    __shared__ double sh[2][256];
    const int b = blockIdx.x;
    float* mu = reinterpret_cast<float*>(sc + 2 * T);
    float* rs = mu + T;
    // per-column sums over the rows: WP columns at a time (WP = the window width rounded up to a power of two, at most 64),
    // 256 / WP threads share one column's rows and are folded by a tree in double precision
    int WP = 1;
    while (WP < T - c0 && WP < 64) WP <<= 1;
    const int NL = 256 / WP, col = threadIdx.x % WP, ln = threadIdx.x / WP;
    for (int tb = c0; tb < T; tb += WP) {
        const int t = tb + col;
        double s = 0.0, q = 0.0;
        if (t < T) {
            const float* xp = x + (long)b * R * T + t;
            auto take = [&](int r, float v) {
                if (pre_slope) v = v >= 0.f ? v : pre_slope[r / F] * v;
                s += v;
                q += (double)v * v;
            };
            int r = ln;
            for (; r + NL * 3 < R; r += NL * 4) {
                float xr[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xr[u] = xp[(long)(r + NL * u) * T];
#pragma unroll
                for (int u = 0; u < 4; ++u) take(r + NL * u, xr[u]);
            }
            for (; r < R; r += NL) take(r, xp[(long)r * T]);
        }
        sh[0][threadIdx.x] = s;
        sh[1][threadIdx.x] = q;
        __syncthreads();
        for (int h = NL >> 1; h >= 1; h >>= 1) {
            if (ln < h) {
                sh[0][threadIdx.x] += sh[0][threadIdx.x + h * WP];
                sh[1][threadIdx.x] += sh[1][threadIdx.x + h * WP];
            }
            __syncthreads();
        }
        if (ln == 0 && t < T) {
            const bool live = tg0 + t >= 0;
            sc[t] = live ? sh[0][col] : 0.0;
            sc[T + t] = live ? sh[1][col] : 0.0;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double a = carry[2 * b], q = carry[2 * b + 1];
        for (int t = c0; t < T; ++t) {
            a += sc[t];
            q += sc[T + t];
            sc[t] = a;
            sc[T + t] = q;
            if (t == c0 + n_new - 1) {
                carry[2 * b] = a;
                carry[2 * b + 1] = q;
            }
        }
    }
    __syncthreads();
    for (int t = c0 + threadIdx.x; t < T; t += 256) {
        const double cnt = (double)R * (double)(tg0 + t >= 0 ? tg0 + t + 1 : 1), m = sc[t] / cnt;
        const double var = (sc[T + t] - 2.0 * m * sc[t]) / cnt + m * m;
        mu[t] = (float)m;
        rs[t] = (float)(1.0 / sqrt(var + 1e-5));
    }
    __syncthreads();
    const int nn = T - t_first, W = T - c0;
    float* nw = rs + T;                  // [R][W] normalised window (FIR flavour only: the launcher sizes the LDS for it)
    if (K > 0) {
        for (long i = threadIdx.x; i < (long)R * W; i += 256) {
            const int r = (int)(i / W), ti = c0 + (int)(i - (long)r * W), c = r / F;
            float v = x[((long)b * R + r) * T + ti];
            v = v >= 0.f ? v : (pre_slope ? pre_slope[c] : 1.f) * v;
            nw[i] = (tg0 + ti >= 0) ? (v - mu[ti]) * rs[ti] * gain[c] + bias[c] : 0.f;
        }
        __syncthreads();
    }
    for (long i = threadIdx.x; i < (long)R * nn; i += 256) {
        const int r = (int)(i / nn), t = t_first + (int)(i - (long)r * nn), c = r / F;
        float o;
        if (K <= 0) {
            float v = x[((long)b * R + r) * T + t];
            v = v >= 0.f ? v : (pre_slope ? pre_slope[c] : 1.f) * v;
            o = (v - mu[t]) * rs[t] * gain[c] + bias[c];
            if (post_slope) o = o >= 0.f ? o : post_slope[c] * o;
        } else {
            o = 0.f;
            const float* wp = nw + (long)r * W + (t - c0) - (K - 1);
            for (int k = 0; k < K; ++k) {
                if (tg0 + t - (K - 1) + k >= 0) o += fir[k] * wp[k];
            }
        }
        y[((long)b * R + r) * T + t] = o;
    }
}

// One or two new frames (the latency-critical push): a thread keeps its share of the C * F * W values in registers - one
// batch of loads, column sums folded by wave shuffles and one LDS step in float64 (fixed order), the stream's running sums
// continued by one thread, normalise + PReLU straight from the registers.  cln_window_kernel walks the rows twice with a
// load per loop iteration (9 us on a 64 x 161 layer; this form: the kernel floor + one round trip).
template <int W, int VPT>
__global__ __launch_bounds__(256) void cln_window_reg_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             const float* __restrict__ gain, const float* __restrict__ bias,
                                                             const float* __restrict__ post_slope, int R, int F, int T, int c0,
                                                             long tg0, double* __restrict__ carry, const float* res) {
    __shared__ double sh[4][2 * W];
    __shared__ float s_mu[W], s_rs[W];
    __shared__ float s_par[3][256];          // gain, bias, PReLU slope per channel (C = R / F <= 256, checked by the launcher)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = R * W;
    const float* xb = x + (long)b * R * T + c0;
    float* yb = y + (long)b * R * T + c0;
    // Everything the kernel reads from global memory is requested here, in ONE batch: the frame, the running sums and the
    // channel parameters.  (Round 5 read the sums behind the block reduction and the parameters behind the second barrier: three
    // dependent round trips in a kernel whose floor is one - 6.6 us a launch, 75 launches per one-frame push of TaylorSENet_new.)
    float v[VPT], rv[VPT];
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int e = tid + 256 * u;
        const int ec = e < N ? e : 0;
        v[u] = xb[(long)(ec / W) * T + (ec % W)];
    }
    // res (optional, may be y itself: an element is read and written by one thread): the residual of a module's last layer rides
    // here instead of on an add_kernel launch behind this one (12 per one-frame push of TaylorSENet_new)
    const float* rb = res ? res + (long)b * R * T + c0 : nullptr;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int e = tid + 256 * u;
        const int ec = e < N ? e : 0;
        rv[u] = rb ? rb[(long)(ec / W) * T + (ec % W)] : 0.f;
    }
    double ca = 0.0, cq = 0.0;
    if (tid == 0) {
        ca = carry[2 * b];
        cq = carry[2 * b + 1];
    }
    {
        const int C = R / F;
        const float pg = tid < C ? gain[tid] : 0.f, pb = tid < C ? bias[tid] : 0.f;
        const float ps = (tid < C && post_slope) ? post_slope[tid] : 1.f;
        s_par[0][tid] = pg;
        s_par[1][tid] = pb;
        s_par[2][tid] = ps;
    }
    double s[W], q[W];
#pragma unroll
    for (int t = 0; t < W; ++t) s[t] = q[t] = 0.0;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int e = tid + 256 * u;
        if (e < N) {
            // (W divides 256: a thread's elements all belong to column tid % W)
#pragma unroll
            for (int t = 0; t < W; ++t)
                if (W == 1 || (tid % W) == t) {
                    s[t] += v[u];
                    q[t] += (double)v[u] * v[u];
                }
        }
    }
#pragma unroll
    for (int t = 0; t < W; ++t)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s[t] += __shfl_xor(s[t], o, 64);
            q[t] += __shfl_xor(q[t], o, 64);
        }
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < W; ++t) {
            sh[wave][2 * t] = s[t];
            sh[wave][2 * t + 1] = q[t];
        }
    }
    __syncthreads();
    if (tid == 0) {
        double a = ca, qq = cq;
#pragma unroll
        for (int t = 0; t < W; ++t) {
            a += (sh[0][2 * t] + sh[1][2 * t]) + (sh[2][2 * t] + sh[3][2 * t]);
            qq += (sh[0][2 * t + 1] + sh[1][2 * t + 1]) + (sh[2][2 * t + 1] + sh[3][2 * t + 1]);
            const double cnt = (double)R * (double)(tg0 + c0 + t + 1), m = a / cnt;
            const double var = (qq - 2.0 * m * a) / cnt + m * m;
            s_mu[t] = (float)m;
            s_rs[t] = (float)(1.0 / sqrt(var + 1e-5));
        }
        carry[2 * b] = a;
        carry[2 * b + 1] = qq;
    }
    __syncthreads();
    const bool act = post_slope != nullptr;
#pragma unroll
    for (int u = 0; u < VPT; ++u) {
        const int e = tid + 256 * u;
        if (e < N) {
            const int r = e / W, t = e % W, c = r / F;
            float o = (v[u] - s_mu[t]) * s_rs[t] * s_par[0][c] + s_par[1][c];
            if (act) o = o >= 0.f ? o : s_par[2][c] * o;
            yb[(long)r * T + t] = o + rv[u];
        }
    }
}

// frame-online chunk: launch_cln(..., res) adds the residual in its own launch (the one- / two-frame register form)
bool cln_stream_takes_res(int C, int F) {
    const StreamCtx* cx = stream_ctx();
    static const bool reg_on = !(getenv("SE_CLN_REG") && atoi(getenv("SE_CLN_REG")) == 0);
    static const bool res_on = !(getenv("SE_CLN_STREAM_RES") && atoi(getenv("SE_CLN_STREAM_RES")) == 0);
    return cx && reg_on && res_on && cx->n <= 2 && (long)C * F <= 256L * 41 && C <= 256;
}
void launch_cln(const float* x, float* y, const float* gain, const float* bias, const float* pre_slope,
                const float* post_slope, const float* fir, int K, int B, int C, int F, int T, hipStream_t s, const float* res) {
    SE_CHECK(!res || (K <= 0 && !pre_slope), "cLN: the residual rides on the plain 2-D form only");
    // per-(b, t) statistics live in a small engine-lifetime buffer (grown on first use, never on the steady-state path)
    const size_t need = (size_t)B * T * (2 * sizeof(double) + 2 * sizeof(float));
    char* stat = device_scratch(1, need, s);
    SE_CHECK(K <= 0 || x != y, "cLN + FIR cannot run in place");
    SE_CHECK((size_t)T * 16 <= 60000, "utterance too long for the LDS-resident cLN scan");
    double* sum = (double*)stat;
    double* sq = sum + (size_t)B * T;
    float* mean = (float*)(sq + (size_t)B * T);
    float* rstd = mean + (size_t)B * T;
    const int R = C * F;
    if (StreamCtx* cx = stream_ctx()) {
        // frame-online chunk: the FIR reaches K - 1 frames back into the history columns of x (brought in here), whose
        // statistics are re-accumulated from the carried totals in the order of the whole-utterance scan
        SE_CHECK(T == cx->H + cx->n && B == cx->B, "cLN: tensor is not a window of the current chunk");
        const int back = K > 0 ? K - 1 : 0, c0 = cx->H - back;
        SE_CHECK(back <= cx->H, "cLN FIR longer than the window's history");
        if (back > 0) stream_exchange(const_cast<float*>(x), (long)R * T, (long)F * T, T, B, C, F, back, s);
        cx->memo_src = nullptr;
        double* carry = static_cast<double*>(cx->slot((size_t)B * 2 * sizeof(double), s));
        const long tg0 = cx->t0 - cx->H;
        static const bool reg_on = !(getenv("SE_CLN_REG") && atoi(getenv("SE_CLN_REG")) == 0);
        if (reg_on && K <= 0 && !pre_slope && cx->n <= 2 && (long)R * cx->n <= 256L * 41 * cx->n && R <= 256 * 41 && C <= 256) {
            // (every frame of the chunk is live: tg0 + c0 = the stream index of the first new frame >= 0)
            if (cx->n == 1)
                hipLaunchKernelGGL((cln_window_reg_kernel<1, 41>), dim3(B), dim3(256), 0, s, x, y, gain, bias, post_slope, R, F, T,
                                   c0, tg0, carry, res);
            else
                hipLaunchKernelGGL((cln_window_reg_kernel<2, 82>), dim3(B), dim3(256), 0, s, x, y, gain, bias, post_slope, R, F, T,
                                   c0, tg0, carry, res);
            SE_HIP(hipGetLastError());
            return;
        }
        SE_CHECK(!res, "frame-online cLN: the residual rides on the register form only (ask cln_stream_takes_res first)");
        if ((long)R * (T - c0) <= 32768 && (K <= 0 || (size_t)R * (T - c0) * 4 + (size_t)T * 24 <= 60000)) {   // small windows: one launch
            const size_t lds = (size_t)T * 24 + (K > 0 ? (size_t)R * (T - c0) * 4 : 0);
            hipLaunchKernelGGL(cln_window_kernel, dim3(B), dim3(256), lds, s, x, y, gain, bias, pre_slope, post_slope,
                               fir, K, R, F, T, c0, cx->H, tg0, cx->n, carry);
            SE_HIP(hipGetLastError());
            return;
        }
        int WP = 1;
        while (WP < T - c0 && WP < 64) WP <<= 1;
        hipLaunchKernelGGL(cln_stats_kernel, dim3((T - c0 + WP - 1) / WP, B), dim3(256), 0, s, x, pre_slope, sum, sq, R, F, T, c0,
                           WP);
        hipLaunchKernelGGL(cln_scan_kernel, dim3(B), dim3(256), (size_t)T * 16, s, sum, sq, mean, rstd, R, T, c0, tg0, cx->n,
                           carry);
        hipLaunchKernelGGL(cln_apply_kernel, dim3((cx->n + 255) / 256, (R + 7) / 8, B), dim3(256), 0, s, x, y, mean, rstd, gain,
                           bias, pre_slope, post_slope, fir, K, R, F, T, cx->H, tg0);
        SE_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(cln_stats_kernel, dim3((T + 63) / 64, B), dim3(256), 0, s, x, pre_slope, sum, sq, R, F, T, 0, 64);
    hipLaunchKernelGGL(cln_scan_kernel, dim3(B), dim3(256), (size_t)T * 16, s, sum, sq, mean, rstd, R, T, 0, 0L, 0, nullptr);
    static const bool plane_on = !(getenv("SE_CLN_PLANE") && atoi(getenv("SE_CLN_PLANE")) == 0);
    if (K <= 0 && !pre_slope && (plane_on || res)) {
        hipLaunchKernelGGL(cln_apply_plane_kernel, dim3(B * C), dim3(256), (size_t)T * 8, s, x, y, mean, rstd, gain, bias, post_slope,
                           res, C, F, T);
        SE_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL(cln_apply_kernel, dim3((T + 255) / 256, (R + 7) / 8, B), dim3(256), 0, s, x, y, mean, rstd, gain,
                       bias, pre_slope, post_slope, fir, K, R, F, T, 0, 0L);
    SE_HIP(hipGetLastError());
}

// offline 2-D cLN + PReLU (+ residual) behind a conv that emitted its per-frame sums (GCParams::cstats): scan + apply, no statistics pass
void launch_cln_parts(const float* x, float* y, const float* gain, const float* bias, const float* post_slope, const float* parts,
                      int B, int C, int F, int T, hipStream_t s, const float* res) {
    SE_CHECK(!stream_ctx() && !ragged_ctx(), "cLN from epilogue statistics: offline equal-length batches only");
    SE_CHECK((size_t)T * 16 <= 60000, "utterance too long for the LDS-resident cLN scan");
    const size_t need = (size_t)B * T * (2 * sizeof(double) + 2 * sizeof(float));
    char* stat = device_scratch(1, need, s);
    float* mean = (float*)((double*)stat + 2 * (size_t)B * T);
    float* rstd = mean + (size_t)B * T;
    hipLaunchKernelGGL(cln_scan_parts_kernel, dim3(B), dim3(256), (size_t)T * 16, s, parts, F, mean, rstd, C * F, T);
    hipLaunchKernelGGL(cln_apply_plane_kernel, dim3(B * C), dim3(256), (size_t)T * 8, s, x, y, mean, rstd, gain, bias, post_slope, res, C,
                       F, T);
    SE_HIP(hipGetLastError());
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = a[i] + b[i];
}
void launch_add(const float* a, const float* b, float* y, long n, hipStream_t s) {
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, y, n);
    SE_HIP(hipGetLastError());
}

// ragged batch: x[b][r][t] = 0 for t >= tlen[b]  (rows r of T frames).  Operators that look a bounded number of frames
// ahead see the zeros a per-clip decode has past its end (DCCRN's decoder: one frame per transposed conv, the
// `out[..., 1:]` crop of DCCRN_cprs.py:199); only the tail is touched.
__global__ __launch_bounds__(256) void zero_tail_kernel(float* __restrict__ x, long rows, int T, const int* __restrict__ tlen) {
    const int b = blockIdx.y, Tb = tlen[b];
    const int n = T - Tb;
    if (n <= 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
        float* xp = x + ((long)b * rows + r) * T + Tb;
        for (int i = lane; i < n; i += 64) xp[i] = 0.f;
    }
}
void launch_zero_tail(float* x, int B, long rows, int T, hipStream_t s) {
    const Ragged* rg = ragged_ctx();
    if (!rg) return;
    const unsigned gx = (unsigned)std::min<long>((rows + 3) / 4, 4096);
    hipLaunchKernelGGL(zero_tail_kernel, dim3(gx, B), dim3(256), 0, s, x, rows, T, rg->tlen);
    SE_HIP(hipGetLastError());
}

// ---- running unit-RMS scale of a frame-online stream (se_stream_begin_running) ---------------------------------------
// The decode scripts scale an utterance by c = sqrt(len / sum x^2) (every *_decode_vb.py) - known only when the utterance has
// ended.  A stream that cannot wait uses what it has heard: after a push, c = sqrt(n_total / sum of squares so far) (float64
// sum, one workgroup per stream), every frame this push releases ([t0, t1)) is transformed under that c and taken back by
// it in the iSTFT (ring of 1 / c per frame).  A push that delivers the whole utterance at once therefore IS the offline decode.
__global__ __launch_bounds__(256) void stream_rms_kernel(const float* __restrict__ wav, long pitch, int n_total, int n_new,
                                                         double* __restrict__ sumsq, float* __restrict__ c,
                                                         float* __restrict__ frame_inv, int ring, int t0, int t1) {
    __shared__ double sh[256];
    __shared__ float s_inv;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* x = wav + (long)b * pitch + (n_total - n_new);
    double s = 0.0;
    for (int i = tid; i < n_new; i += 256) s += (double)x[i] * x[i];
    sh[tid] = s;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
        if (tid < h) sh[tid] += sh[tid + h];
        __syncthreads();
    }
    if (tid == 0) {
        const double tot = sumsq[b] + sh[0];
        sumsq[b] = tot;
        const float cc = tot > 1e-20 ? (float)sqrt((double)n_total / tot) : 1.f;      // (digital silence so far: no scaling)
        c[b] = cc;
        s_inv = 1.f / cc;
    }
    __syncthreads();
    for (int t = t0 + tid; t < t1; t += 256) frame_inv[(long)b * ring + (t & (ring - 1))] = s_inv;
}
void launch_stream_rms(const float* wav, long pitch, int B, int n_total, int n_new, double* sumsq, float* c, float* frame_inv,
                       int ring, int t0, int t1, hipStream_t s) {
    hipLaunchKernelGGL(stream_rms_kernel, dim3(B), dim3(256), 0, s, wav, pitch, n_total, n_new, sumsq, c, frame_inv, ring, t0, t1);
    SE_HIP(hipGetLastError());
}

// ---- streaming history columns -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hist_kernel(float* __restrict__ buf, float* __restrict__ state, long nrows, int Tw,
                                                   int hc, int save) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nrows * hc) return;
    const long r = i / hc;
    const int k = (int)(i - r * hc);
    if (save) state[i] = buf[r * Tw + (Tw - hc) + k];
    else buf[r * Tw + k] = state[i];
}
void launch_hist_restore(float* buf, const float* state, int B, long rows, int Tw, int hc, hipStream_t s) {
    const long n = (long)B * rows * hc;
    hipLaunchKernelGGL(hist_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, buf, const_cast<float*>(state),
                       (long)B * rows, Tw, hc, 0);
    SE_HIP(hipGetLastError());
}
void launch_hist_save(const float* buf, float* state, int B, long rows, int Tw, int hc, hipStream_t s) {
    const long n = (long)B * rows * hc;
    hipLaunchKernelGGL(hist_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, const_cast<float*>(buf), state,
                       (long)B * rows, Tw, hc, 1);
    SE_HIP(hipGetLastError());
}

// every history tensor of a model in one launch (blockIdx.y = tensor)
__global__ __launch_bounds__(256) void hist_batch_kernel(const HistBatch hb, int B, int Tw, int hc, int save) {
    const int e = blockIdx.y;
    const long nrows = (long)B * hb.rows[e];
    float* __restrict__ buf = hb.buf[e];
    float* __restrict__ state = hb.state[e];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nrows * hc; i += (long)gridDim.x * 256) {
        const long r = i / hc;
        const int k = (int)(i - r * hc);
        if (save) state[i] = buf[r * Tw + (Tw - hc) + k];
        else buf[r * Tw + k] = state[i];
    }
}
void launch_hist_batch(const HistBatch& hb, int B, int Tw, int hc, bool save, hipStream_t s) {
    if (hb.n <= 0) return;
    SE_CHECK(hb.n <= HistBatch::MAX, "launch_hist_batch: too many tensors");
    long most = 0;
    for (int e = 0; e < hb.n; ++e) most = std::max(most, (long)B * hb.rows[e] * hc);
    const unsigned gx = (unsigned)std::min<long>((most + 255) / 256, 256);
    hipLaunchKernelGGL(hist_batch_kernel, dim3(gx, (unsigned)hb.n), dim3(256), 0, s, hb, B, Tw, hc, save ? 1 : 0);
    SE_HIP(hipGetLastError());
}

// ---- frame-online context of the block-built models (kernels.h: StreamCtx) -------------------------------------------
static thread_local StreamCtx* g_stream_ctx = nullptr;
StreamCtx* stream_ctx() { return g_stream_ctx; }
void set_stream_ctx(StreamCtx* c) { g_stream_ctx = c; }
void* StreamCtx::slot(size_t bytes, hipStream_t st) {
    SE_CHECK(slots, "stream context without state");
    if (cursor == slots->size()) {
        void* d = nullptr;
        SE_HIP(hipMalloc(&d, bytes));
        SE_HIP(hipMemsetAsync(d, 0, bytes, st));
        slots->emplace_back(d, bytes);
    }
    SE_CHECK((*slots)[cursor].second == bytes, "frame-online state: launch order differs from the first chunk");
    return (*slots)[cursor++].first;
}
// Columns [H - need, H) of x <- state, then state <- the last `need` columns of the window [H + n - need, H + n).  For
// n < need the new state starts with the tail of the old one: a workgroup owns whole rows (KP = need rounded up to a
// power of two <= 256 threads per row), reads everything it needs, and only then writes.
__global__ __launch_bounds__(256) void stream_hist_kernel(float* __restrict__ x, float* __restrict__ state, long sb, long sc,
                                                          long sf, int C, int F, int need, int KP, int H, int n, long rows) {
    const int k = threadIdx.x % KP;
    const long r = (long)blockIdx.x * (256 / KP) + threadIdx.x / KP;
    const bool on = r < rows && k < need;
    float old = 0.f, nxt = 0.f;
    float* xp = nullptr;
    float* sp = nullptr;
    if (on) {
        const int f = (int)(r % F);
        const long q = r / F;
        const int c = (int)(q % C);
        const long b = q / C;
        xp = x + b * sb + c * sc + f * sf;
        sp = state + r * need;
        old = sp[k];
        nxt = (k + n < need) ? sp[k + n] : xp[H + n - need + k];
    }
    __syncthreads();
    if (on) {
        xp[H - need + k] = old;
        sp[k] = nxt;
    }
}
void stream_exchange(float* x, long sb, long sc, long sf, int B, int C, int F, int need, hipStream_t st) {
    StreamCtx* cx = stream_ctx();
    SE_CHECK(cx && need > 0 && need <= cx->H, "stream_exchange: history deeper than the window keeps");
    const long rows = (long)B * C * F;
    float* state = static_cast<float*>(cx->slot((size_t)rows * need * sizeof(float), st));
    SE_CHECK(need <= 256, "stream_exchange: more than 256 history columns");
    int KP = 1;
    while (KP < need) KP <<= 1;
    const int rpb = 256 / KP;
    hipLaunchKernelGGL(stream_hist_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, st, x, state, sb, sc, sf, C,
                       F, need, KP, cx->H, cx->n, rows);
    SE_HIP(hipGetLastError());
}

// Both sources of a concatenating layer (the decoders' skip inputs) in one launch: blockIdx.y picks the tensor.
struct HistPair {
    float* x[2];
    float* state[2];
    long sb[2], sc[2], sf[2], rows[2];
    int C[2], F[2];
};
__global__ __launch_bounds__(256) void stream_hist_pair_kernel(const HistPair a, int need, int KP, int H, int n) {
    const int s = blockIdx.y;
    const int k = threadIdx.x % KP;
    const long r = (long)blockIdx.x * (256 / KP) + threadIdx.x / KP;
    const bool on = r < a.rows[s] && k < need;
    float old = 0.f, nxt = 0.f;
    float* xp = nullptr;
    float* sp = nullptr;
    if (on) {
        const int f = (int)(r % a.F[s]);
        const long q = r / a.F[s];
        const int c = (int)(q % a.C[s]);
        const long b = q / a.C[s];
        xp = a.x[s] + b * a.sb[s] + c * a.sc[s] + f * a.sf[s];
        sp = a.state[s] + r * need;
        old = sp[k];
        nxt = (k + n < need) ? sp[k + n] : xp[H + n - need + k];
    }
    __syncthreads();
    if (on) {
        xp[H - need + k] = old;
        sp[k] = nxt;
    }
}
void stream_exchange_pair(float* x0, long sb0, long sc0, long sf0, int C0, int F0, float* x1, long sb1, long sc1, long sf1, int C1,
                          int F1, int B, int need, hipStream_t st) {
    static const bool pair_on = !(getenv("SE_STREAM_HIST_PAIR") && atoi(getenv("SE_STREAM_HIST_PAIR")) == 0);
    if (!pair_on) {
        stream_exchange(x0, sb0, sc0, sf0, B, C0, F0, need, st);
        stream_exchange(x1, sb1, sc1, sf1, B, C1, F1, need, st);
        return;
    }
    StreamCtx* cx = stream_ctx();
    SE_CHECK(cx && need > 0 && need <= cx->H && need <= 256, "stream_exchange: history deeper than the window keeps");
    HistPair a;
    a.x[0] = x0; a.sb[0] = sb0; a.sc[0] = sc0; a.sf[0] = sf0; a.C[0] = C0; a.F[0] = F0; a.rows[0] = (long)B * C0 * F0;
    a.x[1] = x1; a.sb[1] = sb1; a.sc[1] = sc1; a.sf[1] = sf1; a.C[1] = C1; a.F[1] = F1; a.rows[1] = (long)B * C1 * F1;
    a.state[0] = static_cast<float*>(cx->slot((size_t)a.rows[0] * need * sizeof(float), st));
    a.state[1] = static_cast<float*>(cx->slot((size_t)a.rows[1] * need * sizeof(float), st));
    int KP = 1;
    while (KP < need) KP <<= 1;
    const int rpb = 256 / KP;
    const long rmax = a.rows[0] > a.rows[1] ? a.rows[0] : a.rows[1];
    hipLaunchKernelGGL(stream_hist_pair_kernel, dim3((unsigned)((rmax + rpb - 1) / rpb), 2), dim3(256), 0, st, a, need, KP, cx->H,
                       cx->n);
    SE_HIP(hipGetLastError());
}

__global__ void fill_rows_kernel(int* d, int MB, int B, int len, int lpad, int tlen, int olen) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    d[b] = len;
    d[MB + b] = lpad;
    d[2 * MB + b] = tlen;
    d[3 * MB + b] = olen;
}
void launch_fill_rows(int* d, int MB, int B, int len, int lpad, int tlen, int olen, hipStream_t s) {
    hipLaunchKernelGGL(fill_rows_kernel, dim3((B + 255) / 256), dim3(256), 0, s, d, MB, B, len, lpad, tlen, olen);
    SE_HIP(hipGetLastError());
}

__global__ void fill_kernel(float* p, long n, float v) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
void launch_fill(float* p, long n, float v, hipStream_t s) {
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
    SE_HIP(hipGetLastError());
}

}  // namespace se
