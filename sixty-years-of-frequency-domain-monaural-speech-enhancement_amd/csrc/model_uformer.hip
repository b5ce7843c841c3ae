// Uformer on the MI355X engine.
//
// Reference: Uformer/uformer.py:30-287 (Uformer.forward :172-287) with its blocks conv2d_cplx.py, conv2d_real.py,
// fusion.py, dilated_dualpath_conformer.py:23-78, ff_cplx.py, ff_real.py, linear_cplx.py, linear_real.py,
// t_att_cplx.py, f_att_cplx.py, t_att_real.py, f_att_real.py, dsconv2d_cplx.py, dsconv2d_real.py; decode loop
// Uformer/uformer_decode_vb.py:34-62.  STFT (512/160, Hann 400) and iSTFT live INSIDE the model's forward.
//
// Engine mapping: a complex tensor [N,C,F,T,2] is stored as [B][2C][F][T] (real planes, then imaginary planes) so
//   * complex convs / deconvs / linears are real tap-table GEMMs over a 2x2 block weight, BatchNorm3d folded, scalar
//     PReLU in the epilogue, skip concatenations as two-source K loops;
//   * LayerNorm over C is one kernel on the [2B][C][F*T] view (real and imaginary parts normalised separately, as
//     `x.transpose(1,4)` does), with the following swish / PReLU / residual fused;
//   * the 24 Q/K/V projections of a complex attention are ONE GEMM (rows read the real or the imaginary half through
//     zero blocks), the 8 real attentions run in one kernel per branch with the A-B-C-D / E+F+G-H combination in
//     registers (T-branch: 401 x 401 online softmax per (b, f) with K/V tiles in LDS; F-branch: 4 x 4 per (b, t));
//   * the dilated 3x3 conv pairs use the sigmoid and gate-product epilogues, `fusion` is one elementwise kernel.
#include "rnn.h"
#include "gauss.h"

namespace se {

namespace {

constexpr int NFFT = 512, HOP = 160, WIN = 400, NBIN = 257, NL = 6, CC = 128, HD = 16, NDS = 8;
constexpr int KN[NL + 1] = {1, 8, 16, 32, 64, 128, 128};
constexpr float UEPS = 1.1920928955078125e-07f;       // torch.finfo(float32).eps (uformer.py:16)

// ---- :187-210  mag = sqrt(clamp(re^2+im^2, EPS)) [**p_in], phase = atan2(im+EPS, re); network inputs drop the DC bin
__global__ __launch_bounds__(256) void uf_prep_kernel(const float* __restrict__ spec, float* __restrict__ mag0,
                                                      float* __restrict__ ph0, float* __restrict__ xc, float* __restrict__ xm,
                                                      int T, float p_in) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const long plane = (long)NBIN * T;
    const long o = ((long)b * 2 * NBIN + k) * T + t;
    const float re = spec[o], im = spec[o + plane];
    float m = sqrtf(fmaxf(re * re + im * im, UEPS));
    if (p_in != 1.f) m = powf(m, p_in);
    const float ph = atan2f(im + UEPS, re);
    mag0[((long)b * NBIN + k) * T + t] = m;
    ph0[((long)b * NBIN + k) * T + t] = ph;
    if (k > 0) {
        const long q = ((long)b * 2 * (NBIN - 1) + (k - 1)) * T + t;
        xc[q] = m * cosf(ph);
        xc[q + (long)(NBIN - 1) * T] = m * sinf(ph);
        xm[((long)b * (NBIN - 1) + (k - 1)) * T + t] = m;
    }
}

// ---- fusion.py:13-19 on cplx [B][2C][P] / mag [B][C][P], in place
__global__ __launch_bounds__(256) void uf_fusion_kernel(float* __restrict__ cplx, float* __restrict__ mag, long CP) {
    // grid (ceil(CP / 256), B): no 64-bit division per element; hardware exp for the two sigmoids
    const long r = (long)blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (r >= CP) return;
    float* cr = cplx + b * 2 * CP + r;
    float* mp = mag + b * CP + r;
    const float re = cr[0], im = cr[CP], m = mp[0];
    const float cm = sqrtf(fmaxf(re * re + im * im, UEPS));
    const float s = (1.f / (1.f + fm_exp(-m)));
    cr[0] = re + s;
    cr[CP] = im + s;
    mp[0] = m + (1.f / (1.f + fm_exp(-cm)));
}

// ---- attention along T (t_att_cplx.py:15-40, :58-67): pq [B][nh*48][F][T] rows (q,k,v) x 16 per head.
// One block = 256 queries of one (b, f); per head K/V [16][T] go through LDS; online softmax; heads are combined with
// signs into out [B][nout*16][F][T] (complex: heads 0-3 -> real (+,-,-,-), heads 4-7 -> imag (+,+,+,-); real: 1 head).
__global__ __launch_bounds__(256) void uf_att_t_kernel(const float* __restrict__ pq, float* __restrict__ out, int F, int T,
                                                       int nh, const int* __restrict__ tlen) {
    extern __shared__ float kv[];          // K [T][16], V [T][16]
    float* Ks = kv;
    float* Vs = kv + HD * T;
    const int f = blockIdx.x % F, b = blockIdx.x / F;
    const int Tkeys = tlen ? tlen[b] : T;  // ragged batch: a clip attends to its own frames only
    const int t = blockIdx.y * 256 + threadIdx.x;
    const long P = (long)F * T;
    const float* base = pq + (long)b * nh * 48 * P + (long)f * T;
    float accr[HD], acci[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) accr[d] = acci[d] = 0.f;
    for (int h = 0; h < nh; ++h) {
        const float* hq = base + (long)h * 48 * P;
        __syncthreads();
        // K / V of the head as [T][16] in LDS (key-major): the 16 values of a key are four broadcast 16 B reads in the loop
        // below instead of sixteen 4 B ones - the loop was bound by LDS instruction issue, not by its FMAs
        for (int i = threadIdx.x; i < HD * T; i += 256) {
            const int d = i / T, s = i - d * T;
            Ks[s * HD + d] = hq[(long)(HD + d) * P + s];
            Vs[s * HD + d] = hq[(long)(2 * HD + d) * P + s];
        }
        __syncthreads();
        if (t < T) {
            float q[HD], o[HD];
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                q[d] = hq[(long)d * P + t] * 0.25f;       // / hidden_channel ** 0.5
                o[d] = 0.f;
            }
            // online softmax over chunks of 8 keys: one running-maximum update (one rescale of the 16 accumulators) per
            // chunk instead of per key, hardware exp (v_exp_f32, ~1 ulp: the weights are normalised by their own sum)
            float mx = -3.0e38f, l = 0.f;
            for (int s0 = 0; s0 < Tkeys; s0 += 8) {
                float e[8];
                float cm = -3.0e38f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int s = min(s0 + k, T - 1);
                    float a = 0.f;
                    const float4* kp = reinterpret_cast<const float4*>(Ks + s * HD);
#pragma unroll
                    for (int d4 = 0; d4 < HD / 4; ++d4) {
                        const float4 kk = kp[d4];
                        a = fmaf(q[4 * d4 + 0], kk.x, a);
                        a = fmaf(q[4 * d4 + 1], kk.y, a);
                        a = fmaf(q[4 * d4 + 2], kk.z, a);
                        a = fmaf(q[4 * d4 + 3], kk.w, a);
                    }
                    e[k] = (s0 + k < Tkeys) ? a : -3.0e38f;
                    cm = fmaxf(cm, e[k]);
                }
                const float mn = fmaxf(mx, cm);
                const float corr = fm_exp(mx - mn);
                l *= corr;
#pragma unroll
                for (int d = 0; d < HD; ++d) o[d] *= corr;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int s = min(s0 + k, T - 1);
                    const float pe = fm_exp(e[k] - mn);          // masked keys: exp(-3e38 - mn) = 0
                    l += pe;
                    const float4* vp = reinterpret_cast<const float4*>(Vs + s * HD);
#pragma unroll
                    for (int d4 = 0; d4 < HD / 4; ++d4) {
                        const float4 vv = vp[d4];
                        o[4 * d4 + 0] = fmaf(pe, vv.x, o[4 * d4 + 0]);
                        o[4 * d4 + 1] = fmaf(pe, vv.y, o[4 * d4 + 1]);
                        o[4 * d4 + 2] = fmaf(pe, vv.z, o[4 * d4 + 2]);
                        o[4 * d4 + 3] = fmaf(pe, vv.w, o[4 * d4 + 3]);
                    }
                }
                mx = mn;
            }
            const float inv = 1.f / l;
            const float sg = (nh == 1) ? 1.f : ((h == 0 || (h >= 4 && h < 7)) ? 1.f : -1.f);
            if (nh == 1 || h < 4) {
#pragma unroll
                for (int d = 0; d < HD; ++d) accr[d] += sg * o[d] * inv;
            } else {
#pragma unroll
                for (int d = 0; d < HD; ++d) acci[d] += sg * o[d] * inv;
            }
        }
    }
    if (t < T) {
        const int nout = nh == 1 ? 1 : 2;
        float* ob = out + (long)b * nout * HD * P + (long)f * T + t;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            ob[(long)d * P] = accr[d];
            if (nout == 2) ob[(long)(HD + d) * P] = acci[d];
        }
    }
}

// ---- the same attention on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 products): scores and P.V of a
// 16-query x 16-key tile are 4 + 4 MFMAs instead of 2 x 16 x 16 x 16 VALU FMAs fed by LDS broadcast reads.
//   S^T[key][query] = K[key][:] . Q[query][:]      A = K tile (lane (m = key, g) holds K[key][4j + g]),  B = Q^T
//   O^T[d][query]  += V^T[d][key] . P^T[key][query] A = V^T (lane (m = d, g) holds V[4g + j][d]),          B = P^T
// The accumulator of S^T (lane (n = query, g), register i  <->  key 4g + i) IS the B operand of the second product with
// the key order 4g + j, so the probabilities never leave their registers; the softmax statistics are per query = per
// lane column, reduced over the four 16-lane groups with two cross-lane swaps.  One wave owns QT query tiles, a block
// (4 waves) 64 * QT queries of one (b, f); K ([16][Tk], dim-major) and V ([T][17], key-major) of a head sit in LDS.
typedef float uf_x4 __attribute__((ext_vector_type(4)));
constexpr int UF_QT = 2;
__global__ __launch_bounds__(256) void uf_att_t_mfma_kernel(const float* __restrict__ pq, float* __restrict__ out, int F,
                                                            int T, int nh, int Tk, int KB, const int* __restrict__ tlen) {
    extern __shared__ float kv[];
    float* Ks = kv;                        // [16][Tk], Tk % 32 == 16: the four dim rows of an A fragment hit distinct banks
    float* Vs = kv + HD * Tk;              // [KB][17]; Tk = KB (+16): one block of KB keys is resident at a time
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int f = blockIdx.x % F, b = blockIdx.x / F;
    const int q0 = blockIdx.y * (64 * UF_QT) + wave * (16 * UF_QT);
    const long P = (long)F * T;
    const float* base = pq + (long)b * nh * 48 * P + (long)f * T;
    const int Tkeys = tlen ? tlen[b] : T;          // ragged batch: a clip attends to its own frames only
    const int nkt = (Tkeys + 15) >> 4;
    const int nks = (T + 15) >> 4;                 // key tiles staged (LDS layout is per launch, not per row)
    uf_x4 accr[UF_QT], acci[UF_QT];
#pragma unroll
    for (int qt = 0; qt < UF_QT; ++qt) accr[qt] = acci[qt] = uf_x4{0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < nh; ++h) {
        const float* hq = base + (long)h * 48 * P;
        float qf[UF_QT][4], mx[UF_QT], l[UF_QT];
        uf_x4 o[UF_QT];
#pragma unroll
        for (int qt = 0; qt < UF_QT; ++qt) {
            const int t = min(q0 + 16 * qt + n, T - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) qf[qt][j] = hq[(long)(4 * j + g) * P + t] * 0.25f;      // / hidden_channel ** 0.5
            mx[qt] = -3.0e38f;
            l[qt] = 0.f;
            o[qt] = uf_x4{0.f, 0.f, 0.f, 0.f};
        }
        // keys stream through LDS in blocks of KB (t_att_cplx.py:25 has no length limit): the online-softmax state
        // (mx, l, o) lives in registers across blocks; a clip of <= KB frames is one block, as before
        for (int kb0 = 0; kb0 < nkt * 16; kb0 += KB) {
        const int kbn = min(KB, nks * 16 - kb0);           // keys staged for this block (whole 16-key tiles)
        __syncthreads();
        for (int i = tid; i < HD * Tk; i += 256) {
            const int d = i / Tk, s = i - d * Tk;
            Ks[i] = (s < kbn && kb0 + s < T) ? hq[(long)(HD + d) * P + kb0 + s] : 0.f;
        }
        for (int i = tid; i < HD * kbn; i += 256) {
            const int d = i / kbn, s = i - d * kbn;
            Vs[s * 17 + d] = kb0 + s < T ? hq[(long)(2 * HD + d) * P + kb0 + s] : 0.f;
        }
        __syncthreads();
        const int kte = min(nkt, (kb0 + KB) >> 4);
        for (int kt = kb0 >> 4; kt < kte; ++kt) {
            const int key0 = kt * 16, kl = key0 - kb0;
            float ka[4], va[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ka[j] = Ks[(4 * j + g) * Tk + kl + n];
                va[j] = Vs[(kl + 4 * g + j) * 17 + n];
            }
#pragma unroll
            for (int qt = 0; qt < UF_QT; ++qt) {
                uf_x4 sc = uf_x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) sc = __builtin_amdgcn_mfma_f32_16x16x4f32(ka[j], qf[qt][j], sc, 0, 0, 0);
                float cm = -3.0e38f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (key0 + 4 * g + i >= Tkeys) sc[i] = -3.0e38f;
                    cm = fmaxf(cm, sc[i]);
                }
                cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
                cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
                const float mn = fmaxf(mx[qt], cm);
                const float corr = fm_exp(mx[qt] - mn);
                float ps = 0.f;
                uf_x4 pe;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pe[i] = fm_exp(sc[i] - mn);          // masked keys: exp(-3e38 - mn) = 0
                    ps += pe[i];
                }
                ps += __shfl_xor(ps, 16, 64);
                ps += __shfl_xor(ps, 32, 64);
                l[qt] = l[qt] * corr + ps;
                o[qt] *= corr;
                mx[qt] = mn;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[qt] = __builtin_amdgcn_mfma_f32_16x16x4f32(va[j], pe[j], o[qt], 0, 0, 0);
            }
        }
        }
        const float sg = (nh == 1) ? 1.f : ((h == 0 || (h >= 4 && h < 7)) ? 1.f : -1.f);
#pragma unroll
        for (int qt = 0; qt < UF_QT; ++qt) {
            const float w = sg / l[qt];
            if (nh == 1 || h < 4) accr[qt] += o[qt] * w;
            else acci[qt] += o[qt] * w;
        }
    }
    const int nout = nh == 1 ? 1 : 2;
#pragma unroll
    for (int qt = 0; qt < UF_QT; ++qt) {
        const int t = q0 + 16 * qt + n;
        if (t >= T) continue;
        float* ob = out + (long)b * nout * HD * P + (long)f * T + t;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ob[(long)(4 * g + i) * P] = accr[qt][i];
            if (nout == 2) ob[(long)(HD + 4 * g + i) * P] = acci[qt][i];
        }
    }
}

// ---- attention along F (f_att_cplx.py:13-29): one thread per (b, f_q, t)
__global__ __launch_bounds__(256) void uf_att_f_kernel(const float* __restrict__ pq, float* __restrict__ out, int F, int T,
                                                       int nh) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int fq = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const long P = (long)F * T;
    const float* base = pq + (long)b * nh * 48 * P + t;
    float accr[HD], acci[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) accr[d] = acci[d] = 0.f;
    for (int h = 0; h < nh; ++h) {
        const float* hq = base + (long)h * 48 * P;
        float q[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) q[d] = hq[(long)d * P + (long)fq * T] * 0.25f;
        float e[8];
        float mx = -3.0e38f;
        for (int g = 0; g < F; ++g) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) s += q[d] * hq[(long)(HD + d) * P + (long)g * T];
            e[g] = s;
            mx = fmaxf(mx, s);
        }
        float l = 0.f;
        for (int g = 0; g < F; ++g) {
            e[g] = fm_exp(e[g] - mx);
            l += e[g];
        }
        const float inv = 1.f / l;
        const float sg = (nh == 1) ? 1.f : ((h == 0 || (h >= 4 && h < 7)) ? 1.f : -1.f);
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            float o = 0.f;
            for (int g = 0; g < F; ++g) o += e[g] * hq[(long)(2 * HD + d) * P + (long)g * T];
            if (nh == 1 || h < 4) accr[d] += sg * o * inv;
            else acci[d] += sg * o * inv;
        }
    }
    const int nout = nh == 1 ? 1 : 2;
    float* ob = out + (long)b * nout * HD * P + (long)fq * T + t;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
        ob[(long)d * P] = accr[d];
        if (nout == 2) ob[(long)(HD + d) * P] = acci[d];
    }
}

// The same attention on the matrix cores (round 5; the north star names f_att next to t_att).  The problem is 4 x 4 per (b, t, head) -
// four frequency bins attend to four - which is exactly the block shape of v_mfma_f32_4x4x1_16b_f32: SIXTEEN independent 4 x 4 x 1
// products per instruction.  A wave owns 16 consecutive frames of one utterance (block <-> frame, lane = 4 * block + j):
//   S^T[g][fq] = sum_d k[g][d] q[fq][d]     A row g = k[g][d], B column fq = q[fq][d] / 4, 16 steps over d in 4 accumulator chains
//                                           (a dependent 4x4x1 issues at half rate, tools/mfma4bench.cpp);
//   D: lane (block, fq), register g  ->  all four scores of query fq sit in ONE lane: the softmax needs no cross-lane step, and the
//   probabilities are already the B operand (column fq, step g) of
//   O^T[d][fq] = sum_g v[g][d] p[g][fq]     A row d' = v[g][4 db + d'], four row blocks db, accumulated over the heads with the
//                                           sign of the complex product folded into p (f_att_cplx.py:31-88).
// Every q / k / v element is loaded exactly once (4 x 64 B per wave instruction).
__global__ __launch_bounds__(256) void uf_att_f_mfma_kernel(const float* __restrict__ pq, float* __restrict__ out, int T, int nh,
                                                            int ngroups) {
    constexpr int F = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = blockIdx.x * 4 + wave, b = blockIdx.y;
    if (grp >= ngroups) return;
    const int blk = lane >> 2, j = lane & 3;
    const int t = grp * 16 + blk, tc = min(t, T - 1);
    const long P = (long)F * T;
    const float* base = pq + (long)b * nh * 48 * P + (long)j * T + tc;       // row j of every (head, dim) plane at this lane's frame
    uf_x4 accr[4], acci[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) accr[db] = acci[db] = uf_x4{0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < nh; ++h) {
        const float* hq = base + (long)h * 48 * P;
        float qv[HD], kv[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            qv[d] = hq[(long)d * P] * 0.25f;                  // / hidden_channel ** 0.5
            kv[d] = hq[(long)(HD + d) * P];
        }
        uf_x4 sc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) sc[c] = uf_x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < HD; ++d) sc[d & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(kv[d], qv[d], sc[d & 3], 0, 0, 0);
        float e[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) e[g] = (sc[0][g] + sc[1][g]) + (sc[2][g] + sc[3][g]);
        const float mx = fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3]));
        float l = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            e[g] = fm_exp(e[g] - mx);
            l += e[g];
        }
        const float sg = (nh == 1) ? 1.f : ((h == 0 || (h >= 4 && h < 7)) ? 1.f : -1.f);
        const float inv = sg / l;
        const bool to_r = nh == 1 || h < 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float pg = e[g] * inv;
            const float* vg = hq + (long)(2 * HD) * P + (long)(g - j) * T;       // row g (this lane's base points at row j)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const float vv = vg[(long)(4 * db + j) * P];                     // A row d' = j of row block db: v[g][4 db + j]
                if (to_r) accr[db] = __builtin_amdgcn_mfma_f32_4x4x1f32(vv, pg, accr[db], 0, 0, 0);
                else acci[db] = __builtin_amdgcn_mfma_f32_4x4x1f32(vv, pg, acci[db], 0, 0, 0);
            }
        }
    }
    if (t >= T) return;
    const int nout = nh == 1 ? 1 : 2;
    float* ob = out + (long)b * nout * HD * P + (long)j * T + t;                 // lane (block, fq = j): every dim of query row fq
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ob[(long)(4 * db + i) * P] = accr[db][i];
            if (nout == 2) ob[(long)(HD + 4 * db + i) * P] = acci[db][i];
        }
}

// ---- :236-262  sigmoid magnitude mask, tanh complex-magnitude mask + phase add, average, polar -> RI [B][2][257][T]
__global__ __launch_bounds__(256) void uf_post_kernel(const float* __restrict__ dc, const float* __restrict__ dm,
                                                      const float* __restrict__ mag0, const float* __restrict__ ph0,
                                                      float* __restrict__ est, int T, float p_out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const long i0 = ((long)b * NBIN + k) * T + t;
    const float m0 = mag0[i0], p0 = ph0[i0];
    float mmask = 0.f, cm = 0.f, cph = 0.f;
    if (k > 0) {
        const long q = ((long)b * 2 * (NBIN - 1) + (k - 1)) * T + t;
        const float mr = dc[q], mi = dc[q + (long)(NBIN - 1) * T];
        const float mg = dm[((long)b * (NBIN - 1) + (k - 1)) * T + t];
        mmask = 1.f / (1.f + expf(-mg));
        const float mm = sqrtf(fmaxf(mr * mr + mi * mi, UEPS));
        const float rp = mr / (mm + UEPS), ip = mi / (mm + UEPS);
        cm = tanhf(mm + UEPS);
        cph = atan2f(ip + UEPS, rp);
    }
    float em = (cm * m0 + mmask * m0) * 0.5f;
    if (p_out != 1.f) em = powf(em, p_out);
    const float ep = p0 + cph;
    const long o = ((long)b * 2 * NBIN + k) * T + t;
    est[o] = em * cosf(ep);
    est[o + (long)NBIN * T] = em * sinf(ep);
}

// ---- :182-194  src_cplx = |S| e^{j angle S} of the clean source's STFT with the clamp / EPS of the reference,
// [B][2][257][T] (spec and out share the row pitch T)
__global__ __launch_bounds__(256) void uf_src_cplx_kernel(const float* __restrict__ spec, float* __restrict__ out, int T,
                                                          float p_in) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const long plane = (long)NBIN * T;
    const long o = ((long)b * 2 * NBIN + k) * T + t;
    const float re = spec[o], im = spec[o + plane];
    float m = sqrtf(fmaxf(re * re + im * im, UEPS));
    if (p_in != 1.f) m = powf(m, p_in);
    const float ph = atan2f(im + UEPS, re);
    out[o] = m * cosf(ph);
    out[o + plane] = m * sinf(ph);
}

// ------------------------------------------------------------------------------------------------ weight helpers
HostTensor dup2(const HostTensor& t) {         // BatchNorm3d(C) acts on the real and the imaginary planes alike
    HostTensor o = t;
    o.shape = {2 * t.numel()};
    o.data.insert(o.data.end(), t.data.begin(), t.data.end());
    return o;
}
DenseW clinear(const TrackedSD& sd, const std::string& p, int out, int in, float scale = 1.f) {   // Complex_Linear
    DenseW w = complex_expand(linear_weights(sd.get(p + "real_linear.weight", {out, in}), &sd.get(p + "real_linear.bias", {out})),
                              linear_weights(sd.get(p + "imag_linear.weight", {out, in}), &sd.get(p + "imag_linear.bias", {out})));
    for (auto& v : w.w) v *= scale;
    for (auto& v : w.bias) v *= scale;
    return w;
}
DenseW rlinear(const TrackedSD& sd, const std::string& p, int out, int in, float scale = 1.f) {   // Real_Linear
    DenseW w = linear_weights(sd.get(p + "linear.weight", {out, in}), &sd.get(p + "linear.bias", {out}));
    for (auto& v : w.w) v *= scale;
    for (auto& v : w.bias) v *= scale;
    return w;
}
DenseW cconv(const TrackedSD& sd, const std::string& p, int co, int ci, int kf, int kt, bool deconv) {
    std::vector<int64_t> sh = deconv ? std::vector<int64_t>{ci, co, kf, kt} : std::vector<int64_t>{co, ci, kf, kt};
    auto one = [&](const std::string& n) {
        return deconv ? deconv_weights(sd.get(p + n + ".weight", sh), &sd.get(p + n + ".bias", {co}), false)
                      : conv_weights(sd.get(p + n + ".weight", sh), &sd.get(p + n + ".bias", {co}), false);
    };
    return complex_expand(one("real_conv"), one("imag_conv"));
}
struct LnW {
    float *w = nullptr, *b = nullptr;
    void load(const TrackedSD& sd, const std::string& p) {
        w = to_device(sd.get(p + "weight").data);
        b = to_device(sd.get(p + "bias").data);
    }
    void free() {
        if (w) (void)hipFree(w);
        if (b) (void)hipFree(b);
    }
};
float* dev_scalar(const TrackedSD& sd, const std::string& key) { return to_device(sd.get(key, {1}).data); }

struct FFBlock {          // FF_Cplx / FF_Real
    LnW ln;
    GCPlan l1, l2;
    void load(const TrackedSD& sd, const std::string& p, bool cplx) {
        ln.load(sd, p + "layernorm_linear.");
        DenseW a = cplx ? clinear(sd, p + "linear1.", 64, CC) : rlinear(sd, p + "linear1.", 64, CC);
        DenseW b = cplx ? clinear(sd, p + "linear2.", CC, 64, 0.5f) : rlinear(sd, p + "linear2.", CC, 64, 0.5f);   // y*0.5 + x
        l1 = make_pointwise_plan(a, ACT_PRELU, prelu_slopes(sd.get(p + "prelu.weight"), a.M), 1604);
        l2 = make_pointwise_plan(b, ACT_NONE, {}, 1604, EPI_ADD);
    }
    void free() {
        ln.free();
        gc_free_plan(l1);
        gc_free_plan(l2);
    }
};

struct AttBlock {         // Multihead_Attention_{T,F}_Branch[_real]
    LnW ln1, ln2, ln3;
    GCPlan proj, trans;
    float* slope = nullptr;
    int nh = 8;
    void load(const TrackedSD& sd, const std::string& p, bool cplx, const char* nm) {
        nh = cplx ? 8 : 1;
        const std::string h = p + "attn_heads.0.";
        ln1.load(sd, h + "layernorm1.");
        ln2.load(sd, h + "layernorm2.");
        ln3.load(sd, p + "layernorm3.");
        slope = dev_scalar(sd, p + "prelu.weight");
        const int K = cplx ? 2 * CC : CC;
        DenseW w;
        w.M = nh * 48;
        w.Cin = K;
        w.w.assign((size_t)w.M * K, 0.f);
        w.bias.assign(w.M, 0.f);
        const char* combos[8] = {"rrr", "rii", "iri", "iir", "rri", "rir", "irr", "iii"};   // (q,k,v) sources :58-65
        const char* names[3] = {"query", "key", "value"};
        for (int n = 0; n < nh; ++n) {
            const std::string a = h + nm + (cplx ? "_att" + std::to_string(n + 1) : std::string("_att")) + ".";
            for (int j = 0; j < 3; ++j) {
                const HostTensor& lw = sd.get(a + names[j] + ".linear.weight", {HD, CC});
                const HostTensor& lb = sd.get(a + names[j] + ".linear.bias", {HD});
                const int off = (cplx && combos[n][j] == 'i') ? CC : 0;
                for (int d = 0; d < HD; ++d) {
                    const int row = n * 48 + j * HD + d;
                    for (int c = 0; c < CC; ++c) w.w[(size_t)row * K + off + c] = lw.data[d * CC + c];
                    w.bias[row] = lb.data[d];
                }
            }
        }
        proj = make_pointwise_plan(w, ACT_NONE, {}, 1604);
        DenseW t = cplx ? clinear(sd, p + "transform_linear.", CC, HD) : rlinear(sd, p + "transform_linear.", CC, HD);
        trans = make_pointwise_plan(t, ACT_NONE, {}, 1604);
    }
    void free() {
        ln1.free(); ln2.free(); ln3.free();
        gc_free_plan(proj);
        gc_free_plan(trans);
        if (slope) (void)hipFree(slope);
    }
};

struct DsBlock {          // DSConv2d / DSConv2d_Real
    LnW ln1, ln2;
    GCPlan c1, d1, d2, sc;
    void load(const TrackedSD& sd, const std::string& p, bool cplx, int dil1, int dil2) {
        ln1.load(sd, p + "layernorm_conv1.");
        ln2.load(sd, p + "layernorm_conv2.");
        auto conv = [&](const std::string& n, int co, int ci, int k) {
            if (cplx) return cconv(sd, p + n + ".", co, ci, k, k, false);
            return conv_weights(sd.get(p + n + ".conv.weight", {co, ci, k, k}), &sd.get(p + n + ".conv.bias", {co}), false);
        };
        DenseW a = conv("conv1x1", 32, CC, 1);
        c1 = make_conv_plan(a, 1, 0, 0, 1, 1, ACT_PRELU, prelu_slopes(sd.get(p + "prelu.weight"), a.M), EPI_ACT, 401);
        d2 = make_conv_plan(conv("dconv2", 32, 32, 3), 1, 1, dil2, 1, dil2, ACT_SIGMOID, {}, EPI_ACT, 401);
        d1 = make_conv_plan(conv("dconv1", 32, 32, 3), 1, 1, dil1, 1, dil1, ACT_NONE, {}, EPI_MUL, 401);
        sc = make_conv_plan(conv("sconv", CC, 32, 1), 1, 0, 0, 1, 1, ACT_NONE, {}, EPI_ADD, 401);
    }
    void free() {
        ln1.free(); ln2.free();
        for (GCPlan* g : {&c1, &d1, &d2, &sc}) gc_free_plan(*g);
    }
};

class Uformer final : public Model {
  public:
    explicit Uformer(EngineCtx& c) : Model(c) {}
    int frame_multiple() const override { return 16; }
    ~Uformer() override {
        for (int k = 0; k < NL; ++k) {
            gc_free_plan(encC[k]);
            gc_free_plan(encR[k]);
            free_deconv_plan(decC[k]);
            free_deconv_plan(decR[k]);
        }
        for (int j = 0; j < 2; ++j) {
            ffC[j].free(); ffR[j].free(); attC[j].free(); attR[j].free();
        }
        for (int k = 0; k < NDS; ++k) {
            dsC[k].free();
            dsR[k].free();
        }
        lnC.free();
        lnR.free();
        for (auto& g : genc) g.free();
        for (auto& g : gdec) g.free();
    }
    StftGeom default_geom() const override { return StftGeom{NFFT, HOP, WIN}; }
    int64_t output_samples(int L) const override { return (int64_t)HOP * (L / HOP); }   // istft without length (:276)

    void finalize(const TrackedSD& sd) override {
        for (int k = 0; k < NL; ++k) {        // :49-83  (5,2) stride (2,1) pad (2,1) then [..., :T]  -> causal in time
            const std::string p = "encoder." + std::to_string(k) + ".";
            DenseW w = cconv(sd, p + "0.", KN[k + 1], KN[k], 5, 2, false);
            fold_bn(w, dup2(sd.get(p + "1.weight")), dup2(sd.get(p + "1.bias")), dup2(sd.get(p + "1.running_mean")),
                    dup2(sd.get(p + "1.running_var")));
            encC[k] = make_conv_plan(w, 2, 2, 1, 1, 1, ACT_PRELU, prelu_slopes(sd.get(p + "2.weight"), w.M), EPI_ACT, 401);
            const std::string q = "encoder_real." + std::to_string(k) + ".";
            DenseW r = conv_weights(sd.get(q + "0.conv.weight", {KN[k + 1], KN[k], 5, 2}), &sd.get(q + "0.conv.bias", {KN[k + 1]}), false);
            fold_bn(r, sd.get(q + "1.weight"), sd.get(q + "1.bias"), sd.get(q + "1.running_mean"), sd.get(q + "1.running_var"));
            encR[k] = make_conv_plan(r, 2, 2, 1, 1, 1, ACT_PRELU, prelu_slopes(sd.get(q + "2.weight"), r.M), EPI_ACT, 401);
        }
        for (int k = 0; k < NL; ++k) {        // :90-158  ConvTranspose2d (5,2) stride (2,1) pad (2,0) out_pad (1,0), [..., :T]
            const int idx = NL - k, ci = KN[idx], co = KN[idx - 1];
            const std::string p = "decoder." + std::to_string(k) + ".";
            DenseW w = cconv(sd, p + "0.", co, 2 * ci, 5, 2, true);
            // reference channel order of the cat([skip, out]) per part: [skip_r, out_r | skip_i, out_i];
            // engine two-source order: [skip_r, skip_i | out_r, out_i]
            std::vector<int> perm(4 * ci);
            for (int c = 0; c < ci; ++c) {
                perm[c] = c;
                perm[ci + c] = 2 * ci + c;
                perm[2 * ci + c] = ci + c;
                perm[3 * ci + c] = 3 * ci + c;
            }
            permute_cin(w, perm);
            const std::string q = "decoder_real." + std::to_string(k) + ".";
            DenseW r = deconv_weights(sd.get(q + "0.conv.weight", {2 * ci, co, 5, 2}), &sd.get(q + "0.conv.bias", {co}), false);
            std::vector<float> sc, sr;
            int act = ACT_NONE;
            if (k < NL - 1) {
                fold_bn(w, dup2(sd.get(p + "1.weight")), dup2(sd.get(p + "1.bias")), dup2(sd.get(p + "1.running_mean")),
                        dup2(sd.get(p + "1.running_var")));
                fold_bn(r, sd.get(q + "1.weight"), sd.get(q + "1.bias"), sd.get(q + "1.running_mean"), sd.get(q + "1.running_var"));
                sc = prelu_slopes(sd.get(p + "2.weight"), w.M);
                sr = prelu_slopes(sd.get(q + "2.weight"), r.M);
                act = ACT_PRELU;
            }
            decC[k] = make_deconv_plan(w, 2, 2, 0, act, sc, 401, 2 * ci);
            decR[k] = make_deconv_plan(r, 2, 2, 0, act, sr, 401, ci);
        }
        gauss_on = !(getenv("SE_UF_GAUSS") && atoi(getenv("SE_UF_GAUSS")) == 0);
        if (gauss_on) {
            // BatchNorm3d(C) (both parts alike) + the conv biases (real part b_r - b_i, imaginary b_r + b_i) + PReLU of a layer
            auto tail = [&](gauss::GaussLayer& g, const std::string& p, const DenseW& wr, const DenseW& wi) {
                const int co = wr.M;
                const HostTensor &ga = sd.get(p + "1.weight", {co}), &be = sd.get(p + "1.bias", {co}), &mu = sd.get(p + "1.running_mean", {co}),
                                 &va = sd.get(p + "1.running_var", {co});
                std::vector<float> sc(2 * co), sh(2 * co);
                for (int m = 0; m < 2 * co; ++m) {
                    const int c = m % co;
                    const float bias = m < co ? wr.bias[c] - wi.bias[c] : wr.bias[c] + wi.bias[c];
                    const double k = (double)ga.data[c] / std::sqrt((double)va.data[c] + 1e-5);
                    sc[m] = (float)k;
                    sh[m] = (float)((double)be.data[c] - (double)mu.data[c] * k + (double)bias * k);
                }
                g.sc = to_device(sc);
                g.sh = to_device(sh);
                g.slope = to_device(prelu_slopes(sd.get(p + "2.weight"), 2 * co));
            };
            for (int j = 0; j < 2; ++j) {
                const int k = 4 + j, ci = KN[k], co = KN[k + 1];
                const std::string p = "encoder." + std::to_string(k) + ".";
                DenseW wr = conv_weights(sd.get(p + "0.real_conv.weight", {co, ci, 5, 2}), &sd.get(p + "0.real_conv.bias", {co}), false);
                DenseW wi = conv_weights(sd.get(p + "0.imag_conv.weight", {co, ci, 5, 2}), &sd.get(p + "0.imag_conv.bias", {co}), false);
                gauss::make_conv_plans(genc[j], wr, wi, 401);
                tail(genc[j], p, wr, wi);
            }
            for (int k = 0; k < 2; ++k) {
                const int idx = NL - k, ci = KN[idx], co = KN[idx - 1];
                const std::string p = "decoder." + std::to_string(k) + ".";
                // input channels per part in the reference's cat order [skip (ci) | out (ci)] = (first source | second source)
                DenseW wr = deconv_weights(sd.get(p + "0.real_conv.weight", {2 * ci, co, 5, 2}), &sd.get(p + "0.real_conv.bias", {co}), false);
                DenseW wi = deconv_weights(sd.get(p + "0.imag_conv.weight", {2 * ci, co, 5, 2}), &sd.get(p + "0.imag_conv.bias", {co}), false);
                gauss::make_deconv_plans(gdec[k], wr, wi, 0, ci, 401);
                tail(gdec[k], p, wr, wi);
            }
        }
        const std::string c = "conformer.";
        ffC[0].load(sd, c + "ff1_cplx.", true);
        ffR[0].load(sd, c + "ff1_mag.", false);
        ffC[1].load(sd, c + "ff2_cplx.", true);
        ffR[1].load(sd, c + "ff2_mag.", false);
        attC[0].load(sd, c + "cplx_tatt.", true, "T");
        attR[0].load(sd, c + "mag_tatt.", false, "T");
        attC[1].load(sd, c + "cplx_fatt.", true, "F");
        attR[1].load(sd, c + "mag_fatt.", false, "F");
        const int dil[NDS] = {1, 2, 4, 8, 16, 32, 64, 128};
        for (int k = 0; k < NDS; ++k) {
            dsC[k].load(sd, c + "dsconv_cplx." + std::to_string(k) + ".", true, dil[k], dil[NDS - 1 - k]);
            dsR[k].load(sd, c + "dsconv_real." + std::to_string(k) + ".", false, dil[k], dil[NDS - 1 - k]);
        }
        lnC.load(sd, c + "ln_conformer_cplx.");
        lnR.load(sd, c + "ln_conformer_mag.");
    }

    void plan_buffers(int B, int T) override {
        cur.B = 0;
        bufs(B, T);
    }

    // model(wav, wav)[0]: [B, L] waveform -> [B, 160*floor(L/160)]  (the STFT / iSTFT are inside the model)
    void forward(const float* in, const int64_t* shape, int ndim, float* out, hipStream_t st) override {
        SE_CHECK(ndim == 2, "Uformer forward expects waveforms [B, L]");
        run(in, shape[1], (int)shape[0], (int)shape[1], out, (long)output_samples((int)shape[1]), false, st);
    }
    void enhance(const float* wav, long pitch, int B, int L, float* out, long out_pitch, hipStream_t st) override {
        run(wav, pitch, B, L, out, out_pitch, true, st);                  // uformer_decode_vb.py:35-36,62
    }
    // output, src, output_cplx, src_cplx = model(inputs, src)   (uformer.py:172-287)
    void forward_uformer(const float* inputs, const float* src, int B, int L, float* output, float* src_out, float* output_cplx,
                         float* src_cplx, hipStream_t st) override {
        const long olen = output_samples(L);
        run(inputs, L, B, L, output, olen, false, st, output_cplx);
        if (!src || (!src_out && !src_cplx)) return;
        // :182-194: the clean source only goes through the front end - STFT, waveform round trip, polar re-synthesis
        const int T = 1 + L / HOP;
        Bufs& b = bufs(cur.B, cur.T);                 // the buffers of the run above (its spectra are consumed by now)
        launch_stft(ctx.geom, src, L, B, L, L, nullptr, 1.f, b.spec, nullptr, T, T, st);
        if (src_out) launch_istft(ctx.geom, b.spec, B, T, T, b.frames, nullptr, src_out, olen, HOP * (T - 1), st);
        if (src_cplx) {
            hipLaunchKernelGGL(uf_src_cplx_kernel, dim3((T + 255) / 256, NBIN, B), dim3(256), 0, st, b.spec, src_cplx, T, ctx.p_in);
            SE_HIP(hipGetLastError());
        }
    }

  private:
    struct Bufs {
        int B = 0, T = 0;
        float *c, *spec, *est, *frames, *mag0, *ph0, *xc, *xm;
        float *EC[NL], *ER[NL], *DC[NL], *DR[NL];
        float *XC[2], *XR[2], *t1, *t2, *t3, *pq;
        float *EC5g, *XCg, *K;      // three-product layers: three-plane copies of the encoder's last output / the conformer's output, k1..k3
    } cur;
    // encoder layers 4 - 5 and decoder layers 0 - 1 (>= 64 complex output channels at 128 / 256 complex inputs) as Gauss' three real
    // products (gauss.h): EC[3], EC[4], DC[0] are then three-plane tensors [S | R | I] (SE_UF_GAUSS=0: block GEMMs)
    gauss::GaussLayer genc[2], gdec[2];
    bool gauss_on = false;
    GCPlan encC[NL], encR[NL];
    DeconvPlan decC[NL], decR[NL];
    FFBlock ffC[2], ffR[2];
    AttBlock attC[2], attR[2];
    DsBlock dsC[NDS], dsR[NDS];
    LnW lnC, lnR;

    Bufs& bufs(int B, int T) {
        if (cur.B == B && cur.T == T) return cur;
        Arena& a = ctx.arena;
        a.reset();
        Bufs b;
        b.B = B;
        b.T = T;
        const size_t BT = (size_t)B * T;
        b.c = a.alloc_f(B);
        b.spec = a.alloc_f(BT * 2 * NBIN);
        b.est = a.alloc_f(BT * 2 * NBIN);
        b.frames = nullptr;      // the fused iSTFT keeps its frames in LDS (k_stft.hip); kept in the struct for the launcher signature
        b.mag0 = a.alloc_f(BT * NBIN);
        b.ph0 = a.alloc_f(BT * NBIN);
        b.xc = a.alloc_f(BT * 2 * 256);
        b.xm = a.alloc_f(BT * 256);
        int F = 256;
        for (int k = 0; k < NL; ++k) {
            F /= 2;
            b.EC[k] = a.alloc_f(BT * ((gauss_on && (k == 3 || k == 4)) ? 3 : 2) * KN[k + 1] * F);
            b.ER[k] = a.alloc_f(BT * KN[k + 1] * F);
        }
        F = 4;
        for (int k = 0; k < NL; ++k) {
            F *= 2;
            b.DC[k] = a.alloc_f(BT * ((gauss_on && k == 0) ? 3 : 2) * KN[NL - k - 1] * F);
            b.DR[k] = a.alloc_f(BT * KN[NL - k - 1] * F);
        }
        const size_t P = BT * 4;
        for (int j = 0; j < 2; ++j) {
            b.XC[j] = a.alloc_f(P * 2 * CC);
            b.XR[j] = a.alloc_f(P * CC);
        }
        b.EC5g = b.XCg = b.K = nullptr;
        if (gauss_on) {
            b.EC5g = a.alloc_f(P * 3 * CC);
            b.XCg = a.alloc_f(P * 3 * CC);
            b.K = a.alloc_f(BT * 3 * 1024);      // k1..k3 of the widest layer: 128 channels x 8 rows (64 x 16)
        }
        b.t1 = a.alloc_f(P * 2 * CC);
        b.t2 = a.alloc_f(P * 2 * CC);
        b.t3 = a.alloc_f(P * 2 * CC);
        b.pq = a.alloc_f(P * 8 * 48);
        cur = b;
        return cur;
    }

    // pointwise GEMM on a [B][C][P] tensor
    void pw(const GCPlan& pl, const float* x, int Cin, float* y, int Cout, const float* res, int B, long P, hipStream_t st,
            float* fz = nullptr) {
        GCParams p = pl.p;
        if (fz) { p.fz = fz; p.fz_im = Cout * P; p.fz_b = 2 * Cout * P; p.fz_c = P; p.fz_f = 0; }
        p.src0 = x; p.s0_b = Cin * P; p.s0_c = P; p.s0_f = 0; p.src1 = nullptr;
        p.Fin = 1; p.Tin = (int)P; p.B = B; p.Q = 1; p.Tout = (int)P;
        p.dst = y; p.d_b = Cout * P; p.d_c = P; p.d_f = 0;
        if (res) { p.aux = res; p.x_b = Cout * P; p.x_c = P; p.x_f = 0; }
        gc_launch_prof(pl, p, st, &ctx.prof);
    }
    // SE_UF_FOLD=0: the interaction of the two branches (fusion.py:13-19) as its own launch after every layer pair instead of
    // inside the store of the magnitude branch's last launch (GCParams::fz)
    static bool fold_env() {
        static const bool v = !(getenv("SE_UF_FOLD") && atoi(getenv("SE_UF_FOLD")) == 0);
        return v;
    }
    void fusion(float* c, float* m, int B, long CP, hipStream_t st) {
        hipLaunchKernelGGL(uf_fusion_kernel, dim3((unsigned)((CP + 255) / 256), B), dim3(256), 0, st, c, m, CP);
    }
    // ---- three-product layers (gauss.h).  A three-plane tensor [B][3 C][F][T]: S = R + I at +0, R at + C F T, I at + 2 C F T
    static Act4 view3(const float* t3, int C, int F, int T) {      // its [R | I] planes as a 2 C-channel tensor
        return Act4{t3 + (long)C * F * T, 2 * C, F, 3L * C * F * T, (long)F * T, (long)T};
    }
    void gauss_sum(float* t3, int B, long CP, hipStream_t st) {
        Profiler* pf = &ctx.prof;
        const bool timed = pf->on;
        if (timed) pf->begin(st);
        hipLaunchKernelGGL(gauss::gauss_sum_kernel, dim3((unsigned)((CP / 4 + 255) / 256 + 1), B), dim3(256), 0, st, t3, CP);
        SE_HIP(hipGetLastError());
        if (timed) pf->end(st, 0.0);
    }
    void gauss_planes23(const float* x2, float* x3, int B, long CP, hipStream_t st) {
        Profiler* pf = &ctx.prof;
        const bool timed = pf->on;
        if (timed) pf->begin(st);
        hipLaunchKernelGGL(gauss::gauss_planes23_kernel, dim3((unsigned)((CP / 4 + 255) / 256 + 1), B), dim3(256), 0, st, x2, x3, CP);
        SE_HIP(hipGetLastError());
        if (timed) pf->end(st, 0.0);
    }
    // y = PReLU(BN(complex (de)conv(x))): the grouped launch(es) into b.K, then the combine pass.  src0 / src1: three-plane tensors of
    // C0 / C1 complex channels; dst3: three-plane output (else [R | I])
    void gauss_layer(const gauss::GaussLayer& g, Bufs& b, const float* src0, int C0, const float* src1, int C1, int Fin, int Fout, int T,
                     float* dst, bool dst3, hipStream_t st) {
        const int B = b.B, co = g.co;
        // SE_GAUSS_CMB=0: three products into scratch + the combine pass (round 4).  Rows of whole 16 B groups only (the combine
        // epilogue has no trimming variant; PadFrames gives every offline decode such rows)
        static const bool cmb_env = !(getenv("SE_GAUSS_CMB") && atoi(getenv("SE_GAUSS_CMB")) == 0);
        const bool cmb = cmb_env && T % 4 == 0 && co >= 64;
        Profiler* pf = &ctx.prof;
        const long kz = (long)B * co * Fout * T;
        const Ragged* rg = ragged_ctx();
        for (const GCPlan& pl : g.pl) {
            GCParams p = pl.p;
            p.src0 = src0; p.C0 = C0; p.s0_b = 3L * C0 * Fin * T; p.s0_c = (long)Fin * T; p.s0_f = T; p.src0_z = (long)C0 * Fin * T;
            if (src1) {
                p.src1 = src1; p.C1 = C1; p.s1_b = 3L * C1 * Fin * T; p.s1_c = (long)Fin * T; p.s1_f = T; p.src1_z = (long)C1 * Fin * T;
            } else {
                p.src1 = nullptr; p.C1 = 0;
            }
            p.Fin = Fin; p.Tin = T; p.B = B; p.Tout = T;
            p.Q = (Fout - p.po + p.so - 1) / p.so;
            p.dst = b.K; p.d_b = (long)co * Fout * T; p.d_c = (long)Fout * T; p.d_f = T; p.dst_z = kz;
            if (rg) p.tlen = rg->tlen;
            if (cmb) {
                // k1 = Wr (xr + xi) alone, then k2 / k3 as a grouped launch of two whose epilogue (EPI_CMB) reads k1 and stores the
                // finished planes: I = f(k1 + k2), R = f(k1 - k3) - no k2 / k3 scratch, no combine pass (round 4: 5 % of a step)
                GCParams p1 = p;
                p1.Z = 1;
                p1.tlen = nullptr;
                gc_launch_prof(pl, p1, st, pf);
                GCParams q = p;
                q.Z = 2;
                q.A = pl.p.A + pl.p.A_z;
                q.src0 = p.src0 + p.src0_z;
                if (src1) q.src1 = p.src1 + p.src1_z;
                q.epi = EPI_CMB;
                q.bias = nullptr;
                q.aux = b.K; q.x_b = p.d_b; q.x_c = p.d_c; q.x_f = p.d_f; q.aux_z = 0;
                q.post_scale = g.sc + co; q.post_shift = g.sh + co; q.slope = g.slope + co; q.ps_z = -co;
                q.cmb_neg = 2;                                   // z = 0: I = f(k1 + k2); z = 1: R = f(k1 - k3)
                const long CPo = (long)co * Fout * T, oR = dst3 ? CPo : 0L, oI = dst3 ? 2 * CPo : CPo;
                q.dst = dst + oI; q.dst_z = oR - oI; q.d_b = (dst3 ? 3 : 2) * CPo;
                gc_launch_prof(pl, q, st, pf);
                continue;
            }
            gc_launch_prof(pl, p, st, pf);
        }
        const long CP = (long)co * Fout * T;
        if (cmb) return;      // (the sum plane of a three-plane output is refreshed by the caller, behind the folded interaction)
        const bool timed = pf->on;
        if (timed) pf->begin(st);
        hipLaunchKernelGGL(gauss::gauss_combine_kernel, dim3(Fout, co, B), dim3(128), 0, st, b.K, dst, co, Fout, T, kz, dst3 ? 3 * CP : 2 * CP,
                           -1L, dst3 ? CP : 0L, dst3 ? 2 * CP : CP, g.sc, g.sh, g.slope, rg ? rg->tlen : nullptr);
        SE_HIP(hipGetLastError());
        if (timed) pf->end(st, 0.0);
    }
    // LayerNorm over C of a [Bv][C][P] view
    void ln(const LnW& w, const float* x, float* y, int Bv, int C, long P, hipStream_t st, int post = 0, const float* slope = nullptr,
            const float* res = nullptr) {
        launch_layernorm_cf(x, res, w.w, w.b, y, Bv, C, 1, (int)P, 1e-5f, st, post, slope);
    }

    // fz (magnitude branch only): the complex branch's output of the same step - the interaction goes into the last store
    void ff(const FFBlock& f, bool cplx, const float* x, float* y, Bufs& b, long P, hipStream_t st, float* fz = nullptr) {
        const int m = cplx ? 2 : 1;
        ln(f.ln, x, b.t1, m * b.B, CC, P, st);
        pw(f.l1, b.t1, m * CC, b.t2, m * 64, nullptr, b.B, P, st);
        pw(f.l2, b.t2, m * 64, y, m * CC, x, b.B, P, st, fz);
    }
    void att(const AttBlock& a, bool cplx, bool along_t, const float* x, float* y, Bufs& b, int F, int T, hipStream_t st) {
        const int m = cplx ? 2 : 1, B = b.B;
        const long P = (long)F * T;
        ln(a.ln1, x, b.t1, m * B, CC, P, st);
        pw(a.proj, b.t1, m * CC, b.pq, a.nh * 48, nullptr, B, P, st);
        if (along_t) {
            // SE_UF_ATT_MFMA=0: the round-1 VALU kernel (kept for the A/B measurement in profiles/)
            static const bool mfma = !(getenv("SE_UF_ATT_MFMA") && atoi(getenv("SE_UF_ATT_MFMA")) == 0);
            if (mfma) {
                // key blocks of <= 512 frames (68 KB of K / V per workgroup: two workgroups per CU at any clip length)
                static const int kbmax = getenv("SE_UF_ATT_KB") ? std::max(16, atoi(getenv("SE_UF_ATT_KB")) / 16 * 16) : 512;
                const int KB = std::min((T + 15) / 16 * 16, kbmax);
                int Tk = KB;
                if (Tk % 32 != 16) Tk += 16;
                const size_t lds = ((size_t)HD * Tk + (size_t)KB * 17) * sizeof(float);
                SE_CHECK(lds <= 150 * 1024, "SE_UF_ATT_KB too large for the LDS-resident T-attention K/V block");
                static bool seen[64] = {};
                if (first_on_device(seen))
                    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(uf_att_t_mfma_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                hipLaunchKernelGGL(uf_att_t_mfma_kernel, dim3(B * F, (T + 64 * UF_QT - 1) / (64 * UF_QT)), dim3(256), lds, st,
                                   b.pq, b.t2, F, T, a.nh, Tk, KB, ragged_ctx() ? ragged_ctx()->tlen : nullptr);
            } else {
                const size_t lds = (size_t)2 * HD * T * sizeof(float);
                SE_CHECK(lds <= 64 * 1024, "utterance too long for the LDS-resident T-attention K/V tiles");
                hipLaunchKernelGGL(uf_att_t_kernel, dim3(B * F, (T + 255) / 256), dim3(256), lds, st, b.pq, b.t2, F, T, a.nh,
                                   ragged_ctx() ? ragged_ctx()->tlen : nullptr);
            }
        } else {
            SE_CHECK(F <= 8, "F-attention kernel is built for the 4-bin bottleneck");
            // SE_UF_ATT_F_MFMA=0: the VALU kernel of rounds 1-4 (one thread per query)
            static const bool fmfma = !(getenv("SE_UF_ATT_F_MFMA") && atoi(getenv("SE_UF_ATT_F_MFMA")) == 0);
            if (fmfma && F == 4) {
                const int ng = (T + 15) / 16;
                hipLaunchKernelGGL(uf_att_f_mfma_kernel, dim3((ng + 3) / 4, B), dim3(256), 0, st, b.pq, b.t2, T, a.nh, ng);
            } else {
                hipLaunchKernelGGL(uf_att_f_kernel, dim3((T + 255) / 256, F, B), dim3(256), 0, st, b.pq, b.t2, F, T, a.nh);
            }
        }
        ln(a.ln2, b.t2, b.t1, m * B, HD, P, st);
        pw(a.trans, b.t1, m * HD, b.t3, m * CC, nullptr, B, P, st);
        ln(a.ln3, b.t3, y, m * B, CC, P, st, 0, a.slope, x);
    }
    void ds(const DsBlock& d, bool cplx, const float* x, float* y, Bufs& b, int F, int T, hipStream_t st, float* fz = nullptr) {
        const int m = cplx ? 2 : 1, B = b.B;
        const long P = (long)F * T;
        Profiler* pf = &ctx.prof;
        ln(d.ln1, x, b.t1, m * B, CC, P, st);
        run_conv(d.c1, act4(b.t1, m * CC, F, T), nullptr, b.t2, m * 32, F, B, T, T, st, pf);
        // ragged batch: the two dilated convs pad symmetrically in time (dsconv2d_cplx.py:29-36) - a clip decoded alone has
        // zeros past its last frame there
        launch_zero_tail(b.t2, B, (long)m * 32 * F, T, st);
        run_conv(d.d2, act4(b.t2, m * 32, F, T), nullptr, b.t3, m * 32, F, B, T, T, st, pf);
        {
            GCParams p = d.d1.p;
            Act4 a = act4(b.t2, m * 32, F, T);
            p.src0 = a.p; p.s0_b = a.sb; p.s0_c = a.sc; p.s0_f = a.sf; p.src1 = nullptr;
            p.Fin = F; p.Tin = T; p.B = B; p.Q = F; p.Tout = T;
            p.dst = b.t1; p.d_b = (long)m * 32 * P; p.d_c = P; p.d_f = T;
            p.aux = b.t3; p.x_b = (long)m * 32 * P; p.x_c = P; p.x_f = T;
            gc_launch_prof(d.d1, p, st, pf);
        }
        ln(d.ln2, b.t1, b.t2, m * B, 32, P, st, 1);
        {
            GCParams p = d.sc.p;
            Act4 a = act4(b.t2, m * 32, F, T);
            p.src0 = a.p; p.s0_b = a.sb; p.s0_c = a.sc; p.s0_f = a.sf; p.src1 = nullptr;
            p.Fin = F; p.Tin = T; p.B = B; p.Q = F; p.Tout = T;
            p.dst = y; p.d_b = (long)m * CC * P; p.d_c = P; p.d_f = T;
            p.aux = x; p.x_b = (long)m * CC * P; p.x_c = P; p.x_f = T;
            if (fz) { p.fz = fz; p.fz_im = (long)CC * P; p.fz_b = 2L * CC * P; p.fz_c = P; p.fz_f = T; }
            gc_launch_prof(d.sc, p, st, pf);
        }
    }

    void run(const float* wav, long pitch, int B, int L, float* out, long out_pitch, bool normalise, hipStream_t st,
             float* out_cplx = nullptr) {
        const int T_true = 1 + L / HOP;
        PadFrames pad(ctx, B, L, L, T_true, HOP * (T_true - 1), st, 16);      // symmetric dilated convs: rows of whole 16 B groups (and, at 16, whole 128 B lines for T = 401)
        const int T = pad.T;
        Bufs& b = bufs(B, T);
        Profiler* pf = &ctx.prof;
        if (normalise) launch_rms_scale(wav, B, L, pitch, b.c, st);
        const float* cs = normalise ? b.c : nullptr;
        launch_stft(ctx.geom, wav, pitch, B, L, L, cs, 1.f, b.spec, nullptr, T, T, st);          // uformer.py:178
        hipLaunchKernelGGL(uf_prep_kernel, dim3((T + 255) / 256, NBIN, B), dim3(256), 0, st, b.spec, b.mag0, b.ph0, b.xc, b.xm, T,
                           ctx.p_in);
        // ---- encoder (:214-219)
        Act4 xc = act4(b.xc, 2, 256, T), xm = act4(b.xm, 1, 256, T);
        int F = 256;
        for (int k = 0; k < NL; ++k) {
            F /= 2;
            const int co = KN[k + 1];
            const long CPk = (long)co * F * T;
            // three-plane outputs (inputs of the three-product layers 4 / 5 and skips of decoder layers 1 / 2): EC[3], EC[4]
            const bool out3 = gauss_on && (k == 3 || k == 4);
            float* ecR = out3 ? b.EC[k] + CPk : b.EC[k];          // the [R | I] planes
            if (gauss_on && k >= 4) gauss_layer(genc[k - 4], b, b.EC[k - 1], KN[k], nullptr, 0, 2 * F, F, T, b.EC[k], out3, st);
            else run_conv(encC[k], xc, nullptr, ecR, (out3 ? 3 : 2) * co, F, B, T, T, st, pf);      // (dstC only sets the batch stride)
            const bool fold = fold_env() && conv_folds_interaction(encR[k]);
            run_conv(encR[k], xm, nullptr, b.ER[k], co, F, B, T, T, st, pf, nullptr, 0, fold ? ecR : nullptr, out3 ? 3 : 2);
            if (!fold) {
                SE_CHECK(!out3, "Uformer: three-plane encoder tensors need the folded interaction (SE_UF_FOLD=0 with SE_UF_GAUSS=1)");
                fusion(b.EC[k], b.ER[k], B, CPk, st);
            }
            if (out3 && !(fold && conv_fold_writes_sum())) gauss_sum(b.EC[k], B, CPk, st);      // S = R + I of the tensor the interaction has just rewritten (else: stored by the fold's epilogue)
            xc = out3 ? view3(b.EC[k], co, F, T) : act4(b.EC[k], 2 * co, F, T);
            xm = act4(b.ER[k], co, F, T);
        }
        if (gauss_on) gauss_planes23(b.EC[NL - 1], b.EC5g, B, (long)CC * 4 * T, st);      // skip of decoder layer 0
        // ---- dilated dual-path conformer at [B][128][4][T] (dilated_dualpath_conformer.py:53-78)
        const long P = 4L * T, CP = (long)CC * P;
        const float *c = b.EC[NL - 1], *m = b.ER[NL - 1];
        int pp = 0;
        auto step = [&](auto&& fc, auto&& fr, bool fold = false) {
            fc(c, b.XC[pp]);
            fr(m, b.XR[pp]);
            if (!fold) fusion(b.XC[pp], b.XR[pp], B, CP, st);
            c = b.XC[pp];
            m = b.XR[pp];
            pp ^= 1;
        };
        const bool fold_pw = fold_env() && conv_folds_interaction(ffR[0].l2) && conv_folds_interaction(dsR[0].sc);
        step([&](const float* x, float* y) { ff(ffC[0], true, x, y, b, P, st); },
             [&](const float* x, float* y) { ff(ffR[0], false, x, y, b, P, st, fold_pw ? b.XC[pp] : nullptr); }, fold_pw);
        step([&](const float* x, float* y) { att(attC[0], true, true, x, y, b, 4, T, st); },
             [&](const float* x, float* y) { att(attR[0], false, true, x, y, b, 4, T, st); });
        step([&](const float* x, float* y) { att(attC[1], true, false, x, y, b, 4, T, st); },
             [&](const float* x, float* y) { att(attR[1], false, false, x, y, b, 4, T, st); });
        for (int k = 0; k < NDS; ++k)
            step([&](const float* x, float* y) { ds(dsC[k], true, x, y, b, 4, T, st); },
                 [&](const float* x, float* y) { ds(dsR[k], false, x, y, b, 4, T, st, fold_pw ? b.XC[pp] : nullptr); }, fold_pw);
        step([&](const float* x, float* y) { ff(ffC[1], true, x, y, b, P, st); },
             [&](const float* x, float* y) { ff(ffR[1], false, x, y, b, P, st, fold_pw ? b.XC[pp] : nullptr); }, fold_pw);
        ln(lnC, c, b.XC[pp], 2 * B, CC, P, st);
        ln(lnR, m, b.XR[pp], B, CC, P, st);
        c = b.XC[pp];
        m = b.XR[pp];
        // ---- decoder (:225-232): cat([skip, out]) two-source, fusion after every layer
        F = 4;
        if (gauss_on) gauss_planes23(c, b.XCg, B, CP, st);
        bool c3 = false;          // `c` is a three-plane tensor
        for (int k = 0; k < NL; ++k) {
            const int ci = KN[NL - k], co = KN[NL - k - 1];
            const int ek = NL - 1 - k;                             // the skip: encoder output ek
            const bool skip3 = gauss_on && (ek == 3 || ek == 4);
            const bool out3 = gauss_on && k == 0;                  // DC[0] feeds the three-product layer 1
            const long CPo = (long)co * 2 * F * T;
            float* dcR = out3 ? b.DC[k] + CPo : b.DC[k];
            if (gauss_on && k < 2) {
                const float* s0 = k == 0 ? b.EC5g : b.EC[ek];
                const float* s1 = k == 0 ? b.XCg : b.DC[0];
                gauss_layer(gdec[k], b, s0, ci, s1, ci, F, 2 * F, T, b.DC[k], out3, st);
            } else {
                Act4 s0 = skip3 ? view3(b.EC[ek], ci, F, T) : act4(b.EC[ek], 2 * ci, F, T);
                Act4 s1 = c3 ? view3(c, ci, F, T) : act4(c, 2 * ci, F, T);
                run_deconv(decC[k], s0, &s1, b.DC[k], 2 * co, 2 * F, B, T, T, st, pf);
            }
            Act4 r0 = act4(b.ER[ek], ci, F, T), r1 = act4(m, ci, F, T);
            const bool fold = fold_env() && conv_folds_interaction(decR[k]);
            run_deconv(decR[k], r0, &r1, b.DR[k], co, 2 * F, B, T, T, st, pf, nullptr, 0, -1, false, fold ? dcR : nullptr, out3 ? 3 : 2);
            F *= 2;
            if (!fold) {
                SE_CHECK(!out3, "Uformer: three-plane decoder tensors need the folded interaction (SE_UF_FOLD=0 with SE_UF_GAUSS=1)");
                fusion(b.DC[k], b.DR[k], B, (long)co * F * T, st);
            }
            if (out3 && !(fold && conv_fold_writes_sum())) gauss_sum(b.DC[k], B, CPo, st);
            c = b.DC[k];
            c3 = out3;
            m = b.DR[k];
        }
        hipLaunchKernelGGL(uf_post_kernel, dim3((T + 255) / 256, NBIN, B), dim3(256), 0, st, c, m, b.mag0, b.ph0, b.est, T, ctx.p_out);
        SE_HIP(hipGetLastError());
        if (out_cplx) {      // :264-286 output_cplx [B][2][257][T_true] (the engine's rows may be padded to whole 16 B groups)
            const float* oc = b.est;
            if (ctx.p_out != 1.f) {      // the reference collects it BEFORE the (commented-out) decompression, :273
                hipLaunchKernelGGL(uf_post_kernel, dim3((T + 255) / 256, NBIN, B), dim3(256), 0, st, c, m, b.mag0, b.ph0, b.spec, T, 1.f);
                oc = b.spec;             // the input spectrum was consumed by uf_prep_kernel
            }
            SE_HIP(hipMemcpy2DAsync(out_cplx, (size_t)T_true * sizeof(float), oc, (size_t)T * sizeof(float),
                                    (size_t)T_true * sizeof(float), (size_t)B * 2 * NBIN, hipMemcpyDeviceToDevice, st));
        }
        launch_istft(ctx.geom, b.est, B, T, T, b.frames, cs, out, out_pitch, HOP * (T_true - 1), st);  // :276
    }
};

}  // namespace

std::unique_ptr<Model> make_uformer(EngineCtx& ctx) { return std::unique_ptr<Model>(new Uformer(ctx)); }

}  // namespace se
