// One TCM / GLU block of the gated-U-Net models in ONE kernel (CTSNet Step1_network.py:158-188 `Glu`,
// Step2_network.py:126-158 `glu`, G2Net_VB/gaf_net_320.py:245-274 `Glu`, TaylorSENet/TaylorSENet.py:641-685 `SqueezedTCM`):
//
//   x [256][T] -> 1x1 conv (256 -> 64) -> per branch { PReLU -> InstanceNorm1d -> [shared causal FIR] -> dilated causal
//   Conv1d (64 -> 64, k taps) } -> [left * sigmoid(right)] -> PReLU -> InstanceNorm1d -> 1x1 conv (64 -> 256) -> + x
//
// Round 1 ran this as 3 (4) tap-table GEMM launches + 2 (3) norm/FIR launches per block - 64-row tiles that fill a quarter
// of the chip at 40 us each and an HBM round trip of the [64][T] tensor between every pair.  The
// InstanceNorm statistics span the whole utterance, so the unit that can run a block without leaving the chip is ONE
// UTTERANCE: a 512-thread workgroup per utterance keeps the 64-channel bottleneck tensor in its registers (the MFMA
// accumulators of the 1x1 conv ARE the tensor: 2 row tiles x 2 column tiles x 16 per lane) and the normalised copy that
// the next GEMM reads in LDS ([64][Tp] floats, 104 KB at T = 401).  All three GEMMs run on v_mfma_f32_32x32x2_f32 (exact
// fp32 products); weights are packed on the host in MFMA fragment order and come straight from L2 into registers, the
// 256-channel input is read once from HBM as the B operand of the first GEMM and once as the residual.
//
// Used when the batch fills the chip (one workgroup per CU); small batches keep the multi-launch path (blocks.h).
#include "kernels.h"
#include "common.h"

namespace se {

static int tcm_dbg_env() {
    static const int v = getenv("SE_TCM_DBG") ? atoi(getenv("SE_TCM_DBG")) : 0;
    return v;
}

typedef float tcm_x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int TCM_C = 64, TCM_CIO = 256, TCM_NW = 8;      // bottleneck channels, block in/out channels, waves per workgroup

// sum over the 32 lanes that share (lane >> 5), on the DPP network: quad swaps, half-row and row mirrors give every lane its
// 16-lane row's total, row_bcast15 then adds row 0's / row 2's total into rows 1 / 3 - the result is valid in lanes 16-31 and
// 48-63 (callers read it from lane 31 of their half).  Five v_add_f32_dpp per sum; the `__shfl_xor` form of rounds 2-4 was five
// ds_bpermute_b32 each - 640 LDS round trips per lane and InstanceNorm head, 20 of the 29 us the statistics took per block.
__device__ __forceinline__ float half_sum32(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));    // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));    // row_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));   // row_bcast15 into rows 1, 3
    return v;
}
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }      // MFMA 32x32 D layout

}  // namespace

struct HeadParams {
    const float *slope, *gamma, *beta, *fir;
};

struct TcmFusedArgs {
    const float* x; float* y;
    int B, T, Tp;
    const float *w1, *w2L, *w2R, *w3;
    HeadParams hL, hR, hO;
    int dil, K;
    const int* tlen;
    int strip;     // 1: the output tile leaves through a per-wave LDS strip (needs 36 KB more LDS)
    int dbg;       // tuning ablations (SE_TCM_DBG): 1 no GEMM 1, 2 no head statistics, 4 no dilated conv, 8 no GEMM 3, 16 no FIR
};

// CUM: the heads normalise with CumulativeLayerNorm1d (the `_new` directories, CTSNet_new/Step1_network.py:213-251: frame t by
// the mean / biased variance of all 64 * (t + 1) values of frames 0..t) instead of InstanceNorm1d (per channel over the
// utterance): the statistics are per COLUMN - a lane sums its 32 rows of its two columns, the partner lane (lane ^ 32) has the
// other 32 - and a float64 prefix scan over the <= 512 columns by one wave turns them into running sums.
template <int KS, bool GATED, bool CUM>
__global__ __launch_bounds__(512) void tcm_fused_kernel(const TcmFusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int Tp = a.Tp, T = a.T;
    float* A = lds;                            // [64][Tp] normalised tensor = B operand of the next GEMM
    float* part = lds + TCM_C * Tp;            // [8 waves][64] per-wave partial sums
    float* prm = part + TCM_NW * TCM_C;        // [5][64]: slope, scale = rstd * gamma, shift = beta - mean * scale, mean, FIR taps
    float* Ws = prm + 5 * TCM_C;               // [2][16 k-pairs][2 row tiles][64]: weight fragments of the running GEMM, double-buffered
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int b = blockIdx.x;
    const int Tv = a.tlen ? a.tlen[b] : T;     // frames the statistics cover (ragged batch: the row's own)
    const float* xb = a.x + (long)b * TCM_CIO * T;
    float* yb = a.y + (long)b * TCM_CIO * T;
    const int ntiles = (T + 31) >> 5;
    const int tj0 = 32 * wave + l31, tj1 = 32 * (wave + TCM_NW) + l31;      // this lane's two columns
    const bool v1 = wave + TCM_NW < ntiles;                                 // wave-uniform: second column tile exists
    const int tc0 = min(tj0, T - 1), tc1 = min(tj1, T - 1);

    // Weights: every wave needs the same A fragments, so a GEMM's packed weights stream through LDS in chunks of CK k-pairs
    // x 2 row tiles (8 KB: one float4 per thread), the next chunk's global load in flight under this chunk's MFMAs.  (Loading
    // fragments straight into registers, per wave and without lookahead, left every phase bound by one L2 round trip per
    // 4 k-pairs: 117 us for the last GEMM against 27 us of matrix work.)
    constexpr int CK = 16;
    typedef float wf4 __attribute__((ext_vector_type(4)));
    // fragment (kp, mt) of a [K/2][MT][64] packed matrix; this thread's float4 of chunk c (row tiles mt0, mt0 + 1)
    auto w_load = [&](const float* __restrict__ wg, int MT, int mt0, int c) __attribute__((always_inline)) {
        const int e = 4 * tid, kpl = e >> 7, mt = (e >> 6) & 1, l0 = e & 63;
        return *reinterpret_cast<const wf4*>(wg + ((long)(c * CK + kpl) * MT + mt0 + mt) * 64 + l0);
    };
    auto w_store = [&](int buf, wf4 v) __attribute__((always_inline)) {
        *reinterpret_cast<wf4*>(Ws + buf * (CK * 128) + 4 * tid) = v;
    };

    // ---------------------------------------------------------------- GEMM 1: h = W_in (64 x 256) . x (256 x T)
    tcm_x16 h[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) h[i][j][r] = 0.f;
    if (!(a.dbg & 1)) {
        // B operand (this wave's two 32-frame column tiles of x) straight from HBM into registers, one whole chunk ahead
        const float* xr = xb + hi * T;
        float b0[CK][2], b1[CK][2];
        auto x_load = [&](float (&bv)[CK][2], int c) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < CK; ++u) {
                bv[u][0] = xr[(2 * (c * CK + u)) * T + tc0];          // 32-bit offsets: 256 * T fits easily
                bv[u][1] = xr[(2 * (c * CK + u)) * T + tc1];
            }
        };
        auto mma = [&](const float (&bv)[CK][2], int buf) __attribute__((always_inline)) {
            const float* wb = Ws + buf * (CK * 128) + lane;
#pragma unroll
            for (int u = 0; u < CK; ++u) {
                const float a0 = wb[u * 128], a1 = wb[u * 128 + 64];
                // all four tiles unconditionally: a wave without a second column tile (T = 401: waves 5-7) would only idle
                // at the next barrier, and a branch around 16-register accumulators costs copies
                h[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[u][0], h[0][0], 0, 0, 0);
                h[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[u][0], h[1][0], 0, 0, 0);
                h[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[u][1], h[0][1], 0, 0, 0);
                h[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[u][1], h[1][1], 0, 0, 0);
            }
        };
        constexpr int NC = TCM_CIO / 2 / CK;       // 8 chunks
        x_load(b0, 0);
        w_store(0, w_load(a.w1, 2, 0, 0));
        __syncthreads();
        for (int c = 0; c < NC; c += 2) {
            wf4 wn = w_load(a.w1, 2, 0, c + 1);
            x_load(b1, c + 1);
            mma(b0, 0);
            w_store(1, wn);
            __syncthreads();
            if (c + 2 < NC) {
                wn = w_load(a.w1, 2, 0, c + 2);
                x_load(b0, c + 2);
            }
            mma(b1, 1);
            if (c + 2 < NC) w_store(0, wn);
            __syncthreads();
        }
    }

    // PReLU -> InstanceNorm1d (two-pass statistics over the Tv valid frames) -> A[c][t], then the shared causal FIR in place
    auto head = [&](const tcm_x16 (&src)[2][2], const HeadParams& hp, int K) __attribute__((always_inline)) {
        __syncthreads();                                   // every reader of the previous A / prm is done
        if (tid < TCM_C) prm[tid] = hp.slope[tid];
        if (K > 0 && tid >= TCM_C && tid < TCM_C + K) prm[4 * TCM_C + tid - TCM_C] = hp.fir[tid - TCM_C];     // K <= 64
        __syncthreads();
        if constexpr (CUM) {
            // Ws is free between the GEMMs: [Tp] float64 column sums, [Tp] sums of squares, [Tp] mean, [Tp] rstd
            double* cs = reinterpret_cast<double*>(Ws);
            double* cq = cs + Tp;
            float* cmu = reinterpret_cast<float*>(cq + Tp);
            float* crs = cmu + Tp;
            if (tid < TCM_C) {
                prm[TCM_C + tid] = hp.gamma[tid];
                prm[2 * TCM_C + tid] = hp.beta[tid];
            }
            float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sl = prm[32 * mt + acc_row(r, hi)];
                    float v0 = src[mt][0][r], vv1 = src[mt][1][r];
                    v0 = v0 >= 0.f ? v0 : sl * v0;
                    vv1 = vv1 >= 0.f ? vv1 : sl * vv1;
                    s0 += v0; q0 += v0 * v0;
                    s1 += vv1; q1 += vv1 * vv1;
                }
            s0 += __shfl_xor(s0, 32, 64); q0 += __shfl_xor(q0, 32, 64);
            s1 += __shfl_xor(s1, 32, 64); q1 += __shfl_xor(q1, 32, 64);
            if (hi == 0) {
                if (tj0 < Tp) { cs[tj0] = s0; cq[tj0] = q0; }
                if (v1 && tj1 < Tp) { cs[tj1] = s1; cq[tj1] = q1; }
            }
            __syncthreads();
            if (wave == 0) {            // 64 lanes x 8 consecutive columns cover Tp <= 512
                double ls[8], lq[8], as = 0.0, aq = 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int t = lane * 8 + i;
                    if (t < Tp) { as += cs[t]; aq += cq[t]; }
                    ls[i] = as; lq[i] = aq;
                }
                double ps = as, pq = aq;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const double ts = __shfl_up(ps, o, 64), tq = __shfl_up(pq, o, 64);
                    if (lane >= o) { ps += ts; pq += tq; }
                }
                const double bs = ps - as, bq = pq - aq;        // sums of all columns in front of this lane's eight
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int t = lane * 8 + i;
                    if (t < Tp) {
                        const double S = bs + ls[i], Q = bq + lq[i], cnt = 64.0 * (double)(t + 1), mm = S / cnt;
                        const double var = (Q - 2.0 * mm * S) / cnt + mm * mm;
                        cmu[t] = (float)mm;
                        crs[t] = (float)(1.0 / sqrt(var + 1e-5));
                    }
                }
            }
            __syncthreads();
            const int c0i = min(tj0, Tp - 1), c1i = min(tj1, Tp - 1);
            const float mu0 = cmu[c0i], rs0 = crs[c0i], mu1 = cmu[c1i], rs1 = crs[c1i];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * mt + acc_row(r, hi);
                    const float sl = prm[row], ga = prm[TCM_C + row], be = prm[2 * TCM_C + row];
                    float v0 = src[mt][0][r], vv1 = src[mt][1][r];
                    v0 = v0 >= 0.f ? v0 : sl * v0;
                    vv1 = vv1 >= 0.f ? vv1 : sl * vv1;
                    if (tj0 < Tp) A[row * Tp + tj0] = (v0 - mu0) * rs0 * ga + be;
                    if (v1 && tj1 < Tp) A[row * Tp + tj1] = (vv1 - mu1) * rs1 * ga + be;
                }
            __syncthreads();
        } else {
        // pass 1: mean of PReLU(src) per channel
        const bool in0 = tj0 < Tv, in1 = v1 && tj1 < Tv;
        if (a.dbg & 2) {          // (ablation: identity normalisation, so that the decode stays finite)
            if (tid < TCM_C) {
                prm[1 * TCM_C + tid] = 1.f;
                prm[2 * TCM_C + tid] = 0.f;
            }
            __syncthreads();
        }
        if (!(a.dbg & 2)) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * mt + acc_row(r, hi);
                const float sl = prm[row];
                float v0 = src[mt][0][r], vv1 = src[mt][1][r];
                v0 = v0 >= 0.f ? v0 : sl * v0;
                vv1 = vv1 >= 0.f ? vv1 : sl * vv1;
                const float s = half_sum32((in0 ? v0 : 0.f) + (in1 ? vv1 : 0.f));
                if (l31 == 31) part[wave * TCM_C + row] = s;
            }
        __syncthreads();
        if (tid < TCM_C) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < TCM_NW; ++w) s += part[w * TCM_C + tid];
            prm[3 * TCM_C + tid] = s / (float)Tv;
        }
        __syncthreads();
        // pass 2: biased variance around that mean
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * mt + acc_row(r, hi);
                const float sl = prm[row], mu = prm[3 * TCM_C + row];
                float v0 = src[mt][0][r], vv1 = src[mt][1][r];
                v0 = (v0 >= 0.f ? v0 : sl * v0) - mu;
                vv1 = (vv1 >= 0.f ? vv1 : sl * vv1) - mu;
                const float s = half_sum32((in0 ? v0 * v0 : 0.f) + (in1 ? vv1 * vv1 : 0.f));
                if (l31 == 31) part[wave * TCM_C + row] = s;
            }
        __syncthreads();
        if (tid < TCM_C) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < TCM_NW; ++w) s += part[w * TCM_C + tid];
            const float rstd = 1.f / sqrtf(s / (float)Tv + 1e-5f);
            const float sc = rstd * hp.gamma[tid];
            prm[1 * TCM_C + tid] = sc;
            prm[2 * TCM_C + tid] = hp.beta[tid] - prm[3 * TCM_C + tid] * sc;
        }
        __syncthreads();
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * mt + acc_row(r, hi);
                const float sl = prm[row], sc = prm[TCM_C + row], sh = prm[2 * TCM_C + row];
                float v0 = src[mt][0][r], vv1 = src[mt][1][r];
                v0 = v0 >= 0.f ? v0 : sl * v0;
                vv1 = vv1 >= 0.f ? vv1 : sl * vv1;
                if (tj0 < Tp) A[row * Tp + tj0] = v0 * sc + sh;
                if (v1 && tj1 < Tp) A[row * Tp + tj1] = vv1 * sc + sh;
            }
        __syncthreads();
        }
        if (K > 0 && !(a.dbg & 16)) {
            // ShareSepConv (Step1_network.py:190-204): y[t] = sum_k fir[k] * n[t - (K-1) + k], in place, one wave per row.
            // A lane owns EIGHT CONSECUTIVE frames and first pulls the 72 values [8 lane - 64, 8 lane + 8) of the row into
            // registers (18 x ds_read_b128), then runs all taps out of them: 18 LDS reads per lane and row instead of 8 per tap
            // (round 2: 504 for K = 63 - 129 of CTSNet's 434 us per block).  The taps sit right-aligned in a 63-entry table
            // (zeros in front), so one fully unrolled loop serves every K <= 63.
            typedef float fv4 __attribute__((ext_vector_type(4)));
            float* tp63 = prm + 4 * TCM_C;                 // [64]: prm[4*64 + k] holds tap k of K; re-pack right-aligned
            __syncthreads();
            float mytap = 0.f;
            if (tid < 64) mytap = (tid >= 63 - K && tid < 63) ? tp63[tid - (63 - K)] : 0.f;
            __syncthreads();
            if (tid < 64) tp63[tid] = mytap;
            __syncthreads();
#pragma unroll 1
            for (int row = wave; row < TCM_C; row += TCM_NW) {
                float* Ar = A + row * Tp;
                const int t0 = 8 * lane;
                float o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = 0.f;
                // four passes of 16 (15) taps, each out of a 28-value register window (one 72-value window next to the block's
                // 128 accumulator registers spilled 150-190 VGPRs): output t0 + i reads n[t0 + i - 62 + k'] for the
                // right-aligned tap k' - index 2 + i + (k' - k0) of the window that starts at t0 - 64 + k0
#pragma unroll 1
                for (int q = 0; q < 4; ++q) {
                    const int k0 = 16 * q;
                    float xw[28];
#pragma unroll
                    for (int g = 0; g < 7; ++g) {
                        const int ti = t0 - 64 + k0 + 4 * g;
                        fv4 v = {0.f, 0.f, 0.f, 0.f};
                        if (ti >= 0 && ti + 3 < Tp) v = *reinterpret_cast<const fv4*>(Ar + ti);
                        xw[4 * g] = v[0]; xw[4 * g + 1] = v[1]; xw[4 * g + 2] = v[2]; xw[4 * g + 3] = v[3];
                    }
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        {
                            const float w = tp63[k0 + k];          // entry 63 is zero
#pragma unroll
                            for (int i = 0; i < 8; ++i) o[i] = fmaf(w, xw[2 + i + k], o[i]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (t0 + 7 < Tp) {
                    *reinterpret_cast<fv4*>(Ar + t0) = fv4{o[0], o[1], o[2], o[3]};
                    *reinterpret_cast<fv4*>(Ar + t0 + 4) = fv4{o[4], o[5], o[6], o[7]};
                }
            }
            __syncthreads();
        }
    };

    // dilated causal Conv1d 64 -> 64 from A: acc[co][t] = sum_{tap, ci} W[co][ci][tap] * A[ci][t - (KS-1-tap) * dil]
    // (GEMM K index tap-major: k-pair kp = tap * 32 + ci / 2, so a 16-k-pair weight chunk lies inside one tap)
    auto dconv = [&](tcm_x16 (&acc)[2][2], const float* __restrict__ w2) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        if (a.dbg & 4) return;
        constexpr int NC = KS * (TCM_C / 2) / CK;      // 2 chunks per tap
        const float* Ah = A + hi * Tp;
        __syncthreads();                               // Ws is free (the previous GEMM's last chunk has been read)
        w_store(0, w_load(w2, 2, 0, 0));
        __syncthreads();
        for (int c = 0; c < NC; ++c) {
            wf4 wn;
            if (c + 1 < NC) wn = w_load(w2, 2, 0, c + 1);
            const int tap = c >> 1, cp0 = (c & 1) * CK;
            const int shift = (KS - 1 - tap) * a.dil;
            const int t0 = tj0 - shift, t1 = tj1 - shift;
            const bool ok0 = t0 >= 0 && t0 < Tp, ok1 = t1 >= 0 && t1 < Tp;
            const float* wb = Ws + (c & 1) * (CK * 128) + lane;
            const float* A0 = Ah + (2 * cp0) * Tp + (ok0 ? t0 : 0);
            const float* A1 = Ah + (2 * cp0) * Tp + (ok1 ? t1 : 0);
#pragma unroll
            for (int u = 0; u < CK; ++u) {
                const float a0 = wb[u * 128], a1 = wb[u * 128 + 64];
                float bx = A0[2 * u * Tp], by = A1[2 * u * Tp];
                bx = ok0 ? bx : 0.f;
                by = ok1 ? by : 0.f;
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bx, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bx, acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, by, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, by, acc[1][1], 0, 0, 0);
            }
            if (c + 1 < NC) w_store((c + 1) & 1, wn);
            __syncthreads();
        }
    };

    tcm_x16 m[2][2];
    if constexpr (GATED) {
        tcm_x16 g[2][2];
        head(h, a.hR, a.K);
        dconv(g, a.w2R);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) g[i][j][r] = fm_sigmoid(g[i][j][r]);      // hardware exp / reciprocal, as in the gc_kernel GLU epilogue
        head(h, a.hL, a.K);
        dconv(m, a.w2L);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) m[i][j] *= g[i][j];
    } else {
        head(h, a.hL, a.K);
        dconv(m, a.w2L);
    }
    head(m, a.hO, 0);

    // ---------------------------------------------------------------- GEMM 3: y = W_out (256 x 64) . A + x, 2 row tiles per pass
    if (!(a.dbg & 8)) {
        const float* Ah = A + hi * Tp;
        const int ta0 = min(tj0, Tp - 1), ta1 = min(tj1, Tp - 1);
        constexpr int NC = (TCM_CIO / 64) * (TCM_C / 2 / CK);      // 4 passes x 2 chunks
        __syncthreads();
        w_store(0, w_load(a.w3, 8, 0, 0));
        __syncthreads();
        tcm_x16 acc[2][2];
        wf4 xres[2][2][4];
        for (int c = 0; c < NC; ++c) {
            const int ps = c >> 1, half = c & 1;
            wf4 wn;
            // chunk c + 1: pass (c + 1) / 2, k-pairs 16 * ((c + 1) % 2) ..  -> packed fragment row (kp, 2 ps' + mt)
            if (c + 1 < NC) {
                const int e = 4 * tid, kpl = e >> 7, mt = (e >> 6) & 1, l0 = e & 63, psn = (c + 1) >> 1, hn = (c + 1) & 1;
                wn = *reinterpret_cast<const wf4*>(a.w3 + ((long)(hn * CK + kpl) * 8 + 2 * psn + mt) * 64 + l0);
            }
            if (half == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                // the residual rows of this pass (x, in the layout the strips are read back in) are requested HERE, ahead of
                // the pass's 128 matrix instructions: issued in the epilogue they were 16 exposed HBM round trips per pass
                // (rounds 2-4: ~36 us of a block's 172 with the matrix pipe idle)
                if (a.strip && !(a.dbg & 64)) {
                    const int lr = lane >> 3, lc = (lane & 7) * 4;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int tg = 32 * (wave + TCM_NW * j) + lc;
                            if ((j == 1 && !v1) || tg + 3 >= T) continue;
#pragma unroll
                            for (int it = 0; it < 4; ++it)
                                xres[mt][j][it] = *reinterpret_cast<const wf4*>(xb + (64 * ps + 32 * mt + it * 8 + lr) * T + tg);
                        }
                }
            }
            const float* wb = Ws + (c & 1) * (CK * 128) + lane;
            const float* A0 = Ah + (2 * half * CK) * Tp + ta0;
            const float* A1 = Ah + (2 * half * CK) * Tp + ta1;
#pragma unroll
            for (int u = 0; u < CK; ++u) {
                const float a0 = wb[u * 128], a1 = wb[u * 128 + 64];
                const float bx = A0[2 * u * Tp], by = A1[2 * u * Tp];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bx, acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bx, acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, by, acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, by, acc[1][1], 0, 0, 0);
            }
            if (c + 1 < NC) w_store((c + 1) & 1, wn);
            if (half == 1 && !(a.dbg & 32)) {
                if (a.strip) {
                    // the accumulator layout gives 2 rows x 128 B per memory instruction (measured 2.1 / 3.4 TB/s for the
                    // residual read / the store); a 32 x 32 tile goes through a per-wave LDS strip instead and leaves as
                    // 8 rows x 128 B per instruction, 16 B per lane along t
                    float* strip = Ws + 2 * CK * 128 + wave * (32 * 36);
                    const int lr = lane >> 3, lc = (lane & 7) * 4;
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            if (j == 1 && !v1) continue;
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the previous tile has been read back
#pragma unroll
                            for (int r = 0; r < 16; ++r) strip[acc_row(r, hi) * 36 + l31] = acc[mt][j][r];
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            const int tg = 32 * (wave + TCM_NW * j) + lc;
#pragma unroll
                            for (int it = 0; it < 4; ++it) {
                                const int row = it * 8 + lr;
                                wf4 v = *reinterpret_cast<const wf4*>(strip + row * 36 + lc);
                                const int o = (64 * ps + 32 * mt + row) * T + tg;
                                if (tg + 3 < T) {
                                    v += xres[mt][j][it];          // (SE_TCM_DBG=64: not loaded - timing ablation only)
                                    *reinterpret_cast<wf4*>(yb + o) = v;
                                } else {
#pragma unroll
                                    for (int k = 0; k < 4; ++k)
                                        if (tg + k < T) yb[o + k] = v[k] + xb[o + k];
                                }
                            }
                        }
                } else {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        // 32-bit element offsets (a 64-bit address per element held 128 VGPRs and spilled the accumulators)
                        const int row = 64 * ps + 32 * mt + acc_row(r, hi);
                        const int o0 = row * T + tj0, o1 = row * T + tj1;
                        if (tj0 < T) yb[o0] = acc[mt][0][r] + ((a.dbg & 64) ? 0.f : xb[o0]);
                        if (v1 && tj1 < T) yb[o1] = acc[mt][1][r] + ((a.dbg & 64) ? 0.f : xb[o1]);
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
// w[m][k] (row-major, M x K) -> MFMA A fragments [K/2][M/32][64]: lane l of fragment (kp, mt) holds w[32 mt + l % 32][2 kp + l / 32]
static std::vector<float> pack_frag(const std::vector<float>& w, int M, int K) {
    std::vector<float> p((size_t)M * K);
    for (int kp = 0; kp < K / 2; ++kp)
        for (int mt = 0; mt < M / 32; ++mt)
            for (int l = 0; l < 64; ++l)
                p[((size_t)kp * (M / 32) + mt) * 64 + l] = w[(size_t)(32 * mt + (l & 31)) * K + 2 * kp + (l >> 5)];
    return p;
}

TcmFusedW tcm_fused_build(const std::vector<float>& w_in, const std::vector<float>& w_left, const std::vector<float>* w_right,
                          const std::vector<float>& w_out, int ks) {
    TcmFusedW f;
    f.ks = ks;
    f.w1 = to_device(pack_frag(w_in, TCM_C, TCM_CIO));
    // dilated conv weights arrive as [co][ci][tap]; GEMM K index is tap-major: k = tap * 64 + ci
    auto tapmajor = [&](const std::vector<float>& w) {
        std::vector<float> r((size_t)TCM_C * TCM_C * ks);
        for (int co = 0; co < TCM_C; ++co)
            for (int ci = 0; ci < TCM_C; ++ci)
                for (int t = 0; t < ks; ++t) r[(size_t)co * (TCM_C * ks) + t * TCM_C + ci] = w[((size_t)co * TCM_C + ci) * ks + t];
        return pack_frag(r, TCM_C, TCM_C * ks);
    };
    f.w2L = to_device(tapmajor(w_left));
    if (w_right) f.w2R = to_device(tapmajor(*w_right));
    f.w3 = to_device(pack_frag(w_out, TCM_CIO, TCM_C));
    return f;
}

void tcm_fused_free(TcmFusedW& f) {
    for (float* p : {f.w1, f.w2L, f.w2R, f.w3})
        if (p) (void)hipFree(p);
    f = TcmFusedW{};
}

bool tcm_fused_supported(int T) { return T >= 32 && T <= 512; }

void launch_tcm_fused(const TcmFusedW& f, const TcmFusedHeads& hd, const float* x, float* y, int B, int T, int dil, int K,
                      hipStream_t s, bool cum) {
    SE_CHECK(tcm_fused_supported(T) && f.w1, "fused TCM block: unsupported shape");
    const int Tp = (T + 31) / 32 * 32;
    const Ragged* rg = ragged_ctx();
    TcmFusedArgs a{x, y, B, T, Tp, f.w1, f.w2L, f.w2R, f.w3,
                   {hd.sL, hd.gL, hd.bL, hd.firL}, {hd.sR, hd.gR, hd.bR, hd.firR}, {hd.sO, hd.gO, hd.bO, nullptr},
                   dil, K, rg ? rg->tlen : nullptr, 0, tcm_dbg_env()};
    size_t lds = ((size_t)TCM_C * Tp + TCM_NW * TCM_C + 5 * TCM_C + 2 * 16 * 128) * sizeof(float);
    static const bool strip_env = !(getenv("SE_TCM_STRIP") && atoi(getenv("SE_TCM_STRIP")) == 0);
    const size_t strip_bytes = (size_t)TCM_NW * 32 * 36 * sizeof(float);
    if (strip_env && lds + strip_bytes <= 160 * 1024) {      // T <= 416: the strips fit next to the [64][Tp] tensor
        a.strip = 1;
        lds += strip_bytes;
    }
    const bool gated = f.w2R != nullptr;
    {
        static bool seen[64] = {};
        if (first_on_device(seen)) {        // (all variants share one function-pointer type: raise every limit once per device)
            auto up = [](auto kern) {
                SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            };
            up(tcm_fused_kernel<5, true, false>); up(tcm_fused_kernel<5, false, false>); up(tcm_fused_kernel<3, true, false>);
            up(tcm_fused_kernel<3, false, false>); up(tcm_fused_kernel<5, true, true>); up(tcm_fused_kernel<5, false, true>);
            up(tcm_fused_kernel<3, true, true>); up(tcm_fused_kernel<3, false, true>);
        }
    }
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(B), dim3(512), lds, s, a); };
    if (cum) {
        if (f.ks == 5 && gated) go(tcm_fused_kernel<5, true, true>);
        else if (f.ks == 5) go(tcm_fused_kernel<5, false, true>);
        else if (f.ks == 3 && gated) go(tcm_fused_kernel<3, true, true>);
        else if (f.ks == 3) go(tcm_fused_kernel<3, false, true>);
        else SE_CHECK(false, "fused TCM block: kernel size must be 3 or 5");
    } else {
        if (f.ks == 5 && gated) go(tcm_fused_kernel<5, true, false>);
        else if (f.ks == 5) go(tcm_fused_kernel<5, false, false>);
        else if (f.ks == 3 && gated) go(tcm_fused_kernel<3, true, false>);
        else if (f.ks == 3) go(tcm_fused_kernel<3, false, false>);
        else SE_CHECK(false, "fused TCM block: kernel size must be 3 or 5");
    }
    SE_HIP(hipGetLastError());
}

}  // namespace se
