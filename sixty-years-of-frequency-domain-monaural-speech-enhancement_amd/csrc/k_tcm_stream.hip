// Frame-online TCM / GLU block of the cumulative-LayerNorm variants as ONE kernel per block and chunk.
//
// Reference: the block itself is CTSNet_new/Step1_network.py:158-204 (`Glu`: 1x1 in-conv 256 -> 64, two branches
// [PReLU -> CumulativeLayerNorm1d (:213-251) -> ShareSepConv -> causal pad -> dilated Conv1d], gate product, PReLU -> cLN ->
// 1x1 out-conv 64 -> 256, + residual), Step2_network.py:126-158 (`glu`), G2Net_new/gaf_net_320.py `Glu` (k = 3, single
// branch) and TaylorSENet_new's `SqueezedTCM`.  The reference only runs them offline; the frame-online mode
// (include/se_engine.h se_stream_*) is this engine's.
//
// Round 2 ran a block's chunk as ~15 launches of 3-8 us each (1x1 GEMM, per branch: history exchange + cLN window + history
// exchange + dilated conv, gate, cLN, 1x1 GEMM) - 420-620 launches per 10 ms frame on the three `_new` models, 2.9-3.4 ms
// per push, launch bound.  A chunk of a few frames is ~75 K multiply-adds per frame and block: far too
// small for a tiled GEMM, so here ONE workgroup per stream walks the whole block on the VALU: the 256-channel input
// columns, every intermediate [64][n] tensor and the FIR / dilated-conv windows live in LDS, weights are read transposed
// (consecutive lanes = consecutive output rows) from L2, and the block's state - the three running cLN sums (float64, as
// the offline scan), the K-1 normalised columns the shared FIR looks back on, the (ks-1)*dilation columns the dilated conv
// looks back on - sits in one per-call-site slot of the stream context (zero-filled at se_stream_begin: zeros ARE the causal
// padding).  Chunks of any length are walked in sub-chunks of up to 8 frames, so the launch sequence never depends on the
// chunk size.
#include "kernels.h"
#include "common.h"

namespace se {

namespace {

constexpr int NS = 1;              // frames per sub-chunk.  The latency-critical push is ONE frame; with 2 / 4 / 8 frames per
                                   // sub-chunk the compiler hoists every LDS operand of the unrolled multiply-add phases in
                                   // front of them (512 VGPRs + 130 / 230 / 300 spilled), so longer chunks walk frame by frame
#define SE_TS_FENCE() __builtin_amdgcn_sched_barrier(0)   // keeps the compiler from hoisting a whole phase's LDS reads

struct TcmStreamArgs {
    const float* x; float* y;       // windows [B][256][Tw], new frames in columns [H, H + n)
    int Tw, H, n; long t0;          // t0: index of the first new frame in the stream
    const float *w_in, *w_l, *w_r, *w_out;      // transposed: [256][64], [64*ks][64] (k = ci * ks + tap), same, [64][256]
    TcmFusedHeads hd;
    int K, ks, dil;                 // ShareSepConv length (0 = none), dilated-conv taps, dilation
    char* state; long state_stride; // per stream: double carry[3][2] | firL [64][K-1] | firR | cvL [64][(ks-1)*dil] | cvR
};

__device__ __forceinline__ float sigm_(float v) { return 1.f / (1.f + fm_exp(-v)); }

// Latency, not arithmetic, bounds a chunk of one or two frames (~75 K multiply-adds): every dependent global access is a
// ~1 us round trip (first version, a load per loop iteration with runtime trip counts: 125 us per block; second, the whole
// shifted history windows loaded and stored per push: 50 us).  So
//   * the FIR / dilated-conv histories are RINGS in the state slot (column of frame t at t mod R, R a power of two >= look-back
//     + 8): a push writes its n new columns and gathers only what it reads - the K-1 FIR columns and the KS tap columns per
//     new frame - instead of moving (KS-1)*dilation columns through the chip;
//   * everything a sub-chunk reads from global memory besides weights is issued in ONE batch at its start (input columns,
//     both branches' history gathers, and at kernel entry the head parameters, FIR taps and cLN sums);
//   * a thread's weights of a GEMM phase (64, or 16 * KS) are loaded into registers by one fully unrolled batch that is
//     issued BEFORE the LDS-only phase in front of it (cLN, FIR), so its round trip hides there;
//   * column sums of the cLN are wave reductions in float64 (fixed order), not a 64-step serial loop.
// Ring columns written by an earlier sub-chunk of the same launch are read back with agent-scope loads (past the L1).
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ring_of(int need) {
    int r = 16;
    while (r < need + NS) r <<= 1;
    return r;
}

template <int KS>
__global__ __launch_bounds__(256) void tcm_stream_kernel(const TcmStreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KH = a.K > 0 ? a.K - 1 : 0, CH = (KS - 1) * a.dil;
    const int RF = ring_of(KH), RC = ring_of(CH);
    const int FP = KH + NS;                            // pitch of the FIR window
    const bool gated = a.w_r != nullptr;
    const int nb = gated ? 2 : 1;
    float* xs = sm;                        // [256][NS] input columns of the sub-chunk
    float* hb = xs + 256 * NS;             // [64][NS]  in-conv output
    float* part = hb + 64 * NS;            // [4][64][NS] partial sums of the four k-parts (waves)
    float* ab = part + 4 * 64 * NS;        // [64][NS]  branch value
    float* rg = ab + 64 * NS;              // [64][NS]  gate
    float* hp = rg + 64 * NS;              // [9][64]   sL gL bL sR gR bR sO gO bO
    float* ft = hp + 9 * 64;               // [2][64]   FIR taps (K <= 64)
    float* mu = ft + 2 * 64;               // [NS], then rstd [NS]
    double* ds = reinterpret_cast<double*>(mu + 2 * NS);       // [NS][2] column sums, then carry [3][2]
    double* carry = ds + 2 * NS;
    float* ct = reinterpret_cast<float*>(carry + 8);           // [nb][64][KS][NS] tap columns of the dilated conv
    float* wf = ct + nb * 64 * KS * NS;                        // [nb][64][FP] FIR windows: K-1 history columns, then the sub-chunk
    char* stb = a.state + (long)b * a.state_stride;
    double* g_carry = reinterpret_cast<double*>(stb);
    float* g_fir = reinterpret_cast<float*>(stb + 64);         // [nb][64][RF] ring of normalised columns (FIR input)
    float* g_cv = g_fir + (KH > 0 ? nb * 64 * RF : 0);         // [nb][64][RC] ring of dilated-conv input columns
    const float* xb = a.x + (long)b * 256 * a.Tw + a.H;
    float* yb = a.y + (long)b * 256 * a.Tw + a.H;
    const int r = tid & 63, p = tid >> 6;

    // the first sub-chunk's in-conv weights go first: the parameter loads below are stored to LDS (a wait), and loads
    // issued before that wait stay in flight under it
    float wreg[64 > 16 * KS ? 64 : 16 * KS];
    {
        const float* w = a.w_in + (long)(64 * p) * 64 + r;
#pragma unroll
        for (int k = 0; k < 64; ++k) wreg[k] = w[k * 64];
    }
    if (tid < 64) {
        hp[tid] = a.hd.sL[tid]; hp[64 + tid] = a.hd.gL[tid]; hp[128 + tid] = a.hd.bL[tid];
        hp[384 + tid] = a.hd.sO[tid]; hp[448 + tid] = a.hd.gO[tid]; hp[512 + tid] = a.hd.bO[tid];
        if (gated) { hp[192 + tid] = a.hd.sR[tid]; hp[256 + tid] = a.hd.gR[tid]; hp[320 + tid] = a.hd.bR[tid]; }
        if (tid < a.K) {
            ft[tid] = a.hd.firL[tid];
            if (gated) ft[64 + tid] = a.hd.firR[tid];
        }
    }
    if (tid < 6) carry[tid] = g_carry[tid];

    // PReLU -> cumulative LayerNorm of v [64][nn] in place (CTSNet_new/Step1_network.py:213-251: frame t is normalised by the
    // mean / biased variance of all 64 * (t + 1) values of frames 0..t; sums in float64 like the offline scan)
    auto cln = [&](float* v, int nn, long tbase, const float* prm, int which) __attribute__((always_inline)) {
        // wave w owns columns w, w + 4: lane = channel; PReLU, then the column's (sum, sum of squares) by a butterfly
        for (int j = wave; j < nn; j += 4) {
            float t = v[lane * NS + j];
            t = t >= 0.f ? t : prm[lane] * t;
            v[lane * NS + j] = t;
            double s1 = t, s2 = (double)t * t;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                s1 += __shfl_xor(s1, o, 64);
                s2 += __shfl_xor(s2, o, 64);
            }
            if (lane == 0) {
                ds[2 * j] = s1;
                ds[2 * j + 1] = s2;
            }
        }
        __syncthreads();
        if (tid == 0) {
            double s = carry[2 * which], q = carry[2 * which + 1];
            for (int j = 0; j < nn; ++j) {
                s += ds[2 * j];
                q += ds[2 * j + 1];
                const double cnt = 64.0 * (double)(tbase + j + 1), m = s / cnt;
                const double var = (q - 2.0 * m * s) / cnt + m * m;
                mu[j] = (float)m;
                mu[NS + j] = (float)(1.0 / sqrt(var + 1e-5));
            }
            carry[2 * which] = s;
            carry[2 * which + 1] = q;
        }
        __syncthreads();
        for (int i = tid; i < 64 * nn; i += 256) {
            const int c = i / nn, j = i - c * nn;
            v[c * NS + j] = (v[c * NS + j] - mu[j]) * mu[NS + j] * prm[64 + c] + prm[128 + c];
        }
        __syncthreads();
    };

    for (int c0 = 0; c0 < a.n; c0 += NS) {
        const int nn = min(NS, a.n - c0);
        const long tbase = a.t0 + c0;                  // stream index of the sub-chunk's first frame
        // ---- one batch of global reads: in-conv weights, input columns, FIR history, history taps of the dilated convs
        if (c0 > 0) {
            const float* w = a.w_in + (long)(64 * p) * 64 + r;
#pragma unroll
            for (int k = 0; k < 64; ++k) wreg[k] = w[k * 64];
        }
        for (int i = tid; i < 256 * nn; i += 256) {
            const int k = i / nn, j = i - k * nn;
            xs[k * NS + j] = xb[(long)k * a.Tw + c0 + j];
        }
        {
            // FIR history: window column i < KH holds frame tbase - KH + i
            const int n = nb * 64 * KH;
            for (int i0 = tid; i0 < n; i0 += 16 * 256) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = i0 + u * 256;
                    if (i < n) {
                        const int row = i / KH, col = i - row * KH;
                        v[u] = ld_agent(g_fir + (long)row * RF + ((int)(tbase - KH + col) & (RF - 1)));
                    }
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int i = i0 + u * 256;
                    if (i < n) wf[(i / KH) * FP + (i % KH)] = v[u];
                }
            }
            // tap (ci, tap, j) of the dilated conv reads frame tbase + j - (KS - 1 - tap) * dil: from the ring when it is older
            // than the sub-chunk (the others are filled from LDS once the sub-chunk's own columns exist)
            const int m = nb * 64 * KS * nn;
            for (int i = tid; i < m; i += 256) {
                const int j = i % nn, q = i / nn, tap = q % KS, row = q / KS;
                const int back = (KS - 1 - tap) * a.dil - j;
                if (back > 0) ct[(row * KS + tap) * NS + j] = ld_agent(g_cv + (long)row * RC + ((int)(tbase - back) & (RC - 1)));
            }
        }
        __syncthreads();
        {
            float acc[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) acc[j] = 0.f;
            const float* xv = xs + 64 * p * NS;
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                if ((k & 7) == 0) SE_TS_FENCE();
#pragma unroll
                for (int j = 0; j < NS; ++j) acc[j] += wreg[k] * xv[k * NS + j];
            }
#pragma unroll
            for (int j = 0; j < NS; ++j) part[(p * 64 + r) * NS + j] = acc[j];
        }
        __syncthreads();
        for (int i = tid; i < 64 * NS; i += 256) hb[i] = part[i] + part[64 * NS + i] + part[2 * 64 * NS + i] + part[3 * 64 * NS + i];
        __syncthreads();
        // ---- branches: right (gate) first, then left
        for (int br = gated ? 1 : 0; br >= 0; --br) {
            const float* wcv = br ? a.w_r : a.w_l;
            // the branch's conv weights start their round trip now; cLN and FIR below only touch LDS
#pragma unroll
            for (int q = 0; q < 16 * KS; ++q) wreg[q] = wcv[(long)((16 * p) * KS + q) * 64 + r];
            for (int i = tid; i < 64 * NS; i += 256) ab[i] = hb[i];
            __syncthreads();
            cln(ab, nn, tbase, hp + br * 192, br ? 1 : 0);
            if (a.K > 0) {          // ShareSepConv: one causal FIR shared by all channels
                float* w = wf + br * 64 * FP;
                float* ring = g_fir + (long)br * 64 * RF;
                for (int i = tid; i < 64 * nn; i += 256) {
                    const int c = i / nn, j = i - c * nn;
                    const float t = ab[c * NS + j];
                    w[c * FP + KH + j] = t;
                    if (KH > 0) st_agent(ring + (long)c * RF + ((int)(tbase + j) & (RF - 1)), t);
                }
                __syncthreads();
                for (int i = tid; i < 64 * nn; i += 256) {
                    const int c = i / nn, j = i - c * nn;
                    float o = 0.f;
                    for (int k = 0; k < a.K; ++k) o += ft[br * 64 + k] * w[c * FP + j + k];
                    ab[c * NS + j] = o;
                }
                __syncthreads();
            }
            float* tp = ct + br * 64 * KS * NS;
            {
                float* ring = g_cv + (long)br * 64 * RC;
                // the sub-chunk's own columns: into the ring for later pushes, and into the taps that read them now
                for (int i = tid; i < 64 * nn; i += 256) {
                    const int c = i / nn, j = i - c * nn;
                    st_agent(ring + (long)c * RC + ((int)(tbase + j) & (RC - 1)), ab[c * NS + j]);
                }
                const int m = 64 * KS * nn;
                for (int i = tid; i < m; i += 256) {
                    const int j = i % nn, q = i / nn, tap = q % KS, c = q / KS;
                    const int back = (KS - 1 - tap) * a.dil - j;
                    if (back <= 0) tp[(c * KS + tap) * NS + j] = ab[c * NS - back];
                }
            }
            __syncthreads();
            {
                // dilated causal conv: out[r][j] = sum_ci sum_tap W[r][ci][tap] * frame(j - (KS - 1 - tap) * dil)[ci]
                float acc[NS];
#pragma unroll
                for (int j = 0; j < NS; ++j) acc[j] = 0.f;
#pragma unroll
                for (int q = 0; q < 16 * KS; ++q) {
                    if ((q & 7) == 0) SE_TS_FENCE();
                    const float* wp = tp + (16 * p * KS + q) * NS;
#pragma unroll
                    for (int j = 0; j < NS; ++j) acc[j] += wreg[q] * wp[j];
                }
#pragma unroll
                for (int j = 0; j < NS; ++j) part[(p * 64 + r) * NS + j] = acc[j];
            }
            __syncthreads();
            if (br == 0) {          // the out-conv's weights fly under the gate product and the last cLN
                const float* wo = a.w_out + tid;
#pragma unroll
                for (int k = 0; k < 64; ++k) wreg[k] = wo[k * 256];
            }
            for (int i = tid; i < 64 * NS; i += 256) {
                const float t = part[i] + part[64 * NS + i] + part[2 * 64 * NS + i] + part[3 * 64 * NS + i];
                if (br) rg[i] = sigm_(t);
                else ab[i] = gated ? t * rg[i] : t;
            }
            __syncthreads();
        }
        // ---- out head: PReLU -> cLN -> W_out + residual
        cln(ab, nn, tbase, hp + 384, 2);
        {
            float acc[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) acc[j] = 0.f;
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                if ((k & 7) == 0) SE_TS_FENCE();
#pragma unroll
                for (int j = 0; j < NS; ++j) acc[j] += wreg[k] * ab[k * NS + j];
            }
            for (int j = 0; j < nn; ++j) yb[(long)tid * a.Tw + c0 + j] = acc[j] + xs[tid * NS + j];
        }
        // the next sub-chunk gathers ring columns written above: they went out as agent-scope (write-through) stores and
        // come back through agent-scope loads - waiting for the stores' acknowledgement is enough, an agent-scope FENCE
        // (L2 write-back + invalidate) cost ~10 us per block here
        if (c0 + NS < a.n) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
        __syncthreads();
    }
    if (tid < 6) g_carry[tid] = carry[tid];
}

// ---- a CHAIN of blocks in one launch ------------------------------------------------------------------------------------
// A kernel of this size costs ~4.7 us before its first instruction does anything useful (dispatch, cache maintenance at the
// kernel boundaries) and another ~1.5 us until its head parameters have arrived - a third to a half of a block's time at one
// frame per push.  The blocks of a TCM stack feed each other (6 per group in CTSNet, 8 per stack in G2Net / TaylorSENet), so
// ONE launch walks the whole chain frame by frame: every block is causal and advances its own state (rings, cLN sums) by
// one frame, the 256-channel column between two blocks never leaves LDS, all blocks' head parameters, FIR taps and cLN
// sums arrive in one batch at kernel entry, and the next block's in-conv weights are requested before the current block's
// output is written.
constexpr int TCM_CHAIN_MAX = 8;
struct TcmStreamBlk {
    const float *w_in, *w_l, *w_r, *w_out;
    TcmFusedHeads hd;
    int K, dil;
    char* state; long state_stride;
};
struct TcmChainArgs {
    const float* x; float* y;       // windows [B][256][Tw] of the chain's input / output, new frames in columns [H, H + n)
    int Tw, H, n; long t0;
    int nblk, fp_max;               // fp_max: widest FIR window of the chain (K - 1 + NS)
    TcmStreamBlk blk[TCM_CHAIN_MAX];
};

template <int KS>
__global__ __launch_bounds__(256) void tcm_chain_kernel(const TcmChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NB = a.nblk;
    float* xs = sm;                        // [256] the column that enters the current block (its residual)
    float* hb = xs + 256;                  // [64]  in-conv output
    float* part = hb + 64;                 // [4][64] partial sums of the four k-parts (waves)
    float* ab = part + 4 * 64;             // [64]  branch value
    float* rg = ab + 64;                   // [64]  gate
    float* mu = rg + 64;                   // mean, rstd
    double* ds = reinterpret_cast<double*>(mu + 2);            // [2] column sums
    double* carry_all = ds + 2;                                // [NB][8] running cLN sums (6 used)
    float* hp_all = reinterpret_cast<float*>(carry_all + 8 * TCM_CHAIN_MAX);      // [NB][9][64]
    float* ft_all = hp_all + TCM_CHAIN_MAX * 9 * 64;           // [NB][2][64] FIR taps
    float* ct = ft_all + TCM_CHAIN_MAX * 2 * 64;               // [2][64][KS] tap columns of the dilated conv
    float* wf = ct + 2 * 64 * KS;                              // [2][64][fp_max] FIR windows
    const float* xb = a.x + (long)b * 256 * a.Tw + a.H;
    float* yb = a.y + (long)b * 256 * a.Tw + a.H;
    const int r = tid & 63, p = tid >> 6;

    float wreg[64 > 16 * KS ? 64 : 16 * KS];
    {
        const float* w = a.blk[0].w_in + (long)(64 * p) * 64 + r;
#pragma unroll
        for (int k = 0; k < 64; ++k) wreg[k] = w[k * 64];
    }
    for (int bi = 0; bi < NB; ++bi) {
        const TcmStreamBlk& B = a.blk[bi];
        float* hp = hp_all + bi * 9 * 64;
        float* ft = ft_all + bi * 2 * 64;
        const bool gated = B.w_r != nullptr;
        if (tid < 64) {
            hp[tid] = B.hd.sL[tid]; hp[64 + tid] = B.hd.gL[tid]; hp[128 + tid] = B.hd.bL[tid];
            hp[384 + tid] = B.hd.sO[tid]; hp[448 + tid] = B.hd.gO[tid]; hp[512 + tid] = B.hd.bO[tid];
            if (gated) { hp[192 + tid] = B.hd.sR[tid]; hp[256 + tid] = B.hd.gR[tid]; hp[320 + tid] = B.hd.bR[tid]; }
            if (tid < B.K) {
                ft[tid] = B.hd.firL[tid];
                if (gated) ft[64 + tid] = B.hd.firR[tid];
            }
        }
        if (tid < 6) carry_all[bi * 8 + tid] = reinterpret_cast<const double*>(B.state + (long)b * B.state_stride)[tid];
    }

    for (int c0 = 0; c0 < a.n; ++c0) {
        const long tbase = a.t0 + c0;                  // stream index of this frame
        xs[tid] = xb[(long)tid * a.Tw + c0];
        for (int bi = 0; bi < NB; ++bi) {
            const TcmStreamBlk& B = a.blk[bi];
            const int KH = B.K > 0 ? B.K - 1 : 0, CH = (KS - 1) * B.dil;
            const int RF = ring_of(KH), RC = ring_of(CH), FP = a.fp_max;
            const bool gated = B.w_r != nullptr;
            const int nb = gated ? 2 : 1;
            float* hp = hp_all + bi * 9 * 64;
            const float* ft = ft_all + bi * 2 * 64;
            double* carry = carry_all + bi * 8;
            char* stb = B.state + (long)b * B.state_stride;
            float* g_fir = reinterpret_cast<float*>(stb + 64);
            float* g_cv = g_fir + (KH > 0 ? nb * 64 * RF : 0);
            // ---- one batch of global reads: FIR history, history taps of the dilated convs
            {
                const int n = nb * 64 * KH;
                for (int i0 = tid; i0 < n; i0 += 16 * 256) {
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int i = i0 + u * 256;
                        if (i < n) {
                            const int row = i / KH, col = i - row * KH;
                            v[u] = ld_agent(g_fir + (long)row * RF + ((int)(tbase - KH + col) & (RF - 1)));
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int i = i0 + u * 256;
                        if (i < n) wf[(i / KH) * FP + (i % KH)] = v[u];
                    }
                }
                const int m = nb * 64 * KS;
                for (int i = tid; i < m; i += 256) {
                    const int tap = i % KS, row = i / KS;
                    const int back = (KS - 1 - tap) * B.dil;
                    if (back > 0) ct[row * KS + tap] = ld_agent(g_cv + (long)row * RC + ((int)(tbase - back) & (RC - 1)));
                }
            }
            __syncthreads();           // xs of this block is complete
            {
                float acc = 0.f;
                const float* xv = xs + 64 * p;
#pragma unroll
                for (int k = 0; k < 64; ++k) {
                    if ((k & 7) == 0) SE_TS_FENCE();
                    acc += wreg[k] * xv[k];
                }
                part[p * 64 + r] = acc;
            }
            __syncthreads();
            if (tid < 64) hb[tid] = part[tid] + part[64 + tid] + part[128 + tid] + part[192 + tid];
            __syncthreads();
            // PReLU -> cumulative LayerNorm of ab [64] in place (one frame: wave 0 reduces, see tcm_stream_kernel)
            auto cln = [&](const float* prm, int which) __attribute__((always_inline)) {
                if (wave == 0) {
                    float t = ab[lane];
                    t = t >= 0.f ? t : prm[lane] * t;
                    ab[lane] = t;
                    double s1 = t, s2 = (double)t * t;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        s1 += __shfl_xor(s1, o, 64);
                        s2 += __shfl_xor(s2, o, 64);
                    }
                    if (lane == 0) {
                        const double s = carry[2 * which] + s1, q = carry[2 * which + 1] + s2;
                        const double cnt = 64.0 * (double)(tbase + 1), m = s / cnt;
                        const double var = (q - 2.0 * m * s) / cnt + m * m;
                        mu[0] = (float)m;
                        mu[1] = (float)(1.0 / sqrt(var + 1e-5));
                        carry[2 * which] = s;
                        carry[2 * which + 1] = q;
                    }
                }
                __syncthreads();
                if (tid < 64) ab[tid] = (ab[tid] - mu[0]) * mu[1] * prm[64 + tid] + prm[128 + tid];
                __syncthreads();
            };
            for (int br = gated ? 1 : 0; br >= 0; --br) {
                const float* wcv = br ? B.w_r : B.w_l;
#pragma unroll
                for (int q = 0; q < 16 * KS; ++q) wreg[q] = wcv[(long)((16 * p) * KS + q) * 64 + r];
                if (tid < 64) ab[tid] = hb[tid];
                __syncthreads();
                cln(hp + br * 192, br ? 1 : 0);
                if (B.K > 0) {          // ShareSepConv: one causal FIR shared by all channels
                    float* w = wf + br * 64 * FP;
                    float* ring = g_fir + (long)br * 64 * RF;
                    if (tid < 64) {
                        const float t = ab[tid];
                        w[tid * FP + KH] = t;
                        if (KH > 0) st_agent(ring + (long)tid * RF + ((int)tbase & (RF - 1)), t);
                    }
                    __syncthreads();
                    {
                        // 4 threads per channel walk a quarter of the taps each (K <= 64), folded through part
                        const int c = tid >> 2, qd = tid & 3;
                        float o = 0.f;
                        for (int k = qd; k < B.K; k += 4) o += ft[br * 64 + k] * w[c * FP + k];
                        o += __shfl_xor(o, 1, 64);
                        o += __shfl_xor(o, 2, 64);
                        if (qd == 0) ab[c] = o;
                    }
                    __syncthreads();
                }
                float* tp = ct + br * 64 * KS;
                {
                    float* ring = g_cv + (long)br * 64 * RC;
                    if (tid < 64) {
                        const float t = ab[tid];
                        st_agent(ring + (long)tid * RC + ((int)tbase & (RC - 1)), t);
                        tp[tid * KS + KS - 1] = t;          // the newest tap reads this frame
                    }
                }
                __syncthreads();
                {
                    float acc = 0.f;
#pragma unroll
                    for (int q = 0; q < 16 * KS; ++q) {
                        if ((q & 7) == 0) SE_TS_FENCE();
                        acc += wreg[q] * tp[16 * p * KS + q];
                    }
                    part[p * 64 + r] = acc;
                }
                __syncthreads();
                if (br == 0) {          // the out-conv's weights fly under the gate product and the last cLN
                    const float* wo = B.w_out + tid;
#pragma unroll
                    for (int k = 0; k < 64; ++k) wreg[k] = wo[k * 256];
                }
                if (tid < 64) {
                    const float t = part[tid] + part[64 + tid] + part[128 + tid] + part[192 + tid];
                    if (br) rg[tid] = sigm_(t);
                    else ab[tid] = gated ? t * rg[tid] : t;
                }
                __syncthreads();
            }
            // ---- out head: PReLU -> cLN -> W_out + residual
            cln(hp + 384, 2);
            {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 64; ++k) {
                    if ((k & 7) == 0) SE_TS_FENCE();
                    acc += wreg[k] * ab[k];
                }
                acc += xs[tid];
                // the in-conv weights of the next block (or of the first block for the next frame) start their trip now
                const bool more = bi + 1 < NB || c0 + 1 < a.n;
                if (more) {
                    const float* w = a.blk[bi + 1 < NB ? bi + 1 : 0].w_in + (long)(64 * p) * 64 + r;
#pragma unroll
                    for (int k = 0; k < 64; ++k) wreg[k] = w[k * 64];
                }
                __syncthreads();                    // every wave has read its part of xs (residual)
                if (bi + 1 < NB) xs[tid] = acc;     // the next block's input never leaves LDS
                else yb[(long)tid * a.Tw + c0] = acc;
            }
        }
        // the next frame gathers ring columns written above (agent-scope stores, read back by agent-scope loads: waiting for
        // the stores' acknowledgement is enough)
        if (c0 + 1 < a.n) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
        __syncthreads();
    }
    for (int bi = 0; bi < NB; ++bi)
        if (tid < 6) reinterpret_cast<double*>(a.blk[bi].state + (long)b * a.blk[bi].state_stride)[tid] = carry_all[bi * 8 + tid];
}

int ring_host(int need) {
    int r = 16;
    while (r < need + NS) r <<= 1;
    return r;
}

}  // namespace

TcmStreamW tcm_stream_build(const std::vector<float>& w_in, const std::vector<float>& w_left, const std::vector<float>* w_right,
                            const std::vector<float>& w_out, int ks) {
    // host layouts (DenseW): w_in [64][256], w_left / w_right [64][64][ks], w_out [256][64]
    TcmStreamW f;
    f.ks = ks;
    std::vector<float> t((size_t)256 * 64);
    for (int m = 0; m < 64; ++m)
        for (int k = 0; k < 256; ++k) t[(size_t)k * 64 + m] = w_in[(size_t)m * 256 + k];
    f.w_in = to_device(t);
    auto conv_t = [&](const std::vector<float>& w) {
        std::vector<float> u((size_t)64 * ks * 64);
        for (int m = 0; m < 64; ++m)
            for (int ci = 0; ci < 64; ++ci)
                for (int tap = 0; tap < ks; ++tap) u[(size_t)(ci * ks + tap) * 64 + m] = w[((size_t)m * 64 + ci) * ks + tap];
        return to_device(u);
    };
    f.w_l = conv_t(w_left);
    if (w_right) f.w_r = conv_t(*w_right);
    for (int m = 0; m < 256; ++m)
        for (int k = 0; k < 64; ++k) t[(size_t)k * 256 + m] = w_out[(size_t)m * 64 + k];
    f.w_out = to_device(t);
    return f;
}

void tcm_stream_free(TcmStreamW& f) {
    for (float** p : {&f.w_in, &f.w_l, &f.w_r, &f.w_out}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
}

bool tcm_stream_enabled() {
    static const bool on = !(getenv("SE_TCM_STREAM") && atoi(getenv("SE_TCM_STREAM")) == 0);
    return on;
}

// x, y: windows [B][256][H + n] of the current chunk (stream context); only the n new columns are read / written
void launch_tcm_stream(const TcmStreamW& f, const TcmFusedHeads& hd, const float* x, float* y, int dil, int K, hipStream_t s) {
    StreamCtx* cx = stream_ctx();
    SE_CHECK(cx && f.w_in, "launch_tcm_stream outside a frame-online chunk");
    const bool gated = f.w_r != nullptr;
    const int KH = K > 0 ? K - 1 : 0, CH = (f.ks - 1) * dil, nb = gated ? 2 : 1;
    const int RF = ring_host(KH), RC = ring_host(CH);
    const long stride = 64 + (long)nb * 64 * ((KH > 0 ? RF : 0) + RC) * sizeof(float);
    char* state = static_cast<char*>(cx->slot((size_t)cx->B * stride, s));
    cx->memo_src = nullptr;
    TcmStreamArgs a{x, y, cx->H + cx->n, cx->H, cx->n, cx->t0, f.w_in, f.w_l, f.w_r, f.w_out, hd, K, f.ks, dil, state, stride};
    SE_CHECK(K <= 64, "tcm_stream: FIR longer than 64 taps");
    const size_t lds = ((size_t)256 * NS + 64 * NS * 7 + 9 * 64 + 2 * 64 + 2 * NS) * sizeof(float) + (2 * NS + 8) * sizeof(double) +
                       (size_t)nb * 64 * (f.ks * NS + (KH + NS)) * sizeof(float);
    SE_CHECK(lds <= 150 * 1024, "tcm_stream: windows too large for LDS");
    static bool seen[64] = {};
    if (first_on_device(seen)) {
        SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tcm_stream_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tcm_stream_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    }
    if (f.ks == 3) hipLaunchKernelGGL(tcm_stream_kernel<3>, dim3(cx->B), dim3(256), lds, s, a);
    else if (f.ks == 5) hipLaunchKernelGGL(tcm_stream_kernel<5>, dim3(cx->B), dim3(256), lds, s, a);
    else SE_CHECK(false, "tcm_stream: dilated conv with 3 or 5 taps expected");
    SE_HIP(hipGetLastError());
}

bool tcm_chain_enabled() {
    static const bool on = !(getenv("SE_TCM_CHAIN") && atoi(getenv("SE_TCM_CHAIN")) == 0);
    return on && tcm_stream_enabled();
}

void launch_tcm_chain(const TcmStreamW* const* f, const TcmFusedHeads* hd, const int* dil, const int* K, int nblk, const float* x,
                      float* y, hipStream_t s) {
    StreamCtx* cx = stream_ctx();
    SE_CHECK(cx && nblk >= 1 && nblk <= TCM_CHAIN_MAX, "launch_tcm_chain: 1..8 blocks inside a frame-online chunk");
    TcmChainArgs a{};
    a.x = x; a.y = y; a.Tw = cx->H + cx->n; a.H = cx->H; a.n = cx->n; a.t0 = cx->t0; a.nblk = nblk;
    const int ks = f[0]->ks;
    int fp = 1;
    for (int i = 0; i < nblk; ++i) {
        SE_CHECK(f[i]->w_in && f[i]->ks == ks, "launch_tcm_chain: blocks of one chain share the dilated conv's tap count");
        SE_CHECK(K[i] <= 64, "tcm_stream: FIR longer than 64 taps");
        const bool gated = f[i]->w_r != nullptr;
        const int KH = K[i] > 0 ? K[i] - 1 : 0, CH = (ks - 1) * dil[i], nb = gated ? 2 : 1;
        const int RF = ring_host(KH), RC = ring_host(CH);
        const long stride = 64 + (long)nb * 64 * ((KH > 0 ? RF : 0) + RC) * sizeof(float);
        char* state = static_cast<char*>(cx->slot((size_t)cx->B * stride, s));       // same slot layout as launch_tcm_stream
        a.blk[i] = TcmStreamBlk{f[i]->w_in, f[i]->w_l, f[i]->w_r, f[i]->w_out, hd[i], K[i], dil[i], state, stride};
        fp = std::max(fp, KH + NS);
    }
    cx->memo_src = nullptr;
    a.fp_max = fp;
    const size_t lds = (size_t)(256 + 64 * 8 + 2) * sizeof(float) + (2 + 8 * TCM_CHAIN_MAX) * sizeof(double) +
                       (size_t)TCM_CHAIN_MAX * 11 * 64 * sizeof(float) + (size_t)2 * 64 * (ks + fp) * sizeof(float) + 64;
    SE_CHECK(lds <= 150 * 1024, "tcm_chain: windows too large for LDS");
    static bool seen[64] = {};
    if (first_on_device(seen)) {
        SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tcm_chain_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(tcm_chain_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    }
    if (ks == 3) hipLaunchKernelGGL(tcm_chain_kernel<3>, dim3(cx->B), dim3(256), lds, s, a);
    else if (ks == 5) hipLaunchKernelGGL(tcm_chain_kernel<5>, dim3(cx->B), dim3(256), lds, s, a);
    else SE_CHECK(false, "tcm_stream: dilated conv with 3 or 5 taps expected");
    SE_HIP(hipGetLastError());
}

}  // namespace se
