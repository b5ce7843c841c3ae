// Tap-table implicit-GEMM convolution on f32 MFMA for gfx950 (see gemmconv.h).
//
// Block = 256 threads = 4 waves, output tile BM (channels) x BN (consecutive t) of one (z, b, q) row.
// K is walked in chunks of CI_C input channels x all taps.  Per chunk the block stages
//   As[k][m]    : KCp x BM packed weights (float4 global loads, m contiguous)
//   Bs[ci][r][w]: the raw input patch (CI_C channels x nrows distinct frequency rows x (BN + dt span)),
//                 NOT an im2col copy - every tap reads the same patch at a shifted offset,
// double-buffered through registers (global loads for chunk c+1 are in flight while chunk c runs on MFMA).
// MFMA operand reads are ds_read_b32 with lanes 0-31 on 32 consecutive floats of row k and lanes 32-63
// on row k+1: conflict-free for both operands at any tap offset.
#include "gemmconv.h"
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstring>

namespace se {

typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    switch (act) {
        case ACT_PRELU: return v >= 0.f ? v : slope * v;
        case ACT_ELU: return v > 0.f ? v : expm1f(v);
        case ACT_SOFTPLUS: return v > 20.f ? v : log1pf(expf(v));
        case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case ACT_TANH: return tanhf(v);
        case ACT_RELU: return fmaxf(v, 0.f);
        default: return v;
    }
}
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gc_kernel(const GCParams p) {
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr int A_IT = (GC_MAX_KCP * BM / 4 + 255) / 256;
    constexpr int ROW_IT = 6;
    constexpr int W_IT = 3;
    static_assert(ROW_IT * W_IT <= GC_MAX_BLD, "prefetch budget");
    static_assert(WM * WN == 4, "4 waves");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int rows = p.CI_C * p.nrows;
    const int As_sz = p.KCp * BM;
    const int Bs_sz = rows * p.Wp;
    float* As = smem;
    float* Bs = smem + 2 * As_sz;
    int* koff = reinterpret_cast<int*>(Bs + 2 * Bs_sz);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int hi = lane >> 5;

    // ---- block -> (z, b, q, t-tile, m-tile); XCD-aware: the m-tiles of one activation patch and
    //      neighbouring patches share an XCD's L2 (block id i runs on XCD i % 8).
    int lid;
    {
        const int nblk = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, slot = id >> 3, q8 = nblk >> 3, r8 = nblk & 7;
        lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    }
    const int mt = lid % p.n_mtiles;
    int rest = lid / p.n_mtiles;
    const int ttile = rest % p.n_ttiles;
    rest /= p.n_ttiles;
    const int q = rest % p.Q;
    rest /= p.Q;
    const int b = rest % p.B;
    const int z = rest / p.B;
    const int t0 = ttile * BN;
    const int m0 = mt * BM;

    const float* __restrict__ Ag = p.A + (long)z * p.A_z + m0;
    const float* __restrict__ s0 = p.src0 ? p.src0 + (long)z * p.src0_z + (long)b * p.s0_b : nullptr;
    const float* __restrict__ s1 = p.src1 ? p.src1 + (long)z * p.src1_z + (long)b * p.s1_b : nullptr;
    const int Cin = p.C0 + p.C1;

    // tap offset table for one chunk (identical for every chunk)
    for (int k = tid; k < p.KCp; k += 256) {
        int off = 0;
        if (k < p.KC) {
            const int cil = k / p.ntaps, j = k - cil * p.ntaps;
            off = cil * (p.nrows * p.Wp) + p.tap_row[j] * p.Wp + (p.tap_dt[j] - p.dtmin);
        }
        koff[k] = off;
    }

    float4 preA[A_IT];
    float preB[ROW_IT * W_IT];
    const int nA4 = p.KCp * (BM / 4);

    auto load_chunk = [&](int chunk) {
        const float* Ac = Ag + (long)chunk * p.KCp * p.Mp;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + i * 256;
            if (idx < nA4) {
                const int k = idx / (BM / 4), m4 = idx % (BM / 4);
                preA[i] = *reinterpret_cast<const float4*>(Ac + (long)k * p.Mp + m4 * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < ROW_IT; ++i) {
            const int rr = wave + 4 * i;
            if (rr < rows) {
                const int cil = rr / p.nrows, r = rr - cil * p.nrows;
                const int ci = chunk * p.CI_C + cil;
                const int f = q * p.si + p.row_df[r];
                const bool rowok = (ci < Cin) && (f >= 0) && (f < p.Fin);
                const float* rp = nullptr;
                if (rowok) rp = (ci < p.C0) ? s0 + (long)ci * p.s0_c + (long)f * p.s0_f
                                            : s1 + (long)(ci - p.C0) * p.s1_c + (long)f * p.s1_f;
#pragma unroll
                for (int j = 0; j < W_IT; ++j) {
                    const int w = lane + 64 * j;
                    const int t = t0 + p.dtmin + w;
                    float v = 0.f;
                    if (rowok && w < p.Wp && t >= 0 && t < p.Tin) v = rp[t];
                    preB[i * W_IT + j] = v;
                }
            }
        }
    };
    auto store_chunk = [&](int buf) {
        float* Ad = As + buf * As_sz;
        float* Bd = Bs + buf * Bs_sz;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int idx = tid + i * 256;
            if (idx < nA4) *reinterpret_cast<float4*>(Ad + idx * 4) = preA[i];
        }
#pragma unroll
        for (int i = 0; i < ROW_IT; ++i) {
            const int rr = wave + 4 * i;
            if (rr < rows) {
#pragma unroll
                for (int j = 0; j < W_IT; ++j) {
                    const int w = lane + 64 * j;
                    if (w < p.Wp) Bd[rr * p.Wp + w] = preB[i * W_IT + j];
                }
            }
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wm = wave / WN, wn = wave % WN;
    const int am = wm * (TM * 32) + l31;      // A column base inside the tile
    const int bn = wn * (TN * 32) + l31;      // B column base inside the tile

    if (p.nchunks > 0) {
        load_chunk(0);
        store_chunk(0);
    }
    __syncthreads();

    for (int c = 0; c < p.nchunks; ++c) {
        const int buf = c & 1;
        if (c + 1 < p.nchunks) load_chunk(c + 1);
        const float* Ab = As + buf * As_sz + hi * BM + am;
        const float* Bb = Bs + buf * Bs_sz + bn;
        const int* kb = koff + hi;
        const int npair = p.KCp >> 1;
#pragma unroll 2
        for (int kp = 0; kp < npair; ++kp) {
            const int ob = kb[2 * kp];
            float a[TM], bb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Ab[(2 * kp) * BM + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bb[j] = Bb[ob + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
        if (c + 1 < p.nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---------------------------------------------------------------- epilogue
    const float* __restrict__ bias = p.bias ? p.bias + (long)z * p.bias_z : nullptr;
    const int fo = q * p.so + p.po;
    float* __restrict__ dst = p.dst + (long)z * p.dst_z + (long)b * p.d_b + (long)fo * p.d_f;

    if (p.epi == EPI_ACT || p.epi == EPI_ADD) {
        const float* __restrict__ res =
            (p.epi == EPI_ADD) ? p.aux + (long)z * p.aux_z + (long)b * p.x_b + (long)fo * p.x_f : nullptr;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int t = t0 + wn * (TN * 32) + j * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (m < p.M && t < p.Tout) {
                        float v = acc[i][j][r] + (bias ? bias[m] : 0.f);
                        v = act_apply(v, p.act, p.slope ? p.slope[m] : 0.f);
                        if (res) v += res[(long)m * p.x_c + t];
                        dst[(long)m * p.d_c + t] = v;
                    }
                }
            }
    } else if (p.epi == EPI_GLU) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int t = t0 + wn * (TN * 32) + j * 32 + l31;
#pragma unroll
                for (int r2 = 0; r2 < 8; ++r2) {
                    const int r = 2 * r2;
                    const int m = m0 + wm * (TM * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;   // even row
                    if (m + 1 < p.M && t < p.Tout) {
                        const float a = acc[i][j][r] + (bias ? bias[m] : 0.f);
                        const float g = acc[i][j][r + 1] + (bias ? bias[m + 1] : 0.f);
                        dst[(long)(m >> 1) * p.d_c + t] = a * sigmoidf_(g);
                    }
                }
            }
    } else {   // EPI_LSTM
        const float* __restrict__ gx = p.aux + (long)z * p.aux_z + (long)b * p.x_b + (long)fo * p.x_f;
        float* __restrict__ cell = p.cell + (long)z * p.cell_z + (long)b * p.d_b + (long)fo * p.d_f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int t = t0 + wn * (TN * 32) + j * 32 + l31;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int m = m0 + wm * (TM * 32) + i * 32 + 8 * g4 + 4 * hi;   // row of gate i of unit m/4
                    if (m + 3 < p.M && t < p.Tout) {
                        const float* gp = gx + (long)m * p.x_c + t;
                        const float gi = acc[i][j][4 * g4 + 0] + gp[0];
                        const float gf = acc[i][j][4 * g4 + 1] + gp[p.x_c];
                        const float gg = acc[i][j][4 * g4 + 2] + gp[2 * p.x_c];
                        const float go = acc[i][j][4 * g4 + 3] + gp[3 * p.x_c];
                        const long oi = (long)(m >> 2) * p.d_c + t;
                        const float cprev = p.first_step ? 0.f : cell[oi];
                        const float cn = sigmoidf_(gf) * cprev + sigmoidf_(gi) * tanhf(gg);
                        cell[oi] = cn;
                        dst[oi] = sigmoidf_(go) * tanhf(cn);
                    }
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static size_t gc_lds_bytes(const GCParams& p, int BM) {
    return (size_t)(2 * p.KCp * BM + 2 * p.CI_C * p.nrows * p.Wp) * 4 + (size_t)p.KCp * 4;
}

GCPlan gc_make_plan(int M, int Cin, const TapSpec& taps, const std::vector<float>& w, const std::vector<float>& bias,
                    const std::vector<float>& slope, int act, int epi, int si, int so, int po, int tout_hint, int Z) {
    SE_CHECK(taps.ntaps >= 1 && taps.ntaps <= GC_MAX_TAPS, "tap count");
    SE_CHECK((long)w.size() == (long)Z * M * Cin * taps.ntaps, "weight size mismatch in gc_make_plan");
    GCPlan pl;
    GCParams& p = pl.p;
    pl.BM = M >= 96 ? 128 : (M >= 48 ? 64 : 32);
    pl.BN = (tout_hint >= 96 || pl.BM == 32) ? 128 : 64;
    // distinct rows / dt span
    int dtmin = taps.dt[0], dtmax = taps.dt[0];
    std::vector<int> rows;
    for (int j = 0; j < taps.ntaps; ++j) {
        dtmin = std::min(dtmin, taps.dt[j]);
        dtmax = std::max(dtmax, taps.dt[j]);
        if (std::find(rows.begin(), rows.end(), taps.df[j]) == rows.end()) rows.push_back(taps.df[j]);
    }
    std::sort(rows.begin(), rows.end());
    SE_CHECK((int)rows.size() <= GC_MAX_ROWS, "too many distinct frequency rows");
    p.ntaps = taps.ntaps;
    p.nrows = (int)rows.size();
    p.dtmin = dtmin;
    p.Wp = pl.BN + (dtmax - dtmin);
    SE_CHECK(p.Wp <= 192, "time span of taps too wide for one patch");
    for (int r = 0; r < p.nrows; ++r) p.row_df[r] = (signed char)rows[r];
    for (int j = 0; j < taps.ntaps; ++j) {
        p.tap_row[j] = (unsigned char)(std::find(rows.begin(), rows.end(), taps.df[j]) - rows.begin());
        p.tap_dt[j] = (signed char)taps.dt[j];
    }
    // chunking: largest CI_C within the staging budgets
    const int wit = (p.Wp + 63) / 64;
    int cic = 1;
    for (int c = 1; c <= std::max(Cin, 1); ++c) {
        int kcp = (c * taps.ntaps + 1) & ~1;
        int rit = (c * p.nrows + 3) / 4;
        if (kcp <= GC_MAX_KCP && rit <= 6 && rit * wit <= GC_MAX_BLD) cic = c;
    }
    p.CI_C = cic;
    p.KC = cic * taps.ntaps;
    p.KCp = (p.KC + 1) & ~1;
    SE_CHECK(p.KCp <= GC_MAX_KCP, "single-channel chunk exceeds K budget");
    p.nchunks = Cin > 0 ? (Cin + cic - 1) / cic : 0;
    p.M = M;
    p.Mp = ((M + pl.BM - 1) / pl.BM) * pl.BM;
    p.n_mtiles = p.Mp / pl.BM;
    p.act = act;
    p.epi = epi;
    p.si = si;
    p.so = so;
    p.po = po;
    p.Z = Z;
    p.C0 = Cin;
    p.C1 = 0;
    // pack weights: [z][chunk][k_local][Mp]
    const size_t per_z = (size_t)std::max(p.nchunks, 1) * p.KCp * p.Mp;
    std::vector<float> packed(per_z * Z, 0.f);
    for (int z = 0; z < Z; ++z)
        for (int m = 0; m < M; ++m)
            for (int ci = 0; ci < Cin; ++ci)
                for (int j = 0; j < taps.ntaps; ++j) {
                    const int chunk = ci / cic, cil = ci % cic;
                    const size_t dst = z * per_z + ((size_t)chunk * p.KCp + (cil * taps.ntaps + j)) * p.Mp + m;
                    packed[dst] = w[(((size_t)z * M + m) * Cin + ci) * taps.ntaps + j];
                }
    p.A_z = (long)per_z;
    pl.dA = to_device(packed);
    p.A = pl.dA;
    if (!bias.empty()) {
        SE_CHECK((long)bias.size() == (long)Z * M, "bias size");
        pl.dBias = to_device(bias);
        p.bias = pl.dBias;
        p.bias_z = M;
    }
    if (!slope.empty()) {
        SE_CHECK((long)slope.size() == M, "slope size");
        pl.dSlope = to_device(slope);
        p.slope = pl.dSlope;
    }
    return pl;
}

void gc_free_plan(GCPlan& pl) {
    if (pl.dA) (void)hipFree(pl.dA);
    if (pl.dBias) (void)hipFree(pl.dBias);
    if (pl.dSlope) (void)hipFree(pl.dSlope);
    pl.dA = pl.dBias = pl.dSlope = nullptr;
}

template <int BM, int BN, int WM, int WN>
static void gc_launch_t(const GCParams& p, hipStream_t stream) {
    const size_t lds = gc_lds_bytes(p, BM);
    static bool attr_set = false;
    if (!attr_set) {
        SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gc_kernel<BM, BN, WM, WN>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const long nblk = (long)p.Z * p.B * p.Q * p.n_ttiles * p.n_mtiles;
    SE_CHECK(nblk > 0 && nblk < (1L << 31), "grid size");
    hipLaunchKernelGGL((gc_kernel<BM, BN, WM, WN>), dim3((unsigned)nblk), dim3(256), lds, stream, p);
    SE_HIP(hipGetLastError());
}

void gc_launch(const GCPlan& pl, GCParams p, hipStream_t stream) {
    p.n_ttiles = (p.Tout + pl.BN - 1) / pl.BN;
    if (p.epi == EPI_LSTM && p.first_step) p.nchunks = 0;
    if (pl.BM == 128 && pl.BN == 128) gc_launch_t<128, 128, 2, 2>(p, stream);
    else if (pl.BM == 64 && pl.BN == 128) gc_launch_t<64, 128, 2, 2>(p, stream);
    else if (pl.BM == 32 && pl.BN == 128) gc_launch_t<32, 128, 1, 4>(p, stream);
    else if (pl.BM == 128 && pl.BN == 64) gc_launch_t<128, 64, 4, 1>(p, stream);
    else if (pl.BM == 64 && pl.BN == 64) gc_launch_t<64, 64, 2, 2>(p, stream);
    else SE_CHECK(false, "no gemmconv tile config");
}

}  // namespace se
