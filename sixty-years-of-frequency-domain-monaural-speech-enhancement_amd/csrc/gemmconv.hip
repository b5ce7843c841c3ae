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
#include "fastmath.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <type_traits>

namespace se {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// compile-time loop: indices are constants at the IR level, so per-thread arrays always live in registers
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// (v_rcp_f32, 1 ulp: `__frcp_rn` is a correctly rounded division - ten vector instructions on the matrix pipe's clock, five
// times per LSTM cell and twice per gated value)
__device__ __forceinline__ float fsig_(float x) { return fm_sigmoid(x); }
__device__ __forceinline__ float ftanh_(float x) { return fm_tanh(x); }
__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    switch (act) {
        case ACT_PRELU: return v >= 0.f ? v : slope * v;
        // (ELU's negative branch: hardware exp2 - libm's expm1f is ~40 instructions per value in CRN's / GCRN's conv epilogues;
        // near zero, where exp(v) - 1 cancels, the series v + v^2 / 2 is exact to 2e-10)
        case ACT_ELU: return v > 0.f ? v : fm_expm1(v);
        case ACT_SOFTPLUS: return fm_softplus(v);      // (hardware exp2 / log2: absolute error < 1e-7)
        case ACT_SIGMOID: return fsig_(v);
        case ACT_TANH: return ftanh_(v);
        case ACT_RELU: return fmaxf(v, 0.f);
        default: return v;
    }
}
__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }
// GCParams::fz - one frame of the branch interaction (the same expressions as model_uformer.hip: uf_fusion_kernel)
__device__ __forceinline__ float gc_fuse1(float v, float* __restrict__ zr, long im, long so = 0) {
    const float re = zr[0], ii = zr[im];
    const float cm = fm_sqrt(fmaxf(re * re + ii * ii, 1.1920928955078125e-07f));      // v_sqrt_f32 (1 ulp)
    const float s = fsig_(v);
    zr[0] = re + s;
    zr[im] = ii + s;
    if (so) zr[so] = (re + s) + (ii + s);      // the sum plane of a three-plane tensor (GCParams::fz_s)
    return v + fsig_(cm);
}

// 16 B per lane global -> LDS copy without a register round trip (global_load_lds_dwordx4: lane l lands at
// lds_base + 16*l, lds_base wave-uniform in M0).  Issued from inline asm on purpose: through the builtin the compiler
// has to assume that the DMA write may alias the LDS reads of the MFMA loop and parks an s_waitcnt vmcnt(0) right
// behind the issue, i.e. a full memory round trip per K chunk.  The kernel's own ordering makes that wait unnecessary:
// vmcnt retires in order and the activation-patch loads of the same chunk are consumed behind an s_waitcnt vmcnt(0)
// (the compiler's wait for its newest load drains every older or younger VMEM op) before the __syncthreads() that
// publishes the buffer.
__device__ __forceinline__ void gc_dma16(const float* g, float* lds_wave_base) {
    const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(g) : "memory");
}
// 4 B per lane, only for the lanes whose `pred` is non-zero (EXEC is narrowed inside the asm block and restored):
// lane l lands at lds_base + 4*l, masked lanes leave their LDS word untouched (the patch buffers are zeroed once per
// block, so a never-written word IS the zero of the padding).
__device__ __forceinline__ void gc_dma4_masked(const float* g, unsigned lds_wave_base, unsigned pred) {
    unsigned long long saved;
    asm volatile(
        "s_mov_b64 %0, exec\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %1\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %3, off\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(saved)
        : "v"(pred), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base)), "v"(g)
        : "memory", "vcc");
}
// 16 B per lane variant (pointwise layers: a lane's 4 consecutive frames are one aligned group of the patch row)
__device__ __forceinline__ void gc_dma16_masked(const float* g, unsigned lds_wave_base, unsigned pred) {
    unsigned long long saved;
    asm volatile(
        "s_mov_b64 %0, exec\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %1\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, off\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(saved)
        : "v"(pred), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base)), "v"(g)
        : "memory", "vcc");
}
// the same three with a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane BYTE offset: no 64-bit vector address per
// staged group (two VALU instructions and two registers each, live across the matrix loop once the issue is spread over it)
__device__ __forceinline__ void gc_dma16_s(const float* base, unsigned voff, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte), "v"(voff), "s"(base) : "memory");
}
// (the validity test of a slot - bit `mask` of the lane's `vbits` - inside the asm block: as C++ the scheduler hoisted the
// thirteen `vbits & mask` of a chunk to the top of the matrix loop and kept them alive)
__device__ __forceinline__ void gc_dma4_masked_s(const float* base, unsigned voff, unsigned lds_byte, unsigned vbits, unsigned mask) {
    unsigned long long saved;
    unsigned tmp;
    asm volatile(
        "s_mov_b64 %0, exec\n\t"
        "v_and_b32_e32 %1, %6, %2\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %1\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %4, %5\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(saved), "=&v"(tmp)
        : "v"(vbits), "s"(lds_byte), "v"(voff), "s"(base), "s"(mask)
        : "memory", "vcc");
}
__device__ __forceinline__ void gc_dma16_masked_s(const float* base, unsigned voff, unsigned lds_byte, unsigned vbits, unsigned mask) {
    unsigned long long saved;
    unsigned tmp;
    asm volatile(
        "s_mov_b64 %0, exec\n\t"
        "v_and_b32_e32 %1, %6, %2\n\t"
        "v_cmp_ne_u32_e32 vcc, 0, %1\n\t"
        "s_and_b64 exec, exec, vcc\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %5\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(saved), "=&v"(tmp)
        : "v"(vbits), "s"(lds_byte), "v"(voff), "s"(base), "s"(mask)
        : "memory", "vcc");
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const float*)p;
}

// Per-thread staging descriptors are chunk-invariant: element e = (row rr = wave + 4*i, column w = lane + 64*j)
// of the activation patch reads  base_chunk[boff[e]]  (always an in-bounds address) and is zeroed when its validity
// bit is clear, so the per-chunk staging code is one load + one select per element with a uniform 64-bit base.
// Phase timing for tuning builds (make gcbench_timing: -DGC_TIMING): wave 0 of every block accumulates s_memtime
// deltas per phase into p.timing[0..5] (prologue, load issue, mfma, vmcnt wait, barrier, epilogue), [6] = blocks.
#ifdef GC_TIMING
#define GC_T(SLOT)                                                               \
    {                                                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();            \
        if (tid == 0) tacc[SLOT] += now_ - tlast;                                \
        tlast = __builtin_amdgcn_s_memtime();                                    \
    }
#else
#define GC_T(SLOT)
#endif

// sum over the 8 lanes that share (lane >> 3): quad swaps, then the mirrored half row; every lane ends with the total
__device__ __forceinline__ float dpp_add8(float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));   // row_half_mirror
    return x;
}

// RES (tiny launches, e.g. the few-frame chunks of the frame-online mode): every K chunk of a source has its own staging
// buffer, all are requested at once and waited for once - the block's life is then 1-2 memory round trips instead of one
// per chunk (16 for a 64 x 320 TCM conv, whose matrix work is a few hundred cycles).
// TRIM: the launch stages 16 B groups under taps that look ahead in time (GC_TRIM_TAIL).  A variant of its own: the mere
// presence of the LDS stores in the K loop costs the other launches 2-10 % (gcbench, 32- / 64-row tiles most).
// NRM: the sources are raw conv outputs, their InstanceNorm + PReLU is applied to the B-operand fragments (GCParams::nrm0 / nrm1)
// FLATW > 0 (GCParams::flat_upr): the tile's columns are BN / 32 consecutive 32-frame UNITS of the batch rows flattened, each staged
// with its own halo - FLATW = LDS columns per unit and patch row (32: pointwise layers, 36: causal taps that look back <= 4 frames)
template <int BM, int BN, int WM, int WN, int EPI, bool RES = false, bool TRIM = false, bool FZ = false, bool NRM = false, int FLATW = 0>
__global__ __launch_bounds__(256, RES ? 1 : (FZ || (BM >= 128 && BN >= 256)) ? (BM <= 32 ? 4 : 2) : gc_blocks_per_cu(BM)) void gc_kernel(const GCParams p) {   // (FZ: room for the prefetched pair; 128 x 256: 128 accumulators per lane)
#ifdef GC_TIMING
    unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
    constexpr int TM = BM / (WM * 32);
    constexpr int TN = BN / (WN * 32);
    constexpr bool FLAT = FLATW > 0;
    constexpr int UW = FLAT ? FLATW : 32;            // LDS columns between the 32-column sub-tiles of a wave
    constexpr int UPT = BN / 32;                     // units per tile
    static_assert(!FLAT || (!RES && !TRIM && !FZ && EPI != EPI_LSTM), "flattened column tiles: plain offline variants only");
    // small-M tiles do little MFMA work per staged K row, so they stage twice the K depth per barrier to keep the
    // global-load latency under the matrix work
    constexpr int KCP_MAX = gc_kcp_max(BM);
    constexpr int A_IT = (KCP_MAX * BM / 4 + 255) / 256;
    // patch slots staged per thread (flat index e = tid + 256 * i); the 256-column tile is only launched with 16 B groups
    constexpr int NB = BN >= 256 ? (BM >= 128 ? 6 : 5) : gc_bld_max(BM);      // (128 x 256: pointwise layers, 24 rows x 64 groups)
    static_assert(WM * WN == 4, "4 waves");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int rows = p.CI_C * p.nrows;
    const int npatch = rows * p.Wp;                  // floats of one staged activation patch
    const int bit = (npatch + 255) >> 8;             // patch elements per thread (uniform)
    const bool pw4 = p.pw4 != 0;                     // the patch is staged in 16 B groups (decided per launch, gc_launch)
    // 16 B groups under taps that look ahead in time, rows that are not whole groups: see GC_TRIM_TAIL (block-uniform)
    constexpr bool fixt = TRIM;
    const int bit4 = (npatch / 4 + 255) >> 8;
    const int nA4 = p.KCp * (BM / 4);
    const int ait = (nA4 + 255) >> 8;                // float4 groups of the weight chunk per thread (uniform)
    const int As_sz = ait * 1024;                    // padded so that every thread stores unconditionally
    const int Bs_sz = bit * 256;                     // padded: every thread stores unconditionally
    const int nbuf = RES ? p.nbuf : 2;               // double-buffered staging (RES: one buffer per chunk of a source)
    float* As = smem;
    float* Bs = smem + nbuf * As_sz;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const unsigned wave_u = __builtin_amdgcn_readfirstlane((unsigned)wave);      // in an SGPR for the DMA's LDS bases
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
    // (FLAT: the output rows q of one unit group are neighbours in the block order - rows q and q + 1 of a stride-2 layer share
    // an input row, and with the B * 13 / 8 tiles of a whole batch between them that row came from HBM twice: the round-6 PMC pass
    // of TaylorSENet read 1.44 GB per launch of the 64 x 256 tile against 0.96 GB with plain tiles)
    const int ttile = FLAT ? (rest / p.Qt) % p.n_ttiles : rest % p.n_ttiles;
    if (FLAT) rest = (rest % p.Qt) + (rest / (p.Qt * p.n_ttiles)) * p.Qt;
    else rest /= p.n_ttiles;
    // two-row tiles (p.qt2, gc_launch): the tile's columns are 2 output rows x BN / 2 frames - neighbouring output rows share
    // most of their input rows (5 taps at stride 2: 7 staged rows instead of 10), so a chunk stages ~30 % fewer bytes for the
    // same matrix work.  The waves of the upper column half take the second row; only block-level scalars differ.
    const bool qt2 = !FLAT && p.qt2 != 0;
    const int q = (rest % p.Qt) << (qt2 ? 1 : 0);
    rest /= p.Qt;
    // FLAT: `b` is the batch row of the tile's FIRST unit (the 64-bit bases are taken there, a unit of the next row adds its
    // batch stride to the 32-bit offsets), `t0` the first frame of that unit; every unit has its own entry in `utab`
    const int u0 = FLAT ? ttile * UPT : 0;
    const int b = FLAT ? u0 / p.flat_upr : rest % p.B;
    const int z = FLAT ? rest : rest / p.B;
    const int t0 = FLAT ? p.t_base + (u0 - b * p.flat_upr) * 32 : p.t_base + ttile * (qt2 ? BN / 2 : BN);
    const int m0 = mt * BM;

    const float* __restrict__ Ag = p.A + (long)z * p.A_z + m0;

    // patch offsets of the K rows of one chunk (identical for every chunk): lane half `hi` serves row 2*kp + hi of
    // k-pair kp.  They live in registers indexed at compile time (the MFMA loop is fully unrolled over k-pairs), so no
    // operand address depends on an LDS read or a v_readlane (both measured slower, tools/mfmabench.cpp).
    constexpr int NPAIR = KCP_MAX / 2;
    constexpr bool KOFF_REGS = (BM >= 128 || TM * TN >= 4);      // small tiles keep the table in LDS: registers buy occupancy there
    int koffv[KOFF_REGS ? NPAIR : 1];
    // the whole device table (frequency rows, taps, per-K-row patch offsets) is staged once into LDS: every later
    // lookup is a short LDS read instead of a dependent global load in the block prologue
    // LDS map: [staging buffers | epilogue strips (aliased)] [tap table] [per-row epilogue parameters]
    constexpr int STRIPS = 4 * (TM * 32) * 36;       // floats of the four per-wave epilogue transposition strips
    // (GCParams::cstats: 2 048 floats of partial column sums behind the strips, see `cpart` - the host sizes the area the same way)
    int* tabl = reinterpret_cast<int*>(smem + max(nbuf * (As_sz + Bs_sz), STRIPS + (p.cstats ? 2048 : 0)));
    int* koff_lds = tabl + GC_TAB_KOFF;
    float* ep = reinterpret_cast<float*>(tabl + GC_TAB_KOFF + KCP_MAX + 8);      // [4 * BM]
    // NRM: per input channel {scale, shift, slope - 1, x0} of this batch row, and per K row of a staged chunk the same with
    // scale = shift = 0 where the row's frequency tap lies outside the plane (double-buffered with the chunks)
    // (FLAT: a tile's units may lie in two batch rows - both rows' channel parameters, and per buffer both rows' K-row parameters)
    constexpr int NRMB = FLAT ? 2 : 1;
    floatx4* nrmC = reinterpret_cast<floatx4*>(ep + 4 * BM);                     // [NRMB][GC_NRM_MAXC]
    floatx4* nrmK = nrmC + NRMB * GC_NRM_MAXC;                                   // [2][NRMB][KCP_MAX] + 2 (the pipeline reads one pair ahead)
    // FLAT: per unit k of the tile {batch row - b, first frame, valid, -}
    int* utab = reinterpret_cast<int*>(nrmC + (NRM ? NRMB * GC_NRM_MAXC + 2 * NRMB * KCP_MAX + 2 : 0));
    // GCParams::cstats: partial column sums of the block's tile, [WM][4 row groups][BN][2] floats (<= 8 KB), combined in a fixed
    // order after a barrier.  Behind the epilogue strips INSIDE the (by then dead) staging area: as 8 KB of their own they pushed the
    // 64 x 256 tile from three to two workgroups per CU (+ 6 ... 23 % per launch in the first form of this epilogue)
    float* cpart = smem + STRIPS;
    GC_T(7);      /* kernel entry .. index decode */
    // one barrier for both block-wide LDS initialisations: the tap table (its global load is in flight while the patch
    // buffers are cleared) and the zeros of the padding (masked DMA lanes never touch their LDS words again)
    const int tabv = (tid < GC_TAB_KOFF + KCP_MAX + 8) ? p.tab[tid] : 0;
    // per-row epilogue parameters (bias, PReLU slope, GLU post scale / shift) are fetched here, under the rest of the
    // prologue, and parked in LDS: the epilogue then has no global load and no block barrier in front of its stores
    const int fo = q * p.so + p.po;
    const float* __restrict__ bias = (fo < p.pad_lo) ? p.bias_pad : (p.bias ? p.bias + (long)z * p.bias_z : nullptr);
    float epv[4] = {0.f, 0.f, 1.f, 0.f};
    if (EPI != EPI_LSTM && tid < BM) {
        const int m = min(m0 + tid, p.M - 1);
        epv[0] = bias ? bias[m] : 0.f;
        // (identity and ReLU ride the PReLU form of the epilogue with slopes 1 and 0)
        const float lin_slope = p.act == ACT_NONE ? 1.f : 0.f;
        if (EPI == EPI_CMB) {
            epv[0] = p.post_shift[(long)z * p.ps_z + m];
            epv[1] = p.slope ? p.slope[(long)z * p.ps_z + m] : 1.f;
            epv[2] = p.post_scale[(long)z * p.ps_z + m];
        } else if (EPI != EPI_GLU) {
            epv[1] = (p.act == ACT_PRELU && p.slope) ? p.slope[m] : lin_slope;
        } else if (tid < BM / 2) {
            const int oc = min((m0 >> 1) + tid, (p.M >> 1) - 1);
            epv[1] = (p.act == ACT_PRELU && p.slope) ? p.slope[oc] : lin_slope;
            epv[2] = p.post_scale ? p.post_scale[oc] : 1.f;
            epv[3] = p.post_scale ? p.post_shift[oc] : 0.f;
        }
    }
    if constexpr (NRM) {
        const int ct_ = p.C0 + p.C1;
        if (tid < NRMB * ct_) {
            const int db_ = tid >= ct_ ? 1 : 0, ch_ = tid - db_ * ct_;
            const bool s1 = ch_ >= p.C0;
            const float* __restrict__ np_ = s1 ? p.nrm1 : p.nrm0;
            floatx4 v = {1.f, 0.f, 0.f, 0.f};
            if (np_) v = reinterpret_cast<const floatx4*>(np_)[(long)min(b + db_, p.B - 1) * (s1 ? p.C1 : p.C0) + (s1 ? ch_ - p.C0 : ch_)];
            nrmC[db_ * GC_NRM_MAXC + ch_] = v;
        }
    }
    if constexpr (FLAT) {
        if (tid < UPT) {
            const int u_ = u0 + tid, bk_ = u_ / p.flat_upr;
            utab[4 * tid + 0] = bk_ - b;
            utab[4 * tid + 1] = p.t_base + (u_ - bk_ * p.flat_upr) * 32;
            utab[4 * tid + 2] = u_ < p.flat_units ? 1 : 0;
            utab[4 * tid + 3] = p.tlen ? p.tlen[min(bk_, p.B - 1)] : 0x7fffffff;      // the unit's row: its own frame count (ragged batches)
        }
    }
    for (int i = tid; i < nbuf * Bs_sz / 4; i += 256) reinterpret_cast<floatx4*>(Bs)[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (tid < GC_TAB_KOFF + KCP_MAX + 8) tabl[tid] = tabv;
    if (EPI != EPI_LSTM && tid < BM) {
        ep[tid] = epv[0];
        ep[BM + tid] = epv[1];
        if (EPI == EPI_GLU && tid < BM / 2) {
            ep[2 * BM + tid] = epv[2];
            ep[3 * BM + tid] = epv[3];
        }
        if (EPI == EPI_CMB) ep[2 * BM + tid] = epv[2];
    }
    __syncthreads();
    if constexpr (KOFF_REGS) {
        static_for<NPAIR>([&](auto KP) {
            constexpr int kp = decltype(KP)::value;
            koffv[kp] = koff_lds[2 * kp + hi];
        });
    }

    // NRM, chunk-invariant part of the per-K-row parameters: thread k < KCP_MAX serves K row k = (cil, tap) - its channel inside
    // the chunk and whether its frequency row lies inside the plane (block-uniform per row; padded K rows count as outside);
    // threads < CI_C * nrows also own one patch row each for the left-pad frames of the first time tile
    int nk_cil = 0, pb_cil = 0;
    bool nk_ok = false;
    const int npadL = (NRM && !FLAT) ? max(0, -(t0 + p.dtmin)) : 0;       // staged columns in front of frame 0 (block-uniform, multiple of 4; FLAT: per unit, see GC_LOAD_CHUNK)
    if constexpr (NRM) {
        if (tid < NRMB * KCP_MAX) {
            const int kr_ = tid & (KCP_MAX - 1);      // K row (FLAT: threads KCP_MAX.. serve the second batch row's copy)
            nk_cil = kr_ / p.ntaps;
            const int f = q * p.si + tabl[tabl[GC_MAX_ROWS + kr_ - nk_cil * p.ntaps]];
            nk_ok = kr_ < p.KC && f >= 0 && f < p.Fin;
        }
        pb_cil = tid / p.nrows;
    }

    // ---- chunk-invariant staging descriptors (all staging loops have uniform bounds: no exec masking)
    unsigned aoff[A_IT];
    static_for<A_IT>([&](auto I) {
        constexpr int i = decltype(I)::value;
        int idx = tid + i * 256;
        if (idx >= nA4) idx = nA4 - 1;
        aoff[i] = 4u * (unsigned)((idx / (BM / 4)) * p.Mp + (idx % (BM / 4)) * 4);      // bytes
    });
    unsigned boff[NB];
    unsigned vbits = 0;      // bit e set when patch element e lies inside the tensor (else it is a zero of the padding)
    unsigned fullbits = 0;   // wave-uniform: bit e set when that holds for all 64 lanes of slot e

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // 64 x 256 tile: the wave -> column-strip assignment rotates with the time tile, so that the idle waves of a partly
    // filled last tile (T = 401: 145 of 256 columns) sit on a different SIMD in every workgroup of a CU instead of all on SIMD 3
    const int wm = wave / WN, wn = (WN == 4 && BN >= 256) ? ((wave + ttile) & 3) : wave % WN;
    constexpr int WNT = WN >= 2 ? WN / 2 : 1;          // waves per output row of a two-row tile
    const int wq = (qt2 && WN >= 2) ? wn / WNT : 0;     // output row of this wave inside the tile
    const int wt = (qt2 && WN >= 2) ? wn % WNT : wn;    // its 32 * TN-frame strip inside the row
    const int am = wm * (TM * 32) + l31;      // A column base inside the tile
    const int bn = FLAT ? wn * (TN * UW) + l31 : wq * p.qq_off + wt * (TN * 32) + l31;      // B column base inside the staged patch
    // 32-column sub-tiles of this wave with a frame below Tout (wave-uniform; FLAT: units below the end of the batch)
    const int jact = FLAT ? max(0, min(TN, p.flat_units - u0 - wn * TN))
                          : (q + wq >= p.Q) ? 0 : max(0, min(TN, (p.Tout - t0 - wt * (TN * 32) + 31) >> 5));

    int gchunk = 0;          // global chunk counter (weights are packed segment after segment)
    int buf = 0;

#define GC_MAKE_DESC(LIM)                                                                          \
    {                                                                                              \
        const int lim_ = (LIM);                                                                    \
        vbits = 0;                                                                                 \
        /* host-built descriptors of the flat patch slots fe = tid + 256 e: w | r << 12 | cil << 16 | staged << 31.    */ \
        /* All NB loads are issued as one batch into block-local registers before the first use: fused with their use */ \
        /* the compiler waited for each load in turn (13 serial global round trips per block prologue).               */ \
        unsigned dsc_[NB];                                                                         \
        {                                                                                          \
            const unsigned* __restrict__ dp_ = (pw4 ? p.desc4 : p.desc) + tid;                     \
            static_for<NB>([&](auto E) {                                                           \
                constexpr int e = decltype(E)::value;                                              \
                dsc_[e] = dp_[256 * e];                                                            \
            });                                                                                    \
        }                                                                                          \
        static_for<NB>([&](auto E) {                                                               \
            constexpr int e = decltype(E)::value;                                                  \
            const unsigned d_ = dsc_[e];                                                           \
            const int w = d_ & 0xfff, r = (d_ >> 12) & 0xf, cil = (d_ >> 16) & 0x7fff;             \
            const bool staged = (d_ >> 31) != 0;                                                   \
            const int f = q * p.si + tabl[staged ? r : 0];                                         \
            /* FLAT: column w of the patch row belongs to unit w / UW - its own batch row, first frame and validity */ \
            const int uk_ = FLAT ? min(w / UW, UPT - 1) : 0;                                       \
            const int t = FLAT ? utab[4 * uk_ + 1] + p.dtmin + (w - uk_ * UW) : t0 + p.dtmin + w;  \
            const bool uok_ = !FLAT || utab[4 * uk_ + 2] != 0;                                     \
            const unsigned ub_ = (FLAT && uok_) ? (unsigned)utab[4 * uk_] * sb32 : 0u;             \
            const int fc = f < 0 ? 0 : (f >= p.Fin ? p.Fin - 1 : f);                               \
            const int tc = t < 0 ? 0 : (t >= p.Tin ? p.Tin - 1 : t);                               \
            const int cc = cil < lim_ ? cil : lim_ - 1;                                            \
            boff[e] = staged ? 4u * (ub_ + (unsigned)cc * sc32 + (unsigned)fc * sf32 + (unsigned)tc) : 0u;   /* bytes */ \
            /* pw4: the slot is a group of 4 frames; never straddles frame 0, may straddle Tin (see gc_launch)   */ \
            vbits |= (staged && uok_ && (cil < lim_) && (f >= 0) && (f < p.Fin) && (t >= 0) && (t < p.Tin)) ? (1u << e) : 0u; \
            /* non-causal taps: a 16 B group that straddles the end of its row is trimmed in LDS after it lands (bits 16..) */ \
            if (fixt) vbits |= (staged && (t < p.Tin) && (t + 4 > p.Tin)) ? (0x10000u << e) : 0u;   \
        });                                                                                        \
        /* slots whose 64 lanes are all inside the tensor (every slot of an interior tile): staged without the EXEC detour */ \
        fullbits = 0;      /* (SE_GC_DBG=32: always the masked form, for the A/B measurement) */ \
        {                                                                                          \
            if (!(p.dbg & 32)) static_for<NB>([&](auto E) {                                                           \
                constexpr int e = decltype(E)::value;                                              \
                if (e < bit4 && __builtin_amdgcn_ballot_w64((vbits >> e) & 1u) == ~0ull) fullbits |= 1u << e; \
            });                                                                                    \
        }                                                                                          \
    }
// the frames >= Tin of a straddling group hold the head of the next row: the thread that staged the group zeroes them once
// its own DMAs have landed (behind GC_WAIT_CHUNK, before the barrier that publishes the buffer)
#define GC_TRIM_TAIL(BUF)                                                                          \
    if (fixt && (vbits >> 16)) {                                                                   \
        float* fb_ = Bs + (BUF) * Bs_sz + 4 * tid;                                                 \
        const int nv_ = p.Tin & 3;                                                                 \
        static_for<NB>([&](auto E) {                                                               \
            constexpr int e = decltype(E)::value;                                                  \
            if (vbits & (0x10000u << e)) {                                                         \
                _Pragma("unroll") for (int k = 1; k < 4; ++k)                                      \
                    if (k >= nv_) fb_[1024 * e + k] = 0.f;                                         \
            }                                                                                      \
        });                                                                                        \
    }
#define GC_LOAD_CHUNK(CH, BUF)                                                                     \
    {                                                                                              \
        if constexpr (NRM) {                                                                       \
            /* the buffer is free (its readers passed the last barrier): its K-row parameters and, in the first time tile, */ \
            /* the left-pad frames as the value the normalisation maps to zero (the masked DMA never touches them)        */ \
            const int cb_ = (CH) * p.CI_C, co_ = seg ? p.C0 : 0;                                   \
            if (tid < NRMB * KCP_MAX) {                                                            \
                const int db_ = tid >= KCP_MAX ? 1 : 0;                                            \
                floatx4 v_ = nrmC[db_ * GC_NRM_MAXC + min(cb_ + nk_cil, Cseg - 1) + co_];          \
                if (!nk_ok) {                                                                      \
                    v_[0] = 0.f;                                                                   \
                    v_[1] = 0.f;                                                                   \
                }                                                                                  \
                nrmK[(BUF) * (NRMB * KCP_MAX) + tid] = v_;                                         \
            }                                                                                      \
            if (npadL > 0 && tid < rows) {                                                         \
                const float x0_ = nrmC[min(cb_ + pb_cil, Cseg - 1) + co_][3];                      \
                float* d_ = Bs + (BUF) * Bs_sz + tid * p.Wp;                                       \
                for (int w_ = 0; w_ < npadL; w_ += 4) *reinterpret_cast<floatx4*>(d_ + w_) = floatx4{x0_, x0_, x0_, x0_}; \
            }                                                                                      \
            if constexpr (FLAT) {      /* every unit that starts a row: its halo columns (frames < 0) as that row's x0 */ \
                if (tid < rows && p.dtmin < 0) {                                                   \
                    _Pragma("unroll") for (int k_ = 0; k_ < UPT; ++k_) {                           \
                        if (utab[4 * k_ + 1] + p.dtmin < 0 && utab[4 * k_ + 2]) {                  \
                            const float x0_ = nrmC[utab[4 * k_] * GC_NRM_MAXC + min(cb_ + pb_cil, Cseg - 1) + co_][3]; \
                            float* d_ = Bs + (BUF) * Bs_sz + tid * p.Wp + k_ * UW;                 \
                            for (int w_ = 0; w_ < -(utab[4 * k_ + 1] + p.dtmin); w_ += 4)          \
                                *reinterpret_cast<floatx4*>(d_ + w_) = floatx4{x0_, x0_, x0_, x0_}; \
                        }                                                                          \
                    }                                                                              \
                }                                                                                  \
            }                                                                                      \
        }                                                                                          \
        /* both operands go global -> LDS by DMA: no staging registers, no ds_write phase; the activation patch first */ \
        /* (HBM latency), the weights (L2-resident) behind it                                                        */ \
        const float* __restrict__ Bc = sbase + (long)(CH) * p.CI_C * s_c;                          \
        const unsigned bb = __builtin_amdgcn_readfirstlane(lds_addr(Bs + (BUF) * Bs_sz));          \
        const unsigned bl = bb + 256u * wave_u, bl4 = bb + 1024u * wave_u;                         \
        if (pw4 && fullbits == (1u << bit4) - 1u) {     /* every group of the chunk inside the tensor: no EXEC detours */ \
            static_for<NB>([&](auto E) {                                                           \
                constexpr int e = decltype(E)::value;                                              \
                if (e < bit4) gc_dma16_s(Bc, boff[e], bl4 + 4096u * e);                            \
            });                                                                                    \
        } else if (pw4) {                                                                          \
            static_for<NB>([&](auto E) {                                                           \
                constexpr int e = decltype(E)::value;                                              \
                if (e < bit4) gc_dma16_masked_s(Bc, boff[e], bl4 + 4096u * e, vbits, 1u << e);     \
            });                                                                                    \
        } else {                                                                                   \
            static_for<NB>([&](auto E) {                                                           \
                constexpr int e = decltype(E)::value;                                              \
                if (e < bit) gc_dma4_masked_s(Bc, boff[e], bl + 1024u * e, vbits, 1u << e);         \
            });                                                                                    \
        }                                                                                          \
        const float* __restrict__ Ac = Ag + (long)(gchunk + (CH)) * p.KCp * p.Mp;                  \
        /* wave-uniform LDS base of this wave's 1 KB slice */                                      \
        const unsigned Adw = __builtin_amdgcn_readfirstlane(lds_addr(As + (BUF) * As_sz)) + 1024u * wave_u; \
        static_for<A_IT>([&](auto I) {                                                             \
            constexpr int i = decltype(I)::value;                                                  \
            if (i < ait) gc_dma16_s(Ac, aoff[i], Adw + 4096u * i);                                 \
        });                                                                                        \
    }
// the DMA writes are invisible to the compiler: drain them by hand before the barrier that publishes the buffer
#define GC_WAIT_CHUNK() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

    for (int seg = 0; seg < 2; ++seg) {
        const int Cseg = seg ? p.C1 : p.C0;
        if (Cseg <= 0) continue;
        const float* __restrict__ sbase =
            (seg ? p.src1 + (long)z * p.src1_z + (long)b * p.s1_b : p.src0 + (long)z * p.src0_z + (long)b * p.s0_b);
        const long s_c = seg ? p.s1_c : p.s0_c, s_f = seg ? p.s1_f : p.s0_f;
        const unsigned sc32 = (unsigned)s_c, sf32 = (unsigned)s_f;      // patch offsets inside one chunk fit 32 bits
        const unsigned sb32 = (unsigned)(seg ? p.s1_b : p.s0_b);        // (FLAT: a unit of the next batch row; gc_launch checks the range)
        const int nch = (Cseg + p.CI_C - 1) / p.CI_C;
        const int tail = Cseg - (nch - 1) * p.CI_C;       // channels in the last chunk

        GC_T(9);      /* Bs zero + sync */
        GC_MAKE_DESC(nch > 1 ? p.CI_C : tail);
        GC_T(10);     /* descriptors */
        if constexpr (RES) {
            buf = 0;
            for (int c2 = 0; c2 < min(nch, nbuf); ++c2) {
                if (c2 + 1 == nch && nch > 1 && tail != p.CI_C) GC_MAKE_DESC(tail);
                GC_LOAD_CHUNK(c2, c2);
            }
        } else {
            GC_LOAD_CHUNK(0, buf);       // `buf` is free: the previous segment's last chunk was read from buf ^ 1
        }
        GC_T(1);
        GC_WAIT_CHUNK();
        GC_TRIM_TAIL(buf);
        GC_T(3);
        __syncthreads();
        GC_T(4);

        for (int c = 0; c < nch; ++c) {
            if constexpr (RES) {
                if (c > 0 && buf == nbuf) {      // next group of resident chunks
                    __syncthreads();
                    for (int c2 = c; c2 < min(nch, c + nbuf); ++c2) {
                        if (c2 + 1 == nch && tail != p.CI_C) GC_MAKE_DESC(tail);
                        GC_LOAD_CHUNK(c2, c2 - c);
                    }
                    GC_WAIT_CHUNK();
                    __syncthreads();
                    buf = 0;
                }
            }
            // (round 4 also tried the next chunk's DMAs one per matrix-instruction group instead of as a batch up front - gcbench_timing
            // shows the batch stalling a wave at issue for half as long as its matrix work takes.  Measured on every tile, with a
            // clean build as the baseline, it was a loss: the other workgroups of the CU already cover one another's issue stalls,
            // and the slots' code - sixteen asm blocks per matrix path - cost the 64 x 64 / 128 x 32 / 32 x 128 tiles 20 - 40 % at small
            // batches and 2 - 6 % at batch 256.  DESIGN.md 3.1)
            if (!RES && c + 1 < nch && !(p.dbg & 1)) {
                if (__builtin_expect(c + 2 == nch && tail != p.CI_C, 0)) GC_MAKE_DESC(tail);      // (cold code out of the loop's way)
                GC_LOAD_CHUNK(c + 1, buf ^ 1);
            }
            GC_T(1);
            // ---- MFMA over the staged chunk: two k-pairs (8 MFMAs at TM = TN = 2) per operand fetch
            const float* Ab = As + buf * As_sz + hi * BM + am;
            const float* Bb = Bs + buf * Bs_sz + bn;
            const floatx4* nkb = nrmK + buf * (NRMB * KCP_MAX) + hi;       // (NRM) parameters of K row 2 kp + hi
            // (FLAT) the batch row (0 / 1 relative to the tile's first) of each of this wave's units, as offsets into nrmK
            int nkd[FLAT && NRM ? TN : 1];
            if constexpr (FLAT && NRM) {
#pragma unroll
                for (int j = 0; j < TN; ++j) nkd[j] = utab[4 * min(wn * TN + j, UPT - 1)] * KCP_MAX;
            }
            const int npair = p.KCp >> 1;
            // software pipeline, depth 1: the operands of k-pair kp+1 are fetched (ds_read2_b32) before the MFMAs of
            // k-pair kp issue; sched_group_barrier pins "2 DS reads, then 4 MFMAs" so the LDS latency sits under
            // 256 cycles of matrix work.  Reads one pair past the chunk (valid LDS, result unused).
#define GC_FETCH(KP, AR, BR)                                                                       \
    {                                                                                              \
        int o_;                                                                                    \
        if constexpr (KOFF_REGS) o_ = koffv[(KP) < NPAIR ? (KP) : NPAIR - 1];                      \
        else o_ = koff_lds[2 * (KP) + hi];                                                         \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) AR[i] = Ab[(2 * (KP)) * BM + i * 32];       \
        _Pragma("unroll") for (int j = 0; j < JN; ++j) BR[j] = Bb[o_ + j * UW];                    \
        if constexpr (NRM && !FLAT) {                                                              \
            const floatx4 pr_ = nkb[2 * (KP)];                                                     \
            _Pragma("unroll") for (int j = 0; j < JN; ++j) {                                       \
                const float t_ = fmaf(BR[j], pr_[0], pr_[1]);                                      \
                BR[j] = fmaf(fminf(t_, 0.f), pr_[2], t_);                                          \
            }                                                                                      \
        }                                                                                          \
        if constexpr (NRM && FLAT) {                                                               \
            _Pragma("unroll") for (int j = 0; j < JN; ++j) {                                       \
                const floatx4 pr_ = nkb[2 * (KP) + nkd[j]];                                        \
                const float t_ = fmaf(BR[j], pr_[0], pr_[1]);                                      \
                BR[j] = fmaf(fminf(t_, 0.f), pr_[2], t_);                                          \
            }                                                                                      \
        }                                                                                          \
    }
#define GC_MMA(AR, BR)                                                                             \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                 \
        _Pragma("unroll") for (int j = 0; j < JN; ++j)                                             \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AR[i], BR[j], acc[i][j], 0, 0, 0);
            // JN = 32-column sub-tiles of this wave that hold at least one frame < Tout: in the last time tile of a row
            // (e.g. 17 of 128 columns at T = 401) most waves own only padding and skip the matrix work altogether
            auto mma_chunk = [&](auto JN_) {
                constexpr int JN = decltype(JN_)::value;
                float ax[TM], bx[JN], ay[TM], by[JN];
                GC_FETCH(0, ax, bx);
                static_for<NPAIR / 2>([&](auto KP2) {
                    constexpr int kp = 2 * decltype(KP2)::value;
                    // (the test against a copy the compiler cannot see through: as loop invariants of the chunk loop the eight
                    // tests were hoisted, their masks ran out of SGPRs and every one came back through two v_readlane - vector
                    // instructions on the matrix pipe's clock - per two k-pairs; now one s_cmp each)
                    int np_ = npair;
                    asm volatile("" : "+s"(np_));
                    if (kp < np_) {
                        // (NRM: one more DS read - the K rows' parameters - and three vector instructions per B value, placed
                        // behind the matrix instructions of the k-pair in flight)
                        constexpr int NDS = (TM + 1) / 2 + (JN + 1) / 2 + (NRM ? (FLAT ? JN : 1) : 0);
                        GC_FETCH(kp + 1, ay, by);
                        __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
                        GC_MMA(ax, bx);
                        __builtin_amdgcn_sched_group_barrier(0x008, TM * JN, 0);
                        if constexpr (NRM) __builtin_amdgcn_sched_group_barrier(0x002, 3 * JN, 0);
                        GC_FETCH(kp + 2, ax, bx);
                        __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
                        GC_MMA(ay, by);
                        __builtin_amdgcn_sched_group_barrier(0x008, TM * JN, 0);
                        if constexpr (NRM) __builtin_amdgcn_sched_group_barrier(0x002, 3 * JN, 0);
                    }
                });
            };
            if (!(p.dbg & 4)) {
                // (the 4-tile wave keeps ONE matrix path: with three, the register allocator moved its 128 accumulators between
                // the paths' own ranges through scratch memory - a partly filled last tile is rare on the layers that use it)
                if (jact == 1 && TN == 2) mma_chunk(std::integral_constant<int, 1>{});
                else if (jact >= 1) mma_chunk(std::integral_constant<int, TN>{});
            }
#undef GC_FETCH
#undef GC_MMA
            GC_T(2);
            if constexpr (RES) {
                buf += 1;
            } else {
                GC_WAIT_CHUNK();
                GC_TRIM_TAIL(buf ^ 1);
                GC_T(3);
                __syncthreads();
                GC_T(4);
                buf ^= 1;
            }
        }
        if constexpr (RES) __syncthreads();      // every chunk has been read: the next source (or the epilogue strips) may overwrite
        gchunk += nch;
    }
#undef GC_MAKE_DESC
#undef GC_LOAD_CHUNK
#undef GC_WAIT_CHUNK
#undef GC_TRIM_TAIL

    // ---------------------------------------------------------------- epilogue
    if (p.dbg & 8) return;
    GC_T(0);
    const int fow = fo + wq * p.so;      // output row of this wave (two-row tiles)
    float* __restrict__ dst = p.dst + (long)z * p.dst_z + (long)b * p.d_b + (long)fow * p.d_f;

    // The activated tile is transposed through LDS inside each wave so that a lane stores 16 B runs along t (4x fewer,
    // 4x wider stores than the MFMA accumulator layout gives).  The strips alias the staging buffers (the K loop's last
    // barrier has retired every reader) and are private to a wave: DS operations of one wave execute in order, so the
    // write -> read-back hand-over needs a wave-level fence only, no block barrier.
    constexpr int OROWS = (EPI == EPI_GLU) ? TM * 16 : TM * 32;   // output rows of one wave's strip
    // statistics epilogue only where an InstanceNorm can follow (64-channel layers): not in the 128-row EPI_ACT tile, whose
    // register budget (3 workgroups per CU) is the tightest and which carries the DCCRN bench
    constexpr bool GC_STATS = (EPI == EPI_GLU) || (EPI == EPI_ACT && BM == 64);
    constexpr int OST = 32 + 4;                                    // LDS row stride of the strip (one 32-column MFMA tile wide)
    float* strip = smem + wave * (TM * 32 * OST);
#define GC_WAVE_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
    if (EPI == EPI_ACT || EPI == EPI_ADD || EPI == EPI_MUL || EPI == EPI_GLU || EPI == EPI_CMB) {
        const int mw = wm * (TM * 32) + 4 * hi;                 // first tile row of this lane
        const int lr = lane >> 3, lc = (lane & 7) * 4;           // read-back role: (row within 8, 4 consecutive t)
        const int mo0 = (EPI == EPI_GLU ? (m0 >> 1) : m0) + wm * OROWS;      // first output row of the strip
        // (an odd row count leaves the second row of the last two-row tile outside the plane: nothing of it is stored)
        const int Mo = (q + wq < p.Q) ? ((EPI == EPI_GLU) ? (p.M >> 1) : p.M) : 0;
        // p.tlen (ragged batches, layers in front of operators that look ahead): frames >= tlen[b] of the output are zeros
        const int tvalid0 = p.tlen ? p.tlen[b] : 0x7fffffff;
        const float* __restrict__ res =
            (EPI == EPI_ADD || EPI == EPI_MUL || EPI == EPI_CMB) ? p.aux + (long)z * p.aux_z + (long)b * p.x_b + (long)fow * p.x_f : nullptr;
        const float cmb_sg = (EPI == EPI_CMB && ((p.cmb_neg >> z) & 1)) ? -1.f : 1.f;
        // EPI_CMB with the sum plane: this launch finishes R, reads the finished I and writes S = R + I beside it
        const float* __restrict__ cmbi = (EPI == EPI_CMB && p.cmb_s) ? p.cmb_i + (long)b * p.d_b + (long)fow * p.d_f : nullptr;
        float* __restrict__ cmbs = (EPI == EPI_CMB && p.cmb_s) ? p.cmb_s + (long)b * p.d_b + (long)fow * p.d_f : nullptr;
        float* __restrict__ delu = (EPI == EPI_GLU && p.dst_elu) ? p.dst_elu + (long)b * p.d_b + (long)fow * p.d_f : nullptr;      // GCParams::dst_elu
        float* __restrict__ fzb = FZ ? p.fz + (long)b * p.fz_b + (long)fow * p.fz_f : nullptr;      // FZ: GCParams::fz (its own instantiations: the
                                                                                                      // extra live registers of the epilogue spill in the 128-row tile otherwise)
        // (one column tile per call, its index a compile-time constant: with the interaction operands in the body the unroller
        // gave up on a `for j` and the accumulators, indexed by a run-time j, went to scratch memory)
        auto epi_tile = [&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            if (j > 0) GC_WAVE_FENCE();                          // the previous column tile has been read back
            // FLAT: this column tile is one unit - its own batch row (an offset on every per-row pointer), first frame and validity
            const int uk = FLAT ? min(wn * TN + j, UPT - 1) : 0;
            const long dbJ = FLAT ? utab[4 * uk] : 0;
            const int tj0 = FLAT ? utab[4 * uk + 1] : t0 + wt * (TN * 32) + j * 32;
            const int MoJ = (!FLAT || utab[4 * uk + 2]) ? Mo : 0;
            // frames >= the row's own count are stored as zeros and do not count in the statistics riders (ragged batches)
            const int tvalid = FLAT ? utab[4 * uk + 3] : tvalid0;
            const int tstat = min(p.Tout, tvalid);
            float* __restrict__ dstJ = dst + dbJ * p.d_b;
            const float* __restrict__ resJ = res ? res + dbJ * p.x_b : nullptr;
            const float* __restrict__ cmbiJ = cmbi ? cmbi + dbJ * p.d_b : nullptr;
            float* __restrict__ cmbsJ = cmbs ? cmbs + dbJ * p.d_b : nullptr;
            float* __restrict__ deluJ = delu ? delu + dbJ * p.d_b : nullptr;
            // residual / interaction operands of this column tile: all of a wave's loads issued here, ahead of the transposition
            // (one load -> wait -> store per 8 rows left an EPI_ADD tile waiting on 8 HBM round trips in a row: the pointwise
            // layers of Uformer's conformer ran at 0.22 of the matrix peak)
            constexpr bool PRE = (EPI == EPI_ADD || EPI == EPI_MUL || EPI == EPI_CMB);
            const int tgp = tj0 + lc;
            floatx4 rvp[PRE ? OROWS / 8 : 1], zre[FZ ? OROWS / 8 : 1], zim[FZ ? OROWS / 8 : 1], ivp[EPI == EPI_CMB ? OROWS / 8 : 1];
            if ((PRE || FZ) && tgp + 3 < p.Tout && MoJ > 0) {
#pragma unroll
                for (int it = 0; it < OROWS / 8; ++it) {
                    const int m = min(mo0 + it * 8 + lr, MoJ - 1);      // clamped: rows past M are loaded, not used
                    if (PRE) rvp[it] = *reinterpret_cast<const floatx4*>(resJ + (long)m * p.x_c + tgp);
                    if (EPI == EPI_CMB && cmbiJ) ivp[it] = *reinterpret_cast<const floatx4*>(cmbiJ + (long)m * p.d_c + tgp);
                    if (FZ) {
                        zre[it] = *reinterpret_cast<const floatx4*>(fzb + (long)m * p.fz_c + tgp);
                        zim[it] = *reinterpret_cast<const floatx4*>(fzb + (long)m * p.fz_c + tgp + p.fz_im);
                    }
                }
            }
            // The activation is chosen ONCE per column tile, not per value: as `act_apply(v, p.act, ..)` inside the unrolled loops
            // every one of the 32 - 64 values of a tile carried the whole switch - libm's expm1f / log1pf / tanhf expansions - and the
            // plain PReLU path hopped over them value by value: 22 500 instructions (180 KB) per kernel, a multiple of the
            // instruction cache, fetched again by every workgroup.  LIN: identity / ReLU / PReLU as one `v >= 0 ? v : s v`.
            auto write_strip = [&](auto ACT_) __attribute__((always_inline)) {
                constexpr int ACT = decltype(ACT_)::value;      // -1: the linear family, -2: identity (no slope read), else the activation itself
                auto actf = [&](float v, float sl) __attribute__((always_inline)) -> float {
                    if constexpr (ACT == -2) return v;
                    else if constexpr (ACT < 0) return v >= 0.f ? v : sl * v;
                    else return act_apply(v, ACT, sl);
                };
                // (round 6) a lane's 16 rows of a 32 x 32 tile are four groups of four consecutive rows: their parameters come as
                // 16 B / 8 B LDS reads - 8 instead of 32 (plain) and 16 instead of 40 (gated) per tile and lane
                typedef float floatx2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (EPI == EPI_CMB) {        // raw products: scale / shift / PReLU follow the sum with k1 in the read-back
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int dm = i * 32 + (r & 3) + 8 * (r >> 2);
                            strip[(4 * hi + dm) * OST + l31] = acc[i][j][r];
                        }
                    } else if (EPI != EPI_GLU) {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const int dm0 = i * 32 + 8 * g4;
                            const floatx4 b4 = *reinterpret_cast<const floatx4*>(ep + mw + dm0);
                            floatx4 s4 = {1.f, 1.f, 1.f, 1.f};
                            if constexpr (ACT != -2) s4 = *reinterpret_cast<const floatx4*>(ep + BM + mw + dm0);
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float v = acc[i][j][4 * g4 + k] + b4[k];
                                strip[(4 * hi + dm0 + k) * OST + l31] = actf(v, s4[k]);
                            }
                        }
                    } else {
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {      // rows dm0 .. dm0 + 3 = (value, gate) x 2 outputs ol0, ol0 + 1
                            const int dm0 = i * 32 + 8 * g4;
                            const int ol0 = (mw + dm0) >> 1;
                            const floatx4 b4 = *reinterpret_cast<const floatx4*>(ep + mw + dm0);
                            const floatx2 sc2 = *reinterpret_cast<const floatx2*>(ep + 2 * BM + ol0);
                            const floatx2 sh2 = *reinterpret_cast<const floatx2*>(ep + 3 * BM + ol0);
                            floatx2 sl2 = {1.f, 1.f};
                            if constexpr (ACT != -2) sl2 = *reinterpret_cast<const floatx2*>(ep + BM + ol0);
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const float a = acc[i][j][4 * g4 + 2 * k] + b4[2 * k];
                                const float g = acc[i][j][4 * g4 + 2 * k + 1] + b4[2 * k + 1];
                                float v = a * fsig_(g);          // hardware exp + reciprocal (as in the LSTM cells): the libm pair was ~15 % of a small-K GLU tile
                                v = v * sc2[k] + sh2[k];
                                strip[(((4 * hi + dm0) >> 1) + k) * OST + l31] = actf(v, sl2[k]);
                            }
                        }
                    }
                }
            };
            switch (p.act) {
                case ACT_NONE: write_strip(std::integral_constant<int, -2>{}); break;       // (raw layers: the norm-folded U^2-Net levels, k1 planes, projections)
                case ACT_RELU:
                case ACT_PRELU: write_strip(std::integral_constant<int, -1>{}); break;      // (slopes 0 / the layer's: see `ep`)
                case ACT_ELU: write_strip(std::integral_constant<int, ACT_ELU>{}); break;
                case ACT_SOFTPLUS: write_strip(std::integral_constant<int, ACT_SOFTPLUS>{}); break;
                case ACT_SIGMOID: write_strip(std::integral_constant<int, ACT_SIGMOID>{}); break;
                default: write_strip(std::integral_constant<int, ACT_TANH>{}); break;
            }
            GC_WAVE_FENCE();
            // read back as rows: 8 lanes x 16 B cover the 32 columns of one row, 8 rows per wave instruction
            const int tg = tj0 + lc;
            // (a rolled loop where nothing indexes registers by `it`: eight copies of the store path - vector form, ragged mask,
            // frame-by-frame tail - were 6 KB of instructions per kernel that every workgroup streamed through once)
            constexpr int RB_UNROLL = (PRE || FZ) ? OROWS / 8 : 1;
            float ccs[GC_STATS ? 4 : 1] = {}, ccq[GC_STATS ? 4 : 1] = {};      // GCParams::cstats: this lane's 4 frames over its rows
            // (round 6) the row's offsets run along with the loop: as `(long)m * p.d_c` inside the rolled loop every iteration paid
            // two 64-bit multiply-adds and a dozen v_readlane for their spilled scalars
            long doff = (long)(mo0 + lr) * p.d_c + tg;                 // dst / dst_elu / cmb_s offset of (row m, frame tg)
            const long dstep = 8L * p.d_c;
            long soff = 0;
            if constexpr (GC_STATS) soff = (long)(b + dbJ) * p.st_b + (long)(mo0 + lr) * p.st_c + (long)fow * p.st_f + (tj0 >> 5) * 2;
            const long sstep = GC_STATS ? 8L * p.st_c : 0;
#pragma unroll RB_UNROLL
            for (int it = 0; it < OROWS / 8; ++it, doff += dstep, soff += sstep) {
                const int row = it * 8 + lr, m = mo0 + row;
                floatx4 v = *reinterpret_cast<const floatx4*>(strip + row * OST + lc);
                if constexpr (GC_STATS) {
                    if (p.cstats && m < MoJ) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float xk = (tg + k < tstat) ? v[k] : 0.f;
                            ccs[k] += xk;
                            ccq[k] = fmaf(xk, xk, ccq[k]);
                        }
                    }
                }
                if constexpr (GC_STATS) {
                    // statistics of the stored values for the InstanceNorm behind this layer: the 8 lanes that hold one
                    // row of the 32-column sub-tile add up their 4 frames each (frames >= Tout masked), fixed order
                    if (p.stats) {
                        float s = 0.f, q = 0.f;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float xk = (tg + k < tstat) ? v[k] : 0.f;
                            s += xk;
                            q = fmaf(xk, xk, q);
                        }
                        s = dpp_add8(s);
                        q = dpp_add8(q);
                        const int tb = tg - lc;                  // first frame of the sub-tile (= tj0)
                        if ((lane & 7) == 0 && m < MoJ && tb < p.Tout) {
                            float* sp = p.stats + soff;
                            sp[0] = s;
                            sp[1] = q;
                        }
                    }
                }
                if (m < MoJ) {
                    float* __restrict__ dp = dstJ + doff;
                    if (__builtin_expect(tg + 3 < p.Tout, 1)) {
                        if (EPI == EPI_ADD || EPI == EPI_MUL) v = (EPI == EPI_ADD) ? v + rvp[it] : v * rvp[it];
                        if (EPI == EPI_CMB) {
                            const int rt = wm * (TM * 32) + row;
                            const float sh_ = ep[rt], sl_ = ep[BM + rt], sc_ = ep[2 * BM + rt];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float u = fmaf(fmaf(cmb_sg, v[k], rvp[it][k]), sc_, sh_);
                                v[k] = u >= 0.f ? u : sl_ * u;
                            }
                        }
                        if (__builtin_expect(tg + 3 >= tvalid, 0)) {      // rows of a ragged batch: frames past the row's own end are stored as zeros
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] = (tg + k < tvalid) ? v[k] : 0.f;
                        }
                        if (FZ) {
                            float* __restrict__ zr = fzb + (long)m * p.fz_c + tg;
                            floatx4 re = zre[it], ii = zim[it];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float cm = fm_sqrt(fmaxf(re[k] * re[k] + ii[k] * ii[k], 1.1920928955078125e-07f));
                                const float sg = fsig_(v[k]);
                                re[k] += sg;
                                ii[k] += sg;
                                v[k] += fsig_(cm);
                            }
                            *reinterpret_cast<floatx4*>(zr) = re;
                            *reinterpret_cast<floatx4*>(zr + p.fz_im) = ii;
                            if (p.fz_s) *reinterpret_cast<floatx4*>(zr + p.fz_s) = re + ii;      // S = R + I of a three-plane tensor
                        }
                        *reinterpret_cast<floatx4*>(dp) = v;
                        if (EPI == EPI_GLU && deluJ) {
                            floatx4 e4;
#pragma unroll
                            for (int k = 0; k < 4; ++k) e4[k] = v[k] > 0.f ? v[k] : fm_expm1(v[k]);
                            *reinterpret_cast<floatx4*>(deluJ + doff) = e4;
                        }
                        if (EPI == EPI_CMB && cmbsJ) {
                            floatx4 s4 = v + ivp[it];
                            if (__builtin_expect(tg + 3 >= tvalid, 0)) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) s4[k] = (tg + k < tvalid) ? s4[k] : 0.f;
                            }
                            *reinterpret_cast<floatx4*>(cmbsJ + doff) = s4;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (tg + k < p.Tout) {
                                float o = v[k];
                                if (EPI == EPI_ADD) o += resJ[(long)m * p.x_c + tg + k];
                                if (EPI == EPI_MUL) o *= resJ[(long)m * p.x_c + tg + k];
                                if (EPI == EPI_CMB) {
                                    const int rt = wm * (TM * 32) + row;
                                    const float u = fmaf(fmaf(cmb_sg, o, resJ[(long)m * p.x_c + tg + k]), ep[2 * BM + rt], ep[rt]);
                                    o = u >= 0.f ? u : ep[BM + rt] * u;
                                }
                                o = (tg + k < tvalid) ? o : 0.f;
                                if (FZ) o = gc_fuse1(o, fzb + (long)m * p.fz_c + tg + k, p.fz_im, p.fz_s);
                                dp[k] = o;
                                if (EPI == EPI_GLU && deluJ) deluJ[doff + k] = o > 0.f ? o : fm_expm1(o);
                                if (EPI == EPI_CMB && cmbsJ) {
                                    const float iv = cmbiJ[doff + k];
                                    cmbsJ[doff + k] = (tg + k < tvalid) ? o + iv : 0.f;
                                }
                            }
                    }
                }
            }
            if constexpr (GC_STATS) {
                if (p.cstats) {
                    // the 8 lanes (lane >> 3 = row within 8) that hold the same 4 frames: lanes l and l ^ 8 meet on the DPP network
                    // (row_ror:8), the four pairs (lane >> 4) go to LDS and are added with the wave rows' partials below
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        ccs[k] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ccs[k]), 0x128, 0xF, 0xF, true));
                        ccq[k] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(ccq[k]), 0x128, 0xF, 0xF, true));
                    }
                    if (!(lane & 8)) {
                        float* cp_ = cpart + (((wm * 4 + (lane >> 4)) * BN) + wt * (TN * 32) + j * 32 + lc) * 2;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            cp_[2 * k] = ccs[k];
                            cp_[2 * k + 1] = ccq[k];
                        }
                    }
                }
            }
        };
        epi_tile(std::integral_constant<int, 0>{});
        if constexpr (TN > 1) epi_tile(std::integral_constant<int, 1>{});
        if constexpr (TN > 2) {
            epi_tile(std::integral_constant<int, 2>{});
            epi_tile(std::integral_constant<int, 3>{});
        }
        static_assert(TN == 1 || TN == 2 || TN == 4, "epilogue: column tiles per wave");
        if constexpr (GC_STATS) {
            if (p.cstats) {      // (block-uniform) the wave rows' partial column sums -> one (sum, sum of squares) per frame of this row
                __syncthreads();
                for (int col = tid; col < BN; col += 256) {
                    float a = 0.f, c2 = 0.f;
#pragma unroll
                    for (int w = 0; w < 4 * WM; ++w) {
                        a += cpart[(w * BN + col) * 2];
                        c2 += cpart[(w * BN + col) * 2 + 1];
                    }
                    const int ukc = FLAT ? col >> 5 : 0;      // FLAT: the column's unit - its batch row and first frame
                    const int t = FLAT ? utab[4 * ukc + 1] + (col & 31) : t0 + col;
                    const long bc = FLAT ? b + utab[4 * ukc] : b;
                    if (t < p.Tout && fo < p.so * p.Q + p.po && (!FLAT || utab[4 * ukc + 2])) {
                        float* cg = p.cstats + bc * p.cs_b + (long)fo * p.cs_f + 2 * t;
                        cg[0] = a;
                        cg[1] = c2;
                    }
                }
            }
        }
    } else {   // EPI_LSTM
        // a lane's 16 accumulators of one MFMA tile are the i,f,g,o gates of 4 cells: all 16 gate pre-activations and
        // the 4 cell states are fetched by unconditional (clamped) loads in one batch, then the cells update
        // (p.aux == nullptr: the input projection is part of this GEMM - a second source in the K loop - and the gate
        // pre-activations only need the bias, FullSubNet's first sub-band layer)
        const bool has_gx = p.aux != nullptr;
        const float* __restrict__ gx = has_gx ? p.aux + (long)z * p.aux_z + (long)b * p.x_b + (long)fo * p.x_f : p.bias;
        float* __restrict__ cell = p.cell + (long)z * p.cell_z + (long)b * p.d_b + (long)fo * p.d_f;
        // the cell states of ALL the wave's tiles are requested first (one HBM round trip instead of one per tile)
        float cpa[TM][TN][4];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int tc = min(t0 + wn * (TN * 32) + j * 32 + l31, p.Tout - 1);
                const int mb = m0 + wm * (TM * 32) + i * 32 + 4 * hi;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int m = min(mb + 8 * g4, p.M - 4);
                    cpa[i][j][g4] = p.first_step ? 0.f : cell[(long)(m >> 2) * p.d_c + tc];
                }
            }
        // (cell type and gate source are chosen once per workgroup, not inside each of the 32 - 64 unrolled cells)
        auto cells = [&](auto GRU_, auto HGX_) __attribute__((always_inline)) {
            constexpr bool GRU = decltype(GRU_)::value, HGX = decltype(HGX_)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int t = t0 + wn * (TN * 32) + j * 32 + l31;
                const int tc = min(t, p.Tout - 1);
                const int mb = m0 + wm * (TM * 32) + i * 32 + 4 * hi;      // gate-i row of this lane's first cell
                float g[16], cp[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int m = min(mb + 8 * g4, p.M - 4);
                    const float* gp = HGX ? gx + (long)m * p.x_c + tc : gx + m;
                    const long gs = HGX ? p.x_c : 1;
#pragma unroll
                    for (int k = 0; k < 4; ++k) g[4 * g4 + k] = gp[(long)k * gs];
                    cp[g4] = cpa[i][j][g4];
                }
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int m = mb + 8 * g4;
                    const float gi = acc[i][j][4 * g4 + 0] + g[4 * g4 + 0];
                    const float gf = acc[i][j][4 * g4 + 1] + g[4 * g4 + 1];
                    const float gg = acc[i][j][4 * g4 + 2] + g[4 * g4 + 2];
                    const float go = acc[i][j][4 * g4 + 3] + g[4 * g4 + 3];
                    // fast exp / reciprocal as in the persistent LSTM kernels (5 transcendentals per cell, 16 cells per lane
                    // and tile: libm's tanhf / expf made this the longest phase of a step block)
                    float cn, hn;
                    if constexpr (GRU) {
                        // torch.nn.GRU: r = s(W_ir x + b_ir + W_hr h + b_hr), z likewise, n = tanh(W_in x + b_in + r (W_hn h + b_hn)),
                        // h' = (1 - z) n + z h.  Rows (r, z, n, -): gi / gf / gg hold the r / z / n sums, the aux row of the unused
                        // fourth gate carries b_hn (go = 0 + b_hn), `cell` keeps h for the next step
                        const float r = fsig_(gi), zz = fsig_(gf);
                        const float hw = acc[i][j][4 * g4 + 2] + go;            // W_hn h + b_hn
                        const float nn = ftanh_(g[4 * g4 + 2] + r * hw);
                        hn = (1.f - zz) * nn + zz * cp[g4];
                        cn = hn;
                    } else {
                        cn = fsig_(gf) * cp[g4] + fsig_(gi) * ftanh_(gg);
                        hn = fsig_(go) * ftanh_(cn);
                    }
                    if (m + 3 < p.M && t < p.Tout) {
                        const long oi = (long)(m >> 2) * p.d_c + t;
                        cell[oi] = cn;
                        dst[oi] = hn;
                    }
                }
            }
    
        };
        if (p.gru) {
            if (has_gx) cells(std::true_type{}, std::true_type{});
            else cells(std::true_type{}, std::false_type{});
        } else {
            if (has_gx) cells(std::false_type{}, std::true_type{});
            else cells(std::false_type{}, std::false_type{});
        }
    }
#ifdef GC_TIMING
    GC_T(5);
    if (tid == 0) {
        for (int i = 0; i < 6; ++i) atomicAdd(p.timing + i, tacc[i]);
        for (int i = 7; i < 12; ++i) atomicAdd(p.timing + i, tacc[i]);
        atomicAdd(p.timing + 6, 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Direct (VALU) path for layers with at most 4 output channels - the final (de)convs back to one spectrum plane or
// an RI pair (DCCRN decoder layer 5: 64 -> 2, DCCRN_cprs.py:108-132; CRN/DPCRN/GCRN/CTSNet last layers alike).  On
// the MFMA path their M pads to 32 rows (16x wasted matrix work and a full LDS staging per 2 useful rows); here a
// thread owns one (q, t) output position, walks the same tap table with coalesced loads straight from L1/L2 (every
// input element is re-read by ~ntaps neighbouring positions: cache hits), and the weights are wave-uniform scalar
// loads.  The layer is then bound by reading its input once from HBM.
// ------------------------------------------------------------------------------------------------
template <int MM, int EPI>
__global__ __launch_bounds__(256) void gc_small_kernel(const GCParams p) {
    __shared__ int s_df[GC_MAX_TAPS], s_dt[GC_MAX_TAPS];
    if (threadIdx.x < p.ntaps) {
        s_df[threadIdx.x] = p.tab[p.tab[GC_MAX_ROWS + threadIdx.x]];
        s_dt[threadIdx.x] = p.tab[GC_MAX_ROWS + GC_MAX_TAPS + threadIdx.x];
    }
    __syncthreads();
    // XCD-aware block order (block id i runs on XCD i % 8): neighbouring frequency rows q of one (b, t-tile) read the
    // same input rows through their taps, so they are made neighbours inside one XCD's L2
    int lid;
    {
        const int nblk = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, slot = id >> 3, q8 = nblk >> 3, r8 = nblk & 7;
        lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    }
    const int q = lid % p.Q;
    int rest = lid / p.Q;
    const int nt = (p.Tout + 255) >> 8;
    const int tt = rest % nt;
    rest /= nt;
    const int b = rest % p.B, z = rest / p.B;
    const int t = tt * 256 + threadIdx.x;
    const int tc = min(t, p.Tout - 1);
    float acc[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[m] = 0.f;
    const float* __restrict__ W = p.Ws + (long)z * (p.C0 + p.C1) * p.ntaps * MM;
#pragma unroll 1
    for (int seg = 0; seg < 2; ++seg) {
        const int C = seg ? p.C1 : p.C0;
        if (C == 0) continue;
        const float* __restrict__ src = seg ? p.src1 + (long)z * p.src1_z + (long)b * p.s1_b
                                            : p.src0 + (long)z * p.src0_z + (long)b * p.s0_b;
        const long s_c = seg ? p.s1_c : p.s0_c, s_f = seg ? p.s1_f : p.s0_f;
        const float* __restrict__ Wc = W + (long)(seg ? p.C0 : 0) * p.ntaps * MM;
        for (int j = 0; j < p.ntaps; ++j) {
            const int fi = q * p.si + s_df[j], ti = tc + s_dt[j];
            const bool ok = fi >= 0 && fi < p.Fin && ti >= 0 && ti < p.Tin;
            const float* __restrict__ xp = src + (long)min(max(fi, 0), p.Fin - 1) * s_f + min(max(ti, 0), p.Tin - 1);
            const float keep = ok ? 1.f : 0.f;
#pragma unroll 8
            for (int c = 0; c < C; ++c) {
                const float x = xp[(long)c * s_c] * keep;
                const float* __restrict__ w = Wc + ((long)c * p.ntaps + j) * MM;
#pragma unroll
                for (int m = 0; m < MM; ++m) acc[m] = fmaf(w[m], x, acc[m]);
            }
        }
    }
    if (t >= p.Tout) return;
    const int fo = q * p.so + p.po;
    const float* __restrict__ bias = (fo < p.pad_lo) ? p.bias_pad : (p.bias ? p.bias + (long)z * p.bias_z : nullptr);
    float* __restrict__ dst = p.dst + (long)z * p.dst_z + (long)b * p.d_b + (long)fo * p.d_f;
    const float* __restrict__ res = (EPI == EPI_ADD || EPI == EPI_MUL) ? p.aux + (long)z * p.aux_z + (long)b * p.x_b + (long)fo * p.x_f : nullptr;
    if constexpr (EPI == EPI_GLU) {
        // rows are (value, gate) pairs: out[j] = act(((a + bias) * sigmoid(g + bias)) * post_scale + post_shift), same fast
        // sigmoid as the MFMA tile's GLU epilogue
#pragma unroll
        for (int j = 0; j < MM / 2; ++j) {
            if (2 * j + 1 < p.M) {
                const float a = acc[2 * j] + (bias ? bias[2 * j] : 0.f), g = acc[2 * j + 1] + (bias ? bias[2 * j + 1] : 0.f);
                float v = a * fsig_(g);
                if (p.post_scale) v = v * p.post_scale[j] + p.post_shift[j];
                v = act_apply(v, p.act, p.slope ? p.slope[j] : 0.f);
                if (p.pair) {       // parity classes as virtual row groups: (value, gate) pairs [cls * pair / 2, ...) belong to class cls
                    const int cls = (2 * j) / p.pair, fo2 = q * p.so + (cls ? p.po2 : p.po);
                    if (fo2 < p.fo_lim) dst[(long)(fo2 - fo) * p.d_f + (long)(j - cls * (p.pair >> 1)) * p.d_c + t] = v;
                } else {
                    dst[(long)j * p.d_c + t] = v;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < MM; ++m) {
        if (m < p.M) {
            float v = acc[m] + (bias ? bias[m] : 0.f);
            v = act_apply(v, p.act, p.slope ? p.slope[m] : 0.f);
            if (EPI == EPI_ADD) v += res[(long)m * p.x_c + t];
            if (EPI == EPI_MUL) v *= res[(long)m * p.x_c + t];
            if (EPI == EPI_ACT && p.pair) {
                // both parity classes of a transposed conv in one launch: rows [cls * pair, (cls + 1) * pair) are class cls
                const int cls = m / p.pair, fo2 = q * p.so + (cls ? p.po2 : p.po);
                if (fo2 < p.fo_lim) dst[(long)(fo2 - fo) * p.d_f + (long)(m - cls * p.pair) * p.d_c + t] = v;
                continue;
            }
            dst[(long)m * p.d_c + t] = v;
        }
    }
}

// LDS-tiled form of the direct path: a workgroup owns SQB neighbouring output rows x 256 frames, stages the input rows all
// of their taps touch (SQB * si + tap span rows, CC channels at a time) in LDS once and reads every tap from there.  With
// one row per workgroup (gc_small_kernel) the three frequency taps of a row came from three workgroups and the
// L2 <-> fabric counters showed the input fetched 3x; two rows per thread straight from L1 made it 5x.
constexpr int SQB = 8;
constexpr int SWT = 264;        // LDS row stride of the patch: 256 frames + tap span (<= 8), compile time so that the
                                // reads of the 8 rows are one base register + immediate offsets
template <int MM, int EPI, int SI>
__global__ __launch_bounds__(256) void gc_small_lds_kernel(const GCParams p, const int CC, const int NR, const int dfmin,
                                                           const int dtlo) {
    extern __shared__ float patch[];                 // [CC][NR][SWT]
    __shared__ int s_off[GC_MAX_TAPS];
    __shared__ float s_w[8 * GC_MAX_TAPS * MM];      // weights of the staged channels (a global load per tap would be
                                                     // waited for inside the tap loop)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < p.ntaps)                               // patch offset of tap j for the block's first output row
        s_off[tid] = (p.tab[p.tab[GC_MAX_ROWS + tid]] - dfmin) * SWT + (p.tab[GC_MAX_ROWS + GC_MAX_TAPS + tid] - dtlo);
    const int QG = (p.Q + SQB - 1) / SQB;
    int rest = blockIdx.x;
    const int q0 = (rest % QG) * SQB;
    rest /= QG;
    const int nt = (p.Tout + 255) >> 8;
    const int tt = rest % nt;
    rest /= nt;
    const int b = rest % p.B, z = rest / p.B;
    const int t = tt * 256 + tid;
    const int f0 = q0 * SI + dfmin, tb = tt * 256 + dtlo;            // input coordinates of patch element (0, 0)
    float acc[SQB][MM];
#pragma unroll
    for (int qq = 0; qq < SQB; ++qq)
#pragma unroll
        for (int m = 0; m < MM; ++m) acc[qq][m] = 0.f;
    const float* __restrict__ W = p.Ws + (long)z * (p.C0 + p.C1) * p.ntaps * MM;
    const int plane = NR * SWT;
#pragma unroll 1
    for (int seg = 0; seg < 2; ++seg) {
        const int C = seg ? p.C1 : p.C0;
        if (C == 0) continue;
        const float* __restrict__ src = seg ? p.src1 + (long)z * p.src1_z + (long)b * p.s1_b
                                            : p.src0 + (long)z * p.src0_z + (long)b * p.s0_b;
        const long s_c = seg ? p.s1_c : p.s0_c, s_f = seg ? p.s1_f : p.s0_f;
        const float* __restrict__ Wc = W + (long)(seg ? p.C0 : 0) * p.ntaps * MM;
        // The next chunk's patch rows and weights travel in registers while the current chunk's taps run: a wave owns <= SRB rows
        // of a chunk (CC * NR <= 23 rows at the 24 KB patch budget), SWN loads each - issued right after the previous chunk's values
        // went to LDS, written behind the barrier that ends the current chunk's reads.  (Round 5, ablations SE_GC_DBG 64 / 128: the
        // staging was 85 % of this kernel at 2.2 TB/s - one row per wave in flight - against a third for the taps.)
        constexpr int SRB = 6, SWN = (SWT + 63) / 64, SWW = (8 * GC_MAX_TAPS * MM + 255) / 256;
        float sv[SRB][SWN], wv[SWW];
        auto issue = [&](int c0n) {
            const int ccn = min(CC, C - c0n);
#pragma unroll
            for (int i = 0; i < SRB; ++i) {
                const int row = min(wave + 4 * i, ccn * NR - 1);
                const int c = row / NR, r = row - c * NR;
                const float* __restrict__ sp = src + (long)(c0n + c) * s_c + (long)min(max(f0 + r, 0), p.Fin - 1) * s_f;
#pragma unroll
                for (int k = 0; k < SWN; ++k) sv[i][k] = sp[min(max(tb + lane + 64 * k, 0), p.Tin - 1)];
            }
#pragma unroll
            for (int k = 0; k < SWW; ++k) wv[k] = Wc[(long)c0n * p.ntaps * MM + min(tid + 256 * k, ccn * p.ntaps * MM - 1)];
        };
        issue(0);
#pragma unroll 1
        for (int c0 = 0; c0 < C; c0 += CC) {
            const int cc = min(CC, C - c0);
            __syncthreads();                         // the previous chunk has been consumed (and s_off is visible)
#pragma unroll
            for (int k = 0; k < SWW; ++k)
                if (tid + 256 * k < cc * p.ntaps * MM) s_w[tid + 256 * k] = wv[k];
            // stage [cc][NR][SWT]: a wave takes whole rows, 64 consecutive frames per load; outside the plane: zeros
#pragma unroll
            for (int i = 0; i < SRB; ++i) {
                const int row = wave + 4 * i;
                if (row < cc * NR) {
                    const int c = row / NR, r = row - c * NR;
                    const int fi = f0 + r;
                    const bool rok = fi >= 0 && fi < p.Fin;
                    float* __restrict__ dp = patch + c * plane + r * SWT;
#pragma unroll
                    for (int k = 0; k < SWN; ++k) {
                        const int w = lane + 64 * k, ti = tb + w;
                        if (w < SWT) dp[w] = (rok && ti >= 0 && ti < p.Tin) ? sv[i][k] : 0.f;
                    }
                }
            }
            if (c0 + CC < C) issue(c0 + CC);
            __syncthreads();
            for (int c = 0; c < cc; ++c) {
                const float* __restrict__ pc = patch + c * plane + tid;
                const float* wc = s_w + c * p.ntaps * MM;
#pragma unroll 2
                for (int j = 0; j < p.ntaps; ++j) {
                    const float* __restrict__ pj = pc + s_off[j];
                    float w[MM];
#pragma unroll
                    for (int m = 0; m < MM; ++m) w[m] = wc[j * MM + m];
#pragma unroll
                    for (int qq = 0; qq < SQB; ++qq) {
                        const float x = pj[qq * SI * SWT];
#pragma unroll
                        for (int m = 0; m < MM; ++m) acc[qq][m] = fmaf(w[m], x, acc[qq][m]);
                    }
                }
            }
        }
    }
    if (t >= p.Tout) return;
#pragma unroll
    for (int qq = 0; qq < SQB; ++qq) {
        if (q0 + qq >= p.Q) break;
        const int fo = (q0 + qq) * p.so + p.po;
        const float* __restrict__ bias = (fo < p.pad_lo) ? p.bias_pad : (p.bias ? p.bias + (long)z * p.bias_z : nullptr);
        float* __restrict__ dst = p.dst + (long)z * p.dst_z + (long)b * p.d_b + (long)fo * p.d_f;
        const float* __restrict__ res = (EPI == EPI_ADD || EPI == EPI_MUL) ? p.aux + (long)z * p.aux_z + (long)b * p.x_b + (long)fo * p.x_f : nullptr;
        if constexpr (EPI == EPI_GLU) {
#pragma unroll
            for (int j = 0; j < MM / 2; ++j) {
                if (2 * j + 1 < p.M) {
                    const float a = acc[qq][2 * j] + (bias ? bias[2 * j] : 0.f), g = acc[qq][2 * j + 1] + (bias ? bias[2 * j + 1] : 0.f);
                    float v = a * fsig_(g);
                    if (p.post_scale) v = v * p.post_scale[j] + p.post_shift[j];
                    v = act_apply(v, p.act, p.slope ? p.slope[j] : 0.f);
                    if (p.pair) {       // see gc_small_kernel
                        const int cls = (2 * j) / p.pair, fo2 = (q0 + qq) * p.so + (cls ? p.po2 : p.po);
                        if (fo2 < p.fo_lim) dst[(long)(fo2 - fo) * p.d_f + (long)(j - cls * (p.pair >> 1)) * p.d_c + t] = v;
                    } else {
                        dst[(long)j * p.d_c + t] = v;
                    }
                }
            }
            continue;
        }
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            if (m < p.M) {
                float v = acc[qq][m] + (bias ? bias[m] : 0.f);
                v = act_apply(v, p.act, p.slope ? p.slope[m] : 0.f);
                if (EPI == EPI_ADD) v += res[(long)m * p.x_c + t];
                if (EPI == EPI_MUL) v *= res[(long)m * p.x_c + t];
                if (EPI == EPI_ACT && p.pair) {      // see gc_small_kernel
                    const int cls = m / p.pair, fo2 = (q0 + qq) * p.so + (cls ? p.po2 : p.po);
                    if (fo2 < p.fo_lim) dst[(long)(fo2 - fo) * p.d_f + (long)(m - cls * p.pair) * p.d_c + t] = v;
                    continue;
                }
                dst[(long)m * p.d_c + t] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Thin path: launches that produce at most 8 frames per row (the chunks of the frame-online mode).  A 64-column MFMA tile
// would spend its whole matrix time on padding (a 64 x 320 TCM conv: 5 us of MFMAs for one useful column); here the layer is
// a handful of dot products.  A workgroup owns 16 output rows of one (b, q): thread (mi, kg) walks every 16th K row
// (ci, tap) of the packed weight matrix - 16 lanes read 64 B of one K row, the activation is the same address for all of
// them - with up to 8 frame accumulators, the 16 partial sums per output are folded through LDS, and the epilogues of the
// MFMA kernel follow.
// ------------------------------------------------------------------------------------------------
// NT = frames per row (1, 2, 4, 8: no masked loads for the frames a short chunk does not have).  A thread's whole K walk is
// issued in batches of 8 independent (weight, activation) loads - a block lives for 2-3 memory round trips - and the tap
// table travels in the kernel arguments (GCParams::tdf / tdt), so that no address waits for a table fetch.
template <int EPI, int NT>
__device__ __forceinline__ void gc_thin_body(const GCParams& p, const int blk) {
    constexpr int RB = 8, KG = 32, UN = 8;          // rows per workgroup, K groups, K rows in flight per thread
    __shared__ float part[KG][RB][NT + 1];
    __shared__ int s_df[GC_MAX_TAPS], s_dt[GC_MAX_TAPS];
    const int tid = threadIdx.x, mi = tid & (RB - 1), kg = tid / RB;
#pragma unroll
    for (int j = 0; j < GC_MAX_TAPS; ++j)       // constant indices: scalar loads from the argument segment
        if (tid == j) {
            s_df[j] = p.tdf[j];
            s_dt[j] = p.tdt[j];
        }
    __syncthreads();
    const int nm = (p.M + RB - 1) / RB;          // (rows M .. Mp of the packed matrix are padding)
    const int mt = blk % nm;
    const int rest = blk / nm;
    const int q = rest % p.Q, b = rest / p.Q;
    const int m = mt * RB + mi, t0 = p.t_base, n = p.Tout - p.t_base;
    const int ntaps = p.ntaps, Ktot = (p.C0 + p.C1) * ntaps;
    const int nch0 = p.C0 > 0 ? (p.C0 + p.CI_C - 1) / p.CI_C : 0;
    float acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = 0.f;
    for (int k0 = kg; k0 < Ktot; k0 += KG * UN) {
        float w[UN], xv[UN][NT];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kk = k0 + u * KG;
            const bool kok = kk < Ktot;
            const int kc = kok ? kk : 0;
            const int ci = kc / ntaps, j = kc - ci * ntaps;
            const bool seg = ci >= p.C0;
            const int cs = seg ? ci - p.C0 : ci;
            const int chunk = (seg ? nch0 : 0) + cs / p.CI_C, cil = cs % p.CI_C;
            const float wv = p.A[((long)chunk * p.KCp + cil * ntaps + j) * p.Mp + m];
            const int f = q * p.si + s_df[j];
            const bool fok = kok && f >= 0 && f < p.Fin;
            const int fc = fok ? f : 0;
            const float* __restrict__ src = seg ? p.src1 + (long)b * p.s1_b + (long)cs * p.s1_c + (long)fc * p.s1_f
                                                : p.src0 + (long)b * p.s0_b + (long)cs * p.s0_c + (long)fc * p.s0_f;
            const int tb = t0 + s_dt[j];
            w[u] = fok ? wv : 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int ti = tb + t;
                const bool ok = t < n && ti >= 0 && ti < p.Tin;
                const float x = src[ok ? ti : 0];
                xv[u][t] = ok ? x : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = fmaf(w[u], xv[u][t], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) part[kg][mi][t] = acc[t];
    __syncthreads();
    const bool fold = tid < RB * NT;          // (NT <= 16: at most 128 threads)
    const int r = tid & (RB - 1), t = (tid / RB) & (NT - 1);
    float tot = 0.f;
    if (fold) {
#pragma unroll
        for (int g = 0; g < KG; ++g) tot += part[g][r][t];
    }
    __syncthreads();
    if (fold) part[0][r][t] = tot;      // row pairs of the GLU epilogue read their neighbour's total
    __syncthreads();
    if (!fold || t >= n) return;
    const int fo = q * p.so + p.po, mr = mt * RB + r;
    const float* __restrict__ bias = (fo < p.pad_lo) ? p.bias_pad : p.bias;
    float* __restrict__ dst = p.dst + (long)b * p.d_b + (long)fo * p.d_f + t0 + t;
    if (EPI == EPI_LSTM) {
        // rows 4u + {i, f, g, o} (GCParams::gru: {r, z, n, -}): the thread of a unit's first row updates the cell, as the
        // MFMA kernel's epilogue does (a frame-online step of a 1024-wide LSTM is a 16.7 MB matrix-vector product: 64 us as
        // a latency-bound K loop of one MFMA workgroup per 128 rows, ~12 us here)
        if ((r & 3) || mr + 3 >= p.M) return;
        const int tc = t0 + t;
        const bool has_gx = p.aux != nullptr;
        const float* __restrict__ gx = has_gx ? p.aux + (long)b * p.x_b + (long)fo * p.x_f + (long)mr * p.x_c + tc : p.bias + mr;
        const long gs = has_gx ? p.x_c : 1;
        float* __restrict__ cell = p.cell + (long)b * p.d_b + (long)fo * p.d_f + (long)(mr >> 2) * p.d_c + tc;
        const float g0 = gx[0], g1 = gx[gs], g2 = gx[2 * gs], g3 = gx[3 * gs];
        const float cp = p.first_step ? 0.f : *cell;
        const float gi = part[0][r][t] + g0, gf = part[0][r + 1][t] + g1, gg = part[0][r + 2][t] + g2, go = part[0][r + 3][t] + g3;
        float cn, hn;
        if (p.gru) {
            const float rr = fsig_(gi), zz = fsig_(gf);
            const float hw = part[0][r + 2][t] + go;            // W_hn h + b_hn
            const float nn = ftanh_(g2 + rr * hw);
            hn = (1.f - zz) * nn + zz * cp;
            cn = hn;
        } else {
            cn = fsig_(gf) * cp + fsig_(gi) * ftanh_(gg);
            hn = fsig_(go) * ftanh_(cn);
        }
        *cell = cn;
        dst[(long)(mr >> 2) * p.d_c] = hn;
    } else if (EPI == EPI_GLU) {
        if ((r & 1) || mr + 1 >= p.M) return;
        const int oc = mr >> 1;
        const float a = part[0][r][t] + (bias ? bias[mr] : 0.f), g = part[0][r + 1][t] + (bias ? bias[mr + 1] : 0.f);
        float v = a * fsig_(g);
        if (p.post_scale) v = v * p.post_scale[oc] + p.post_shift[oc];
        dst[(long)oc * p.d_c] = act_apply(v, p.act, p.slope ? p.slope[oc] : 0.f);
    } else {
        if (mr >= p.M) return;
        float v = part[0][r][t] + (bias ? bias[mr] : 0.f);
        v = act_apply(v, p.act, p.slope ? p.slope[mr] : 0.f);
        if (EPI == EPI_ADD || EPI == EPI_MUL) {
            const float rv = p.aux[(long)b * p.x_b + (long)fo * p.x_f + (long)mr * p.x_c + t0 + t];
            v = (EPI == EPI_ADD) ? v + rv : v * rv;
        }
        dst[(long)mr * p.d_c] = v;
    }
}
template <int EPI, int NT>
__global__ __launch_bounds__(256) void gc_thin_kernel(const GCParams p) {
    gc_thin_body<EPI, NT>(p, (int)blockIdx.x);
}
// Both frequency-parity classes of a transposed conv in one launch: a class is a block range with its own taps / weights /
// output rows (a kernel of this size costs ~4.7 us before it does anything: a one-frame push of TaylorSENet_new made 36 such
// second launches)
struct GCThinPair {
    GCParams p[2];
    int nblk0;
};
template <int EPI, int NT>
__global__ __launch_bounds__(256) void gc_thin_pair_kernel(const GCThinPair a) {
    const int cls = (int)blockIdx.x >= a.nblk0 ? 1 : 0;
    gc_thin_body<EPI, NT>(a.p[cls], (int)blockIdx.x - (cls ? a.nblk0 : 0));
}
template <int EPI>
static void gc_thin_launch_n(const GCParams& p, dim3 grid, int n, hipStream_t stream) {
    if (n <= 1) hipLaunchKernelGGL((gc_thin_kernel<EPI, 1>), grid, dim3(256), 0, stream, p);
    else if (n <= 2) hipLaunchKernelGGL((gc_thin_kernel<EPI, 2>), grid, dim3(256), 0, stream, p);
    else if (n <= 4) hipLaunchKernelGGL((gc_thin_kernel<EPI, 4>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gc_thin_kernel<EPI, 8>), grid, dim3(256), 0, stream, p);
}
// number of workgroups of the launch on the thin path, 0: not a thin launch
static long gc_thin_blocks(const GCParams& p) {
    static const int thin_env = getenv("SE_GC_THIN") ? atoi(getenv("SE_GC_THIN")) : 1;
    const int n = p.Tout - p.t_base;
    // (layers with <= 4 output channels have the packed matrix too: a one-frame launch of the direct kernel walks its whole K
    // in one thread per output - 20-40 us; the fused parity pair of a transposed conv only exists there)
    if (!thin_env || (p.Ws && p.pair) || n > GC_THIN_NT || p.Z > 1 || p.stats || p.cstats || p.nrm0 || p.nrm1) return 0;
    if (p.epi != EPI_ACT && p.epi != EPI_ADD && p.epi != EPI_MUL && p.epi != EPI_GLU && p.epi != EPI_LSTM) return 0;
    if (p.epi == EPI_LSTM && (p.M & 3)) return 0;
    const long nblk = (long)((p.M + 7) >> 3) * p.Q * p.B;
    // (a few frames per row is not yet a small launch: the LSTM input projections of a batch-1 decode are 1 "frame" wide
    // and 401 rows high with K = 1024 - matrix work)
    static const long thin_max = getenv("SE_GC_THIN_MAX") ? atol(getenv("SE_GC_THIN_MAX")) : 8192;
    if (nblk > thin_max) return 0;
    // Both paths are latency-bound at these sizes: a thin block walks K / 32 rows with 1 + NT loads each (~0.08 us per row
    // and load, ~2 048 blocks in flight) and re-reads its 8 weight rows from L2 (32 B x K per block: 16 streams x one frame of
    // DCCRN's 256-channel layers = 4 096 blocks x 80 KB - 110 us against 80 us of the MFMA tiles); an MFMA workgroup makes K / 16
    // staged steps of ~0.5 us (~512 in flight).  Measured on the frame-online chunks of CRN / GCRN / DPCRN / DCCRN with 1 and 16
    // streams x 1 and 8 frames: up to 64 columns in all go to the thin kernel unless its weight traffic or its waves say no.
    if (!p.Ws) {
        const int nt = n <= 1 ? 1 : (n <= 2 ? 2 : (n <= 4 ? 4 : 8));
        const double waves_thin = std::ceil((double)nblk / 2048.0);
        const double waves_mfma = std::ceil((double)std::max(p.Z, 1) * p.B * p.Q * std::max(p.n_mtiles, 1) / 512.0);
        if ((long)p.B * n > 64 || nblk > 2800 * waves_mfma || waves_thin * (1 + nt) > 12 * waves_mfma) return 0;
    }
    return nblk;
}
static bool gc_thin_launch(const GCParams& p, hipStream_t stream) {
    const long nblk = (p.fz || p.dst_elu) ? 0 : gc_thin_blocks(p);          // (GCParams::fz / dst_elu: MFMA kernel only)
    if (nblk <= 0) return false;
    const int n = p.Tout - p.t_base;
    dim3 grid((unsigned)nblk);
    switch (p.epi) {
        case EPI_ACT: gc_thin_launch_n<EPI_ACT>(p, grid, n, stream); break;
        case EPI_ADD: gc_thin_launch_n<EPI_ADD>(p, grid, n, stream); break;
        case EPI_MUL: gc_thin_launch_n<EPI_MUL>(p, grid, n, stream); break;
        case EPI_GLU: gc_thin_launch_n<EPI_GLU>(p, grid, n, stream); break;
        case EPI_LSTM: gc_thin_launch_n<EPI_LSTM>(p, grid, n, stream); break;
        default: return false;
    }
    SE_HIP(hipGetLastError());
    return true;
}

template <int EPI>
static void gc_thin_pair_launch_n(const GCThinPair& a, dim3 grid, int n, hipStream_t stream) {
    if (n <= 1) hipLaunchKernelGGL((gc_thin_pair_kernel<EPI, 1>), grid, dim3(256), 0, stream, a);
    else if (n <= 2) hipLaunchKernelGGL((gc_thin_pair_kernel<EPI, 2>), grid, dim3(256), 0, stream, a);
    else if (n <= 4) hipLaunchKernelGGL((gc_thin_pair_kernel<EPI, 4>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((gc_thin_pair_kernel<EPI, 8>), grid, dim3(256), 0, stream, a);
}
bool gc_launch_thin_pair(const GCParams& p0, const GCParams& p1, hipStream_t stream) {
    static const int pair_env = getenv("SE_GC_THIN_PAIR") ? atoi(getenv("SE_GC_THIN_PAIR")) : 1;
    const long n0 = gc_thin_blocks(p0), n1 = gc_thin_blocks(p1);
    if (!pair_env || n0 <= 0 || n1 <= 0 || p0.epi != p1.epi || p0.Tout - p0.t_base != p1.Tout - p1.t_base) return false;
    GCThinPair a;
    a.p[0] = p0;
    a.p[1] = p1;
    a.nblk0 = (int)n0;
    const int n = p0.Tout - p0.t_base;
    dim3 grid((unsigned)(n0 + n1));
    switch (p0.epi) {
        case EPI_ACT: gc_thin_pair_launch_n<EPI_ACT>(a, grid, n, stream); break;
        case EPI_ADD: gc_thin_pair_launch_n<EPI_ADD>(a, grid, n, stream); break;
        case EPI_MUL: gc_thin_pair_launch_n<EPI_MUL>(a, grid, n, stream); break;
        case EPI_GLU: gc_thin_pair_launch_n<EPI_GLU>(a, grid, n, stream); break;
        default: return false;
    }
    SE_HIP(hipGetLastError());
    return true;
}

template <int MM, int EPI>
static void gc_small_lds_launch(const GCParams& p, dim3 grid, size_t shm, int CC, int NR, const GCSmallGeom& sg, hipStream_t stream) {
    if (p.si == 1) hipLaunchKernelGGL((gc_small_lds_kernel<MM, EPI, 1>), grid, dim3(256), shm, stream, p, CC, NR, sg.dfmin, sg.dtmin);
    else hipLaunchKernelGGL((gc_small_lds_kernel<MM, EPI, 2>), grid, dim3(256), shm, stream, p, CC, NR, sg.dfmin, sg.dtmin);
}

template <int MM>
static void gc_small_launch(const GCParams& p, const GCSmallGeom& sg, hipStream_t stream) {
    // LDS-tiled form when the launch still fills the chip with 8-row workgroups and the patch of at least 2 channels fits
    static const int lds_env = getenv("SE_GC_SMALL_LDS") ? atoi(getenv("SE_GC_SMALL_LDS")) : 1;
    const int NR = (SQB - 1) * p.si + (sg.dfmax - sg.dfmin) + 1;
    static const int small_kb = getenv("SE_GC_SMALL_KB") ? atoi(getenv("SE_GC_SMALL_KB")) : 24;       // LDS budget of the staged patch (KB): 2-channel chunks, 6-7 workgroups per CU (40 KB: DCCRN -0.4 %, CTSNet -1.7 %)
    // <= 8: s_w holds 8 channels; <= 24 patch rows per chunk: a wave carries 6 rows of the next chunk in registers (gc_small_lds_kernel)
    const int CC = std::min(std::min(8, 24 / std::max(NR, 1)), (int)((size_t)small_kb * 1024 / ((size_t)NR * SWT * sizeof(float))));
    const long nblk8 = (long)((p.Tout + 255) / 256) * ((p.Q + SQB - 1) / SQB) * p.B * p.Z;
    // (worth it from three frequency rows per output row on: with one or two the plain kernel's caches do as well - CRN /
    // DPCRN last layers measured 1-3 % slower here, DCCRN's 5-tap deconv 1 % faster with a third of the fetches)
    if (lds_env && CC >= 2 && p.Q >= SQB && (p.si == 1 || p.si == 2) && sg.dtmax - sg.dtmin <= SWT - 256 &&
        ((nblk8 >= 4 * 256 && (sg.dfmax - sg.dfmin >= 2 || (sg.dfmax - sg.dfmin >= 1 && MM >= 2 && p.C0 + p.C1 >= 64))) ||
         lds_env == 2)) {        // 2: always (tests)
        SE_CHECK(nblk8 < (1L << 31), "grid size");
        const size_t shm = (size_t)CC * NR * SWT * sizeof(float);
        dim3 grid((unsigned)nblk8);
        switch (p.epi) {
            case EPI_ACT: gc_small_lds_launch<MM, EPI_ACT>(p, grid, shm, CC, NR, sg, stream); break;
            case EPI_ADD: gc_small_lds_launch<MM, EPI_ADD>(p, grid, shm, CC, NR, sg, stream); break;
            case EPI_MUL: gc_small_lds_launch<MM, EPI_MUL>(p, grid, shm, CC, NR, sg, stream); break;
            case EPI_GLU:
                if constexpr (MM >= 2) gc_small_lds_launch<MM, EPI_GLU>(p, grid, shm, CC, NR, sg, stream);
                break;
            default: SE_CHECK(false, "direct small-M path: unsupported epilogue");
        }
        SE_HIP(hipGetLastError());
        return;
    }
    const long nblk = (long)((p.Tout + 255) / 256) * p.Q * p.B * p.Z;
    SE_CHECK(nblk > 0 && nblk < (1L << 31), "grid size");
    dim3 grid((unsigned)nblk);
    switch (p.epi) {
        case EPI_ACT: hipLaunchKernelGGL((gc_small_kernel<MM, EPI_ACT>), grid, dim3(256), 0, stream, p); break;
        case EPI_ADD: hipLaunchKernelGGL((gc_small_kernel<MM, EPI_ADD>), grid, dim3(256), 0, stream, p); break;
        case EPI_MUL: hipLaunchKernelGGL((gc_small_kernel<MM, EPI_MUL>), grid, dim3(256), 0, stream, p); break;
        case EPI_GLU:
            if constexpr (MM >= 2) hipLaunchKernelGGL((gc_small_kernel<MM, EPI_GLU>), grid, dim3(256), 0, stream, p);
            break;
        default: SE_CHECK(false, "direct small-M path: unsupported epilogue");
    }
    SE_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static size_t gc_lds_bytes(const GCParams& p, int BM, size_t epi_bytes, int nbuf = 2) {
    const size_t as = (size_t)((p.KCp * (BM / 4) + 255) / 256) * 1024, bs = (size_t)((p.CI_C * p.nrows * p.Wp + 255) / 256) * 256;
    const size_t nrb = p.flat_upr ? 2 : 1;      // FLAT: parameters of two batch rows
    const size_t nrm = ((p.nrm0 || p.nrm1) ? (size_t)(nrb * GC_NRM_MAXC + 2 * nrb * gc_kcp_max(BM) + 2) * 16 : 0) + (p.flat_upr ? 128 : 0);      // gc_kernel NRM: nrmC + nrmK; FLAT: utab
    const size_t cst = p.cstats ? (size_t)4 * 256 * 2 * 4 : 0;      // GCParams::cstats: cpart [WM][4][BN][2] <= 8 KB, behind the strips inside the staging area
    return std::max((size_t)nbuf * (as + bs) * 4, epi_bytes + cst) + (GC_TAB_KOFF + GC_MAX_KCP + 8) * 4 + (size_t)4 * BM * 4 + 64 + nrm;
}

// Device tables of one patch geometry (row stride Wp): frequency rows / tap table / K-row patch offsets, and the
// per-(thread, slot) descriptors of the patch seen as single frames and as 16 B groups of 4 frames.
GCGeom gc_build_geom(const TapSpec& taps, const std::vector<int>& rows, int dtmin, int nrows, int Wp, int cic, int KC, int NB) {
    GCGeom g;
    std::vector<int> tab(GC_TAB_KOFF + GC_MAX_KCP + 8, 0);
    for (int r = 0; r < nrows; ++r) tab[r] = rows[r];
    for (int j = 0; j < taps.ntaps; ++j) {
        tab[GC_MAX_ROWS + j] = (int)(std::find(rows.begin(), rows.end(), taps.df[j]) - rows.begin());
        tab[GC_MAX_ROWS + GC_MAX_TAPS + j] = taps.dt[j];
    }
    for (int k = 0; k < KC; ++k) {       // k = cil * ntaps + j
        const int cil = k / taps.ntaps, j = k - cil * taps.ntaps;
        tab[GC_TAB_KOFF + k] = cil * (nrows * Wp) + tab[GC_MAX_ROWS + j] * Wp + (taps.dt[j] - dtmin);
    }
    SE_HIP(hipMalloc(&g.tab, tab.size() * sizeof(int)));
    SE_HIP(hipMemcpy(g.tab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
    // slot fe = tid + 256 e -> (row rr = fe / Wp, column w), rr -> (cil, r); groups: slot fg <-> LDS floats 4 fg .. 4 fg + 3
    const int npatch = cic * nrows * Wp, gpr = Wp / 4;
    SE_CHECK(Wp % 4 == 0 && Wp < 4096 && nrows <= 16 && cic < 32768, "patch descriptor field overflow");
    std::vector<unsigned> desc((size_t)NB * 256, 0u), d4((size_t)NB * 256, 0u);
    for (int e = 0; e < NB; ++e)
        for (int t = 0; t < 256; ++t) {
            const int fe = t + 256 * e;
            if (fe < npatch) {
                const int rr = fe / Wp, w = fe - rr * Wp, cil = rr / nrows, r = rr - cil * nrows;
                desc[(size_t)e * 256 + t] = (unsigned)w | ((unsigned)r << 12) | ((unsigned)cil << 16) | 0x80000000u;
            }
            if (4 * fe < npatch) {
                const int rr = fe / gpr, w = 4 * (fe - rr * gpr), cil = rr / nrows, r = rr - cil * nrows;
                d4[(size_t)e * 256 + t] = (unsigned)w | ((unsigned)r << 12) | ((unsigned)cil << 16) | 0x80000000u;
            }
        }
    SE_HIP(hipMalloc(&g.desc, desc.size() * sizeof(unsigned)));
    SE_HIP(hipMemcpy(g.desc, desc.data(), desc.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    SE_HIP(hipMalloc(&g.desc4, d4.size() * sizeof(unsigned)));
    SE_HIP(hipMemcpy(g.desc4, d4.data(), d4.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    return g;
}

GCPlan gc_make_plan(int M, int Cin, const TapSpec& taps, const std::vector<float>& w, const std::vector<float>& bias,
                    const std::vector<float>& slope, int act, int epi, int si, int so, int po, int tout_hint, int Z,
                    int C0split) {
    const int C0 = (C0split < 0 || C0split > Cin) ? Cin : C0split;
    SE_CHECK(taps.ntaps >= 1 && taps.ntaps <= GC_MAX_TAPS, "tap count");
    SE_CHECK((long)w.size() == (long)Z * M * Cin * taps.ntaps, "weight size mismatch in gc_make_plan");
    GCPlan pl;
    GCParams& p = pl.p;
    pl.BM = M >= 96 ? 128 : (M >= 48 ? 64 : 32);
    {   // memory-bound pointwise layers (<= 128 input channels, <= 384 rows: Uformer's conformer, the TCM blocks' 64 -> 256 layers) on
        // 64-row tiles: twice the workgroups in flight, each small enough to spread its DMAs over the matrix loop
        // (SE_GC_PW_BM64=k: input-channel limit, 0 = 128-row tiles; Uformer + 1.1 %, the others unchanged)
        static const int pw64 = getenv("SE_GC_PW_BM64") ? atoi(getenv("SE_GC_PW_BM64")) : 128;
        if (pw64 > 0 && taps.ntaps == 1 && Cin <= pw64 && M >= 96 && M <= 384 && epi != EPI_LSTM) pl.BM = 64;
    }
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
    for (int j = 0; j < taps.ntaps; ++j) {       // the thin kernel reads the taps from its arguments
        SE_CHECK(taps.df[j] >= -32768 && taps.df[j] <= 32767 && taps.dt[j] >= -32768 && taps.dt[j] <= 32767, "tap offset range");
        p.tdf[j] = (short)taps.df[j];
        p.tdt[j] = (short)taps.dt[j];
    }
    // patch geometry: the staged time window starts a multiple of 4 frames before the tile (origin t0 - pad4) so that it
    // decomposes into 16 B groups that never straddle frame 0 (t0 is a multiple of BN): LDS column w <-> frame t0 + dtmin + w
    p.causal = dtmax <= 0;
    pl.lookback = std::max(-dtmin, 0);                      // frames of history the taps reach back (frame-online mode)
    dtmin = -((std::max(-dtmin, 0) + 3) & ~3);
    dtmax = (std::max(dtmax, 0) + 3) & ~3;
    p.dtmin = dtmin;
    p.Wp = pl.BN + (dtmax - dtmin);                         // LDS row stride of the patch (multiple of 4)
    SE_CHECK(p.nrows * p.Wp <= gc_bld_max(pl.BM) * 256, "tap span too wide for one staged patch");
    // chunking: largest CI_C within the staging budgets (SE_GC_KCP: tuning override)
    static const int kcp_cap = getenv("SE_GC_KCP") ? atoi(getenv("SE_GC_KCP")) : GC_MAX_KCP;
    static const bool pw_chunks = !(getenv("SE_GC_PW4") && atoi(getenv("SE_GC_PW4")) == 0);
    // chunk = CI_C input channels x all taps.  Among the sizes that fit the staging budgets take the one that wastes the
    // fewest K rows on padding to a multiple of 4 (rows of zeros cost full MFMAs), then the largest: measured on the DCCRN
    // bench, 20 exact rows per barrier beat 30 rows padded to 32 (taps = 10), 24 beat 30 / 32 (taps = 6)
    static const bool wide_chunk_env = !(getenv("SE_GC_WIDE_CHUNK") && atoi(getenv("SE_GC_WIDE_CHUNK")) == 0);
    int cic = 1;
    double best = -1.0;
    for (int c = 1; c <= std::max(std::max(C0, Cin - C0), 1); ++c) {
        const int kc = c * taps.ntaps, kcp = (kc + 3) & ~3;
        // pointwise layers stage 16 B groups (4x the slots) and size the chunk by the LDS budget of 3 blocks per CU
        const bool pw = taps.ntaps == 1 && pw_chunks;
        const int kmax = pw ? (pl.BM >= 128 ? 24 : 32) : gc_kcp_max(pl.BM);
        const int cap = gc_bld_max(pl.BM) * 256 * (pw ? 4 : 1);
        if (kcp > std::min(kmax, kcp_cap) || c * p.nrows * p.Wp > cap) continue;
        // 64-row layers: keep the chunk small enough for the 64 x 256 tile's patch (3 workgroups per CU) when that still
        // leaves >= 12 K rows per barrier - G2Net's 3-tap convs would stage 8 channels x 3 rows and fall back to 64 x 128
        // tiles (measured: + 4 % for the whole model at batch 256 with 4-channel chunks)
        if (wide_chunk_env && pl.BM == 64 && pl.BN == 128 && taps.ntaps > 1 && epi != EPI_LSTM && kc > 12 &&
            (long)c * p.nrows * (256 + (dtmax - dtmin)) > 4608)
            continue;
        const double score = (double)kc / kcp + 1e-4 * kc;       // padding efficiency first, size second
        if (score > best) {
            best = score;
            cic = c;
        }
    }
    p.CI_C = cic;
    p.KC = cic * taps.ntaps;
    p.KCp = (p.KC + 3) & ~3;
    SE_CHECK(p.KCp <= gc_kcp_max(pl.BM), "single-channel chunk exceeds K budget");
    {
        GCGeom g = gc_build_geom(taps, rows, dtmin, p.nrows, p.Wp, cic, p.KC, gc_bld_max(pl.BM));
        pl.dTab = g.tab;
        pl.dDesc = g.desc;
        pl.dDesc4 = g.desc4;
        p.tab = g.tab;
        p.desc = g.desc;
        p.desc4 = g.desc4;
        // narrower geometries for the last, mostly empty time tile of a row (T = 401 fills 17 of 128 columns of its 4th
        // tile): same weights and chunking, own patch tables; gc_launch sends that tile to a 32- / 64-column kernel
        static const int tail_env = getenv("SE_GC_TAIL") ? atoi(getenv("SE_GC_TAIL")) : 1;
        // (only where the narrow kernel keeps all four waves busy: 128 rows x 32 columns; a 64-row layer would leave half
        // its waves on padding again - measured slower than the skip logic of the full-width tile)
        // (pointwise layers get the geometry for tiny launches only: as a separate tail launch it doubled the launch count
        // of FullSubNet's per-step LSTM GEMMs)
        pl.tail_split = !(taps.ntaps == 1 && pw_chunks);
        if (tail_env && pl.BN == 128 && pl.BM == 128) {
            for (int i = 0; i < 1; ++i) {
                const int bn = 32;
                pl.tail[i].BN = bn;
                pl.tail[i].Wp = bn + (dtmax - dtmin);
                pl.tail[i].g = gc_build_geom(taps, rows, dtmin, p.nrows, pl.tail[i].Wp, cic, p.KC, gc_bld_max(pl.BM));
            }
        }
        // 64-column geometry of the whole layer (tail[1]) for launches that would not fill the chip with 128-column
        // tiles: the batch is only known at launch time, gc_launch picks
        // two-row geometry (gc_kernel: qt2): rows of the patch = union of the input rows of two neighbouring output rows,
        // when that union is an interval and smaller than twice the single-row set
        if (pl.BN == 128 && epi != EPI_LSTM && taps.ntaps > 1) {
            std::vector<int> r2 = rows;
            for (int r : rows)
                if (std::find(r2.begin(), r2.end(), r + si) == r2.end()) r2.push_back(r + si);
            std::sort(r2.begin(), r2.end());
            const bool interval = r2.back() - r2.front() + 1 == (int)r2.size();
            const int wp2 = 64 + (dtmax - dtmin);
            if (interval && (int)r2.size() <= GC_MAX_ROWS && (int)r2.size() < 2 * p.nrows &&
                cic * (int)r2.size() * wp2 <= gc_bld_max(pl.BM) * 256) {
                pl.qt2.BN = 64;
                pl.qt2.Wp = wp2;
                pl.qt2.g = gc_build_geom(taps, r2, dtmin, (int)r2.size(), wp2, cic, p.KC, gc_bld_max(pl.BM));
                pl.qt2_nrows = (int)r2.size();
                pl.qt2_qoff = si * wp2;
            }
        }
        // 256-column geometry of a 64-row layer (tail[2]): 1 x 4 waves of 64 x 64 do the 128 x 128 tile's matrix work per
        // staged K row (the 64 x 128 tile does half of it), for big launches whose rows fill the wide tiles (T = 501: 98 %)
        static const int wide_env = getenv("SE_GC_WIDE") ? atoi(getenv("SE_GC_WIDE")) : 1;
        if (wide_env && pl.BN == 128 && pl.BM == 64 && epi != EPI_LSTM && taps.ntaps > 1) {
            pl.tail[2].BN = 256;
            pl.tail[2].Wp = 256 + (dtmax - dtmin);
            pl.tail[2].g = gc_build_geom(taps, rows, dtmin, p.nrows, pl.tail[2].Wp, cic, p.KC, gc_bld_max(pl.BM));
        }
        // 256-column geometry of the per-step LSTM GEMM (tail[2] as well; FullSubNet's sub-band layers): 2 x 2 waves of 64 x 128 -
        // a staged K row feeds twice the matrix work of the 128 x 128 tile and a weight chunk is fetched once per 256 columns.
        // Measured in round 4 on every 128-row pointwise layer: + 1.7 % on FullSubNet (two streams of step launches fill each
        // other's tails), 1 - 8 % SLOWER on the LSTM input projections, DPCRN's and the conformer's pointwise layers (two
        // workgroups per CU instead of three, coarser tails; with loads and epilogue ablated both tiles run the same
        // 132 - 134 TFLOP/s, so the K loop itself gains nothing) - hence only here
        static const int wide128_env = getenv("SE_GC_WIDE128") ? atoi(getenv("SE_GC_WIDE128")) : 1;
        if (wide128_env && pl.BN == 128 && pl.BM == 128 && taps.ntaps == 1 && pw_chunks && epi == EPI_LSTM && cic * 256 <= 6 * 1024) {
            pl.tail[2].BN = 256;
            pl.tail[2].Wp = 256;
            pl.tail[2].g = gc_build_geom(taps, rows, dtmin, p.nrows, 256, cic, p.KC, gc_bld_max(pl.BM));
        }
        // unit-flattened geometries (GCParams::flat_upr): every 32-frame unit of the tile with its own halo
        static const int flat_env = getenv("SE_GC_FLAT") ? atoi(getenv("SE_GC_FLAT")) : 1;
        if (flat_env && pl.BM == 64 && pl.BN == 128 && epi != EPI_LSTM && p.causal && dtmin >= -4 && Z == 1) {
            const int uw = 32 - dtmin;      // 32 or 36
            pl.flat_uw = uw;
            const int nbk = gc_bld_max(pl.BM);
            if ((long)cic * p.nrows * 4 * uw <= (long)nbk * 256 * 4) {
                pl.flat[0].BN = 128;
                pl.flat[0].Wp = 4 * uw;
                pl.flat[0].g = gc_build_geom(taps, rows, dtmin, p.nrows, 4 * uw, cic, p.KC, nbk);
            }
            if (pl.BM == 64 && (long)cic * p.nrows * 8 * uw <= 4608) {      // (5 x 256 groups, 3 workgroups' LDS - as the plain 64 x 256 tile)
                pl.flat[1].BN = 256;
                pl.flat[1].Wp = 8 * uw;
                pl.flat[1].g = gc_build_geom(taps, rows, dtmin, p.nrows, 8 * uw, cic, p.KC, nbk);
            }
        }
        if (pl.BN == 128 && (pl.BM == 64 || pl.BM == 128) && epi != EPI_LSTM) {
            pl.tail[1].BN = 64;
            pl.tail[1].Wp = 64 + (dtmax - dtmin);
            pl.tail[1].g = gc_build_geom(taps, rows, dtmin, p.nrows, pl.tail[1].Wp, cic, p.KC, gc_bld_max(pl.BM));
        }
    }
    const int nch0 = (C0 + cic - 1) / cic, nch1 = (Cin - C0 + cic - 1) / cic;
    p.nchunks = nch0 + nch1;
    p.M = M;
    p.Mp = ((M + pl.BM - 1) / pl.BM) * pl.BM;
    p.n_mtiles = p.Mp / pl.BM;
    p.act = act;
    p.epi = epi;
    p.si = si;
    p.so = so;
    p.po = po;
    p.Z = Z;
    p.C0 = C0;
    p.C1 = Cin - C0;
    // pack weights: [z][chunk][k_local][Mp]; chunks of segment 0 (channels < C0) first, then segment 1
    const size_t per_z = (size_t)std::max(p.nchunks, 1) * p.KCp * p.Mp;
    std::vector<float> packed(per_z * Z, 0.f);
    for (int z = 0; z < Z; ++z)
        for (int m = 0; m < M; ++m)
            for (int ci = 0; ci < Cin; ++ci)
                for (int j = 0; j < taps.ntaps; ++j) {
                    const int cs = ci < C0 ? ci : ci - C0;
                    const int chunk = (ci < C0 ? 0 : nch0) + cs / cic, cil = cs % cic;
                    const size_t dst = z * per_z + ((size_t)chunk * p.KCp + (cil * taps.ntaps + j)) * p.Mp + m;
                    packed[dst] = w[(((size_t)z * M + m) * Cin + ci) * taps.ntaps + j];
                }
    p.A_z = (long)per_z;
    pl.dA = to_device(packed);
    p.A = pl.dA;
    // direct path (gc_small_kernel): plain weights [z][ci][tap][MM] for layers with <= 4 output channels
    static const int small_env = getenv("SE_GC_SMALL") ? atoi(getenv("SE_GC_SMALL")) : 1;
    if (small_env && M <= 4 && (epi == EPI_ACT || epi == EPI_ADD || epi == EPI_MUL || (epi == EPI_GLU && M % 2 == 0))) {
        const int MM = M <= 1 ? 1 : (M <= 2 ? 2 : 4);
        std::vector<float> ws((size_t)Z * Cin * taps.ntaps * MM, 0.f);
        for (int z = 0; z < Z; ++z)
            for (int m = 0; m < M; ++m)
                for (int ci = 0; ci < Cin; ++ci)
                    for (int j = 0; j < taps.ntaps; ++j)
                        ws[(((size_t)z * Cin + ci) * taps.ntaps + j) * MM + m] = w[(((size_t)z * M + m) * Cin + ci) * taps.ntaps + j];
        pl.dWs = to_device(ws);
        p.Ws = pl.dWs;
        // exact tap extents for the LDS-tiled form of the direct kernel
        pl.small.dfmin = pl.small.dfmax = taps.df[0];
        pl.small.dtmin = pl.small.dtmax = taps.dt[0];
        for (int j = 1; j < taps.ntaps; ++j) {
            pl.small.dfmin = std::min(pl.small.dfmin, taps.df[j]);
            pl.small.dfmax = std::max(pl.small.dfmax, taps.df[j]);
            pl.small.dtmin = std::min(pl.small.dtmin, taps.dt[j]);
            pl.small.dtmax = std::max(pl.small.dtmax, taps.dt[j]);
        }
    }
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

// Device ranges whose tensors may be over-read by up to 12 B past their last element (the engine arenas: every
// allocation is followed by more arena or by the arena's tail slack).  Pointwise layers stage their patch in 16 B groups;
// a group that straddles the end of a row only feeds output columns that are never stored, so the only requirement on
// the over-read is that it stays inside mapped memory.
static std::vector<std::pair<const char*, const char*>>& gc_safe_ranges() {
    static std::vector<std::pair<const char*, const char*>> r;
    return r;
}
static std::mutex& gc_safe_mutex() {       // engines may be created / destroyed from different host threads
    static std::mutex m;
    return m;
}
void gc_register_overread_range(const void* lo, size_t bytes) {
    std::lock_guard<std::mutex> lk(gc_safe_mutex());
    gc_safe_ranges().emplace_back(static_cast<const char*>(lo), static_cast<const char*>(lo) + bytes);
}
void gc_unregister_overread_range(const void* lo) {
    std::lock_guard<std::mutex> lk(gc_safe_mutex());
    auto& r = gc_safe_ranges();
    for (size_t i = 0; i < r.size(); ++i)
        if (r[i].first == lo) {
            r.erase(r.begin() + i);
            return;
        }
}
static bool gc_overread_ok(const void* ptr) {
    if (!ptr) return true;
    std::lock_guard<std::mutex> lk(gc_safe_mutex());
    for (const auto& r : gc_safe_ranges())
        if (ptr >= r.first && static_cast<const char*>(ptr) + 16 <= r.second) return true;
    return false;
}

void gc_free_plan(GCPlan& pl) {
    if (pl.dA) (void)hipFree(pl.dA);
    if (pl.dWs) (void)hipFree(pl.dWs);
    if (pl.dDesc) (void)hipFree(pl.dDesc);
    if (pl.dDesc4) (void)hipFree(pl.dDesc4);
    pl.dDesc4 = nullptr;
    if (pl.qt2.g.tab) (void)hipFree(pl.qt2.g.tab);
    if (pl.qt2.g.desc) (void)hipFree(pl.qt2.g.desc);
    if (pl.qt2.g.desc4) (void)hipFree(pl.qt2.g.desc4);
    pl.qt2 = GCTail{};
    for (auto& t : pl.tail) {
        if (t.g.tab) (void)hipFree(t.g.tab);
        if (t.g.desc) (void)hipFree(t.g.desc);
        if (t.g.desc4) (void)hipFree(t.g.desc4);
        t = GCTail{};
    }
    for (auto& t : pl.flat) {
        if (t.g.tab) (void)hipFree(t.g.tab);
        if (t.g.desc) (void)hipFree(t.g.desc);
        if (t.g.desc4) (void)hipFree(t.g.desc4);
        t = GCTail{};
    }
    pl.dWs = nullptr;
    pl.dDesc = nullptr;
    if (pl.dBias) (void)hipFree(pl.dBias);
    if (pl.dSlope) (void)hipFree(pl.dSlope);
    if (pl.dTab) (void)hipFree(pl.dTab);
    if (pl.dBiasPad) (void)hipFree(pl.dBiasPad);
    if (pl.dPostScale) (void)hipFree(pl.dPostScale);
    if (pl.dPostShift) (void)hipFree(pl.dPostShift);
    pl.dPostScale = pl.dPostShift = nullptr;
    pl.dTab = nullptr;
    pl.dBiasPad = nullptr;
    pl.dA = pl.dBias = pl.dSlope = nullptr;
}

// kernel variants with unit-flattened column tiles (GCParams::flat_upr) that are instantiated: the tiles whose last time tile
// of a T = 401 row is mostly padding, for the epilogues the zoo uses on them
// (64-row tiles only.  Measured in round 6 on one box, batch 256, T = 401, flattened against plain tiles with identical output
// hashes: 64 -> 64 channels at F = 79 on the 64 x 256 tile 3.43 -> 3.17 ms; 32 -> 32 channels on the 32 x 128 tile 1.13 -> 1.26 ms -
// that tile is bound by its staging, not by matrix slots, and a unit's own halo is 9 % more bytes; 64 -> 128 channels on the
// 128 x 128 tile 5.61 -> 5.63 ms against three full tiles + the 32-column tail launch, CTSNet as a whole 3 138 -> 3 017 utt/s)
constexpr bool gc_flat_inst(int BM, int BN, int EPI, int UWv) {
    const bool tile = BM == 64 && (BN == 256 || BN == 128);
    if (!tile) return false;
    if (UWv == 36) return EPI == EPI_ACT || EPI == EPI_GLU || (EPI == EPI_CMB && BN == 256);
    if (UWv == 32) return EPI == EPI_ACT || (EPI == EPI_ADD && BN == 128);
    return false;
}
static bool gc_flat_supported(int BM, int BN, int epi, int uw) {
    switch (epi) {
        case EPI_ACT: return gc_flat_inst(BM, BN, EPI_ACT, uw);
        case EPI_ADD: return gc_flat_inst(BM, BN, EPI_ADD, uw);
        case EPI_GLU: return gc_flat_inst(BM, BN, EPI_GLU, uw);
        case EPI_CMB: return gc_flat_inst(BM, BN, EPI_CMB, uw);
        default: return false;
    }
}
template <int BM, int BN, int WM, int WN, int EPI, bool NRMv, int UWv>
static void gc_launch_flat_k(const GCParams& p, long nblk, size_t lds, hipStream_t stream) {
    static bool attr_fl[64] = {};
    if (first_on_device(attr_fl)) {
        SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gc_kernel<BM, BN, WM, WN, EPI, false, false, false, NRMv, UWv>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL((gc_kernel<BM, BN, WM, WN, EPI, false, false, false, NRMv, UWv>), dim3((unsigned)nblk), dim3(256), lds, stream, p);
    SE_HIP(hipGetLastError());
}

template <int BM, int BN, int WM, int WN, int EPI, bool RES = false>
static void gc_launch_e(const GCParams& p_in, hipStream_t stream) {
    // epilogue: 4*BM row parameters + one transposition strip per wave (rows x (cols + 4))
    const size_t epi = (size_t)(4 * (BM / WM) * 36) * sizeof(float);
    const size_t lds = gc_lds_bytes(p_in, BM, epi, RES ? p_in.nbuf : 2);
    static bool attr_set[64] = {};
    if (first_on_device(attr_set)) {
        SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gc_kernel<BM, BN, WM, WN, EPI, RES>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const long nblk = (long)p_in.Z * (p_in.flat_upr ? 1 : p_in.B) * p_in.Qt * p_in.n_ttiles * p_in.n_mtiles;
    SE_CHECK(nblk > 0 && nblk < (1L << 31), "grid size");
    const GCParams& p = p_in;
    if (p.flat_upr) {        // unit-flattened column tiles (gc_launch decided; only the instantiated variants get here)
        const int uw = p.Wp / (BN / 32);
        SE_CHECK(!RES && !p.fz && !p.trim && !p.qt2 && p.pw4 && p.t_base == 0, "gc_launch: flattened tiles on a launch they do not cover");
        const bool nrm = p.nrm0 || p.nrm1;
        if constexpr (!RES && gc_flat_inst(BM, BN, EPI, 36)) {
            if (uw == 36) {
                if constexpr (EPI == EPI_ACT && BM == 64) {
                    if (nrm) {
                        SE_CHECK(p.causal && p.Z == 1 && p.C0 + p.C1 <= GC_NRM_MAXC, "gc_launch: on-the-fly InstanceNorm needs causal taps and <= 128 input channels");
                        gc_launch_flat_k<BM, BN, WM, WN, EPI, true, 36>(p, nblk, lds, stream);
                        return;
                    }
                }
                SE_CHECK(!nrm, "gc_launch: this flattened variant cannot normalise its sources on the fly");
                gc_launch_flat_k<BM, BN, WM, WN, EPI, false, 36>(p, nblk, lds, stream);
                return;
            }
        }
        if constexpr (!RES && gc_flat_inst(BM, BN, EPI, 32)) {
            if (uw == 32) {
                if constexpr (EPI == EPI_ACT && BM == 64) {
                    if (nrm) {
                        SE_CHECK(p.causal && p.Z == 1 && p.C0 + p.C1 <= GC_NRM_MAXC, "gc_launch: on-the-fly InstanceNorm needs causal taps and <= 128 input channels");
                        gc_launch_flat_k<BM, BN, WM, WN, EPI, true, 32>(p, nblk, lds, stream);
                        return;
                    }
                }
                SE_CHECK(!nrm, "gc_launch: this flattened variant cannot normalise its sources on the fly");
                gc_launch_flat_k<BM, BN, WM, WN, EPI, false, 32>(p, nblk, lds, stream);
                return;
            }
        }
        SE_CHECK(false, "gc_launch: no flattened variant of this kernel");
    }
    if constexpr ((EPI == EPI_ACT || EPI == EPI_ADD) && !RES) {
        if (p.fz) {          // branch interaction folded into the store (GCParams::fz)
            SE_CHECK(!p.trim && !p.stats, "gc_launch: no trimming / statistics variant of the kernel with the folded interaction");
            static bool attr_fz[64] = {};
            if (first_on_device(attr_fz)) {
                SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gc_kernel<BM, BN, WM, WN, EPI, false, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            }
            hipLaunchKernelGGL((gc_kernel<BM, BN, WM, WN, EPI, false, false, true>), dim3((unsigned)nblk), dim3(256), lds, stream, p);
            SE_HIP(hipGetLastError());
            return;
        }
    }
    SE_CHECK(!p.fz, "gc_launch: this kernel variant cannot fold the branch interaction into its store");
    if constexpr (EPI == EPI_ACT && BM == 64 && !RES) {
        if (p.nrm0 || p.nrm1) {      // sources normalised on the fly (GCParams::nrm0 / nrm1)
            SE_CHECK(!p.trim && p.causal && !p.qt2 && p.Z == 1 && p.C0 + p.C1 <= GC_NRM_MAXC,
                     "gc_launch: on-the-fly InstanceNorm needs causal taps, one-row tiles and <= 128 input channels");
            static bool attr_nrm[64] = {};
            if (first_on_device(attr_nrm)) {
                SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gc_kernel<BM, BN, WM, WN, EPI, false, false, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            }
            hipLaunchKernelGGL((gc_kernel<BM, BN, WM, WN, EPI, false, false, false, true>), dim3((unsigned)nblk), dim3(256), lds, stream, p);
            SE_HIP(hipGetLastError());
            return;
        }
    }
    SE_CHECK(!p.nrm0 && !p.nrm1, "gc_launch: this kernel variant cannot normalise its sources on the fly");
    if constexpr (EPI == EPI_ACT && !RES) {
        if (p.trim) {
            static bool attr_trim[64] = {};
            if (first_on_device(attr_trim)) {
                SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gc_kernel<BM, BN, WM, WN, EPI, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            }
            hipLaunchKernelGGL((gc_kernel<BM, BN, WM, WN, EPI, false, true>), dim3((unsigned)nblk), dim3(256), lds, stream, p);
            SE_HIP(hipGetLastError());
            return;
        }
    }
    SE_CHECK(!p.trim, "gc_launch: no trimming variant of this kernel");
    hipLaunchKernelGGL((gc_kernel<BM, BN, WM, WN, EPI, RES>), dim3((unsigned)nblk), dim3(256), lds, stream, p);
    SE_HIP(hipGetLastError());
}

template <int BM, int BN, int WM, int WN, bool RES = false>
static void gc_launch_t(const GCParams& p, hipStream_t stream) {
    switch (p.epi) {
        case EPI_ACT: gc_launch_e<BM, BN, WM, WN, EPI_ACT, RES>(p, stream); break;
        case EPI_ADD: gc_launch_e<BM, BN, WM, WN, EPI_ADD, RES>(p, stream); break;
        case EPI_MUL: gc_launch_e<BM, BN, WM, WN, EPI_MUL, RES>(p, stream); break;
        case EPI_GLU: gc_launch_e<BM, BN, WM, WN, EPI_GLU, RES>(p, stream); break;
        case EPI_CMB:
            if constexpr (!RES && BM >= 64) gc_launch_e<BM, BN, WM, WN, EPI_CMB>(p, stream);
            else SE_CHECK(false, "no resident-K / 32-row form of the combine epilogue");
            break;
        case EPI_LSTM:
            if constexpr (!RES) gc_launch_e<BM, BN, WM, WN, EPI_LSTM>(p, stream);
            else SE_CHECK(false, "no resident-K form of the LSTM step");
            break;
        default: SE_CHECK(false, "unknown epilogue");
    }
}
// all chunks of a source resident in LDS at once (gc_kernel RES): launches of at most one workgroup per CU whose staging fits
static bool gc_resident_fits(GCParams& p, int BM, int BM_div_WM) {
    static const int res_env = getenv("SE_GC_RES") ? atoi(getenv("SE_GC_RES")) : 1;
    if (!res_env || p.trim || p.epi == EPI_LSTM || p.epi == EPI_CMB || p.fz || p.nrm0 || p.nrm1 || p.cstats) return false;
    const long nblk = (long)p.Z * p.B * p.Q * p.n_ttiles * p.n_mtiles;
    const int nch0 = p.C0 > 0 ? (p.C0 + p.CI_C - 1) / p.CI_C : 0, nch1 = p.C1 > 0 ? (p.C1 + p.CI_C - 1) / p.CI_C : 0;
    int nb = std::max(std::max(nch0, nch1), 1);
    // (few frames per row: the chunks of the frame-online mode.  A batch-1 offline decode has as few workgroups, but 64
    // frames of matrix work per chunk to hide the next load under - the double-buffered loop is 1-3 % faster there)
    if (nblk > 256 || nb <= 2 || p.Tout - p.t_base > 64) return false;
    // as many chunks per round trip as the LDS holds (a longer source goes in groups)
    const size_t epi = (size_t)(4 * BM_div_WM * 36) * sizeof(float);
    while (nb > 2 && gc_lds_bytes(p, BM, epi, nb) > 158 * 1024) --nb;
    if (nb <= 2) return false;
    p.nbuf = nb;
    return true;
}

bool gc_nrm_supported(const GCPlan& pl) {
    static const bool on = !(getenv("SE_IN_FOLD") && atoi(getenv("SE_IN_FOLD")) == 0);
    return on && !pl.p.Ws && pl.p.epi == EPI_ACT && pl.BM == 64 && pl.p.causal && pl.p.Z == 1 && pl.p.C0 + pl.p.C1 <= GC_NRM_MAXC;
}

bool gc_stats_supported(const GCPlan& pl) {
    return !pl.p.Ws && (pl.p.epi == EPI_GLU || (pl.p.epi == EPI_ACT && pl.BM == 64));
}

void gc_launch(const GCPlan& pl, GCParams p, hipStream_t stream) {
    // p.t_base (default 0): first output frame of the launch - frame-online chunks only produce the frames behind their
    // history columns.  A multiple of 4, so that the 16 B staging groups keep their alignment to frame 0.
    SE_CHECK(p.C0 == pl.p.C0 && p.C1 == pl.p.C1, "gc_launch: source channel split differs from the plan");
    SE_CHECK(!p.dst_elu || (p.epi == EPI_GLU && p.Z == 1), "gc_launch: the second (ELU) store belongs to the gated epilogue");
    if (p.epi == EPI_LSTM && p.first_step && p.C1 == 0) p.C0 = 0;       // h_{-1} = 0: no matrix work (a step that also projects its input keeps both)
    if (p.t_base > 0 && gc_thin_launch(p, stream)) return;       // (any first frame)
    if (p.tb_soft) p.t_base &= ~3;
    const int tb = p.t_base, Tspan = p.Tout - tb;
    SE_CHECK(tb >= 0 && (tb & 3) == 0 && Tspan > 0, "gc_launch: first output frame must be a multiple of 4 below Tout");
    SE_CHECK(tb == 0 || (!p.stats && !p.cstats), "gc_launch: the statistics epilogue needs whole rows");
    p.n_ttiles = (Tspan + pl.BN - 1) / pl.BN;
    p.Qt = p.Q;
    p.qt2 = 0;
    p.qq_off = 0;
    static const int dbg_env = getenv("SE_GC_DBG") ? atoi(getenv("SE_GC_DBG")) : 0;
    p.dbg = dbg_env;
    // patch offsets inside one staged chunk are 32-bit (the 64-bit part of an address is the per-block / per-chunk base)
    SE_CHECK((double)p.CI_C * (double)std::max(p.s0_c, p.s1_c) + (double)p.Fin * (double)std::max(p.s0_f, p.s1_f) + p.Tin < 1.0e9,
             "gc_launch: source plane too large for 32-bit patch byte offsets");
    static const int pw4_env = getenv("SE_GC_PW4") ? atoi(getenv("SE_GC_PW4")) : 1;
    // 16 B staging groups: exact when no group straddles the end of a row (Tin % 4 == 0); for causal tap sets a straddling
    // group only feeds output frames >= Tin, which are never stored - then it merely has to stay inside mapped memory
    // (a tap set that looks ahead would read the straddling group's foreign frames into stored outputs: the kernel trims
    // them in LDS, GC_TRIM_TAIL)
    static const int trim_env = getenv("SE_GC_TRIM") ? atoi(getenv("SE_GC_TRIM")) : 1;
    // the trimming variant exists for the plain epilogue (DCCRN's decoder); it pays from a few thousand workgroups on - small
    // launches are latency-bound and the extra LDS stores per chunk cost them 3 % (batch 1)
    static const long trim_min = getenv("SE_GC_TRIM_MIN") ? atol(getenv("SE_GC_TRIM_MIN")) : 8192;
    const bool trim_ok = trim_env && p.epi == EPI_ACT && (long)p.Z * p.B * p.Q * p.n_ttiles * p.n_mtiles >= trim_min;
    p.pw4 = (pw4_env && p.desc4 && (p.Tin % 4 == 0 || ((p.causal || trim_ok) && gc_overread_ok(p.src0) && gc_overread_ok(p.src1)))) ? 1 : 0;
    p.trim = (p.pw4 && !p.causal && p.Tin % 4 != 0) ? 1 : 0;
    SE_CHECK(!p.stats || gc_stats_supported(pl), "gc_launch: this tile configuration has no statistics epilogue");
    SE_CHECK(!p.cstats || (gc_stats_supported(pl) && p.n_mtiles == 1 && p.Z == 1),
             "gc_launch: column statistics need a statistics tile configuration and one m-tile (M " + std::to_string(p.M) + ", BM " +
                 std::to_string(pl.BM) + ", m-tiles " + std::to_string(p.n_mtiles) + ", epilogue " + std::to_string(p.epi) + ", direct " +
                 std::to_string(p.Ws != nullptr) + ")");
    SE_CHECK(p.pw4 || p.CI_C * p.nrows * p.Wp <= gc_bld_max(pl.BM) * 256,
             "pointwise layer with a row length that is not a multiple of 4 needs its sources inside the engine arena");
    if (gc_thin_launch(p, stream)) return;
    if (p.Ws) {
        SE_CHECK(!p.fz, "gc_launch: the direct (<= 4 channel) path cannot fold the branch interaction into its store");
        SE_CHECK(!p.dst_elu, "gc_launch: the direct (<= 4 channel) path has no second (ELU) store");
        SE_CHECK(!p.nrm0 && !p.nrm1, "gc_launch: the direct (<= 4 channel) path cannot normalise its sources on the fly");
        if (p.M <= 1) gc_small_launch<1>(p, pl.small, stream);
        else if (p.M <= 2) gc_small_launch<2>(p, pl.small, stream);
        else gc_small_launch<4>(p, pl.small, stream);
        return;
    }
    // small launches: 64-column tiles double the workgroup count (and waste less of the last time tile: T = 401 is
    // 6.3 x 64).  Measured on the TCM models: 64-row layers gain up to ~1 500 workgroups of 128 columns (G2Net + 7 % at
    // B = 64, + 4 % at B = 256, + 16 % at B = 8; DCCRN's big grids lose 1 %), 128-row layers only below one workgroup
    // per CU (B = 8: + 15 %; B = 64: - 3 %)
    {
        static const int alt_env = getenv("SE_GC_ALT") ? atoi(getenv("SE_GC_ALT")) : 1;
        const GCTail& alt = pl.tail[1];
        const long nblk = (long)p.Z * p.B * p.Q * p.n_ttiles * p.n_mtiles;
        static const int alt_n64 = getenv("SE_GC_ALT_N64") ? atoi(getenv("SE_GC_ALT_N64")) : 4096;
        if (alt_env && alt.BN == 64 && nblk < (pl.BM == 64 ? alt_n64 : 256)) {
            GCParams pa = p;
            pa.n_ttiles = (Tspan + 63) / 64;
            pa.Wp = alt.Wp;
            pa.tab = alt.g.tab;
            pa.desc = alt.g.desc;
            pa.desc4 = alt.g.desc4;
            static const int alt32_env = getenv("SE_GC_ALT32") ? atoi(getenv("SE_GC_ALT32")) : 512;
            const long nblk64 = (long)p.Z * p.B * p.Q * pa.n_ttiles * p.n_mtiles;
            if (pl.BM == 128 && pl.tail[0].BN == 32 && nblk64 < alt32_env) {       // tiny launches: 32 columns
                pa.n_ttiles = (Tspan + 31) / 32;
                pa.Wp = pl.tail[0].Wp;
                pa.tab = pl.tail[0].g.tab;
                pa.desc = pl.tail[0].g.desc;
                pa.desc4 = pl.tail[0].g.desc4;
                if (gc_resident_fits(pa, 128, 32)) gc_launch_t<128, 32, 4, 1, true>(pa, stream);
                else gc_launch_t<128, 32, 4, 1>(pa, stream);
                return;
            }
            if (pl.BM == 64) {
                if (gc_resident_fits(pa, 64, 32)) gc_launch_t<64, 64, 2, 2, true>(pa, stream);
                else gc_launch_t<64, 64, 2, 2>(pa, stream);
            } else {
                gc_launch_t<128, 64, 4, 1>(pa, stream);
            }
            return;
        }
    }
    // unit-flattened column tiles (GCParams::flat_upr): equal-length offline batches whose rows do not fill their last time tile.
    // T = 401 is 13 units of 32 frames: 2 tiles of 256 / 4 tiles of 128 columns per row hold 16, the flattened batch needs
    // 13 B / 8 (or / 4) tiles - 19 % fewer workgroups for the same stored values
    {
        static const int flat_env = getenv("SE_GC_FLAT") ? atoi(getenv("SE_GC_FLAT")) : 1;
        static const long flat_min = getenv("SE_GC_FLAT_MIN") ? atol(getenv("SE_GC_FLAT_MIN")) : 1024;
        static const long wide_min = getenv("SE_GC_WIDE_MIN") ? atol(getenv("SE_GC_WIDE_MIN")) : 6144;
        const long nblk = (long)p.Z * p.B * p.Q * p.n_ttiles * p.n_mtiles;      // (128-column tiles)
        const int upr = (Tspan + 31) / 32;
        const double sbmax = 4.0 * (double)std::max(p.s0_b, p.s1_b);
        // (A/B switch: SE_GC_FLAT_QT2=0 keeps the two-row tiles where they apply)
        static const int flat_qt2 = getenv("SE_GC_FLAT_QT2") ? atoi(getenv("SE_GC_FLAT_QT2")) : 1;
        static const int qt2_env_f = getenv("SE_GC_QT2") ? atoi(getenv("SE_GC_QT2")) : 1;
        static const long qt2_min_f = getenv("SE_GC_QT2_MIN") ? atol(getenv("SE_GC_QT2_MIN")) : 4096;
        const bool would_qt2 = qt2_env_f && pl.qt2.BN == 64 && pl.BN == 128 && p.Q >= 2 && !p.stats && !p.cstats && p.pad_lo == 0 &&
                               !p.nrm0 && !p.nrm1 && nblk >= qt2_min_f;
        if (flat_env && pl.BM == 64 && pl.flat_uw && tb == 0 && p.pw4 && !p.trim && !p.fz && p.Z == 1 && nblk >= flat_min && p.B > 1 &&
            true) {
            // tile width as the plain path would choose it
            const bool wide = pl.BM == 64 && pl.flat[1].BN == 256 && nblk >= wide_min && gc_flat_supported(64, 256, p.epi, pl.flat_uw);
            const GCTail& fg = wide ? pl.flat[1] : pl.flat[0];
            const int upt = wide ? 8 : 4;
            const long tiles_plain = (long)p.B * ((Tspan + 32 * upt - 1) / (32 * upt));
            const long tiles_flat = ((long)p.B * upr + upt - 1) / upt;
            // (a unit of a later batch row adds (rows ahead) x batch stride to a 32-bit byte offset)
            const double span = ((double)(upt + upr - 1) / upr + 1.0) * sbmax;
            if (fg.BN && gc_flat_supported(pl.BM, fg.BN, p.epi, pl.flat_uw) && tiles_flat * 100 <= tiles_plain * 94 && span < 3.0e9 &&
                (wide || flat_qt2 || !would_qt2)) {
                GCParams pa = p;
                pa.flat_upr = upr;
                pa.flat_units = p.B * upr;
                pa.n_ttiles = (int)tiles_flat;
                pa.Wp = fg.Wp;
                pa.tab = fg.g.tab;
                pa.desc = fg.g.desc;
                pa.desc4 = fg.g.desc4;
                if (wide) gc_launch_t<64, 256, 1, 4>(pa, stream);
                else gc_launch_t<64, 128, 2, 2>(pa, stream);
                return;
            }
        }
    }
    // big launches of a 64-row layer: 64 x 256 tiles when the patch goes in 16 B groups (4 x fewer slots), the rows fill the
    // wide tiles and the staging buffers leave room for 3 workgroups per CU
    {
        const GCTail& wd = pl.tail[2];
        static const long wide_min = getenv("SE_GC_WIDE_MIN") ? atol(getenv("SE_GC_WIDE_MIN")) : 6144;
        const long nblk = (long)p.Z * p.B * p.Q * p.n_ttiles * p.n_mtiles;
        const int nt = (Tspan + 255) / 256;
        static const int wide_fill = getenv("SE_GC_WIDE_FILL") ? atoi(getenv("SE_GC_WIDE_FILL")) : 75;
        if (wd.BN == 256 && pl.BM == 64 && p.pw4 && nblk >= wide_min && Tspan * 100 >= nt * 256 * wide_fill &&
            (long)p.CI_C * p.nrows * wd.Wp <= 4608 /* 5 x 256 groups, 3 workgroups' LDS */) {
            GCParams pa = p;
            pa.n_ttiles = nt;
            pa.Wp = wd.Wp;
            pa.tab = wd.g.tab;
            pa.desc = wd.g.desc;
            pa.desc4 = wd.g.desc4;
            gc_launch_t<64, 256, 1, 4>(pa, stream);
            return;
        }
    }
    // big launches of the per-step LSTM GEMM: 128 x 256 tiles (two workgroups per CU)
    {
        const GCTail& wd = pl.tail[2];
        static const long wide128_min = getenv("SE_GC_WIDE128_MIN") ? atol(getenv("SE_GC_WIDE128_MIN")) : 1536;
        static const int wide128_fill = getenv("SE_GC_WIDE128_FILL") ? atoi(getenv("SE_GC_WIDE128_FILL")) : 75;
        const long nblk = (long)p.Z * p.B * p.Q * p.n_ttiles * p.n_mtiles;
        const int nt = (Tspan + 255) / 256;
        if (wd.BN == 256 && pl.BM == 128 && p.epi == EPI_LSTM && p.pw4 && !p.trim && !p.stats && nblk >= wide128_min &&
            Tspan * 100L >= (long)nt * 256 * wide128_fill) {
            GCParams pa = p;
            pa.n_ttiles = nt;
            pa.Wp = wd.Wp;
            pa.tab = wd.g.tab;
            pa.desc = wd.g.desc;
            pa.desc4 = wd.g.desc4;
            gc_launch_e<128, 256, 2, 2, EPI_LSTM>(pa, stream);
            return;
        }
    }
    // big launches of layers whose neighbouring output rows share input rows: two-row tiles (2 x 64 frames)
    {
        static const int qt2_env = getenv("SE_GC_QT2") ? atoi(getenv("SE_GC_QT2")) : 1;
        static const long qt2_min = getenv("SE_GC_QT2_MIN") ? atol(getenv("SE_GC_QT2_MIN")) : 4096;
        const long nblk = (long)p.Z * p.B * p.Q * p.n_ttiles * p.n_mtiles;
        if (qt2_env && pl.qt2.BN == 64 && pl.BN == 128 && p.Q >= 2 && !p.stats && !p.cstats && p.pad_lo == 0 && p.epi != EPI_LSTM &&
            !p.nrm0 && !p.nrm1 && nblk >= qt2_min) {
            GCParams pa = p;
            pa.qt2 = 1;
            pa.qq_off = pl.qt2_qoff;
            pa.nrows = pl.qt2_nrows;
            pa.Qt = (p.Q + 1) / 2;
            pa.n_ttiles = (Tspan + 63) / 64;
            pa.Wp = pl.qt2.Wp;
            pa.tab = pl.qt2.g.tab;
            pa.desc = pl.qt2.g.desc;
            pa.desc4 = pl.qt2.g.desc4;
            if (pl.BM == 128) gc_launch_t<128, 128, 2, 2>(pa, stream);
            else if (pl.BM == 64) gc_launch_t<64, 128, 2, 2>(pa, stream);
            else gc_launch_t<32, 128, 1, 4>(pa, stream);
            return;
        }
    }
    // the last time tile of a row, when it is at most half full, goes to a narrower kernel (own launch, same weights)
    const int full = Tspan / pl.BN, rem = Tspan - full * pl.BN;
    const GCTail* tl = nullptr;
    if (pl.tail_split && full >= 1 && rem > 0 && pl.tail[0].BN && rem <= pl.tail[0].BN) tl = &pl.tail[0];
    if (tl) {
        GCParams pt = p;
        pt.t_base = tb + full * pl.BN;
        pt.n_ttiles = 1;
        pt.Wp = tl->Wp;
        pt.tab = tl->g.tab;
        pt.desc = tl->g.desc;
        pt.desc4 = tl->g.desc4;
        p.n_ttiles = full;
        if (pl.BM == 128 && tl->BN == 32) gc_launch_t<128, 32, 4, 1>(pt, stream);
        else if (pl.BM == 128 && tl->BN == 64) gc_launch_t<128, 64, 4, 1>(pt, stream);
        else if (pl.BM == 64 && tl->BN == 64) gc_launch_t<64, 64, 2, 2>(pt, stream);
        else SE_CHECK(false, "no gemmconv tail tile config");
    }
    if (pl.BM == 128 && pl.BN == 128) gc_launch_t<128, 128, 2, 2>(p, stream);
    else if (pl.BM == 64 && pl.BN == 128) gc_launch_t<64, 128, 2, 2>(p, stream);
    else if (pl.BM == 32 && pl.BN == 128) gc_launch_t<32, 128, 1, 4>(p, stream);
    else if (pl.BM == 128 && pl.BN == 64) gc_launch_t<128, 64, 4, 1>(p, stream);
    else if (pl.BM == 64 && pl.BN == 64) gc_launch_t<64, 64, 2, 2>(p, stream);
    else SE_CHECK(false, "no gemmconv tile config");
}

}  // namespace se
