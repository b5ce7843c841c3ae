// LSTM over SHORT sequences with the input projection inside the recurrence.
//
// Reference: DPCRN's intra-frame BiLSTM (DPCRN/DPCRN.py:51-54 `nn.LSTM(128, 64, bidirectional)`, applied at :65-71 to the
// 4 frequency rows of every (utterance, frame) pair: 102 656 sequences of 4 steps per direction at batch 256); cell as in
// k_lstm.hip (gate order i,f,g,o).
//
// Why its own kernel: the persistent kernel of k_lstm.hip is built for hundreds of steps - a workgroup loads W_hh into
// registers (64 KB) and then walks ONE tile of 16 sequences; at 4 steps the weight load and the launch of 13 312 workgroups
// are most of a tile's life (0.60 ms per layer at 0.3 MFMA-busy), and the input projection in front of it is a GEMM whose
// [4H x columns] gate tensor (840 MB per layer, both directions) is written and read back once (0.63 ms at 85 TFLOP/s: its
// output stores are 80 % of its traffic).  Here a workgroup keeps W_ih AND W_hh of one direction in registers (192 VGPRs
// per lane as v_mfma_f32_16x16x4_f32 A fragments, one wave per SIMD) and walks MANY tiles; per (tile, step) it
//   1. adds W_hh h_{t-1} (h through an LDS tile, as k_lstm.hip) to the accumulators that already hold W_ih x_t + b,
//   2. runs the cells, writes h_t (LDS + HBM),
//   3. computes W_ih x_{t+1} + b of the NEXT step (of this tile or the next one) from an LDS tile of x - two thirds of the
//      matrix work, independent of the recurrence, issued while the other waves finish step t,
//   4. moves x_{t+2} (global loads issued at the top of the step) into the free x tile.
// No gate tensor exists; x is read once per direction (8 KB per tile and step).
#include "k_lstm_short.h"
#include "common.h"
#include <algorithm>
#include <type_traits>

namespace se {

namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int N, typename F>
__device__ __forceinline__ void static_for_s(F&& f) {
    if constexpr (N > 0) {
        static_for_s<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

constexpr int LS_H = 64, LS_I = 128, LS_N = 16;      // hidden units, input features, sequences per tile

// acc[mt] += A[mt][kg] * B(kg), B from an LDS tile [K][16] (element (k, n) at k * 16 + n): the operand reads run PF k-groups
// ahead of the matrix instructions that consume them
// between(kg): code placed behind the matrix instructions of k-group kg - the cells of the step in flight run in the shadow of
// the projection's matrix instructions (one wave per SIMD: nothing else would fill the matrix pipe while the cells compute)
template <int KG, int MT, int VSLOT = 0, typename Between>
__device__ __forceinline__ void mma_lds(const float (&wa)[MT][KG], const float* __restrict__ tile, int lane, floatx4 (&acc)[MT],
                                        Between&& between) {
    const float* hb = tile + lane;       // (k = 4 kg + (lane >> 4), n = lane & 15) -> kg * 64 + lane
    constexpr int PF = 4;
    float bq[2 * PF];
    static_for_s<PF>([&](auto K_) {
        constexpr int kg = decltype(K_)::value;
        bq[kg] = hb[kg * 64];
    });
    static_for_s<KG>([&](auto K_) {
        constexpr int kg = decltype(K_)::value;
        if constexpr (kg + PF < KG) bq[(kg + PF) % (2 * PF)] = hb[(kg + PF) * 64];
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        const float bv = bq[kg % (2 * PF)];
        static_for_s<MT>([&](auto M_) {
            constexpr int mt = decltype(M_)::value;
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt][kg], bv, acc[mt], 0, 0, 0);
        });
        __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);
        between(K_);
        if constexpr (VSLOT > 0) {      // slots for the vector / transcendental instructions of `between`, spread over the k-groups
            __builtin_amdgcn_sched_group_barrier(0x002, VSLOT, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
        }
    });
}

__global__ __launch_bounds__(256, 1) void lstm_short_kernel(const LstmShortArgs a) {
    constexpr int H = LS_H, I = LS_I, MT = H / 16, KH = H / 4, KX = I / 4;
    __shared__ float hs[2][H * LS_N];
    __shared__ float xs[2][I * LS_N];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int z = blockIdx.y;
    const bool rev = (a.reverse >> z) & 1;
    const int ntn = (a.S + LS_N - 1) / LS_N;                   // tiles per outer item
    const int ntiles = ntn * a.O;
    const int G = gridDim.x;
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    if (my_tiles <= 0) return;
    const int NV = my_tiles * a.T;                             // (tile, step) pairs of this workgroup, in walking order

    // ---- weights of this direction -> registers (A fragments: lane (l15, l4) holds row m = l15 of a 16-row tile, k = l4)
    float wx[MT][KX], wh[MT][KH];
    float bs[MT][4];
    {
        const float* __restrict__ Wx = a.wih + (long)z * a.wih_z;
        const float* __restrict__ Wh = a.whh + (long)z * a.whh_z;
        const float* __restrict__ Bv = a.bias + (long)z * a.bias_z;
        static_for_s<MT>([&](auto M_) {
            constexpr int mt = decltype(M_)::value;
            const int row = wave * H + mt * 16 + l15;
            static_for_s<KX>([&](auto K_) {
                constexpr int kg = decltype(K_)::value;
                wx[mt][kg] = Wx[(long)row * I + 4 * kg + l4];
            });
            static_for_s<KH>([&](auto K_) {
                constexpr int kg = decltype(K_)::value;
                wh[mt][kg] = Wh[(long)row * H + 4 * kg + l4];
            });
#pragma unroll
            for (int g = 0; g < 4; ++g) bs[mt][g] = Bv[wave * H + mt * 16 + l4 * 4 + g];     // accumulator row of this lane
        });
    }

    // x staging: thread (n = tid & 15, channel group cg = tid >> 4) moves channels cg + 16 j, j < 8, of one (tile, step)
    const int xn = tid & 15, xcg = tid >> 4;
    float xr[I / 16];
    // walking state of the three positions in flight: v (cells), v + 1 (projection), v + 2 (loads)
    struct Pos { int tile, step; };
    auto advance = [&](Pos& p) {
        if (++p.step == a.T) {
            p.step = 0;
            p.tile += G;
        }
    };
    // (columns past the last sequence of an outer item repeat its last sequence: they compute the same h and store the same
    // value to the same address - no masks, no branches inside a step)
    auto load_x = [&](const Pos& p) {
        const int o = p.tile / ntn, nt = p.tile - o * ntn;
        const int t = rev ? a.T - 1 - p.step : p.step;
        const float* xp = a.x + (long)o * a.x_o + (long)t * a.x_t + min(nt * LS_N + xn, a.S - 1);
#pragma unroll
        for (int j = 0; j < I / 16; ++j) xr[j] = xp[(long)(xcg + 16 * j) * a.x_c];
    };
    auto store_x = [&](float* tile) {
#pragma unroll
        for (int j = 0; j < I / 16; ++j) tile[(xcg + 16 * j) * LS_N + xn] = xr[j];
    };
    auto project = [&](const float* tile, floatx4 (&acc)[MT], auto VS_, auto&& between) {
        static_for_s<MT>([&](auto M_) {
            constexpr int mt = decltype(M_)::value;
            acc[mt] = floatx4{bs[mt][0], bs[mt][1], bs[mt][2], bs[mt][3]};
        });
        mma_lds<KX, MT, decltype(VS_)::value>(wx, tile, lane, acc, between);
    };
    auto nothing = [](auto) {};

    Pos pv{(int)blockIdx.x, 0}, pl{(int)blockIdx.x, 0};
    floatx4 acc[MT], accn[MT];
    // prologue: x(0) -> xs[0], projection of v = 0, x(1) -> xs[1], loads of x(2) start inside the loop
    load_x(pl);
    store_x(xs[0]);
    advance(pl);
    if (NV > 1) load_x(pl);
    __syncthreads();
    project(xs[0], acc, std::integral_constant<int, 0>{}, nothing);
    if (NV > 1) {
        store_x(xs[1]);
    } else {      // a single (tile, step): the projection "of the next pair" below reads xs[1] - defined values, result unused
#pragma unroll
        for (int j = 0; j < I / 16; ++j) xs[1][(xcg + 16 * j) * LS_N + xn] = 0.f;
    }
    advance(pl);
    __syncthreads();

    float c[MT];
    for (int v = 0; v < NV; ++v) {
        const int cur = v & 1;
        if (v + 2 < NV) load_x(pl);                                  // x(v + 2): lands while this step's matrix work runs
        // ---- 1. recurrent term (h_{t-1} of this tile is in hs[cur]; step 0 starts from zeros)
        if (pv.step > 0) mma_lds<KH, MT, 0>(wh, hs[cur], lane, acc, nothing);
        // ---- 2. cells, and under them 3. the input projection of the next (tile, step): independent of the recurrence (behind
        // the last pair it projects a stale tile - valid LDS, result unused)
        {
            const int o = pv.tile / ntn, nt = pv.tile - o * ntn;
            const int t = rev ? a.T - 1 - pv.step : pv.step;
            const int n = min(nt * LS_N + l15, a.S - 1);
            float* __restrict__ op = a.out + (long)z * a.out_z + (long)o * a.out_o + (long)t * a.out_t + n;
            const bool first = pv.step == 0;
            // the cells of 16-row tile mt behind k-group 1 + 7 mt of the projection
            project(xs[cur ^ 1], accn, std::integral_constant<int, 5>{}, [&](auto K_) {
                constexpr int kg = decltype(K_)::value;
                if constexpr (kg % 7 == 1 && kg / 7 < MT) {
                    constexpr int mt = kg / 7;
                    const int u = wave * (H / 4) + mt * 4 + l4;
                    const float cp = first ? 0.f : c[mt];
                    const float cn = fm_sigmoid(acc[mt][1]) * cp + fm_sigmoid(acc[mt][0]) * fm_tanh(acc[mt][2]);
                    c[mt] = cn;
                    const float h = fm_sigmoid(acc[mt][3]) * fm_tanh(cn);
                    hs[cur ^ 1][u * LS_N + l15] = h;
                    op[(long)u * a.out_row] = h;
                }
            });
        }
        // ---- 4. x(v + 2) into the tile the projection of step v read (free since the last barrier)
        if (v + 2 < NV) {
            store_x(xs[cur]);
            advance(pl);
        }
        __syncthreads();
        static_for_s<MT>([&](auto M_) {
            constexpr int mt = decltype(M_)::value;
            acc[mt] = accn[mt];
        });
        advance(pv);
    }
}

}  // namespace

bool lstm_short_supported(int H, int I, int T) {
    static const bool on = !(getenv("SE_LSTM_SHORT") && atoi(getenv("SE_LSTM_SHORT")) == 0);
    return on && H == LS_H && I == LS_I && T >= 1 && T <= 16;
}

void launch_lstm_short(const LstmShortArgs& a, hipStream_t s) {
    SE_CHECK(a.T >= 1 && a.S >= 1 && a.O >= 1 && a.Z >= 1, "lstm_short: empty problem");
    const long ntiles = (long)((a.S + LS_N - 1) / LS_N) * a.O;
    int dev = 0, ncu = 256;
    SE_HIP(hipGetDevice(&dev));
    SE_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    // one workgroup per CU (192 weight registers per lane: one wave per SIMD), the directions side by side
    const int G = (int)std::min<long>(ntiles, std::max(1, ncu / a.Z));
    hipLaunchKernelGGL(lstm_short_kernel, dim3(G, a.Z), dim3(256), 0, s, a);
    SE_HIP(hipGetLastError());
}

}  // namespace se
