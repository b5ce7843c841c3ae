// Persistent LSTM recurrence for small hidden sizes (H = 64 / 128): one launch walks all time steps.
//
// Reference: torch.nn.LSTM cell (gate order i,f,g,o; c' = sig(f) c + sig(i) tanh(g); h = sig(o) tanh(c')) as used by
// DCCRN's NavieComplexLSTM (DCCRN/DCCRN_cprs.py:82-92, hidden 128) and DPCRN's intra/inter LSTMs
// (DPCRN/DPCRN.py:51-54, hidden 64 / 128).  The input projection W_ih x + b is a separate big GEMM (gemmconv);
// this kernel adds the recurrent term and runs the cell.
//
// MI355X mapping: a workgroup owns 16 sequences (MFMA N = 16) of one LSTM for the whole utterance.  W_hh (4H x H
// fp32 = 256 KB at H = 128) does not fit LDS, so it lives in REGISTERS as v_mfma_f32_16x16x4_f32 A-fragments:
// wave w holds the 4H/4 gate rows of its H/4 units (H*H/64 VGPRs per lane = 256 at H = 128, one wave per SIMD),
// h_{t-1} is the B operand read from an 8 KB LDS tile, the cell state never leaves registers, and the gate
// pre-activations of step t+1 are prefetched from HBM while step t runs on the matrix pipe.  Rows are
// gate-interleaved (row 4u+g) so that a lane's four accumulator registers are exactly the i,f,g,o gates of one
// (unit, sequence) pair and the cell update is lane-local.
#include "kernels.h"
#include "common.h"
#include <type_traits>

namespace se {

typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int N, typename F>
__device__ __forceinline__ void static_for_l(F&& f) {
    if constexpr (N > 0) {
        static_for_l<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__device__ __forceinline__ float fast_sigmoid(float x) { return fm_sigmoid(x); }
__device__ __forceinline__ float fast_tanh(float x) { return fm_tanh(x); }

template <int H>
__global__ __launch_bounds__(256, 1) void lstm_persist_kernel(const LstmPersistArgs a) {
    constexpr int MT = H / 16;     // 16-row M tiles per wave (wave owns H gate rows = H/4 units)
    constexpr int KG = H / 4;      // k groups of 4
    __shared__ float hs[2][H * 16];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    // XCD-aware order (see lstm_persist4_kernel): the two blocks that share a 128 B line of every gate row meet in one L2
    int lid;
    {
        const int nblk = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, slot = id >> 3, q8 = nblk >> 3, r8 = nblk & 7;
        lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    }
    const int n0 = lid * 16;
    const int z = blockIdx.y % a.Z, o = blockIdx.y / a.Z;
    const int n = n0 + l15;
    const bool rev = (a.reverse >> z) & 1;
    const bool col_ok = n < a.S;

    // ---- W_hh fragments -> registers (row-major [4H][H], rows gate-interleaved)
    const float* __restrict__ W = a.whh + (long)z * a.whh_z;
    float wa[MT][KG];
    static_for_l<MT>([&](auto M_) {
        constexpr int mt = decltype(M_)::value;
        static_for_l<KG>([&](auto K_) {
            constexpr int kg = decltype(K_)::value;
            wa[mt][kg] = W[(long)(wave * H + mt * 16 + l15) * H + 4 * kg + l4];
        });
    });

    const float* __restrict__ gx = a.gx + (long)z * a.gx_z + (long)o * a.gx_o + n;
    float* __restrict__ out = a.out + (long)z * a.out_z + (long)o * a.out_o + n;

    float c[MT];
    float gcur[MT][4], gnxt[MT][4];
    // streaming: continue from the carried state (a.st_h / a.st_c, [H][S] per LSTM) instead of zeros
    const bool carry = a.st_h != nullptr;
    float* __restrict__ sth = carry ? a.st_h + (long)z * a.st_z : nullptr;
    float* __restrict__ stc = carry ? a.st_c + (long)z * a.st_z : nullptr;
    static_for_l<MT>([&](auto M_) {
        constexpr int mt = decltype(M_)::value;
        const int u = wave * (H / 4) + mt * 4 + l4;
        c[mt] = (carry && col_ok) ? stc[(long)u * a.S + n] : 0.f;
    });
    for (int i = tid; i < H * 16; i += 256) {
        const int u = i >> 4, col = n0 + (i & 15);
        hs[0][i] = (carry && col < a.S) ? sth[(long)u * a.S + col] : 0.f;
    }

    auto load_gx = [&](int step, float (&g)[MT][4]) {
        const int t = rev ? a.T - 1 - step : step;
        const float* gp = gx + (long)t * a.gx_t;
        static_for_l<MT>([&](auto M_) {
            constexpr int mt = decltype(M_)::value;
            const int row = wave * H + mt * 16 + l4 * 4;          // row of gate i of this lane's unit
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) g[mt][g4] = col_ok ? gp[(long)(row + g4) * a.gx_row] : 0.f;
        });
    };
    load_gx(0, gcur);
    __syncthreads();

    for (int step = 0; step < a.T; ++step) {
        const int cur = step & 1;
        if (step + 1 < a.T) load_gx(step + 1, gnxt);
        floatx4 acc[MT];
        static_for_l<MT>([&](auto M_) {
            constexpr int mt = decltype(M_)::value;
            acc[mt] = floatx4{0.f, 0.f, 0.f, 0.f};
        });
        if (step > 0 || carry) {
            // h_{t-1} operand reads run PF k-groups ahead of the MFMAs that consume them (ring of 2 * PF registers): the
            // compiler's own schedule waits lgkmcnt(0) behind every LDS read, 16-32 exposed LDS round trips per step
            const float* hb = &hs[cur][l4 * 16 + l15];
            constexpr int PF = 4;
            float bq[2 * PF];
            static_for_l<PF>([&](auto K_) {
                constexpr int kg = decltype(K_)::value;
                bq[kg] = hb[kg * 64];
            });
            static_for_l<KG>([&](auto K_) {
                constexpr int kg = decltype(K_)::value;
                if constexpr (kg + PF < KG) bq[(kg + PF) % (2 * PF)] = hb[(kg + PF) * 64];
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                const float bv = bq[kg % (2 * PF)];
                static_for_l<MT>([&](auto M_) {
                    constexpr int mt = decltype(M_)::value;
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[mt][kg], bv, acc[mt], 0, 0, 0);
                });
                __builtin_amdgcn_sched_group_barrier(0x008, MT, 0);
            });
        }
        const int t = rev ? a.T - 1 - step : step;
        float* op = out + (long)t * a.out_t;
        static_for_l<MT>([&](auto M_) {
            constexpr int mt = decltype(M_)::value;
            const int u = wave * (H / 4) + mt * 4 + l4;
            const float gi = acc[mt][0] + gcur[mt][0];
            const float gf = acc[mt][1] + gcur[mt][1];
            const float gg = acc[mt][2] + gcur[mt][2];
            const float go = acc[mt][3] + gcur[mt][3];
            const float cn = fast_sigmoid(gf) * c[mt] + fast_sigmoid(gi) * fast_tanh(gg);
            c[mt] = cn;
            const float h = fast_sigmoid(go) * fast_tanh(cn);
            hs[cur ^ 1][u * 16 + l15] = h;
            if (col_ok) op[(long)u * a.out_row] = h;
            if (carry && col_ok && step == a.T - 1) {
                sth[(long)u * a.S + n] = h;
                stc[(long)u * a.S + n] = cn;
            }
        });
        static_for_l<MT>([&](auto M_) {
            constexpr int mt = decltype(M_)::value;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) gcur[mt][g4] = gnxt[mt][g4];
        });
        __syncthreads();
    }
}

// ---- small sequence counts: 4 sequences per workgroup on v_mfma_f32_4x4x1_16b_f32 -----------------------------------
// DCCRN's complex LSTM runs S = B sequences per (weight set, input): 16-sequence tiles give 16 * Z * O = 64 workgroups at
// B = 256 and leave 3/4 of the chip idle for 2 x 3.6 ms of every step.  The 4x4x1 MFMA (16 independent 4x4 blocks, K = 1)
// maps block <-> hidden unit, block row <-> gate, block column <-> sequence: a workgroup then owns 4 sequences, there are
// 4x as many workgroups, and a lane still holds the four gates of one (unit, sequence) pair for a lane-local cell update.
// Same register-resident W_hh (H*H/64 VGPRs per lane), h_{t-1} through a [4][H] LDS tile read as 16 B per 4 k values.
template <int H>
__global__ __launch_bounds__(256, 1) void lstm_persist4_kernel(const LstmPersistArgs a) {
    constexpr int RG = H / 64;         // 64-row groups (16 units) per wave: the wave owns H gate rows = H/4 units
    __shared__ __attribute__((aligned(16))) float hs[2][4 * H];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = lane >> 2, j = lane & 3;                  // MFMA block (unit) and column (sequence) of this lane
    // XCD-aware order (block id i runs on XCD i % 8): the 4 sequences of a block are 16 B of every gate-row line, so the
    // blocks that share those lines are made neighbours inside one XCD's L2 (round robin put them on 4-8 different
    // XCDs and the L2 <-> fabric counters showed the gate pre-activations fetched 4x)
    int lid;
    {
        const int nblk = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, slot = id >> 3, q8 = nblk >> 3, r8 = nblk & 7;
        lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    }
    const int n0 = lid * 4;
    const int z = blockIdx.y % a.Z, o = blockIdx.y / a.Z;
    const int n = n0 + j;
    const bool rev = (a.reverse >> z) & 1;
    const bool col_ok = n < a.S;

    // A operand of block `blk`, row i = lane & 3: W[4 * unit + i][k]   (rows are gate-interleaved)
    const float* __restrict__ W = a.whh + (long)z * a.whh_z;
    float wa[RG][H];
    static_for_l<RG>([&](auto R_) {
        constexpr int rg = decltype(R_)::value;
        const int row = 4 * (wave * (H / 4) + rg * 16 + blk) + j;
        static_for_l<H>([&](auto K_) {
            constexpr int k = decltype(K_)::value;
            wa[rg][k] = W[(long)row * H + k];
        });
    });

    const float* __restrict__ gx = a.gx + (long)z * a.gx_z + (long)o * a.gx_o + min(n, a.S - 1);
    float* __restrict__ out = a.out + (long)z * a.out_z + (long)o * a.out_o + n;
    float c[RG], gcur[RG][4], gnxt[RG][4];
    static_for_l<RG>([&](auto R_) {
        constexpr int rg = decltype(R_)::value;
        c[rg] = 0.f;
    });
    for (int i = tid; i < 4 * H; i += 256) hs[0][i] = 0.f;

    auto load_gx = [&](int step, float (&g)[RG][4]) {
        const int t = rev ? a.T - 1 - step : step;
        const float* gp = gx + (long)t * a.gx_t;
        static_for_l<RG>([&](auto R_) {
            constexpr int rg = decltype(R_)::value;
            const int row = 4 * (wave * (H / 4) + rg * 16 + blk);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) g[rg][g4] = gp[(long)(row + g4) * a.gx_row];
        });
    };
    load_gx(0, gcur);
    __syncthreads();

    for (int step = 0; step < a.T; ++step) {
        const int cur = step & 1;
        if (step + 1 < a.T) load_gx(step + 1, gnxt);
        // four accumulator chains per row group (k mod 4): a v_mfma_f32_4x4x1 that depends on the previous one issues 16 cycles
        // behind it, an independent one 8 (tools/mfma4bench.cpp) - with one chain per row group this loop ran at half rate
        floatx4 acc4[RG][4];
        static_for_l<RG>([&](auto R_) {
            constexpr int rg = decltype(R_)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc4[rg][q] = floatx4{0.f, 0.f, 0.f, 0.f};
        });
        if (step > 0) {
            const floatx4* hb = reinterpret_cast<const floatx4*>(&hs[cur][j * H]);     // B operand: h_{t-1}[k] of sequence j
            // operand reads two 16 B groups ahead of the MFMAs (ring of 3), see lstm_persist_kernel.  At 490 of 512
            // registers the compiler's scheduler will not hoist a read on its own: the reads and their waits are asm
            // (the wait names the register it releases so that no consumer can move above it)
            floatx4 bq[3];
            const unsigned haddr = (unsigned)(size_t)hb;           // LDS byte address (low 32 bits of the flat pointer)
#define P4_READ(Q, K4) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(Q) : "v"(haddr), "n"((K4) * 16) : "memory")
#define P4_WAIT(Q, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(Q))
            P4_READ(bq[0], 0);
            P4_READ(bq[1], 1);
            static_for_l<H / 4>([&](auto K4_) {
                constexpr int k4 = decltype(K4_)::value;
                if constexpr (k4 + 2 < H / 4) {
                    P4_READ(bq[(k4 + 2) % 3], k4 + 2);
                    P4_WAIT(bq[k4 % 3], 2);
                } else if constexpr (k4 + 1 < H / 4) {
                    P4_WAIT(bq[k4 % 3], 1);
                } else {
                    P4_WAIT(bq[k4 % 3], 0);
                }
                const floatx4 b = bq[k4 % 3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    static_for_l<RG>([&](auto R_) {
                        constexpr int rg = decltype(R_)::value;
                        acc4[rg][q] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[rg][4 * k4 + q], b[q], acc4[rg][q], 0, 0, 0);
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
#undef P4_READ
#undef P4_WAIT
        }
        floatx4 acc[RG];
        static_for_l<RG>([&](auto R_) {
            constexpr int rg = decltype(R_)::value;
            acc[rg] = (acc4[rg][0] + acc4[rg][1]) + (acc4[rg][2] + acc4[rg][3]);
        });
        const int t = rev ? a.T - 1 - step : step;
        float* op = out + (long)t * a.out_t;
        static_for_l<RG>([&](auto R_) {
            constexpr int rg = decltype(R_)::value;
            const int u = wave * (H / 4) + rg * 16 + blk;
            const float gi = acc[rg][0] + gcur[rg][0];
            const float gf = acc[rg][1] + gcur[rg][1];
            const float gg = acc[rg][2] + gcur[rg][2];
            const float go = acc[rg][3] + gcur[rg][3];
            const float cn = fast_sigmoid(gf) * c[rg] + fast_sigmoid(gi) * fast_tanh(gg);
            c[rg] = cn;
            const float h = fast_sigmoid(go) * fast_tanh(cn);
            hs[cur ^ 1][j * H + u] = h;
            if (col_ok) op[(long)u * a.out_row] = h;
        });
        static_for_l<RG>([&](auto R_) {
            constexpr int rg = decltype(R_)::value;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) gcur[rg][g4] = gnxt[rg][g4];
        });
        __syncthreads();
    }
}

void launch_lstm_persist(const LstmPersistArgs& a, hipStream_t s) {
    SE_CHECK(a.H == 64 || a.H == 128, "persistent LSTM kernel is built for H = 64 / 128");
    // few sequences: 4 per workgroup (4x4x1 MFMA) fill the chip where 16-sequence tiles would not
    static const int p4_max = getenv("SE_LSTM_P4") ? atoi(getenv("SE_LSTM_P4")) : 128;      // 0 disables
    SE_CHECK(!a.st_h || (a.st_c && a.O == 1), "persistent LSTM: carried state needs both tensors and O = 1");
    if (a.H == 128 && !a.st_h && ((a.S + 15) / 16) * a.Z * a.O <= p4_max) {
        hipLaunchKernelGGL(lstm_persist4_kernel<128>, dim3((a.S + 3) / 4, a.Z * a.O), dim3(256), 0, s, a);
        SE_HIP(hipGetLastError());
        return;
    }
    dim3 grid((a.S + 15) / 16, a.Z * a.O);
    if (a.H == 128) hipLaunchKernelGGL(lstm_persist_kernel<128>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(lstm_persist_kernel<64>, grid, dim3(256), 0, s, a);
    SE_HIP(hipGetLastError());
}

}  // namespace se
