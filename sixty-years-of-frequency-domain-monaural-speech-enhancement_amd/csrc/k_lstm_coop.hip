// Weight-stationary LSTM recurrence for LARGE hidden sizes (H = 512 / 1024): ONE cooperative launch walks all time
// steps with W_hh spread over the register files of the whole chip.
//
// Reference: the nn.LSTM layers of LSTM/LSTM.py:18-20 (3 x 1024), CRN/CRN.py:19 (2 x 1024), GCRN's GLSTM
// (GCRN/GCRN_noncprs.py:5-39, 2 groups x 512 per layer) and FullSubNet's full-band model (sequence_model.py:66-84,
// 2 x 512).  W_hh is 4H x H fp32 = 16 MB at H = 1024: it fits neither LDS nor one CU, and streaming it from L2 /
// Infinity Cache once per time step (the per-step GEMM launch this replaces) is what bounded those models.
//
// MI355X mapping: a workgroup owns 16 hidden units (64 gate rows); wave w keeps the 16 rows of its 4 units as
// v_mfma_f32_16x16x4_f32 A-fragments in H/4 VGPRs for the whole utterance.  H/16 workgroups ("unit slices") cover
// one LSTM; the remaining CUs are filled by slicing the sequences (SS "sequence slices", weights replicated).  Per
// step and 16-sequence tile a workgroup stages h_{t-1} ([16][H], coalesced 8 B agent-scope loads) into LDS, every
// wave multiplies its rows against it, the cell update is lane-local (gate-interleaved rows) and h_t is published with
// agent-scope write-through stores to a double-buffered exchange tensor hx[2][S][H]; unit slices of the same
// sequence slice then meet at a flag barrier in global memory (one relaxed agent-scope flag word per producer, written
// after the producer's stores have been acknowledged; no cache write-back / invalidate per step).  The K order is
// permuted (k = l4*H/4 + kg) identically on both operands so that a lane's B values are contiguous (ds_read_b128).
// DESIGN.md 3.2 has the measurements and the protocols that were tried and dropped.
#include "kernels.h"
#include "common.h"
#include <cstring>
#include <type_traits>

namespace se {

typedef float floatx4 __attribute__((ext_vector_type(4)));

namespace {

template <int N, typename F>
__device__ __forceinline__ void static_for_c(F&& f) {
    if constexpr (N > 0) {
        static_for_c<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__device__ __forceinline__ float sigm(float x) { return fm_sigmoid(x); }
__device__ __forceinline__ float tanhf_fast(float x) { return fm_tanh(x); }

template <int H>
__global__ __launch_bounds__(256, 1) void lstm_coop_kernel(const LstmCoopArgs a) {
    constexpr int KQ = H / 4;          // k values per MFMA k-slot (l4)
    constexpr int LDW = H + 4;         // LDS row stride: 16 lanes x 16 B land in 64 distinct banks
    constexpr int US = H / 16;         // unit slices per LSTM
    extern __shared__ float hs[];      // [2][16][LDW]  h_{t-1} of the current / the next 16 sequences

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int us = blockIdx.x % US, ss = (blockIdx.x / US) % a.SS, z = blockIdx.x / (US * a.SS);
    const bool rev = (a.reverse >> z) & 1;
    const int NT = (a.S + 15) >> 4;

    // ---- W_hh rows of this wave's 4 units -> registers
    const int r0 = us * 64 + wave * 16;                           // first gate row (rows are 4u+g)
    floatx4 wa[KQ / 4];
    {
        const floatx4* __restrict__ W =
            reinterpret_cast<const floatx4*>(a.whh + (long)z * a.whh_z + (long)(r0 + l15) * H + l4 * KQ);
        static_for_c<KQ / 4>([&](auto J_) {
            constexpr int j = decltype(J_)::value;
            wa[j] = W[j];
        });
    }
    const int u = us * 16 + wave * 4 + l4;                        // hidden unit of this lane's accumulator
    const float* __restrict__ gx = a.gx + (long)z * a.gx_z + (long)(4 * u) * a.gx_row;
    float* __restrict__ out = a.out + (long)z * a.out_z + (long)u * a.out_row;
    float* __restrict__ cell = a.cell + ((long)z * H + u) * a.S;
    float* __restrict__ hx = a.hx + (long)z * 2 * a.S * H;
    // one arrival flag per unit slice of this (z, sequence slice): a producer stores the step number into its own word,
    // a consumer reads all US words with one wave load - no read-modify-write traffic serialised on a single counter
    unsigned* flags = a.bar + (z * a.SS + ss) * 64;

    // gate pre-activations of this lane's (unit, sequence) for the block's first tile of a step (prefetched before
    // the exchange barrier of the previous step - they do not depend on h)
    auto load_g = [&](int step, int nt, float (&g)[4]) {
        const int t = rev ? a.T - 1 - step : step;
        const int n = min(nt * 16 + l15, a.S - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = gx[(long)t * a.gx_t + q * a.gx_row + n];
    };
    float gfirst[4], cfirst = 0.f;
    load_g(0, ss, gfirst);

    // retire the weight loads here: left pending, the compiler re-waits for them inside the MFMA loop (vmcnt retires in
    // order, so those waits would also drain the next tile's h loads that are meant to fly under the matrix work)
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
    for (int step = 0; step < a.T; ++step) {
        const int t = rev ? a.T - 1 - step : step;
        const float* hprev = hx + (long)(step & 1) * a.S * H;
        float* hnext = hx + (long)((step + 1) & 1) * a.S * H;
        // h_{t-1} of a tile's 16 sequences, [16][H] fp32, comes through agent-coherent (sc1) 8 B loads - they bypass this
        // XCD's possibly stale L2 lines, so the exchange needs no cache invalidate (measured: ordinary cached loads from
        // a one-slab-per-step tensor, which cannot go stale, are no faster - the exchange is not bandwidth bound).  The
        // loads of the block's NEXT tile (and its gate pre-activations / cell state) are issued before the MFMAs of the
        // current one and land in the other half of the double-buffered LDS tile: staging overlaps the matrix work and a
        // tile costs one barrier.  Every step runs the same code (h_{-1} = 0 and c_{-1} = 0 are zero-filled buffers, not
        // branches): a conditional VMEM issue makes the compiler's wait counts collapse to 0
        constexpr int NLD = 16 * (H / 2) / 256;
        unsigned long long v[NLD];
        auto issue_h = [&](int nt_) {
            static_for_c<NLD>([&](auto I_) {
                constexpr int i = decltype(I_)::value;
                // row is uniform per i (H / 512 loads of 256 x 8 B per row): scalar row base + one per-lane offset
                constexpr int row = i / (H / 512), c2i = 256 * (i % (H / 512));
                const int nn = min(nt_ * 16 + row, a.S - 1);
                const unsigned long long* src =
                    reinterpret_cast<const unsigned long long*>(hprev + (long)nn * H + 2 * c2i) + tid;
                v[i] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            });
        };
        issue_h(ss);
        float g[4], cprev = cfirst;
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = gfirst[q];
        // a tile's results go out one tile late (after the next tile's LDS publish): stores issued behind the prefetch
        // would be the youngest VMEM ops when the next tile waits for its h loads, and vmcnt retires in order
        float h_p = 0.f, c_p = 0.f;
        int n_p = 0;
        auto flush = [&](bool first) {
            if (n_p < a.S) {
                if (!first) cell[n_p] = c_p;
                out[(long)t * a.out_t + n_p] = h_p;
                __hip_atomic_store(hnext + (long)n_p * H + u, h_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
        int kbuf = 0;
        for (int nt = ss; nt < NT; nt += a.SS, kbuf ^= 1) {
            const int n = nt * 16 + l15;
            floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
            float* hsb = hs + kbuf * (16 * LDW);
            static_for_c<NLD>([&](auto I_) {
                constexpr int i = decltype(I_)::value;
                constexpr int row = i / (H / 512), c2i = 256 * (i % (H / 512));
                reinterpret_cast<unsigned long long*>(hsb + row * LDW + 2 * c2i)[tid] = v[i];
            });
            // publishes hsb; the other half was last read two tiles ago, i.e. before the previous tile's barrier
            __syncthreads();
            if (nt != ss) flush(nt - a.SS == ss);
            float gn[4] = {0.f, 0.f, 0.f, 0.f}, cnext = 0.f;
            if (nt + a.SS < NT) {
                load_g(step, nt + a.SS, gn);
                cnext = cell[min(n + 16 * a.SS, a.S - 1)];
                issue_h(nt + a.SS);
            }
            {
                // B operand reads run two 16 B groups ahead of the MFMAs that consume them (ring of 3).  The compiler's
                // own schedule parks an lgkmcnt(0) behind every ds_read_b128 (H/16 exposed LDS round trips per step) and
                // at ~500 registers it will not hoist them: reads and counted waits are asm, the wait names the
                // register it releases so that no consumer moves above it
                const unsigned haddr = (unsigned)(size_t)(hsb + l15 * LDW + l4 * KQ);
                floatx4 bq[3];
#define CO_READ(Q, J) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(Q) : "v"(haddr), "n"((J) * 16) : "memory")
#define CO_WAIT(Q, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(Q))
                CO_READ(bq[0], 0);
                CO_READ(bq[1], 1);
                static_for_c<KQ / 4>([&](auto J_) {
                    constexpr int j = decltype(J_)::value;
                    if constexpr (j + 2 < KQ / 4) {
                        CO_READ(bq[(j + 2) % 3], j + 2);
                        CO_WAIT(bq[j % 3], 2);
                    } else if constexpr (j + 1 < KQ / 4) {
                        CO_WAIT(bq[j % 3], 1);
                    } else {
                        CO_WAIT(bq[j % 3], 0);
                    }
                    const floatx4 b = bq[j % 3];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][0], b[0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][1], b[1], acc1, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][2], b[2], acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][3], b[3], acc3, 0, 0, 0);
                });
#undef CO_READ
#undef CO_WAIT
            }
            const floatx4 acc = (acc0 + acc1) + (acc2 + acc3);
            const float cn = sigm(acc[1] + g[1]) * cprev + sigm(acc[0] + g[0]) * tanhf_fast(acc[2] + g[2]);
            const float h = sigm(acc[3] + g[3]) * tanhf_fast(cn);
            if (nt == ss) cfirst = cn;
            h_p = h; c_p = cn; n_p = n;
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] = gn[q];
            cprev = cnext;
        }
        flush(NT - ss <= a.SS);                 // the step's last tile (it is also the first when the block owns one)
        if (step + 1 < a.T) {
            // unit slices of this (z, sequence slice) exchange h_t: release our writes, wait for the others'
            // h_t went out as agent-coherent (sc1) write-through stores: each wave waits for the acknowledgement of its
            // own stores (vmcnt(0) - a workgroup-scope barrier alone does not wait for global stores), the workgroup
            // barrier collects the four waves, and a relaxed arrival is then enough: no L2 write-back / invalidate per step
            if (!(a.dbg & 8)) __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();                                       // (also: every wave is done reading hs)
            if (tid == 0 && !(a.dbg & 4))
                __hip_atomic_store(flags + us, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            load_g(step + 1, ss, gfirst);
            if (wave == 0 && !(a.dbg & 4)) {
                const unsigned want = (unsigned)(step + 1);
                const unsigned long long t0 = wall_clock64();
                // relaxed polling: an acquire load would invalidate the caches on every iteration
                for (;;) {
                    const unsigned v = lane < US ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
                    if (__builtin_amdgcn_ballot_w64(v < want) == 0) break;
                    if (wall_clock64() - t0 > 400000000ull) __builtin_trap();     // 4 s @ 100 MHz: never hang the GPU
                }
            }
            __syncthreads();
        }
    }
}

// ---- K-split form for at most 16 sequences (one MFMA tile) ----------------------------------------------------------
// At batch 1 ... 16 the kernel above uses 64 of the 256 CUs (one sequence slice) and a step is 3.9 us of MFMAs - a whole
// 16-column tile for one useful column - plus the exchange.  Here EVERY CU takes part: a workgroup owns 4 hidden units (16
// gate rows) and its four waves split K - a wave keeps rows x K / 4 in H / 16 VGPRs, stages only its quarter of h_{t-1}
// (wave-private LDS, no workgroup barrier), runs H / 16 MFMAs (1 us at H = 1024) and leaves a partial tile in LDS; wave 0
// adds the four partials in a fixed order, updates the cell (kept in a register) and publishes h_t.  Every CU then reads
// all of h - 4 KB per sequence and step, nothing at these batch sizes (at batch 64 it would be 128 MB per step through the
// fabric: the sequence-sliced form above stays for more than one tile).  Same flag protocol, H / 4 producers.
// TAG: the exchange carries its own arrival signal - h_t goes out with its lowest mantissa bit replaced by a step tag (h_t lives
// in slab (t + 1) & 1 and carries bit ((t + 1) >> 1) & 1; what the slab held before, h_{t-2}, carries the other value), and a
// wave simply re-loads its quarter of h_{t-1} until every element shows the expected bit: no store acknowledgement, flag store
// and flag poll in front of the load.  The recurrence then runs on h rounded to 23 mantissa bits (<= 1 ulp per step, identical
// from run to run); the output tensor keeps the exact h.  A producer cannot overwrite a value a slower consumer still needs: it
// needs that consumer's h_t - published after the consumer has read h_{t-1} - before it can produce h_{t+1}.
template <int H, int NS, bool TAG>
__global__ __launch_bounds__(256, 1) void lstm_coop_ks_kernel(const LstmCoopArgs a) {
    constexpr int KW = H / 4;          // k values per wave
    constexpr int KQ = KW / 4;         // k values per MFMA k-slot (l4)
    constexpr int LDW = KW + 4;        // LDS row stride
    constexpr int NWG = H / 4;         // workgroups per LSTM
    constexpr int FPL = KW / 64;       // floats of a staged row per lane (2 or 4)
    constexpr int NLD = NS * FPL / 2;  // 8 B loads per lane and step
    extern __shared__ float hs[];      // [4][16][LDW] wave-private h tiles, then [4][64] floatx4 partial tiles
    floatx4* red = reinterpret_cast<floatx4*>(hs + 4 * 16 * LDW);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ug = blockIdx.x % NWG, z = blockIdx.x / NWG;
    const bool rev = (a.reverse >> z) & 1;
    const int r0 = ug * 16;
    floatx4 wa[KQ / 4];
    {
        const floatx4* __restrict__ W =
            reinterpret_cast<const floatx4*>(a.whh + (long)z * a.whh_z + (long)(r0 + l15) * H + wave * KW + l4 * KQ);
        static_for_c<KQ / 4>([&](auto J_) {
            constexpr int j = decltype(J_)::value;
            wa[j] = W[j];
        });
    }
    // wave 0: hidden unit / sequence of this lane's accumulator (NS == 1, the dot-product form: lanes 0, 4, 8, 12 own a unit each)
    const int u = NS == 1 ? ug * 4 + (l15 >> 2) : ug * 4 + l4, n = NS == 1 ? 0 : l15;
    const bool live = NS == 1 ? (lane < 16 && (lane & 3) == 0) : n < a.S;
    const int nc = live ? n : a.S - 1;
    const float* __restrict__ gx = a.gx + (long)z * a.gx_z + (long)(4 * u) * a.gx_row + nc;
    float* __restrict__ out = a.out + (long)z * a.out_z + (long)u * a.out_row + nc;
    float* __restrict__ hx = a.hx + (long)z * 2 * a.S * H;
    unsigned* flags = a.bar + z * 256;
    float* hsw = hs + wave * (16 * LDW);
    for (int i = lane; i < 16 * LDW; i += 64) hsw[i] = 0.f;      // rows >= NS stay zero
    float g[4] = {0.f, 0.f, 0.f, 0.f}, c = 0.f;
    auto load_g = [&](int step) {
        const int t = rev ? a.T - 1 - step : step;
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = gx[(long)t * a.gx_t + q * a.gx_row];
    };
    if (wave == 0) load_g(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the weight loads retire here (see the kernel above)
    __syncthreads();
    for (int step = 0; step < a.T; ++step) {
        const int t = rev ? a.T - 1 - step : step;
        const float* hprev = hx + (long)(step & 1) * a.S * H + wave * KW;
        float* hnext = hx + (long)((step + 1) & 1) * a.S * H;
        unsigned long long v[NLD];
        const unsigned tag_prev = (unsigned)(step >> 1) & 1u, tag_next = (unsigned)((step + 1) >> 1) & 1u;
        auto issue_h = [&]() {
            static_for_c<NLD>([&](auto I_) {
                constexpr int i = decltype(I_)::value;
                constexpr int row = i / (FPL / 2), part = i % (FPL / 2);
                const int nn = min(row, a.S - 1);
                const unsigned long long* src = reinterpret_cast<const unsigned long long*>(hprev + (long)nn * H + lane * FPL) + part;
                v[i] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            });
        };
        issue_h();
        if (TAG && !(a.dbg & 4)) {
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                unsigned bad = 0;
                static_for_c<NLD>([&](auto I_) {
                    constexpr int i = decltype(I_)::value;
                    bad |= ((unsigned)v[i] ^ tag_prev) | ((unsigned)(v[i] >> 32) ^ tag_prev);
                });
                if (__builtin_amdgcn_ballot_w64((bad & 1u) != 0) == 0) break;
                if (wall_clock64() - t0 > 400000000ull) __builtin_trap();     // 4 s @ 100 MHz: never hang the GPU
                issue_h();
            }
        }
        static_for_c<NLD>([&](auto I_) {
            constexpr int i = decltype(I_)::value;
            constexpr int row = i / (FPL / 2), part = i % (FPL / 2);
            reinterpret_cast<unsigned long long*>(hsw + row * LDW + lane * FPL)[part] = v[i];
        });
        floatx4 acc;
        if constexpr (NS == 1) {
            // ONE sequence: 15 of the tile's 16 columns would be padding (1 us of MFMAs per step) - a dot product per lane
            // instead: lane (row l15, slot l4) multiplies its H / 16 weights with its slice of h, the four slots fold by
            // shuffles, the four waves through LDS, and lanes 0 / 4 / 8 / 12 of wave 0 collect the gates of a unit each
            const floatx4* hb = reinterpret_cast<const floatx4*>(hsw + l4 * KQ);
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            static_for_c<KQ / 4>([&](auto J_) {
                constexpr int j = decltype(J_)::value;
                const floatx4 b = hb[j];
                s0 = fmaf(wa[j][0], b[0], s0);
                s1 = fmaf(wa[j][1], b[1], s1);
                s2 = fmaf(wa[j][2], b[2], s2);
                s3 = fmaf(wa[j][3], b[3], s3);
            });
            float pr = (s0 + s1) + (s2 + s3);
            pr += __shfl_xor(pr, 16, 64);
            pr += __shfl_xor(pr, 32, 64);
            float* redf = reinterpret_cast<float*>(red);
            if (lane < 16) redf[wave * 16 + lane] = pr;
            __syncthreads();
            const float tot = (redf[l15] + redf[16 + l15]) + (redf[32 + l15] + redf[48 + l15]);
            const int b0 = lane & 12;
            acc = floatx4{__shfl(tot, b0, 64), __shfl(tot, b0 + 1, 64), __shfl(tot, b0 + 2, 64), __shfl(tot, b0 + 3, 64)};
        } else {
            floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
            const floatx4* hb = reinterpret_cast<const floatx4*>(hsw + l15 * LDW + l4 * KQ);
            static_for_c<KQ / 4>([&](auto J_) {
                constexpr int j = decltype(J_)::value;
                const floatx4 b = hb[j];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][0], b[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][1], b[1], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][2], b[2], acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][3], b[3], acc3, 0, 0, 0);
            });
            red[wave * 64 + lane] = (acc0 + acc1) + (acc2 + acc3);
            __syncthreads();
            acc = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
        }
        if (wave == 0) {
            const float cn = sigm(acc[1] + g[1]) * c + sigm(acc[0] + g[0]) * tanhf_fast(acc[2] + g[2]);
            const float h = sigm(acc[3] + g[3]) * tanhf_fast(cn);
            c = cn;
            if (live) {
                out[(long)t * a.out_t] = h;
                const float ht = TAG ? __uint_as_float((__float_as_uint(h) & ~1u) | tag_next) : h;
                __hip_atomic_store(hnext + (long)n * H + u, ht, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (TAG) {
                if (step + 1 < a.T) load_g(step + 1);
            } else if (step + 1 < a.T) {
                if (!(a.dbg & 8)) __builtin_amdgcn_s_waitcnt(0x0F70);      // our h_t has been acknowledged
                if (lane == 0 && !(a.dbg & 4))
                    __hip_atomic_store(flags + ug, (unsigned)(step + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                load_g(step + 1);
                if (!(a.dbg & 4)) {
                    const unsigned want = (unsigned)(step + 1);
                    const unsigned long long t0 = wall_clock64();
                    for (;;) {
                        unsigned lo = want;
#pragma unroll
                        for (int q = 0; q < NWG / 64; ++q)
                            lo = min(lo, __hip_atomic_load(flags + lane + 64 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        if (__builtin_amdgcn_ballot_w64(lo < want) == 0) break;
                        if (wall_clock64() - t0 > 400000000ull) __builtin_trap();     // 4 s @ 100 MHz: never hang the GPU
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---- a STACK of LSTM layers on one sequence as one launch -----------------------------------------------------------------
// With one sequence a step of the K-split kernel is the exchange latency and little else (2.65 us, 0.15 us of it arithmetic),
// and a stack of L layers pays it L x T times.  Here the layers run as a wavefront: at global step s layer l works on frame
// t = s - l, so every layer's inputs - its own h_l[t-1] and, for l >= 1, the h_{l-1}[t] it projects itself - were published
// in global step s - 1, and ONE exchange latency serves all layers: T + L - 1 steps instead of L x T.  A workgroup owns the
// same 4 hidden units (16 gate rows) of every layer; a wave keeps its K quarter of W_hh of all layers and of W_ih of the
// layers above the first (the first layer's input projection stays one batched GEMM) - 5 x H / 16 = 320 VGPRs for three
// 1024-wide layers.  Dot products, the tagged exchange (one slab pair per layer) and the fold through LDS are those of the
// one-sequence K-split kernel; the cell states live in registers of wave 0.
template <int H, int L>
__global__ __launch_bounds__(256, 1) void lstm_stack_kernel(const LstmStackArgs a) {
    constexpr int KW = H / 4, KQ = KW / 4, NWG = H / 4, FPL = KW / 64, NLD1 = FPL / 2, NV = 2 * L - 1;
    extern __shared__ float hs[];      // [4 waves][NV vectors][KW] staged h quarters, [L][4][16] row partials, then W_ih
    float* redf = hs + 4 * NV * KW;
    // the input matrices of the layers above the first live in LDS ([layer][j][thread] 16 B fragments: every read is one
    // conflict-free ds_read_b128 per lane) - with them in registers too, three 1024-wide layers need 5 x 64 VGPRs for weights
    // alone and the kernel spills
    floatx4* wl = reinterpret_cast<floatx4*>(redf + L * 64);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ug = blockIdx.x;
    const int r0 = ug * 16;
    floatx4 whh[L][KQ / 4];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const floatx4* __restrict__ W = reinterpret_cast<const floatx4*>(a.whh[l] + (long)(r0 + l15) * H + wave * KW + l4 * KQ);
        static_for_c<KQ / 4>([&](auto J_) {
            constexpr int j = decltype(J_)::value;
            whh[l][j] = W[j];
        });
        if (l > 0) {
            const floatx4* __restrict__ V = reinterpret_cast<const floatx4*>(a.wih[l] + (long)(r0 + l15) * H + wave * KW + l4 * KQ);
            static_for_c<KQ / 4>([&](auto J_) {
                constexpr int j = decltype(J_)::value;
                wl[((l - 1) * (KQ / 4) + j) * 256 + tid] = V[j];
            });
        }
    }
    const int u = ug * 4 + (l15 >> 2);                           // wave 0, lanes 0 / 4 / 8 / 12: the unit this lane updates
    const bool live = wave == 0 && lane < 16 && (lane & 3) == 0;
    const float* __restrict__ gx = a.gx0 + (long)(4 * u) * a.gx_row;
    float bias[L > 1 ? L - 1 : 1][4];
#pragma unroll
    for (int l = 1; l < L; ++l)
#pragma unroll
        for (int q = 0; q < 4; ++q) bias[l - 1][q] = a.bias[l][4 * u + q];
    float c[L], g0[4];
#pragma unroll
    for (int l = 0; l < L; ++l) c[l] = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) g0[q] = gx[q * a.gx_row];
    float* hsw = hs + wave * (NV * KW);
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): the weight loads retire here
    __syncthreads();
    const int nsteps = a.T + L - 1;
    for (int s = 0; s < nsteps; ++s) {
        // vector 2 l: h_l[t - 1] (layer l's own state), vector 2 l - 1: h_{l-1}[t] (its input), t = s - l
        unsigned long long v[NV][NLD1];
        auto issue = [&]() {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int t = s - l;
                if (t < 0 || t >= a.T) continue;
                const float* own = a.hx + ((long)l * 2 + (t & 1)) * H + wave * KW + lane * FPL;
#pragma unroll
                for (int i = 0; i < NLD1; ++i)
                    v[2 * l][i] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(own) + i, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
                if (l > 0) {
                    const float* in = a.hx + ((long)(l - 1) * 2 + ((t + 1) & 1)) * H + wave * KW + lane * FPL;
#pragma unroll
                    for (int i = 0; i < NLD1; ++i)
                        v[2 * l - 1][i] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(in) + i, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        };
        issue();
        {
            const unsigned long long t0 = wall_clock64();
            for (;;) {
                unsigned bad = 0;
#pragma unroll
                for (int l = 0; l < L; ++l) {
                    const int t = s - l;
                    if (t < 0 || t >= a.T) continue;
                    const unsigned town = (unsigned)(t >> 1) & 1u, tin = (unsigned)((t + 1) >> 1) & 1u;
#pragma unroll
                    for (int i = 0; i < NLD1; ++i) {
                        bad |= ((unsigned)v[2 * l][i] ^ town) | ((unsigned)(v[2 * l][i] >> 32) ^ town);
                        if (l > 0) bad |= ((unsigned)v[2 * l - 1][i] ^ tin) | ((unsigned)(v[2 * l - 1][i] >> 32) ^ tin);
                    }
                }
                if (__builtin_amdgcn_ballot_w64((bad & 1u) != 0) == 0) break;
                if (wall_clock64() - t0 > 400000000ull) __builtin_trap();     // 4 s @ 100 MHz: never hang the GPU
                issue();
            }
        }
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int t = s - l;
            if (t < 0 || t >= a.T) continue;
#pragma unroll
            for (int i = 0; i < NLD1; ++i) {
                reinterpret_cast<unsigned long long*>(hsw + (2 * l) * KW + lane * FPL)[i] = v[2 * l][i];
                if (l > 0) reinterpret_cast<unsigned long long*>(hsw + (2 * l - 1) * KW + lane * FPL)[i] = v[2 * l - 1][i];
            }
        }
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const int t = s - l;
            if (t < 0 || t >= a.T) continue;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            {
                const floatx4* hb = reinterpret_cast<const floatx4*>(hsw + (2 * l) * KW + l4 * KQ);
                static_for_c<KQ / 4>([&](auto J_) {
                    constexpr int j = decltype(J_)::value;
                    const floatx4 b = hb[j];
                    s0 = fmaf(whh[l][j][0], b[0], s0);
                    s1 = fmaf(whh[l][j][1], b[1], s1);
                    s2 = fmaf(whh[l][j][2], b[2], s2);
                    s3 = fmaf(whh[l][j][3], b[3], s3);
                });
            }
            if (l > 0) {
                const floatx4* xb = reinterpret_cast<const floatx4*>(hsw + (2 * l - 1) * KW + l4 * KQ);
                static_for_c<KQ / 4>([&](auto J_) {
                    constexpr int j = decltype(J_)::value;
                    const floatx4 b = xb[j];
                    const floatx4 w = wl[((l > 0 ? l - 1 : 0) * (KQ / 4) + j) * 256 + tid];
                    s0 = fmaf(w[0], b[0], s0);
                    s1 = fmaf(w[1], b[1], s1);
                    s2 = fmaf(w[2], b[2], s2);
                    s3 = fmaf(w[3], b[3], s3);
                });
            }
            float pr = (s0 + s1) + (s2 + s3);
            pr += __shfl_xor(pr, 16, 64);
            pr += __shfl_xor(pr, 32, 64);
            if (lane < 16) redf[(l * 4 + wave) * 16 + lane] = pr;
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const int t = s - l;
                if (t < 0 || t >= a.T) continue;
                const float* rp = redf + l * 64;
                const float tot = (rp[l15] + rp[16 + l15]) + (rp[32 + l15] + rp[48 + l15]);
                const int b0 = lane & 12;
                float gi = __shfl(tot, b0, 64), gf = __shfl(tot, b0 + 1, 64), gg = __shfl(tot, b0 + 2, 64), go = __shfl(tot, b0 + 3, 64);
                if (l == 0) { gi += g0[0]; gf += g0[1]; gg += g0[2]; go += g0[3]; }
                else { gi += bias[l > 0 ? l - 1 : 0][0]; gf += bias[l > 0 ? l - 1 : 0][1]; gg += bias[l > 0 ? l - 1 : 0][2]; go += bias[l > 0 ? l - 1 : 0][3]; }
                const float cn = sigm(gf) * c[l] + sigm(gi) * tanhf_fast(gg);
                const float h = sigm(go) * tanhf_fast(cn);
                c[l] = cn;
                if (live) {
                    if (l == L - 1) a.out[(long)t * a.out_t + (long)u * a.out_row] = h;
                    const unsigned tag = (unsigned)((t + 1) >> 1) & 1u;
                    __hip_atomic_store(a.hx + ((long)l * 2 + ((t + 1) & 1)) * H + u, __uint_as_float((__float_as_uint(h) & ~1u) | tag),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (s + 1 < a.T) {
#pragma unroll
                for (int q = 0; q < 4; ++q) g0[q] = gx[(long)(s + 1) * a.gx_t + q * a.gx_row];
            }
        }
        __syncthreads();
    }
}

// ---- sub-tile pipelined form (round 4): 4-sequence sub-tiles on v_mfma_f32_4x4x1_16b_f32, eight independent waves ---------------
// lstm_coop_kernel above gives a workgroup ONE 16-sequence tile per step at batch 64 (H = 1024): 3.4 us of matrix work, then the
// serial chain store -> acknowledge -> flag -> poll -> 64 KB of h -> LDS -> barrier with nothing to overlap it - 8.2 us per step
// (CRN at batch 64: 46 % of the decode); with more tiles per workgroup it still pays 6.2 - 7 us per tile.
// Here the unit of work is a SUB-TILE of 4 sequences, and a workgroup walks its sub-tiles round robin (4 per step at batch 64,
// 16 at batch 256): while sub-tile j's h_t travels, sub-tiles j + 1 ... run their matrix work.
//   * v_mfma_f32_4x4x1_16b_f32 = 16 independent 4 x 4 x 1 blocks.  Block <-> K slice s16 (lane >> 2), block row <-> gate, block
//     column <-> sequence: lane (s16, g) holds W[4 u + g][(16 i + s16) 4 + kk] of the wave's TWO units u (H / 8 VGPRs), lane
//     (s16, n) feeds h[n][(16 i + s16) 4 + kk] - one 16 B LDS read per 8 matrix instructions, rows in LDS are plain
//     [sequence][H + 16] (bank = 16 n + 4 s16 + kk: conflict free).  Four accumulator chains per wave (unit x k parity): a
//     dependent 4x4x1 issues 16 cycles behind its producer, an independent one 8 (tools/mfma4bench.cpp).
//   * EIGHT waves per workgroup, two per SIMD (<= 256 registers each): a wave owns its 2 units for ALL of K, so there is no
//     cross-wave reduction - the 16 K slices fold inside the wave (v_permlane32_swap: unit 0 | unit 1 into the wave halves,
//     v_permlane16_swap + two row_ror adds inside a half) - and while one wave of a SIMD folds, updates its cells and stores,
//     the other one keeps the matrix pipe busy (with one wave per SIMD that tail was as long as the matrix work itself).
//   * The exchange carries its own arrival signal (mantissa-LSB step tag, as in lstm_coop_ks_kernel).  Nothing the compiler
//     tracks is pending across slots (its wait-count pass parks `s_waitcnt vmcnt(0)` at the loop header otherwise): h rows AND
//     the slot's gate pre-activations travel global -> LDS by DMA (`global_load_lds_dwordx4 ... sc1`, issued from asm) LEAD slots
//     ahead into a ring of LEAD + 1 buffers, the cell state lives in LDS for the whole launch, the only compiler-issued vector
//     memory operations of a slot are its two stores, and the wait in front of a slot's tag check is a counted
//     `s_waitcnt vmcnt(N)` that names exactly what was issued after that slot's DMA group.  A wave checks the tags of the words
//     it fetched itself and only then re-fetches (rare path, bounded spin); h_t goes out as one 8 B sc1 store per sequence and wave.
//     One workgroup barrier per sub-tile (LDS publish); no flags, no store acknowledgement.
// A producer cannot overwrite a value a slower consumer still needs: it writes h_{t+1}[j] over h_{t-1}[j] only after it has read
// h_t[j] from everyone, which everyone published after reading h_{t-1}[j].  The recurrence runs on h rounded to 23 mantissa
// bits (bit-identical from run to run); `out` keeps the exact h.
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float swap_add32(float x, float y) {      // lanes 0-31: x.lo + x.hi, lanes 32-63: y.lo + y.hi
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap_add16(float x, float y) {      // 16-lane rows: [x.r0 + x.r1, y.r0 + y.r1, x.r2 + x.r3, y.r2 + y.r3]
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float row_ror_add(float x) {              // x + x rotated by CTRL - 0x120 lanes inside its 16-lane row
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ void c4_dma16(const float* g, unsigned lds_byte) {      // lane l -> LDS[lds_byte + 16 l], agent-coherent
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1" ::"s"(lds_byte), "v"(g) : "memory");
}
__device__ __forceinline__ void c4_dma4(const float* g, unsigned lds_byte) {       // lane l -> LDS[lds_byte + 4 l]
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(lds_byte), "v"(g) : "memory");
}
__device__ __forceinline__ void c4_dma16g(const float* base, unsigned voff, unsigned lds_byte) {     // the same, ordinary cache policy
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte), "v"(voff), "s"(base) : "memory");
}
template <int N>
__device__ __forceinline__ void c4_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one hardware exp2 and one hardware reciprocal per activation (v_exp_f32 / v_rcp_f32, 1 ulp each): `__frcp_rn` is a correctly
// rounded division - ten instructions - and a slot's tail is what the matrix pipe waits for
__device__ __forceinline__ float sigm_hw(float x) { return kExactMath ? fm_sigmoid(x) : fm_rcp(1.f + fm_exp2(-1.44269504088896f * x)); }
__device__ __forceinline__ float tanh_hw(float x) { return kExactMath ? fm_tanh(x) : 1.f - 2.f * fm_rcp(1.f + fm_exp2(2.88539008177793f * x)); }
// 16 B per lane global -> LDS with a wave-uniform 64-bit base and a 32-bit per-lane byte offset (no 64-bit vector address arithmetic per slot)
__device__ __forceinline__ void c4_dma16s(const float* base, unsigned voff, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1" ::"s"(lds_byte), "v"(voff), "s"(base) : "memory");
}
__device__ __forceinline__ void c4_dma4s(const float* base, unsigned voff, unsigned lds_byte) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds_byte), "v"(voff), "s"(base) : "memory");
}

template <int H, int LEAD, int NW>
__global__ __launch_bounds__(64 * NW) void lstm_coop8_kernel(const LstmCoopArgs a) {
    constexpr int UW = 16 / NW;        // hidden units per wave (NW = 8 waves: 2, two waves per SIMD; NW = 4: 4, one wave per SIMD)
    constexpr int KS = H / 16;         // k values of one K slice (lane group s16)
    constexpr int NI = KS / 4;         // 16 B operand reads per sub-tile and lane
    constexpr int NSTR = H + 16;       // LDS row stride (floats): rows 16 banks apart
    constexpr int US = H / 16;         // unit slices (workgroups) per LSTM
    constexpr int ND = H / (64 * NW);  // 16 B DMA instructions per wave and fetch: a wave stages 4 / NW of a sequence row
    constexpr int FW = H * 4 / NW;     // floats of a sub-tile one wave fetches
    constexpr int NR = LEAD + 1;       // ring buffers
    constexpr int GRP = ND + 1;        // DMA instructions of one slot's group (h part + gate pre-activations)
    // registers: UW x H / 16 x 4 weights per lane = 256 at NW = 4, H = 1024 - with the operand ring and the tail's temporaries more
    // than the 256 architectural VGPRs, and the compiler then parks one unit's weights in AGPRs and copies them back four at a
    // time inside the matrix loop (64 extra vector instructions per slot).  That unit's weights live in LDS instead (64 KB per
    // workgroup) and come in with the operand reads, 16 B per 4 matrix instructions.
    constexpr int UL = (NW == 4 && H == 1024) ? 1 : 0;       // units of a wave whose weights are LDS-resident
    constexpr int UG = UW - UL;                              // ... register-resident
    extern __shared__ __attribute__((aligned(16))) float hs4[];      // [NR][4][NSTR] h ring, [NR][NW][64] gx ring, [NW][NI][64][4] weights, [NW][NSUB][4 UW] cells
    float* gxs = hs4 + NR * 4 * NSTR;
    float* wl = gxs + NR * NW * 64;
    float* cs = wl + UL * NW * NI * 256;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s16 = lane >> 2, n = lane & 3;
    const int ul = NW == 8 ? lane >> 5 : lane >> 4;          // the unit (of this wave's UW) this lane holds after the fold
    const int us = blockIdx.x % US, ss = (blockIdx.x / US) % a.SS, z = blockIdx.x / (US * a.SS);
    const bool rev = (a.reverse >> z) & 1;
    const int NS4 = (a.S + 3) >> 2;                          // sub-tiles of the launch
    const int NSUB = (NS4 - ss + a.SS - 1) / a.SS;           // ... of this sequence slice: ss, ss + SS, ...
    const int NSUBmax = (NS4 + a.SS - 1) / a.SS;             // (the launch's LDS is sized for it)
    const int U0 = us * 16 + wave * UW;                      // first hidden unit of this wave

    floatx4 wa[UG][NI];
    static_for_c<UG>([&](auto U_) {
        constexpr int u = decltype(U_)::value;
        const floatx4* __restrict__ W =
            reinterpret_cast<const floatx4*>(a.whh + (long)z * a.whh_z + (long)(4 * (U0 + u) + n) * H + s16 * 4);
        static_for_c<NI>([&](auto I_) {
            constexpr int i = decltype(I_)::value;
            wa[u][i] = W[i * 16];
        });
    });
    if constexpr (UL > 0) {
        const floatx4* __restrict__ W =
            reinterpret_cast<const floatx4*>(a.whh + (long)z * a.whh_z + (long)(4 * (U0 + UG) + n) * H + s16 * 4);
        for (int i = 0; i < NI; ++i) reinterpret_cast<floatx4*>(wl)[(wave * NI + i) * 64 + lane] = W[i * 16];
    }
    float* cw = cs + wave * (NSUBmax * 4 * UW);
    for (int i = lane; i < NSUBmax * 4 * UW; i += 64) cw[i] = 0.f;  // c_{-1} = 0
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): the weight loads retire here, before the counted waits below

    const float* __restrict__ hx = a.hx + (long)z * 2 * a.S * H;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hx), 0, 2 * a.S * H * 4, 0x00020000);
    const int Ur = U0 + ul;                                  // the unit this lane updates after the fold
    // per-lane parts of the addresses as 32-bit offsets, the per-slot parts are wave-uniform (scalar unit): on this chip a vector
    // instruction does not hide under another wave's f32 matrix instructions (tools/coissuebench.cpp: the times add), so every
    // vector instruction of a slot's tail is paid in matrix time
    float* __restrict__ outz = a.out + (long)z * a.out_z;
    const unsigned ov = (unsigned)((long)Ur * a.out_row + n);        // element offset of (unit, sequence n) inside a frame of `out`
    // gate pre-activations of a slot as ONE 4 B DMA per wave: lane (u, n, k) fetches gate k of unit U0 + u for sequence n (the
    // four gates of a (unit, sequence) pair are then one 16 B LDS read); NW = 8: lanes >= 32 repeat the first 32
    const int gl = lane & (16 * UW - 1);
    const float* __restrict__ gxz = a.gx + (long)z * a.gx_z;
    const unsigned gv = (unsigned)(((long)(4 * (U0 + (gl >> 4)) + (gl & 3)) * a.gx_row + ((gl >> 2) & 3)) * 4);   // byte offset
    const bool writer = NW == 8 ? (lane & 28) == 0 : (lane & 12) == 0;      // one lane per (unit, sequence) of the wave
    const unsigned lds0 = (unsigned)(size_t)hs4;             // LDS byte address of the ring (low 32 bits of the flat pointer)
    const unsigned gxs0 = (unsigned)(size_t)gxs;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int frow_u = wave_u * 4 / NW, fcol_u = (wave_u * 4 % NW) * (H / 4);   // the part of a sub-tile this wave fetches
    const int fpart = frow_u * NSTR + fcol_u;

    auto issue_h = [&](int t, int g, int buf) {
        const int nq = min(4 * g + frow_u, a.S - 1);         // (clamped inside this sub-tile: 4 g < S)
        const float* src = hx + ((long)(t & 1) * a.S + nq) * H + fcol_u;      // wave-uniform
        const unsigned dst = lds0 + (unsigned)(buf * 4 * NSTR + fpart) * 4u;
        if (a.dbg & 32) return;
        static_for_c<ND>([&](auto I_) {
            constexpr int i = decltype(I_)::value;
            c4_dma16s(src, lane * 16 + i * 1024, dst + i * 1024);
        });
    };
    auto issue = [&](int t, int g, int buf) {                // the DMA group of slot (t, sub-tile g) into ring buffer `buf`
        issue_h(t, g, buf);
        const int tt = min(t, a.T - 1), tr = rev ? a.T - 1 - tt : tt;
        // (the last sub-tile of a launch whose sequence count is no multiple of 4 fetches past its last column: values of the next
        // row or of the tensor's tail slack, never used - the tail stores nothing for n >= S)
        c4_dma4s(gxz + (long)tr * a.gx_t + 4 * g, gv, gxs0 + (unsigned)((buf * NW + wave_u) * 64) * 4u);
    };
    auto stale = [&](const float* part, unsigned tag) -> bool {      // this lane's words of the part its wave fetched
        unsigned bad = 0;
        static_for_c<ND>([&](auto I_) {
            constexpr int i = decltype(I_)::value;
            const uintx4 w = *reinterpret_cast<const uintx4*>(part + (i * 64 + lane) * 4);
            bad |= (w[0] ^ tag) | (w[1] ^ tag) | (w[2] ^ tag) | (w[3] ^ tag);
        });
        return __builtin_amdgcn_ballot_w64((bad & 1u) != 0) != 0;
    };

    const int Q = a.T * NSUB;                                // slots of this workgroup: (step, sub-tile) pairs
    int tf = 0, jf = 0, bf = 0;                              // the next slot to fetch, its ring buffer
    __syncthreads();                                         // (cell states zeroed)
#pragma unroll
    for (int p = 0; p < LEAD; ++p) {
        issue(tf, ss + jf * a.SS, bf);
        if (++jf == NSUB) { jf = 0; ++tf; }
        if (++bf == NR) bf = 0;
    }
    // fold the 16 K slices of slot (tt, jj), update the cells, publish h.  (Tried: waves 4 - 7 of the 8-wave form running this one
    // slot late, in front of their next matrix work, so that the two waves of a SIMD are in opposite phases - no gain, the tail
    // does not hide under the other wave's matrix instructions either way.)
    auto tail = [&](const floatx4 (&acc)[UW][2], const float (&gg)[4], int tt, int jj) {
        const int n0 = 4 * (ss + jj * a.SS);
        const float cprev = cw[jj * 4 * UW + 4 * ul + n];
        float gate[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float r;
            if constexpr (NW == 8) {       // wave halves (unit 0 | unit 1), then the two rows of a half
                const float p = swap_add32(acc[0][0][k] + acc[0][1][k], acc[1][0][k] + acc[1][1][k]);
                r = swap_add16(p, p);
            } else {                       // wave halves (units 0 | 2 and 1 | 3), then rows: unit r in row r
                const float p02 = swap_add32(acc[0][0][k] + acc[0][1][k], acc[2 % UW][0][k] + acc[2 % UW][1][k]);
                const float p13 = swap_add32(acc[1][0][k] + acc[1][1][k], acc[3 % UW][0][k] + acc[3 % UW][1][k]);
                r = swap_add16(p02, p13);
            }
            r = row_ror_add<0x124>(r);     // ... and the 4 slices of a row
            r = row_ror_add<0x128>(r);
            gate[k] = r + gg[k];
        }
        const float cn = sigm_hw(gate[1]) * cprev + sigm_hw(gate[0]) * tanh_hw(gate[2]);
        const float h = sigm_hw(gate[3]) * tanh_hw(cn);
        const unsigned hb_ = (__float_as_uint(h) & ~1u) | ((unsigned)((tt + 1) >> 1) & 1u);
        if (writer) cw[jj * 4 * UW + 4 * ul + n] = cn;
        if (!(a.dbg & 64)) {
            // both stores are issued by every wave (lane 0 always qualifies: 4 g < S): the counted wait relies on it
            const int tr = rev ? a.T - 1 - tt : tt;
            float* __restrict__ of = outz + (long)tr * a.out_t + n0;         // wave-uniform
            if (writer && n0 + n < a.S) of[ov] = h;
            const unsigned ho = (unsigned)((((tt + 1) & 1) * a.S + n0 + n) * (H * 4) + U0 * 4);
            // h of the wave's units gathered into lanes 0 - 3 (sequence n): one 8 B / 16 B store per sequence, not a 4 B fabric write each
            if constexpr (NW == 8) {
                const auto e = __builtin_amdgcn_permlane32_swap(hb_, hb_, false, false);       // [u0 | u0], [u1 | u1]
                if (lane < 4 && n0 + n < a.S) __builtin_amdgcn_raw_buffer_store_b64(uintx2{e[0], e[1]}, rs, ho, 0, 16);
            } else {
                const auto r1 = __builtin_amdgcn_permlane16_swap(hb_, hb_, false, false);      // rows [r0, r0, r2, r2], [r1, r1, r3, r3]
                const auto e = __builtin_amdgcn_permlane32_swap(r1[0], r1[0], false, false);   // [r0 x 4], [r2 x 4]
                const auto o = __builtin_amdgcn_permlane32_swap(r1[1], r1[1], false, false);   // [r1 x 4], [r3 x 4]
                if (lane < 4 && n0 + n < a.S) __builtin_amdgcn_raw_buffer_store_b128(uintx4{e[0], o[0], e[1], o[1]}, rs, ho, 0, 16);
            }
        }
    };
    int tq = 0, jq = 0, bq_ = 0;
    for (int q = 0; q < Q; ++q) {
        const int g = ss + jq * a.SS;
        const unsigned tag = (unsigned)(tq >> 1) & 1u;
        // younger than this slot's group: LEAD - 1 groups and two stores per slot in between (the first slots take the count
        // without stores: it only waits longer)
        if (q >= LEAD) c4_wait_vm<(LEAD - 1) * GRP + 2 * LEAD>();
        else c4_wait_vm<(LEAD - 1) * GRP>();
        float* hb = hs4 + bq_ * (4 * NSTR);
        if (!(a.dbg & (4 | 32)) && stale(hb + fpart, tag)) {
            const unsigned long long t0 = wall_clock64();
            do {
                if (wall_clock64() - t0 > 400000000ull) __builtin_trap();     // 4 s @ 100 MHz: never hang the GPU
                issue_h(tq, g, bq_);
                c4_wait_vm<0>();
            } while (stale(hb + fpart, tag));
        }
        float gcur[4];
        {
            const floatx4 g4 = *reinterpret_cast<const floatx4*>(gxs + (bq_ * NW + wave) * 64 + 16 * ul + 4 * n);
#pragma unroll
            for (int k = 0; k < 4; ++k) gcur[k] = g4[k];
        }
        if (!(a.dbg & 128)) __syncthreads();                 // publishes hb; every wave is done with the buffer of slot q - 1
        issue(tf, ss + jf * a.SS, bf);                       // slot q + LEAD -> the buffer slot q - 1 used
        if (++jf == NSUB) { jf = 0; ++tf; }
        if (++bf == NR) bf = 0;
        floatx4 acc[UW][2];
#pragma unroll
        for (int u = 0; u < UW; ++u) acc[u][0] = acc[u][1] = floatx4{0.f, 0.f, 0.f, 0.f};
        {
            const unsigned haddr = (unsigned)(size_t)(hb + n * NSTR + s16 * 4);
            const unsigned waddr = (unsigned)(size_t)(wl + (wave * NI * 64 + lane) * 4);
            floatx4 bq[3], wq[3];
            // operand reads run two 16 B groups ahead of the matrix instructions (ring of 3); the reads and their counted waits are
            // asm: the compiler's own schedule parks an lgkmcnt(0) behind every read (the wait names the register it releases so
            // that no consumer moves above it)
#define C4_READ(Qr, AD, J, ST) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(Qr) : "v"(AD), "n"((J) * (ST)) : "memory")
#define C4_WAIT(Qr, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(Qr))
#define C4_WAIT2(Qr, Wr, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(Qr), "+v"(Wr))
            {
            C4_READ(bq[0], haddr, 0, 256);
            if constexpr (UL > 0) C4_READ(wq[0], waddr, 0, 1024);
            C4_READ(bq[1], haddr, 1, 256);
            if constexpr (UL > 0) C4_READ(wq[1], waddr, 1, 1024);
            static_for_c<NI>([&](auto I_) {
                constexpr int i = decltype(I_)::value;
                if constexpr (i + 2 < NI) {
                    C4_READ(bq[(i + 2) % 3], haddr, i + 2, 256);
                    if constexpr (UL > 0) {
                        C4_READ(wq[(i + 2) % 3], waddr, i + 2, 1024);
                        C4_WAIT2(bq[i % 3], wq[i % 3], 4);
                    } else {
                        C4_WAIT(bq[i % 3], 2);
                    }
                } else if constexpr (i + 1 < NI) {
                    if constexpr (UL > 0) C4_WAIT2(bq[i % 3], wq[i % 3], 2);
                    else C4_WAIT(bq[i % 3], 1);
                } else {
                    if constexpr (UL > 0) C4_WAIT2(bq[i % 3], wq[i % 3], 0);
                    else C4_WAIT(bq[i % 3], 0);
                }
                const floatx4 b = bq[i % 3];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int u = 0; u < UG; ++u)
                        acc[u][kk & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[u][i][kk], b[kk], acc[u][kk & 1], 0, 0, 0);
                    if constexpr (UL > 0)
                        acc[UG][kk & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wq[i % 3][kk], b[kk], acc[UG][kk & 1], 0, 0, 0);
                    if (UW == 4) __builtin_amdgcn_sched_barrier(0);
                }
                if (UW != 4) __builtin_amdgcn_sched_barrier(0);
            });
            }
#undef C4_READ
#undef C4_WAIT
#undef C4_WAIT2
        }
        tail(acc, gcur, tq, jq);
        if (++jq == NSUB) { jq = 0; ++tq; }
        if (++bq_ == NR) bq_ = 0;
    }
}

// ---- 16-sequence tiles with the sub-tile form's exchange (round 4) ---------------------------------------------------------------
// What the sub-tile form taught: (1) on this chip a vector instruction never hides under f32 matrix work - v_mfma_f32_* runs at
// the vector rate on the same pipes, and tools/coissuebench.cpp shows the times of a matrix wave and a vector wave on one SIMD
// simply add - so a tile's non-matrix instructions are paid in full, and a 4-sequence sub-tile pays the fold / cell / publish
// tail four times per 16 sequences (1.45 us of work per sub-tile for 1.0 us of matrix instructions); (2) the tagged exchange
// fetched by LDS-DMA needs no flags, no acknowledgement and - per 64 KB tile - 20 DMA instructions per wave instead of 32 loads +
// 32 LDS stores per THREAD.  So: the 16 x 16 x 4 tile of lstm_coop_kernel (no K fold: a lane's accumulator holds the four gates
// of one (unit, sequence) pair) with that exchange.  A wave fetches 4 rows of the next tile (16 x 1 KB DMA + 4 x 256 B of gate
// pre-activations) right behind the barrier that publishes the current one, checks the tags of its own rows when the tile comes
// up, re-fetches only then; cell state in LDS; h_t leaves as 16 B sc1 stores (the four units of a wave, gathered by three lane
// swaps).  Per tile: 256 matrix instructions (8 250 cycles) + ~230 vector instructions.
template <int H>
__global__ __launch_bounds__(256, 1) void lstm_coop16_kernel(const LstmCoopArgs a) {
    constexpr int KQ = H / 4;          // k values per MFMA k-slot (lane >> 4)
    constexpr int LDW = H + 4;         // LDS row stride: 16 lanes x 16 B land in 64 distinct banks
    constexpr int US = H / 16;
    constexpr int ND = H / 256;        // 1 KB DMA instructions per row
    constexpr int GRP = 4 * ND + 1;    // DMA instructions of one slot's group: 4 rows + the gate pre-activations
    extern __shared__ __attribute__((aligned(16))) float hs16[];     // [2][16][LDW] h tiles, [2][4][256] gx, [4][NTL][64] cells
    float* gxs = hs16 + 2 * 16 * LDW;
    float* cs = gxs + 2 * 4 * 256;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int us = blockIdx.x % US, ss = (blockIdx.x / US) % a.SS, z = blockIdx.x / (US * a.SS);
    const bool rev = !a.pz && ((a.reverse >> z) & 1);
    const int NT = (a.S + 15) >> 4;
    const int NTL = (NT - ss + a.SS - 1) / a.SS;             // tiles of this workgroup: ss, ss + SS, ...
    const int NTLmax = (NT + a.SS - 1) / a.SS;
    const int r0 = us * 64 + wave * 16, U0 = us * 16 + wave * 4;
    // chunked layer pipeline (a.pz): this LSTM is layer lzz of a stack, on steps tb .. tb + Tn of its recurrence
    const int lzz = a.pz ? a.lz[z] : z, tb = a.pz ? a.t0[z] : 0, Tn = a.pz ? a.Tz[z] : a.T;

    floatx4 wa[KQ / 4];
    {
        const floatx4* __restrict__ W =
            reinterpret_cast<const floatx4*>((a.pz ? a.whhp[z] : a.whh + (long)z * a.whh_z) + (long)(r0 + l15) * H + l4 * KQ);
        static_for_c<KQ / 4>([&](auto J_) {
            constexpr int j = decltype(J_)::value;
            wa[j] = W[j];
        });
    }
    float* cw = cs + wave * (NTLmax * 64);
    float* __restrict__ cellg = a.cell + ((long)lzz * H + U0 + l4) * a.S + l15;      // this lane's (unit, sequence l15 of tile 0) in `cell`
    for (int j = 0; j < NTL; ++j) {                          // c_{-1} = 0, or the cells the previous range of steps left
        const int n = (ss + j * a.SS) * 16 + l15;
        cw[j * 64 + lane] = (a.pz && tb > 0 && n < a.S) ? cellg[(ss + j * a.SS) * 16] : 0.f;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): the weight loads retire here, before the counted waits below

    const float* __restrict__ hx = a.hx + (long)lzz * 2 * a.S * H;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hx), 0, 2 * a.S * H * 4, 0x00020000);
    float* __restrict__ outz = a.pz ? a.outp[z] : a.out + (long)z * a.out_z;
    const unsigned ov = (unsigned)((long)(U0 + l4) * a.out_row + l15);       // (unit, sequence) of this lane inside a frame of `out`
    const float* __restrict__ gxz = a.pz ? a.gxp[z] : a.gx + (long)z * a.gx_z;
    // the wave's 16 gate rows x 16 sequences of a tile as ONE 16 B DMA: lane (row = lane >> 2, sequences 4 (lane & 3) ..) lands
    // at gxs[..][lane][4]; the lane that updates (unit l4, sequence l15) then reads gate k at [(4 l4 + k) 4 + l15 / 4][l15 % 4]
    const unsigned gv = (unsigned)(((long)(4 * U0 + (lane >> 2)) * a.gx_row + 4 * (lane & 3)) * 4);
    const int gidx = (16 * l4 + (l15 >> 2)) * 4 + (l15 & 3);
    const unsigned lds0 = (unsigned)(size_t)hs16, gxs0 = (unsigned)(size_t)gxs;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);

    auto issue_h = [&](int t, int nt, int buf) {             // rows 4 w .. 4 w + 3 of tile nt (h_{t-1}) into buffer buf
        if (a.dbg & 32) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nq = min(nt * 16 + 4 * wave_u + r, a.S - 1);
            const float* src = hx + ((long)((tb + t) & 1) * a.S + nq) * H;
            const unsigned dst = lds0 + (unsigned)((buf * 16 + 4 * wave_u + r) * LDW) * 4u;
            static_for_c<ND>([&](auto I_) {
                constexpr int i = decltype(I_)::value;
                c4_dma16s(src, lane * 16 + i * 1024, dst + i * 1024);
            });
        }
    };
    auto issue = [&](int t, int nt, int buf) {
        issue_h(t, nt, buf);
        const int tt = min(t, Tn - 1), tr = rev ? Tn - 1 - tt : tt;
        const float* gb = gxz + (long)tr * a.gx_t + nt * 16;         // (a last tile with fewer than 16 sequences fetches past its columns: unused)
        c4_dma16g(gb, gv, gxs0 + (unsigned)((buf * 4 + wave_u) * 256) * 4u);
    };
    // this lane's words of the 4 rows its wave fetched: all low bits = tag?  (tag is wave-uniform: an OR chain for tag 0, an AND
    // chain for tag 1 - two three-input operations per 16 B instead of four XORs and two ORs)
    auto stale = [&](const float* tile, unsigned tag) -> bool {
        unsigned o = 0, n = ~0u;
        if (tag) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                static_for_c<ND>([&](auto I_) {
                    constexpr int i = decltype(I_)::value;
                    const uintx4 w = *reinterpret_cast<const uintx4*>(tile + (4 * wave + r) * LDW + (i * 64 + lane) * 4);
                    n = (n & w[0] & w[1]) & (w[2] & w[3]);
                });
            o = ~n;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                static_for_c<ND>([&](auto I_) {
                    constexpr int i = decltype(I_)::value;
                    const uintx4 w = *reinterpret_cast<const uintx4*>(tile + (4 * wave + r) * LDW + (i * 64 + lane) * 4);
                    o = (o | w[0] | w[1]) | (w[2] | w[3]);
                });
        }
        return __builtin_amdgcn_ballot_w64((o & 1u) != 0) != 0;
    };

    const int Q = Tn * NTL;                                  // slots: (step, tile) pairs
    const bool ahead = NTL >= 2;                             // with one tile per step the next slot's h does not exist yet: fetch at the check
    __syncthreads();                                         // (cell states zeroed)
    issue(0, ss, 0);
    int tq = 0, jq = 0, par = 0;
    for (int q = 0; q < Q; ++q) {
        const int nt = ss + jq * a.SS;
        const unsigned tag = (unsigned)((tb + tq) >> 1) & 1u;
        // younger than this slot's group: the previous slot's two stores (the first slot: nothing)
        if (q >= 1) c4_wait_vm<2>();
        else c4_wait_vm<0>();
        float* hb = hs16 + par * (16 * LDW);
        // (one tile per step: nothing was fetched ahead, the buffer still holds h_{t-3} - straight to the fetch loop)
        if (!(a.dbg & (4 | 32)) && ((!ahead && q > 0) || stale(hb, tag))) {
            const unsigned long long t0 = wall_clock64();
            do {
                if (wall_clock64() - t0 > 400000000ull) __builtin_trap();     // 4 s @ 100 MHz: never hang the GPU
                issue_h(tq, nt, par);
                c4_wait_vm<0>();
            } while (stale(hb, tag));
        }
        float gcur[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) gcur[k] = gxs[(par * 4 + wave) * 256 + gidx + 16 * k];
        const float cprev = cw[jq * 64 + lane];
        if (!(a.dbg & 128)) __syncthreads();                 // publishes hb; every wave is done with the other buffer
        // the next slot's group goes out BETWEEN the matrix instructions below (a DMA instruction waits ~60 - 180 cycles for its
        // turn at the memory pipe; behind a 128-cycle group of matrix instructions that wait is free, in front of the loop it is not)
        int tn = tq, jn = jq + 1;
        if (jn == NTL) { jn = 0; ++tn; }
        const int ntn = ss + jn * a.SS;
        const float* nsrc[4];
        unsigned ndst[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int nq = min(ntn * 16 + 4 * wave_u + r, a.S - 1);
            nsrc[r] = hx + ((long)((tb + tn) & 1) * a.S + nq) * H;
            ndst[r] = lds0 + (unsigned)(((par ^ 1) * 16 + 4 * wave_u + r) * LDW) * 4u;
        }
        const bool fetch_h = ahead && !(a.dbg & 32);
        {
            const int tt = min(tn, Tn - 1), tr = rev ? Tn - 1 - tt : tt;
            c4_dma16g(gxz + (long)tr * a.gx_t + ntn * 16, gv, gxs0 + (unsigned)(((par ^ 1) * 4 + wave_u) * 256) * 4u);
        }
        floatx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
        {
            const unsigned haddr = (unsigned)(size_t)(hb + l15 * LDW + l4 * KQ);
            floatx4 bq[3];
#define CO_READ(Qr, J) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(Qr) : "v"(haddr), "n"((J) * 16) : "memory")
#define CO_WAIT(Qr, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(Qr))
            CO_READ(bq[0], 0);
            CO_READ(bq[1], 1);
            static_for_c<KQ / 4>([&](auto J_) {
                constexpr int j = decltype(J_)::value;
                if constexpr (j + 2 < KQ / 4) {
                    CO_READ(bq[(j + 2) % 3], j + 2);
                    CO_WAIT(bq[j % 3], 2);
                } else if constexpr (j + 1 < KQ / 4) {
                    CO_WAIT(bq[j % 3], 1);
                } else {
                    CO_WAIT(bq[j % 3], 0);
                }
                const floatx4 b = bq[j % 3];
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][0], b[0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][1], b[1], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][2], b[2], acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][3], b[3], acc3, 0, 0, 0);
                if constexpr (j < 4 * ND) {
                    if (fetch_h) c4_dma16s(nsrc[j / ND], lane * 16 + (j % ND) * 1024, ndst[j / ND] + (j % ND) * 1024);
                }
            });
#undef CO_READ
#undef CO_WAIT
        }
        const floatx4 acc = (acc0 + acc1) + (acc2 + acc3);
        const float cn = sigm_hw(acc[1] + gcur[1]) * cprev + sigm_hw(acc[0] + gcur[0]) * tanh_hw(acc[2] + gcur[2]);
        const float h = sigm_hw(acc[3] + gcur[3]) * tanh_hw(cn);
        cw[jq * 64 + lane] = cn;
        if (!(a.dbg & 64)) {
            // both stores are issued by every wave (lane 0 always qualifies: 16 nt < S): the counted wait relies on it
            const int n0 = nt * 16, tr = rev ? Tn - 1 - tq : tq;
            float* __restrict__ of = outz + (long)tr * a.out_t + n0;         // wave-uniform
            if (n0 + l15 < a.S) of[ov] = h;
            // h of the wave's 4 units (lane groups l4 = 0 .. 3 = 16-lane rows) gathered into lanes 0 - 15: one 16 B store per sequence
            const unsigned hb_ = (__float_as_uint(h) & ~1u) | ((unsigned)((tb + tq + 1) >> 1) & 1u);
            const auto r1 = __builtin_amdgcn_permlane16_swap(hb_, hb_, false, false);      // rows [r0, r0, r2, r2], [r1, r1, r3, r3]
            const auto e = __builtin_amdgcn_permlane32_swap(r1[0], r1[0], false, false);   // [r0 x 4], [r2 x 4]
            const auto o = __builtin_amdgcn_permlane32_swap(r1[1], r1[1], false, false);   // [r1 x 4], [r3 x 4]
            if (lane < 16 && n0 + l15 < a.S)
                __builtin_amdgcn_raw_buffer_store_b128(uintx4{e[0], o[0], e[1], o[1]}, rs,
                                                       (unsigned)((((tb + tq + 1) & 1) * a.S + n0 + l15) * (H * 4) + U0 * 4), 0, 16);
        }
        if (++jq == NTL) { jq = 0; ++tq; }
        par ^= 1;
    }
    if (a.pz)                                                // the cells go back for the layer's next range of steps
        for (int j = 0; j < NTL; ++j)
            if ((ss + j * a.SS) * 16 + l15 < a.S) cellg[(ss + j * a.SS) * 16] = cw[j * 64 + lane];
}

char* coop_scratch(size_t need, hipStream_t s) { return device_scratch(0, need, s); }

template <int H>
void launch_t(LstmCoopArgs a, int n_cu, hipStream_t s) {
    constexpr int US = H / 16;
    const int NT = (a.S + 15) / 16;
    SE_CHECK(US * a.Z <= n_cu, "cooperative LSTM: more unit slices than CUs");
    a.SS = std::max(1, std::min(NT, n_cu / (US * a.Z)));
    static const int dbg = getenv("SE_COOP_DBG") ? atoi(getenv("SE_COOP_DBG")) : 0;
    a.dbg = dbg;
    // 64 arrival flags per (z, sequence slice) + exchange tensor hx [Z][2][S][H]
    constexpr size_t NFLAG = 256 * 64;        // (z, sequence slice) groups x 64 unit-slice words
    const size_t slab = (size_t)a.S * H, hx_bytes = (size_t)a.Z * 2 * slab * sizeof(float);
    char* sc = coop_scratch(NFLAG * sizeof(unsigned) + hx_bytes, s);
    a.bar = reinterpret_cast<unsigned*>(sc);
    a.hx = reinterpret_cast<float*>(sc + NFLAG * sizeof(unsigned));
    // zeroed by a kernel, not a memset node: under hipGraph replay the memset was observed not to be ordered before the
    // cooperative kernel (stale arrival counts let every barrier fall through)
    launch_fill(reinterpret_cast<float*>(a.bar), (long)a.Z * a.SS * 64, 0.f, s);
    // h_{-1} = 0 and c_{-1} = 0 are data, not branches in the kernel
    launch_fill(a.hx, (long)a.Z * 2 * slab, 0.f, s);
    launch_fill(a.cell, (long)a.Z * H * a.S, 0.f, s);
    const size_t shmem = (size_t)2 * 16 * (H + 4) * sizeof(float);
    static bool attr_set[64] = {};
    if (first_on_device(attr_set)) {
        SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_coop_kernel<H>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    }
    void* params[] = {&a};
    SE_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&lstm_coop_kernel<H>), dim3(US * a.SS * a.Z), dim3(256),
                                      params, (unsigned)shmem, s));
}

}  // namespace

bool lstm_coop_supported(int H, int S, int Z) { return (H == 512 || H == 1024) && (H / 16) * Z <= 256 && S <= 4096; }

// sub-tile pipelined form: sequence slices by 4-sequence sub-tiles, LEAD = slots a fetch runs ahead (at most the sub-tiles a
// workgroup owns: the padding slots at the end of the launch are sub-tiles of step T)
template <int H, int LEAD, int NW>
static void launch_c8(LstmCoopArgs a, hipStream_t s) {
    constexpr int US = H / 16;
    static const int dbg = getenv("SE_COOP_DBG") ? atoi(getenv("SE_COOP_DBG")) : 0;
    a.dbg = dbg;
    constexpr size_t NFLAG = 256 * 64;
    const size_t slab = (size_t)a.S * H, hx_bytes = (size_t)a.Z * 2 * slab * sizeof(float);
    char* sc = coop_scratch(NFLAG * sizeof(unsigned) + hx_bytes, s);
    a.bar = reinterpret_cast<unsigned*>(sc);
    a.hx = reinterpret_cast<float*>(sc + NFLAG * sizeof(unsigned));
    launch_fill(a.hx, (long)a.Z * 2 * slab, 0.f, s);          // h_{-1} = 0 (tag 0) ...
    const unsigned one = 1u;
    float stale;
    memcpy(&stale, &one, sizeof(float));
    for (int z = 0; z < a.Z; ++z) launch_fill(a.hx + ((long)z * 2 + 1) * slab, (long)slab, stale, s);   // ... slab 1 must not look like h_0
    const int nsub_max = ((a.S + 3) / 4 + a.SS - 1) / a.SS;
    const size_t wl_floats = (NW == 4 && H == 1024) ? (size_t)NW * (H / 64) * 256 : 0;      // one unit's weights per wave (kernel: UL)
    const size_t shmem = ((size_t)(LEAD + 1) * 4 * (H + 16) + (size_t)(LEAD + 1) * NW * 64 + wl_floats + (size_t)64 * nsub_max) * sizeof(float);
    SE_CHECK(shmem <= 160 * 1024, "cooperative LSTM (sub-tile form): too many sequences per workgroup for the LDS-resident cell state");
    static bool attr_set[64] = {};
    if (first_on_device(attr_set)) {
        SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_coop8_kernel<H, LEAD, NW>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    void* params[] = {&a};
    SE_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&lstm_coop8_kernel<H, LEAD, NW>), dim3(US * a.SS * a.Z),
                                      dim3(64 * NW), params, (unsigned)shmem, s));
}
template <int H>
static bool launch_c16(LstmCoopArgs a, int n_cu, hipStream_t s) {
    static const int on = getenv("SE_COOP16") ? atoi(getenv("SE_COOP16")) : 1;
    static const int min_s = getenv("SE_COOP16_MINS") ? atoi(getenv("SE_COOP16_MINS")) : 17;
    constexpr int US = H / 16;
    if (!on || a.S < min_s || US * a.Z > n_cu || (long)a.S * H * 8 >= (1L << 31) || (double)a.gx_row * 4 * H * 4 >= 4.0e9 ||
        (double)a.out_row * H >= 4.0e9)
        return false;
    const int NT = (a.S + 15) / 16;
    a.SS = std::max(1, std::min(NT, n_cu / (US * a.Z)));
    const int ntl_max = (NT + a.SS - 1) / a.SS;
    const size_t shmem = ((size_t)2 * 16 * (H + 4) + (size_t)2 * 4 * 256 + (size_t)4 * ntl_max * 64) * sizeof(float);
    if (shmem > 160 * 1024) return false;
    static const int dbg = getenv("SE_COOP_DBG") ? atoi(getenv("SE_COOP_DBG")) : 0;
    a.dbg = dbg;
    constexpr size_t NFLAG = 256 * 64;
    const size_t slab = (size_t)a.S * H, hx_bytes = (size_t)a.Z * 2 * slab * sizeof(float);
    char* sc = coop_scratch(NFLAG * sizeof(unsigned) + hx_bytes, s);
    a.bar = reinterpret_cast<unsigned*>(sc);
    a.hx = reinterpret_cast<float*>(sc + NFLAG * sizeof(unsigned));
    launch_fill(a.hx, (long)a.Z * 2 * slab, 0.f, s);          // h_{-1} = 0 (tag 0) ...
    const unsigned one = 1u;
    float stale;
    memcpy(&stale, &one, sizeof(float));
    for (int z = 0; z < a.Z; ++z) launch_fill(a.hx + ((long)z * 2 + 1) * slab, (long)slab, stale, s);   // ... slab 1 must not look like h_0
    static bool attr_set[64] = {};
    if (first_on_device(attr_set)) {
        SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_coop16_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
    }
    void* params[] = {&a};
    SE_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&lstm_coop16_kernel<H>), dim3(US * a.SS * a.Z), dim3(256), params,
                                      (unsigned)shmem, s));
    return true;
}
static int coop_n_cu() {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        SE_HIP(hipGetDevice(&dev));
        SE_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    return n_cu;
}
bool lstm_coop_chunk_supported(int H, int S, int n_layers) {
    static const int on = getenv("SE_LSTM_CHUNK") ? atoi(getenv("SE_LSTM_CHUNK")) : 1;
    if (!on || (H != 512 && H != 1024) || n_layers < 2 || n_layers > 4 || S < 17) return false;
    const int n_cu = coop_n_cu(), US = H / 16, NT = (S + 15) / 16;
    // worth it where one layer alone leaves a workgroup a single tile per step (its exchange is then exposed) and the layers
    // together still fit the chip
    return US * n_layers <= n_cu && NT <= n_cu / US && (long)S * H * 8 < (1L << 31);
}
template <int H>
static void launch_chunk_t(LstmCoopArgs a, int n_layers, hipStream_t s) {
    constexpr int US = H / 16;
    const int n_cu = coop_n_cu(), NT = (a.S + 15) / 16;
    a.SS = std::max(1, std::min(NT, n_cu / (US * a.Z)));
    const int ntl_max = (NT + a.SS - 1) / a.SS;
    const size_t shmem = ((size_t)2 * 16 * (H + 4) + (size_t)2 * 4 * 256 + (size_t)4 * ntl_max * 64) * sizeof(float);
    SE_CHECK(shmem <= 160 * 1024, "chunked cooperative LSTM: too many tiles per workgroup");
    static const int dbg = getenv("SE_COOP_DBG") ? atoi(getenv("SE_COOP_DBG")) : 0;
    a.dbg = dbg;
    a.pz = 1;
    constexpr size_t NFLAG = 256 * 64;
    const size_t slab = (size_t)a.S * H, hx_bytes = (size_t)n_layers * 2 * slab * sizeof(float);
    char* sc = coop_scratch(NFLAG * sizeof(unsigned) + hx_bytes, s);
    a.bar = reinterpret_cast<unsigned*>(sc);
    a.hx = reinterpret_cast<float*>(sc + NFLAG * sizeof(unsigned));
    const unsigned one = 1u;
    float stale;
    memcpy(&stale, &one, sizeof(float));
    for (int z = 0; z < a.Z; ++z)
        if (a.t0[z] == 0) {      // a layer's first range: h_{-1} = 0 (tag 0), and slab 1 must not look like h_0
            launch_fill(a.hx + (long)a.lz[z] * 2 * slab, (long)slab, 0.f, s);
            launch_fill(a.hx + ((long)a.lz[z] * 2 + 1) * slab, (long)slab, stale, s);
        }
    static bool attr_set[64] = {};
    if (first_on_device(attr_set)) {
        SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_coop16_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
    }
    void* params[] = {&a};
    SE_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&lstm_coop16_kernel<H>), dim3(US * a.SS * a.Z), dim3(256), params,
                                      (unsigned)shmem, s));
}
void launch_lstm_coop_chunk(const LstmCoopArgs& a, int n_layers, hipStream_t s) {
    SE_CHECK(a.Z >= 1 && a.Z <= 4 && n_layers <= 4 && (a.H == 512 || a.H == 1024), "launch_lstm_coop_chunk: 1 - 4 layers of 512 / 1024 units");
    SE_CHECK((double)a.gx_row * 4 * a.H * 4 < 4.0e9 && (double)a.out_row * a.H < 4.0e9, "launch_lstm_coop_chunk: tensor too large for 32-bit lane offsets");
    if (a.H == 1024) launch_chunk_t<1024>(a, n_layers, s);
    else launch_chunk_t<512>(a, n_layers, s);
}
template <int H>
static bool launch_c4_n(LstmCoopArgs a, int n_cu, hipStream_t s) {
    static const int on = getenv("SE_COOP4") ? atoi(getenv("SE_COOP4")) : 1;
    static const int min_s = getenv("SE_COOP4_MINS") ? atoi(getenv("SE_COOP4_MINS")) : 17;
    static const int lead_env = getenv("SE_COOP4_LEAD") ? atoi(getenv("SE_COOP4_LEAD")) : 0;
    constexpr int US = H / 16;
    // (the per-lane parts of the gate / output addresses are 32-bit offsets inside one LSTM's tensors)
    if (!on || a.S < min_s || US * a.Z > n_cu || (long)a.S * H * 8 >= (1L << 31) || (double)a.gx_row * 4 * H * 4 >= 4.0e9 ||
        (double)a.out_row * H >= 4.0e9)
        return false;
    const int NS4 = (a.S + 3) / 4;
    a.SS = std::max(1, std::min(NS4, n_cu / (US * a.Z)));
    if ((size_t)((NS4 + a.SS - 1) / a.SS) * 256 > 24 * 1024) return false;        // cell state of the slice must fit LDS
    const int nsub_min = NS4 / a.SS;                           // the fewest sub-tiles a workgroup owns (>= 1)
    int lead = std::min(3, nsub_min);
    if (nsub_min >= 3 && nsub_min < 6) lead = 2;             // a fetch issued 3 slots ahead of a 4-slot cycle would read before h_{t-1} is out
    if (lead_env > 0) lead = std::min(lead_env, std::min(3, nsub_min));
    if (lead <= 1) launch_c8<H, 1, 4>(a, s);
    else if (lead == 2) launch_c8<H, 2, 4>(a, s);
    else launch_c8<H, 3, 4>(a, s);
    return true;
}

template <int H, int NS, bool TAG>
static void launch_ks(LstmCoopArgs a, hipStream_t s) {
    constexpr int NWG = H / 4;
    static const int dbg = getenv("SE_COOP_DBG") ? atoi(getenv("SE_COOP_DBG")) : 0;
    a.dbg = dbg;
    a.SS = 1;
    constexpr size_t NFLAG = 256 * 64;
    const size_t slab = (size_t)a.S * H, hx_bytes = (size_t)a.Z * 2 * slab * sizeof(float);
    char* sc = coop_scratch(NFLAG * sizeof(unsigned) + hx_bytes, s);
    a.bar = reinterpret_cast<unsigned*>(sc);
    a.hx = reinterpret_cast<float*>(sc + NFLAG * sizeof(unsigned));
    launch_fill(reinterpret_cast<float*>(a.bar), (long)a.Z * 256, 0.f, s);
    launch_fill(a.hx, (long)a.Z * 2 * slab, 0.f, s);
    if (TAG) {      // slab 1 must not look like h_0 (tag 0) before h_0 has been written
        const unsigned one = 1u;
        float stale;
        memcpy(&stale, &one, sizeof(float));
        for (int z = 0; z < a.Z; ++z) launch_fill(a.hx + ((long)z * 2 + 1) * slab, (long)slab, stale, s);
    }
    const size_t shmem = (size_t)4 * 16 * (H / 4 + 4) * sizeof(float) + (size_t)4 * 64 * 16;
    static bool attr_set[64] = {};
    if (first_on_device(attr_set)) {
        SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_coop_ks_kernel<H, NS, TAG>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    }
    void* params[] = {&a};
    SE_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&lstm_coop_ks_kernel<H, NS, TAG>), dim3(NWG * a.Z), dim3(256),
                                      params, (unsigned)shmem, s));
}
template <int H>
static void launch_ks_n(const LstmCoopArgs& a, hipStream_t s) {
    // (the tagged exchange pays while a wave's share of h is one or two loads per lane: 4.9 -> 3.1 us per step at one sequence,
    // 5.4 -> 4.2 at four, nothing at sixteen; SE_COOP_TAG=0: flags everywhere)
    static const bool tag = !(getenv("SE_COOP_TAG") && atoi(getenv("SE_COOP_TAG")) == 0);
    if (tag && a.S <= 4) {
        if (a.S <= 1) launch_ks<H, 1, true>(a, s);
        else launch_ks<H, 4, true>(a, s);
        return;
    }
    if (a.S <= 1) launch_ks<H, 1, false>(a, s);
    else if (a.S <= 4) launch_ks<H, 4, false>(a, s);
    else launch_ks<H, 16, false>(a, s);
}

template <int H, int L>
static void launch_stack_t(LstmStackArgs a, hipStream_t s) {
    constexpr int NWG = H / 4, NV = 2 * L - 1;
    const size_t hx_bytes = (size_t)L * 2 * H * sizeof(float);
    char* sc = coop_scratch(256 * 64 * sizeof(unsigned) + hx_bytes, s);
    a.hx = reinterpret_cast<float*>(sc + 256 * 64 * sizeof(unsigned));
    launch_fill(a.hx, (long)L * 2 * H, 0.f, s);
    const unsigned one = 1u;
    float stale;
    memcpy(&stale, &one, sizeof(float));
    for (int l = 0; l < L; ++l) launch_fill(a.hx + ((long)l * 2 + 1) * H, H, stale, s);     // slab 1 must not look like h_0
    const size_t shmem = ((size_t)4 * NV * (H / 4) + (size_t)L * 64) * sizeof(float) + (size_t)(L - 1) * 16 * H * sizeof(float);
    static bool attr_set[64] = {};
    if (first_on_device(attr_set)) {
        SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_stack_kernel<H, L>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)shmem));
    }
    void* params[] = {&a};
    SE_HIP(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&lstm_stack_kernel<H, L>), dim3(NWG), dim3(256), params,
                                      (unsigned)shmem, s));
}
bool lstm_stack_supported(int H, int L) {
    static const bool on = !(getenv("SE_LSTM_STACK") && atoi(getenv("SE_LSTM_STACK")) == 0);
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        SE_HIP(hipGetDevice(&dev));
        SE_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    return on && (H == 1024 || H == 512) && (L == 2 || L == 3) && H / 4 <= n_cu;
}
void launch_lstm_stack(const LstmStackArgs& a, hipStream_t s) {
    SE_CHECK(lstm_stack_supported(a.H, a.L), "launch_lstm_stack: 2 or 3 layers of 512 / 1024 units on one sequence");
    if (a.H == 1024) a.L == 2 ? launch_stack_t<1024, 2>(a, s) : launch_stack_t<1024, 3>(a, s);
    else a.L == 2 ? launch_stack_t<512, 2>(a, s) : launch_stack_t<512, 3>(a, s);
}

void launch_lstm_coop(const LstmCoopArgs& a, hipStream_t s) {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        SE_HIP(hipGetDevice(&dev));
        SE_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    SE_CHECK(a.Z * std::max(1, std::min((a.S + 15) / 16, n_cu / ((a.H / 16) * a.Z))) <= 256, "cooperative LSTM: too many slices");
    static const bool ks_on = !(getenv("SE_COOP_KS") && atoi(getenv("SE_COOP_KS")) == 0);
    // one tile: every CU on the K-split form (H = 1024: 7.5 -> 5.1 ... 5.9 us per step for 1 ... 16 sequences; H = 512: 4.4 -> 3.8 at
    // one sequence, nothing from 8 on or with two LSTMs per launch - tools/coopbench.cpp)
    static const int ks_z = getenv("SE_COOP_KS_Z") ? atoi(getenv("SE_COOP_KS_Z")) : 2;      // (two LSTMs per launch, GCRN: 4.4 -> 2.5 ... 3.0 us)
    if (ks_on && a.S <= (a.H == 1024 ? 16 : 4) && (a.H == 1024 || a.Z <= ks_z) && (a.H / 4) * a.Z <= n_cu && (a.H / 4) * a.Z <= 256) {
        if (a.H == 1024) launch_ks_n<1024>(a, s);
        else if (a.H == 512) launch_ks_n<512>(a, s);
        else SE_CHECK(false, "cooperative LSTM kernel is built for H = 512 / 1024");
        return;
    }
    // 16-sequence tiles (lstm_coop16_kernel) unless a workgroup would own a single tile per step AND fewer than 16 sequences per
    // unit slice's share of the chip are left to split: there the 4-sequence sub-tiles overlap their own exchange (batch 32,
    // H = 1024: 5.1 us per step against 7.4; batch 64: 8.1 against 7.8; SE_COOP4_FIRST = 1 / 0 forces / forbids)
    static const int sub4_env = getenv("SE_COOP4_FIRST") ? atoi(getenv("SE_COOP4_FIRST")) : -1;
    const int us_z = (a.H / 16) * a.Z, nt16 = (a.S + 15) / 16;
    const bool sub4 = sub4_env >= 0 ? sub4_env != 0 : (us_z <= n_cu && nt16 <= n_cu / us_z && a.S * us_z < 60 * n_cu / 4 && a.H == 1024);
    if (sub4 && a.H == 1024 && launch_c4_n<1024>(a, n_cu, s)) return;
    if (sub4 && a.H == 512 && launch_c4_n<512>(a, n_cu, s)) return;
    if (a.H == 1024 && launch_c16<1024>(a, n_cu, s)) return;
    if (a.H == 512 && launch_c16<512>(a, n_cu, s)) return;
    if (a.H == 1024 && launch_c4_n<1024>(a, n_cu, s)) return;
    if (a.H == 512 && launch_c4_n<512>(a, n_cu, s)) return;
    if (a.H == 1024) launch_t<1024>(a, n_cu, s);
    else if (a.H == 512) launch_t<512>(a, n_cu, s);
    else SE_CHECK(false, "cooperative LSTM kernel is built for H = 512 / 1024");
}

}  // namespace se
