// Register-resident STFT / iSTFT-OLA for gfx950 (round 3; the round-1/2 LDS-Stockham kernels stay in k_stft.hip behind
// SE_STFT_V1=1 for A/B runs).
//
// Reference behaviour: see k_stft.hip (torch.stft / librosa.stft, centre = True, reflect pad, periodic Hann, one-sided;
// torch.istft / librosa.istft with window-sum-square normalisation) - e.g. DCCRN/dccrn_decode_vb.py:37-38, :59-60.
//
// FFT: one 64-lane wave transforms one complex N-point sequence (= two real frames, two-for-one) with its points in
// REGISTERS: N = R1 * 8 * 8 (512: R1 = 8; 320: R1 = 5).  Lane l starts with the R1 points n = l + 64 j; radix-R1
// butterfly, twiddle, exchange through a wave-private LDS strip, radix-8, twiddle, exchange, radix-8: two LDS exchanges
// instead of the five Stockham round trips, no block barrier inside a transform (the DS operations of a wave execute in
// order).  Both exchange layouts are rotated so that the b64 writes and reads are bank-conflict free.  Twiddles and the
// window come from a table built once per (device, N, win) on the host in float64 - the old kernels evaluated N double
// sincospi per 16-frame block.
//
// Spectrogram access ([B][2][F][Tp], frames contiguous): a block owns 32 consecutive frames (16 transforms, four per
// wave) and moves them through an LDS tile [F][33], one plane at a time, so that every global access is a 128 B run
// along T (the old kernels: 64 B runs; PMC: 3.9x read amplification in the inverse).  Forward: the untangled spectra
// wait in registers while the planes leave one after the other; inverse: the two planes are gathered into registers one
// after the other, the windowed frames stay in LDS (a wave's exchange strip IS the two rows its frames end up in) and
// the overlap-add reads them from there.
#include "kernels.h"
#include "common.h"
#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace se {

namespace {

constexpr int NFB = 32;            // frames per block
// waves per block (16 transforms = NQ per wave).  Measured at batch 256 (DCCRN geometry): the forward kernel is fastest with
// four waves (two independent blocks per CU: 0.135 ms against 0.166 ms with one block of eight - its untangled spectra wait
// in 80 registers per wave either way), the inverse with eight (half the per-thread work in its load / overlap-add phases
// and 126 VGPRs: two blocks = four waves per SIMD, 0.144 ms against 0.184 ms)
template <bool INVERSE> struct Shape {
    static constexpr int NW = INVERSE ? 8 : 4;
    static constexpr int NT = NW * 64, NQ = 16 / NW, KSTEP = NT / NFB;
};
constexpr int TPITCH = NFB + 1;    // tile row pitch (floats): lanes walk bins at stride 33 words - conflict free

#define SE_WAVE_FENCE2() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// complex values as 2-wide vectors: adds / subs / scalings compile to v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 (one VALU
// instruction per complex operation instead of two), a complex product to two packed instructions
typedef float cf __attribute__((ext_vector_type(2)));
#define float2 cf
#define make_float2(a, b) (cf{(a), (b)})
__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
__device__ __forceinline__ cf cmul2(cf a, cf b) { return a.xx * b + a.yy * cf{-b.y, b.x}; }
// multiply by -i (forward) / +i (inverse)
template <bool INV>
__device__ __forceinline__ cf rot90(cf d) { return INV ? cf{-d.y, d.x} : cf{d.y, -d.x}; }

// out[k] = sum_j x[j] w^{jk}, w = exp(-+ 2 pi i / 4)
template <bool INV>
__device__ __forceinline__ void bfly4(float2& x0, float2& x1, float2& x2, float2& x3) {
    const float2 a = cadd(x0, x2), b = csub(x0, x2), c = cadd(x1, x3), d = rot90<INV>(csub(x1, x3));
    x0 = cadd(a, c);
    x2 = csub(a, c);
    x1 = cadd(b, d);
    x3 = csub(b, d);
}
// radix-8, natural order in and out
template <bool INV>
__device__ __forceinline__ void bfly8(float2 (&x)[8]) {
    float2 e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6], o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
    bfly4<INV>(e0, e1, e2, e3);
    bfly4<INV>(o0, o1, o2, o3);
    constexpr float h = 0.70710678118654752f;
    // w^1 = (1 -+ i) / sqrt2, w^2 = -+ i, w^3 = (-1 -+ i) / sqrt2:  w^1 o = h (o + rot90 o),  w^3 o = h (rot90 o - o)
    const float2 t1 = (o1 + rot90<INV>(o1)) * h;
    const float2 t2 = rot90<INV>(o2);
    const float2 t3 = (rot90<INV>(o3) - o3) * h;
    x[0] = cadd(e0, o0); x[4] = csub(e0, o0);
    x[1] = cadd(e1, t1); x[5] = csub(e1, t1);
    x[2] = cadd(e2, t2); x[6] = csub(e2, t2);
    x[3] = cadd(e3, t3); x[7] = csub(e3, t3);
}
template <bool INV>
__device__ __forceinline__ void bfly5(float2 (&x)[8]) {
    constexpr float c1 = 0.30901699437494742f, s1 = 0.95105651629515357f;    // cos / sin 2 pi / 5
    constexpr float c2 = -0.80901699437494742f, s2 = 0.58778525229247313f;   // cos / sin 4 pi / 5
    const float sg = INV ? 1.f : -1.f;
    const float2 t1 = cadd(x[1], x[4]), t2 = cadd(x[2], x[3]), t3 = csub(x[1], x[4]), t4 = csub(x[2], x[3]);
    const float2 m1 = x[0] + t1 * c1 + t2 * c2;
    const float2 m2 = x[0] + t1 * c2 + t2 * c1;
    const float2 u1 = t3 * s1 + t4 * s2;
    const float2 u2 = t3 * s2 - t4 * s1;
    const float2 j1 = make_float2(-sg * u1.y, sg * u1.x), j2 = make_float2(-sg * u2.y, sg * u2.x);
    x[0] = x[0] + t1 + t2;
    x[1] = cadd(m1, j1);
    x[4] = csub(m1, j1);
    x[2] = cadd(m2, j2);
    x[3] = csub(m2, j2);
}

// Per-lane constants of the transform (loaded once per block from the host-built table)
template <int N>
struct FftConst {
    static constexpr int R1 = N / 64;
    float2 tw1[8];     // W_N^{lane * k1}
    float2 tw2[8];     // W_64^{b * c}, b = lane & 7
    int k1, b;         // stage-2/3 role of this lane: lane = k1 * 8 + b (active while lane < 8 * R1)
    bool act;
    __device__ __forceinline__ void load(const float2* __restrict__ tab, int lane) {
        k1 = min(lane >> 3, R1 - 1);
        b = lane & 7;
        act = lane < 8 * R1;
#pragma unroll
        for (int k = 0; k < R1; ++k) tw1[k] = tab[(lane * k) % N];
#pragma unroll
        for (int c = 0; c < 8; ++c) tw2[c] = tab[((N / 64) * b * c) % N];
    }
};

// The transform.  x[j], j < R1: points n = lane + 64 j.  Returns with v[d] = X[k1 + R1 * (c + 8 d)] where the lane's
// (k1, c) = (lane >> 3, lane & 7); lanes >= 8 * R1 hold nothing.  `ws`: wave-private strip of N float2.
template <int N, bool INV>
__device__ __forceinline__ void fft_regs(float2 (&x)[8], float2* __restrict__ ws, const FftConst<N>& K, int lane) {
    constexpr int R1 = N / 64;
    if (R1 == 8) bfly8<INV>(x);
    else bfly5<INV>(x);
#pragma unroll
    for (int k = 1; k < R1; ++k) {
        float2 w = K.tw1[k];
        if (INV) w.y = -w.y;
        x[k] = cmul2(x[k], w);
    }
    // exchange 1: element (k1, n2 = 8 a + b) at k1 * 64 + ((a + k1) & 7) * 8 + b
    {
        const int a = lane >> 3, bb = lane & 7;
#pragma unroll
        for (int k = 0; k < R1; ++k) ws[k * 64 + ((a + k) & 7) * 8 + bb] = x[k];
    }
    SE_WAVE_FENCE2();
    float2 u[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) u[a] = ws[K.k1 * 64 + ((a + K.k1) & 7) * 8 + K.b];
    bfly8<INV>(u);
#pragma unroll
    for (int c = 1; c < 8; ++c) {
        float2 w = K.tw2[c];
        if (INV) w.y = -w.y;
        u[c] = cmul2(u[c], w);
    }
    // exchange 2: element (k1, c, b) at k1 * 64 + ((c + k1) & 7) * 8 + ((b + c) & 7)
    if (K.act) {
#pragma unroll
        for (int c = 0; c < 8; ++c) ws[K.k1 * 64 + ((c + K.k1) & 7) * 8 + ((K.b + c) & 7)] = u[c];
    }
    SE_WAVE_FENCE2();
    {
        const int c = K.b;          // the lane's low three bits now name the stage-2 output it gathers
#pragma unroll
        for (int bb = 0; bb < 8; ++bb) x[bb] = ws[K.k1 * 64 + ((c + K.k1) & 7) * 8 + ((bb + c) & 7)];
    }
    bfly8<INV>(x);
    SE_WAVE_FENCE2();
}

// ------------------------------------------------------------------------------------------------ tables
struct FftTab {
    float* dev = nullptr;     // [N] float2 twiddles exp(-2 pi i j / N), then [N] float window (centred, zero padded)
};
const float* fft_table(int N, int win) {
    static std::map<std::tuple<int, int, int>, FftTab> tabs;
    static std::mutex mu;
    int dev = 0;
    SE_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    FftTab& t = tabs[std::make_tuple(dev, N, win)];
    if (!t.dev) {
        std::vector<float> h((size_t)3 * N, 0.f);
        const double pi = 3.14159265358979323846;
        for (int j = 0; j < N; ++j) {
            h[2 * j] = (float)std::cos(2.0 * pi * j / N);
            h[2 * j + 1] = (float)(-std::sin(2.0 * pi * j / N));
        }
        const int left = (N - win) / 2;
        for (int j = 0; j < win; ++j) h[2 * N + left + j] = (float)(0.5 - 0.5 * std::cos(2.0 * pi * j / win));
        SE_HIP(hipMalloc(&t.dev, h.size() * sizeof(float)));
        SE_HIP(hipMemcpy(t.dev, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return t.dev;
}

struct Stft2Args {
    const float* wav; long pitch; int B, L, Lpad; const float* c_scale; float p_in;
    float* spec; float* mag; int T, Tp, hop;
    const int *len, *lpad, *tlen;
    int t_first, col0;
    const float* tab;
};

// ------------------------------------------------------------------------------------------------ forward
// CP: magnitude exponent of the decode script: 0 none (1.0), 1 square root (0.5, every script's compressed variant), 2 powf
template <int N, bool MAG, int CP>
__global__ __launch_bounds__(Shape<false>::NT) void stft2_kernel(const Stft2Args a) {
    constexpr int NW = Shape<false>::NW, NT = Shape<false>::NT, NQ = Shape<false>::NQ, KSTEP = Shape<false>::KSTEP;
    constexpr int F = N / 2 + 1, R1 = N / 64, NB = (F + 63) / 64;     // NB bins per lane (k = lane + 64 m)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [NW][N] float2 exchange strips; the tile [F][33] floats of the output phases aliases them (used after a barrier)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float2* ws = reinterpret_cast<float2*>(smem) + wave * N;
    float* tile = smem;
    const int b = blockIdx.y, t0 = a.t_first + blockIdx.x * NFB;
    const int cshift = a.col0 - a.t_first;
    const int L = a.len ? a.len[b] : a.L, Lpad = a.lpad ? a.lpad[b] : a.Lpad, Tb = a.tlen ? a.tlen[b] : a.T;
    float* sre = a.spec ? a.spec + ((long)b * 2 + 0) * F * a.Tp + cshift : nullptr;
    float* sim = a.spec ? a.spec + ((long)b * 2 + 1) * F * a.Tp + cshift : nullptr;
    float* smg = a.mag ? a.mag + (long)b * F * a.Tp + cshift : nullptr;
    if (t0 >= Tb) {            // ragged batch: the block lies wholly in the row's zero tail (block-uniform)
        for (int idx = tid; idx < F * NFB; idx += NT) {
            const int t = t0 + (idx & (NFB - 1)), k = idx >> 5;
            if (t >= a.T) continue;
            if (sre) {
                sre[(long)k * a.Tp + t] = 0.f;
                sim[(long)k * a.Tp + t] = 0.f;
            }
            if (smg) smg[(long)k * a.Tp + t] = 0.f;
        }
        return;
    }
    FftConst<N> K;
    K.load(reinterpret_cast<const float2*>(a.tab), lane);
    const float* wtab = a.tab + 2 * N;
    float wn[8];
#pragma unroll
    for (int j = 0; j < R1; ++j) wn[j] = wtab[lane + 64 * j];
    const float c = a.c_scale ? a.c_scale[b] : 1.f;
    const float* x = a.wav + (long)b * a.pitch;
    auto sample = [&](int t, int n, float w) {
        float v = 0.f;
        if (t < Tb) {
            int idx = t * a.hop + n - N / 2;
            if (idx < 0) idx = -idx;
            if (idx >= Lpad) idx = 2 * (Lpad - 1) - idx;
            if (idx >= 0 && idx < L) v = x[idx] * c * w;
        }
        return v;
    };
    float vre[NQ][2 * NB], vim[NQ][2 * NB];     // [transform][bin slot m, frame parity]: untangled, compressed spectra
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int P = wave + NW * q, t = t0 + 2 * P;       // frames (t, t + 1)
        float2 z[8];
        bool nz0 = false, nz1 = false;
        // frames wholly inside the clip (all but the first two and the last few of a row): no reflection, no bounds - the
        // wave-uniform fast path loads x[t * hop - N/2 + lane + 64 j] with immediate offsets
        if (t + 1 < Tb && t * a.hop >= N / 2 && (t + 1) * a.hop + N / 2 <= L) {
            const float* p0 = x + (t * a.hop - N / 2 + lane);
#pragma unroll
            for (int j = 0; j < R1; ++j) {
                const float cw = c * wn[j];
                z[j] = make_float2(p0[64 * j] * cw, p0[a.hop + 64 * j] * cw);
            }
        } else {
#pragma unroll
            for (int j = 0; j < R1; ++j) z[j] = make_float2(sample(t, lane + 64 * j, wn[j]), sample(t + 1, lane + 64 * j, wn[j]));
        }
#pragma unroll
        for (int j = 0; j < R1; ++j) {
            nz0 |= (z[j].x != 0.f);
            nz1 |= (z[j].y != 0.f);
        }
        // a frame of digital silence must transform to EXACT zeros (k_stft.hip: the decode scripts re-use atan2 of it)
        const bool a0 = __any(nz0), a1 = __any(nz1);
        fft_regs<N, false>(z, ws, K, lane);
        // natural order into the strip: Z[k1 + R1 * (c + 8 d)]
        if (K.act) {
#pragma unroll
            for (int d = 0; d < 8; ++d) ws[K.k1 + R1 * (K.b + 8 * d)] = z[d];
        }
        SE_WAVE_FENCE2();
        // two-for-one: X_t[k] = (Z[k] + conj Z[N-k]) / 2,  X_{t+1}[k] = (Z[k] - conj Z[N-k]) / (2i)
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            const int k = lane + 64 * m;
            float2 v0 = make_float2(0.f, 0.f), v1 = make_float2(0.f, 0.f);
            if (k < F) {
                const float2 zk = ws[k], zc = ws[k == 0 ? 0 : N - k];
                v0 = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));
                v1 = make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x));
            }
            if (!a0) v0 = make_float2(0.f, 0.f);
            if (!a1) v1 = make_float2(0.f, 0.f);
            vre[q][2 * m] = v0.x; vim[q][2 * m] = v0.y;
            vre[q][2 * m + 1] = v1.x; vim[q][2 * m + 1] = v1.y;
        }
        SE_WAVE_FENCE2();
    }
    // |X|^p e^{j angle X} (decode scripts' compression, e.g. dccrn_decode_vb.py:40-42) and the optional magnitude plane
    const bool want_mag = MAG && smg != nullptr;
    constexpr bool cprs = CP != 0;
    float vmg[MAG ? NQ : 1][MAG ? 2 * NB : 1];
    if (cprs || MAG) {
        constexpr bool half = CP == 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int s = 0; s < 2 * NB; ++s) {
                const float m2 = vre[q][s] * vre[q][s] + vim[q][s] * vim[q][s];
                const bool pos = !(m2 <= 1e-37f);                  // (below: the hardware square root flushes to zero; NaN / inf pass through)
                const float m = pos ? fm_sqrt(m2) : 0.f;       // v_sqrt_f32, 1 ulp
                float mp = m;
                if (cprs) {
                    float sc;
                    if (half) {
                        sc = pos ? fm_rsq(m) : 0.f;            // |X|^0.5 / |X| = |X|^-0.5
                        mp = m * sc;
                    } else {
                        mp = powf(m, a.p_in);
                        sc = m > 0.f ? mp / m : 0.f;
                    }
                    vre[q][s] *= sc;
                    vim[q][s] *= sc;
                }
                if (MAG) vmg[q][s] = mp;
            }
    }
    // the planes leave one after the other through the tile (128 B runs along T)
    const int nplanes = (sre ? 2 : 0) + (want_mag ? 1 : 0);
    for (int pl = 0; pl < nplanes; ++pl) {
        const int which = sre ? pl : 2;            // 0 re, 1 im, 2 mag
        __syncthreads();                            // the strips / the previous plane's tile are done with
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int f0 = 2 * (wave + NW * q);
#pragma unroll
            for (int m = 0; m < NB; ++m) {
                const int k = lane + 64 * m;
                if (k >= F) continue;
                const float e0 = which == 0 ? vre[q][2 * m] : (which == 1 ? vim[q][2 * m] : vmg[MAG ? q : 0][MAG ? 2 * m : 0]);
                const float e1 = which == 0 ? vre[q][2 * m + 1] : (which == 1 ? vim[q][2 * m + 1] : vmg[MAG ? q : 0][MAG ? 2 * m + 1 : 0]);
                tile[k * TPITCH + f0] = e0;
                tile[k * TPITCH + f0 + 1] = e1;
            }
        }
        __syncthreads();
        float* dst = which == 0 ? sre : (which == 1 ? sim : smg);
        {
            const int f = tid & (NFB - 1), t = t0 + f;
#pragma unroll 8
            for (int it = 0; it < (F * NFB + NT - 1) / NT; ++it) {
                const int k = (tid >> 5) + KSTEP * it;
                if (k < F && t < a.T) dst[(long)k * a.Tp + t] = tile[k * TPITCH + f];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ inverse
struct Istft2Args {
    const float* spec; int B, T, Tp; int hop;
    const float* c_scale; float* out; long out_pitch; int Lout;
    int own, halo;
    const int *tlen, *olen;
    int t_off, t_lo, o_lo, pos_base;
    const float* tab;
    const float* frame_inv; int ring_mask;      // optional [B][ring]: frame t is multiplied by frame_inv[b][t & ring_mask]
};

// FSC: per-frame scales (Istft2Args::frame_inv) - a variant of its own: the two extra values per transform lift the common
// kernel from 126 to 134 VGPRs, i.e. from four to three waves per SIMD (0.138 -> 0.219 ms at batch 256)
template <int N, bool FSC>
__global__ __launch_bounds__(Shape<true>::NT) void istft2_kernel(const Istft2Args a) {
    constexpr int NW = Shape<true>::NW, NT = Shape<true>::NT, NQ = Shape<true>::NQ, KSTEP = Shape<true>::KSTEP;
    constexpr int F = N / 2 + 1, R1 = N / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // frames [32][N] floats: rows 2P, 2P+1 double as the exchange strip of transform P (N float2 = 2 N floats); the
    // gather tile [F][33] aliases the same memory before any transform starts; then the window [N]
    float* frames = smem;
    float* tile = smem;
    float* wls = smem + NFB * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Neighbouring blocks of a row read overlapping frame windows (halo) out of the same 128 B lines, but consecutive
    // workgroup ids go to different XCDs (id mod 8), each with its own L2: remap so that an XCD walks a contiguous range of
    // (row, window) pairs and the shared lines are L2 hits (PMC: 0.598 -> 0.266 GB read per launch at batch 256, 2.3x -> 1.0x
    // the algorithmic bytes)
    int bx = blockIdx.x, b = blockIdx.y;
    {
        const unsigned nx = gridDim.x, total = nx * gridDim.y, lin = blockIdx.x + nx * blockIdx.y;
        const unsigned per = total >> 3, body = per << 3;
        if (per > 0 && lin < body) {
            const unsigned m = (lin & 7) * per + (lin >> 3);
            bx = m % nx;
            b = m / nx;
        }
    }
    const int Tb = a.tlen ? a.tlen[b] : a.T;
    const int Lo = a.olen ? a.olen[b] : a.Lout;
    const int span = a.own * a.hop;
    const int pos0 = a.pos_base + bx * span;
    const int tb = pos0 / a.hop - a.halo;
    float* outp = a.out + (long)b * a.out_pitch - a.o_lo;
    if (pos0 - N / 2 >= Lo || tb >= Tb) {       // nothing of this row left: zeros up to Lout (block-uniform)
        for (int i = tid; i < span; i += NT) {
            const int o = pos0 + i - N / 2;
            if (o >= a.o_lo && o < a.Lout) outp[o] = 0.f;
        }
        return;
    }
    FftConst<N> K;
    K.load(reinterpret_cast<const float2*>(a.tab), lane);
    const float* wtab = a.tab + 2 * N;
    for (int n = tid; n < N; n += NT) wls[n] = wtab[n];
    // ---- gather: plane by plane through the tile; z[q][j] = Z[lane + 64 j] of transform q of this wave.
    // All 33 row groups of a plane are loaded in one batch (independent 128 B-run loads in flight per thread), and the
    // imaginary plane's batch is issued before the real plane is gathered out of the tile, so its latency hides there.
    float2 z[NQ][8];
    constexpr int NIT = (F * NFB + NT - 1) / NT;
    const int tf = tid & (NFB - 1), tt = tb + tf, tk = tid >> 5;
    const bool tv = tt >= a.t_lo && tt < Tb;
    auto load_plane = [&](int pl, float (&v)[NIT]) {
        const float* src = a.spec + ((long)b * 2 + pl) * F * a.Tp - a.t_off + tt;
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int k = tk + KSTEP * u;
            v[u] = (k < F && tv) ? src[(long)k * a.Tp] : 0.f;
        }
    };
    auto store_tile = [&](const float (&v)[NIT]) {
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int k = tk + KSTEP * u;
            if (k < F) tile[k * TPITCH + tf] = v[u];
        }
    };
    auto gather = [&](int pl) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int f0 = 2 * (wave + NW * q);
#pragma unroll
            for (int j = 0; j < R1; ++j) {
                const int ki = lane + 64 * j;
                const bool hi = ki > N / 2;
                const int k = hi ? N - ki : ki;
                const float va = tile[k * TPITCH + f0], vb = tile[k * TPITCH + f0 + 1];
                const bool edge = (k == 0 || k == N / 2);      // C2R ignores the imaginary part of DC / Nyquist
                if (pl == 0) {
                    z[q][j] = make_float2(va, vb);             // real parts: Z = (xa.re -+ xb.im, +-xa.im + xb.re)
                } else if (!edge) {
                    // k <= N/2: Z[k] = X_a[k] + i X_b[k];  k > N/2: conj X_a[N-k] + i conj X_b[N-k]
                    if (!hi) z[q][j] = make_float2(z[q][j].x - vb, z[q][j].y + va);
                    else z[q][j] = make_float2(z[q][j].x + vb, z[q][j].y - va);
                }
            }
        }
    };
    {
        float v0[NIT], v1[NIT];
        load_plane(0, v0);
        store_tile(v0);
        load_plane(1, v1);          // in flight while plane 0 is gathered
        __syncthreads();
        gather(0);
        __syncthreads();
        store_tile(v1);
        __syncthreads();
        gather(1);
    }
    __syncthreads();            // the tile is dead: its memory becomes the frames / exchange strips
    const float invN = 1.f / N;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int P = wave + NW * q;
        float2* ws = reinterpret_cast<float2*>(frames + (size_t)(2 * P) * N);
        fft_regs<N, true>(z[q], ws, K, lane);
        // z[q][d] = time sample n = k1 + R1 * (c + 8 d): real part -> frame 2P, imaginary part -> frame 2P + 1, windowed
        if (K.act) {
            // frame-online streams with a running scale: every frame was transformed under its own c and is taken back by it
            float s0 = invN, s1 = invN;
            if (FSC) {
                const float* fr = a.frame_inv + (long)b * (a.ring_mask + 1);
                s0 *= fr[(tb + 2 * P) & a.ring_mask];
                s1 *= fr[(tb + 2 * P + 1) & a.ring_mask];
            }
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const int n = K.k1 + R1 * (K.b + 8 * d);
                const float w = wls[n];
                frames[(size_t)(2 * P) * N + n] = z[q][d].x * (w * s0);
                frames[(size_t)(2 * P + 1) * N + n] = z[q][d].y * (w * s1);
            }
        }
    }
    __syncthreads();
    // overlap-add out of LDS (as k_stft.hip): position pos gets frame t = tb + fi at n = pos - t * hop
    const float cinv = a.c_scale ? 1.f / a.c_scale[b] : 1.f;
    for (int i = tid; i < span; i += NT) {
        const int pos = pos0 + i, o = pos - N / 2;
        if (o < a.o_lo || o >= a.Lout) continue;
        if (o >= Lo) {
            outp[o] = 0.f;
            continue;
        }
        int thi = pos / a.hop;
        if (thi > Tb - 1) thi = Tb - 1;
        float acc = 0.f, env = 0.f;
        for (int t = thi; t >= a.t_lo && pos - t * a.hop < N; --t) {
            const int n = pos - t * a.hop, fi = t - tb;
            const float w = wls[n];
            acc += frames[(size_t)fi * N + n];
            env += w * w;
        }
        float y = env > 1e-11f ? acc / env : acc;
        outp[o] = y * cinv;
    }
}

template <int N>
constexpr size_t stft2_lds() { return std::max((size_t)Shape<false>::NW * N * 8, (size_t)(N / 2 + 1) * TPITCH * 4); }
template <int N>
constexpr size_t istft2_lds() { return std::max((size_t)NFB * N * 4, (size_t)(N / 2 + 1) * TPITCH * 4) + (size_t)N * 4; }

template <typename Kern>
void set_lds(Kern kernel, size_t bytes) {
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

}  // namespace

bool stft2_enabled() {
    static const bool on = !(getenv("SE_STFT_V1") && atoi(getenv("SE_STFT_V1")) != 0);
    return on;
}

void launch_stft2(const StftGeom& g, const float* wav, long pitch, int B, int L, int Lpad, const float* c_scale, float p_in,
                  float* spec_ri, float* mag, int T, int Tp, hipStream_t s, int t_first, int col0) {
    const Ragged* rg = ragged_ctx();
    Stft2Args a{wav, pitch, B, L, Lpad, c_scale, p_in, spec_ri, mag, T, Tp, g.hop,
                rg ? rg->len : nullptr, rg ? rg->lpad : nullptr, rg ? rg->tlen : nullptr, t_first, col0,
                fft_table(g.n_fft, g.win)};
    dim3 grid((T - t_first + NFB - 1) / NFB, B);
    // (every variant stays below the 64 KB of LDS a kernel may use without raising its limit)
    const int cp = p_in == 1.f ? 0 : (p_in == 0.5f ? 1 : 2);
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, dim3(Shape<false>::NT), g.n_fft == 512 ? stft2_lds<512>() : stft2_lds<320>(), s, a); };
#define SE_STFT2_CASE(NN, MM) \
    (cp == 0 ? go(stft2_kernel<NN, MM, 0>) : cp == 1 ? go(stft2_kernel<NN, MM, 1>) : go(stft2_kernel<NN, MM, 2>))
    if (g.n_fft == 512) {
        if (mag) SE_STFT2_CASE(512, true);
        else SE_STFT2_CASE(512, false);
    } else {
        if (mag) SE_STFT2_CASE(320, true);
        else SE_STFT2_CASE(320, false);
    }
#undef SE_STFT2_CASE
    SE_HIP(hipGetLastError());
}

void launch_istft2(const StftGeom& g, const float* spec_ri, int B, int T, int Tp, const float* c_scale, float* wav_out,
                   long out_pitch, int Lout, hipStream_t s, int t_off, int t_lo, int o_lo, const float* frame_inv, int ring) {
    const Ragged* rg = ragged_ctx();
    int halo = (g.n_fft + g.hop - 1) / g.hop - 1;
    halo += halo & 1;                                       // frames are transformed in pairs
    SE_CHECK(halo < NFB, "hop too small for the fused overlap-add window");
    SE_CHECK(!frame_inv || (ring > 0 && (ring & (ring - 1)) == 0), "launch_istft2: per-frame scales live in a power-of-two ring");
    const int own = NFB - halo;
    Istft2Args a{spec_ri, B, T, Tp, g.hop, c_scale, wav_out, out_pitch, Lout, own, halo,
                 rg ? rg->tlen : nullptr, rg ? rg->olen : nullptr, t_off, t_lo, o_lo, 0, fft_table(g.n_fft, g.win), frame_inv, frame_inv ? ring - 1 : 0};
    const int span = own * g.hop;
    a.pos_base = (o_lo + g.n_fft / 2) / g.hop * g.hop;
    dim3 grid((g.n_fft / 2 + Lout - a.pos_base + span - 1) / span, B);
    static bool seen[64] = {};
    if (first_on_device(seen)) {
        set_lds(istft2_kernel<512, false>, istft2_lds<512>());
        set_lds(istft2_kernel<512, true>, istft2_lds<512>());
        set_lds(istft2_kernel<320, false>, istft2_lds<320>());
        set_lds(istft2_kernel<320, true>, istft2_lds<320>());
    }
    const dim3 blk(Shape<true>::NT);
    if (g.n_fft == 512) {
        if (frame_inv) hipLaunchKernelGGL((istft2_kernel<512, true>), grid, blk, istft2_lds<512>(), s, a);
        else hipLaunchKernelGGL((istft2_kernel<512, false>), grid, blk, istft2_lds<512>(), s, a);
    } else {
        if (frame_inv) hipLaunchKernelGGL((istft2_kernel<320, true>), grid, blk, istft2_lds<320>(), s, a);
        else hipLaunchKernelGGL((istft2_kernel<320, false>), grid, blk, istft2_lds<320>(), s, a);
    }
    SE_HIP(hipGetLastError());
}

}  // namespace se
