// Framed STFT / iSTFT-OLA for gfx950.
//
// Reference behaviour (file:line in /root/reference):
//   torch.stft(x, n_fft, hop, win, window=torch.hann_window(win))   DCCRN/dccrn_decode_vb.py:37-38,
//       FullSubNet/fullsubnet_sa_decode_vb.py:46-47, CTSNet/two_stage_com_decode_vb.py:70-71,
//       TaylorSENet/taylorsenet_decode_vb.py:36-37, Uformer/uformer.py:178,182
//   librosa.stft(x, n_fft=320, hop_length=160, window='hanning')    LSTM/lstm_decode_vb.py:37, CRN/crn_decode_vb.py:36,
//       GCRN/gcrn_decode_vb.py:37, DPCRN/dpcrn_decode_vb.py:37, G2Net_VB/com_decode.py:49
//   torch.istft / librosa.istft with Hann synthesis window, window-sum-square normalisation, `length=`.
// Both front ends are centre=True / reflect pad n_fft/2 / periodic Hann / one-sided.
//
// Kernel shape: one 64-lane wave transforms one frame with a Stockham autosort FFT in LDS (radix 4/4/4/4/2 for
// n_fft=512, 4/4/4/5 for n_fft=320 - the radix-5 and radix-2 tail stages need no twiddles), twiddles and the
// window staged once per block in LDS.  A block owns 16 consecutive frames, transformed in pairs as 8 complex FFTs
// (two-for-one real FFT), so that the [F][T]-major spectrogram is written / read in 64-byte runs along T and the
// waveform is read in coalesced rows.
#include "kernels.h"
#include "common.h"

namespace se {

static thread_local const Ragged* g_ragged = nullptr;
const Ragged* ragged_ctx() { return g_ragged; }
void set_ragged_ctx(const Ragged* r) { g_ragged = r; }

static thread_local StageProf* g_stage_prof = nullptr;
StageProf* stage_prof() { return g_stage_prof; }
void set_stage_prof(StageProf* p) { g_stage_prof = p; }
void StageProf::begin(int stage, hipStream_t st) {
    Slot& s = slot[stage];
    if (s.used + 2 > s.ev.size())
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e;
            SE_HIP(hipEventCreate(&e));
            s.ev.push_back(e);
        }
    SE_HIP(hipEventRecord(s.ev[s.used], st));
}
void StageProf::end(int stage, hipStream_t st, double bytes) {
    Slot& s = slot[stage];
    SE_HIP(hipEventRecord(s.ev[s.used + 1], st));
    s.used += 2;
    s.bytes += bytes;
    s.launches += 1;
}
void StageProf::reset() {
    for (auto& s : slot) {
        s.used = 0;
        s.bytes = 0.0;
        s.launches = 0;
    }
}
double StageProf::ms(int stage) {
    Slot& s = slot[stage];
    double tot = 0.0;
    for (size_t i = 0; i + 1 < s.used; i += 2) {
        SE_HIP(hipEventSynchronize(s.ev[i + 1]));
        float m = 0.f;
        SE_HIP(hipEventElapsedTime(&m, s.ev[i], s.ev[i + 1]));
        tot += m;
    }
    return tot;
}
StageProf::~StageProf() {
    for (auto& s : slot)
        for (auto e : s.ev) (void)hipEventDestroy(e);
}

constexpr int FPB = 16;  // frames per block
constexpr int PPB = 8;   // complex transforms per block: frames are transformed in pairs (two-for-one real FFT)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// one Stockham stage of radix R; s = product of earlier radices (power of two), ncur = N / s
template <int N, int R, bool INV>
__device__ __forceinline__ void fft_stage(const float2* __restrict__ x, float2* __restrict__ y,
                                          const float2* __restrict__ tw, int ncur, int log2s, int lane) {
    const int s = 1 << log2s;
    const int m = ncur / R;
    constexpr int NB = N / R;
    for (int i = lane; i < NB; i += 64) {
        const int p = i >> log2s, q = i & (s - 1);
        float2 in[R], out[R];
#pragma unroll
        for (int j = 0; j < R; ++j) in[j] = x[q + s * (p + m * j)];
        if (R == 2) {
            out[0] = make_float2(in[0].x + in[1].x, in[0].y + in[1].y);
            out[1] = make_float2(in[0].x - in[1].x, in[0].y - in[1].y);
        } else if (R == 4) {
            const float2 a = in[0], b = in[1], c = in[2], d = in[3];
            const float2 apc = make_float2(a.x + c.x, a.y + c.y), amc = make_float2(a.x - c.x, a.y - c.y);
            const float2 bpd = make_float2(b.x + d.x, b.y + d.y), bmd = make_float2(b.x - d.x, b.y - d.y);
            // forward: -i*(b-d) = (bmd.y, -bmd.x);  inverse: +i*(b-d) = (-bmd.y, bmd.x)
            const float2 jb = INV ? make_float2(-bmd.y, bmd.x) : make_float2(bmd.y, -bmd.x);
            out[0] = make_float2(apc.x + bpd.x, apc.y + bpd.y);
            out[1] = make_float2(amc.x + jb.x, amc.y + jb.y);
            out[2] = make_float2(apc.x - bpd.x, apc.y - bpd.y);
            out[3] = make_float2(amc.x - jb.x, amc.y - jb.y);
        } else {   // R == 5
            constexpr float c1 = 0.30901699437494742f, s1 = 0.95105651629515357f;    // cos/sin 2pi/5
            constexpr float c2 = -0.80901699437494742f, s2 = 0.58778525229247313f;   // cos/sin 4pi/5
            const float sg = INV ? 1.f : -1.f;
            const float2 t1 = make_float2(in[1].x + in[4].x, in[1].y + in[4].y);
            const float2 t2 = make_float2(in[2].x + in[3].x, in[2].y + in[3].y);
            const float2 t3 = make_float2(in[1].x - in[4].x, in[1].y - in[4].y);
            const float2 t4 = make_float2(in[2].x - in[3].x, in[2].y - in[3].y);
            out[0] = make_float2(in[0].x + t1.x + t2.x, in[0].y + t1.y + t2.y);
            const float2 m1 = make_float2(in[0].x + c1 * t1.x + c2 * t2.x, in[0].y + c1 * t1.y + c2 * t2.y);
            const float2 m2 = make_float2(in[0].x + c2 * t1.x + c1 * t2.x, in[0].y + c2 * t1.y + c1 * t2.y);
            // sg * i * (s1*t3 + s2*t4)  and  sg * i * (s2*t3 - s1*t4)
            const float2 u1 = make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
            const float2 u2 = make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
            const float2 j1 = make_float2(-sg * u1.y, sg * u1.x);
            const float2 j2 = make_float2(-sg * u2.y, sg * u2.x);
            out[1] = make_float2(m1.x + j1.x, m1.y + j1.y);
            out[4] = make_float2(m1.x - j1.x, m1.y - j1.y);
            out[2] = make_float2(m2.x + j2.x, m2.y + j2.y);
            out[3] = make_float2(m2.x - j2.x, m2.y - j2.y);
        }
        if (m > 1) {
            const int step = p * (N / ncur);
#pragma unroll
            for (int k = 1; k < R; ++k) {
                float2 w = tw[step * k];
                if (INV) w.y = -w.y;
                out[k] = cmul(out[k], w);
            }
        }
#pragma unroll
        for (int k = 0; k < R; ++k) y[q + s * (R * p + k)] = out[k];
    }
}

// Full transform of the frame held in buf0; returns the buffer holding the natural-order result.
// A frame pair belongs to ONE wave from its first stage to its last (the wave also filled buf0 in the forward kernel), so
// the stages hand over through LDS inside the wave: DS operations of a wave execute in order, a wave-level fence (no
// block barrier) is all the write -> read hand-over needs.  Ten block barriers per block made every wave wait for the
// slowest one at each of the five stages; the callers keep one block barrier before data crosses waves.
#define SE_WAVE_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
template <int N, bool INV>
__device__ __forceinline__ float2* fft_frame(float2* buf0, float2* buf1, const float2* tw, int lane) {
    SE_WAVE_FENCE();
    if (N == 512) {
        fft_stage<512, 4, INV>(buf0, buf1, tw, 512, 0, lane); SE_WAVE_FENCE();
        fft_stage<512, 4, INV>(buf1, buf0, tw, 128, 2, lane); SE_WAVE_FENCE();
        fft_stage<512, 4, INV>(buf0, buf1, tw, 32, 4, lane);  SE_WAVE_FENCE();
        fft_stage<512, 4, INV>(buf1, buf0, tw, 8, 6, lane);   SE_WAVE_FENCE();
        fft_stage<512, 2, INV>(buf0, buf1, tw, 2, 8, lane);   SE_WAVE_FENCE();
        return buf1;
    } else {   // 320
        fft_stage<320, 4, INV>(buf0, buf1, tw, 320, 0, lane); SE_WAVE_FENCE();
        fft_stage<320, 4, INV>(buf1, buf0, tw, 80, 2, lane);  SE_WAVE_FENCE();
        fft_stage<320, 4, INV>(buf0, buf1, tw, 20, 4, lane);  SE_WAVE_FENCE();
        fft_stage<320, 5, INV>(buf1, buf0, tw, 5, 6, lane);   SE_WAVE_FENCE();
        return buf0;
    }
}

template <int N>
__device__ __forceinline__ void init_tables(float2* tw, float* win, int win_len, int tid) {
    const int left = (N - win_len) / 2;
    for (int j = tid; j < N; j += 256) {
        double sv, cv;
        sincospi(2.0 * j / N, &sv, &cv);
        tw[j] = make_float2((float)cv, (float)(-sv));
        float w = 0.f;
        if (j >= left && j < left + win_len) w = (float)(0.5 - 0.5 * cospi(2.0 * (j - left) / win_len));
        win[j] = w;
    }
}

struct StftArgs {
    const float* wav; long pitch; int B, L, Lpad; const float* c_scale; float p_in;
    float* spec; float* mag; int T, Tp, hop, win;
    const int *len, *lpad, *tlen;      // ragged batch: per-row L, Lpad, T (else null)
    int t_first, col0;                 // frames [t_first, T) are transformed, frame t lands in column t - t_first + col0
};

// two-for-one: a block owns 16 consecutive frames as 8 complex transforms z = x_{2p} + i x_{2p+1}
//   X_{2p}[k] = (Z[k] + conj Z[N-k]) / 2,   X_{2p+1}[k] = (Z[k] - conj Z[N-k]) / (2i)
template <int N>
__global__ __launch_bounds__(256) void stft_kernel(const StftArgs a) {
    constexpr int F = N / 2 + 1;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float2* tw = reinterpret_cast<float2*>(smem_f);
    float2* bufs = tw + N;                                  // [PPB][2][N]
    float* win = reinterpret_cast<float*>(bufs + PPB * 2 * N);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, t0 = a.t_first + blockIdx.x * FPB;
    const int cshift = a.col0 - a.t_first;          // output column of frame t = t + cshift
    init_tables<N>(tw, win, a.win, tid);
    __syncthreads();
    const float c = a.c_scale ? a.c_scale[b] : 1.f;
    const float* x = a.wav + (long)b * a.pitch;
    // ragged batch: this row's own length, padded length and frame count; frames in [Tb, T) come out as zeros
    const int L = a.len ? a.len[b] : a.L, Lpad = a.lpad ? a.lpad[b] : a.Lpad, Tb = a.tlen ? a.tlen[b] : a.T;
    auto sample = [&](int t, int n) {
        float v = 0.f;
        if (t < Tb) {
            int idx = t * a.hop + n - N / 2;
            if (idx < 0) idx = -idx;
            if (idx >= Lpad) idx = 2 * (Lpad - 1) - idx;
            if (idx >= 0 && idx < L) v = x[idx] * c * win[n];
        }
        return v;
    };
    if (t0 >= Tb) {            // ragged batch: the block lies wholly in the row's zero tail (block-uniform)
        for (int idx = tid; idx < F * FPB; idx += 256) {
            const int t = t0 + (idx & (FPB - 1)), k = idx >> 4;
            if (t >= a.T) continue;
            if (a.spec) {
                a.spec[(((long)b * 2 + 0) * F + k) * a.Tp + t + cshift] = 0.f;
                a.spec[(((long)b * 2 + 1) * F + k) * a.Tp + t + cshift] = 0.f;
            }
            if (a.mag) a.mag[((long)b * F + k) * a.Tp + t + cshift] = 0.f;
        }
        return;
    }
    // A frame whose windowed samples are all exactly zero (digital silence) must transform to EXACT zeros: the decode
    // scripts take atan2 of the spectrum (np.angle(0) = 0) and the mapping models re-use that phase at full magnitude
    // (LSTM/lstm_decode_vb.py:47-49, CTSNet/two_stage_com_decode_vb.py:80-81).  The two-for-one split of a (silent,
    // non-silent) frame pair would leave rounding residue of the partner's spectrum - a random phase - in the silent one.
    __shared__ int nzflag[FPB];
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int pi = wave + 4 * rep, t = t0 + 2 * pi;
        float2* b0 = bufs + (pi * 2) * N;
        float2* b1 = b0 + N;
        bool nz0 = false, nz1 = false;
        for (int n = lane; n < N; n += 64) {
            const float2 v = make_float2(sample(t, n), sample(t + 1, n));
            nz0 |= (v.x != 0.f);
            nz1 |= (v.y != 0.f);
            b0[n] = v;
        }
        const bool a0 = __any(nz0), a1 = __any(nz1);
        if (lane == 0) {
            nzflag[2 * pi] = a0;
            nzflag[2 * pi + 1] = a1;
        }
        fft_frame<N, false>(b0, b1, tw, lane);
    }
    __syncthreads();        // the spectra of all eight frame pairs (and the silence flags) are read across waves below
    // write [F][T]-major: the 16 frames of one bin are 64 contiguous bytes
    for (int idx = tid; idx < F * FPB; idx += 256) {
        const int fi = idx & (FPB - 1), k = idx >> 4;
        const int t = t0 + fi;
        if (t >= a.T) continue;
        // natural-order result sits in buf1 after 5 stages (512) / buf0 after 4 stages (320)
        const float2* src = bufs + ((fi >> 1) * 2) * N + ((N == 512) ? N : 0);
        const float2 zk = src[k], zc = src[k == 0 ? 0 : N - k];       // Z[k], Z[(N - k) mod N]
        float2 v;
        if (fi & 1) v = make_float2(0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x));      // (Z[k] - conj Z[N-k]) / (2i)
        else v = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y));              // (Z[k] + conj Z[N-k]) / 2
        if (!nzflag[fi]) v = make_float2(0.f, 0.f);                                    // silent frame: exact zeros
        const float m = sqrtf(v.x * v.x + v.y * v.y);
        float mp = m;
        if (a.p_in != 1.f) {
            mp = powf(m, a.p_in);
            const float sc = m > 0.f ? mp / m : 0.f;
            v.x *= sc;
            v.y *= sc;
        }
        if (a.spec) {
            a.spec[(((long)b * 2 + 0) * F + k) * a.Tp + t + cshift] = v.x;
            a.spec[(((long)b * 2 + 1) * F + k) * a.Tp + t + cshift] = v.y;
        }
        if (a.mag) a.mag[((long)b * F + k) * a.Tp + t + cshift] = mp;
    }
}

struct IstftArgs {
    const float* spec; int B, T, Tp; int hop, win;
    const float* c_scale; float* out; long out_pitch; int Lout;
    int own, halo;        // overlap-add positions a block owns = own * hop; frames it transforms = FPB = own + halo
    const int *tlen, *olen;     // ragged batch: frames / output samples of row b (else null); samples in [olen, Lout) = 0
    // streaming window (offline: all zero): frame t sits in spec column t - t_off, frames [t_lo, T) exist, the launch emits
    // output samples [o_lo, Lout) into out[o - o_lo], block 0 starts at overlap-add position pos_base (a hop multiple)
    int t_off, t_lo, o_lo, pos_base;
};

// Inverse STFT with the overlap-add fused in: a block transforms FPB = 16 consecutive frames (two-for-one: Z = X_{2p} +
// i X_{2p+1} on the Hermitian-extended spectra -> z = x_{2p} + i x_{2p+1}), keeps the windowed frames in LDS and writes
// the `own * hop` output samples whose contributing frames all lie inside its window: the first `halo` (= ceil(N/hop) - 1,
// rounded up to even for the frame pairing) frames are recomputed by the neighbouring block instead of travelling through
// a [B][T][N] scratch tensor in HBM (round 1: frames written and re-read by a second kernel, 4.2x the algorithmic bytes).
// Divides by the overlap-added squared window and by the utterance's c.
template <int N>
__global__ __launch_bounds__(256) void istft_ola_kernel(const IstftArgs a) {
    constexpr int F = N / 2 + 1;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float2* tw = reinterpret_cast<float2*>(smem_f);
    float2* bufs = tw + N;
    float* win = reinterpret_cast<float*>(bufs + PPB * 2 * N);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int Tb = a.tlen ? a.tlen[b] : a.T;
    const int Lo = a.olen ? a.olen[b] : a.Lout;
    const int span = a.own * a.hop;
    const int pos0 = a.pos_base + blockIdx.x * span;        // first overlap-add position (pos = sample + N/2) of the block
    const int tb = pos0 / a.hop - a.halo;                   // first frame of the window (may be negative)
    float* outp = a.out + (long)b * a.out_pitch - a.o_lo;
    // nothing of this row left (ragged batch, or the rounding of the grid): zeros up to Lout (block-uniform branch)
    if (pos0 - N / 2 >= Lo || tb >= Tb) {
        for (int i = tid; i < span; i += 256) {
            const int o = pos0 + i - N / 2;
            if (o >= a.o_lo && o < a.Lout) outp[o] = 0.f;
        }
        return;
    }
    init_tables<N>(tw, win, a.win, tid);
    // 8 lanes x 2 frames = 16 consecutive frames of one bin (64 contiguous bytes per plane)
    for (int idx = tid; idx < F * PPB; idx += 256) {
        const int pi = idx & (PPB - 1), k = idx >> 3;
        const int t = tb + 2 * pi;
        float2 xa = make_float2(0.f, 0.f), xb = make_float2(0.f, 0.f);
        const float* re = a.spec + (((long)b * 2 + 0) * F + k) * a.Tp - a.t_off;
        const float* im = a.spec + (((long)b * 2 + 1) * F + k) * a.Tp - a.t_off;
        if (t >= a.t_lo && t < Tb) xa = make_float2(re[t], im[t]);
        if (t + 1 >= a.t_lo && t + 1 < Tb) xb = make_float2(re[t + 1], im[t + 1]);
        float2* b0 = bufs + (pi * 2) * N;
        if (k == 0 || k == N / 2) {
            b0[k] = make_float2(xa.x, xb.x);                // C2R ignores the imaginary part of DC / Nyquist
        } else {
            b0[k] = make_float2(xa.x - xb.y, xa.y + xb.x);              // X_a[k] + i X_b[k]
            b0[N - k] = make_float2(xa.x + xb.y, xb.x - xa.y);          // conj X_a[k] + i conj X_b[k]
        }
    }
    __syncthreads();
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int pi = wave + 4 * rep;
        float2* b0 = bufs + (pi * 2) * N;
        fft_frame<N, true>(b0, b0 + N, tw, lane);
    }
    __syncthreads();        // the overlap-add reads every wave's frames
    // overlap-add out of LDS: position pos gets frame t = tb + fi at n = pos - t * hop for every frame that covers it
    const float invN = 1.f / N;
    const float cinv = a.c_scale ? 1.f / a.c_scale[b] : 1.f;
    for (int i = tid; i < span; i += 256) {
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
            const int n = pos - t * a.hop, fi = t - tb;          // fi >= 0 by construction of halo
            const float2 z = (bufs + ((fi >> 1) * 2) * N + ((N == 512) ? N : 0))[n];
            const float w = win[n];
            acc += ((fi & 1) ? z.y : z.x) * invN * w;
            env += w * w;
        }
        // summed newest frame first above; the reference order (oldest first) differs only in fp32 rounding
        float y = env > 1e-11f ? acc / env : acc;
        outp[o] = y * cinv;
    }
}

// c[b] = sqrt(L / sum x^2) in two steps: RMS_SPLIT blocks per utterance each sum a slice in fp64 (one block per utterance
// pulled 256 kB through a single CU: 100 us flat whatever the batch), then one thread per utterance adds the slices in
// slice order - deterministic, no atomics.
constexpr int RMS_SPLIT = 16;
__global__ __launch_bounds__(256) void rms_partial_kernel(const float* __restrict__ wav, int L, long pitch,
                                                          double* __restrict__ part, const int* __restrict__ len) {
    const int b = blockIdx.y, k = blockIdx.x;
    if (len) L = len[b];
    const int per = (L + RMS_SPLIT - 1) / RMS_SPLIT;
    const int lo = k * per, hi = min(L, lo + per);
    const float* x = wav + (long)b * pitch;
    double s = 0.0;
    int i = lo + threadIdx.x;
    for (; i + 3 * 256 < hi; i += 4 * 256) {          // four independent loads in flight per thread
        const float v0 = x[i], v1 = x[i + 256], v2 = x[i + 512], v3 = x[i + 768];
        s += (double)v0 * v0;
        s += (double)v1 * v1;
        s += (double)v2 * v2;
        s += (double)v3 * v3;
    }
    for (; i < hi; i += 256) {
        const double v = x[i];
        s += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    __shared__ double sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[(long)b * RMS_SPLIT + k] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void rms_finish_kernel(const double* __restrict__ part, int L, float* __restrict__ c_out,
                                  const int* __restrict__ len, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (len) L = len[b];
    double tot = 0.0;
    for (int k = 0; k < RMS_SPLIT; ++k) tot += part[(long)b * RMS_SPLIT + k];
    c_out[b] = (float)sqrt((double)L / tot);
}

void launch_rms_scale(const float* wav, int B, int L, long pitch, float* c_out, hipStream_t s) {
    const Ragged* rg = ragged_ctx();
    StageScope prof(STAGE_RMS, s, 4.0 * L * B);
    double* part = reinterpret_cast<double*>(device_scratch(3, (size_t)B * RMS_SPLIT * sizeof(double), s));
    hipLaunchKernelGGL(rms_partial_kernel, dim3(RMS_SPLIT, B), dim3(256), 0, s, wav, L, pitch, part, rg ? rg->len : nullptr);
    hipLaunchKernelGGL(rms_finish_kernel, dim3((B + 63) / 64), dim3(64), 0, s, part, L, c_out, rg ? rg->len : nullptr, B);
    SE_HIP(hipGetLastError());
}

template <int N>
static size_t fft_lds_bytes() { return (size_t)N * 8 + (size_t)PPB * 2 * N * 8 + (size_t)N * 4; }

template <typename K>
static void set_lds_attr(K kernel, size_t bytes) {
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes));
}

void launch_stft(const StftGeom& g, const float* wav, long pitch, int B, int L, int Lpad, const float* c_scale,
                 float p_in, float* spec_ri, float* mag, int T, int Tp, hipStream_t s, int t_first, int col0) {
    const Ragged* rg = ragged_ctx();
    SE_CHECK(t_first >= 0 && t_first < T, "launch_stft: empty frame range");
    StageScope prof(STAGE_STFT, s, 4.0 * L * B + (spec_ri ? 8.0 : 0.0) * g.F() * T * B + (mag ? 4.0 : 0.0) * g.F() * T * B);
    StftArgs a{wav, pitch, B, L, Lpad, c_scale, p_in, spec_ri, mag, T, Tp, g.hop, g.win,
               rg ? rg->len : nullptr, rg ? rg->lpad : nullptr, rg ? rg->tlen : nullptr, t_first, col0};
    if (stft2_enabled() && (g.n_fft == 512 || g.n_fft == 320)) {       // register-resident FFT, 128 B runs (k_stft2.hip)
        launch_stft2(g, wav, pitch, B, L, Lpad, c_scale, p_in, spec_ri, mag, T, Tp, s, t_first, col0);
        return;
    }
    dim3 grid((T - t_first + FPB - 1) / FPB, B);
    if (g.n_fft == 512) {
        static bool seen[64] = {};
        if (first_on_device(seen)) set_lds_attr(stft_kernel<512>, fft_lds_bytes<512>());
        hipLaunchKernelGGL(stft_kernel<512>, grid, dim3(256), fft_lds_bytes<512>(), s, a);
    } else if (g.n_fft == 320) {
        static bool seen[64] = {};
        if (first_on_device(seen)) set_lds_attr(stft_kernel<320>, fft_lds_bytes<320>());
        hipLaunchKernelGGL(stft_kernel<320>, grid, dim3(256), fft_lds_bytes<320>(), s, a);
    } else {
        SE_CHECK(false, "unsupported n_fft (320 and 512 are the reference geometries)");
    }
    SE_HIP(hipGetLastError());
}

void launch_istft(const StftGeom& g, const float* spec_ri, int B, int T, int Tp, float* /*frames: unused since the fused kernel*/,
                  const float* c_scale, float* wav_out, long out_pitch, int Lout, hipStream_t s, int t_off, int t_lo, int o_lo,
                  const float* frame_inv, int ring) {
    const Ragged* rg = ragged_ctx();
    StageScope prof(STAGE_ISTFT, s, 8.0 * g.F() * T * B + 4.0 * Lout * B);
    SE_CHECK(Lout > o_lo, "launch_istft: empty output range");
    if (stft2_enabled() && (g.n_fft == 512 || g.n_fft == 320)) {
        launch_istft2(g, spec_ri, B, T, Tp, c_scale, wav_out, out_pitch, Lout, s, t_off, t_lo, o_lo, frame_inv, ring);
        return;
    }
    SE_CHECK(!frame_inv, "per-frame scales (running-RMS streams) need the round-3 iSTFT kernel (SE_STFT_V1 unset, n_fft 320 / 512)");
    int halo = (g.n_fft + g.hop - 1) / g.hop - 1;
    halo += halo & 1;                                       // frames are transformed in pairs
    SE_CHECK(halo < FPB, "hop too small for the fused overlap-add window");
    const int own = FPB - halo;
    IstftArgs a{spec_ri, B, T, Tp, g.hop, g.win, c_scale, wav_out, out_pitch, Lout, own, halo,
                rg ? rg->tlen : nullptr, rg ? rg->olen : nullptr, t_off, t_lo, o_lo, 0};
    const int span = own * g.hop;
    // first block: the hop-aligned position at or below the first emitted sample
    a.pos_base = (o_lo + g.n_fft / 2) / g.hop * g.hop;
    SE_CHECK(Lout > o_lo, "launch_istft: empty output range");
    dim3 grid((g.n_fft / 2 + Lout - a.pos_base + span - 1) / span, B);
    if (g.n_fft == 512) {
        static bool seen[64] = {};
        if (first_on_device(seen)) set_lds_attr(istft_ola_kernel<512>, fft_lds_bytes<512>());
        hipLaunchKernelGGL(istft_ola_kernel<512>, grid, dim3(256), fft_lds_bytes<512>(), s, a);
    } else if (g.n_fft == 320) {
        static bool seen[64] = {};
        if (first_on_device(seen)) set_lds_attr(istft_ola_kernel<320>, fft_lds_bytes<320>());
        hipLaunchKernelGGL(istft_ola_kernel<320>, grid, dim3(256), fft_lds_bytes<320>(), s, a);
    } else {
        SE_CHECK(false, "unsupported n_fft");
    }
    SE_HIP(hipGetLastError());
}

}  // namespace se
