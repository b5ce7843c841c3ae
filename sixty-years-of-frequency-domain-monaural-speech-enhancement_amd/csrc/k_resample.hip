// Band-limited sinc resampler (48 kHz VoiceBank+DEMAND clips -> the models' 16 kHz): the GPU side of
// `librosa.resample(y, orig_sr, 16000, fix=True, scale=False)` at DCCRN/dccrn_decode_vb.py:26, LSTM/lstm_decode_vb.py:34.
//
// librosa / resampy are third-party, absent and unversioned in the reference: this follows the
// published resampy 'kaiser_best' algorithm - Kaiser-windowed sinc table (64 zero crossings x 512 samples), linear
// interpolation between table entries, float64 accumulation - restated in oracle/resample.py; parity is pinned to
// that restatement only.  One thread per output sample; the table (256 KB of doubles) lives in L2.
#include "kernels.h"
#include "common.h"
#include <cmath>
#include <mutex>
#include <vector>

namespace se {

namespace {

constexpr int RS_ZEROS = 64, RS_BITS = 512, RS_NWIN = RS_ZEROS * RS_BITS + 1;
constexpr double RS_ROLLOFF = 0.9475937167399596, RS_BETA = 14.769656459379492;

double bessel_i0(double x) {
    double s = 1.0, term = 1.0;
    const double q = 0.25 * x * x;
    for (int k = 1; k < 200; ++k) {
        term *= q / ((double)k * k);
        s += term;
        if (term < 1e-18 * s) break;
    }
    return s;
}

// right half of the filter, as resampy.filters.sinc_window builds it
std::vector<double> build_window() {
    const int n = RS_ZEROS * RS_BITS;
    std::vector<double> w(n + 1);
    const double i0b = bessel_i0(RS_BETA);
    const double pi = 3.14159265358979323846;
    for (int k = 0; k <= n; ++k) {
        const double t = (double)k * RS_ZEROS / n;                  // np.linspace(0, num_zeros, n + 1)
        const double a = RS_ROLLOFF * t;
        const double sinc = a == 0.0 ? 1.0 : std::sin(pi * a) / (pi * a);
        const double r = (double)k / n;                             // kaiser(2n+1)[n + k]: position k of n from the centre
        const double taper = bessel_i0(RS_BETA * std::sqrt(std::max(0.0, 1.0 - r * r))) / i0b;
        w[k] = taper * RS_ROLLOFF * sinc;
    }
    return w;
}

struct ResampleArgs {
    const float* x; long in_pitch; int n_in;
    float* y; long out_pitch; int n_calc, n_out;
    const double* win;       // [RS_NWIN] (already scaled by the ratio when downsampling)
    const double* treg;      // [n_calc] read positions, or nullptr -> t * inc exactly (integer decimation)
    double inc, scale;
    int index_step;
};

__global__ __launch_bounds__(256) void resample_kernel(const ResampleArgs a) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= a.n_out) return;
    float* yp = a.y + (long)blockIdx.y * a.out_pitch;
    if (t >= a.n_calc) {                                            // librosa fix_length: zero tail up to ceil(n * ratio)
        yp[t] = 0.f;
        return;
    }
    const float* __restrict__ x = a.x + (long)blockIdx.y * a.in_pitch;
    const double* __restrict__ win = a.win;
    const double tr = a.treg ? a.treg[t] : (double)t * a.inc;
    const int n = (int)tr;
    double acc = 0.0;
    {
        const double frac = a.scale * (tr - n);
        const double index_frac = frac * RS_BITS;
        const int offset = (int)index_frac;
        const double eta = index_frac - offset;
        const int i_max = min(n + 1, (RS_NWIN - offset) / a.index_step);
        for (int i = 0; i < i_max; ++i) {
            const int idx = offset + i * a.index_step;
            const double d = idx + 1 < RS_NWIN ? win[idx + 1] - win[idx] : 0.0;
            acc += (win[idx] + eta * d) * (double)x[n - i];
        }
        const double frac2 = a.scale - frac;
        const double index_frac2 = frac2 * RS_BITS;
        const int offset2 = (int)index_frac2;
        const double eta2 = index_frac2 - offset2;
        const int k_max = min(a.n_in - n - 1, (RS_NWIN - offset2) / a.index_step);
        for (int k = 0; k < k_max; ++k) {
            const int idx = offset2 + k * a.index_step;
            const double d = idx + 1 < RS_NWIN ? win[idx + 1] - win[idx] : 0.0;
            acc += (win[idx] + eta2 * d) * (double)x[n + k + 1];
        }
    }
    yp[t] = (float)acc;
}

struct Tables {
    std::mutex mu;
    std::vector<double> host;
    double* dev_unit = nullptr;      // table as built (upsampling)
    double* dev_scaled = nullptr;    // table * ratio of the last downsampling ratio seen
    double scaled_ratio = 0.0;
    double* dev_treg = nullptr;
    size_t treg_cap = 0;
};
Tables& tables() {
    static Tables t[64];              // one set of tables per device
    int dev = 0;
    SE_HIP(hipGetDevice(&dev));
    SE_CHECK(dev >= 0 && dev < 64, "device ordinal");
    return t[dev];
}

}  // namespace

// ---- PCM_16 <-> float at the two ends of the decode driver (e.g. DCCRN/dccrn_decode_vb.py:25,64).  soundfile.read:
// int16 / 32768 -> float (libsndfile s2f, 1 / 0x8000).  soundfile.write, default subtype of .wav = PCM_16: libsndfile's
// float -> short conversion with normalisation scales by 0x7FFF (NOT 0x8000) and rounds to nearest even (lrint); it does not
// clip unless SFC_SET_CLIPPING is on - a sample beyond +-1 wraps there.  The engine takes the 0x7FFF scale and CLIPS instead of
// wrapping (deliberate: a wrapped sample is a full-scale click).  soundfile is absent from this image: restated from
// libsndfile's pcm.c as published, unpinned.  x / 32768 is exact in fp32; y * 32767 is formed in fp64 so that the
// device and the host path (wavio.pcm16_bytes) agree bit for bit, while the host only moves raw 2-byte samples.
namespace {
__global__ __launch_bounds__(256) void pcm16_decode_kernel(const short* __restrict__ in, long in_pitch, float* __restrict__ out,
                                                           long out_pitch, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[(long)blockIdx.y * out_pitch + i] = (float)in[(long)blockIdx.y * in_pitch + i] * (1.f / 32768.f);
}
__global__ __launch_bounds__(256) void pcm16_encode_kernel(const float* __restrict__ in, long in_pitch, short* __restrict__ out,
                                                           long out_pitch, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = rint((double)in[(long)blockIdx.y * in_pitch + i] * 32767.0);   // round half to even, like lrint / np.rint
    out[(long)blockIdx.y * out_pitch + i] = (short)fmin(fmax(v, -32768.0), 32767.0);
}
}  // namespace
void launch_pcm16_decode(const short* in, long in_pitch, int batch, int n, float* out, long out_pitch, hipStream_t s) {
    SE_CHECK(batch > 0 && n > 0, "pcm16 decode: bad arguments");
    hipLaunchKernelGGL(pcm16_decode_kernel, dim3((n + 255) / 256, batch), dim3(256), 0, s, in, in_pitch, out, out_pitch, n);
    SE_HIP(hipGetLastError());
}
void launch_pcm16_encode(const float* in, long in_pitch, int batch, int n, short* out, long out_pitch, hipStream_t s) {
    SE_CHECK(batch > 0 && n > 0, "pcm16 encode: bad arguments");
    hipLaunchKernelGGL(pcm16_encode_kernel, dim3((n + 255) / 256, batch), dim3(256), 0, s, in, in_pitch, out, out_pitch, n);
    SE_HIP(hipGetLastError());
}

long resample_out_samples(int n_in, int sr_in, int sr_out) {
    if (sr_in == sr_out) return n_in;
    return (long)std::ceil((double)n_in * ((double)sr_out / sr_in));
}

void launch_resample(const float* x, long in_pitch, int batch, int n_in, int sr_in, int sr_out, float* y, long out_pitch,
                     hipStream_t s) {
    SE_CHECK(sr_in > 0 && sr_out > 0 && n_in > 0 && batch > 0, "resample: bad arguments");
    const double ratio = (double)sr_out / sr_in;
    Tables& T = tables();
    if (T.host.empty()) {
        T.host = build_window();
        SE_HIP(hipMalloc(&T.dev_unit, RS_NWIN * sizeof(double)));
        SE_HIP(hipMalloc(&T.dev_scaled, RS_NWIN * sizeof(double)));
        SE_HIP(hipMemcpy(T.dev_unit, T.host.data(), RS_NWIN * sizeof(double), hipMemcpyHostToDevice));
    }
    ResampleArgs a{};
    a.x = x; a.in_pitch = in_pitch; a.n_in = n_in; a.y = y; a.out_pitch = out_pitch;
    a.n_calc = (int)((double)n_in * ratio);                         // resampy: int(shape * sample_ratio)
    a.n_out = (int)resample_out_samples(n_in, sr_in, sr_out);
    a.scale = std::min(1.0, ratio);
    a.index_step = (int)(a.scale * RS_BITS);
    a.inc = 1.0 / ratio;
    if (ratio < 1.0) {
        if (T.scaled_ratio != ratio) {                              // interp_win *= sample_ratio
            std::vector<double> w(T.host);
            for (auto& v : w) v *= ratio;
            SE_HIP(hipStreamSynchronize(s));
            SE_HIP(hipMemcpy(T.dev_scaled, w.data(), RS_NWIN * sizeof(double), hipMemcpyHostToDevice));
            T.scaled_ratio = ratio;
        }
        a.win = T.dev_scaled;
    } else {
        a.win = T.dev_unit;
    }
    if (sr_in % sr_out == 0) {
        a.treg = nullptr;                                           // t * inc is exact: same values as the running sum
    } else {
        // resampy advances the read position by repeated float64 addition: reproduce that rounding on the host
        std::vector<double> tr((size_t)a.n_calc);
        double acc = 0.0;
        for (int i = 0; i < a.n_calc; ++i) {
            tr[i] = acc;
            acc += a.inc;
        }
        if (tr.size() > T.treg_cap) {
            if (T.dev_treg) (void)hipFree(T.dev_treg);
            SE_HIP(hipMalloc(&T.dev_treg, tr.size() * sizeof(double)));
            T.treg_cap = tr.size();
        }
        SE_HIP(hipStreamSynchronize(s));
        SE_HIP(hipMemcpy(T.dev_treg, tr.data(), tr.size() * sizeof(double), hipMemcpyHostToDevice));
        a.treg = T.dev_treg;
    }
    hipLaunchKernelGGL(resample_kernel, dim3((a.n_out + 255) / 256, batch), dim3(256), 0, s, a);
    SE_HIP(hipGetLastError());
}

}  // namespace se
