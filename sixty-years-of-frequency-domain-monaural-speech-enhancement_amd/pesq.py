"""Wide-band PESQ (ITU-T P.862 + P.862.2 mapping), host side, numpy / scipy - SURVEY 8(f) rank 2.

Restates `DeepXi/deepxi/pesq.m` (Hu / Wojcicki / Loizou's MATLAB implementation of the ITU reference code, the only PESQ
the reference ships) function by function for the 16 kHz / wide-band branch the paper's tables use: level alignment
(`fix_power_level`), the P.862.2 input filter (`apply_filters_WB`), the alignment chain (`input_filter`, `apply_VAD`,
`crude_align`, `utterance_locate` = `id_searchwindows` + per-utterance `crude_align` / `time_align` + `id_utterances` +
`utterance_split` / `split_align`), the psychoacoustic model (`pesq_psychoacoustic_model` with Bark warping, intensity
warping, asymmetry, bad-interval realignment, `Lpq_weight`) and the MOS-LQO mapping.  The band tables of `setup_global`
(pesq.m:1863-2060) are the constants of the Recommendation.

Arrays keep MATLAB's 1-based indices (element 0 is a dummy), so every index expression reads like the line it restates.

PINNING: the reference holds no test vector for pesq.m and there is no MATLAB / Octave in the build container, so this
restatement is UNPINNED against the reference; tests/test_metrics.py checks the known anchors of the measure (identical
signals -> raw 4.5 / MOS-LQO 4.64, monotone in SNR, insensitive to level and to a constant delay) and that the engine's
and the oracle's outputs score within 0.01.
"""
import numpy as np

FS = 16000
DOWNSAMPLE = 64
ALIGN_NFFT = 1024
DATAPADDING_MSECS = 320
SEARCHBUFFER = 75
MINSPEECHLGTH = 4
JOINSPEECHLGTH = 50
MAXNUTTERANCES = 50
MINUTTLENGTH = 50
WHOLE_SIGNAL = -1
NB = 49
SP = 6.910853e-006
SL = 1.866055e-001
PAD = DATAPADDING_MSECS * (FS // 1000)          # 5120 samples
SB = SEARCHBUFFER * DOWNSAMPLE                  # 4800 samples

WB_SOS = np.array([[2.740826, -5.4816519, 2.740826, 1.0, -1.9444777, 0.94597794]])
IIR_SOS_16K = np.array([
    [0.325631521, -0.086782860, -0.238848661, -1.079416490, 0.434583902],
    [0.403961804, -0.556985881, 0.153024077, -0.415115835, 0.696590244],
    [4.736162769, 3.287251046, 1.753289019, -1.859599046, 0.876284034],
    [0.365373469, 0.000000000, 0.000000000, -0.634626531, 0.000000000],
    [0.884811506, 0.000000000, 0.000000000, -0.256725271, 0.141536777],
    [0.723593055, -1.447186099, 0.723593044, -1.129587469, 0.657232737],
    [1.644910855, -1.817280902, 1.249658063, -1.778403899, 0.801724355],
    [0.633692689, -0.284644314, -0.319789663, 0.000000000, 0.000000000],
    [1.032763031, 0.268428979, 0.602913323, 0.000000000, 0.000000000],
    [1.001616361, -0.823749013, 0.439731942, -0.885778255, 0.000000000],
    [0.752472096, -0.375388990, 0.188977609, -0.077258216, 0.247230734],
    [1.023700575, 0.001661628, 0.521284240, -0.183867259, 0.354324187]])

NR_HZ_PER_BARK = np.array([1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 2, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 4, 3, 4, 5, 4, 5, 6, 6, 7,
                           8, 9, 9, 12, 12, 15, 16, 18, 21, 25, 20])
CENTRE_BARK = np.array([
    0.078672, 0.316341, 0.636559, 0.961246, 1.290450, 1.624217, 1.962597, 2.305636, 2.653383, 3.005889, 3.363201, 3.725371,
    4.092449, 4.464486, 4.841533, 5.223642, 5.610866, 6.003256, 6.400869, 6.803755, 7.211971, 7.625571, 8.044611, 8.469146,
    8.899232, 9.334927, 9.776288, 10.223374, 10.676242, 11.134952, 11.599563, 12.070135, 12.546731, 13.029408, 13.518232,
    14.013264, 14.514566, 15.022202, 15.536238, 16.056736, 16.583761, 17.117382, 17.657663, 18.204674, 18.758478, 19.319147,
    19.886751, 20.461355, 21.043034])
WIDTH_BARK = np.array([
    0.157344, 0.317994, 0.322441, 0.326934, 0.331474, 0.336061, 0.340697, 0.345381, 0.350114, 0.354897, 0.359729, 0.364611,
    0.369544, 0.374529, 0.379565, 0.384653, 0.389794, 0.394989, 0.400236, 0.405538, 0.410894, 0.416306, 0.421773, 0.427297,
    0.432877, 0.438514, 0.444209, 0.449962, 0.455774, 0.461645, 0.467577, 0.473569, 0.479621, 0.485736, 0.491912, 0.498151,
    0.504454, 0.510819, 0.517250, 0.523745, 0.530308, 0.536934, 0.543629, 0.550390, 0.557220, 0.564119, 0.571085, 0.578125,
    0.585232])
POW_CORR = np.array([
    100.000000, 99.999992, 100.000000, 100.000008, 100.000008, 100.000015, 99.999992, 99.999969, 50.000027, 100.000000, 99.999969,
    100.000015, 99.999947, 100.000061, 53.047077, 110.000046, 117.991989, 65.000000, 68.760147, 69.999931, 71.428818, 75.000038,
    76.843384, 80.968781, 88.646126, 63.864388, 68.155350, 72.547775, 75.584831, 58.379192, 80.950836, 64.135651, 54.384785,
    73.821884, 64.437073, 59.176456, 65.521278, 61.399822, 58.144047, 57.004543, 64.126297, 54.311001, 61.114979, 55.077751,
    56.849335, 55.628868, 53.137054, 54.985844, 79.546974])
ABS_THRESH = np.array([
    51286152.00, 2454709.500, 70794.593750, 4897.788574, 1174.897705, 389.045166, 104.712860, 45.708820, 17.782795, 9.772372,
    4.897789, 3.090296, 1.905461, 1.258925, 0.977237, 0.724436, 0.562341, 0.457088, 0.389045, 0.331131, 0.295121, 0.269153,
    0.257040, 0.251189, 0.251189, 0.251189, 0.251189, 0.263027, 0.288403, 0.309030, 0.338844, 0.371535, 0.398107, 0.436516,
    0.467735, 0.489779, 0.501187, 0.501187, 0.512861, 0.524807, 0.524807, 0.524807, 0.512861, 0.478630, 0.426580, 0.371535,
    0.363078, 0.416869, 0.537032])
_BAND_EDGES = np.concatenate([[0], np.cumsum(NR_HZ_PER_BARK)])       # Hz-bin ranges of the Bark bands (0-based bins)
_WINDOW = 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(ALIGN_NFFT) / ALIGN_NFFT))


def _one(x):
    """1-based view: prepend a dummy element."""
    return np.concatenate([[0.0], np.asarray(x, dtype=np.float64)])


def _sosfilt(sos5, x):
    """dfilt.df2sos over rows [b0 b1 b2 a1 a2] (a0 = 1), zero initial state (pesq.m:265-306)."""
    from scipy.signal import sosfilt
    sos = np.zeros((len(sos5), 6))
    sos[:, :3] = sos5[:, :3]
    sos[:, 3] = 1.0
    sos[:, 4:] = sos5[:, 3:5]
    return sosfilt(sos, x)


class _State:
    """The `global` variables of pesq.m (utterance bookkeeping), 1-based arrays."""

    def __init__(self):
        n = MAXNUTTERANCES + 2
        self.UttSearch_Start = np.zeros(n, dtype=np.int64)
        self.UttSearch_End = np.zeros(n, dtype=np.int64)
        self.Utt_DelayEst = np.zeros(n, dtype=np.int64)
        self.Utt_Delay = np.zeros(n, dtype=np.int64)
        self.Utt_DelayConf = np.zeros(n)
        self.Utt_Start = np.zeros(n, dtype=np.int64)
        self.Utt_End = np.zeros(n, dtype=np.int64)
        self.Nutterances = 0
        self.Crude_DelayEst = 0
        self.Best = None


# ------------------------------------------------------------------------------------------------ level / filters
def _apply_filter(data, nsamples, table):
    """pesq.m:226-257: zero-phase FFT filter of the signal part, gain normalised at 1 kHz."""
    n = nsamples - 2 * SB + PAD
    p2 = int(2 ** np.ceil(np.log2(n)))
    f_t, db_t = table[:, 0], table[:, 1]
    gain_1k = np.interp(1000.0, f_t, db_t)
    x = np.zeros(p2)
    x[:n] = data[SB + 1: SB + n + 1]
    X = np.fft.rfft(x)
    fdb = np.interp(np.arange(p2 // 2 + 1) * (FS / p2), f_t, db_t) - gain_1k
    y = np.fft.irfft(X * 10.0 ** (fdb / 20.0), p2)
    out = data.copy()
    out[SB + 1: SB + n + 1] = y[:n]
    return out


_ALIGN_FILTER_DB = np.array([[0, -500], [50, -500], [100, -500], [125, -500], [160, -500], [200, -500], [250, -500], [300, -500],
                             [350, 0], [400, 0], [500, 0], [600, 0], [630, 0], [800, 0], [1000, 0], [1250, 0], [1600, 0], [2000, 0],
                             [2500, 0], [3000, 0], [3250, 0], [3500, -500], [4000, -500], [5000, -500], [6300, -500], [8000, -500]],
                            dtype=np.float64)


def _pow_of(data, a, b, div):
    return float(np.sum(data[a: b + 1] ** 2) / div)


def _fix_power_level(data, nsamples, max_nsamples):
    """pesq.m:609-630."""
    filt = _apply_filter(data, nsamples, _ALIGN_FILTER_DB)
    p = _pow_of(filt, SB + 1, nsamples - SB + PAD, max_nsamples - 2 * SB + PAD)
    return data * np.sqrt(1e7 / p)


def _dc_block(data, nsamples):
    """pesq.m:550-568."""
    out = data.copy()
    facc = np.sum(data[SB + 1: nsamples - SB + 1]) / nsamples
    out[SB + 1: nsamples - SB + 1] -= facc
    ramp = (0.5 + np.arange(DOWNSAMPLE)) / DOWNSAMPLE
    out[SB + 1: SB + DOWNSAMPLE + 1] *= ramp
    idx = np.arange(nsamples - SB, nsamples - SB - DOWNSAMPLE, -1)
    out[idx] *= ramp
    return out


# ------------------------------------------------------------------------------------------------ VAD
def _apply_vad(data, nsamples):
    """pesq.m:308-466 -> (VAD, logVAD), 1-based, length Nwindows."""
    nw = nsamples // DOWNSAMPLE
    V = np.zeros(nw + 1)
    V[1:] = np.sum(data[1: nw * DOWNSAMPLE + 1].reshape(nw, DOWNSAMPLE) ** 2, axis=1) / DOWNSAMPLE
    thresh = np.sum(V[1:]) / nw
    lmin = np.max(V[1:])
    lmin = lmin * 1.0e-4 if lmin > 0 else 1.0
    V[1:][V[1:] < lmin] = lmin
    for _ in range(12):
        low = V[1:][V[1:] <= thresh]
        noise, std = 0.0, 0.0
        if len(low) > 0:
            noise = np.sum(low) / len(low)
            std = np.sqrt(np.sum((low - noise) ** 2) / len(low))
        thresh = 1.001 * (noise + 2 * std)
    hi = V[1:][V[1:] > thresh]
    n_hi = len(hi)
    lsig = np.sum(hi)
    lnoise = np.sum(V[1:][V[1:] <= thresh])
    if n_hi > 0:
        lsig /= n_hi
    else:
        thresh = -1
    lnoise = lnoise / (nw - n_hi) if n_hi < nw else 1.0
    m = V[1:] <= thresh
    V[1:][m] = -V[1:][m]
    V[1] = -lmin
    V[nw] = -lmin
    start = finish = 0
    for c in range(2, nw + 1):
        if V[c] > 0.0 and V[c - 1] <= 0.0:
            start = c
        if V[c] <= 0.0 and V[c - 1] > 0.0:
            finish = c
            if finish - start <= MINSPEECHLGTH:
                V[start: finish] = -V[start: finish]
    if lsig >= lnoise * 1000:
        for c in range(2, nw + 1):
            if V[c] > 0 and V[c - 1] <= 0:
                start = c
            if V[c] <= 0 and V[c - 1] > 0:
                finish = c
                g = np.sum(V[start: finish])
                if g < 3.0 * thresh * (finish - start):
                    V[start: finish] = -V[start: finish]
    start = finish = 0
    for c in range(2, nw + 1):
        if V[c] > 0.0 and V[c - 1] <= 0.0:
            start = c
            if finish > 0 and (start - finish) <= JOINSPEECHLGTH:
                V[finish: start] = lmin
        if V[c] <= 0.0 and V[c - 1] > 0.0:
            finish = c
    start = 0
    for c in range(2, nw + 1):
        if V[c] > 0 and V[c - 1] <= 0:
            start = c
    if start == 0:
        V[1:] = np.abs(V[1:])
        V[1] = -lmin
        V[nw] = -lmin
    c = 4
    while c < nw - 1:
        if V[c] > 0 and V[c - 2] <= 0:
            V[c - 2] = V[c] * 0.1
            V[c - 1] = V[c] * 0.3
            c += 1
        if V[c] <= 0 and V[c - 1] > 0:
            V[c] = V[c - 1] * 0.3
            V[c + 1] = V[c - 1] * 0.1
            c += 3
        c += 1
    V[1:][V[1:] < 0] = 0
    if thresh <= 0:
        thresh = lmin
    logV = np.zeros(nw + 1)
    m = V[1:] > thresh
    logV[1:][m] = np.log(V[1:][m] / thresh)
    return V, logV


# ------------------------------------------------------------------------------------------------ alignment
def _fftnxcorr(ref_v, startr, nr, deg_v, startd, nd):
    """pesq.m:570-607 (1-based result Y(1..nr+nd-1))."""
    nx = int(2 ** np.ceil(np.log2(max(nr, nd))))
    startd, startr = max(1, startd), max(1, startr)
    x1 = np.zeros(2 * nx)
    x2 = np.zeros(2 * nx)
    x1[:nr] = ref_v[startr: startr + nr][::-1]
    x2[:nd] = deg_v[startd: startd + nd]
    y = np.fft.irfft(np.fft.rfft(x1) * np.fft.rfft(x2), 2 * nx)
    return _one(y[: nr + nd - 1])


def _crude_align(S, ref_log, ref_n, deg_log, deg_n, utt_id):
    """pesq.m:468-548."""
    if utt_id == WHOLE_SIGNAL:
        nr, nd, startr, startd = ref_n // DOWNSAMPLE, deg_n // DOWNSAMPLE, 1, 1
    elif utt_id == MAXNUTTERANCES:
        startr = int(S.UttSearch_Start[MAXNUTTERANCES])
        startd = startr + S.Utt_DelayEst[MAXNUTTERANCES] / DOWNSAMPLE
        if startd < 0:
            startr = 1 - S.Utt_DelayEst[MAXNUTTERANCES] / DOWNSAMPLE
            startd = 1
        nr = S.UttSearch_End[MAXNUTTERANCES] - startr
        nd = nr
        if startd + nd > deg_n // DOWNSAMPLE:
            nd = deg_n // DOWNSAMPLE - startd
    else:
        startr = int(S.UttSearch_Start[utt_id])
        startd = startr + S.Crude_DelayEst / DOWNSAMPLE
        if startd < 0:
            startr = 1 - S.Crude_DelayEst / DOWNSAMPLE
            startd = 1
        nr = S.UttSearch_End[utt_id] - startr
        nd = nr
        if startd + nd > deg_n // DOWNSAMPLE + 1:
            nd = deg_n // DOWNSAMPLE - startd + 1
    # delays are multiples of the 4 ms window wherever they are divided by it; keep MATLAB's doubles -> whole numbers
    startr, startd, nr, nd = int(round(max(1, startr))), int(round(max(1, startd))), int(round(nr)), int(round(nd))
    max_y, i_max = 0.0, nr
    if nr > 1 and nd > 1:
        Y = _fftnxcorr(ref_log, startr, nr, deg_log, startd, nd)
        i_max = int(np.argmax(Y[1:])) + 1
        max_y = Y[i_max]
        if max_y <= 0:
            max_y, i_max = 0.0, nr
    if utt_id == WHOLE_SIGNAL:
        S.Crude_DelayEst = (i_max - nr) * DOWNSAMPLE
    elif utt_id == MAXNUTTERANCES:
        S.Utt_Delay[MAXNUTTERANCES] = (i_max - nr) * DOWNSAMPLE + S.Utt_DelayEst[MAXNUTTERANCES]
    else:
        S.Utt_DelayEst[utt_id] = (i_max - nr) * DOWNSAMPLE + S.Crude_DelayEst


def _xcorr_frame(ref, deg, startr, startd):
    """|ifft(conj(fft(x1 w)) fft(x2 w))| of one Align_Nfft frame pair (1-based starts)."""
    x1 = ref[startr: startr + ALIGN_NFFT] * _WINDOW
    x2 = deg[startd: startd + ALIGN_NFFT] * _WINDOW
    return np.abs(np.fft.ifft(np.conj(np.fft.fft(x1)) * np.fft.fft(x2)))


def _time_align(S, ref, ref_n, deg, deg_n, utt_id):
    """pesq.m:2479-2550."""
    est = int(S.Utt_DelayEst[utt_id])
    H = np.zeros(ALIGN_NFFT)
    startr = (int(S.UttSearch_Start[utt_id]) - 1) * DOWNSAMPLE + 1
    startd = startr + est
    if startd < 0:
        startr = 1 - est
        startd = 1
    while (startd + ALIGN_NFFT) <= deg_n and (startr + ALIGN_NFFT) <= (int(S.UttSearch_End[utt_id]) - 1) * DOWNSAMPLE:
        X1 = _xcorr_frame(ref, deg, startr, max(startd, 1))
        v_max = np.max(X1) * 0.99
        H[X1 > v_max] += v_max ** 0.125
        startr += ALIGN_NFFT // 4
        startd += ALIGN_NFFT // 4
    hsum = np.sum(H)
    kernel = ALIGN_NFFT // 64
    X2 = np.zeros(ALIGN_NFFT)
    X2[0] = 1.0
    for c in range(2, kernel + 1):
        X2[c - 1] = 1 - (c - 1) / kernel
        X2[ALIGN_NFFT - c + 1] = 1 - (c - 1) / kernel
    X1 = np.fft.ifft(np.fft.fft(H) * np.fft.fft(X2))
    Hn = np.abs(X1) / hsum if hsum > 0 else np.zeros(ALIGN_NFFT)
    i_max = int(np.argmax(Hn)) + 1
    v_max = Hn[i_max - 1]
    if i_max - 1 >= ALIGN_NFFT // 2:
        i_max -= ALIGN_NFFT
    S.Utt_Delay[utt_id] = est + i_max - 1
    S.Utt_DelayConf[utt_id] = v_max


def _id_searchwindows(S, ref_vad, ref_n, deg_n):
    """pesq.m:632-688."""
    utt, flag = 1, 0
    vlen = ref_n // DOWNSAMPLE
    del_start = MINUTTLENGTH - S.Crude_DelayEst / DOWNSAMPLE
    del_end = np.floor((deg_n - S.Crude_DelayEst) / DOWNSAMPLE) - MINUTTLENGTH
    this_start = 0
    for c in range(1, vlen + 1):
        v = ref_vad[c]
        if v > 0 and flag == 0:
            flag = 1
            this_start = c
            S.UttSearch_Start[utt] = max(c - SEARCHBUFFER, 1)
        if (v == 0 or c == vlen - 1) and flag == 1:
            flag = 0
            S.UttSearch_End[utt] = min(c + SEARCHBUFFER, vlen)
            if (c - this_start) >= MINUTTLENGTH and this_start < del_end and c > del_start:
                utt += 1
                if utt > MAXNUTTERANCES - 1:
                    break
    S.Nutterances = utt - 1


def _id_utterances(S, ref_n, ref_vad, deg_n):
    """pesq.m:690-772."""
    utt, flag = 1, 0
    vlen = ref_n // DOWNSAMPLE
    del_start = MINUTTLENGTH - S.Crude_DelayEst / DOWNSAMPLE
    del_end = np.floor((deg_n - S.Crude_DelayEst) / DOWNSAMPLE) - MINUTTLENGTH
    this_start = 0
    for c in range(1, vlen + 1):
        v = ref_vad[c]
        if v > 0.0 and flag == 0:
            flag = 1
            this_start = c
            S.Utt_Start[utt] = c
        if (v == 0 or c == vlen) and flag == 1:
            flag = 0
            S.Utt_End[utt] = c
            if (c - this_start) >= MINUTTLENGTH and this_start < del_end and c > del_start:
                utt += 1
                if utt > MAXNUTTERANCES - 1:
                    break
    S.Utt_Start[1] = SEARCHBUFFER + 1
    S.Nutterances = max(1, S.Nutterances)
    N = S.Nutterances
    S.Utt_End[N] = vlen - SEARCHBUFFER + 1
    for u in range(2, N + 1):
        this_start = S.Utt_Start[u] - 1
        last_end = S.Utt_End[u - 1] - 1
        c = (this_start + last_end) // 2
        S.Utt_Start[u] = c + 1
        S.Utt_End[u - 1] = c + 1
    this_start = (S.Utt_Start[1] - 1) * DOWNSAMPLE + S.Utt_Delay[1]
    if this_start < SB:
        c = SEARCHBUFFER + int(np.floor((DOWNSAMPLE - 1 - S.Utt_Delay[1]) / DOWNSAMPLE))
        S.Utt_Start[1] = c + 1
    last_end = (S.Utt_End[N] - 1) * DOWNSAMPLE + 1 + S.Utt_Delay[N]
    if last_end > deg_n - SB + 1:
        c = int(np.floor((deg_n - S.Utt_Delay[N]) / DOWNSAMPLE)) - SEARCHBUFFER
        S.Utt_End[N] = c + 1
    for u in range(2, N + 1):
        this_start = (S.Utt_Start[u] - 1) * DOWNSAMPLE + S.Utt_Delay[u]
        last_end = (S.Utt_End[u - 1] - 1) * DOWNSAMPLE + S.Utt_Delay[u - 1]
        if this_start < last_end:
            c = int(np.floor((this_start + last_end) / 2))
            this_start = int(np.floor((DOWNSAMPLE - 1 + c - S.Utt_Delay[u]) / DOWNSAMPLE))
            last_end = int(np.floor((c - S.Utt_Delay[u - 1]) / DOWNSAMPLE))
            S.Utt_Start[u] = this_start + 1
            S.Utt_End[u - 1] = last_end + 1


def _hist_pass(ref, deg, startr, startd, cond, step, H, hsum, kernel):
    """The histogram accumulation loop shared by the four scans of split_align (pesq.m:2180-2202 etc.)."""
    tri = kernel - np.abs(np.arange(1 - kernel, kernel))
    while cond(startr, startd):
        X1 = _xcorr_frame(ref, deg, startr, startd)
        v_max = np.max(X1) * 0.99
        n_max = (v_max ** 0.125) / kernel
        for count in np.nonzero(X1 > v_max)[0]:
            hsum += n_max * kernel
            idx = (count + np.arange(1 - kernel, kernel) + ALIGN_NFFT) % ALIGN_NFFT
            np.add.at(H, idx, n_max * tri)
        startr += step
        startd += step
    return startr, startd, hsum


def _split_align(S, ref, ref_n, ref_log, deg, deg_n, deg_log, start_l, sp_start, sp_end, end_l, est_l, conf_l):
    """pesq.m:2109-2477: try to split one utterance in two with different delays; result in S.Best (or None)."""
    utt_len = sp_end - sp_start
    T = MAXNUTTERANCES
    best = dict(DC1=0.0, DC2=0.0)
    kernel = ALIGN_NFFT // 64
    delta = ALIGN_NFFT // (4 * DOWNSAMPLE)
    step = int(np.floor((0.801 * utt_len + 40 * delta - 1) / (40 * delta))) * delta
    pad = max(utt_len // 10, 75)
    BPs = np.zeros(43, dtype=np.int64)
    BPs[1] = sp_start + pad
    n_bps = 1
    while True:
        n_bps += 1
        BPs[n_bps] = BPs[n_bps - 1] + step
        if not (BPs[n_bps] <= (sp_end - pad) and n_bps <= 40):
            break
    S.Best = best
    if n_bps <= 1:
        return
    ED1, ED2 = np.zeros(43, dtype=np.int64), np.zeros(43, dtype=np.int64)
    D1, D2 = np.zeros(43, dtype=np.int64), np.zeros(43, dtype=np.int64)
    DC1, DC2 = np.zeros(43), np.zeros(43)
    for bp in range(1, n_bps):
        S.Utt_DelayEst[T] = est_l
        S.UttSearch_Start[T] = start_l
        S.UttSearch_End[T] = BPs[bp]
        _crude_align(S, ref_log, ref_n, deg_log, deg_n, MAXNUTTERANCES)
        ED1[bp] = S.Utt_Delay[T]
        S.Utt_DelayEst[T] = est_l
        S.UttSearch_Start[T] = BPs[bp]
        S.UttSearch_End[T] = end_l
        _crude_align(S, ref_log, ref_n, deg_log, deg_n, MAXNUTTERANCES)
        ED2[bp] = S.Utt_Delay[T]
    DC1[1: n_bps] = -2.0
    q = ALIGN_NFFT // 4

    def peak(H, hsum, est):
        i_max = int(np.argmax(H)) + 1
        v_max = H[i_max - 1]
        if i_max - 1 >= ALIGN_NFFT // 2:
            i_max -= ALIGN_NFFT
        return est + i_max - 1, (v_max / hsum if hsum > 0.0 else 0.0)

    while True:
        bp = 1
        while bp <= n_bps - 1 and DC1[bp] > -2.0:
            bp += 1
        if bp >= n_bps:
            break
        est = int(ED1[bp])
        H, hsum = np.zeros(ALIGN_NFFT), 0.0
        startr = (start_l - 1) * DOWNSAMPLE + 1
        startd = startr + est
        if startd < 0:
            startr = -est + 1
            startd = 1
        startr, startd = max(1, startr), max(1, startd)

        def cond_fwd(sr, sd, b=bp):
            return (sd + ALIGN_NFFT) <= 1 + deg_n and (sr + ALIGN_NFFT) <= 1 + (BPs[b] - 1) * DOWNSAMPLE
        startr, startd, hsum = _hist_pass(ref, deg, startr, startd, cond_fwd, q, H, hsum, kernel)
        D1[bp], DC1[bp] = peak(H, hsum, est)
        while bp < n_bps - 1:
            bp += 1
            if ED1[bp] == est and DC1[bp] <= -2.0:
                startr, startd, hsum = _hist_pass(ref, deg, startr, startd, lambda sr, sd, b=bp: (sd + ALIGN_NFFT) <= 1 + deg_n and
                                                  (sr + ALIGN_NFFT) <= (BPs[b] - 1) * DOWNSAMPLE + 1, q, H, hsum, kernel)
                D1[bp], DC1[bp] = peak(H, hsum, est)
    for bp in range(1, n_bps):
        DC2[bp] = -2.0 if DC1[bp] > conf_l else 0.0
    while True:
        bp = n_bps - 1
        while bp >= 1 and DC2[bp] > -2.0:
            bp -= 1
        if bp < 1:
            break
        est = int(ED2[bp])
        H, hsum = np.zeros(ALIGN_NFFT), 0.0
        startr = (end_l - 1) * DOWNSAMPLE + 1 - ALIGN_NFFT
        startd = startr + est
        if (startd + ALIGN_NFFT) > deg_n + 1:
            startd = deg_n - ALIGN_NFFT + 1
            startr = startd - est
        startr, startd, hsum = _hist_pass(ref, deg, startr, startd, lambda sr, sd, b=bp: sd >= 1 and sr >= (BPs[b] - 1) * DOWNSAMPLE + 1,
                                          -q, H, hsum, kernel)
        D2[bp], DC2[bp] = peak(H, hsum, est)
        while bp > 1:
            bp -= 1
            if ED2[bp] == est and DC2[bp] <= -2.0:
                startr, startd, hsum = _hist_pass(ref, deg, startr, startd,
                                                  lambda sr, sd, b=bp: sd >= 1 and sr >= (BPs[b] - 1) * DOWNSAMPLE + 1, -q, H, hsum, kernel)
                D2[bp], DC2[bp] = peak(H, hsum, est)
    for bp in range(1, n_bps):
        if (abs(D2[bp] - D1[bp]) >= DOWNSAMPLE and (DC1[bp] + DC2[bp]) > (best['DC1'] + best['DC2']) and DC1[bp] > conf_l and
                DC2[bp] > conf_l):
            best.update(ED1=int(ED1[bp]), D1=int(D1[bp]), DC1=float(DC1[bp]), ED2=int(ED2[bp]), D2=int(D2[bp]), DC2=float(DC2[bp]),
                        BP=int(BPs[bp]))


def _utterance_split(S, ref, ref_n, ref_vad, ref_log, deg, deg_n, deg_log):
    """pesq.m:2585-2707."""
    u = 1
    while u <= S.Nutterances and S.Nutterances <= MAXNUTTERANCES - 2:
        est_l, conf_l = int(S.Utt_DelayEst[u]), float(S.Utt_DelayConf[u])
        start_l, end_l = int(S.Utt_Start[u]), int(S.Utt_End[u])
        sp_start = max(1, start_l)
        while sp_start < end_l and ref_vad[sp_start] <= 0.0:
            sp_start += 1
        sp_end = end_l
        while sp_end > start_l and ref_vad[min(sp_end, len(ref_vad) - 1)] <= 0:
            sp_end -= 1
        sp_end += 1
        if sp_end - sp_start >= 200:
            _split_align(S, ref, ref_n, ref_log, deg, deg_n, deg_log, start_l, sp_start, sp_end, end_l, est_l, conf_l)
            b = S.Best
            if b['DC1'] > conf_l and b['DC2'] > conf_l:
                for step in range(S.Nutterances, u, -1):
                    S.Utt_DelayEst[step + 1] = S.Utt_DelayEst[step]
                    S.Utt_Delay[step + 1] = S.Utt_Delay[step]
                    S.Utt_DelayConf[step + 1] = S.Utt_DelayConf[step]
                    S.Utt_Start[step + 1] = S.Utt_Start[step]
                    S.Utt_End[step + 1] = S.Utt_End[step]
                    S.UttSearch_Start[step + 1] = S.Utt_Start[step]
                    S.UttSearch_End[step + 1] = S.Utt_End[step]
                S.Nutterances += 1
                S.Utt_DelayEst[u], S.Utt_Delay[u], S.Utt_DelayConf[u] = b['ED1'], b['D1'], b['DC1']
                S.Utt_DelayEst[u + 1], S.Utt_Delay[u + 1], S.Utt_DelayConf[u + 1] = b['ED2'], b['D2'], b['DC2']
                S.UttSearch_Start[u + 1] = S.UttSearch_Start[u]
                S.UttSearch_End[u + 1] = S.UttSearch_End[u]
                if b['D2'] < b['D1']:
                    S.Utt_Start[u], S.Utt_End[u] = start_l, b['BP']
                    S.Utt_Start[u + 1], S.Utt_End[u + 1] = b['BP'], end_l
                else:
                    h = int(np.floor((b['D2'] - b['D1']) / (2 * DOWNSAMPLE)))
                    S.Utt_Start[u], S.Utt_End[u] = start_l, b['BP'] + h
                    S.Utt_Start[u + 1], S.Utt_End[u + 1] = b['BP'] - h, end_l
                if (S.Utt_Start[u] - SEARCHBUFFER - 1) * DOWNSAMPLE + 1 + b['D1'] < 0:
                    S.Utt_Start[u] = SEARCHBUFFER + 1 + int(np.floor((DOWNSAMPLE - 1 - b['D1']) / DOWNSAMPLE))
                if ((S.Utt_End[u + 1] - 1) * DOWNSAMPLE + 1 + b['D2']) > (deg_n - SB):
                    S.Utt_End[u + 1] = int(np.floor((deg_n - b['D2']) / DOWNSAMPLE)) - SEARCHBUFFER + 1
            else:
                u += 1
        else:
            u += 1


# ------------------------------------------------------------------------------------------------ psychoacoustic model
def _short_term_fft(nf, data, whann, start):
    X = np.fft.fft(data[start: start + nf] * whann)
    hz = np.abs(X[: nf // 2]) ** 2
    hz[0] = 0
    return hz


def _freq_warping(hz):
    """pesq.m:1703-1722 -> pitch power density of the 49 Bark bands."""
    cs = np.concatenate([[0.0], np.cumsum(hz)])
    return (cs[_BAND_EDGES[1:]] - cs[_BAND_EDGES[:-1]]) * POW_CORR * SP


def _total_audible(ppd, factor):
    h = ppd[1:]
    return float(np.sum(h[h > factor * ABS_THRESH[1:]]))


_ZW_H = np.where(CENTRE_BARK < 4, 6.0 / (CENTRE_BARK + 2.0), 1.0)
_ZW_POW = 0.23 * np.minimum(_ZW_H, 2.0) ** 0.15


def _intensity_warping(ppd):
    """pesq.m:1600-1630."""
    out = np.zeros(NB)
    m = ppd > ABS_THRESH
    out[m] = ((ABS_THRESH[m] / 0.5) ** _ZW_POW[m]) * ((0.5 + 0.5 * ppd[m] / ABS_THRESH[m]) ** _ZW_POW[m] - 1.0)
    return out * SL


def _pseudo_lp(x, p):
    """pesq.m:1632-1648 (bands 2..Nb)."""
    w = WIDTH_BARK[1:]
    tot = np.sum(w)
    r = np.sum((np.abs(x[1:]) * w) ** p)
    return (r / tot) ** (1.0 / p) * tot


def _asym(dist, ppd_ref, ppd_deg):
    """pesq.m:1582-1598."""
    h = ((ppd_deg + 50.0) / (ppd_ref + 50.0)) ** 1.2
    h = np.where(h > 12.0, 12.0, np.where(h < 3.0, 0.0, h))
    return dist * h


def _disturbance(ppd_ref, ppd_deg):
    """Loudness difference with the 0.25 * min deadzone (pesq.m:1044-1070) -> (D, DA frame disturbances, raw density)."""
    lr, ld = _intensity_warping(ppd_ref), _intensity_warping(ppd_deg)
    d = ld - lr
    m = 0.25 * np.minimum(ld, lr)
    d = np.where(d > m, d - m, np.where(d < -m, d + m, 0.0))
    return _pseudo_lp(d, 2), _pseudo_lp(_asym(d, ppd_ref, ppd_deg), 1)


def _compute_delay(start, stop, search_range, t1, t2):
    """pesq.m:1527-1580 (1-based series)."""
    n = stop - start + 1
    p2 = int(2 ** np.ceil(np.log2(2 * n)))
    power1 = _pow_of(t1, start, stop, n) * n / p2
    power2 = _pow_of(t2, start, stop, n) * n / p2
    norm = np.sqrt(power1 * power2)
    x1, x2 = np.zeros(p2), np.zeros(p2)
    x1[:n] = np.abs(t1[start: stop + 1])
    x2[:n] = np.abs(t2[start: stop + 1])
    y = np.fft.ifft(np.conj(np.fft.fft(x1) / p2) * np.fft.fft(x2))
    best, maxc = 0, 0.0
    if norm > 0:
        for i in range(-search_range, 0):
            h = abs(y[i + p2]) / norm
            if h > maxc:
                maxc, best = h, i
        for i in range(0, search_range):
            h = abs(y[i]) / norm
            if h > maxc:
                maxc, best = h, i
    return best - 1, maxc


def _lpq_weight(start_frame, stop_frame, p_syl, p_time, fd, tw):
    """pesq.m:1479-1525 (fd / tw indexed by 0-based frame)."""
    res, tot_w = 0.0, 0.0
    for s0 in range(start_frame, stop_frame + 1, 10):
        acc = 0.0
        for f in range(s0, s0 + 20):
            if f <= stop_frame:
                acc += fd[f] ** p_syl
        acc = (acc / 20.0) ** (1.0 / p_syl)
        w = tw[s0 - start_frame]
        res += (w * acc) ** p_time
        tot_w += w ** p_time
    return (res / tot_w) ** (1.0 / p_time)


def _psychoacoustic(S, ref, ref_n, deg, deg_n):
    """pesq.m:785-1477 -> raw PESQ score."""
    max_n = max(ref_n, deg_n)
    nf = DOWNSAMPLE * 8
    half = nf // 2
    whann = 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(nf) / nf))
    # leading / trailing silence of the reference (sum of 5 |samples| < 500)
    skip_start = 0
    while skip_start < max_n / 2 and np.sum(np.abs(ref[skip_start + SB + 1: skip_start + SB + 6])) < 500:
        skip_start += 1
    skip_end = 0
    e0 = max_n - SB + PAD
    while skip_end < max_n / 2 and np.sum(np.abs(ref[e0 - skip_end - 4: e0 - skip_end + 1])) < 500:
        skip_end += 1
    start_frame = skip_start // half
    stop_frame = (max_n - 2 * SB + PAD - skip_end) // half - 1
    nfr = stop_frame + 1
    N = S.Nutterances
    utt_first = (S.Utt_Start[1: N + 1] - 1) * DOWNSAMPLE + 1          # first sample of each utterance

    def delay_at(sample):
        u = int(np.searchsorted(utt_first, sample, side='right'))      # last utterance starting at or before `sample`
        return int(S.Utt_Delay[max(u, 1)])

    ppd_ref = np.zeros((nfr, NB))
    ppd_deg = np.zeros((nfr, NB))
    silent = np.zeros(nfr, dtype=bool)
    for fr in range(nfr):
        s_ref = 1 + SB + fr * half
        ppd_ref[fr] = _freq_warping(_short_term_fft(nf, ref, whann, s_ref))
        s_deg = s_ref + delay_at(s_ref)
        if s_deg > 0 and s_deg + nf - 1 < max_n + PAD:
            ppd_deg[fr] = _freq_warping(_short_term_fft(nf, deg, whann, s_deg))
        silent[fr] = _total_audible(ppd_ref[fr], 1e2) < 1e7
    tot_frames = (max_n - 2 * SB + PAD) // half - 1

    def time_avg(ppd):
        m = (~silent)[:, None] & (ppd > 100.0 * ABS_THRESH[None, :])
        return np.sum(np.where(m, ppd, 0.0), axis=0) / tot_frames
    avg_ref, avg_deg = time_avg(ppd_ref), time_avg(ppd_deg)
    x = np.clip((avg_deg + 1000.0) / (avg_ref + 1000.0), 0.01, 100.0)
    ppd_ref = ppd_ref * x[None, :]                                      # freq_resp_compensation

    fd = np.zeros(nfr)
    fda = np.zeros(nfr)
    total_power_ref = np.zeros(nfr)
    old = 1.0
    bad = False
    for fr in range(nfr):
        ta_ref, ta_deg = _total_audible(ppd_ref[fr], 1), _total_audible(ppd_deg[fr], 1)
        total_power_ref[fr] = ta_ref
        scale = (ta_ref + 5e3) / (ta_deg + 5e3)
        if fr > 0:
            scale = 0.2 * old + 0.8 * scale
        old = scale
        scale = min(max(scale, 3e-4), 5.0)
        ppd_deg[fr] *= scale
        fd[fr], fda[fr] = _disturbance(ppd_ref[fr], ppd_deg[fr])
        if fd[fr] > 30:
            bad = True
    # frames around a delay jump between utterances are skipped
    for u in range(2, N + 1):
        frame1 = int(np.floor(((S.Utt_Start[u] - 1 - SEARCHBUFFER) * DOWNSAMPLE + 1 + S.Utt_Delay[u]) / half))
        j = int(np.floor(np.floor((S.Utt_End[u - 1] - 1 - SEARCHBUFFER) * DOWNSAMPLE + 1 + S.Utt_Delay[u - 1]) / half))
        jump = int(S.Utt_Delay[u] - S.Utt_Delay[u - 1])
        frame1 = max(min(frame1, j), 0)
        if jump < -half:
            frame2 = int(np.floor(((S.Utt_Start[u] - 1 - SEARCHBUFFER) * DOWNSAMPLE + 1 + max(0, abs(jump))) / half)) + 1
            for fr in range(frame1, frame2 + 1):
                if fr < stop_frame:
                    fd[fr] = 0
                    fda[fr] = 0
    if bad:
        # delay-compensated degraded signal
        nn = PAD + max_n
        idx = np.arange(SB + 1, nn - SB + 1)
        utt_first_i = (S.Utt_Start[1: N + 1] - 1) * DOWNSAMPLE
        # MATLAB: utt decreases while (Utt_Start(utt)-1)*DS > i  -> largest utt with (Utt_Start-1)*DS <= i
        which = np.maximum(np.searchsorted(utt_first_i, idx, side='right'), 1)
        j = np.clip(idx + S.Utt_Delay[which], SB + 1, nn - SB)
        tweaked = np.zeros(nn + 1)
        tweaked[idx] = deg[j]
        is_bad = fd > 30
        is_bad[0] = False
        smeared = np.zeros(nfr, dtype=bool)
        for fr in range(2, stop_frame - 1 - 2 + 1):
            left = np.max(is_bad[fr - 2: fr + 1])
            right = np.max(is_bad[fr: fr + 3])
            smeared[fr] = min(left, right)
        intervals = []
        fr = 0
        while fr <= stop_frame:
            while fr <= stop_frame and not smeared[fr]:
                fr += 1
            if fr <= stop_frame:
                st = 1 + fr
                while fr <= stop_frame and smeared[fr]:
                    fr += 1
                if fr <= stop_frame:
                    sp = 1 + fr
                    if sp - st >= 5:
                        intervals.append([st, sp])
        srange = 4 * nf
        fixes = []
        for st, sp in intervals:
            s_samp = (st - 1) * half + SB + 1
            e_samp = (sp - 1) * half + nf + SB
            sp = min(sp, stop_frame + 1)
            n_in = e_samp - s_samp + 1
            r = np.zeros(2 * srange + n_in + 1)
            r[srange + 1: srange + n_in + 1] = ref[s_samp + 1: s_samp + n_in + 1]
            jj = np.clip(s_samp - srange + np.arange(0, 2 * srange + n_in), SB + 1, max_n - SB + PAD)
            d = _one(tweaked[jj])
            delay, corr = _compute_delay(1, 2 * srange + n_in, srange, r, d)
            fixes.append((st, sp, s_samp, e_samp, delay if corr >= 0.5 else 0))
        if fixes:
            doubly = tweaked[: max_n + PAD + 1].copy()
            for st, sp, s_samp, e_samp, delay in fixes:
                i = np.arange(s_samp, e_samp + 1)
                doubly[i] = tweaked[np.clip(i + delay, 1, max_n)]
            for st, sp, s_samp, e_samp, delay in fixes:
                for fr in range(st - 1, sp - 1):
                    ppd_deg[fr] = _freq_warping(_short_term_fft(nf, doubly, whann, SB + fr * half + 1))
                old = 1.0
                for fr in range(st - 1, sp - 1):
                    ta_ref, ta_deg = _total_audible(ppd_ref[fr], 1), _total_audible(ppd_deg[fr], 1)
                    scale = (ta_ref + 5e3) / (ta_deg + 5e3)
                    if fr > 0:
                        scale = 0.2 * old + 0.8 * scale
                    old = scale
                    scale = min(max(scale, 3e-4), 5.0)
                    ppd_deg[fr] *= scale
                    d_, a_ = _disturbance(ppd_ref[fr], ppd_deg[fr])
                    fd[fr] = min(fd[fr], d_)
                    fda[fr] = min(fda[fr], a_)
    tw = np.ones(nfr)
    if nfr > 1000:
        n = (max_n - 2 * SB) // half - 1
        twf = min((n - 1000) / 5500.0, 0.5)
        tw = (1.0 - twf) + twf * np.arange(nfr) / n
    h = ((total_power_ref + 1e5) / 1e7) ** 0.04
    fd = np.minimum(fd / h, 45.0)
    fda = np.minimum(fda / h, 45.0)
    d_ind = _lpq_weight(start_frame, stop_frame, 6, 2, fd, tw)
    a_ind = _lpq_weight(start_frame, stop_frame, 6, 2, fda, tw)
    return 4.5 - 0.1 * d_ind - 0.0309 * a_ind


# ------------------------------------------------------------------------------------------------ entry point
def pesq_raw(ref, deg, fs=16000):
    """Raw P.862 score of `deg` against `ref` (float waveforms in [-1, 1], 16 kHz), wide-band input filter."""
    if fs != FS:
        raise ValueError('this restatement covers the 16 kHz wide-band branch of pesq.m (the one the paper reports)')
    ref = np.asarray(ref, dtype=np.float64).ravel() * 32768.0
    deg = np.asarray(deg, dtype=np.float64).ravel() * 32768.0
    ref_n, deg_n = len(ref) + 2 * SB, len(deg) + 2 * SB
    ref_d = _one(np.concatenate([np.zeros(SB), ref, np.zeros(PAD + SB)]))
    deg_d = _one(np.concatenate([np.zeros(SB), deg, np.zeros(PAD + SB)]))
    max_n = max(ref_n, deg_n)
    ref_d = _fix_power_level(ref_d, ref_n, max_n)
    deg_d = _fix_power_level(deg_d, deg_n, max_n)
    ref_d[1:] = _sosfilt(WB_SOS[:, [0, 1, 2, 4, 5]], ref_d[1:])       # apply_filters_WB
    deg_d[1:] = _sosfilt(WB_SOS[:, [0, 1, 2, 4, 5]], deg_d[1:])
    model_ref, model_deg = ref_d.copy(), deg_d.copy()
    # input_filter: DC block + the IRS-like IIR, only for VAD / alignment
    ref_a, deg_a = _dc_block(ref_d, ref_n), _dc_block(deg_d, deg_n)
    ref_a[1:] = _sosfilt(IIR_SOS_16K, ref_a[1:])
    deg_a[1:] = _sosfilt(IIR_SOS_16K, deg_a[1:])
    ref_vad, ref_log = _apply_vad(ref_a, ref_n)
    deg_vad, deg_log = _apply_vad(deg_a, deg_n)
    S = _State()
    _crude_align(S, ref_log, ref_n, deg_log, deg_n, WHOLE_SIGNAL)
    _id_searchwindows(S, ref_vad, ref_n, deg_n)
    for u in range(1, S.Nutterances + 1):
        _crude_align(S, ref_log, ref_n, deg_log, deg_n, u)
        _time_align(S, ref_a, ref_n, deg_a, deg_n, u)
    _id_utterances(S, ref_n, ref_vad, deg_n)
    _utterance_split(S, ref_a, ref_n, ref_vad, ref_log, deg_a, deg_n, deg_log)
    if ref_n < deg_n:
        model_ref = np.concatenate([model_ref, np.zeros(deg_n + PAD + 1 - len(model_ref))])
    elif ref_n > deg_n:
        model_deg = np.concatenate([model_deg, np.zeros(ref_n + PAD + 1 - len(model_deg))])
    return _psychoacoustic(S, model_ref, ref_n, model_deg, deg_n)


def pesq(ref, deg, fs=16000):
    """Wide-band MOS-LQO (P.862.2 mapping, pesq.m:203-207) - the `wb-PESQ` column of the reference's tables."""
    raw = pesq_raw(ref, deg, fs)
    return float(0.999 + (4.999 - 0.999) / (1.0 + np.exp(-1.3669 * raw + 3.8224)))
