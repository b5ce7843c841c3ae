// U^2-Net style gated encoder / decoder shared by TaylorSENet and G2Net.
//
// Reference: TaylorSENet/TaylorSENet.py:336-603 (U2Net_Encoder, U2Net_Decoder, En_unet_module, Conv2dunit,
// Deconv2dunit, GateConv2d, GateConvTranspose2d); G2Net_VB/gaf_net_320.py:277-486 has the same encoder.
// Every (de)conv is a tap-table GEMM (gated ones as (value, gate) row pairs), InstanceNorm2d(affine) + PReLU(C) is one
// plane-wise kernel, 'cat' skips are two-source K loops, 'add' skips / residuals are one elementwise kernel.
#pragma once
#include "blocks.h"

namespace se {

struct ConvIN {        // conv (plain or gated) -> InstanceNorm2d -> PReLU
    GCPlan plan;
    NormAct na;
    int cout = 0, kf = 3;
    void free() {
        gc_free_plan(plan);
        na.free();
    }
};
struct DeconvIN {
    DeconvPlan plan;
    NormAct na;
    int cout = 0, kf = 3;
    void free() {
        free_deconv_plan(plan);
        na.free();
    }
};

// GateConv2d / GateConvTranspose2d produce 2C channels [value ; gate] from ONE conv: interleave to (value, gate) pairs
inline DenseW gate_pairs(DenseW w) {
    const int C = w.M / 2;
    std::vector<int> perm(w.M);
    for (int j = 0; j < C; ++j) {
        perm[2 * j] = j;
        perm[2 * j + 1] = C + j;
    }
    permute_rows(w, perm);
    return w;
}

inline ConvIN load_gate_conv_in(const TrackedSD& sd, const std::string& p, int cin, int cout, int kt, int kf, int c0 = -1) {
    ConvIN c;
    c.cout = cout;
    c.kf = kf;
    const std::string key = p + (kt > 1 ? "0.conv.1." : "0.conv.");
    DenseW w = conv_weights(sd.get(key + "weight", {2 * cout, cin, kt, kf}), &sd.get(key + "bias", {2 * cout}), true);
    c.plan = make_conv_plan(gate_pairs(w), 2, 0, kt - 1, 1, 1, ACT_NONE, {}, EPI_GLU, 401, c0);
    c.na.load(sd, p + "1.", p + "2.");
    return c;
}
inline DeconvIN load_gate_deconv_in(const TrackedSD& sd, const std::string& p, int cin, int cout, int kt, int kf, int c0) {
    DeconvIN c;
    c.cout = cout;
    c.kf = kf;
    const std::string key = p + (kt > 1 ? "0.conv.0." : "0.conv.");
    DenseW w = deconv_weights(sd.get(key + "weight", {cin, 2 * cout, kt, kf}), &sd.get(key + "bias", {2 * cout}), true);
    c.plan = make_deconv_plan(gate_pairs(w), 2, 0, 0, ACT_NONE, {}, 401, c0, nullptr, EPI_GLU);
    c.na.load(sd, p + "1.", p + "2.");
    return c;
}

// Key / kernel conventions that differ between the two reference files that share this encoder.
struct UnetStyle {
    bool two_conv_gate;   // true: Gate_2dconv = two convs `conv` / `gate_conv` (G2Net); false: one conv to 2C (TaylorSENet)
    int inner_kt;         // time extent of the nested (de)convs: 2 (TaylorSENet k2 = (2,3)) or 1 (G2Net k2 = (1,3))
};
constexpr UnetStyle UNET_TAYLOR{false, 2};
constexpr UnetStyle UNET_G2NET{true, 1};

// Gate_2dconv (G2Net_VB/gaf_net_320.py:465-486): value and gate are separate convs, always behind a ConstantPad2d
inline ConvIN load_gate2_conv_in(const TrackedSD& sd, const std::string& p, int cin, int cout, int kt, int kf) {
    ConvIN c;
    c.cout = cout;
    c.kf = kf;
    DenseW a = conv_weights(sd.get(p + "0.conv.1.weight", {cout, cin, kt, kf}), &sd.get(p + "0.conv.1.bias", {cout}), true);
    DenseW g = conv_weights(sd.get(p + "0.gate_conv.1.weight", {cout, cin, kt, kf}), &sd.get(p + "0.gate_conv.1.bias", {cout}), true);
    c.plan = make_conv_plan(interleave_rows(a, g), 2, 0, kt - 1, 1, 1, ACT_NONE, {}, EPI_GLU, 401);
    c.na.load(sd, p + "1.", p + "2.");
    return c;
}

struct UnetScratch {
    float* lev[5] = {};     // encoder levels 1..4 of the nested U-Net
    float* dec[5] = {};     // decoder outputs at levels 0..3
    float* nrm[10] = {};    // [B][64] float4 each: on-the-fly InstanceNorm parameters of the module's raw tensors (UnetModule::run)
    void alloc(Arena& a, size_t BT, int B) {
        const int F[5] = {79, 39, 19, 9, 4};
        for (int i = 0; i < 5; ++i) {
            lev[i] = a.alloc_f(BT * 64 * F[i]);
            dec[i] = a.alloc_f(BT * 64 * F[i]);
        }
        for (auto& n : nrm) n = a.alloc_f((size_t)B * 64 * 4);
    }
};

struct UnetModule {     // En_unet_module (TaylorSENet.py:441-496)
    bool de = false;
    int scale = 1, k1f = 3;
    ConvIN in_c;
    DeconvIN in_d;
    ConvIN enco[4];
    DeconvIN deco[4];

    void load(const TrackedSD& sd, const std::string& p, int cin, int k1t, int k1f_, int scale_, bool de_, int c0,
              UnetStyle sty = UNET_TAYLOR) {
        de = de_;
        scale = scale_;
        k1f = k1f_;
        if (de) in_d = load_gate_deconv_in(sd, p + "in_conv.", cin, 64, k1t, k1f, c0);
        else if (sty.two_conv_gate) in_c = load_gate2_conv_in(sd, p + "in_conv.", cin, 64, k1t, k1f);
        else in_c = load_gate_conv_in(sd, p + "in_conv.", cin, 64, k1t, k1f, c0);
        const int kt = sty.inner_kt;
        // Sequential indices: with a time pad / chomp the conv sits at 1 (enc) and the norm at 2; without, 0 and 1
        const std::string ec = kt > 1 ? "1." : "0.", en = kt > 1 ? "2." : "1.", ep = kt > 1 ? "3." : "2.";
        const std::string dn = kt > 1 ? "2." : "1.", dp = kt > 1 ? "3." : "2.";
        for (int i = 0; i < scale; ++i) {
            const std::string q = p + "enco." + std::to_string(i) + ".conv.";     // Conv2dunit k2 = (2,3)
            enco[i].cout = 64;
            DenseW w = conv_weights(sd.get(q + ec + "weight", {64, 64, kt, 3}), &sd.get(q + ec + "bias", {64}), true);
            enco[i].plan = make_conv_plan(w, 2, 0, kt - 1, 1, 1, ACT_NONE, {}, EPI_ACT, 401);
            enco[i].na.load(sd, q + en, q + ep);
            const std::string r = p + "deco." + std::to_string(i) + ".deconv.";   // Deconv2dunit ('add' for i = 0, else 'cat')
            const int ci = i == 0 ? 64 : 128;
            deco[i].cout = 64;
            DenseW dw = deconv_weights(sd.get(r + "0.weight", {ci, 64, kt, 3}), &sd.get(r + "0.bias", {64}), true);
            deco[i].plan = make_deconv_plan(dw, 2, 0, 0, ACT_NONE, {}, 401, i == 0 ? -1 : 64);
            deco[i].na.load(sd, r + dn, r + dp);
        }
    }
    void free() {
        if (de) in_d.free();
        else in_c.free();
        for (int i = 0; i < scale; ++i) {
            enco[i].free();
            deco[i].free();
        }
    }
    int out_F(int Fin) const { return de ? (Fin - 1) * 2 + k1f : (Fin - k1f) / 2 + 1; }

    // every nested (de)conv can hand statistics to its consumers and normalise its own sources on the fly
    bool fold_ok() const {
        bool ok = !(de ? in_d.na.cum : in_c.na.cum) && (de ? in_d.na.gain_nonzero : in_c.na.gain_nonzero) &&
                  (de ? deconv_stats_supported(in_d.plan) : conv_stats_supported(in_c.plan));
        for (int i = 0; i < scale && ok; ++i)
            ok = !enco[i].na.cum && !deco[i].na.cum && enco[i].na.gain_nonzero && deco[i].na.gain_nonzero &&
                 conv_stats_supported(enco[i].plan) && conv_nrm_supported(enco[i].plan) && deconv_stats_supported(deco[i].plan) &&
                 deconv_nrm_supported(deco[i].plan);
        // (round 6) the fold is a per-module decision taken from the consumers' taps: the on-the-fly normalisation costs three vector
        // instructions per B VALUE, i.e. per staged input value times the taps that read it at different FRAMES (frequency taps are
        // different patch rows); the pass it deletes costs one read + one write per value.  With kernels that have no extent in time
        // (G2Net's (1, 3)) every staged value is normalised once and the fold wins (6 382 against 6 216 utt/s at batch 256); with two
        // time taps (TaylorSENet's k2 = (2, 3)) it is normalised twice - its U^2-Net levels run at 87-92 TFLOP/s folded against
        // 105-109 unfolded and the whole model 2 372 against 2 395 - so such modules keep the apply pass.  SE_IN_FOLD=2: always.
        static const int fold_env = getenv("SE_IN_FOLD") ? atoi(getenv("SE_IN_FOLD")) : 1;
        if (fold_env != 2)
            for (int i = 0; i < scale && ok; ++i) ok = enco[i].plan.p.causal && enco[i].plan.lookback == 0;
        return ok;
    }
    // InstanceNorm + PReLU of every tensor inside the module applied by its consumers (gc_kernel NRM): the nested U-Net's tensors
    // exist only raw; ONE elementwise pass remains - the module's output, PReLU(IN(last deconv)) + PReLU(IN(in_conv)), which the
    // next module reads.  Rounds 2-4 made a read + write pass over every level (2 * scale + 1 passes, 13-15 % of a step).
    void run_folded(const Act4& in0, const Act4* in1, float* out, const UnetScratch& s, int B, int T, hipStream_t st,
                    Profiler* pf) const {
        const int F0 = out_F(in0.F);
        if (de) deconv_stats_nrm(in_d.plan, in_d.na, in0, in1, out, s.nrm[0], 64, F0, B, T, st, pf);
        else conv_stats_nrm(in_c.plan, in_c.na, in0, in1, out, s.nrm[0], 64, F0, B, T, st, pf);
        int Fs[6];
        Fs[0] = F0;
        float* xs[5];
        xs[0] = out;
        auto lvl = [](int F) { return F >= 79 ? 0 : F >= 39 ? 1 : F >= 19 ? 2 : F >= 9 ? 3 : 4; };
        for (int i = 0; i < scale; ++i) {
            Fs[i + 1] = (Fs[i] - 3) / 2 + 1;
            float* y = s.lev[lvl(Fs[i + 1])];
            conv_stats_nrm(enco[i].plan, enco[i].na, act4(xs[i], 64, Fs[i], T).with_nrm(s.nrm[i]), nullptr, y, s.nrm[i + 1], 64,
                           Fs[i + 1], B, T, st, pf);
            xs[i + 1] = y;
        }
        const float* x = xs[scale];
        const float* nx = s.nrm[scale];
        for (int i = 0; i < scale; ++i) {
            const int Fi = Fs[scale - i], Fo = Fs[scale - i - 1];
            float* y = s.dec[lvl(Fo)];
            const Act4 a0 = act4(x, 64, Fi, T).with_nrm(nx);
            const Act4 a1 = act4(xs[scale - i], 64, Fi, T).with_nrm(s.nrm[scale - i]);
            float* ny = s.nrm[5 + i];
            deconv_stats_nrm(deco[i].plan, deco[i].na, a0, i == 0 ? nullptr : &a1, y, ny, 64, Fo, B, T, st, pf);
            if (i + 1 == scale) launch_instnorm_apply2(y, ny, out, s.nrm[0], out, B, 64, Fo * T, st);      // + the module's residual
            x = y;
            nx = ny;
        }
    }

    // in0 (+ optional in1 concatenated on channels), both with F = Fin  ->  out [B][64][out_F][T]
    void run(const Act4& in0, const Act4* in1, float* out, const UnetScratch& s, int B, int T, hipStream_t st,
             Profiler* pf) const {
        if (in_stats_enabled() && fold_ok()) return run_folded(in0, in1, out, s, B, T, st, pf);
        const int F0 = out_F(in0.F);
        if (de) {
            deconv_norm2d_prelu(in_d.plan, in_d.na, in0, in1, out, out, 64, F0, B, T, st, pf);
        } else {
            conv_norm2d_prelu(in_c.plan, in_c.na, in0, in1, out, out, 64, F0, B, T, st, pf);
        }
        int Fs[6];
        Fs[0] = F0;
        float* xs[5];
        xs[0] = out;
        // which scratch level holds F: levels are sized {79,39,19,9,4}
        auto lvl = [](int F) { return F >= 79 ? 0 : F >= 39 ? 1 : F >= 19 ? 2 : F >= 9 ? 3 : 4; };
        for (int i = 0; i < scale; ++i) {
            Fs[i + 1] = (Fs[i] - 3) / 2 + 1;
            float* y = s.lev[lvl(Fs[i + 1])];
            conv_norm2d_prelu(enco[i].plan, enco[i].na, act4(xs[i], 64, Fs[i], T), nullptr, y, y, 64, Fs[i + 1], B, T, st, pf);
            xs[i + 1] = y;
        }
        const float* x = xs[scale];
        for (int i = 0; i < scale; ++i) {
            const int Fi = Fs[scale - i], Fo = Fs[scale - i - 1];
            float* y = s.dec[lvl(Fo)];
            Act4 a0 = act4(x, 64, Fi, T);
            Act4 a1 = act4(xs[scale - i], 64, Fi, T);              // x_list[-(i+1)] ('cat' levels, i > 0)
            // the last decoder level also adds the module's residual (x_resi + x): folded into its norm pass
            const bool lastlvl = i + 1 == scale;
            deconv_norm2d_prelu(deco[i].plan, deco[i].na, a0, i == 0 ? nullptr : &a1, y, lastlvl ? out : y, 64, Fo, B, T, st, pf,
                                lastlvl ? out : nullptr);
            x = y;
        }
    }
};

struct U2Encoder {      // U2Net_Encoder (TaylorSENet.py:336-370)
    UnetModule m[4];
    ConvIN last;
    // k1t: time extent of the inter-module gated convs (TaylorSENet k1 = (1,3) -> 1; G2Net k1 = (2,3) -> 2)
    void load(const TrackedSD& sd, const std::string& p, int cin, UnetStyle sty = UNET_TAYLOR, int k1t = 1) {
        m[0].load(sd, p + "meta_unet_list.0.", cin, 2, 5, 4, false, -1, sty);
        m[1].load(sd, p + "meta_unet_list.1.", 64, k1t, 3, 3, false, -1, sty);
        m[2].load(sd, p + "meta_unet_list.2.", 64, k1t, 3, 2, false, -1, sty);
        m[3].load(sd, p + "meta_unet_list.3.", 64, k1t, 3, 1, false, -1, sty);
        last = sty.two_conv_gate ? load_gate2_conv_in(sd, p + "last_conv.", 64, 64, k1t, 3)
                                 : load_gate_conv_in(sd, p + "last_conv.", 64, 64, k1t, 3);
    }
    void free() {
        for (auto& x : m) x.free();
        last.free();
    }
    // in [B][cin][161][T] -> ens[0..4] with F = 79, 39, 19, 9, 4 (ens[4] is the bottleneck)
    void run(const Act4& in, float* const ens[5], const UnetScratch& s, int B, int T, hipStream_t st, Profiler* pf) const {
        Act4 x = in;
        const int F[5] = {79, 39, 19, 9, 4};
        for (int i = 0; i < 4; ++i) {
            m[i].run(x, nullptr, ens[i], s, B, T, st, pf);
            x = act4(ens[i], 64, F[i], T);
        }
        conv_norm2d_prelu(last.plan, last.na, x, nullptr, ens[4], ens[4], 64, 4, B, T, st, pf);
    }
};

}  // namespace se
