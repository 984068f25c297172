// ablation.hpp -- timing experiments on the walk kernel.  NOT part of the product: the library sources include this
// file only when they are compiled with -DRT_ABLATION, and such a build reports a different ABI version
// (rt_abi_version() | 0x1000), so the Python host refuses to load it unless RT_AMD_ALLOW_ABLATION=1 is set (the
// timing tools under tools/ set it).  Every switch changes results or speed in a way no parity claim covers.
//
//   -DRT_ABL_PHILOX_ROUNDS=n   Philox rounds (product: 10)
//   -DRT_ABL_UNIFORM_53        53-bit uniforms by two conversions and an fma instead of the 52-bit mantissa fill
//   -DRT_ABL_PLAIN_FMA         Horner steps as the compiler's fma (two-address v_fmac_f64 + copy)
//   -DRT_ABL_SQRT_GENERAL      the general fp64 square root instead of the unscaled sequence
//   -DRT_ABL_DIV_GENERAL       the general fp64 division instead of the unscaled sequence
//   -DRT_ABL_WAVE_TRACE        per-wave start / last-chunk / end clocks of the last walk launch (tools/wave_trace.py)
//   RT_AMD_CHUNK=thin,thick    (environment, read at load) chunk lengths for atmospheres below / above tau = 0.5
//   RT_AMD_ILV=n               (environment) chunks per resident wave from which the queue is walked in interleaved rows
//   RT_AMD_TAIL=per_wave,div   (environment) length of the ordered tail of the queue
//   RT_AMD_MIX=tau,beta        (environment) heavy / light alternation of the queue instead of the rows
//   RT_AMD_QUORUM=n            (environment) scattering lanes a wave gathers before it scatters (0 = never wait)
#pragma once
#include <cstdio>
#include <cstdlib>

namespace rt {

struct Ablation {
#ifdef RT_ABL_PHILOX_ROUNDS
    static constexpr int kPhiloxRounds = RT_ABL_PHILOX_ROUNDS;
#else
    static constexpr int kPhiloxRounds = 10;
#endif
#ifdef RT_ABL_UNIFORM_53
    static constexpr bool kUniform53 = true;
#else
    static constexpr bool kUniform53 = false;
#endif
#ifdef RT_ABL_PLAIN_FMA
    static constexpr bool kPlainFma = true;
#else
    static constexpr bool kPlainFma = false;
#endif
#ifdef RT_ABL_SQRT_GENERAL
    static constexpr bool kSqrtGeneral = true;
#else
    static constexpr bool kSqrtGeneral = false;
#endif
#ifdef RT_ABL_DIV_GENERAL
    static constexpr bool kDivGeneral = true;
#else
    static constexpr bool kDivGeneral = false;
#endif
#ifdef RT_ABL_WAVE_TRACE
    static constexpr bool kWaveTrace = true;
#else
    static constexpr bool kWaveTrace = false;
#endif
    static constexpr int kAbiTag = 0x1000;
    // chunk lengths from the environment; false = keep the product's
    static bool chunk_rule(unsigned& thin, unsigned& thick) {
        const char* e = getenv("RT_AMD_CHUNK");
        unsigned a = 0, b = 0;
        if (!e || sscanf(e, "%u,%u", &a, &b) != 2 || a < 64 || b < 64 || a > 65536 || b > 65536) return false;
        thin = a;
        thick = b;
        return true;
    }
    // RT_AMD_ILV=n (environment): items per resident wave from which the queue is walked in interleaved rows
    static int interleave_min(int product) {
        const char* e = getenv("RT_AMD_ILV");
        const int v = e ? atoi(e) : -1;
        return v >= 0 ? v : product;
    }
    // RT_AMD_TAIL=per_wave,cap_div (environment): the cheapest per_wave chunks per resident wave stay in order at the end
    // of the queue, at most items / cap_div of them
    static void tail_rule(unsigned& per_wave, unsigned& cap_div) {
        const char* e = getenv("RT_AMD_TAIL");
        unsigned a = 0, b = 0;
        if (e && sscanf(e, "%u,%u", &a, &b) == 2 && b >= 1) {
            per_wave = a;
            cap_div = b;
        }
    }
    // RT_AMD_MIX=tau_split,beta (environment): chunks of atmospheres with tau >= tau_split alternate with the share beta
    // of the chunks of the thinner ones; the rest of the thin chunks follows in order
    static bool mix_rule(double& tau_split, double& beta) {
        const char* e = getenv("RT_AMD_MIX");
        double a = 0, b = 0;
        if (!e || sscanf(e, "%lf,%lf", &a, &b) != 2 || !(b > 0.0 && b <= 1.0)) return false;
        tau_split = a;
        beta = b;
        return true;
    }
    // RT_AMD_QUORUM=n (environment): how many scattering lanes a wave gathers before it runs the scattering code
    static int scatter_quorum(int product) {
        const char* e = getenv("RT_AMD_QUORUM");
        const int v = e ? atoi(e) : -1;
        return v >= 0 && v <= 64 ? v : product;
    }
};

}  // namespace rt
