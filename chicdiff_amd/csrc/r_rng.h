// r_rng.h — R's default random number stream (host side, plain C++), as far as DESeq2's estimateDispersionsPriorVar
// draws from it after its set.seed(2) (prior_mc.h): Mersenne-Twister with set.seed()'s scrambling, unif_rand(),
// norm_rand() by inversion, exp_rand() and rgamma().  R is an un-vendored dependency of the reference (SURVEY.md
// §8c): the algorithms are the published ones R implements, in R's draw order —
//   MT19937 (Matsumoto & Nishimura 1998) seeded through the LCG 69069 x + 1 after 50 warm-up rounds;
//   normal deviates as qnorm((floor(2^27 u1) + u2) / 2^27) with Wichura's AS 241 (PPND16);
//   exponential deviates by Ahrens & Dieter (1972); gamma deviates by Ahrens & Dieter GD (1982) for shape >= 1 and
//   GS (1974) below.
// Checked against R outputs everybody can reproduce (set.seed(1); runif/rnorm/rexp ...) through
// chicdiff_hip_r_random() in tests/test_r_rng.py.
#pragma once
#include <math.h>
#include <stdint.h>

namespace cd {

class RStream {
    static constexpr int kN = 624, kM = 397;
    uint32_t mt_[kN];
    int pos_ = kN;
    // what rgamma() keeps between calls
    double g_a1_ = 0, g_a2_ = 0, g_s_ = 0, g_s2_ = 0, g_d_ = 0, g_q0_ = 0, g_b_ = 0, g_si_ = 0, g_c_ = 0;

    void refill() {
        auto twist = [](uint32_t hi, uint32_t lo, uint32_t far) {
            const uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
            return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        };
        for (int k = 0; k < kN - kM; k++) mt_[k] = twist(mt_[k], mt_[k + 1], mt_[k + kM]);
        for (int k = kN - kM; k < kN - 1; k++) mt_[k] = twist(mt_[k], mt_[k + 1], mt_[k + kM - kN]);
        mt_[kN - 1] = twist(mt_[kN - 1], mt_[0], mt_[kM - 1]);
        pos_ = 0;
    }

  public:
    explicit RStream(uint32_t seed) { set_seed(seed); }
    void set_seed(uint32_t seed) {
        for (int j = 0; j < 50; j++) seed = 69069u * seed + 1u;  // initial scrambling
        seed = 69069u * seed + 1u;                               // the word R then overwrites with the position
        for (int j = 0; j < kN; j++) mt_[j] = seed = 69069u * seed + 1u;
        pos_ = kN;
        g_a1_ = g_a2_ = 0;
    }
    double unif() {  // inside (0, 1)
        if (pos_ >= kN) refill();
        uint32_t y = mt_[pos_++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        const double x = y * 2.3283064365386963e-10, half_ulp = 0.5 * 2.328306437080797e-10;
        if (x <= 0.0) return half_ulp;
        if (1.0 - x <= 0.0) return 1.0 - half_ulp;
        return x;
    }
    static double horner(const double *c, int n, double r) {  // c[0] r^(n-1) + ... + c[n-1]
        double v = c[0];
        for (int k = 1; k < n; k++) v = v * r + c[k];
        return v;
    }
    static double qnorm(double p) {  // AS 241, 0 < p < 1
        static const double A[] = {2509.0809287301226727, 33430.575583588128105, 67265.770927008700853,
                                   45921.953931549871457, 13731.693765509461125, 1971.5909503065514427,
                                   133.14166789178437745, 3.387132872796366608};
        static const double B[] = {5226.495278852545925, 28729.085735721942674, 39307.89580009271061,
                                   21213.794301586595867, 5394.1960214247511077, 687.1870074920579083,
                                   42.313330701600911252, 1.0};
        static const double C[] = {7.7454501427834140764e-4, 0.0227238449892691845833, 0.24178072517745061177,
                                   1.27045825245236838258, 3.64784832476320460504, 5.7694972214606914055,
                                   4.6303378461565452959, 1.42343711074968357734};
        static const double D[] = {1.05075007164441684324e-9, 5.475938084995344946e-4, 0.0151986665636164571966,
                                   0.14810397642748007459, 0.68976733498510000455, 1.6763848301838038494,
                                   2.05319162663775882187, 1.0};
        static const double E[] = {2.01033439929228813265e-7, 2.71155556874348757815e-5, 0.0012426609473880784386,
                                   0.026532189526576123093, 0.29656057182850489123, 1.7848265399172913358,
                                   5.4637849111641143699, 6.6579046435011037772};
        static const double F[] = {2.04426310338993978564e-15, 1.4215117583164458887e-7, 1.8463183175100546818e-5,
                                   7.868691311456132591e-4, 0.0148753612908506148525, 0.13692988092273580531,
                                   0.59983220655588793769, 1.0};
        const double q = p - 0.5;
        if (fabs(q) <= 0.425) {
            const double r = 0.180625 - q * q;
            return q * horner(A, 8, r) / horner(B, 8, r);
        }
        double r = sqrt(-log(q < 0 ? p : 1.0 - p));
        double v;
        if (r <= 5.0) {
            r -= 1.6;
            v = horner(C, 8, r) / horner(D, 8, r);
        } else {
            r -= 5.0;
            v = horner(E, 8, r) / horner(F, 8, r);
        }
        return q < 0 ? -v : v;
    }
    double norm() {
        const double big = 134217728.0;  // 2^27
        double u = unif();
        u = (double)(int)(big * u) + unif();
        return qnorm(u / big);
    }
    double expo() {
        // partial sums of log(2)^k / k!
        static const double Q[16] = {0.6931471805599453, 0.9333736875190459, 0.9888777961838675, 0.9984589039328340,
                                     0.9998292811061389, 0.9999833164100727, 0.9999985691438767, 0.9999998906925558,
                                     0.9999999924734159, 0.9999999995283275, 0.9999999999728814, 0.9999999999985598,
                                     0.9999999999999289, 0.9999999999999968, 0.9999999999999999, 1.0000000000000000};
        double a = 0, u = unif();
        while (u <= 0.0 || u >= 1.0) u = unif();
        for (u += u; u <= 1.0; u += u) a += Q[0];
        u -= 1.0;
        if (u <= Q[0]) return a + u;
        int i = 0;
        double umin = unif();
        do {
            const double v = unif();
            if (v < umin) umin = v;
            i++;
        } while (u > Q[i]);
        return a + umin * Q[0];
    }
    double gamma(double a, double scale) {  // a > 0, scale > 0
        if (a < 1) {                        // GS
            const double e = 1.0 + 0.36787944117144233 * a;
            double x;
            for (;;) {
                const double p = e * unif();
                if (p >= 1.0) {
                    x = -log((e - p) / a);
                    if (expo() >= (1.0 - a) * log(x)) break;
                } else {
                    x = exp(log(p) / a);
                    if (expo() >= x) break;
                }
            }
            return scale * x;
        }
        // GD
        static const double QC[7] = {2.424e-4, 2.4511e-4, -7.388e-5, 0.00144121, 0.00801191, 0.02083148, 0.04166669};
        static const double AC[7] = {0.1233795, -0.1367177, 0.1423657, -0.1662921, 0.2000062, -0.250003, 0.3333333};
        if (a != g_a1_) {
            g_a1_ = a;
            g_s2_ = a - 0.5;
            g_s_ = sqrt(g_s2_);
            g_d_ = 5.656854 - g_s_ * 12;
        }
        double t = norm();
        double x = g_s_ + 0.5 * t;
        const double first = x * x;
        if (t >= 0) return scale * first;  // immediate acceptance
        double u = unif();
        if (g_d_ * u <= t * t * t) return scale * first;  // squeeze acceptance
        if (a != g_a2_) {
            g_a2_ = a;
            const double r = 1 / a;
            g_q0_ = horner(QC, 7, r) * r;
            if (a <= 3.686) {
                g_b_ = 0.463 + g_s_ + 0.178 * g_s2_;
                g_si_ = 1.235;
                g_c_ = 0.195 / g_s_ - 0.079 + 0.16 * g_s_;
            } else if (a <= 13.022) {
                g_b_ = 1.654 + 0.0076 * g_s2_;
                g_si_ = 1.68 / g_s_ + 0.275;
                g_c_ = 0.062 / g_s_ + 0.024;
            } else {
                g_b_ = 1.77;
                g_si_ = 0.75;
                g_c_ = 0.1515 / g_s_;
            }
        }
        auto quotient = [&](double tt) {
            const double v = tt / (g_s_ + g_s_);
            if (fabs(v) <= 0.25) return g_q0_ + 0.5 * tt * tt * horner(AC, 7, v) * v;
            return g_q0_ - g_s_ * tt + 0.25 * tt * tt + (g_s2_ + g_s2_) * log(1.0 + v);
        };
        if (x > 0.0 && log(1.0 - u) <= quotient(t)) return scale * first;  // quotient acceptance
        for (;;) {  // double exponential rejection
            const double e = expo();
            u = unif();
            u = u + u - 1.0;
            t = u < 0.0 ? g_b_ - g_si_ * e : g_b_ + g_si_ * e;
            if (t >= -0.71874483771719) {
                const double q = quotient(t);
                if (q > 0.0 && g_c_ * fabs(u) <= expm1(q) * exp(e - 0.5 * t * t)) break;
            }
        }
        x = g_s_ + 0.5 * t;
        return scale * x * x;
    }
    double chisq(double df) { return gamma(df / 2.0, 2.0); }
};

}  // namespace cd
