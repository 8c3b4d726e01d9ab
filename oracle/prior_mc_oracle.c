/*
 * oracle/prior_mc_oracle.c — TEST INFRASTRUCTURE (parity oracle), not product code.
 *
 * estimateDispersionsPriorVar for residual degrees of freedom <= 3 (DESeq2): the prior
 * variance is the grid value x whose simulated residual distribution log(chisq_df) + N(0, x) - log(df) is closest
 * (KL over hist(breaks = -20:20/2) densities, loess-smoothed over the 200 grid values) to the observed one.
 * R draws from its unseeded session RNG there, so the reference's own value changes from run to run; this
 * restatement uses a fixed-seed xoshiro256++ stream (Box-Muller normals) and evaluates the loess fit directly
 * (local quadratic, tricube weights, 40 nearest of the 200 grid points).  PARITY UNPINNED, like the rest of the
 * DESeq2 boundary.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "oracle.h"

#define NB 40
#define NG 200
#define NF 1000
#define ND 10000

typedef struct { uint64_t s[4]; double spare; int has; } rng_t;
static uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static void rng_seed(rng_t *r, uint64_t seed) {
    for (int k = 0; k < 4; k++) {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        r->s[k] = z ^ (z >> 31);
    }
    r->spare = 0;
    r->has = 0;
}
static uint64_t rng_next(rng_t *r) {
    uint64_t *s = r->s;
    const uint64_t out = rotl64(s[0] + s[3], 23) + s[0], t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl64(s[3], 45);
    return out;
}
static double rng_unif(rng_t *r) { return ((double)(rng_next(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
static double rng_normal(rng_t *r) {
    if (r->has) { r->has = 0; return r->spare; }
    const double rad = sqrt(-2.0 * log(rng_unif(r))), a = 6.283185307179586476925 * rng_unif(r);
    r->spare = rad * sin(a);
    r->has = 1;
    return rad * cos(a);
}

int oracle_prior_mc_bin(double x) {
    if (!(x > -10.0 && x < 10.0)) return -1;
    int b = (int)ceil((x + 10.0) * 2.0) - 1;
    return b < 0 ? 0 : (b >= NB ? NB - 1 : b);
}

static double g_dens[4][NG][NB];
static int g_ready[4];
static void build(int df) {
    rng_t rng;
    rng_seed(&rng, 20190123ull * 1000003ull + (uint64_t)df);
    const double ldf = log((double)df);
    for (int g = 0; g < NG; g++) {
        const double sd = sqrt(8.0 * g / (NG - 1));
        double cnt[NB];
        memset(cnt, 0, sizeof cnt);
        int inside = 0;
        for (int k = 0; k < ND; k++) {
            double chi = 0;
            for (int q = 0; q < df; q++) { const double z = rng_normal(&rng); chi += z * z; }
            const int b = oracle_prior_mc_bin(log(chi) + sd * rng_normal(&rng) - ldf);
            if (b >= 0) { cnt[b] += 1; inside++; }
        }
        for (int b = 0; b < NB; b++) g_dens[df][g][b] = inside ? cnt[b] / (inside * 0.5) : 0.0;
    }
    g_ready[df] = 1;
}

/* obs_counts[40]: histogram of the log dispersion residuals inside (-10, 10); df = m - p in 1..3 */
double oracle_prior_var_mc(const double *obs_counts, int df) {
    if (df < 1 || df > 3) return NAN;
#pragma omp critical(oracle_prior_mc)
    if (!g_ready[df]) build(df);
    double nobs = 0;
    for (int b = 0; b < NB; b++) nobs += obs_counts[b];
    if (!(nobs > 0)) return NAN;
    double obs[NB], kl[NG], xs[NG];
    for (int b = 0; b < NB; b++) obs[b] = obs_counts[b] / (nobs * 0.5);
    for (int g = 0; g < NG; g++) {
        xs[g] = 8.0 * g / (NG - 1);
        double small = INFINITY;
        for (int b = 0; b < NB; b++) {
            if (obs[b] > 0 && obs[b] < small) small = obs[b];
            if (g_dens[df][g][b] > 0 && g_dens[df][g][b] < small) small = g_dens[df][g][b];
        }
        double s = 0;
        for (int b = 0; b < NB; b++) s += obs[b] * (log(obs[b] + small) - log(g_dens[df][g][b] + small));
        kl[g] = s;
    }
    const int q = (int)floor(NG * 0.2 + 1e-5);
    double best = INFINITY, arg = 0;
    for (int f = 0; f < NF; f++) {
        const double x0 = 8.0 * f / (NF - 1);
        int lo = (int)floor(x0 / (8.0 / (NG - 1))) - q / 2;
        if (lo < 0) lo = 0;
        if (lo > NG - q) lo = NG - q;
        while (lo > 0 && fabs(xs[lo - 1] - x0) < fabs(xs[lo + q - 1] - x0)) lo--;
        while (lo < NG - q && fabs(xs[lo + q] - x0) < fabs(xs[lo] - x0)) lo++;
        const double h = fmax(fabs(xs[lo] - x0), fabs(xs[lo + q - 1] - x0));
        double S0 = 0, S1 = 0, S2 = 0, S3 = 0, S4 = 0, T0 = 0, T1 = 0, T2 = 0;
        for (int k = lo; k < lo + q; k++) {
            const double d = xs[k] - x0, u = fabs(d) / h;
            if (u >= 1.0) continue;
            const double c = 1.0 - u * u * u, w = c * c * c;
            S0 += w; S1 += w * d; S2 += w * d * d; S3 += w * d * d * d; S4 += w * d * d * d * d;
            T0 += w * kl[k]; T1 += w * d * kl[k]; T2 += w * d * d * kl[k];
        }
        const double det = S0 * (S2 * S4 - S3 * S3) - S1 * (S1 * S4 - S3 * S2) + S2 * (S1 * S3 - S2 * S2);
        const double num = T0 * (S2 * S4 - S3 * S3) - S1 * (T1 * S4 - S3 * T2) + S2 * (T1 * S3 - S2 * T2);
        const double fit = num / det;
        if (fit < best) { best = fit; arg = x0; }
    }
    return arg > 0.25 ? arg : 0.25;
}
