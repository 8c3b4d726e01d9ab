// prior_mc.h — estimateDispersionsPriorVar for residual degrees of freedom <= 3 (host side, plain C++).
//
// DESeq2: with m - p <= 3 the trigamma approximation of the sampling variance of the log
// dispersion residuals is poor, so the prior variance is matched by simulation instead:
//     obsDist   <- residuals inside (-10, 10);  obsHist <- hist(obsDist, breaks = -20:20/2)$density
//     for x in seq(0, 8, length = 200):
//         randDist <- log(rchisq(1e4, df = m - p)) + rnorm(1e4, 0, sqrt(x)) - log(m - p)     (inside (-10, 10))
//         kl[x]    <- sum(obs * (log(obs + small) - log(rand + small))),  small = min positive density of both
//     lofit <- loess(kl ~ x, span = .2);  argminKL <- (seq(0, 8, length = 1000))[which.min(predict(lofit, .))]
//     dispPriorVar <- max(argminKL, 0.25)
// R draws from its session RNG, which Chicdiff never seeds: the reference itself gives a slightly different value
// on every run.  Here the draws come from a fixed-seed xoshiro256++ stream, so the simulated histograms are
// constants per df (built once per process) and the result is reproducible; loess is evaluated directly (local
// quadratic, tricube weights, the q = 40 nearest of the 200 grid points) instead of through R's kd-tree
// interpolation of that same local fit.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define PMC_HD __host__ __device__ inline
#else
#define PMC_HD inline
#endif

namespace cd {

constexpr int kPmcBins = 40, kPmcGrid = 200, kPmcFine = 1000, kPmcDraws = 10000;

struct PmcRng {  // xoshiro256++ seeded through splitmix64
    uint64_t s[4];
    explicit PmcRng(uint64_t seed) {
        for (int k = 0; k < 4; k++) {
            seed += 0x9E3779B97F4A7C15ull;
            uint64_t z = seed;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            s[k] = z ^ (z >> 31);
        }
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[0] + s[3], 23) + s[0], t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double unif() { return ((double)(next() >> 11) + 0.5) * (1.0 / 9007199254740992.0); }  // (0, 1)
    double spare = 0;
    bool has_spare = false;
    double normal() {  // Box-Muller
        if (has_spare) { has_spare = false; return spare; }
        const double r = sqrt(-2.0 * log(unif())), a = 6.283185307179586476925 * unif();
        spare = r * sin(a);
        has_spare = true;
        return r * cos(a);
    }
};

// bin of hist(breaks = -20:20/2) (right-closed) for x inside (-10, 10), else -1
inline int pmc_bin(double x) {
    if (!(x > -10.0 && x < 10.0)) return -1;
    int b = (int)ceil((x + 10.0) * 2.0) - 1;
    return b < 0 ? 0 : (b >= kPmcBins ? kPmcBins - 1 : b);
}

struct PmcTable {
    double dens[kPmcGrid][kPmcBins];  // density of the simulated residuals per grid variance
};
inline void pmc_build(int df, PmcTable &t) {
    PmcRng rng(20190123ull * 1000003ull + (uint64_t)df);
    const double ldf = log((double)df);
    for (int g = 0; g < kPmcGrid; g++) {
        const double sd = sqrt(8.0 * g / (kPmcGrid - 1));
        double cnt[kPmcBins] = {0};
        int inside = 0;
        for (int k = 0; k < kPmcDraws; k++) {
            double chi = 0;
            for (int q = 0; q < df; q++) { const double z = rng.normal(); chi += z * z; }
            const int b = pmc_bin(log(chi) + sd * rng.normal() - ldf);
            if (b >= 0) { cnt[b] += 1; inside++; }
        }
        for (int b = 0; b < kPmcBins; b++) t.dens[g][b] = inside ? cnt[b] / (inside * 0.5) : 0.0;
    }
}

// KL divergence of the observed density from the simulated one of grid point g (DESeq2's `small` included)
PMC_HD double pmc_kl(const double *obs, const double *dens_g) {
    double small = INFINITY;
    for (int b = 0; b < kPmcBins; b++) {
        if (obs[b] > 0 && obs[b] < small) small = obs[b];
        if (dens_g[b] > 0 && dens_g[b] < small) small = dens_g[b];
    }
    double s = 0;
    for (int b = 0; b < kPmcBins; b++) s += obs[b] * (log(obs[b] + small) - log(dens_g[b] + small));
    return s;
}
PMC_HD double pmc_grid_x(int g) { return 8.0 * g / (kPmcGrid - 1); }
PMC_HD double pmc_fine_x(int f) { return 8.0 * f / (kPmcFine - 1); }
// loess(kl ~ x, span = .2, degree = 2) at fine-grid point f, evaluated directly: local quadratic, tricube weights,
// the q = 40 nearest of the 200 grid points
PMC_HD double pmc_loess_at(int f, const double *kl) {
    const int q = (int)floor(kPmcGrid * 0.2 + 1e-5);
    const double x0 = pmc_fine_x(f);
    // the q nearest grid points form a window [lo, lo + q): slide it to the one with the smallest reach
    int lo = (int)floor(x0 / (8.0 / (kPmcGrid - 1))) - q / 2;
    if (lo < 0) lo = 0;
    if (lo > kPmcGrid - q) lo = kPmcGrid - q;
    while (lo > 0 && fabs(pmc_grid_x(lo - 1) - x0) < fabs(pmc_grid_x(lo + q - 1) - x0)) lo--;
    while (lo < kPmcGrid - q && fabs(pmc_grid_x(lo + q) - x0) < fabs(pmc_grid_x(lo) - x0)) lo++;
    const double h = fmax(fabs(pmc_grid_x(lo) - x0), fabs(pmc_grid_x(lo + q - 1) - x0));
    double S0 = 0, S1 = 0, S2 = 0, S3 = 0, S4 = 0, T0 = 0, T1 = 0, T2 = 0;
    for (int k = lo; k < lo + q; k++) {
        const double d = pmc_grid_x(k) - x0, u = fabs(d) / h;
        if (u >= 1.0) continue;
        const double c = 1.0 - u * u * u, w = c * c * c;
        S0 += w; S1 += w * d; S2 += w * d * d; S3 += w * d * d * d; S4 += w * d * d * d * d;
        T0 += w * kl[k]; T1 += w * d * kl[k]; T2 += w * d * d * kl[k];
    }
    // intercept of the weighted quadratic fit in d = x - x0 (Cramer's rule on the 3x3 normal equations)
    const double det = S0 * (S2 * S4 - S3 * S3) - S1 * (S1 * S4 - S3 * S2) + S2 * (S1 * S3 - S2 * S2);
    const double num = T0 * (S2 * S4 - S3 * S3) - S1 * (T1 * S4 - S3 * T2) + S2 * (T1 * S3 - S2 * T2);
    return num / det;
}

// obs_counts[kPmcBins]: histogram of the observed residuals inside (-10, 10).  Returns max(argminKL, 0.25)
// (NaN when there are no residuals).  Host form; the library runs the same pieces in a kernel (prior_mc_kernel).
inline double pmc_prior_var(const double *obs_counts, const PmcTable &t) {
    double nobs = 0;
    for (int b = 0; b < kPmcBins; b++) nobs += obs_counts[b];
    if (!(nobs > 0)) return NAN;
    double obs[kPmcBins], kl[kPmcGrid];
    for (int b = 0; b < kPmcBins; b++) obs[b] = obs_counts[b] / (nobs * 0.5);
    for (int g = 0; g < kPmcGrid; g++) kl[g] = pmc_kl(obs, t.dens[g]);
    double best = INFINITY, arg = 0;
    for (int f = 0; f < kPmcFine; f++) {
        const double fit = pmc_loess_at(f, kl);
        if (fit < best) { best = fit; arg = pmc_fine_x(f); }  // which.min: the first minimum
    }
    return arg > 0.25 ? arg : 0.25;
}

}  // namespace cd
