// prior_mc.h — estimateDispersionsPriorVar for residual degrees of freedom <= 3 (the reference's own 2v2 design;
// `~1` with 4 samples in the theta grid).  Reached from chicdiff.R:1573/1602/1643/1673 through DESeq2
// estimateDispersions -> estimateDispersionsMAP.
//
// With m - p <= 3 DESeq2 matches the prior variance by simulation, under a FIXED seed (it saves .Random.seed, calls
// set.seed(2) and restores it), so its value is deterministic:
//     obsDist   <- residuals inside (-10, 10);  obsHist <- hist(obsDist, breaks = -20:20/2)$density
//     for x in seq(0, 8, length = 200):
//         randDist <- log(rchisq(1e4, df = m - p)) + rnorm(1e4, 0, sqrt(x)) - log(m - p)     (inside (-10, 10))
//         kl[x]    <- sum(obs * (log(obs + small) - log(rand + small))),  small = min positive density of both
//     lofit <- loess(kl ~ x, span = .2);  argminKL <- (seq(0, 8, length = 1000))[which.min(predict(lofit, .))]
//     dispPriorVar <- max(argminKL, 0.25)
// The simulated densities are therefore constants per d.f.: built once per process on the host from R's own
// stream (r_rng.h) in R's draw order (1e4 rchisq, then 1e4 rnorm — none when sqrt(x) == 0), binned as hist.default
// does (right-closed, breaks shifted by 1e-7 * median(diff(breaks))).  loess(span = .2, degree = 2,
// surface = "interpolate") is a linear operator of kl: a k-d tree over the 200 grid values (cells cut between their
// two middle points until <= floor(200 * .2 * .2) = 8 points; bounding box widened by 0.5 %), at each of its 33
// vertices a local quadratic tricube fit over the 40 nearest grid values giving value and slope, cubic Hermite
// blending inside a cell.  The vertex operator (2 x 40 coefficients per vertex) is built once on the host; the
// device applies it (prior_mc_kernel in global_kernels.hip).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define PMC_HD __host__ __device__ inline
#else
#define PMC_HD inline
#endif

#include "r_rng.h"

namespace cd {

constexpr int kPmcBins = 40, kPmcGrid = 200, kPmcFine = 1000, kPmcDraws = 10000;
constexpr int kPmcNear = 40;   // q = floor(200 * 0.2 + 1e-5) nearest grid values per local fit
constexpr int kPmcLeaf = 8;    // fc = floor(200 * (0.2 * 0.2)) points per k-d leaf
constexpr int kPmcMaxVert = 40;

// bin of hist(breaks = -20:20/2) for x inside (-10, 10), else -1.  hist.default moves every break up by
// diddle = 1e-7 * median(diff(breaks)) = 5e-8 (the first one down) before counting right-closed bins.
PMC_HD int pmc_bin(double x) {
    if (!(x > -10.0 && x < 10.0)) return -1;
    int b = (int)floor((x + 10.0) * 2.0);  // [brk_b, brk_b+1)
    if (b > kPmcBins - 1) b = kPmcBins - 1;
    // right-closed with fuzz: x belongs to bin b when brk_b + 5e-8 < x <= brk_b+1 + 5e-8
    if (b > 0 && !(x > (b - 20) * 0.5 + 5e-8)) b--;
    return b;
}

struct PmcTable {
    double dens[kPmcGrid][kPmcBins];  // density of the simulated residuals per grid variance
    // loess operator: vertex v of the k-d tree sits at vert[v]; its local fit reads kl[lo[v] .. lo[v] + 40)
    int32_t nvert, lo[kPmcMaxVert];
    double vert[kPmcMaxVert];
    double val[kPmcMaxVert][kPmcNear], slope[kPmcMaxVert][kPmcNear];
};

PMC_HD double pmc_grid_x(int g) { return g == kPmcGrid - 1 ? 8.0 : g * (8.0 / (kPmcGrid - 1)); }  // seq(0, 8, length = 200)
PMC_HD double pmc_fine_x(int f) { return f == kPmcFine - 1 ? 8.0 : f * (8.0 / (kPmcFine - 1)); }

inline void pmc_split(int l, int u, double *vert, int &nv) {  // 1-based inclusive range of grid values
    if (u - l + 1 <= kPmcLeaf) return;
    const int m = (l + u) / 2;
    vert[nv++] = (pmc_grid_x(m - 1) + pmc_grid_x(m)) / 2.0;
    pmc_split(l, m, vert, nv);
    pmc_split(m + 1, u, vert, nv);
}

inline void pmc_build_loess(PmcTable &t) {
    int nv = 0;
    const double margin = 0.005 * 8.0;
    t.vert[nv++] = 0.0 - margin;
    t.vert[nv++] = 8.0 + margin;
    pmc_split(1, kPmcGrid, t.vert, nv);
    for (int i = 1; i < nv; i++)  // insertion sort
        for (int j = i; j > 0 && t.vert[j] < t.vert[j - 1]; j--) {
            const double s = t.vert[j];
            t.vert[j] = t.vert[j - 1];
            t.vert[j - 1] = s;
        }
    t.nvert = nv;
    for (int v = 0; v < nv; v++) {
        const long double s = t.vert[v];
        int lo = 0;  // window of the 40 nearest grid values
        while (lo + kPmcNear < kPmcGrid && fabsl(pmc_grid_x(lo + kPmcNear) - s) < fabsl(pmc_grid_x(lo) - s)) lo++;
        t.lo[v] = lo;
        long double h = 0;
        for (int k = 0; k < kPmcNear; k++) h = fmaxl(h, fabsl(pmc_grid_x(lo + k) - s));
        // weighted moments of u = (x - s) / h, then rows 0 and 1 of (X'WX)^-1 X'W
        long double M[5] = {0, 0, 0, 0, 0}, w[kPmcNear], u[kPmcNear];
        for (int k = 0; k < kPmcNear; k++) {
            u[k] = (pmc_grid_x(lo + k) - s) / h;
            const long double c = 1.0L - fabsl(u[k]) * u[k] * u[k];
            w[k] = c > 0 ? c * c * c : 0.0L;
            long double pw = w[k];
            for (int e = 0; e < 5; e++, pw *= u[k]) M[e] += pw;
        }
        const long double a = M[0], b = M[1], c = M[2], d = M[3], e = M[4];
        const long double det = a * (c * e - d * d) - b * (b * e - d * c) + c * (b * d - c * c);
        // inverse of [[a b c][b c d][c d e]], rows 0 and 1
        const long double i00 = (c * e - d * d) / det, i01 = (c * d - b * e) / det, i02 = (b * d - c * c) / det;
        const long double i11 = (a * e - c * c) / det, i12 = (b * c - a * d) / det;
        for (int k = 0; k < kPmcNear; k++) {
            t.val[v][k] = (double)(w[k] * (i00 + i01 * u[k] + i02 * u[k] * u[k]));
            t.slope[v][k] = (double)(w[k] * (i01 + i11 * u[k] + i12 * u[k] * u[k]) / h);
        }
    }
}

inline void pmc_build(int df, PmcTable &t) {
    RStream rng(2u);  // set.seed(2)
    const double ldf = log((double)df);
    static thread_local double chi[kPmcDraws];
    for (int g = 0; g < kPmcGrid; g++) {
        const double sd = sqrt(pmc_grid_x(g));
        double cnt[kPmcBins] = {0};
        int inside = 0;
        for (int k = 0; k < kPmcDraws; k++) chi[k] = log(rng.chisq((double)df));
        for (int k = 0; k < kPmcDraws; k++) {
            const double z = sd == 0.0 ? 0.0 : 0.0 + sd * rng.norm();  // rnorm(mu, 0) returns mu without drawing
            const int b = pmc_bin(chi[k] + z - ldf);
            if (b >= 0) { cnt[b] += 1; inside++; }
        }
        for (int b = 0; b < kPmcBins; b++) t.dens[g][b] = inside ? cnt[b] / (inside * 0.5) : 0.0;
    }
    pmc_build_loess(t);
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
// value and slope of the local fit at vertex v
PMC_HD void pmc_vertex(const PmcTable &t, int v, const double *kl, double &val, double &slope) {
    double a = 0, b = 0;
    const double *y = kl + t.lo[v];
    for (int k = 0; k < kPmcNear; k++) {
        a += t.val[v][k] * y[k];
        b += t.slope[v][k] * y[k];
    }
    val = a;
    slope = b;
}
// predict(lofit, x) from the vertex values: cubic Hermite inside the cell holding x
PMC_HD double pmc_loess_eval(const double *vert, int nvert, const double *val, const double *slope, double x) {
    int c = 0;
    while (c < nvert - 2 && x > vert[c + 1]) c++;
    const double w = vert[c + 1] - vert[c], h = (x - vert[c]) / w;
    const double phi0 = (1 - h) * (1 - h) * (1 + 2 * h), phi1 = h * h * (3 - 2 * h);
    const double psi0 = h * (1 - h) * (1 - h), psi1 = -h * h * (1 - h);
    return phi0 * val[c] + phi1 * val[c + 1] + (psi0 * slope[c] + psi1 * slope[c + 1]) * w;
}

// obs_counts[kPmcBins]: histogram of the observed residuals inside (-10, 10).  Returns max(argminKL, 0.25)
// (NaN when there are no residuals).  Host form; the library runs the same pieces in a kernel (prior_mc_kernel).
inline double pmc_prior_var(const double *obs_counts, const PmcTable &t) {
    double nobs = 0;
    for (int b = 0; b < kPmcBins; b++) nobs += obs_counts[b];
    if (!(nobs > 0)) return NAN;
    double obs[kPmcBins], kl[kPmcGrid], val[kPmcMaxVert], slope[kPmcMaxVert];
    for (int b = 0; b < kPmcBins; b++) obs[b] = obs_counts[b] / (nobs * 0.5);
    for (int g = 0; g < kPmcGrid; g++) kl[g] = pmc_kl(obs, t.dens[g]);
    for (int v = 0; v < t.nvert; v++) pmc_vertex(t, v, kl, val[v], slope[v]);
    double best = INFINITY, arg = 0;
    for (int f = 0; f < kPmcFine; f++) {
        const double fit = pmc_loess_eval(t.vert, t.nvert, val, slope, pmc_fine_x(f));
        if (fit < best) { best = fit; arg = pmc_fine_x(f); }  // which.min: the first minimum
    }
    return arg > 0.25 ? arg : 0.25;
}

}  // namespace cd
