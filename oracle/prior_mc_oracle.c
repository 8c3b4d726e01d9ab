/*
 * oracle/prior_mc_oracle.c — TEST INFRASTRUCTURE (parity oracle), not product code.
 *
 * DESeq2 estimateDispersionsPriorVar for residual degrees of freedom m - p <= 3 (reached
 * from chicdiff.R:1573/1602/1643/1673 for the reference's own 2v2 design and for `~1` with 4 samples):
 *
 *     set.seed(2)                                           # DESeq2 saves and restores .Random.seed around this
 *     obsDist <- dispResiduals[aboveMinDisp];  brks <- -20:20/2
 *     obsDist <- obsDist[obsDist > min(brks) & obsDist < max(brks)]
 *     obsVarGrid <- seq(from=0, to=8, length=200)
 *     obsDistHist <- hist(obsDist, breaks=brks, plot=FALSE)
 *     klDivs <- sapply(obsVarGrid, function(x) {
 *         randDist <- log(rchisq(1e4, df=(m-p))) + rnorm(1e4, 0, sqrt(x)) - log(m - p)
 *         randDist <- randDist[randDist > min(brks) & randDist < max(brks)]
 *         randDistHist <- hist(randDist, breaks=brks, plot=FALSE)
 *         z <- c(obsDistHist$density, randDistHist$density);  small <- min(z[z > 0])
 *         sum(obsDistHist$density * (log(obsDistHist$density + small) - log(randDistHist$density + small))) })
 *     lofit <- loess(klDivs ~ obsVarGrid, span=.2)
 *     obsVarFineGrid <- seq(from=0, to=8, length=1000)
 *     argminKL <- obsVarFineGrid[which.min(predict(lofit, obsVarFineGrid))]
 *     dispPriorVar <- pmax(argminKL, 0.25)
 *
 * The value is DETERMINISTIC (fixed seed); the simulated densities are constants per d.f.  Restated here:
 *   - the random stream: r_rng.c (R's Mersenne-Twister / inversion / rgamma chain; rnorm(n, 0, 0) draws nothing,
 *     so the first grid point consumes only the 1e4 rchisq draws);
 *   - hist(): right-closed bins on breaks + 1e-7 * median(diff(breaks)) "fuzz" (first break - fuzz), as
 *     hist.default does before C_BinCount;
 *   - loess(span=.2, degree=2, family="gaussian", surface="interpolate", cell=.2): k-d tree on the 200 grid
 *     values (cells split at the mean of the two middle points until <= floor(n*span*cell) = 8 points; bounding
 *     box widened by 0.5 %), local quadratic tricube fits (q = floor(n*span + 1e-5) = 40 nearest) at the cell
 *     vertices giving value and slope, cubic Hermite blending inside a cell (netlib dloess: ehg126/ehg129/ehg124
 *     build, ehg127 vertex fit, ehg128 evaluation).
 * PARITY: pinned down to R's published algorithms and known R outputs of the generators (tests/test_r_rng.py);
 * no DESeq2 2v2 golden value exists under /root/reference — tools/make_golden.R writes one wherever R exists.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

#define NB 40
#define NG 200
#define NF 1000
#define ND 10000
#define MAXV 64

/* bin of hist(x, breaks = -20:20/2) for x strictly inside (-10, 10), else -1.  Bin b is (fb[b], fb[b+1]] with the
 * fuzzy breaks fb[0] = -10 - 5e-8, fb[i] = brks[i] + 5e-8 */
int oracle_prior_mc_bin(double x) {
    if (!(x > -10.0 && x < 10.0)) return -1;
    const double diddle = 1e-7 * 0.5;
    int b = (int)ceil((x + 10.0) * 2.0) - 1; /* exact breaks: (brks[b], brks[b+1]] */
    if (b < 0) b = 0;
    if (b > NB - 1) b = NB - 1;
    /* a value within the fuzz above its lower break belongs to the bin below */
    while (b > 0 && !(x > (-20 + b) / 2.0 + diddle)) b--;
    while (b < NB - 1 && x > (-20 + b + 1) / 2.0 + diddle) b++;
    return b;
}

static double g_dens[4][NG][NB];
static int g_ready[4];

static void build(int df) {
    struct oracle_r_rng *rng = (struct oracle_r_rng *)malloc(oracle_r_rng_size());
    double *chi = (double *)malloc(sizeof(double) * ND);
    oracle_r_set_seed(rng, 2u);
    const double ldf = log((double)df), by = 8.0 / (NG - 1);
    for (int g = 0; g < NG; g++) {
        const double x = g == NG - 1 ? 8.0 : 0.0 + g * by, sd = sqrt(x);
        double cnt[NB];
        memset(cnt, 0, sizeof cnt);
        int inside = 0;
        for (int k = 0; k < ND; k++) chi[k] = log(oracle_r_rchisq(rng, (double)df));
        for (int k = 0; k < ND; k++) {
            /* rnorm(mu, sigma): sigma == 0 returns mu without a draw */
            const double z = sd == 0.0 ? 0.0 : 0.0 + sd * oracle_r_norm_rand(rng);
            const int b = oracle_prior_mc_bin(chi[k] + z - ldf);
            if (b >= 0) { cnt[b] += 1; inside++; }
        }
        for (int b = 0; b < NB; b++) g_dens[df][g][b] = inside ? cnt[b] / (inside * 0.5) : 0.0;
    }
    free(chi);
    free(rng);
    g_ready[df] = 1;
}

/* the simulated density table of one d.f. (200 x 40, row-major) — exported for tests */
int oracle_prior_mc_table(int df, double *out) {
    if (df < 1 || df > 3) return -1;
#pragma omp critical(oracle_prior_mc)
    if (!g_ready[df]) build(df);
    memcpy(out, g_dens[df], sizeof(double) * NG * NB);
    return 0;
}

/* ---- loess(y ~ x, span = .2, degree = 2, surface = "interpolate") on n sorted distinct x, evaluated at z[] ---- */
static int cmp_double(const void *a, const void *b) {
    const double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}
/* k-d tree cells (1-based inclusive point ranges): leaf when <= fc points, else cut between the middle points */
static void kd_split(const double *x, int l, int u, int fc, double *vert, int *nv) {
    if (u - l + 1 <= fc) return;
    const int m = (l + u) / 2;
    vert[(*nv)++] = (x[m - 1] + x[m]) / 2.0; /* (x_m + x_{m+1}) / 2, 1-based */
    kd_split(x, l, m, fc, vert, nv);
    kd_split(x, m + 1, u, fc, vert, nv);
}
/* local quadratic fit at s: value and slope (weighted least squares, tricube weights over the q nearest) */
static void vertex_fit(const double *x, const double *y, int n, int q, double s, double *val, double *slope) {
    /* x sorted: the q nearest are a window; slide it to the smallest reach */
    int lo = 0;
    while (lo + q < n && fabs(x[lo + q] - s) < fabs(x[lo] - s)) lo++;
    double rho = 0; /* squared distance of the q-th nearest */
    for (int k = lo; k < lo + q; k++) rho = fmax(rho, (x[k] - s) * (x[k] - s));
    /* modified Gram-Schmidt on the weighted columns 1, d, d^2 (sqrt-weights as in the Fortran) */
    double A[3][64], e[64];
    for (int k = 0; k < q; k++) {
        const double d = x[lo + k] - s, r = sqrt(d * d / rho), c = 1.0 - r * r * r;
        const double w = sqrt(c * c * c);
        A[0][k] = w; A[1][k] = w * d; A[2][k] = w * d * d;
        e[k] = w * y[lo + k];
    }
    double R[3][3] = {{0}}, qty[3];
    for (int j = 0; j < 3; j++) {
        for (int i = 0; i < j; i++) {
            double dot = 0;
            for (int k = 0; k < q; k++) dot += A[i][k] * A[j][k];
            R[i][j] = dot;
            for (int k = 0; k < q; k++) A[j][k] -= dot * A[i][k];
        }
        double nn = 0;
        for (int k = 0; k < q; k++) nn += A[j][k] * A[j][k];
        nn = sqrt(nn);
        R[j][j] = nn;
        for (int k = 0; k < q; k++) A[j][k] /= nn;
        double dot = 0;
        for (int k = 0; k < q; k++) dot += A[j][k] * e[k];
        qty[j] = dot;
        for (int k = 0; k < q; k++) e[k] -= dot * A[j][k];
    }
    const double b2 = qty[2] / R[2][2];
    const double b1 = (qty[1] - R[1][2] * b2) / R[1][1];
    const double b0 = (qty[0] - R[0][1] * b1 - R[0][2] * b2) / R[0][0];
    *val = b0;
    *slope = b1;
}
int oracle_loess_interp(const double *x, const double *y, int n, double span, double cell, const double *z, int nz,
                        double *out) {
    if (n > 4096 || n < 3) return -1;
    const int q = (int)fmin((double)n, floor(n * span + 1e-5)), fc = (int)floor(n * (span * cell));
    if (q < 3 || q > 64) return -1;
    double vert[MAXV * 8];
    int nv = 0;
    const double alpha = x[0], beta = x[n - 1];
    const double mu = 0.005 * fmax(beta - alpha, 1e-10 * fmax(fabs(alpha), fabs(beta)) + 1e-30);
    vert[nv++] = alpha - mu;
    vert[nv++] = beta + mu;
    kd_split(x, 1, n, fc, vert, &nv);
    qsort(vert, nv, sizeof(double), cmp_double);
    double val[MAXV * 8], slp[MAXV * 8];
    for (int v = 0; v < nv; v++) vertex_fit(x, y, n, q, vert[v], &val[v], &slp[v]);
    for (int i = 0; i < nz; i++) {
        int c = 0; /* cell [vert[c], vert[c+1]]; z <= split goes low */
        while (c < nv - 2 && z[i] > vert[c + 1]) c++;
        const double w = vert[c + 1] - vert[c], h = (z[i] - vert[c]) / w;
        const double phi0 = (1 - h) * (1 - h) * (1 + 2 * h), phi1 = h * h * (3 - 2 * h);
        const double psi0 = h * (1 - h) * (1 - h), psi1 = -h * h * (1 - h);
        out[i] = phi0 * val[c] + phi1 * val[c + 1] + (psi0 * slp[c] + psi1 * slp[c + 1]) * w;
    }
    return nv;
}

/* obs_counts[40]: histogram (oracle_prior_mc_bin) of the log dispersion residuals; df = m - p in 1..3 */
double oracle_prior_var_mc(const double *obs_counts, int df) {
    if (df < 1 || df > 3) return NAN;
#pragma omp critical(oracle_prior_mc)
    if (!g_ready[df]) build(df);
    double nobs = 0;
    for (int b = 0; b < NB; b++) nobs += obs_counts[b];
    if (!(nobs > 0)) return NAN;
    double obs[NB], kl[NG], xs[NG], zs[NF], fit[NF];
    for (int b = 0; b < NB; b++) obs[b] = obs_counts[b] / (nobs * 0.5);
    for (int g = 0; g < NG; g++) {
        xs[g] = g == NG - 1 ? 8.0 : g * (8.0 / (NG - 1));
        double small = INFINITY;
        for (int b = 0; b < NB; b++) {
            if (obs[b] > 0 && obs[b] < small) small = obs[b];
            if (g_dens[df][g][b] > 0 && g_dens[df][g][b] < small) small = g_dens[df][g][b];
        }
        double s = 0;
        for (int b = 0; b < NB; b++) s += obs[b] * (log(obs[b] + small) - log(g_dens[df][g][b] + small));
        kl[g] = s;
    }
    for (int f = 0; f < NF; f++) zs[f] = f == NF - 1 ? 8.0 : f * (8.0 / (NF - 1));
    oracle_loess_interp(xs, kl, NG, 0.2, 0.2, zs, NF, fit);
    double best = INFINITY, arg = 0;
    for (int f = 0; f < NF; f++)
        if (fit[f] < best) { best = fit[f]; arg = zs[f]; } /* which.min: the first minimum */
    return arg > 0.25 ? arg : 0.25;
}
