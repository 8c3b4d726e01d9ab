/* locfit_oracle.c — TEST INFRASTRUCTURE (see oracle/README.md): DESeq2's local-regression dispersion trend.
 *
 * estimateDispersionsFit switches to fitType = "local" when the parametric fit fails (or on request):
 *     localDispersionFit: locfit(logDisps ~ logMeans, data = d[disps >= minDisp*10, ], weights = means)
 *     dispFit <- exp(predict(fit, data.frame(logMeans = log(baseMean))))
 * locfit (CRAN, C. Loader; not under /root/reference, un-vendored and unpinned like DESeq2 itself) with its defaults:
 *   alpha = 0.7  nearest-neighbour bandwidth: h(x) = the k-th smallest |x_i - x|, k = floor(0.7 n)   (nbhd / kordstat)
 *   deg = 2, kern = "tcub": local quadratic, weights w_i (1 - (|x_i - x| / h)^3)^3, Taylor basis 1, dx, dx^2/2:
 *                the fitted value and slope at x are the first two coefficients                         (fitfun / locfit)
 *   ev = rbox(cut = 0.8), type "tree": vertices at the two ends of the data range; a cell [l, r] is cut at its
 *                midpoint while (r - l) / min(h_l, h_r) > cut, a new vertex is fitted there          (atree_split / _grow)
 *   predict:     cubic Hermite interpolation of value and slope between the two vertices around x; beyond the data
 *                range the end vertex's value and slope continue linearly                             (atree_int / hermite2)
 * Restated from the published algorithm (Loader 1999, "Local Regression and Likelihood", and the package's C sources
 * as this author remembers them).  PARITY UNPINNED: no R here; tools/make_golden.R writes a fitType = "local" case
 * that tests/test_golden_deseq2.py compares with once it has been run where R + DESeq2 + locfit exist. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

static double kth_smallest(double *a, int64_t n, int64_t k) { /* 1-based k; reorders a (quickselect) */
    int64_t lo = 0, hi = n - 1, want = k - 1;
    while (lo < hi) {
        const double pivot = a[lo + (hi - lo) / 2];
        int64_t i = lo, j = hi;
        while (i <= j) {
            while (a[i] < pivot) i++;
            while (a[j] > pivot) j--;
            if (i <= j) { const double t = a[i]; a[i] = a[j]; a[j] = t; i++; j--; }
        }
        if (want <= j) hi = j; else if (want >= i) lo = i; else break;
    }
    return a[want];
}

/* local quadratic at xv: bandwidth, value, slope.  Returns 0 ok, 1 singular / too few points */
static int vertex_fit(const double *x, const double *y, const double *w, int64_t n, double alpha, double xv, double *scratch,
                      double *h_out, double *f_out, double *d_out) {
    int64_t k = (int64_t)((double)n * alpha);
    if (k < 1) return 1;
    double h;
    for (int64_t i = 0; i < n; i++) scratch[i] = fabs(x[i] - xv);
    if (k < n) h = kth_smallest(scratch, n, k);
    else {
        double mx = 0;
        for (int64_t i = 0; i < n; i++) mx = scratch[i] > mx ? scratch[i] : mx;
        h = mx * exp(log((double)k / (double)n));
    }
    if (!(h > 0)) return 1;
    long double S[5] = {0, 0, 0, 0, 0}, T[3] = {0, 0, 0};
    for (int64_t i = 0; i < n; i++) {
        const double dx = x[i] - xv, u = fabs(dx) / h;
        if (u >= 1.0) continue;
        const double c = 1.0 - u * u * u, ww = w[i] * (c * c * c);
        long double p = ww;
        for (int q = 0; q < 5; q++) {
            S[q] += p;
            if (q < 3) T[q] += p * y[i];
            p *= dx;
        }
    }
    /* normal equations in the basis 1, dx, dx^2/2 */
    long double A[3][4] = {{S[0], S[1], S[2] / 2, T[0]}, {S[1], S[2], S[3] / 2, T[1]}, {S[2] / 2, S[3] / 2, S[4] / 4, T[2] / 2}};
    for (int c = 0; c < 3; c++) { /* Gaussian elimination with partial pivoting */
        int piv = c;
        for (int r = c + 1; r < 3; r++)
            if (fabsl(A[r][c]) > fabsl(A[piv][c])) piv = r;
        if (!(fabsl(A[piv][c]) > 0)) return 1;
        if (piv != c)
            for (int q = 0; q < 4; q++) { const long double t = A[c][q]; A[c][q] = A[piv][q]; A[piv][q] = t; }
        for (int r = c + 1; r < 3; r++) {
            const long double m = A[r][c] / A[c][c];
            for (int q = c; q < 4; q++) A[r][q] -= m * A[c][q];
        }
    }
    long double b[3];
    for (int r = 2; r >= 0; r--) {
        long double s = A[r][3];
        for (int q = r + 1; q < 3; q++) s -= A[r][q] * b[q];
        b[r] = s / A[r][r];
    }
    *h_out = h;
    *f_out = (double)b[0];
    *d_out = (double)b[1];
    return 0;
}

typedef struct {
    const double *x, *y, *w;
    int64_t n;
    double alpha, cut, lo, hi, *scratch;
    oracle_locfit *fit;
    int err;
} lf_build;

static int add_vertex(lf_build *b, double xv) {
    oracle_locfit *f = b->fit;
    if (f->nv >= ORACLE_LOCFIT_MAXV) { b->err = 2; return -1; }
    const int v = f->nv;
    f->x[v] = xv;
    if (vertex_fit(b->x, b->y, b->w, b->n, b->alpha, xv, b->scratch, &f->h[v], &f->f[v], &f->d[v])) { b->err = 1; return -1; }
    f->nv++;
    return v;
}
static void grow(lf_build *b, int il, int ir) { /* atree_grow in one dimension */
    if (b->err) return;
    const oracle_locfit *f = b->fit;
    const double le = f->x[ir] - f->x[il];
    double hmin = 0;
    if (f->h[il] > 0) hmin = f->h[il];
    if (f->h[ir] > 0 && (hmin == 0 || f->h[ir] < hmin)) hmin = f->h[ir];
    const double score = hmin == 0 ? 2 * le / (b->hi - b->lo) : le / hmin;
    if (!(b->cut < score)) return;
    const int im = add_vertex(b, (f->x[il] + f->x[ir]) / 2);
    if (im < 0) return;
    grow(b, il, im);
    grow(b, im, ir);
}

int oracle_locfit_build(const double *x, const double *y, const double *w, int64_t n, double alpha, double cut, oracle_locfit *fit) {
    memset(fit, 0, sizeof *fit);
    if (n < 4) return 1;
    lf_build b = {x, y, w, n, alpha, cut, x[0], x[0], NULL, fit, 0};
    for (int64_t i = 1; i < n; i++) {
        if (x[i] < b.lo) b.lo = x[i];
        if (x[i] > b.hi) b.hi = x[i];
    }
    if (!(b.hi > b.lo)) return 1;
    b.scratch = (double *)malloc(sizeof(double) * (size_t)n);
    if (!b.scratch) return 3;
    const int il = add_vertex(&b, b.lo), ir = b.err ? -1 : add_vertex(&b, b.hi);
    if (!b.err) grow(&b, il, ir);
    free(b.scratch);
    if (b.err) return b.err;
    /* the leaves of a one-dimensional tree are the intervals between neighbouring vertices: keep them sorted */
    for (int i = 1; i < fit->nv; i++)
        for (int j = i; j > 0 && fit->x[j - 1] > fit->x[j]; j--) {
            double t;
#define SWP(a) t = fit->a[j]; fit->a[j] = fit->a[j - 1]; fit->a[j - 1] = t
            SWP(x); SWP(h); SWP(f); SWP(d);
#undef SWP
        }
    return 0;
}

double oracle_locfit_eval(const oracle_locfit *fit, double x) {
    /* atree_int: at every cut the point goes left when x < midpoint, so x lands in [x_j, x_j+1) (ends: the outer cells) */
    int j = 0;
    while (j + 2 < fit->nv && x >= fit->x[j + 1]) j++;
    const double z = fit->x[j + 1] - fit->x[j], t = (x - fit->x[j]) / z;
    double p0, p1, p2, p3; /* hermite2 */
    if (t < 0) { p0 = 1; p1 = 0; p2 = t; p3 = 0; }
    else if (t > 1) { p0 = 0; p1 = 1; p2 = 0; p3 = t - 1; }
    else { p1 = t * t * (3 - 2 * t); p0 = 1 - p1; p2 = t * (1 - t) * (1 - t); p3 = t * t * (t - 1); }
    return p0 * fit->f[j] + p1 * fit->f[j + 1] + (p2 * fit->d[j] + p3 * fit->d[j + 1]) * z;
}

/* localDispersionFit on the rows DESeq2 uses (dispGeneEst > 100 minDisp): returns 0 and the fit, or non-zero */
int oracle_local_dispersion_fit(const double *means, const double *disps, int64_t n, oracle_locfit *fit) {
    double *lx = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1)), *ly = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (!lx || !ly) { free(lx); free(ly); return 3; }
    for (int64_t i = 0; i < n; i++) { lx[i] = log(means[i]); ly[i] = log(disps[i]); }
    const int rc = oracle_locfit_build(lx, ly, means, n, 0.7, 0.8, fit);
    free(lx);
    free(ly);
    return rc;
}
