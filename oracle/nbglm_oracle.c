/*
 * oracle/nbglm_oracle.c — TEST INFRASTRUCTURE (parity oracle), not product code.
 *
 * CPU restatement of the NB-GLM arithmetic Chicdiff delegates to DESeq2 at
 * chicdiff.R:1557-1562 (estimateSizeFactors), :1573/:1602/:1643/:1673
 * (estimateDispersions) and :1574/:1603/:1644/:1674 (nbinomWaldTest).
 *
 * DESeq2 (Bioconductor, unpinned; the reference fixtures date it to 1.20-1.22) is NOT
 * under /root/reference, and R is not installed here: every function below restates
 * DESeq2's published algorithm (Love, Huber & Anders 2014, Genome Biology 15:550, and
 * the package's documented defaults) as summarised in SURVEY.md Appendix A; the
 * section tags (A1..A6) in the comments refer to that appendix.
 *
 * *** PARITY UNPINNED ***: /root/reference holds no DESeq2 input->output vectors for
 * this path (its only fixture, test_results.Rds, has outputs without inputs).  What
 * pins this file: (i) the fixture's stat->pvalue / lfc,lfcSE->stat relations,
 * (ii) first-principles property tests (score equations, APL stationarity checked
 * with mpmath), (iii) an independent numpy/scipy twin in tests/.  See oracle/README.md.
 *
 * Supported designs: ~group with two levels (p = 2; X = [1, g]) and ~1 (p = 1), the two
 * designs the reference builds (chicdiff.R:1557-1559 and :1629-1631).
 */
#include "oracle.h"
#include "rmath_lite.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXS 64
#define LOG2E 1.4426950408889634074

void oracle_nbglm_default_opts(oracle_nbglm_opts *o) {
    o->minDisp = 1e-8;
    o->dispTol = 1e-6;
    o->kappa0 = 1.0;
    o->maxit = 100;
    o->betaMaxit = 100;
    o->betaTol = 1e-8;
    o->minmu = 0.5;
    o->outlierSD = 2.0;
    o->dispPriorVar = NAN;
    o->nthreads = 1;
    o->_pad = 0;
    o->trendCoef[0] = o->trendCoef[1] = NAN;
    o->fitType = 0;
    o->noLocalSubstitute = 0;
    o->dispFitIn = NULL;
    o->varLogDispEsts = NAN;
    o->xim = NAN;
}

/* ---------------------------------------------------------------------------------- */
/* order statistics                                                                     */
static int cmp_double(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* R median(): mean of the two middle order statistics for even n */
double oracle_median(double *x, int64_t n) {
    if (n <= 0) return NAN;
    qsort(x, (size_t)n, sizeof(double), cmp_double);
    if (n & 1) return x[n / 2];
    return (x[n / 2 - 1] + x[n / 2]) / 2.0;
}

/* A1. estimateSizeFactorsForMatrix: loggeomeans = rowMeans(log(counts));
 * sf_j = exp(median((log(cnts_j) - loggeomeans)[is.finite(loggeomeans) & cnts_j > 0])) */
int oracle_size_factors(const int32_t *counts, int64_t n, int32_t S, double *sf) {
    double *lg = (double *)malloc(sizeof(double) * (size_t)n);
    double *buf = (double *)malloc(sizeof(double) * (size_t)n);
    if (!lg || !buf) { free(lg); free(buf); return -1; }
    int64_t nfinite = 0;
    for (int64_t i = 0; i < n; i++) {
        double s = 0;
        for (int j = 0; j < S; j++) s += log((double)counts[(int64_t)j * n + i]); /* log(0) = -Inf */
        lg[i] = s / S;
        if (isfinite(lg[i])) nfinite++;
    }
    if (nfinite == 0) { free(lg); free(buf); return -2; } /* "every gene contains at least one zero" */
    for (int j = 0; j < S; j++) {
        int64_t m = 0;
        for (int64_t i = 0; i < n; i++) {
            int32_t c = counts[(int64_t)j * n + i];
            if (isfinite(lg[i]) && c > 0) buf[m++] = log((double)c) - lg[i];
        }
        sf[j] = exp(oracle_median(buf, m));
    }
    free(lg);
    free(buf);
    return 0;
}

/* ---------------------------------------------------------------------------------- */
/* A2.6 fitDisp objective: Cox-Reid adjusted profile log-likelihood in a = log(alpha)    */

double oracle_log_posterior(double log_alpha, const double *y, const double *mu, const int32_t *group,
                            int32_t S, int32_t p, double prior_mean, double prior_sigmasq,
                            int32_t use_prior) {
    double alpha = exp(log_alpha);
    double wA = 0, wB = 0;
    for (int j = 0; j < S; j++) {
        double w = 1.0 / (1.0 / mu[j] + alpha);
        if (p == 2 && group[j]) wB += w; else wA += w;
    }
    /* det(X'WX): X=[1,g] -> (wA+wB)*wB - wB^2 = wA*wB ; X=[1] -> wA */
    double cr_term = -0.5 * log(p == 2 ? wA * wB : wA);
    double alpha_neg1 = 1.0 / alpha;
    double ll = 0;
    double lg_an1 = oracle_lgamma(alpha_neg1);
    for (int j = 0; j < S; j++)
        ll += oracle_lgamma(y[j] + alpha_neg1) - lg_an1 - y[j] * log(mu[j] + alpha_neg1) -
              alpha_neg1 * log(1.0 + mu[j] * alpha);
    double prior_part = 0;
    if (use_prior) {
        double d = log(alpha) - prior_mean;
        prior_part = -0.5 * d * d / prior_sigmasq;
    }
    return ll + prior_part + cr_term;
}

double oracle_dlog_posterior(double log_alpha, const double *y, const double *mu, const int32_t *group,
                             int32_t S, int32_t p, double prior_mean, double prior_sigmasq,
                             int32_t use_prior) {
    double alpha = exp(log_alpha);
    double wA = 0, wB = 0, dA = 0, dB = 0;
    for (int j = 0; j < S; j++) {
        double t = 1.0 / mu[j] + alpha;
        double w = 1.0 / t, dw = -1.0 / (t * t);
        if (p == 2 && group[j]) { wB += w; dB += dw; } else { wA += w; dA += dw; }
    }
    /* -0.5 * trace(B^-1 dB): for X=[1,g] this is dA/wA + dB/wB */
    double cr_term = -0.5 * (p == 2 ? dA / wA + dB / wB : dA / wA);
    double alpha_neg1 = 1.0 / alpha, alpha_neg2 = alpha_neg1 * alpha_neg1;
    double s = 0;
    double dg_an1 = oracle_digamma(alpha_neg1);
    for (int j = 0; j < S; j++) {
        double ma = mu[j] * alpha;
        s += dg_an1 + log(1.0 + ma) - ma / (1.0 + ma) - oracle_digamma(y[j] + alpha_neg1) +
             y[j] / (mu[j] + alpha_neg1);
    }
    double ll_part = alpha_neg2 * s;
    double prior_part = use_prior ? -1.0 * (log(alpha) - prior_mean) / prior_sigmasq : 0.0;
    return (ll_part + cr_term) * alpha + prior_part;
}

typedef struct {
    double log_alpha, initial_lp, last_lp;
    int iter;
} fitdisp_res;

/* A2.6: backtracking (Armijo) gradient ascent on a = log alpha */
static fitdisp_res fit_disp_row(const double *y, const double *mu, const int32_t *group, int S, int p,
                                double log_alpha0, double prior_mean, double prior_sigmasq, int use_prior,
                                double min_log_alpha, double kappa_0, double tol, int maxit) {
    const double epsilon = 1.0e-4;
    fitdisp_res r;
    double a = log_alpha0;
    double lp = oracle_log_posterior(a, y, mu, group, S, p, prior_mean, prior_sigmasq, use_prior);
    double dlp = oracle_dlog_posterior(a, y, mu, group, S, p, prior_mean, prior_sigmasq, use_prior);
    double kappa = kappa_0;
    int iter = 0, iter_accept = 0;
    r.initial_lp = lp;
    for (int t = 0; t < maxit; t++) {
        iter++;
        double a_propose = a + kappa * dlp;
        if (a_propose < -30.0) kappa = (-30.0 - a) / dlp;
        if (a_propose > 10.0) kappa = (10.0 - a) / dlp;
        double theta_kappa =
            -1.0 * oracle_log_posterior(a + kappa * dlp, y, mu, group, S, p, prior_mean, prior_sigmasq, use_prior);
        double theta_hat_kappa = -1.0 * lp - kappa * epsilon * dlp * dlp;
        if (theta_kappa <= theta_hat_kappa) {
            iter_accept++;
            a = a + kappa * dlp;
            double lpnew = oracle_log_posterior(a, y, mu, group, S, p, prior_mean, prior_sigmasq, use_prior);
            double change = lpnew - lp;
            if (change < tol) { lp = lpnew; break; }
            if (a < min_log_alpha) break;
            lp = lpnew;
            dlp = oracle_dlog_posterior(a, y, mu, group, S, p, prior_mean, prior_sigmasq, use_prior);
            kappa = fmin(kappa * 1.1, kappa_0);
            if (iter_accept % 5 == 0) kappa = kappa / 2.0;
        } else {
            kappa = kappa / 2.0;
        }
    }
    r.log_alpha = a;
    r.last_lp = lp;
    r.iter = iter;
    return r;
}

/* A2.7 fitDispGrid: 20-point coarse grid on [log 1e-8, log max(10,S)], then a 20-point
 * fine grid one coarse step either side of the argmax (first maximum wins) */
static double fit_disp_grid_row(const double *y, const double *mu, const int32_t *group, int S, int p,
                                double prior_mean, double prior_sigmasq, int use_prior) {
    const int G = 20;
    double lo = log(1e-8), hi = log(S > 10 ? (double)S : 10.0);
    double grid[20], best = -INFINITY;
    int ib = 0;
    for (int t = 0; t < G; t++) grid[t] = lo + (hi - lo) * t / (G - 1);
    double delta = grid[1] - grid[0];
    for (int t = 0; t < G; t++) {
        double v = oracle_log_posterior(grid[t], y, mu, group, S, p, prior_mean, prior_sigmasq, use_prior);
        if (v > best) { best = v; ib = t; }
    }
    double a_hat = grid[ib], flo = a_hat - delta, fhi = a_hat + delta, fbest = -INFINITY, fa = a_hat;
    for (int t = 0; t < G; t++) {
        double a = flo + (fhi - flo) * t / (G - 1);
        double v = oracle_log_posterior(a, y, mu, group, S, p, prior_mean, prior_sigmasq, use_prior);
        if (v > fbest) { fbest = v; fa = a; }
    }
    return exp(fa);
}

/* ---------------------------------------------------------------------------------- */
/* A3 parametricDispersionFit: glm(disps ~ I(1/means), Gamma(link="identity"), start=coefs)
 * inside an outer re-selection loop.  R's glm.fit IRLS restated for this family/link:
 * working response z = y, working weight w^2 = 1/mu^2, deviance 2*sum(-log(y/mu)+(y-mu)/mu),
 * epsilon 1e-8, maxit 25, convergence |dev-devold|/(|dev|+0.1) < epsilon. */

static double gamma_dev(const double *x, const double *y, const int64_t *idx, int64_t m, double c0, double c1,
                        int *valid) {
    double dev = 0;
    for (int64_t k = 0; k < m; k++) {
        int64_t i = idx[k];
        double mu = c0 + c1 * x[i];
        if (!(mu > 0) || !isfinite(mu)) { *valid = 0; return NAN; }
        dev += -2.0 * (log(y[i] == 0 ? 1.0 : y[i] / mu) - (y[i] - mu) / mu);
    }
    return dev;
}

int oracle_parametric_dispersion_fit(const double *means, const double *disps, int64_t n, double coefs[2],
                                     int32_t *outer_iter) {
    double c0 = 0.1, c1 = 1.0;
    int iter = 0, rc = 0;
    double *x = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    int64_t *good = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; i++) x[i] = 1.0 / means[i];
    for (;;) {
        int64_t m = 0;
        for (int64_t i = 0; i < n; i++) {
            double r = disps[i] / (c0 + c1 / means[i]);
            if (r > 1e-4 && r < 15) good[m++] = i;
        }
        if (m < 2) { rc = 3; break; }
        /* glm.fit */
        double b0 = c0, b1 = c1;
        int valid = 1, conv = 0;
        double devold = gamma_dev(x, disps, good, m, b0, b1, &valid);
        if (!valid) { rc = 1; break; }
        for (int it = 0; it < 25; it++) {
            double sw = 0, swx = 0, swxx = 0, swy = 0, swxy = 0;
            for (int64_t k = 0; k < m; k++) {
                int64_t i = good[k];
                double mu = b0 + b1 * x[i], w = 1.0 / (mu * mu);
                sw += w; swx += w * x[i]; swxx += w * x[i] * x[i];
                swy += w * disps[i]; swxy += w * x[i] * disps[i];
            }
            double det = sw * swxx - swx * swx;
            double nb0 = (swxx * swy - swx * swxy) / det, nb1 = (sw * swxy - swx * swy) / det;
            if (!isfinite(nb0) || !isfinite(nb1)) { conv = 0; break; }
            b0 = nb0; b1 = nb1;
            double dev = gamma_dev(x, disps, good, m, b0, b1, &valid);
            if (!valid) break; /* R would step-halve; DESeq2 then usually fails on coefs<=0 anyway */
            if (fabs(dev - devold) / (fabs(dev) + 0.1) < 1e-8) { conv = 1; break; }
            devold = dev;
        }
        if (!valid) { rc = 1; break; }
        double o0 = c0, o1 = c1;
        c0 = b0; c1 = b1;
        if (!(c0 > 0 && c1 > 0)) { rc = 1; break; } /* "parametric dispersion fit failed" */
        double l0 = log(c0 / o0), l1 = log(c1 / o1);
        if ((l0 * l0 + l1 * l1 < 1e-6) && conv) break;
        iter++;
        if (iter > 10) { rc = 2; break; } /* "dispersion fit did not converge" */
    }
    coefs[0] = c0;
    coefs[1] = c1;
    if (outer_iter) *outer_iter = iter;
    free(x);
    free(good);
    return rc;
}

/* ---------------------------------------------------------------------------------- */
/* A5 fitBeta for X=[1,g] — ridge IRLS; the (S+2)x2 QR solve of the augmented system is
 * replaced by its normal equations (X'WX + ridge) beta = X'Wz, solved in closed form.  */

typedef struct {
    double b0, b1, v0, v1, dev;
    int iter;
    double hat[MAXS];
} fitbeta_res;

static fitbeta_res fit_beta_row(const double *y, const double *nf, const int32_t *g, int S, double alpha,
                                double b0, double b1, double lambda, double tol, int maxit, double minmu) {
    fitbeta_res r;
    const double large = 30.0;
    double mu[MAXS];
    for (int j = 0; j < S; j++) mu[j] = fmax(nf[j] * exp(b0 + (g[j] ? b1 : 0.0)), minmu);
    double dev = 0, dev_old = 0;
    int iter = 0;
    for (int t = 0; t < maxit; t++) {
        iter++;
        double wA = 0, wB = 0, zA = 0, zB = 0;
        for (int j = 0; j < S; j++) {
            double w = mu[j] / (1.0 + alpha * mu[j]);
            double z = log(mu[j] / nf[j]) + (y[j] - mu[j]) / mu[j];
            if (g[j]) { wB += w; zB += w * z; } else { wA += w; zA += w * z; }
        }
        double m00 = wA + wB + lambda, m01 = wB, m11 = wB + lambda;
        double r0 = zA + zB, r1 = zB;
        double det = m00 * m11 - m01 * m01;
        b0 = (m11 * r0 - m01 * r1) / det;
        b1 = (m00 * r1 - m01 * r0) / det;
        if (fabs(b0) > large || fabs(b1) > large) { iter = maxit; break; }
        for (int j = 0; j < S; j++) mu[j] = fmax(nf[j] * exp(b0 + (g[j] ? b1 : 0.0)), minmu);
        dev = 0;
        for (int j = 0; j < S; j++) dev += -2.0 * oracle_dnbinom_mu_log(y[j], 1.0 / alpha, mu[j]);
        double conv_test = fabs(dev - dev_old) / (fabs(dev) + 0.1);
        if (isnan(conv_test)) { iter = maxit; break; }
        if (t > 0 && conv_test < tol) break;
        dev_old = dev;
    }
    /* covariance: sigma = M^-1 (X'WX) M^-1, M = X'WX + ridge; hat diag = w_j x_j' M^-1 x_j */
    double wA = 0, wB = 0, w[MAXS];
    for (int j = 0; j < S; j++) {
        w[j] = mu[j] / (1.0 + alpha * mu[j]);
        if (g[j]) wB += w[j]; else wA += w[j];
    }
    double m00 = wA + wB + lambda, m01 = wB, m11 = wB + lambda, det = m00 * m11 - m01 * m01;
    double i00 = m11 / det, i01 = -m01 / det, i11 = m00 / det;
    double a00 = wA + wB, a01 = wB, a11 = wB;
    /* T = Minv * A */
    double t00 = i00 * a00 + i01 * a01, t01 = i00 * a01 + i01 * a11;
    double t10 = i01 * a00 + i11 * a01, t11 = i01 * a01 + i11 * a11;
    r.v0 = t00 * i00 + t01 * i01;
    r.v1 = t10 * i01 + t11 * i11;
    for (int j = 0; j < S; j++) r.hat[j] = w[j] * (g[j] ? (i00 + 2 * i01 + i11) : i00);
    r.b0 = b0; r.b1 = b1; r.dev = dev; r.iter = iter;
    return r;
}

/* The IRLS of one row, step by step, WITHOUT the stopping rule: iterate t (1-based) leaves beta in b0[t-1], b1[t-1] (natural
 * log scale) and DESeq2's conv_test = |dev - dev_old| / (|dev| + 0.1) in conv[t-1].  The referee of tests/test_gpu_parity.py
 * for rows on which the GPU and fit_beta_row() stop after a different number of steps: the stop `conv_test < betaTol`
 * is a comparison of two rounded numbers, and a row whose conv_test sits within rounding of betaTol at step t ends at step
 * t in one double-precision implementation and at t + 1 in another (the estimates then differ by ~ sqrt(betaTol)). */
int oracle_irls_trace(const int32_t *counts, const double *nf, int64_t n, int32_t S, const int32_t *group, int64_t row,
                      double alpha, int32_t steps, double *b0_out, double *b1_out, double *conv_out) {
    if (S < 2 || S > MAXS || row < 0 || row >= n) return -1;
    double y[MAXS], nfr[MAXS], mu[MAXS];
    int32_t g[MAXS];
    int cs[2] = {0, 0};
    double lA = 0, lB = 0;
    for (int j = 0; j < S; j++) {
        g[j] = group[j] != 0;
        cs[g[j]]++;
        y[j] = (double)counts[(int64_t)j * n + row];
        nfr[j] = nf[(int64_t)j * n + row];
        const double l = log(y[j] / nfr[j] + 0.1);
        if (g[j]) lB += l; else lA += l;
    }
    if (!cs[0] || !cs[1]) return -2;
    const double lambda = 1e-6 / (M_LN2 * M_LN2);
    double b0 = lA / cs[0], b1 = lB / cs[1] - lA / cs[0], dev_old = 0;
    for (int j = 0; j < S; j++) mu[j] = fmax(nfr[j] * exp(b0 + (g[j] ? b1 : 0.0)), 0.5);
    for (int t = 0; t < steps; t++) {
        double wA = 0, wB = 0, zA = 0, zB = 0;
        for (int j = 0; j < S; j++) {
            double w = mu[j] / (1.0 + alpha * mu[j]);
            double z = log(mu[j] / nfr[j]) + (y[j] - mu[j]) / mu[j];
            if (g[j]) { wB += w; zB += w * z; } else { wA += w; zA += w * z; }
        }
        double m00 = wA + wB + lambda, m01 = wB, m11 = wB + lambda, r0 = zA + zB, r1 = zB, det = m00 * m11 - m01 * m01;
        b0 = (m11 * r0 - m01 * r1) / det;
        b1 = (m00 * r1 - m01 * r0) / det;
        for (int j = 0; j < S; j++) mu[j] = fmax(nfr[j] * exp(b0 + (g[j] ? b1 : 0.0)), 0.5);
        double dev = 0;
        for (int j = 0; j < S; j++) dev += -2.0 * oracle_dnbinom_mu_log(y[j], 1.0 / alpha, mu[j]);
        b0_out[t] = b0;
        b1_out[t] = b1;
        conv_out[t] = fabs(dev - dev_old) / (fabs(dev) + 0.1);
        dev_old = dev;
    }
    return 0;
}

/* A5 fallback.  DESeq2 re-fits rows whose IRLS did not converge with optim(method = "L-BFGS-B",
 * lower = -30, upper = 30) on the log2-scale negative log posterior (fitNbinomGLMsOptim):
 *   -sum_j dnbinom(k_j; mu = nf_j 2^(x_j p), size = 1/alpha, log) - sum_k dnorm(p_k; 0, sd = 1/sqrt(lambda), log).
 * L-BFGS-B itself (R's lbfgsb.c with finite-difference gradients) is not restated: what is
 * reproduced is its target, the posterior mode inside the box, found here by a damped Fisher-scoring
 * iteration with backtracking on the same objective (agrees with R's optimum to optim's own
 * tolerance, factr = 1e7).  Natural-log scale internally; returns 1 when a mode was reached. */
static double optim_objective(const double *y, const double *nf, const int32_t *g, int S, double alpha, double lam,
                              double b0, double b1) {
    double f = 0.5 * lam * (b0 * b0 + b1 * b1);
    for (int j = 0; j < S; j++) {
        double mu = nf[j] * exp(b0 + (g[j] ? b1 : 0.0));
        f -= oracle_dnbinom_mu_log(y[j], 1.0 / alpha, mu);
    }
    return f;
}
static int beta_optim_row(const double *y, const double *nf, const int32_t *g, int S, double alpha, double lam,
                          double *b0io, double *b1io) {
    const double bound = 30.0 * M_LN2; /* +-30 on the log2 scale */
    double b0 = *b0io, b1 = *b1io;
    double f = optim_objective(y, nf, g, S, alpha, lam, b0, b1);
    int converged = 0;
    for (int it = 0; it < 200 && !converged; it++) {
        double g0 = lam * b0, g1 = lam * b1, wA = 0, wB = 0;
        for (int j = 0; j < S; j++) {
            double mu = nf[j] * exp(b0 + (g[j] ? b1 : 0.0));
            double sc = (y[j] - mu) / (1.0 + alpha * mu), w = mu / (1.0 + alpha * mu);
            g0 -= sc;
            if (g[j]) { g1 -= sc; wB += w; } else wA += w;
        }
        double m00 = wA + wB + lam, m01 = wB, m11 = wB + lam, det = m00 * m11 - m01 * m01;
        double d0 = -(m11 * g0 - m01 * g1) / det, d1 = -(m00 * g1 - m01 * g0) / det;
        double t = 1.0;
        int moved = 0;
        for (int h = 0; h < 40; h++, t *= 0.5) {
            double n0 = fmin(fmax(b0 + t * d0, -bound), bound), n1 = fmin(fmax(b1 + t * d1, -bound), bound);
            double fn = optim_objective(y, nf, g, S, alpha, lam, n0, n1);
            if (fn < f) {
                if (f - fn < 1e-13 * (fabs(f) + 1.0)) converged = 1;
                b0 = n0; b1 = n1; f = fn; moved = 1;
                break;
            }
        }
        if (!moved) converged = 1; /* no descent left at working precision: at the mode (or pinned to the box) */
    }
    *b0io = b0;
    *b1io = b1;
    return converged;
}

/* R mean(x, trim): drop floor(n*trim) from each end of the sorted values */
static double trimmed_mean(double *v, int n, double trim) {
    qsort(v, (size_t)n, sizeof(double), cmp_double);
    if (trim >= 0.5) return (n & 1) ? v[n / 2] : (v[n / 2 - 1] + v[n / 2]) / 2.0;
    int lo = (int)floor(n * trim), hi = n - lo;
    double s = 0;
    for (int k = lo; k < hi; k++) s += v[k];
    return s / (hi - lo);
}

static int trim_bin(int n) { return n <= 3 ? 0 : (n <= 23 ? 1 : 2); } /* cut(n, c(0,3.5,23.5,Inf)) */

/* A5 Cook's: robustMethodOfMomentsDisp (trimmedCellVariance over cells with >=3 samples, floor 0.04) */
static double robust_mom_disp(const double *q, const int32_t *g, int S, const int *cellsize) {
    static const double trimratio[3] = {1.0 / 3, 1.0 / 4, 1.0 / 8};
    static const double scale_c[3] = {2.04, 1.86, 1.51};
    double vmax = -INFINITY, m = 0;
    for (int j = 0; j < S; j++) m += q[j];
    m /= S;
    for (int c = 0; c < 2; c++) {
        int nc = cellsize[c];
        if (nc < 3) continue;
        double tmp[MAXS];
        int k = 0;
        for (int j = 0; j < S; j++) if (g[j] == c) tmp[k++] = q[j];
        double tr = trimratio[trim_bin(nc)];
        double cm = trimmed_mean(tmp, nc, tr);
        k = 0;
        for (int j = 0; j < S; j++) if (g[j] == c) { double d = q[j] - cm; tmp[k++] = d * d; }
        double v = scale_c[trim_bin(nc)] * trimmed_mean(tmp, nc, tr);
        if (v > vmax) vmax = v;
    }
    double alpha = (vmax - m) / (m * m);
    return alpha > 0.04 ? alpha : 0.04;
}

/* ---------------------------------------------------------------------------------- */

#define SET(arr, i, v) do { if (out->arr) out->arr[i] = (v); } while (0)

int oracle_nbglm_fit(const int32_t *counts, const double *nf, int64_t n, int32_t S, const int32_t *group,
                     const oracle_nbglm_opts *opts_in, oracle_nbglm_out *out) {
    oracle_nbglm_opts o;
    if (opts_in) o = *opts_in; else oracle_nbglm_default_opts(&o);
    if (S < 2 || S > MAXS || n < 1) return -1;
    int cellsize[2] = {0, 0};
    int32_t g[MAXS];
    for (int j = 0; j < S; j++) { g[j] = group ? (group[j] != 0) : 0; cellsize[g[j]]++; }
    const int p = cellsize[1] > 0 ? 2 : 1;
    if (p == 2 && cellsize[0] == 0) return -2;
    if (S <= p) return -3;
    const int m = S;
    const double maxDisp = S > 10 ? (double)S : 10.0;
    int status = 0;
#ifdef _OPENMP
    int nthreads = o.nthreads > 0 ? o.nthreads : 1;
#endif

    double *baseMean = (double *)malloc(sizeof(double) * (size_t)n);
    double *baseVar = (double *)malloc(sizeof(double) * (size_t)n);
    double *alphaInit = (double *)malloc(sizeof(double) * (size_t)n);
    double *dispGene = (double *)malloc(sizeof(double) * (size_t)n);
    double *dispFit = (double *)malloc(sizeof(double) * (size_t)n);
    double *dispFinal = (double *)malloc(sizeof(double) * (size_t)n);
    char *allZero = (char *)malloc((size_t)n);

    /* A2.1 getBaseMeansAndVariances + column means of nf over non-all-zero rows (A2.3 xi) */
    double colsum[MAXS];
    int64_t nnz = 0;
    memset(colsum, 0, sizeof colsum);
    for (int64_t i = 0; i < n; i++) {
        double q[MAXS], s = 0;
        int64_t tot = 0;
        for (int j = 0; j < S; j++) {
            int32_t c = counts[(int64_t)j * n + i];
            tot += c;
            q[j] = (double)c / nf[(int64_t)j * n + i];
            s += q[j];
        }
        double bm = s / S, v = 0;
        for (int j = 0; j < S; j++) v += (q[j] - bm) * (q[j] - bm);
        baseMean[i] = bm;
        baseVar[i] = v / (S - 1);
        allZero[i] = (tot == 0);
        if (!allZero[i]) {
            nnz++;
            for (int j = 0; j < S; j++) colsum[j] += nf[(int64_t)j * n + i];
        } else
            status |= ORACLE_ST_ALLZERO_ROWS;
    }
    double xim = 0;
    for (int j = 0; j < S; j++) xim += 1.0 / (colsum[j] / (double)nnz);
    xim /= S;
    if (!isnan(o.xim)) xim = o.xim; /* a slice of a larger fit: the whole fit's value */

    /* A2.2-A2.7 gene-wise estimates */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads)
#endif
    for (int64_t i = 0; i < n; i++) {
        SET(baseMean, i, baseMean[i]);
        SET(baseVar, i, baseVar[i]);
        SET(allZero, i, allZero[i]);
        if (allZero[i]) {
            alphaInit[i] = dispGene[i] = NAN;
            SET(dispInit, i, NAN); SET(dispGeneEst, i, NAN); SET(dispGeneIter, i, 0);
            continue;
        }
        double y[MAXS], q[MAXS], mu[MAXS], gm[2] = {0, 0};
        for (int j = 0; j < S; j++) {
            y[j] = (double)counts[(int64_t)j * n + i];
            q[j] = y[j] / nf[(int64_t)j * n + i];
            gm[g[j]] += q[j];
        }
        gm[0] /= cellsize[0];
        if (p == 2) gm[1] /= cellsize[1];
        /* roughDispEstimate: mu = pmax(1, hat-matrix fit of normalised counts) */
        double est = 0;
        for (int j = 0; j < S; j++) {
            double mj = fmax(1.0, gm[g[j]]);
            est += ((q[j] - mj) * (q[j] - mj) - mj) / (mj * mj);
        }
        est = fmax(est / (m - p), 0.0);
        double moments = (baseVar[i] - xim * baseMean[i]) / (baseMean[i] * baseMean[i]);
        double a0 = fmin(est, moments);
        a0 = fmin(fmax(o.minDisp, a0), maxDisp);
        alphaInit[i] = a0;
        /* linearModelMuNormalized, floored at minmu */
        for (int j = 0; j < S; j++) mu[j] = fmax(gm[g[j]] * nf[(int64_t)j * n + i], o.minmu);
        fitdisp_res r = fit_disp_row(y, mu, g, S, p, log(a0), log(a0), 1.0, 0, log(o.minDisp / 10), o.kappa0,
                                     o.dispTol, o.maxit);
        double d = fmin(exp(r.log_alpha), maxDisp);
        if (r.last_lp < r.initial_lp + fabs(r.initial_lp) / 1e6) d = a0; /* noIncrease */
        int conv = (r.iter < o.maxit) && !(r.iter == 1);
        if (!conv && d > o.minDisp * 10) d = fit_disp_grid_row(y, mu, g, S, p, 0.0, 1.0, 0);
        d = fmin(fmax(d, o.minDisp), maxDisp);
        dispGene[i] = d;
        SET(dispInit, i, a0); SET(dispGeneEst, i, d); SET(dispGeneIter, i, r.iter);
    }

    /* A3 trend on rows with dispGeneEst > 100*minDisp */
    int64_t nfit = 0;
    double *fm = (double *)malloc(sizeof(double) * (size_t)n), *fd = (double *)malloc(sizeof(double) * (size_t)n);
    for (int64_t i = 0; i < n; i++)
        if (!allZero[i] && dispGene[i] > 100 * o.minDisp) { fm[nfit] = baseMean[i]; fd[nfit] = dispGene[i]; nfit++; }
    double coefs[2] = {NAN, NAN};
    int32_t outer = 0;
    int trc = 0;
    if (o.dispFitIn) { /* the fitted values themselves are given (below) */
        if (o.fitType == 2) status |= ORACLE_ST_TREND_LOCAL;
    } else if (!isnan(o.trendCoef[0]) && !isnan(o.trendCoef[1])) { /* dispersionFunction<- : a caller-supplied trend */
        coefs[0] = o.trendCoef[0];
        coefs[1] = o.trendCoef[1];
    } else if (o.fitType == 1) {
        /* estimateDispersionsFit(fitType = "mean"): useForMean <- dispGeneEst > 10*minDisp;
         * meanDisp <- mean(dispGeneEst[useForMean], na.rm = TRUE, trim = 0.001); dispFit <- meanDisp for every row.
         * R's mean(x, trim): lo <- floor(n*trim) + 1; hi <- n + 1 - lo; mean(sort(x)[lo:hi]) */
        int64_t m = 0;
        for (int64_t i = 0; i < n; i++)
            if (!allZero[i] && dispGene[i] > 10 * o.minDisp) fd[m++] = dispGene[i];
        if (m == 0) trc = 4;
        else {
            qsort(fd, (size_t)m, sizeof(double), cmp_double);
            const int64_t lo = (int64_t)floor((double)m * 0.001), hi = m - lo;
            long double acc = 0;
            for (int64_t k = lo; k < hi; k++) acc += fd[k];
            long double mean = acc / (long double)(hi - lo), t = 0;
            for (int64_t k = lo; k < hi; k++) t += fd[k] - mean;
            mean += t / (long double)(hi - lo);
            coefs[0] = (double)mean;
            coefs[1] = 0.0;
        }
    } else if (o.fitType != 2)
        trc = nfit > 0 ? oracle_parametric_dispersion_fit(fm, fd, nfit, coefs, &outer) : 4;
    /* estimateDispersionsFit: fitType = "local" on request, and as the substitute for a failed parametric fit
     * ("a local regression fit was automatically substituted") — locfit_oracle.c */
    oracle_locfit lfit;
    int use_local = 0;
    if (!o.dispFitIn && (o.fitType == 2 || (o.fitType == 0 && trc && isnan(o.trendCoef[0]) && !o.noLocalSubstitute))) {
        trc = oracle_local_dispersion_fit(fm, fd, nfit, &lfit);
        if (!trc) {
            use_local = 1;
            status |= ORACLE_ST_TREND_LOCAL;
            coefs[0] = coefs[1] = NAN;
        }
    }
    if (trc) status |= ORACLE_ST_TREND_FAILED;
    out->trendCoef[0] = coefs[0];
    out->trendCoef[1] = coefs[1];
    out->trendOuterIter = outer;

    /* varLogDispEsts = mad(log dispGeneEst - log dispFit)[dispGeneEst >= 100*minDisp]^2 */
    int64_t nres = 0;
    for (int64_t i = 0; i < n; i++) {
        dispFit[i] = allZero[i] ? NAN : (o.dispFitIn ? o.dispFitIn[i] : use_local ? exp(oracle_locfit_eval(&lfit, log(baseMean[i]))) : coefs[0] + coefs[1] / baseMean[i]);
        SET(dispFit, i, dispFit[i]);
        if (!allZero[i] && dispGene[i] >= 100 * o.minDisp) fm[nres++] = log(dispGene[i]) - log(dispFit[i]);
    }
    memcpy(fd, fm, sizeof(double) * (size_t)nres);
    double obs_hist[40];
    memset(obs_hist, 0, sizeof obs_hist);
    for (int64_t k = 0; k < nres; k++) {
        const int b = oracle_prior_mc_bin(fm[k]);
        if (b >= 0) obs_hist[b] += 1;
    }
    double med = oracle_median(fd, nres);
    for (int64_t k = 0; k < nres; k++) fd[k] = fabs(fm[k] - med);
    double madv = 1.4826 * oracle_median(fd, nres);
    double varLogDispEsts = isnan(o.varLogDispEsts) ? madv * madv : o.varLogDispEsts;
    free(fm);
    free(fd);
    out->varLogDispEsts = varLogDispEsts;

    /* A4 estimateDispersionsPriorVar */
    double dispPriorVar = o.dispPriorVar;
    if (isnan(dispPriorVar)) {
        if (m - p <= 3 && m > p) { /* DESeq2 matches the prior variance by simulation here (prior_mc_oracle.c) */
            status |= ORACLE_ST_PRIORVAR_MC;
            dispPriorVar = oracle_prior_var_mc(obs_hist, m - p);
        }
        if (isnan(dispPriorVar)) dispPriorVar = fmax(varLogDispEsts - oracle_trigamma((m - p) / 2.0), 0.25);
    }
    out->dispPriorVar = dispPriorVar;

    /* A4 MAP + A5 Wald */
    const double lambda = 1e-6 / (M_LN2 * M_LN2);
    double sumdev = 0;
    int any_nonconv = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 256) num_threads(nthreads) reduction(+ : sumdev) reduction(| : any_nonconv)
#endif
    for (int64_t i = 0; i < n; i++) {
        if (allZero[i]) {
            dispFinal[i] = NAN;
            SET(dispMAP, i, NAN); SET(dispersion, i, NAN); SET(dispIter, i, 0); SET(dispOutlier, i, 0);
            SET(beta0, i, NAN); SET(beta1, i, NAN); SET(se0, i, NAN); SET(se1, i, NAN);
            SET(stat, i, NAN); SET(pvalue, i, NAN); SET(deviance, i, NAN);
            SET(betaConv, i, 0); SET(betaIter, i, 0); SET(maxCooks, i, NAN); SET(cooksArgmax, i, -1);
            if (out->mu) for (int j = 0; j < S; j++) out->mu[(int64_t)j * n + i] = NAN;
            sumdev += NAN;
            continue;
        }
        double y[MAXS], q[MAXS], mu[MAXS], nfr[MAXS], gm[2] = {0, 0};
        for (int j = 0; j < S; j++) {
            y[j] = (double)counts[(int64_t)j * n + i];
            nfr[j] = nf[(int64_t)j * n + i];
            q[j] = y[j] / nfr[j];
            gm[g[j]] += q[j];
        }
        gm[0] /= cellsize[0];
        if (p == 2) gm[1] /= cellsize[1];
        for (int j = 0; j < S; j++) mu[j] = fmax(gm[g[j]] * nfr[j], o.minmu);
        double dg = dispGene[i], df = dispFit[i];
        double dinit = dg > 0.1 * df ? dg : df;
        fitdisp_res r = fit_disp_row(y, mu, g, S, p, log(dinit), log(df), dispPriorVar, 1, log(o.minDisp / 10),
                                     o.kappa0, o.dispTol, o.maxit);
        double dmap = exp(r.log_alpha);
        if (!(r.iter < o.maxit)) dmap = fit_disp_grid_row(y, mu, g, S, p, log(df), dispPriorVar, 1);
        dmap = fmin(fmax(dmap, o.minDisp), maxDisp);
        int outlier = log(dg) > log(df) + o.outlierSD * sqrt(varLogDispEsts);
        double alpha = outlier ? dg : dmap;
        dispFinal[i] = alpha;
        SET(dispMAP, i, dmap); SET(dispersion, i, alpha); SET(dispIter, i, r.iter); SET(dispOutlier, i, outlier);

        if (p == 1) {
            /* fitNbinomGLMs intercept-only shortcut */
            double bm = 0;
            for (int j = 0; j < S; j++) bm += q[j];
            bm /= S;
            double beta = log2(bm), ll = 0, xtwx = 0;
            for (int j = 0; j < S; j++) {
                double muj = nfr[j] * exp2(beta);
                ll += oracle_dnbinom_mu_log(y[j], 1.0 / alpha, muj);
                xtwx += 1.0 / (1.0 / muj + alpha);
                if (out->mu) out->mu[(int64_t)j * n + i] = muj;
            }
            double se = LOG2E * sqrt(1.0 / xtwx), st = beta / se;
            SET(beta0, i, beta); SET(beta1, i, NAN); SET(se0, i, se); SET(se1, i, NAN);
            SET(stat, i, st); SET(pvalue, i, oracle_pnorm_two_sided(st));
            SET(deviance, i, -2.0 * ll); SET(betaConv, i, 1); SET(betaIter, i, 1); SET(maxCooks, i, NAN); SET(cooksArgmax, i, -1);
            sumdev += -2.0 * ll;
            continue;
        }
        /* beta init: LS of log(q + 0.1) on X (natural log) */
        double lA = 0, lB = 0;
        for (int j = 0; j < S; j++) { double l = log(q[j] + 0.1); if (g[j]) lB += l; else lA += l; }
        lA /= cellsize[0];
        lB /= cellsize[1];
        fitbeta_res fb = fit_beta_row(y, nfr, g, S, alpha, lA, lB - lA, lambda, o.betaTol, o.betaMaxit, o.minmu);
        int bconv = fb.iter < o.betaMaxit, used_optim = 0;
        if (!bconv || isnan(fb.b0) || isnan(fb.b1)) {
            used_optim = 1;
            /* fitNbinomGLMsOptim: start from the LS start values, store the optimum whatever happens */
            double ob0 = lA, ob1 = lB - lA;
            bconv = beta_optim_row(y, nfr, g, S, alpha, lambda, &ob0, &ob1);
            fitbeta_res fo = fit_beta_row(y, nfr, g, S, alpha, ob0, ob1, lambda, o.betaTol, 0, o.minmu); /* covariance at the optimum */
            fo.iter = fb.iter;
            fb = fo;
        }
        if (!bconv || !(fb.v0 > 0) || !(fb.v1 > 0)) any_nonconv |= 1;
        double ll = 0, muf[MAXS];
        for (int j = 0; j < S; j++) {
            muf[j] = nfr[j] * exp(fb.b0 + (g[j] ? fb.b1 : 0.0)); /* no minmu floor here ... */
            /* ... except on the optim path, where DESeq2 evaluates logLike after flooring mu_row */
            ll += oracle_dnbinom_mu_log(y[j], 1.0 / alpha, used_optim ? fmax(muf[j], o.minmu) : muf[j]);
            if (out->mu) out->mu[(int64_t)j * n + i] = muf[j];
        }
        double B0 = LOG2E * fb.b0, B1 = LOG2E * fb.b1;
        double s0 = LOG2E * sqrt(fmax(fb.v0, 0)), s1 = LOG2E * sqrt(fmax(fb.v1, 0));
        double st = B1 / s1;
        SET(beta0, i, B0); SET(beta1, i, B1); SET(se0, i, s0); SET(se1, i, s1);
        SET(stat, i, st); SET(pvalue, i, oracle_pnorm_two_sided(st));
        SET(deviance, i, -2.0 * ll); SET(betaConv, i, bconv); SET(betaIter, i, fb.iter);
        sumdev += -2.0 * ll;
        if (out->maxCooks) {
            double mc = NAN;
            if (cellsize[0] >= 3 || cellsize[1] >= 3) {
                double arob = robust_mom_disp(q, g, S, cellsize);
                mc = -INFINITY;
                double call = -INFINITY;
                int amax = -1;
                for (int j = 0; j < S; j++) {
                    double V = muf[j] + arob * muf[j] * muf[j];
                    double pr = (y[j] - muf[j]) * (y[j] - muf[j]) / V;
                    double h = fb.hat[j];
                    double ck = pr / p * h / ((1 - h) * (1 - h));
                    if (ck > call) { call = ck; amax = j; }
                    if (cellsize[g[j]] >= 3 && ck > mc) mc = ck;
                }
                SET(cooksArgmax, i, amax);
            } else {
                SET(cooksArgmax, i, -1);
            }
            out->maxCooks[i] = mc;
        }
    }
    if (any_nonconv) status |= ORACLE_ST_BETA_NONCONV;
    out->sumDeviance = sumdev;
    out->status = status;
    free(baseMean); free(baseVar); free(alphaInit); free(dispGene); free(dispFit); free(dispFinal); free(allZero);
    return 0;
}

/* ---------------------------------------------------------------------------------- */
/* Arbiter for the gene-wise estimates (tests only).  DESeq2 — and the restatement above — evaluates the Cox-Reid
 * objective with lgamma(y + 1/alpha) - lgamma(1/alpha), which cancels 6-9 digits once 1/alpha reaches 1e6..1e8 (and,
 * for counts ~1e9, loses as much in the sums), so a few of the line search's accept / stop decisions per million rows
 * are taken inside rounding noise — in ANY double-precision implementation, the GPU's included.  This twin runs the
 * SAME control flow (A2.6-A2.7) in IEEE binary128 (libquadmath: 34 digits; lgammaq, and a digamma by recurrence +
 * asymptotic series): what the algorithm gives when its arithmetic is effectively exact — the referee when the GPU
 * and the double-precision oracle disagree on a row. */
#include <quadmath.h>
typedef __float128 qd;
static qd digammaq(qd x) { /* x > 0 */
    qd s = 0;
    while (x < 40) { s -= 1 / x; x += 1; }
    const qd i2 = 1 / (x * x);
    /* B_2k / (2k): 1/12, -1/120, 1/252, -1/240, 1/132, -691/32760, 1/12, -3617/8160, 43867/14364, -174611/6600 */
    const qd series = i2 * (1.0Q / 12 - i2 * (1.0Q / 120 - i2 * (1.0Q / 252 - i2 * (1.0Q / 240 - i2 * (1.0Q / 132 - i2 * (691.0Q / 32760 -
                      i2 * (1.0Q / 12 - i2 * (3617.0Q / 8160 - i2 * (43867.0Q / 14364 - i2 * (174611.0Q / 6600))))))))));
    return s + logq(x) - 0.5Q / x - series;
}
static qd lp_exact(qd a, const double *y, const double *mu, const int32_t *g, int S, int p, int use_prior, qd pmean, qd psig) {
    const qd alpha = expq(a), r = 1 / alpha;
    qd wA = 0, wB = 0, ll = 0;
    const qd lgr = lgammaq(r);
    for (int j = 0; j < S; j++) {
        const qd w = 1 / (1 / (qd)mu[j] + alpha);
        if (p == 2 && g[j]) wB += w; else wA += w;
        ll += lgammaq((qd)y[j] + r) - lgr - (qd)y[j] * logq((qd)mu[j] + r) - r * log1pq((qd)mu[j] * alpha);
    }
    qd pr = 0;
    if (use_prior) pr = -0.5Q * (a - pmean) * (a - pmean) / psig;
    return ll + pr - 0.5Q * logq(p == 2 ? wA * wB : wA);
}
static qd dlp_exact(qd a, const double *y, const double *mu, const int32_t *g, int S, int p, int use_prior, qd pmean, qd psig) {
    const qd alpha = expq(a), r = 1 / alpha;
    qd wA = 0, wB = 0, dA = 0, dB = 0, s = 0;
    const qd dgr = digammaq(r);
    for (int j = 0; j < S; j++) {
        const qd t = 1 / (qd)mu[j] + alpha, w = 1 / t, dw = -1 / (t * t);
        if (p == 2 && g[j]) { wB += w; dB += dw; } else { wA += w; dA += dw; }
        const qd ma = (qd)mu[j] * alpha;
        s += dgr + log1pq(ma) - ma / (1 + ma) - digammaq((qd)y[j] + r) + (qd)y[j] / ((qd)mu[j] + r);
    }
    const qd cr = -0.5Q * (p == 2 ? dA / wA + dB / wB : dA / wA);
    return (r * r * s + cr) * alpha + (use_prior ? -(a - pmean) / psig : 0);
}
/* stage 0: gene-wise estimate (start = dispInit, no prior); stage 1: MAP estimate (start / prior mean from dispGene,
 * dispFit as A4 prescribes, prior variance dispPriorVar) */
int oracle_arbitrate_disp(const int32_t *counts, const double *nf, int64_t n, int32_t S, const int32_t *group,
                          const int64_t *rows, int64_t nrows, int32_t stage, const double *dispInit, const double *dispGene,
                          const double *dispFit, double dispPriorVar, const oracle_nbglm_opts *opts_in, double *out) {
    oracle_nbglm_opts o;
    if (opts_in) o = *opts_in; else oracle_nbglm_default_opts(&o);
    if (S < 2 || S > MAXS) return -1;
    int cellsize[2] = {0, 0};
    int32_t g[MAXS];
    for (int j = 0; j < S; j++) { g[j] = group ? (group[j] != 0) : 0; cellsize[g[j]]++; }
    const int p = cellsize[1] > 0 ? 2 : 1;
    const double maxDisp = S > 10 ? (double)S : 10.0;
    int rc = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int64_t t = 0; t < nrows; t++) {
        const int64_t i = rows[t];
        if (i < 0 || i >= n) { rc = -2; continue; }
        double y[MAXS], mu[MAXS], gm[2] = {0, 0};
        for (int j = 0; j < S; j++) {
            y[j] = (double)counts[(int64_t)j * n + i];
            gm[g[j]] += y[j] / nf[(int64_t)j * n + i];
        }
        gm[0] /= cellsize[0];
        if (p == 2) gm[1] /= cellsize[1];
        for (int j = 0; j < S; j++) mu[j] = fmax(gm[g[j]] * nf[(int64_t)j * n + i], o.minmu);
        const int up = stage == 1;
        double a0;
        qd pmean = 0, psig = 1;
        if (!up) a0 = dispInit[i];
        else {
            a0 = dispGene[i] > 0.1 * dispFit[i] ? dispGene[i] : dispFit[i];
            pmean = logq((qd)dispFit[i]);
            psig = dispPriorVar;
        }
        /* fitDisp */
        const qd eps = 1.0e-4Q, min_log_alpha = logq((qd)o.minDisp / 10);
        qd a = logq((qd)a0), lp = lp_exact(a, y, mu, g, S, p, up, pmean, psig), dlp = dlp_exact(a, y, mu, g, S, p, up, pmean, psig);
        qd kappa = o.kappa0;
        const qd initial_lp = lp;
        int iter = 0, iter_accept = 0;
        for (int it = 0; it < o.maxit; it++) {
            iter++;
            const qd a_propose = a + kappa * dlp;
            if (a_propose < -30) kappa = (-30 - a) / dlp;
            if (a_propose > 10) kappa = (10 - a) / dlp;
            const qd theta_kappa = -lp_exact(a + kappa * dlp, y, mu, g, S, p, up, pmean, psig);
            const qd theta_hat_kappa = -lp - kappa * eps * dlp * dlp;
            if (theta_kappa <= theta_hat_kappa) {
                iter_accept++;
                a = a + kappa * dlp;
                const qd lpnew = lp_exact(a, y, mu, g, S, p, up, pmean, psig), change = lpnew - lp;
                if (change < (qd)o.dispTol) { lp = lpnew; break; }
                if (a < min_log_alpha) break;
                lp = lpnew;
                dlp = dlp_exact(a, y, mu, g, S, p, up, pmean, psig);
                kappa = fminq(kappa * 1.1Q, o.kappa0);
                if (iter_accept % 5 == 0) kappa /= 2;
            } else
                kappa /= 2;
        }
        double d = (double)expq(a);
        int grid;
        if (!up) {
            d = fmin(d, maxDisp);
            if (lp < initial_lp + fabsq(initial_lp) / 1e6Q) d = a0;
            const int conv = (iter < o.maxit) && !(iter == 1);
            grid = !conv && d > o.minDisp * 10;
        } else
            grid = !(iter < o.maxit);
        if (grid) { /* fitDispGrid */
            const int G = 20;
            const qd lo = logq(1e-8Q), hi = logq((qd)maxDisp);
            qd best = -INFINITY, a_hat = lo;
            const qd delta = (hi - lo) / (G - 1);
            for (int q = 0; q < G; q++) {
                const qd aa = lo + (hi - lo) * q / (G - 1), v = lp_exact(aa, y, mu, g, S, p, up, pmean, psig);
                if (v > best) { best = v; a_hat = aa; }
            }
            qd fbest = -INFINITY, fa = a_hat;
            for (int q = 0; q < G; q++) {
                const qd aa = (a_hat - delta) + 2 * delta * q / (G - 1), v = lp_exact(aa, y, mu, g, S, p, up, pmean, psig);
                if (v > fbest) { fbest = v; fa = aa; }
            }
            d = (double)expq(fa);
        }
        out[t] = fmin(fmax(d, o.minDisp), maxDisp);
    }
    return rc;
}
