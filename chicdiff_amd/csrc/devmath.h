// devmath.h — fp64 special functions for the gfx950 fit kernels.
//
// HIP's device library has lgamma but no digamma/trigamma, and the dispersion objective
// (DESeq2 fitDisp) needs lgamma(x) and digamma(x) at the SAME x in
// every evaluation.  lgamma_digamma() computes both from one log and one reciprocal:
// upward recurrence to z >= 10 carried as a product P = prod(x+i) and its derivative Q = dP/dx
// (so the shift costs two FMAs per step, no division), then the Stirling series.
#pragma once
#include <hip/hip_runtime.h>

#include "exp_table.h"
#include "fit_state.h"
#include "log_table.h"

namespace cd {

// d = a*b + c as the three-address v_fma_f64.  For a Horner step whose addend is a loop-invariant constant held
// in a VGPR the compiler emits v_mov_b64 (copy the constant) + v_fmac_f64 — two issue slots; in the line-search
// kernels those copies were 17 % of the per-sample instructions.  (Scalar registers, where such constants
// would cost nothing, are exhausted there.)  Plain asm, no side effects: still CSE'd and scheduled.
__device__ __forceinline__ double fma3(double a, double b, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// 1/x for normal, well-scaled x (no denormal / overflow handling): v_rcp_f64 + two Newton steps,
// ~5 instructions against ~12 for the IEEE division expansion.  < 1 ulp.
__device__ __forceinline__ double rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}

// 1/x with IEEE behaviour outside rcp()'s range (zero, denormal, huge, infinite, NaN, negative: the division)
__device__ __forceinline__ double rcp_or_div(double x) { return (x > 1e-300 && x < 1e300) ? rcp(x) : 1.0 / x; }

// log(x) for positive, finite, normal x.  fdlibm's reduction (x = 2^k (1+f), s = f/(2+f)) with
// the division done by rcp(): ~35 instructions against ~55 for the device-library log.  <= 1 ulp.
__device__ __forceinline__ double flog(double x) {
    int k = __builtin_amdgcn_frexp_exp(x);          // x = m * 2^k, m in [0.5, 1)
    double m = __builtin_amdgcn_frexp_mant(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;                              // m in [sqrt(1/2), sqrt(2))
    k = lo ? k - 1 : k;
    const double f = m - 1.0;
    const double d = 2.0 + f;
    const double rd = rcp(d);
    double s = f * rd;
    s = fma(fma(-s, d, f), rd, s);                   // correctly rounded-ish f/d
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    // k*ln2_hi - ((hfsq - (s*(hfsq+R) + k*ln2_lo)) - f)
    return fma(dk, 6.93147180369123816490e-01, -((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
}

// ---- table-driven log for the hot loops ------------------------------------------------------
// x = 2^k z, z in [0.6875, 1.375); the top 6 mantissa bits of (bits(x) - OFF) pick c_i ~ z with
// (1/c_i, log c_i) tabulated (log_table.h); r = z/c_i - 1, |r| <= 1/64, and
// log x = k ln2 + log c_i + log1p(r) with a degree-9 Taylor polynomial (truncation < 6e-18 relative).
// c = 1 on both sides of z = 1 keeps full relative accuracy for results near 0.  ~24 VALU
// instructions + one ds_read_b128 against ~38 for flog().  The 1 KB table lives in LDS (`tab`).
struct LogEntry {
    double invc, logc;
};
static __device__ const LogEntry kLogTable[64] = {CD_LOG_TABLE_INIT};

__device__ __forceinline__ void log_table_to_lds(LogEntry *s_tab) {
    if (threadIdx.x < 64) s_tab[threadIdx.x] = kLogTable[threadIdx.x];
    __syncthreads();
}

__device__ __forceinline__ double tlog(double x, const LogEntry *tab) {
    const unsigned long long ix = (unsigned long long)__double_as_longlong(x);
    const unsigned long long tmp = ix - 0x3FE6000000000000ull;
    const int hi = (int)(unsigned int)(tmp >> 32);  // the exponent lives in the high word: keep k a 32-bit quantity (as a 64-bit
    const int i = (hi >> 14) & 63;                  // shift the compiler converts it to double as an int64: two cvt, ldexp, add)
    const int k = hi >> 20;
    const double z = __longlong_as_double((long long)(ix - (tmp & 0xFFF0000000000000ull)));
    const LogEntry e = tab[i];
    const double r = fma(z, e.invc, -1.0);
    const double r2 = r * r;
    // log1p(r) = r - r^2/2 + ... + r^9/9, Estrin-style to shorten the dependency chain
    const double p01 = fma(r, -0.5, 1.0);
    const double p23 = fma3(r, -0.25, 1.0 / 3.0);
    const double p45 = fma3(r, -1.0 / 6.0, 0.2);
    const double p67 = fma3(r, -0.125, 1.0 / 7.0);
    double q = fma(r2, 1.0 / 9.0, p67);
    q = fma(r2, q, p45);
    q = fma(r2, q, p23);
    q = fma(r2, q, p01);
    const double dk = (double)k;
    return fma(dk, 6.93147180369123816490e-01, e.logc) + fma(r, q, dk * 1.90821492927058770002e-10);
}
__device__ __forceinline__ double tlog1p_from(double u, double t, double rt, const LogEntry *tab) {
    return fma(u - (t - 1.0), rt, tlog(t, tab));
}

// ---- table-driven exp for the hot loops ------------------------------------------------------
// x = (64 m + j) ln2/64 + r, |r| <= ln2/128: exp x = 2^m 2^(j/64) (1 + expm1 r), 2^(j/64) tabulated as (hi, lo)
// (exp_table.h), expm1 r by its degree-5 Taylor polynomial (truncation r^6/720 < 3.5e-17).  ~16 VALU instructions + one
// ds_read_b128 against ~38 for the device-library exp; <= 1 ulp.  Finite x only (NaN stays NaN; +-Inf gives NaN: the callers'
// arguments are bounded — log alpha in [-30, 10], IRLS coefficients within +-30).  The 1 KB table lives in LDS (`tab`).
struct ExpEntry {
    double hi, lo;
};
static __device__ const ExpEntry kExpTable[64] = {CD_EXP_TABLE_INIT};

// (call before a __syncthreads() that precedes the first use: log_table_to_lds() ends with one)
__device__ __forceinline__ void exp_table_to_lds(ExpEntry *s_tab) {
    if (threadIdx.x < 64) s_tab[threadIdx.x] = kExpTable[threadIdx.x];
}

__device__ __forceinline__ double texp(double x, const ExpEntry *tab) {
    const double kd = rint(x * 92.33248261689366);              // 64 / ln2
    const int k = (int)kd;
    double r = fma(-kd, 0x1.62e42fee00000p-7, x);               // ln2/64, high part: 32 significant bits, k * hi is exact
    r = fma(-kd, 0x1.a39ef35793c76p-39, r);
    const ExpEntry e = tab[k & 63];
    const double r2 = r * r;
    const double a = fma(r, 1.0 / 6.0, 0.5);
    const double b = fma3(r, 1.0 / 120.0, 1.0 / 24.0);
    const double q = fma(r2, b, a);
    const double p = fma(r2, q, r);                             // expm1(r)
    return ldexp(e.hi + fma(e.hi, p, e.lo), k >> 6);
}

// log1p(u) for u >= 0 where t = 1 + u and rt = 1/t are already at hand: log(t) plus the
// first-order correction for the bits of u lost in forming t.
__device__ __forceinline__ double flog1p_from(double u, double t, double rt) {
    return fma(u - (t - 1.0), rt, flog(t));
}

// Stirling tails for z >= 10 (truncation < 4e-17):
//   lgamma(z)  = (z - 1/2) log z - z + log sqrt(2 pi) + lg_tail(1/z)
//   digamma(z) = log z - 1/(2z) - dg_tail(1/z^2)
__device__ __forceinline__ void stirling(double z, double lz, double zi, double &lg, double &dg) {
    const double z2 = zi * zi;
    // the two Horner chains are written interleaved: each step depends on the one two lines up, so a wave always has an
    // independent fp64 instruction to issue (as two separate chains the compiler emitted them back to back, with wait states)
    double s = fma3(z2, 1.0 / 156.0, -691.0 / 360360.0);
    double t = fma3(z2, 1.0 / 12.0, -691.0 / 32760.0);
    s = fma3(z2, s, 1.0 / 1188.0);
    t = fma3(z2, t, 1.0 / 132.0);
    s = fma3(z2, s, -1.0 / 1680.0);
    t = fma3(z2, t, -1.0 / 240.0);
    s = fma3(z2, s, 1.0 / 1260.0);
    t = fma3(z2, t, 1.0 / 252.0);
    s = fma3(z2, s, -1.0 / 360.0);
    t = fma3(z2, t, -1.0 / 120.0);
    s = fma3(z2, s, 1.0 / 12.0);
    t = fma3(z2, t, 1.0 / 12.0);
    lg = fma(z - 0.5, lz, -z) + 0.91893853320467274178 + s * zi;
    dg = fma(-0.5, zi, lz) - t * z2;
}

// ---- log Gamma(y + r) - log Gamma(r) for an integer count y >= 0 and r > 0 ---------------------
// Stable for huge r (alpha -> 1e-8 means r = 1e8, where lgamma(y+r) - lgamma(r) cancels ~9 digits):
//   y <  nr : log prod_{i<y}(r+i)
//   y >= nr : log prod_{i<nr}(r+i) + (z - 1/2) log1p(d/z0) + d (log z0 - 1) + tail(z) - tail(z0),
// with nr = number of unit steps lifting r to z0 = r + nr >= 10, d = y - nr, z = z0 + d, and
// tail() the Stirling correction series.  Everything that depends only on r sits in LgrCtx.
__device__ __forceinline__ double lg_tail(double zi) {
    const double z2 = zi * zi;
    double s = fma(z2, 1.0 / 156.0, -691.0 / 360360.0);
    s = fma(z2, s, 1.0 / 1188.0);
    s = fma(z2, s, -1.0 / 1680.0);
    s = fma(z2, s, 1.0 / 1260.0);
    s = fma(z2, s, -1.0 / 360.0);
    s = fma(z2, s, 1.0 / 12.0);
    return s * zi;
}
struct LgrCtx {
    double r, z0, iz0, lz0m1, tail0, lP;
    int nr;
};
__device__ __forceinline__ LgrCtx lgr_make(double r) {
    LgrCtx c;
    c.r = r;
    c.nr = r < 10.0 ? (int)ceil(10.0 - r) : 0;
    double P = 1.0, zz = r;
    for (int i = 0; i < c.nr; i++) {
        P *= zz;
        zz += 1.0;
    }
    c.z0 = r + (double)c.nr;
    c.iz0 = rcp(c.z0);
    c.lz0m1 = flog(c.z0) - 1.0;
    c.tail0 = lg_tail(c.iz0);
    c.lP = c.nr ? flog(P) : 0.0;
    return c;
}
// the same two with the table-driven logarithms of the fit kernels (LDS table `lt`): ~30 instructions less per sample in wald_prep
// and wald_intercept, whose every lane runs BOTH branches of lgr_eval (a wave holds small and large counts)
__device__ __forceinline__ LgrCtx lgr_make_t(double r, const LogEntry *lt) {
    LgrCtx c;
    c.r = r;
    c.nr = r < 10.0 ? (int)ceil(10.0 - r) : 0;
    double P = 1.0, zz = r;
    for (int i = 0; i < c.nr; i++) {
        P *= zz;
        zz += 1.0;
    }
    c.z0 = r + (double)c.nr;
    c.iz0 = rcp(c.z0);
    c.lz0m1 = tlog(c.z0, lt) - 1.0;
    c.tail0 = lg_tail(c.iz0);
    c.lP = c.nr ? tlog(P, lt) : 0.0;
    return c;
}
__device__ __forceinline__ double lgr_eval_t(const LgrCtx &c, int yi, const LogEntry *lt) {
    if (yi >= c.nr) {
        const double d = (double)(yi - c.nr);
        const double z = c.z0 + d;
        const double u = d * c.iz0, t = 1.0 + u;
        const double l1p = tlog1p_from(u, t, rcp(t), lt);
        return c.lP + fma(z - 0.5, l1p, d * c.lz0m1) + (lg_tail(rcp(z)) - c.tail0);
    }
    double P = 1.0, zz = c.r;
    for (int i = 0; i < yi; i++) {
        P *= zz;
        zz += 1.0;
    }
    return tlog(P, lt);
}
// the context of r = 1, i.e. log(y!) = lgr_eval(lgr_one(), y)
__device__ __forceinline__ LgrCtx lgr_one() {
    LgrCtx c;
    c.r = 1.0;
    c.nr = 9;
    c.z0 = 10.0;
    c.iz0 = 0.1;
    c.lz0m1 = 2.302585092994045684 - 1.0;
    c.tail0 = lg_tail(0.1);
    c.lP = 12.80182748008146961;        // log(9!)
    return c;
}
__device__ __forceinline__ double lgr_eval(const LgrCtx &c, int yi) {
    if (yi >= c.nr) {
        const double d = (double)(yi - c.nr);
        const double z = c.z0 + d;
        const double u = d * c.iz0, t = 1.0 + u;
        const double l1p = flog1p_from(u, t, rcp(t));
        return c.lP + fma(z - 0.5, l1p, d * c.lz0m1) + (lg_tail(rcp(z)) - c.tail0);
    }
    double P = 1.0, zz = c.r;
    for (int i = 0; i < yi; i++) {
        P *= zz;
        zz += 1.0;
    }
    return flog(P);
}

// lgamma(x) and digamma(x), x > 0 (general purpose; the fit kernels use the difference form in
// disp_kernels.hip instead).
__device__ __forceinline__ void lgamma_digamma(double x, double &lg, double &dg) {
    double z = x, P = 1.0, Q = 0.0;
    while (z < 10.0) {
        Q = fma(Q, z, P);
        P *= z;
        z += 1.0;
    }
    stirling(z, flog(z), rcp(z), lg, dg);
    if (x < 10.0) {
        lg -= flog(P);
        dg -= Q * rcp(P);
    }
}

__device__ __forceinline__ double lgamma_pos(double x) {
    double z = x, P = 1.0;
    while (z < 10.0) {
        P *= z;
        z += 1.0;
    }
    const double zi = rcp(z), z2 = zi * zi, lz = flog(z);
    double s = fma(z2, 1.0 / 156.0, -691.0 / 360360.0);
    s = fma(z2, s, 1.0 / 1188.0);
    s = fma(z2, s, -1.0 / 1680.0);
    s = fma(z2, s, 1.0 / 1260.0);
    s = fma(z2, s, -1.0 / 360.0);
    s = fma(z2, s, 1.0 / 12.0);
    double lg = fma(z - 0.5, lz, -z) + 0.91893853320467274178 + s * zi;
    if (x < 10.0) lg -= flog(P);
    return lg;
}

// ---- Loader's saddle-point binomial pieces (R nmath dnbinom_mu), for the reported deviance ----
// stirlerr(n) = lgamma(n+1) - (n+0.5) log n + n - log sqrt(2 pi)
__device__ __forceinline__ double stirlerr(double n) {
    if (n <= 15.0) {
        if (n == 0.0) return 0.0;
        return lgamma_pos(n + 1.0) - (n + 0.5) * log(n) + n - 0.91893853320467274178;
    }
    const double nn = n * n;
    if (n > 500) return (1.0 / 12 - (1.0 / 360) / nn) / n;
    if (n > 80) return (1.0 / 12 - (1.0 / 360 - (1.0 / 1260) / nn) / nn) / n;
    if (n > 35) return (1.0 / 12 - (1.0 / 360 - (1.0 / 1260 - (1.0 / 1680) / nn) / nn) / nn) / n;
    return (1.0 / 12 - (1.0 / 360 - (1.0 / 1260 - (1.0 / 1680 - (1.0 / 1188) / nn) / nn) / nn) / nn) / n;
}

// bd0(x, np) = x log(x/np) + np - x without cancellation near x ~ np
__device__ __forceinline__ double bd0(double x, double np) {
    if (fabs(x - np) < 0.1 * (x + np)) {
        double v = (x - np) / (x + np);
        double s = (x - np) * v;
        if (fabs(s) < 2.2250738585072014e-308) return s;
        double ej = 2 * x * v;
        v = v * v;
        for (int j = 1; j < 1000; j++) {
            ej *= v;
            const double s1 = s + ej / ((j << 1) + 1);
            if (s1 == s) return s1;
            s = s1;
        }
    }
    return x * log(x / np) + np - x;
}

__device__ __forceinline__ double dbinom_raw_log(double x, double n, double p, double q) {
    if (p == 0) return (x == 0) ? 0.0 : -INFINITY;
    if (q == 0) return (x == n) ? 0.0 : -INFINITY;
    if (x == 0) {
        if (n == 0) return 0.0;
        return (p < 0.1) ? -bd0(n, n * q) - n * p : n * log(q);
    }
    if (x == n) return (q < 0.1) ? -bd0(n, n * p) - n * q : n * log(p);
    if (x < 0 || x > n) return -INFINITY;
    const double lc = stirlerr(n) - stirlerr(x) - stirlerr(n - x) - bd0(x, n * p) - bd0(n - x, n * q);
    const double lf = 1.837877066409345483560659472811 + log(x) + log1p(-x / n);
    return lc - 0.5 * lf;
}

// log dnbinom(x; size, mu), x a non-negative integer value, size finite > 0, mu >= 0
__device__ __forceinline__ double dnbinom_mu_log(double x, double size, double mu) {
    if (x == 0) return size * (size < mu ? log(size / (size + mu)) : log1p(-mu / (size + mu)));
    if (x < 1e-10 * size) {
        const double p = (size < mu ? log(size / (1 + size / mu)) : log(mu / (1 + mu / size)));
        return x * p - mu - lgamma_pos(x + 1) + log1p(x * (x - 1) / (2 * size));
    }
    const double p = size / (size + x);
    return log(p) + dbinom_raw_log(size, x + size, size / (size + mu), mu / (size + mu));
}

// 2 * pnorm(-|z|): Cody (1969) rational approximations, the evaluation R's pnorm uses
__device__ __forceinline__ double pnorm_two_sided(double z) {
    const double y = fabs(z);
    if (!(y == y)) return z;
    if (y <= 0.67448975) {
        double xnum = 0.0, xden = 0.0;
        if (y > 1.1102230246251565e-16) {
            const double xsq = y * y;
            xnum = 0.065682337918207449113 * xsq;
            xden = xsq;
            xnum = (xnum + 2.2352520354606839287) * xsq;   xden = (xden + 47.20258190468824187) * xsq;
            xnum = (xnum + 161.02823106855587881) * xsq;   xden = (xden + 976.09855173777669322) * xsq;
            xnum = (xnum + 1067.6894854603709582) * xsq;   xden = (xden + 10260.932208618978205) * xsq;
        }
        const double temp = y * (xnum + 18154.981253343561249) / (xden + 45507.789335026729956);
        return 2.0 * (0.5 - temp);
    }
    double temp;
    if (y <= 5.656854249492380195206754896838) {
        double xnum = 1.0765576773720192317e-8 * y, xden = y;
        xnum = (xnum + 0.39894151208813466764) * y;  xden = (xden + 22.266688044328115691) * y;
        xnum = (xnum + 8.8831497943883759412) * y;   xden = (xden + 235.38790178262499861) * y;
        xnum = (xnum + 93.506656132177855979) * y;   xden = (xden + 1519.377599407554805) * y;
        xnum = (xnum + 597.27027639480026226) * y;   xden = (xden + 6485.558298266760755) * y;
        xnum = (xnum + 2494.5375852903726711) * y;   xden = (xden + 18615.571640885098091) * y;
        xnum = (xnum + 6848.1904505362823326) * y;   xden = (xden + 34900.952721145977266) * y;
        xnum = (xnum + 11602.651437647350124) * y;   xden = (xden + 38912.003286093271411) * y;
        temp = (xnum + 9842.7148383839780218) / (xden + 19685.429676859990727);
    } else if (y < 37.5193) {
        const double xsq = 1.0 / (y * y);
        double xnum = 0.02307344176494017303 * xsq, xden = xsq;
        xnum = (xnum + 0.21589853405795699) * xsq;      xden = (xden + 1.28426009614491121) * xsq;
        xnum = (xnum + 0.1274011611602473639) * xsq;    xden = (xden + 0.468238212480865118) * xsq;
        xnum = (xnum + 0.022235277870649807) * xsq;     xden = (xden + 0.0659881378689285515) * xsq;
        xnum = (xnum + 0.001421619193227893466) * xsq;  xden = (xden + 0.00378239633202758244) * xsq;
        temp = xsq * (xnum + 2.9112874951168792e-5) / (xden + 7.29751555083966205e-5);
        temp = (0.398942280401432677939946059934 - temp) / y;
    } else {
        return 0.0;
    }
    const double xsq = trunc(y * 16) / 16;
    const double del = (y - xsq) * (y + xsq);
    return 2.0 * (exp(-xsq * xsq * 0.5) * exp(-del * 0.5) * temp);
}


// log() of an R double as the offsets need it: NA / zero / negative / subnormal values take the library path (NaN, -Inf: what R's log()
// gives and the NA-row rule of chicdiff.R:1588 reads), the common positive case the table logarithm
__device__ __forceinline__ double rlog_t(double x, const LogEntry *lt) { return (x > 2.3e-308 && x < 1.7e308) ? tlog(x, lt) : log(x); }
// a4 — one row of the offsets (chicdiff.R:1583-1589 M3, :1635-1638 / :1666-1669 the theta mix): v[0..S) = FullMean in, the
// normalisation factors out.  Shared by offsets16_kernel and by prep16_kernel's fused form (global_kernels.hip / disp_kernels.hip),
// so that the two produce the same bits.  S <= 16; sf: the size factors (nullSizeFactors), wave-uniform loads.
__device__ __forceinline__ void offsets_row16(double (&v)[16], int S, const double *__restrict__ sf, double theta, int mix, const LogEntry *lt) {
    const double iS = 1.0 / S;
    double sl = 0;
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (j < S) sl += rlog_t(v[j], lt);
    const double gmean = exp(sl * iS);
    const double ig = (gmean > 1e-300 && gmean < 1e300) ? rcp(gmean) : 1.0 / gmean;
    bool anyna = false;
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (j < S) {
            v[j] = v[j] * ig;
            anyna |= (v[j] != v[j]);
        }
    double sl2 = 0;
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (j < S) {
            if (anyna) v[j] = sf[j];
            if (mix) {
                v[j] = v[j] * (1 - theta) + sf[j] * theta;
                sl2 += rlog_t(v[j], lt);
            }
        }
    if (mix) {
        const double g2 = exp(sl2 * iS);
        const double i2 = (g2 > 1e-300 && g2 < 1e300) ? rcp(g2) : 1.0 / g2;
#pragma unroll
        for (int j = 0; j < 16; j++)
            if (j < S) v[j] = v[j] * i2;
    }
}

}  // namespace cd
