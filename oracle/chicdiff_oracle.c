/*
 * oracle/chicdiff_oracle.c — TEST INFRASTRUCTURE (parity oracle), not product code.
 *
 * CPU restatement of the Chicdiff-side (non-DESeq2) pieces of the hot path, each
 * following /root/reference/Chicdiff/R/chicdiff.R at the cited lines:
 *   a1 count join      chicdiff.R:843-858   (merge RU x chinput, NA -> 0)
 *   a2 window sums     chicdiff.R:1540-1556 (sum(N), sum(FullMean) by region & sample)
 *   a4 offsets         chicdiff.R:1583-1589, 1614-1615, 1635-1638, 1666-1669
 *   a9 BH adjustment   p.adjust(method="BH") as used by DESeq2 results() / chicdiff.R:2049
 * These lines ARE under /root/reference, so parity here is pinned by the reference's
 * own text plus (for BH) the golden table's weighted_padj / padj columns.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* a2. regionData <- fragData[, .(N=sum(N), ..., FullMean=sum(FullMean)), by=.(baitID, regionID, sample)]
 * (chicdiff.R:1540-1547) after setkey(fragData, otherEndID) (:1526): rows of one region are summed in
 * ascending otherEndID order.  The caller passes fragments already grouped per region in that order.
 * Integer sums are exact; fp64 sums are sequential (NA/NaN propagate, as sum() without na.rm). */
int oracle_window_sums(const int32_t *fragN, const double *fragFullMean, int64_t nfrag, int32_t S,
                       const int64_t *region_ptr, int64_t n, int32_t *N, double *FullMean) {
    if (region_ptr[n] > nfrag) return -1;
    for (int j = 0; j < S; j++) {
        for (int64_t i = 0; i < n; i++) {
            int64_t lo = region_ptr[i], hi = region_ptr[i + 1];
            if (fragN) {
                int64_t s = 0;
                for (int64_t f = lo; f < hi; f++) s += fragN[(int64_t)j * nfrag + f];
                N[(int64_t)j * n + i] = (int32_t)s;
            }
            if (fragFullMean) {
                double s = 0;
                for (int64_t f = lo; f < hi; f++) s += fragFullMean[(int64_t)j * nfrag + f];
                FullMean[(int64_t)j * n + i] = s;
            }
        }
    }
    return 0;
}

/* a4. normFactorsM3 <- FullMean / exp(rowMeans(log(FullMean)))       (chicdiff.R:1585-1586)
 *     rows with any NA <- nullSizeFactors                             (:1588-1589)
 *     mix: sc <- M3*(1-theta) + nsf*theta; sc <- sc/exp(rowMeans(log(sc)))   (:1635-1638, :1666-1669)
 * mix == 0 returns M3 itself (norm = "fullmean", :1600). */
int oracle_offsets(const double *FullMean, const double *sizeFactors, int64_t n, int32_t S, double theta,
                   double *out) {
    int mix = !isnan(theta);
    for (int64_t i = 0; i < n; i++) {
        double sl = 0;
        for (int j = 0; j < S; j++) sl += log(FullMean[(int64_t)j * n + i]);
        double gmean = exp(sl / S);
        int anyna = 0;
        double m3[64];
        for (int j = 0; j < S; j++) {
            m3[j] = FullMean[(int64_t)j * n + i] / gmean;
            if (isnan(m3[j])) anyna = 1;
        }
        if (anyna)
            for (int j = 0; j < S; j++) m3[j] = sizeFactors[j];
        if (mix) {
            double sl2 = 0;
            for (int j = 0; j < S; j++) {
                m3[j] = m3[j] * (1 - theta) + sizeFactors[j] * theta;
                sl2 += log(m3[j]);
            }
            double g2 = exp(sl2 / S);
            for (int j = 0; j < S; j++) m3[j] /= g2;
        }
        for (int j = 0; j < S; j++) out[(int64_t)j * n + i] = m3[j];
    }
    return 0;
}

/* a1. merge(RU, chinput[, .(baitID, otherEndID, N)], all.x=TRUE); N[is.na(N)] <- 0  (chicdiff.R:846-853) */
int oracle_count_join(const int32_t *ru_bait, const int32_t *ru_oe, int64_t nru, const int64_t *keys,
                      const int32_t *vals, int64_t nkeys, int32_t *out) {
    for (int64_t r = 0; r < nru; r++) {
        int64_t key = ((int64_t)ru_bait[r] << 32) | (uint32_t)ru_oe[r];
        int64_t lo = 0, hi = nkeys;
        while (lo < hi) {
            int64_t mid = lo + ((hi - lo) >> 1);
            if (keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        out[r] = (lo < nkeys && keys[lo] == key) ? vals[lo] : 0;
    }
    return 0;
}

/* p.adjust(p, "BH"): n = #non-NA; o = order(p, decreasing=TRUE); pmin(1, cummin(n/i * p[o]))[order(o)] */
typedef struct { double p; int64_t i; } pidx;
static int cmp_pidx_desc(const void *a, const void *b) {
    const pidx *x = (const pidx *)a, *y = (const pidx *)b;
    if (x->p > y->p) return -1;
    if (x->p < y->p) return 1;
    return (x->i > y->i) - (x->i < y->i); /* stable for ties (order() is stable) */
}
int oracle_bh_adjust(const double *p, int64_t n, double *padj) {
    pidx *v = (pidx *)malloc(sizeof(pidx) * (size_t)(n > 0 ? n : 1));
    if (!v) return -1;
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++) {
        padj[i] = NAN;
        if (!isnan(p[i])) { v[m].p = p[i]; v[m].i = i; m++; }
    }
    qsort(v, (size_t)m, sizeof(pidx), cmp_pidx_desc);
    double run = INFINITY;
    for (int64_t k = 0; k < m; k++) {
        double rank = (double)(m - k); /* i = lp:1 */
        double val = (double)m / rank * v[k].p;
        if (val < run) run = val;
        padj[v[k].i] = run < 1.0 ? run : 1.0;
    }
    free(v);
    return 0;
}

/* a3. Chicago .distFun: exp of a cubic in log d inside [obs.min, obs.max], log-linear head and tail. */
static double dist_fun(double d, const double *p) {
    const double ld = log(d);
    double out;
    if (ld > p[9]) out = p[6] + ld * p[7];
    else if (ld < p[8]) out = p[4] + ld * p[5];
    else out = p[0] + p[1] * ld + p[2] * (ld * ld) + p[3] * (ld * ld * ld);
    return exp(out);
}

int oracle_fragment_background(const int32_t *bait, const int32_t *oe, int64_t nru, int32_t id_min, int32_t nid,
                               const int64_t *midsum, int32_t S, const double *sj, const double *si,
                               const int32_t *tblb, const int32_t *tlb, const double *T, int32_t ntblb, int32_t ntlb,
                               const double *distfun, double *bmean, double *tmean, double *fullmean) {
    for (int s = 0; s < S; s++) {
        const double *p = distfun + 10 * s;
        for (int64_t r = 0; r < nru; r++) {
            const int32_t b = bait[r] - id_min, o = oe[r] - id_min;
            if (b < 0 || b >= nid || o < 0 || o >= nid) return -1;
            const double dist = nearbyint((double)(midsum[o] - midsum[b]) / 2.0); /* R round(): half to even */
            const double s_j = sj[(int64_t)s * nid + b];
            double s_i = si[(int64_t)s * nid + o];
            if (isnan(s_i)) s_i = 1.0;
            const double B = s_j * s_i * dist_fun(fabs(dist), p);
            const int32_t tb = tblb[(int64_t)s * nid + b], tl = tlb[(int64_t)s * nid + o];
            double Tm = NAN;
            if (tb >= 0 && tl >= 0) {
                Tm = T[((int64_t)s * ntblb + tb) * ntlb + tl];
            } else if (tb >= 0) { /* tlb missing: lowest Tmean of this tblb */
                double m = INFINITY;
                for (int k = 0; k < ntlb; k++) {
                    const double v = T[((int64_t)s * ntblb + tb) * ntlb + k];
                    if (!isnan(v) && v < m) m = v;
                }
                Tm = isfinite(m) ? m : NAN;
            }
            if (bmean) bmean[(int64_t)s * nru + r] = B;
            if (tmean) tmean[(int64_t)s * nru + r] = Tm;
            if (fullmean) fullmean[(int64_t)s * nru + r] = B + Tm;
        }
    }
    return 0;
}

/* f3. Application side of IHWcorrection, chicdiff.R:2038-2049:
 *   out$group  <- as.integer(cut(log(abs(out$avDist)), breaks))          (:2040; cut(): right-closed (b_k, b_k+1])
 *   merge(out, distLookup[, c("group","avWeights")], all.x=TRUE)          (:2045; NA group -> NA weight)
 *   out$weight <- out$avWeights / mean(out$avWeights)                     (:2046; no na.rm)
 *   weighted_pvalue := pvalue / weight ; weighted_padj := p.adjust(., "BH")   (:2047-2049)
 * group codes are 1-based, INT32_MIN = NA_integer_.  Pinned by the golden table's group / weight /
 * weighted_pvalue / weighted_padj columns (tests/test_results_postprocessing.py). */
int oracle_ihw_apply(const double *avDist, const double *pvalue, int64_t n, const double *breaks, const double *avWeights,
                     int32_t ngroups, int32_t *group, double *weight, double *wp, double *wpadj) {
    long double sum = 0;
    for (int64_t i = 0; i < n; i++) {
        const double x = log(fabs(avDist[i]));
        int g = -1;
        if (!isnan(x))
            for (int k = 0; k < ngroups; k++)
                if (x > breaks[k] && x <= breaks[k + 1]) { g = k; break; }
        if (group) group[i] = g >= 0 ? g + 1 : INT32_MIN;
        weight[i] = g >= 0 ? avWeights[g] : NAN;
        sum += weight[i];
    }
    const double mean = (double)(sum / (long double)n);
    for (int64_t i = 0; i < n; i++) {
        weight[i] = weight[i] / mean;
        wp[i] = pvalue[i] / weight[i];
    }
    return oracle_bh_adjust(wp, n, wpadj);
}

/* f4. getRegionUniverse, window mode, chicdiff.R:353-426.  For peak i (regionID i+1) the candidate otherEndIDs
 * are .expandAvoidBait(baitID, oeID, s) (:353-367; R's a:b counts down when a > b), kept when the ID is on the
 * restriction map (the rmap join, :389-391, and `otherEndID <= maxfrag`, :384) and on the bait's chromosome
 * (:399).  chr_of[0..maxfrag], -1 = not on the map.  Rows come out in (regionID, otherEndID) order; pass
 * ru_* = NULL to count only.  Returns the number of rows, or -1 for baitID == oeID (the reference stops). */
int64_t oracle_region_universe(const int32_t *bait, const int32_t *oe, int64_t n, int32_t s, const int32_t *chr_of,
                               int32_t maxfrag, int64_t *region_ptr, int32_t *ru_bait, int32_t *ru_region, int32_t *ru_oe) {
    int64_t pos = 0;
    for (int64_t i = 0; i < n; i++) {
        const int b = bait[i], o = oe[i];
        int a, e;
        if (abs(b - o) > s + 1) { a = o - s; e = o + s; }
        else if (o > b) { a = b + 2; e = o + s; }
        else if (o < b) { a = o - s; e = b - 2; }
        else return -1;
        const int lo = a < e ? a : e, hi = a < e ? e : a;
        if (region_ptr) region_ptr[i] = pos;
        const int bc = (b >= 1 && b <= maxfrag) ? chr_of[b] : -1;
        for (int id = lo; id <= hi; id++) {
            if (id < 1 || id > maxfrag || bc < 0 || chr_of[id] != bc) continue;
            if (ru_bait) { ru_bait[pos] = b; ru_region[pos] = (int32_t)(i + 1); ru_oe[pos] = id; }
            pos++;
        }
    }
    if (region_ptr) region_ptr[n] = pos;
    return pos;
}

/* IHWcorrection's covariate, chicdiff.R:1965-1967 (test set) / :1980-1982 (control set):
 *   RU.distances <- RU.recast[, list(avDist = mean(distSign)), by = "regionID"]
 * with the long table's distSign from chicdiff.R:868-882:
 *   rmap[, midpoint := round(0.5 * (start + end))]                              (:871; R's round(): half to even)
 *   x <- merge(x, rmap, by.x = "otherEndID", ...); x <- merge(x, rmap, by.x = "baitID", ...)   (:873-874: inner joins —
 *        a row whose fragment the map does not hold is dropped)
 *   distSign := ifelse(chr.x == chr.y, midpoint.x - midpoint.y, NA)              (:877-880; .x = other end, .y = bait)
 * The long table repeats every (region, fragment) row once per sample, so its mean equals the mean over the region's
 * RU rows; mean() without na.rm is NA as soon as one row is NA.  Rows [region_ptr[i], region_ptr[i+1]) = region i.
 * Pinned by the reference's own result table (avDist column, 24 863 regions, exact: tests/test_results_postprocessing.py). */
int oracle_region_avdist(const int32_t *ru_bait, const int32_t *ru_oe, const int64_t *region_ptr, int64_t n, int32_t id_min,
                         int32_t nid, const int64_t *midsum, const int32_t *chr, double *avDist) {
    for (int64_t i = 0; i < n; i++) {
        long double sum = 0; /* data.table's gmean accumulates in long double */
        int64_t cnt = 0;
        int na = 0;
        for (int64_t r = region_ptr[i]; r < region_ptr[i + 1]; r++) {
            const int32_t b = ru_bait[r] - id_min, o = ru_oe[r] - id_min;
            if (b < 0 || b >= nid || o < 0 || o >= nid) continue;
            if (chr && (chr[b] < 0 || chr[o] < 0)) continue;
            cnt++;
            if (chr && chr[b] != chr[o]) { na = 1; continue; }
            const double mo = nearbyint(0.5 * (double)midsum[o]), mb = nearbyint(0.5 * (double)midsum[b]);
            sum += (long double)(mo - mb);
        }
        avDist[i] = (na || cnt == 0) ? NAN : (double)(sum / (long double)cnt);
    }
    return 0;
}

/* a1 without chinput files, chicdiff.R:774-807 (getFullRegionData1) = :1202-1260 (getFullRegionData2):
 *   tempForCounts[[i]] <- x[, c("baitID", "otherEndID", "N")], keyed (baitID, otherEndID)              (:742-747)
 *   mergedFiles <- Reduce(merge, tempForCounts)        -- merge() default: INNER join over the replicates (:779)
 *   countData[[i]] <- mergedFiles[, c(baitID, otherEndID, N.i)][J(baits), ]                            (:782-787)
 *   x <- merge(x, temp, all.x = TRUE); x[is.na(N), N := 0]                                             (:799-800)
 * keys[s] ascending (baitID << 32 | otherEndID), one table per replicate; out is nru x S column-major. */
int oracle_count_join_inner(const int32_t *ru_bait, const int32_t *ru_oe, int64_t nru, int32_t S, const int64_t *const *keys,
                            const int32_t *const *vals, const int64_t *nkeys, int32_t *out) {
    for (int64_t r = 0; r < nru; r++) {
        const int64_t key = ((int64_t)ru_bait[r] << 32) | (uint32_t)ru_oe[r];
        int all = 1;
        for (int s = 0; s < S; s++) {
            int64_t lo = 0, hi = nkeys[s];
            while (lo < hi) {
                const int64_t mid = lo + ((hi - lo) >> 1);
                if (keys[s][mid] < key) lo = mid + 1; else hi = mid;
            }
            if (lo < nkeys[s] && keys[s][lo] == key) out[(int64_t)s * nru + r] = vals[s][lo];
            else all = 0;
        }
        if (!all)
            for (int s = 0; s < S; s++) out[(int64_t)s * nru + r] = 0;
    }
    return 0;
}
