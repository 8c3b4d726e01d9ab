// chinput.hip — f2: reading a CHiCAGO .chinput file (host side) and handing its three used columns to the device.
//
// Replaces `x <- fread(targetChFiles[i])` and the column pick `x[, c("baitID", "otherEndID", "N")]` of the reference
// (chicdiff.R:828, :849; file format: SURVEY.md Appendix B — optional '#' comment lines, a header line naming the columns
// `baitID otherEndID N otherEndLen distSign`, then one row per (bait, other end) pair, tab-separated; only the first three
// named columns are used and `distSign` may read NA).  The text is mmap()ed and cut into one slice per host thread at
// line boundaries; every thread parses its slice straight into the context's pinned staging area (three int32 columns),
// which one DMA then moves to the device, where chicdiff_hip_count_table_dev's kernels (bait filter, radix sort) build
// the key table the count join searches.  No R, no data.table: plain POSIX + std::thread.
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace cd {

struct ChinputCols {
    std::vector<int32_t> bait, oe, N;
    std::string error;
};

static inline bool is_sep(char ch) { return ch == '\t' || ch == ' ' || ch == ','; }

// one data line [p, e): fields separated by tab / blank / comma; columns ib, io, in (0-based) must be integers
static inline bool parse_line(const char *p, const char *e, int ib, int io, int in, int32_t out[3]) {
    int col = 0, got = 0;
    const int last = ib > io ? (ib > in ? ib : in) : (io > in ? io : in);
    while (p < e && col <= last) {
        const char *q = p;
        while (q < e && !is_sep(*q)) q++;
        if (col == ib || col == io || col == in) {
            const char *s = p;
            bool neg = false;
            if (s < q && (*s == '-' || *s == '+')) { neg = *s == '-'; s++; }
            if (s == q) return false;
            long long v = 0;
            for (; s < q; s++) {
                if (*s < '0' || *s > '9') return false;
                v = v * 10 + (*s - '0');
                if (v > 2147483647LL) return false;
            }
            out[col == ib ? 0 : (col == io ? 1 : 2)] = (int32_t)(neg ? -v : v);
            got++;
        }
        col++;
        p = q < e ? q + 1 : q;
    }
    return got == 3;
}

// Parses the whole file with `nthreads` threads.  Returns the row count, or -1 with `err` set.
int64_t chinput_parse(const char *path, int nthreads, ChinputCols &c) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { c.error = std::string("cannot open ") + path; return -1; }
    struct stat stt;
    if (fstat(fd, &stt) != 0) { close(fd); c.error = "fstat failed"; return -1; }
    const size_t size = (size_t)stt.st_size;
    if (size == 0) { close(fd); c.error = "empty file"; return -1; }
    const char *base = (const char *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (base == MAP_FAILED) { c.error = "mmap failed"; return -1; }
    const char *end = base + size, *p = base;
    // comment lines, then the header
    auto line_end = [&](const char *s) { const char *q = (const char *)memchr(s, '\n', (size_t)(end - s)); return q ? q : end; };
    while (p < end && *p == '#') p = line_end(p) + 1;
    int ib = -1, io = -1, in = -1;
    if (p < end) {
        const char *e = line_end(p);
        const char *he = e;
        if (he > p && he[-1] == '\r') he--;
        int col = 0;
        for (const char *f = p; f <= he; col++) {
            const char *q = f;
            while (q < he && !is_sep(*q)) q++;
            std::string name(f, q);
            if (name.size() >= 2 && name.front() == '"' && name.back() == '"') name = name.substr(1, name.size() - 2);
            if (name == "baitID") ib = col;
            else if (name == "otherEndID") io = col;
            else if (name == "N") in = col;
            if (q >= he) break;
            f = q + 1;
        }
        p = e < end ? e + 1 : end;
    }
    if (ib < 0 || io < 0 || in < 0) {
        munmap((void *)base, size);
        c.error = "chinput header must name the columns baitID, otherEndID and N";
        return -1;
    }
    if (nthreads < 1) nthreads = 1;
    const size_t body = (size_t)(end - p);
    if ((size_t)nthreads > body / 65536 + 1) nthreads = (int)(body / 65536 + 1);
    // slice boundaries at line starts
    std::vector<const char *> cut(nthreads + 1);
    cut[0] = p;
    cut[nthreads] = end;
    for (int t = 1; t < nthreads; t++) {
        const char *s = p + body / nthreads * t;
        if (s < cut[t - 1]) s = cut[t - 1];
        cut[t] = s >= end ? end : (line_end(s) + 1 > end ? end : line_end(s) + 1);
    }
    std::vector<std::vector<int32_t>> tb(nthreads), to(nthreads), tn(nthreads);
    std::vector<long long> bad(nthreads, -1);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++)
            th.emplace_back([&, t]() {
                const char *s = cut[t], *e = cut[t + 1];
                const size_t guess = (size_t)(e - s) / 16 + 16;
                tb[t].reserve(guess); to[t].reserve(guess); tn[t].reserve(guess);
                while (s < e) {
                    const char *le = (const char *)memchr(s, '\n', (size_t)(e - s));
                    if (!le) le = e;
                    const char *ce = le;
                    if (ce > s && ce[-1] == '\r') ce--;
                    if (ce > s) {  // blank lines are skipped, as fread does
                        int32_t v[3];
                        if (!parse_line(s, ce, ib, io, in, v)) { if (bad[t] < 0) bad[t] = (long long)(s - base); s = le + 1; continue; }
                        tb[t].push_back(v[0]); to[t].push_back(v[1]); tn[t].push_back(v[2]);
                    }
                    s = le + 1;
                }
            });
        for (auto &w : th) w.join();
    }
    munmap((void *)base, size);
    for (int t = 0; t < nthreads; t++)
        if (bad[t] >= 0) {
            c.error = "malformed chinput row at byte offset " + std::to_string(bad[t]) + " (baitID, otherEndID and N must be integers)";
            return -1;
        }
    size_t total = 0;
    for (int t = 0; t < nthreads; t++) total += tb[t].size();
    c.bait.resize(total); c.oe.resize(total); c.N.resize(total);
    size_t off = 0;
    for (int t = 0; t < nthreads; t++) {
        memcpy(c.bait.data() + off, tb[t].data(), tb[t].size() * 4);
        memcpy(c.oe.data() + off, to[t].data(), to[t].size() * 4);
        memcpy(c.N.data() + off, tn[t].data(), tn[t].size() * 4);
        off += tb[t].size();
    }
    return (int64_t)total;
}

}  // namespace cd
