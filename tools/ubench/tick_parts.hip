// tick_parts.hip — one kernel per part of a line-search tick (disp_kernels.hip), so that each part's instructions can be
// counted in isolation (tools/isa_account.py --parts).  Not linked into anything; compiled to assembly only.
#include "../../chicdiff_amd/csrc/disp_kernels.hip"

namespace cd {

#define LT                                 \
    __shared__ LogEntry s_lt[64];          \
    __shared__ ExpEntry s_et[64];          \
    exp_table_to_lds(s_et);                \
    log_table_to_lds(s_lt);

// the empty frame: what every kernel below pays for its loads / stores / table set-up
extern "C" __global__ void part_frame(const double *in, double *out) {
    LT;
    out[threadIdx.x] = in[threadIdx.x] + s_lt[threadIdx.x & 63].invc;
}
extern "C" __global__ void part_row_consts(const double *in, double *out) {
    LT;
    const RowConsts c = row_consts(in[threadIdx.x], s_lt, s_et);
    out[threadIdx.x] = c.alpha + c.r + c.lgS0 + c.dgS0 + (double)c.nr;
}
extern "C" __global__ void part_exp_library(const double *in, double *out) { out[threadIdx.x] = exp(in[threadIdx.x]); }
extern "C" __global__ void part_texp(const double *in, double *out) {
    LT;
    out[threadIdx.x] = texp(in[threadIdx.x], s_et);
}
extern "C" __global__ void part_rcp(const double *in, double *out) { out[threadIdx.x] = rcp(in[threadIdx.x]); }
extern "C" __global__ void part_tlog(const double *in, double *out) {
    LT;
    out[threadIdx.x] = tlog(in[threadIdx.x], s_lt);
}
extern "C" __global__ void part_stirling(const double *in, double *out) {
    double lg, dg;
    stirling(in[threadIdx.x], in[threadIdx.x + 64], in[threadIdx.x + 128], lg, dg);
    out[threadIdx.x] = lg + dg;
}
// one sample, Stirling branch taken / not taken decided per lane (both sides are in the code)
extern "C" __global__ void part_sample(const double *in, const int *iy, double *out) {
    LT;
    RowConsts c;
    c.a = in[0]; c.alpha = in[1]; c.r = in[2]; c.lgS0 = in[3]; c.dgS0 = in[4]; c.nr = iy[0];
    const SampleVals v = sample_values(c, in[8 + threadIdx.x], iy[8 + threadIdx.x], in[72 + threadIdx.x], in[136 + threadIdx.x], s_lt);
    out[threadIdx.x] = v.wj + v.pm + v.tll + v.tsd + (double)v.pe;
}
// the same with the Stirling difference compiled out (a sample with y <= nr)
extern "C" __global__ void part_sample_no_stirling(const double *in, const int *iy, double *out) {
    LT;
    RowConsts c;
    c.a = in[0]; c.alpha = in[1]; c.r = in[2]; c.lgS0 = in[3]; c.dgS0 = in[4]; c.nr = 0x7fffffff;
    const SampleVals v = sample_values(c, in[8 + threadIdx.x], iy[8 + threadIdx.x] & 0xff, in[72 + threadIdx.x], in[136 + threadIdx.x], s_lt);
    out[threadIdx.x] = v.wj + v.pm + v.tll + v.tsd + (double)v.pe;
}
extern "C" __global__ void part_accumulate(const double *in, double *out, int g) {
    Acc acc;
    acc.ll = in[0]; acc.sd = in[1]; acc.wA = in[2]; acc.wB = in[3]; acc.dA = in[4]; acc.dB = in[5]; acc.pm = in[6];
    SampleVals v;
    v.wj = in[8 + threadIdx.x]; v.pm = in[72 + threadIdx.x]; v.tll = in[136 + threadIdx.x]; v.tsd = in[200 + threadIdx.x]; v.pe = g;
    accumulate(acc, v, g & 1);
    out[threadIdx.x] = acc.ll + acc.sd + acc.wA + acc.wB + acc.dA + acc.dB + acc.pm + (double)acc.pe;
}
extern "C" __global__ void part_finish(const double *in, double *out, int p2, int prior) {
    LT;
    Acc acc;
    acc.ll = in[threadIdx.x]; acc.sd = in[64 + threadIdx.x]; acc.wA = in[128 + threadIdx.x]; acc.wB = in[192 + threadIdx.x];
    acc.dA = in[256 + threadIdx.x]; acc.dB = in[320 + threadIdx.x]; acc.pm = in[384 + threadIdx.x]; acc.pe = p2;
    RowConsts c;
    c.a = in[0]; c.alpha = in[1]; c.r = in[2]; c.lgS0 = in[3]; c.dgS0 = in[4]; c.nr = 0;
    double lp, dlp;
    finish_point(acc, c, p2 != 0, prior != 0, in[5], in[6], lp, dlp, s_lt);
    out[threadIdx.x] = lp + dlp;
}
// one step of the prefix walk / table: P *= z; H += rcp(z); z += 1
extern "C" __global__ void part_prefix_step(const double *in, double *out) {
    double P = in[threadIdx.x], H = in[64 + threadIdx.x], zz = in[128 + threadIdx.x];
    P *= zz;
    H += rcp(zz);
    zz += 1.0;
    out[threadIdx.x] = P + H + zz;
}

}  // namespace cd
