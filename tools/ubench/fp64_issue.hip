// Micro-benchmark (diagnostic, not part of the library): issue cost of the fp64 VALU instructions the line-search kernels
// are made of, for 1, 2 and 4 waves per SIMD, dependent chain vs 8 independent chains.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/fp64_issue tools/ubench/fp64_issue.hip && gpurun_out/fp64_issue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP 512
template <int OP, int ILP>
__global__ void k(double *out, unsigned long long *cyc, double seed) {
    double x[8];
    for (int i = 0; i < 8; i++) x[i] = seed + threadIdx.x * 1e-3 + i;
    const double c1 = 1.0000001, c2 = 1e-9;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = ILP == 1 ? 0 : u;
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c1), "v"(c2));
            if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(c1));
            if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(c2));
            if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[i]));
            if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*(float *)&x[i]) : "v"(1.0000001f), "v"(1e-9f));
            if (OP == 5) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[i]) : "v"(c2));
            if (OP == 6) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x[i]));
            if (OP == 7) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(x[i]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP, int ILP>
void run(const char *name, double *d_out, unsigned long long *d_cyc) {
    for (int wps : {1, 2, 4}) {  // waves per SIMD: blocks of 256 threads = 1 wave per SIMD; one block per CU x wps
        const int threads = 256 * wps > 1024 ? 1024 : 256 * wps;
        const int blocks = 256 * (256 * wps / threads);
        k<OP, ILP><<<blocks, threads>>>(d_out, d_cyc, 1.0);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * threads / 64);
        hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += (double)v;
        const double per_inst = s / h.size() / (REP * 8.0);
        printf("%-14s ilp %d  waves/SIMD %d : %.2f cycles per instruction per wave -> %.2f cycles per instruction per SIMD\n", name, ILP, wps, per_inst,
               per_inst / wps);
    }
}

int main() {
    double *d_out;
    unsigned long long *d_cyc;
    hipMalloc((void **)&d_out, 8 * 1024 * 1024);
    hipMalloc((void **)&d_cyc, 8 * 65536);
#define BOTH(OP, NAME) run<OP, 1>(NAME, d_out, d_cyc); run<OP, 8>(NAME, d_out, d_cyc);
    BOTH(0, "v_fma_f64")
    BOTH(1, "v_mul_f64")
    BOTH(2, "v_add_f64")
    BOTH(3, "v_rcp_f64")
    BOTH(4, "v_fma_f32")
    BOTH(5, "v_max_f64")
    BOTH(6, "v_ldexp_f64")
    BOTH(7, "v_frexp_mant_f64")
    return 0;
}
