// Micro-benchmark (diagnostic, not part of the library): what ONE wave per SIMD pays for the operations the launch's end is made
// of besides fp64 arithmetic — LDS round trips, lane permutes, DPP moves, ballots feeding scalar branches, selects, taken
// branches, lane reads — each as a dependent chain (the end of a line-search launch is one serial chain per row).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/lone_wave_latency tools/ubench/lone_wave_latency.hip && tools/ubench/bin/lone_wave_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP 256
template <int OP>
__global__ void k(double *out, unsigned long long *cyc, double seed, int one) {
    __shared__ double s_buf[1024];
    const int lane = threadIdx.x & 63;
    double x = seed + lane * 1e-3, y = 1.0;
    int idx = (lane * 17 + 3) & 63;
    s_buf[threadIdx.x] = x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < REP; r++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) {  // LDS write -> wait -> read (another lane's slot) -> wait
                s_buf[lane + 64 * (threadIdx.x >> 6)] = x;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                x = s_buf[idx + 64 * (threadIdx.x >> 6)] + 1e-9;
                __builtin_amdgcn_wave_barrier();
            }
            if (OP == 1) x = __shfl(x, idx) + 1e-9;            // ds_bpermute x 2 + wait + add
            if (OP == 2) {                                      // DPP row_shl:1 of a double + add
                const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x101, 0xf, 0xf, false);
                const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x101, 0xf, 0xf, false);
                x = x + __hiloint2double(hi, lo) * 1e-9;
            }
            if (OP == 3) {                                      // ballot -> scalar test -> (never taken) branch; select
                const unsigned long long m = __ballot(x > 0.5);
                if (m == 0ull) x = 2.0;
                x = (x > 1e30) ? y : x + 1e-9;
            }
            if (OP == 4) {                                      // LDS read only (table look-up with a data-dependent index) + add
                const int j = (__double2hiint(x) >> 14) & 63;
                x = x + s_buf[j + one * 0] * 1e-9;
            }
            if (OP == 5) {                                      // a taken, wave-uniform branch per step
                if (__builtin_amdgcn_readfirstlane(__double2hiint(x)) & (one << 30)) x += 1e-9; else x += 2e-9;
                asm volatile("" : "+v"(x));
            }
            if (OP == 6) {                                      // divergent if / else (both sides run): exec juggling
                if (lane & one) x += 1e-9; else x -= 1e-9;
                asm volatile("" : "+v"(x));
            }
            if (OP == 7) {                                      // v_readlane to a scalar and back into the chain
                const int h = __builtin_amdgcn_readlane(__double2hiint(x), 5);
                x = x + (double)(h & one) * 1e-9;
            }
            if (OP == 8) {                                      // integer multiply-high / modulo by a constant in the chain
                idx = (idx * 7 + 1) % 5;
                x = x + (double)idx * 1e-9;
            }
            if (OP == 9) {                                      // s_memtime itself
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                x = x + (double)(t & 1) * 1e-9;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + y + idx;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP>
void run(const char *name, double *d_out, unsigned long long *d_cyc) {
    for (int wps : {1, 2}) {
        const int threads = 256 * wps, blocks = 256;
        k<OP><<<blocks, threads>>>(d_out, d_cyc, 1.0, 1);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * threads / 64);
        hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : h) s += (double)v;
        printf("%-52s waves/SIMD %d : %7.1f cycles per step\n", name, wps, s / h.size() / (REP * 8.0));
    }
}

int main() {
    double *d_out;
    unsigned long long *d_cyc;
    hipMalloc((void **)&d_out, 8 * 1024 * 1024);
    hipMalloc((void **)&d_cyc, 8 * 65536);
    run<0>("LDS write, wave barrier, read, + add", d_out, d_cyc);
    run<1>("__shfl of a double (2 ds_bpermute) + add", d_out, d_cyc);
    run<2>("DPP row_shl:1 of a double + fma", d_out, d_cyc);
    run<3>("ballot -> scalar branch (not taken) + select + add", d_out, d_cyc);
    run<4>("LDS table read at a data-dependent index + fma", d_out, d_cyc);
    run<5>("wave-uniform branch + add", d_out, d_cyc);
    run<6>("divergent if / else (both sides) around an add", d_out, d_cyc);
    run<7>("v_readlane -> scalar -> cvt + fma", d_out, d_cyc);
    run<8>("integer (x * 7 + 1) % 5 -> cvt + fma", d_out, d_cyc);
    run<9>("s_memtime -> cvt + fma", d_out, d_cyc);
    return 0;
}
