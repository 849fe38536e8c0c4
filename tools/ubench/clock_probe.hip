// What clock does the chip actually run integer-VALU-bound kernels at?  Every wave brackets a chain of v_mad_u64_u32 (the NTT /
// Poseidon instruction mix is ~45 % of this class) with s_memtime (shader clock) and s_memrealtime (constant 100 MHz) and the host
// prints cycles / real time, plus the issue cost per instruction in shader cycles.  All VALU-roof figures in profiles/ assume
// 2.4 GHz; SQ_BUSY_CYCLES / duration of the NTT passes says ~2.0 GHz.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/clock_probe.hip -o tools/ubench/bin/clock_probe && tools/ubench/bin/clock_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void k_probe(uint64_t* out, int iters, uint64_t seed) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    uint32_t m = (uint32_t)seed | 1u;
    uint64_t t0 = __builtin_readcyclecounter();          // s_memtime
    uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\t"
                         "v_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
                         "v_mad_u64_u32 %2, vcc, %4, %5, %2\n\t"
                         "v_mad_u64_u32 %3, vcc, %4, %5, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"((uint32_t)threadIdx.x) : "vcc");
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64;
    if ((threadIdx.x & 63) == 0) { out[3 * w] = t1 - t0; out[3 * w + 1] = r1 - r0; out[3 * w + 2] = a0 ^ a1 ^ a2 ^ a3; }
}

int main() {
    const int wgs_per_cu[] = {1, 2, 4, 8};
    printf("%-18s %12s %12s %14s %16s\n", "waves per SIMD", "shader MHz", "wall ms", "cycles/instr", "cycles/instr/SIMD");
    for (int wpc : wgs_per_cu) {
        const int blocks = 256 * wpc, iters = 4000 / wpc * 4;
        const size_t waves = (size_t)blocks * 4;
        uint64_t* d; (void)hipMalloc((void**)&d, waves * 24);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, 0, d, iters, 12345ULL);      // warm-up (clocks ramp)
        hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, 0, d, iters, 12345ULL);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(256), 0, 0, d, iters, 999ULL);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<uint64_t> h(waves * 3);
        (void)hipMemcpy(h.data(), d, waves * 24, hipMemcpyDeviceToHost);
        std::vector<double> mhz, cpi;
        for (size_t w = 0; w < waves; w++) {
            mhz.push_back((double)h[3 * w] / (double)h[3 * w + 1] * 100.0);
            cpi.push_back((double)h[3 * w] / ((double)iters * 64.0));
        }
        std::sort(mhz.begin(), mhz.end()); std::sort(cpi.begin(), cpi.end());
        printf("%-18d %12.0f %12.3f %14.2f %16.2f\n", wpc, mhz[waves / 2], ms, cpi[waves / 2], cpi[waves / 2] / wpc);
        (void)hipFree(d);
    }
    return 0;
}
