// Absolute VALU issue rate of gfx950: full-chip launch of a known number of independent integer instructions,
// timed with HIP events.  Prints wave-instructions per second per SIMD and the implied cycles per instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define ITERS 2048
#define K(NAME, ASM)                                                                                               \
    __global__ void k_##NAME(uint64_t* out, uint32_t seed) {                                                       \
        uint64_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed * 7 + threadIdx.x, d = seed + 11;              \
        uint64_t e = a ^ 5, f = b ^ 9, g = c ^ 3, h = d ^ 17;                                                      \
        uint32_t x = seed | 1, y = seed + 77;                                                                      \
        for (int i = 0; i < ITERS; i++) asm volatile(REP16(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "vcc"); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;                                \
    }
K(mad_u64_u32, "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n")
K(lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %5\n v_lshl_add_u64 %2, %2, 0, %6\n v_lshl_add_u64 %3, %3, 0, %7\n")
K(cmp_lt_u64, "v_cmp_lt_u64 vcc, %0, %4\n v_cmp_lt_u64 vcc, %1, %5\n v_cmp_lt_u64 vcc, %2, %6\n v_cmp_lt_u64 vcc, %3, %7\n")
K(lshlrev_b64, "v_lshlrev_b64 %0, 5, %0\n v_lshlrev_b64 %1, 5, %1\n v_lshlrev_b64 %2, 5, %2\n v_lshlrev_b64 %3, 5, %3\n")
K(mov_b64, "v_mov_b64 %0, %4\n v_mov_b64 %1, %5\n v_mov_b64 %2, %6\n v_mov_b64 %3, %7\n")
typedef void (*kern_t)(uint64_t*, uint32_t);
__global__ void k_add_u32(uint64_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed * 7, d = seed + 11, x = seed | 1;
    for (int i = 0; i < ITERS; i++) asm volatile(REP16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x));
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
__global__ void k_mul_lo(uint64_t* out, uint32_t seed) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed * 7, d = seed + 11, x = seed | 1;
    for (int i = 0; i < ITERS; i++) asm volatile(REP16("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x));
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
int main() {
    uint64_t* d; hipMalloc((void**)&d, (size_t)256 * 8 * 1024 * 8 + 64);
    struct { const char* n; kern_t f; } es[] = {{"v_add_u32", k_add_u32}, {"v_mul_lo_u32", k_mul_lo}, {"v_mad_u64_u32", k_mad_u64_u32}, {"v_lshl_add_u64", k_lshl_add_u64},
                                                {"v_cmp_lt_u64", k_cmp_lt_u64}, {"v_lshlrev_b64", k_lshlrev_b64}, {"v_mov_b64", k_mov_b64}};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-16s %8s %14s %12s\n", "instr", "waves/SIMD", "Ginstr/s/SIMD", "cyc@2.4GHz");
    for (auto& e : es)
        for (int wps : {1, 2, 4, 8}) {
            int blocks = 256 * wps;          // 256-thread blocks: 4 waves = 1 per SIMD; wps blocks per CU
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, d, 12345u);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, d, 12345u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double instr_per_simd = 64.0 * ITERS * wps;
            double rate = instr_per_simd / (ms * 1e-3);
            printf("%-16s %8d %14.3f %12.2f\n", e.n, wps, rate / 1e9, 2.4e9 / rate);
        }
    return 0;
}
