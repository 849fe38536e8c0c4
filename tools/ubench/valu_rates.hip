// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the integer VALU ops the Goldilocks
// kernels are made of, on gfx950.  Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x
#define ITERS 256

#define BENCH_KERNEL(NAME, ASM)                                                                      \
    __global__ void k_##NAME(uint64_t* out, uint32_t seed) {                                          \
        uint64_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed * 7 + threadIdx.x, d = seed + 11;  \
        uint64_t e = a ^ 5, f = b ^ 9, g = c ^ 3, h = d ^ 17;                                          \
        uint32_t x = seed | 1, y = seed + 77;                                                          \
        long long t0 = clock64();                                                                      \
        for (int i = 0; i < ITERS; i++) {                                                              \
            asm volatile(REP16(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "vcc"); \
        }                                                                                              \
        long long t1 = clock64();                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;                    \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (uint64_t)(t1 - t0);                  \
    }

#define BENCH_KERNEL32(NAME, ASM)                                                                    \
    __global__ void k_##NAME(uint64_t* out, uint32_t seed) {                                          \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed * 7 + threadIdx.x, d = seed + 11;  \
        uint32_t e = a ^ 5, f = b ^ 9, g = c ^ 3, h = d ^ 17;                                          \
        uint32_t x = seed | 1, y = seed + 77;                                                          \
        long long t0 = clock64();                                                                      \
        for (int i = 0; i < ITERS; i++) {                                                              \
            asm volatile(REP16(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y) : "vcc"); \
        }                                                                                              \
        long long t1 = clock64();                                                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;                    \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (uint64_t)(t1 - t0);                  \
    }

// each ASM body = 4 independent instructions (so 64 per REP16, 64*ITERS per loop)
BENCH_KERNEL(mad_u64_u32, "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n")
BENCH_KERNEL(lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %5\n v_lshl_add_u64 %2, %2, 0, %6\n v_lshl_add_u64 %3, %3, 0, %7\n")
BENCH_KERNEL(cmp_lt_u64, "v_cmp_lt_u64 vcc, %0, %4\n v_cmp_lt_u64 vcc, %1, %5\n v_cmp_lt_u64 vcc, %2, %6\n v_cmp_lt_u64 vcc, %3, %7\n")
BENCH_KERNEL32(add_u32, "v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n")
BENCH_KERNEL32(add_co_u32, "v_add_co_u32 %0, vcc, %0, %8\n v_add_co_u32 %1, vcc, %1, %8\n v_add_co_u32 %2, vcc, %2, %8\n v_add_co_u32 %3, vcc, %3, %8\n")
BENCH_KERNEL32(addc_pair, "v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %9, vcc\n v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %9, vcc\n")
BENCH_KERNEL32(mul_lo_u32, "v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n")
BENCH_KERNEL32(mul_hi_u32, "v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n")
BENCH_KERNEL32(mul_u32_u24, "v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n")
BENCH_KERNEL32(mad_u32_u24, "v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n")
BENCH_KERNEL(lshlrev_b64, "v_lshlrev_b64 %0, 5, %0\n v_lshlrev_b64 %1, 5, %1\n v_lshlrev_b64 %2, 5, %2\n v_lshlrev_b64 %3, 5, %3\n")
BENCH_KERNEL32(alignbit, "v_alignbit_b32 %0, %0, %8, 7\n v_alignbit_b32 %1, %1, %8, 7\n v_alignbit_b32 %2, %2, %8, 7\n v_alignbit_b32 %3, %3, %8, 7\n")
BENCH_KERNEL32(cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n")
BENCH_KERNEL32(add3_u32, "v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n")
BENCH_KERNEL32(pk_add_u16, "v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n")
BENCH_KERNEL32(mad_i32_i24, "v_mad_i32_i24 %0, %0, %8, %9\n v_mad_i32_i24 %1, %1, %8, %9\n v_mad_i32_i24 %2, %2, %8, %9\n v_mad_i32_i24 %3, %3, %8, %9\n")
BENCH_KERNEL(fma_f64, "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %6, %7\n v_fma_f64 %3, %3, %6, %7\n")
BENCH_KERNEL32(mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %8\n v_mul_hi_u32_u24 %1, %1, %8\n v_mul_hi_u32_u24 %2, %2, %8\n v_mul_hi_u32_u24 %3, %3, %8\n")

typedef void (*kern_t)(uint64_t*, uint32_t);
struct Entry { const char* name; kern_t fn; };

int main() {
    uint64_t* d; hipMalloc((void**)&d, ((1 << 20) + 8) * 8);
    Entry es[] = {
#define E(n) {#n, k_##n}
        E(mad_u64_u32), E(lshl_add_u64), E(cmp_lt_u64), E(add_u32), E(add_co_u32), E(addc_pair), E(mul_lo_u32), E(mul_hi_u32),
        E(mul_u32_u24), E(mad_u32_u24), E(mul_hi_u32_u24), E(lshlrev_b64), E(alignbit), E(cndmask), E(add3_u32), E(pk_add_u16), E(mad_i32_i24), E(fma_f64)};
    printf("%-16s %10s %10s %10s   (cycles per wave-instruction per SIMD)\n", "instr", "1w/SIMD", "2w/SIMD", "4w/SIMD");
    for (auto& e : es) {
        double res[3];
        int wpsimd[3] = {1, 2, 4};
        for (int v = 0; v < 3; v++) {
            int threads = 64 * 4 * wpsimd[v];   // one block on one CU: 4 SIMDs x w waves
            hipLaunchKernelGGL(e.fn, dim3(1), dim3(threads), 0, 0, d, 12345u);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(e.fn, dim3(1), dim3(threads), 0, 0, d, 12345u);
            hipDeviceSynchronize();
            uint64_t cyc; hipMemcpy(&cyc, d + (1 << 20), 8, hipMemcpyDeviceToHost);
            double instr_per_wave = 64.0 * ITERS;
            res[v] = (double)cyc / (instr_per_wave * wpsimd[v]);   // SIMD-cycles per instruction
        }
        printf("%-16s %10.2f %10.2f %10.2f\n", e.name, res[0], res[1], res[2]);
    }
    return 0;
}
