// Issue cost of the 32-bit / carry / packed instructions the Goldilocks primitives are built from (gfx950), 1..8 waves per SIMD.
// Complements valu_peak.hip (64-bit forms).  Each kernel issues ITERS x 64 independent instructions per wave.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define ITERS 1024
#define K32(NAME, ASM)                                                                                             \
    __global__ void k_##NAME(uint32_t* out, uint32_t seed) {                                                       \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed * 7 + threadIdx.x, d = seed + 11;              \
        uint32_t x = seed | 1, y = seed + 77;                                                                      \
        for (int i = 0; i < ITERS; i++) asm volatile(REP16(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y) : "vcc", "s10", "s11", "s12", "s13"); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;                                                \
    }
K32(add_u32, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
K32(add_co_u32, "v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, vcc, %1, %4\n v_add_co_u32 %2, vcc, %2, %4\n v_add_co_u32 %3, vcc, %3, %4\n")
K32(add_co_e64, "v_add_co_u32_e64 %0, s[10:11], %0, %4\n v_add_co_u32_e64 %1, s[12:13], %1, %4\n v_add_co_u32_e64 %2, s[10:11], %2, %4\n v_add_co_u32_e64 %3, s[12:13], %3, %4\n")
// carry chain with the two wait states filled by independent work: add_co a; add_co(e64) b; addc a; addc b
K32(addc_pairs, "v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32_e64 %1, s[10:11], %1, %4\n v_add_u32 %2, %2, %5\n v_addc_co_u32 %0, vcc, %0, %5, vcc\n v_add_u32 %3, %3, %5\n v_addc_co_u32_e64 %1, s[10:11], %1, %5, s[10:11]\n")
// the same chain with s_nop instead of independent work (6 VALU + nops)
K32(addc_nops, "v_add_co_u32 %0, vcc, %0, %4\n s_nop 1\n v_addc_co_u32 %0, vcc, %0, %5, vcc\n v_add_co_u32 %1, vcc, %1, %4\n s_nop 1\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n")
K32(cndmask, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n")
K32(cmp_u32, "v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %4\n v_cmp_lt_u32 vcc, %2, %4\n v_cmp_lt_u32 vcc, %3, %4\n")
K32(alignbit, "v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %4, 7\n v_alignbit_b32 %2, %2, %4, 7\n v_alignbit_b32 %3, %3, %4, 7\n")
K32(perm_b32, "v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5\n")
K32(add3_u32, "v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5\n")
K32(lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %4\n v_lshl_add_u32 %1, %1, 3, %4\n v_lshl_add_u32 %2, %2, 3, %4\n v_lshl_add_u32 %3, %3, 3, %4\n")
K32(bfe_u32, "v_bfe_u32 %0, %0, 3, 9\n v_bfe_u32 %1, %1, 3, 9\n v_bfe_u32 %2, %2, 3, 9\n v_bfe_u32 %3, %3, 3, 9\n")
K32(mad_u32_u24, "v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5\n")
K32(mul_u32_u24, "v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4\n")
K32(mul_hi_u32, "v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n")
K32(dot4_u32_u8, "v_dot4_u32_u8 %0, %4, %5, %0\n v_dot4_u32_u8 %1, %4, %5, %1\n v_dot4_u32_u8 %2, %4, %5, %2\n v_dot4_u32_u8 %3, %4, %5, %3\n")
K32(pk_add_u16, "v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4\n")
K32(pk_mad_u16, "v_pk_mad_u16 %0, %0, %4, %5\n v_pk_mad_u16 %1, %1, %4, %5\n v_pk_mad_u16 %2, %2, %4, %5\n v_pk_mad_u16 %3, %3, %4, %5\n")
K32(mov_dpp, "v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n")
// mixed stream: one half-rate mad_u64_u32 per three full-rate adds (is the cost additive?)
__global__ void k_mix(uint32_t* out, uint32_t seed) {
    uint64_t p = seed, q = seed + 5;
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed * 7 + threadIdx.x, d = seed + 11, x = seed | 1, y = seed + 77;
    for (int i = 0; i < ITERS; i++)
        asm volatile(REP16("v_mad_u64_u32 %4, vcc, %6, %7, %4\n v_add_u32 %0, %0, %6\n v_add_u32 %1, %1, %6\n v_add_u32 %2, %2, %6\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(p), "+v"(q) : "v"(x), "v"(y) : "vcc");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + (uint32_t)p + (uint32_t)q;
}
typedef void (*kern_t)(uint32_t*, uint32_t);
int main() {
    uint32_t* d; (void)hipMalloc((void**)&d, (size_t)256 * 8 * 1024 * 4 + 64);
    struct { const char* n; kern_t f; int per; } es[] = {
        {"v_add_u32", k_add_u32, 4}, {"v_add_co_u32 (vcc)", k_add_co_u32, 4}, {"v_add_co_u32_e64 (sgpr)", k_add_co_e64, 4},
        {"add_co+addc x2 interleaved (6)", k_addc_pairs, 6}, {"add_co,nop,addc x2 (4 VALU)", k_addc_nops, 4},
        {"v_cndmask_b32", k_cndmask, 4}, {"v_cmp_lt_u32", k_cmp_u32, 4}, {"v_alignbit_b32", k_alignbit, 4}, {"v_perm_b32", k_perm_b32, 4},
        {"v_add3_u32", k_add3_u32, 4}, {"v_lshl_add_u32", k_lshl_add_u32, 4}, {"v_bfe_u32", k_bfe_u32, 4}, {"v_mad_u32_u24", k_mad_u32_u24, 4},
        {"v_mul_u32_u24", k_mul_u32_u24, 4}, {"v_mul_hi_u32", k_mul_hi_u32, 4}, {"v_dot4_u32_u8", k_dot4_u32_u8, 4}, {"v_pk_add_u16", k_pk_add_u16, 4},
        {"v_pk_mad_u16", k_pk_mad_u16, 4}, {"v_mov_b32_dpp", k_mov_dpp, 4}, {"1 mad_u64_u32 + 3 add_u32", k_mix, 4}};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-34s %10s %14s %12s\n", "instr", "waves/SIMD", "Ginstr/s/SIMD", "cyc@2.4GHz");
    for (auto& e : es)
        for (int wps : {1, 2, 4, 8}) {
            int blocks = 256 * wps;
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, d, 12345u);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, d, 12345u);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            double instr_per_simd = 16.0 * e.per * ITERS * wps;
            double rate = instr_per_simd / (ms * 1e-3);
            printf("%-34s %10d %14.3f %12.2f\n", e.n, wps, rate / 1e9, 2.4e9 / rate);
        }
    return 0;
}
