// Mixed instruction streams on gfx950: is a "simple" 32-bit VOP1/VOP2 instruction cheaper than a 64-bit / multiply / carry one
// when they share a stream?  (valu_peak / valu_rates2 measure pure streams.)  Reports cycles per GROUP at 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define ITERS 1024
#define KMIX(NAME, ASM)                                                                                            \
    __global__ void k_##NAME(uint32_t* out, uint32_t seed) {                                                       \
        uint64_t p = seed, q = seed + 5, r = seed * 3, s = seed + 9;                                               \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed * 7 + threadIdx.x, d = seed + 11, e = seed + 1, f = seed + 2, x = seed | 1, y = seed + 77; \
        for (int i = 0; i < ITERS; i++)                                                                            \
            asm volatile(REP16(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(p), "+v"(q), "+v"(r), "+v"(s) : "v"(x), "v"(y) : "vcc", "s10", "s11", "s12", "s13"); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + (uint32_t)p + (uint32_t)q + (uint32_t)r + (uint32_t)s; \
    }
// operands: %0-%5 32-bit a..f, %6-%9 64-bit p..s, %10 x, %11 y
KMIX(mov4, "v_mov_b32 %0, %10\n v_mov_b32 %1, %11\n v_mov_b32 %2, %10\n v_mov_b32 %3, %11\n")
KMIX(add4, "v_add_u32 %0, %0, %10\n v_add_u32 %1, %1, %11\n v_add_u32 %2, %2, %10\n v_add_u32 %3, %3, %11\n")
KMIX(and_xor_lsh_sub, "v_and_b32 %0, %0, %10\n v_xor_b32 %1, %1, %11\n v_lshlrev_b32 %2, 3, %2\n v_sub_u32 %3, %3, %11\n")
KMIX(mad1_mov3, "v_mad_u64_u32 %6, vcc, %10, %11, %6\n v_mov_b32 %0, %10\n v_mov_b32 %1, %11\n v_mov_b32 %2, %10\n")
KMIX(mad1_add3, "v_mad_u64_u32 %6, vcc, %10, %11, %6\n v_add_u32 %0, %0, %10\n v_add_u32 %1, %1, %11\n v_add_u32 %2, %2, %10\n")
KMIX(mad2_add2, "v_mad_u64_u32 %6, vcc, %10, %11, %6\n v_add_u32 %0, %0, %10\n v_mad_u64_u32 %7, vcc, %10, %11, %7\n v_add_u32 %1, %1, %11\n")
KMIX(mad4, "v_mad_u64_u32 %6, vcc, %10, %11, %6\n v_mad_u64_u32 %7, vcc, %10, %11, %7\n v_mad_u64_u32 %8, vcc, %10, %11, %8\n v_mad_u64_u32 %9, vcc, %10, %11, %9\n")
KMIX(lshladd1_add3, "v_lshl_add_u64 %6, %6, 0, %7\n v_add_u32 %0, %0, %10\n v_add_u32 %1, %1, %11\n v_add_u32 %2, %2, %10\n")
KMIX(addco1_add3, "v_add_co_u32 %3, vcc, %3, %10\n v_add_u32 %0, %0, %10\n v_add_u32 %1, %1, %11\n v_add_u32 %2, %2, %10\n")
KMIX(cmp64_cnd_add2, "v_cmp_lt_u64 vcc, %6, %7\n v_add_u32 %0, %0, %10\n v_add_u32 %1, %1, %11\n v_cndmask_b32 %2, %2, %10, vcc\n")
KMIX(cmp32_cnd_add2, "v_cmp_lt_u32 vcc, %3, %10\n v_add_u32 %0, %0, %10\n v_add_u32 %1, %1, %11\n v_cndmask_b32 %2, %2, %10, vcc\n")
KMIX(cmp_nop_cnd, "v_cmp_lt_u32 vcc, %3, %10\n s_nop 1\n v_cndmask_b32 %2, %2, %10, vcc\n v_cmp_lt_u32 vcc, %4, %10\n s_nop 1\n v_cndmask_b32 %1, %1, %10, vcc\n")
KMIX(cmp_cnd_3way, "v_cmp_lt_u32 vcc, %3, %10\n v_cmp_lt_u32_e64 s[10:11], %4, %10\n v_cmp_lt_u32_e64 s[12:13], %5, %10\n v_cndmask_b32 %0, %0, %10, vcc\n v_cndmask_b32_e64 %1, %1, %10, s[10:11]\n v_cndmask_b32_e64 %2, %2, %10, s[12:13]\n")
KMIX(salu4_add4, "s_and_b64 s[10:11], s[10:11], s[12:13]\n v_add_u32 %0, %0, %10\n s_or_b64 s[12:13], s[10:11], s[12:13]\n v_add_u32 %1, %1, %11\n s_andn2_b64 s[10:11], s[10:11], s[12:13]\n v_add_u32 %2, %2, %10\n s_xor_b64 s[12:13], s[10:11], s[12:13]\n v_add_u32 %3, %3, %11\n")
KMIX(nop4_add4, "s_nop 0\n v_add_u32 %0, %0, %10\n s_nop 0\n v_add_u32 %1, %1, %11\n s_nop 0\n v_add_u32 %2, %2, %10\n s_nop 0\n v_add_u32 %3, %3, %11\n")
KMIX(nop1x4_mad4, "s_nop 1\n v_mad_u64_u32 %6, vcc, %10, %11, %6\n s_nop 1\n v_mad_u64_u32 %7, vcc, %10, %11, %7\n s_nop 1\n v_mad_u64_u32 %8, vcc, %10, %11, %8\n s_nop 1\n v_mad_u64_u32 %9, vcc, %10, %11, %9\n")
KMIX(salu4_mad4, "s_and_b64 s[10:11], s[10:11], s[12:13]\n v_mad_u64_u32 %6, vcc, %10, %11, %6\n s_or_b64 s[12:13], s[10:11], s[12:13]\n v_mad_u64_u32 %7, vcc, %10, %11, %7\n s_andn2_b64 s[10:11], s[10:11], s[12:13]\n v_mad_u64_u32 %8, vcc, %10, %11, %8\n s_xor_b64 s[12:13], s[10:11], s[12:13]\n v_mad_u64_u32 %9, vcc, %10, %11, %9\n")
typedef void (*kern_t)(uint32_t*, uint32_t);
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    uint32_t* d; (void)hipMalloc((void**)&d, (size_t)256 * 8 * 1024 * 4 + 64);
    struct { const char* n; kern_t f; int valu; } es[] = {
        {"4 v_mov_b32", k_mov4, 4}, {"4 v_add_u32 (two sources)", k_add4, 4}, {"and, xor, lshl, sub (e32)", k_and_xor_lsh_sub, 4},
        {"4 v_mad_u64_u32", k_mad4, 4}, {"1 mad + 3 mov", k_mad1_mov3, 4}, {"1 mad + 3 add_u32", k_mad1_add3, 4}, {"mad, add, mad, add", k_mad2_add2, 4},
        {"1 lshl_add_u64 + 3 add_u32", k_lshladd1_add3, 4}, {"1 add_co + 3 add_u32", k_addco1_add3, 4},
        {"cmp_u64, add, add, cndmask", k_cmp64_cnd_add2, 4}, {"cmp_u32, add, add, cndmask", k_cmp32_cnd_add2, 4},
        {"(cmp, s_nop 1, cndmask) x2", k_cmp_nop_cnd, 4}, {"3 cmp + 3 cndmask interleaved", k_cmp_cnd_3way, 6},
        {"4 SALU + 4 add_u32", k_salu4_add4, 4}, {"4 s_nop 0 + 4 add_u32", k_nop4_add4, 4}, {"4 s_nop 1 + 4 mad", k_nop1x4_mad4, 4}, {"4 SALU + 4 mad", k_salu4_mad4, 4}};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-34s %6s %16s %16s\n", "group", "waves", "cyc/group@2.4GHz", "cyc/VALU@2.4GHz");
    for (auto& e : es)
        for (int wps : {2, 8}) {
            int blocks = 256 * wps;
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, d, 12345u);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(e.f, dim3(blocks), dim3(256), 0, 0, d, 12345u);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            double groups_per_simd = 16.0 * ITERS * wps;
            double cyc = ms * 1e-3 * 2.4e9 / groups_per_simd;
            printf("%-34s %6d %16.2f %16.2f\n", e.n, wps, cyc, cyc / e.valu);
        }
    return 0;
}
