// Which gfx950 VALU opcodes run at the fast (~2.4 cycle) rate?  Pure streams of one opcode (4 independent chains, no SGPR result)
// and the same opcode alternating with v_add_u32 / v_mov_b32.  8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define ITERS 1024
#define KC(NAME, A0, A1, A2, A3)                                                                                   \
    __global__ void kp_##NAME(uint32_t* out, uint32_t seed) {                                                      \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed * 7 + threadIdx.x, d = seed + 11, e = seed + 1, f = seed + 2, g = seed + 3, h = seed + 4, x = seed | 1, y = seed + 77; \
        asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a), "v"(x) : "vcc");                                      \
        for (int i = 0; i < ITERS; i++)                                                                            \
            asm volatile(REP16(A0 "\n" A1 "\n" A2 "\n" A3 "\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y)); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;                                \
    }                                                                                                              \
    __global__ void km_##NAME(uint32_t* out, uint32_t seed) {                                                      \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = seed * 7 + threadIdx.x, d = seed + 11, e = seed + 1, f = seed + 2, g = seed + 3, h = seed + 4, x = seed | 1, y = seed + 77; \
        asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(a), "v"(x) : "vcc");                                      \
        for (int i = 0; i < ITERS; i++)                                                                            \
            asm volatile(REP16(A0 "\n v_add_u32 %4, %4, %8\n" A1 "\n v_add_u32 %5, %5, %9\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y)); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;                                \
    }
#define K2(NAME, OP) KC(NAME, OP " %0, %0, %8", OP " %1, %1, %9", OP " %2, %2, %8", OP " %3, %3, %9")
#define K2R(NAME, OP) KC(NAME, OP " %0, %8, %0", OP " %1, %9, %1", OP " %2, %8, %2", OP " %3, %9, %3")
#define K1(NAME, OP) KC(NAME, OP " %0, %8", OP " %1, %9", OP " %2, %8", OP " %3, %9")
#define K3(NAME, OP) KC(NAME, OP " %0, %0, %8, %9", OP " %1, %1, %9, %8", OP " %2, %2, %8, %9", OP " %3, %3, %9, %8")
K2(add_u32, "v_add_u32") K2(sub_u32, "v_sub_u32") K2(subrev_u32, "v_subrev_u32") K2(and_b32, "v_and_b32") K2(or_b32, "v_or_b32") K2(xor_b32, "v_xor_b32")
K2R(lshlrev_b32, "v_lshlrev_b32") K2R(lshrrev_b32, "v_lshrrev_b32") K2R(ashrrev_i32, "v_ashrrev_i32")
K2(max_u32, "v_max_u32") K2(min_u32, "v_min_u32") K1(mov_b32, "v_mov_b32") K1(not_b32, "v_not_b32")
K2(add_f32, "v_add_f32") K2(mul_f32, "v_mul_f32") K3(fma_f32, "v_fma_f32") K2(mul_lo_u32, "v_mul_lo_u32")
KC(cndmask, "v_cndmask_b32 %0, %0, %8, vcc", "v_cndmask_b32 %1, %1, %9, vcc", "v_cndmask_b32 %2, %2, %8, vcc", "v_cndmask_b32 %3, %3, %9, vcc")
KC(addc_in, "v_addc_co_u32_e64 %0, s[20:21], %0, %8, vcc", "v_addc_co_u32_e64 %1, s[22:23], %1, %9, vcc", "v_addc_co_u32_e64 %2, s[20:21], %2, %8, vcc", "v_addc_co_u32_e64 %3, s[22:23], %3, %9, vcc")
KC(add_co_2sg, "v_add_co_u32_e64 %0, s[20:21], %0, %8", "v_add_co_u32_e64 %1, s[22:23], %1, %9", "v_add_co_u32_e64 %2, s[24:25], %2, %8", "v_add_co_u32_e64 %3, s[26:27], %3, %9")
KC(sub_co_2sg, "v_sub_co_u32_e64 %0, s[20:21], %0, %8", "v_sub_co_u32_e64 %1, s[22:23], %1, %9", "v_sub_co_u32_e64 %2, s[24:25], %2, %8", "v_sub_co_u32_e64 %3, s[26:27], %3, %9")
KC(cmp_2sg, "v_cmp_lt_u32_e64 s[20:21], %0, %8", "v_cmp_lt_u32_e64 s[22:23], %1, %9", "v_cmp_lt_u32_e64 s[24:25], %2, %8", "v_cmp_lt_u32_e64 s[26:27], %3, %9")
KC(alignbit, "v_alignbit_b32 %0, %0, %8, 5", "v_alignbit_b32 %1, %1, %9, 5", "v_alignbit_b32 %2, %2, %8, 5", "v_alignbit_b32 %3, %3, %9, 5")
KC(add_e64, "v_add_u32_e64 %0, %0, %8", "v_add_u32_e64 %1, %1, %9", "v_add_u32_e64 %2, %2, %8", "v_add_u32_e64 %3, %3, %9")
KC(and_or, "v_and_or_b32 %0, %0, %8, %9", "v_and_or_b32 %1, %1, %9, %8", "v_and_or_b32 %2, %2, %8, %9", "v_and_or_b32 %3, %3, %9, %8")
KC(bfi, "v_bfi_b32 %0, %0, %8, %9", "v_bfi_b32 %1, %1, %9, %8", "v_bfi_b32 %2, %2, %8, %9", "v_bfi_b32 %3, %3, %9, %8")
typedef void (*kern_t)(uint32_t*, uint32_t);
#define E(NAME) {#NAME, kp_##NAME, km_##NAME}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    uint32_t* d; (void)hipMalloc((void**)&d, (size_t)256 * 8 * 1024 * 4 + 64);
    struct { const char* n; kern_t fp, fm; } es[] = {E(add_u32), E(add_e64), E(sub_u32), E(subrev_u32), E(and_b32), E(or_b32), E(xor_b32), E(lshlrev_b32), E(lshrrev_b32), E(ashrrev_i32),
        E(max_u32), E(min_u32), E(mov_b32), E(not_b32), E(add_f32), E(mul_f32), E(fma_f32), E(mul_lo_u32), E(cndmask), E(addc_in), E(add_co_2sg), E(sub_co_2sg), E(cmp_2sg), E(alignbit), E(and_or), E(bfi)};
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-16s %18s %28s\n", "opcode", "pure: cyc/instr", "alternating with add_u32: cyc/pair");
    for (auto& e : es) {
        double r[2];
        for (int v = 0; v < 2; v++) {
            kern_t f = v ? e.fm : e.fp;
            int blocks = 256 * 8;
            hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d, 12345u);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(f, dim3(blocks), dim3(256), 0, 0, d, 12345u);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            r[v] = ms * 1e-3 * 2.4e9 / (16.0 * ITERS * 8) / (v ? 2 : 4);
        }
        printf("%-16s %18.2f %28.2f\n", e.n, r[0], r[1]);
    }
    return 0;
}
