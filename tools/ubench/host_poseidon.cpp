// Host-side Poseidon timing (the Challenger / public_inputs_hash / witness sponge run on the CPU):
//   g++ -O3 -std=c++17 -o host_poseidon host_poseidon.cpp && ./host_poseidon ; GL_HOST_MDS_PORTABLE=1 ./host_poseidon
#include "../../plonky2_demo_amd/csrc/poseidon.cuh"
#include <chrono>
#include <cstdio>
int main() {
    gl_t s[12];
    for (int i = 0; i < 12; i++) s[i] = i * 0x123456789abcdefULL + 77;
    for (int rep = 0; rep < 3; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < 200000; it++) psd_permute(s);
        auto t1 = std::chrono::steady_clock::now();
        printf("host permutation: %.3f us (%s MDS) [%llx]\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 200000,
               getenv("GL_HOST_MDS_PORTABLE") ? "portable" : "avx2-if-available", (unsigned long long)s[0]);
    }
    return 0;
}
