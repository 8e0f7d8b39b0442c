// Which workgroups share a CU?  2048 workgroups of 256 threads with 74 KB of LDS (two per CU, like k_trunk16<true,2>):
// each records HW_ID, XCC_ID and its start time.  Build: hipcc --offload-arch=gfx950 -O2 -o build/probe_dispatch tools/probes/probe_dispatch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned* out, int spin) {
    extern __shared__ char lds[];
    unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID, all 32 bits
    unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID, bits 3:0
    unsigned long long t = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = hw;
        out[blockIdx.x * 4 + 1] = xcc;
        out[blockIdx.x * 4 + 2] = (unsigned)t;
    }
    // keep the CU busy so later workgroups queue behind the first 512
    volatile float* p = (volatile float*)lds;
    float acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;
    p[threadIdx.x] = acc;
    if (threadIdx.x == 0) out[blockIdx.x * 4 + 3] = (unsigned)__builtin_amdgcn_s_memrealtime();
}
int main() {
    const int G = 2048;
    unsigned* d;
    hipMalloc(&d, G * 16);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 74240);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(G), dim3(256), 74240, 0, d, 200000);
        hipDeviceSynchronize();
    }
    std::vector<unsigned> h(G * 4);
    hipMemcpy(h.data(), d, G * 16, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> cu;   // key: xcc, se, sh, cu
    unsigned t0 = ~0u;
    for (int b = 0; b < G; ++b) if (h[b * 4 + 2] < t0) t0 = h[b * 4 + 2];
    for (int b = 0; b < G; ++b) {
        unsigned hw = h[b * 4];
        unsigned key = (h[b * 4 + 1] & 15) << 16 | ((hw >> 13) & 7) << 12 | ((hw >> 12) & 1) << 8 | ((hw >> 8) & 15);
        cu[key].push_back(b);
    }
    printf("distinct CUs: %zu\n", cu.size());
    int shown = 0;
    for (auto& kv : cu) {
        if (shown++ >= 24) break;
        printf("xcc %u se %u sh %u cu %2u:", kv.first >> 16, (kv.first >> 12) & 7, (kv.first >> 8) & 1, kv.first & 15);
        for (int b : kv.second) {
            unsigned hw = h[b * 4];
            printf("  b%-4d(tg%u w%u s%u t%u)", b, (hw >> 16) & 15, hw & 15, (hw >> 4) & 3, (h[b * 4 + 2] - t0) / 100);
        }
        printf("\n");
    }
    // statistics: among the first two workgroups of every CU: parity of tg, of b, of b>>8 ...
    int n = 0, tgdiff = 0, b0diff = 0, b8diff = 0, b3diff = 0, wdiff = 0;
    std::map<int, int> delta;
    for (auto& kv : cu) {
        if (kv.second.size() < 2) continue;
        // the two earliest starters
        std::vector<int> v = kv.second;
        std::sort(v.begin(), v.end(), [&](int a, int b) { return h[a * 4 + 2] < h[b * 4 + 2]; });
        int a = v[0], b = v[1];
        ++n;
        tgdiff += ((h[a * 4] >> 16) & 1) != ((h[b * 4] >> 16) & 1);
        wdiff += (h[a * 4] & 1) != (h[b * 4] & 1);
        b0diff += (a & 1) != (b & 1);
        b3diff += ((a >> 3) & 1) != ((b >> 3) & 1);
        b8diff += ((a >> 8) & 1) != ((b >> 8) & 1);
        delta[abs(a - b)]++;
    }
    printf("CUs with >= 2 workgroups: %d; first pair differs in: tg parity %d, wave-slot parity %d, b bit0 %d, b bit3 %d, b bit8 %d\n",
           n, tgdiff, wdiff, b0diff, b3diff, b8diff);
    for (auto& kv : delta) printf("  |b1-b0| = %d : %d CUs\n", kv.first, kv.second);
    std::map<size_t, int> hist;
    for (auto& kv : cu) hist[kv.second.size()]++;
    for (auto& kv : hist) printf("  %zu workgroups on a CU: %d CUs\n", kv.first, kv.second);
    return 0;
}
