// Which ds_read_b128 address patterns are bank-conflict-free on gfx950?  One wave, 4096 back-to-back reads per pattern,
// cycles per read from s_memtime.  Pattern: byte address of lane l = (l & 15) * 16 * cstride + (l >> 4) * gstride
// (the 16-lane groups of an MFMA B operand: 16 consecutive 16-byte slots per k-group, k-groups `gstride` bytes apart) and
// a few swizzled forms.  Build: hipcc --offload-arch=gfx950 -O2 -o build/probe_lds_b128 tools/probes/probe_lds_b128.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void k(const int* addr, unsigned long long* out, int npat) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 65536 / 16; i += 64) ((uint4*)lds)[i] = make_uint4(i, i, i, i);
    __syncthreads();
    for (int p = 0; p < npat; ++p) {
        const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (unsigned)addr[p * 64 + threadIdx.x];
        uint4 acc = make_uint4(0, 0, 0, 0);
        unsigned long long t0, t1;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
        for (int it = 0; it < 512; ++it) {
            uint4 v0, v1, v2, v3, v4, v5, v6, v7;
            asm volatile(
                "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %8\n\t"
                "ds_read_b128 %4, %8\n\tds_read_b128 %5, %8\n\tds_read_b128 %6, %8\n\tds_read_b128 %7, %8\n\t"
                "s_waitcnt lgkmcnt(0)"
                : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                : "v"(a)
                : "memory");
            acc.x ^= v0.x ^ v1.x ^ v2.x ^ v3.x ^ v4.x ^ v5.x ^ v6.x ^ v7.x;
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (threadIdx.x == 0) out[p] = t1 - t0;
        if (acc.x == 0x12345678u) out[npat + p] = acc.x;   // keep the reads alive
    }
}
int main() {
    struct Pat { const char* name; std::vector<int> a; };
    std::vector<Pat> pats;
    auto add = [&](const char* name, auto f) {
        Pat p{name, std::vector<int>(64)};
        for (int l = 0; l < 64; ++l) p.a[l] = f(l) & 0xFFF0;
        pats.push_back(p);
    };
    add("linear: lane*16", [](int l) { return l * 16; });
    add("all lanes same address", [](int) { return 0; });
    add("h3 now: n16*16 + g4*1792 (NC=112, k-groups 7x256 B apart)", [](int l) { return (l & 15) * 16 + (l >> 4) * 1792; });
    add("n16*16 + g4*(1792+64)", [](int l) { return (l & 15) * 16 + (l >> 4) * 1856; });
    add("n16*16 + g4*(1792+128)", [](int l) { return (l & 15) * 16 + (l >> 4) * 1920; });
    add("n16*16 + g4*(1792+16)", [](int l) { return (l & 15) * 16 + (l >> 4) * 1808; });
    add("n16*16 + g4*(1792+32)", [](int l) { return (l & 15) * 16 + (l >> 4) * 1824; });
    add("h3 round 2: n16*16 + g4*1568 (NC=98)", [](int l) { return (l & 15) * 16 + (l >> 4) * 1568; });
    add("wino: tile n16 * 2048 + ((g4 ^ n16) << 4)", [](int l) { return (l & 15) * 2048 + (((l >> 4) ^ (l & 15)) << 4); });
    add("n16*2048 + g4*16 (no swizzle)", [](int l) { return (l & 15) * 2048 + (l >> 4) * 16; });
    add("n16*64 + g4*16 (a 64-B row per cell)", [](int l) { return (l & 15) * 64 + (l >> 4) * 16; });
    add("n16*16 + g4*256", [](int l) { return (l & 15) * 16 + (l >> 4) * 256; });
    add("n16*16 + g4*(256+64)", [](int l) { return (l & 15) * 16 + (l >> 4) * 320; });
    add("n16*16 + g4*(256+128)", [](int l) { return (l & 15) * 16 + (l >> 4) * 384; });
    add("n16*16 + g4*512", [](int l) { return (l & 15) * 16 + (l >> 4) * 512; });
    add("n16*32 + g4*16 (stride 32 B)", [](int l) { return (l & 15) * 32 + (l >> 4) * 16; });
    add("n16*16 + 8 (unaligned by 8? no: masked)", [](int l) { return (l & 15) * 16 + (l >> 4) * 1792 + 16 * 3; });
    const int np = (int)pats.size();
    std::vector<int> h(np * 64);
    for (int p = 0; p < np; ++p)
        for (int l = 0; l < 64; ++l) h[p * 64 + l] = pats[p].a[l];
    int* d;
    unsigned long long* o;
    hipMalloc(&d, h.size() * 4);
    hipMalloc(&o, np * 16);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 65536, 0, d, o, np);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> r(np * 2);
    hipMemcpy(r.data(), o, np * 16, hipMemcpyDeviceToHost);
    for (int p = 0; p < np; ++p) printf("%6.2f cycles per ds_read_b128 | %s\n", (double)r[p] / 4096.0, pats[p].name);
    return 0;
}
