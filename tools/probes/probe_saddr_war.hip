// Does a VMEM load in SADDR form read its scalar base at issue?  Each wave issues eight `global_load_dword v, voff, s[20:21]`
// and overwrites s[20:21] with the address of ANOTHER (valid) buffer in the very next instruction -- the pattern hipcc emits
// when it recycles a scalar pair right after a batch of loads.  Buffer A holds 1.0f everywhere, buffer B 2.0f: any load that
// returns 2.0f read the base after the overwrite.  Run at full occupancy so that the vector-memory queues back up.
// Build: hipcc --offload-arch=gfx950 -O2 -o build/probe_saddr_war tools/probes/probe_saddr_war.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(const float* a, const float* b, unsigned* bad, int iters) {
    const unsigned off = (threadIdx.x & 63) * 4 + (blockIdx.x & 63) * 4096;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        float v0, v1, v2, v3, v4, v5, v6, v7;
        asm volatile(
            "s_mov_b64 s[20:21], %[pa]\n\t"
            "s_nop 4\n\t"
            "global_load_dword %0, %[off], s[20:21]\n\t"
            "global_load_dword %1, %[off], s[20:21] offset:256\n\t"
            "global_load_dword %2, %[off], s[20:21] offset:512\n\t"
            "global_load_dword %3, %[off], s[20:21] offset:768\n\t"
            "global_load_dword %4, %[off], s[20:21] offset:1024\n\t"
            "global_load_dword %5, %[off], s[20:21] offset:1280\n\t"
            "global_load_dword %6, %[off], s[20:21] offset:1536\n\t"
            "global_load_dword %7, %[off], s[20:21] offset:1792\n\t"
            "s_mov_b64 s[20:21], %[pb]\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
            : [off] "v"(off), [pa] "s"(a), [pb] "s"(b)
            : "s20", "s21", "memory");
        nbad += (v0 != 1.0f) + (v1 != 1.0f) + (v2 != 1.0f) + (v3 != 1.0f) + (v4 != 1.0f) + (v5 != 1.0f) + (v6 != 1.0f) + (v7 != 1.0f);
    }
    if (nbad) atomicAdd(bad, nbad);
}
// the same for the VGPR-pair address form: the address registers are overwritten by a VALU instruction right after the loads
__global__ __launch_bounds__(256) void k2(const float* a, const float* b, unsigned* bad, int iters) {
    const unsigned long long off = (threadIdx.x & 63) * 4 + (blockIdx.x & 63) * 4096;
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        float v0, v1, v2, v3, v4, v5, v6, v7;
        unsigned long long pa = (unsigned long long)a + off, pb = (unsigned long long)b + off;
        asm volatile(
            "global_load_dword %0, %[pa], off\n\t"
            "global_load_dword %1, %[pa], off offset:256\n\t"
            "global_load_dword %2, %[pa], off offset:512\n\t"
            "global_load_dword %3, %[pa], off offset:768\n\t"
            "global_load_dword %4, %[pa], off offset:1024\n\t"
            "global_load_dword %5, %[pa], off offset:1280\n\t"
            "global_load_dword %6, %[pa], off offset:1536\n\t"
            "global_load_dword %7, %[pa], off offset:1792\n\t"
            "v_lshl_add_u64 %[pa], %[pb], 0, 0\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7), [pa] "+v"(pa)
            : [pb] "v"(pb)
            : "memory");
        nbad += (v0 != 1.0f) + (v1 != 1.0f) + (v2 != 1.0f) + (v3 != 1.0f) + (v4 != 1.0f) + (v5 != 1.0f) + (v6 != 1.0f) + (v7 != 1.0f);
    }
    if (nbad) atomicAdd(bad, nbad);
}
int main() {
    const size_t n = 64 * 4096 / 4 + 4096;   // floats covered by the offsets above
    std::vector<float> ha(n, 1.0f), hb(n, 2.0f);
    float *a, *b;
    unsigned* bad;
    hipMalloc(&a, n * 4);
    hipMalloc(&b, n * 4);
    hipMalloc(&bad, 4);
    hipMemcpy(a, ha.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(b, hb.data(), n * 4, hipMemcpyHostToDevice);
    for (int grid : {256, 2048, 8192}) {
        hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, a, b, bad, 2000);
        hipDeviceSynchronize();
        unsigned h = 0;
        hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("SADDR form, grid %5d x 256 threads, 2000 iterations x 8 loads per lane: %u loads returned the OTHER buffer's value\n", grid, h);
        hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(k2, dim3(grid), dim3(256), 0, 0, a, b, bad, 2000);
        hipDeviceSynchronize();
        hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("VGPR-address form, same launch shape: %u loads returned the OTHER buffer's value\n", h);
    }
    return 0;
}
