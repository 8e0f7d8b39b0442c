// What does a per-layer cross-CU exchange cost?  The small-batch design of DESIGN.md section 7.2 splits one position over C
// workgroups (one per CU); after every layer each writes its slice of the 64 KB Winograd operand to global memory, all C
// meet at a counter, and each reads the whole 64 KB back into LDS.  This probe runs exactly that exchange 200 times for
// groups of C = 2, 4, 8 workgroups (one group or 32 groups at once) and prints microseconds per exchange.  Spins are
// bounded: a group whose partners never arrive gives up and flags it (no hang).  Workgroup b of a group sits at block index
// g * 8 + ... so that, with round-robin dispatch over the 8 XCDs, SAME = 1 puts a group on one XCD (blocks b * 8 + x) and
// SAME = 0 spreads it over XCDs.
// Build: hipcc --offload-arch=gfx950 -O2 -o build/probe_xchg tools/probes/probe_xchg.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int kBytes = 65536;
__global__ __launch_bounds__(512) void k(uint4* buf, unsigned* cnt, unsigned* fail, unsigned long long* out, int C, int groups,
                                         int same, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // group / member of this block
    int grp, mem;
    if (same) {   // members of a group differ by 8 in block index: same XCD under round-robin dispatch
        const int x = blockIdx.x % 8, q = blockIdx.x / 8;   // q in [0, C * groups / 8)
        grp = x + 8 * (q / C);
        mem = q % C;
    } else {
        grp = blockIdx.x / C;
        mem = blockIdx.x % C;
    }
    if (grp >= groups) return;
    uint4* gb = buf + (size_t)grp * 2 * (kBytes / 16);
    unsigned* gc = cnt + grp * 32;
    const int slice = kBytes / 16 / C;   // uint4 per member
    unsigned long long t0 = 0;
    for (int it = 0; it < iters; ++it) {
        if (it == 8) t0 = __builtin_amdgcn_s_memrealtime();
        uint4* cur = gb + (size_t)(it & 1) * (kBytes / 16);
        // 1. write my slice (from LDS contents)
        for (int i = threadIdx.x; i < slice; i += 512) {
            uint4 v = ((uint4*)lds)[mem * slice + i];
            v.x += it;
            cur[mem * slice + i] = v;
        }
        __threadfence();   // release at agent scope
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(gc, 1u);
            unsigned spins = 0;
            while (__hip_atomic_load(gc, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(C * (it + 1))) {
                if (++spins > 4000000u) {   // partners missing: give up, flag, and let every later wait fall through
                    atomicAdd(fail, 1u);
                    atomicAdd(gc, 1000000000u);
                    break;
                }
            }
        }
        __syncthreads();
        __threadfence();   // acquire for the whole block (invalidate L1)
        // 2. read the whole operand back into LDS
        for (int i = threadIdx.x; i < kBytes / 16; i += 512) {
            const uint4 v = cur[i];   // after the acquire fence: L1 holds no stale line of this buffer
            ((uint4*)lds)[i] = v;
        }
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && mem == 0) out[grp] = t1 - t0;
}
int main() {
    uint4* buf;
    unsigned *cnt, *fail;
    unsigned long long* out;
    const int maxg = 32;
    hipMalloc(&buf, (size_t)maxg * 2 * kBytes);
    hipMalloc(&cnt, maxg * 32 * 4);
    hipMalloc(&fail, 4);
    hipMalloc(&out, maxg * 8);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kBytes + 69632);   // 133 KB: one block per CU
    const int iters = 208;
    for (int same = 1; same >= 0; --same)
        for (int C : {2, 4, 8})
            for (int groups : {1, 8, 32}) {
                if (C * groups > 256) continue;
                hipMemset(cnt, 0, maxg * 32 * 4);
                hipMemset(fail, 0, 4);
                hipMemset(out, 0, maxg * 8);
                const int grid = same ? ((groups + 7) / 8) * 8 * C : C * groups;
                hipLaunchKernelGGL(k, dim3(grid), dim3(512), kBytes + 69632, 0, buf, cnt, fail, out, C, groups, same, iters);
                hipDeviceSynchronize();
                std::vector<unsigned long long> h(maxg);
                unsigned f = 0;
                hipMemcpy(h.data(), out, maxg * 8, hipMemcpyDeviceToHost);
                hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
                double mx = 0, sm = 0;
                for (int g = 0; g < groups; ++g) {
                    const double us = (double)h[g] * 0.01 / (iters - 8);
                    sm += us;
                    if (us > mx) mx = us;
                }
                printf("%s XCD, C = %d workgroups per position, %2d positions at once: %.2f us per exchange (max %.2f)%s\n",
                       same ? "same " : "mixed", C, groups, sm / groups, mx, f ? "  [GAVE UP: partners not co-resident]" : "");
            }
    return 0;
}
