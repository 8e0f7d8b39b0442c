// v_pk_fma_f32 with the HIGH dword of a source pair selected for the LOW lane (`op_sel:[0,1,0]`): is it reliable when two
// waves share a SIMD?  (Round 4: the value-head FC1 of k_trunk_h3<32, 8, 1, 4> gave wrong values at ~1-2 % of the positions,
// different ones on every launch, only with two workgroups per CU; builds that differ in NOTHING but this operand form --
// profiles/r04_heads_batch4_variants.txt, variants 6 / 7 / 8 -- are clean with `op_sel_hi:[1,0,1]` (low dword broadcast) and
// with no operand select, and wrong with `op_sel:[0,1,0]`.)
//
// Each wave runs the FC1 shape stand-alone: per batch 32 weight dwords from an L2-resident table and 8 x values from LDS, then
// per x two packed FMAs (4 outputs per lane) in the form under test AND the same four FMAs as scalar v_fma_f32 into a second
// set of accumulators; after `rows` rows the two sets must be bit-identical (packed fp32 is two independent IEEE FMAs).
// MODE 0: x in the LOW dword, op_sel_hi:[1,0,1]   MODE 1: x in the HIGH dword, op_sel:[0,1,0]   MODE 2: {x, x}, no select.
// Launched with 256 workgroups (one per CU: one wave per SIMD), 512 (two waves per SIMD) and 1024 (four; 54 VGPRs).
// Build: hipcc --offload-arch=gfx950 -O2 -o build/probe_pk_opsel tools/probes/probe_pk_opsel.hip ; run: build/probe_pk_opsel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f2 = float __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ wt, const float* __restrict__ xin, unsigned* bad,
                                         unsigned* bad_lanes, int reps) {
    __shared__ float xs[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned nbad = 0;
    for (int rep = 0; rep < reps; ++rep) {
        xs[wave][lane] = xin[((blockIdx.x * 4 + wave + rep) & 1023) * 64 + lane];   // this wave's 64 inputs
        f2 h01 = {0.f, 0.f}, h23 = {0.f, 0.f};
        float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
#pragma unroll 1
        for (int i0 = 0; i0 < 64; i0 += 8) {
            float w[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) w[u][j] = wt[(size_t)(i0 + u) * 256 + lane + 64 * j];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float x = xs[wave][i0 + u];
                f2 xx = MODE == 0 ? f2{x, 0.f} : MODE == 1 ? f2{0.f, x} : f2{x, x};
                asm volatile("" : "+v"(xx));
                const f2 w01 = {w[u][0], w[u][1]}, w23 = {w[u][2], w[u][3]};
                if (MODE == 0) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(h01) : "v"(w01), "v"(xx));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(h23) : "v"(w23), "v"(xx));
                } else if (MODE == 1) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(h01) : "v"(w01), "v"(xx));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(h23) : "v"(w23), "v"(xx));
                } else {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(h01) : "v"(w01), "v"(xx));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(h23) : "v"(w23), "v"(xx));
                }
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r0) : "v"(w[u][0]), "v"(x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r1) : "v"(w[u][1]), "v"(x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r2) : "v"(w[u][2]), "v"(x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r3) : "v"(w[u][3]), "v"(x));
            }
        }
        const bool b = __float_as_uint(h01.x) != __float_as_uint(r0) || __float_as_uint(h01.y) != __float_as_uint(r1) ||
                       __float_as_uint(h23.x) != __float_as_uint(r2) || __float_as_uint(h23.y) != __float_as_uint(r3);
        const unsigned long long m = __ballot(b);
        if (m && lane == 0) {
            ++nbad;
            atomicAdd(bad_lanes, (unsigned)__popcll(m));
        }
    }
    if (nbad && lane == 0) atomicAdd(bad, nbad);
}

// ---- second experiment: the same victim loop on waves 0..3 of an 8-wave workgroup while waves 4..7 -- which share the SIMDs
// of waves 0..3 -- run a PARTNER instruction stream until the victims are done: nothing / MFMAs into VGPR accumulators /
// MFMAs into AGPR accumulators / ds_read_b128 / the epilogue's packed and mixed-precision VALU ops / global loads.
using half8 = _Float16 __attribute__((ext_vector_type(8)));
using f4 = float __attribute__((ext_vector_type(4)));

template <int MODE, int PARTNER>
__global__ __launch_bounds__(512) void k2(const float* __restrict__ wt, const float* __restrict__ xin, unsigned* bad,
                                          unsigned* bad_lanes, int reps, float* sink) {
    __shared__ float xs[4][64];
    __shared__ __attribute__((aligned(16))) float pad[4][64 * 4];
    __shared__ volatile int done;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (wave >= 4) {   // partner waves
        if (PARTNER == 0) return;
        const int w4 = wave - 4;
        half8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
        f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        f2 p0 = {1.f, 2.f}, p1 = {0.5f, 0.25f};
        float g = 0.f;
        for (int it = 0; it < (1 << 22); ++it) {   // bounded: exits on `done` or after 4 M iterations
            if (PARTNER == 1) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc1, 0, 0, 0);
            } else if (PARTNER == 2) {
                asm volatile("v_mfma_f32_16x16x32_f16 a[0:3], %0, %1, a[0:3]\n\tv_mfma_f32_16x16x32_f16 a[4:7], %1, %0, a[4:7]"
                             :: "v"(a), "v"(b) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
            } else if (PARTNER == 3) {
                f4 t = *(volatile f4*)&pad[w4][lane * 4];
                acc0 += t;
            } else if (PARTNER == 4) {
                asm volatile("v_pk_add_f32 %0, %0, %1\n\tv_pk_fma_f32 %1, %1, %0, %1" : "+v"(p0), "+v"(p1));
                unsigned lo;
                asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                             : "=&v"(lo) : "v"(__float_as_uint(p0.x)), "v"(p0.y), "v"(p1.x));
                g += __uint_as_float(lo & 0x3f800000u);
            } else if (PARTNER == 5) {
                g += wt[((it * 64 + lane) * 4) & (64 * 256 - 1)];
            }
            if ((it & 63) == 63 && done) break;
        }
        if (PARTNER == 2) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(g) :: "a0");
        if (sink && g + acc0[0] + acc1[0] + p0.x + p1.y == 12345.678f) sink[0] = g;   // keep the work alive
        return;
    }
    unsigned nbad = 0;
    for (int rep = 0; rep < reps; ++rep) {
        xs[wave][lane] = xin[((blockIdx.x * 4 + wave + rep) & 1023) * 64 + lane];
        f2 h01 = {0.f, 0.f}, h23 = {0.f, 0.f};
        float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
#pragma unroll 1
        for (int i0 = 0; i0 < 64; i0 += 8) {
            float w[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) w[u][j] = wt[(size_t)(i0 + u) * 256 + lane + 64 * j];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float x = xs[wave][i0 + u];
                f2 xx = MODE == 0 ? f2{x, 0.f} : f2{0.f, x};
                asm volatile("" : "+v"(xx));
                const f2 w01 = {w[u][0], w[u][1]}, w23 = {w[u][2], w[u][3]};
                if (MODE == 0) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(h01) : "v"(w01), "v"(xx));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(h23) : "v"(w23), "v"(xx));
                } else {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(h01) : "v"(w01), "v"(xx));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(h23) : "v"(w23), "v"(xx));
                }
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r0) : "v"(w[u][0]), "v"(x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r1) : "v"(w[u][1]), "v"(x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r2) : "v"(w[u][2]), "v"(x));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r3) : "v"(w[u][3]), "v"(x));
            }
        }
        const bool b = __float_as_uint(h01.x) != __float_as_uint(r0) || __float_as_uint(h01.y) != __float_as_uint(r1) ||
                       __float_as_uint(h23.x) != __float_as_uint(r2) || __float_as_uint(h23.y) != __float_as_uint(r3);
        const unsigned long long m = __ballot(b);
        if (m && lane == 0) {
            ++nbad;
            atomicAdd(bad_lanes, (unsigned)__popcll(m));
        }
    }
    if (nbad && lane == 0) atomicAdd(bad, nbad);
    if (lane == 0) atomicAdd((int*)&done, 1);
}

template <int MODE, int PARTNER>
static void run2(const float* dw, const float* dx, unsigned* dbad, float* sink, const char* mname, const char* pname) {
    const int reps = 400;
    for (int launch = 0; launch < 2; ++launch) {
        hipMemset(dbad, 0, 8);
        hipLaunchKernelGGL((k2<MODE, PARTNER>), dim3(256), dim3(512), 0, 0, dw, dx, dbad, dbad + 1, reps, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
        unsigned h[2];
        hipMemcpy(h, dbad, 8, hipMemcpyDeviceToHost);
        printf("victim %-19s partner %-28s launch %d: %u of %d wave-results differ (%u lanes)\n", mname, pname, launch, h[0],
               256 * 4 * reps, h[1]);
    }
}

int main() {
    std::vector<float> w(64 * 256), x(1024 * 64);
    srand(1);
    for (auto& v : w) v = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
    for (auto& v : x) v = rand() / (float)RAND_MAX;
    float *dw, *dx;
    unsigned* dbad;
    hipMalloc(&dw, w.size() * 4);
    hipMalloc(&dx, x.size() * 4);
    hipMalloc(&dbad, 8);
    hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    const char* names[3] = {"op_sel_hi:[1,0,1] (x in the low dword)", "op_sel:[0,1,0]   (x in the HIGH dword)", "no operand select ({x, x})"};
    const int reps = 400;
    for (int grid : {256, 512, 1024}) {
        for (int mode = 0; mode < 3; ++mode) {
            for (int launch = 0; launch < 3; ++launch) {
                hipMemset(dbad, 0, 8);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, dw, dx, dbad, dbad + 1, reps);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, dw, dx, dbad, dbad + 1, reps);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, dw, dx, dbad, dbad + 1, reps);
                if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                unsigned h[2];
                hipMemcpy(h, dbad, 8, hipMemcpyDeviceToHost);
                printf("grid %4d (%s per SIMD)  %-40s launch %d: %u of %d wave-results differ from the scalar FMAs (%u lanes)\n", grid,
                       grid <= 256 ? "1 wave " : grid <= 512 ? "2 waves" : "4 waves", names[mode], launch, h[0], grid * 4 * reps, h[1]);
            }
        }
    }
    float* sink;
    hipMalloc(&sink, 4);
#define RUN(P, PN) run2<0, P>(dw, dx, dbad, sink, "op_sel_hi:[1,0,1]", PN); run2<1, P>(dw, dx, dbad, sink, "op_sel:[0,1,0]", PN);
    RUN(0, "none")
    RUN(1, "MFMA, VGPR accumulators")
    RUN(2, "MFMA, AGPR accumulators")
    RUN(3, "ds_read_b128")
    RUN(4, "v_pk_add/fma_f32 + v_fma_mix")
    RUN(5, "global_load_dword")
    return 0;
}
