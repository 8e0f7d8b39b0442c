// Round 5: the siblings of the gfx950 `v_pk_fma_f32 ... op_sel:[0,1,0]` defect that round 4 never probed (VERDICT r4 item 7,
// ADVICE r4).  Round 4 showed: a packed-fp32 FMA that takes the HIGH dword of a source pair for its LOW lane returns wrong
// results whenever ANOTHER wave of the same SIMD is issuing MFMAs (tools/probes/probe_pk_opsel.hip, 24-29 % of wave-results),
// and never otherwise.  tools/check_mfma_hazards.py rule (3) forbids what was shown to fail; this probe asks which OTHER
// operand-select forms fail beside the same MFMA partner -- the other source positions of v_pk_fma_f32, the low-to-high
// routes (op_sel_hi), v_pk_mul / v_pk_add, v_pk_mov_b32, the packed-f16 VALU forms, and the forms the shipped epilogue
// uses (net_epilogue.h: v_fma_mixlo/mixhi_f16 with op_sel, v_pk_add/fma_f32 with neg_lo / neg_hi on plain pairs).
//
// Every victim computes the same arithmetic twice per step -- in the form under test and in a reference form that uses no
// operand select (scalar ops, or the packed op on an explicitly built {x, x} pair) -- into two accumulator sets that must end
// bit-identical.  Waves 0..3 of an 8-wave workgroup are the victims; waves 4..7 (same SIMDs) run the partner: nothing, or
// v_mfma_f32_16x16x32_f16 back to back until the victims are done (bounded).  A form is UNRELIABLE when it is clean alone and
// differs beside the partner; a form that already differs alone means this file's reading of its semantics is wrong
// (reported as such, inconclusive).  ONE run: build/probe_pk_opsel2 > profiles/r05_probe_pk_opsel.txt
// Build: hipcc --offload-arch=gfx950 -O2 -o build/probe_pk_opsel2 tools/probes/probe_pk_opsel2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f2 = float __attribute__((ext_vector_type(2)));
using half8 = _Float16 __attribute__((ext_vector_type(8)));
using f4 = float __attribute__((ext_vector_type(4)));
using h2 = _Float16 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float s_fma(float a, float b, float c) { float d; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float s_mul(float a, float b) { float d; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float s_add(float a, float b) { float d; asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float s_sub(float a, float b) { float d; asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t pack16(float a, float b) { return __builtin_bit_cast(uint32_t, h2{(_Float16)a, (_Float16)b}); }

// state: t* = form under test, r* = reference; four fp32 lanes or two packed-f16 words each
struct St { f2 t01, t23; float r0, r1, r2, r3; uint32_t th0, th1, rh0, rh1; };

template <int V>
__device__ __forceinline__ void step(St& s, float w0, float w1, float w2, float w3, float x) {
    const f2 w01 = {w0, w1}, w23 = {w2, w3};
    if constexpr (V <= 3) {   // ---- v_pk_fma_f32 with one operand-select bit
        f2 xx = (V == 3) ? f2{x, 0.f} : f2{0.f, x};
        asm volatile("" : "+v"(xx));
        if (V == 0) {          // CONTROL (known bad, round 4): src1 high -> low lane
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(s.t01) : "v"(w01), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(s.t23) : "v"(w23), "v"(xx));
        } else if (V == 1) {   // src0 high -> low lane
            asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,0,0]" : "+v"(s.t01) : "v"(w01), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,0,0]" : "+v"(s.t23) : "v"(w23), "v"(xx));
        } else if (V == 2) {   // src2 high -> low lane: d = w * {x, x} + {c.hi, c.hi} with c = {junk, previous low result}
            f2 x2 = {x, x};
            asm volatile("" : "+v"(x2));
            f2 c01 = {123.f, s.t01.x}, c23 = {456.f, s.t23.x};
            asm volatile("" : "+v"(c01), "+v"(c23));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(s.t01) : "v"(w01), "v"(x2), "v"(c01));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(s.t23) : "v"(w23), "v"(x2), "v"(c23));
            const float p0 = s.r0, p2 = s.r2;
            s.r0 = s_fma(w0, x, p0); s.r1 = s_fma(w1, x, p0); s.r2 = s_fma(w2, x, p2); s.r3 = s_fma(w3, x, p2);
            return;
        } else {               // src0 low -> HIGH lane (op_sel_hi cleared): the mirror route
            asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel_hi:[0,1,1]" : "+v"(s.t01) : "v"(w01), "v"(xx));
            asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel_hi:[0,1,1]" : "+v"(s.t23) : "v"(w23), "v"(xx));
        }
        s.r0 = s_fma(w0, x, s.r0); s.r1 = s_fma(w1, x, s.r1); s.r2 = s_fma(w2, x, s.r2); s.r3 = s_fma(w3, x, s.r3);
    } else if constexpr (V == 4) {   // ---- v_pk_mul_f32 op_sel:[0,1] (src1 high -> low), then a plain packed add
        f2 xx = {0.f, x};
        asm volatile("" : "+v"(xx));
        f2 m01, m23;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(m01) : "v"(w01), "v"(xx));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(m23) : "v"(w23), "v"(xx));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(s.t01) : "v"(m01));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(s.t23) : "v"(m23));
        s.r0 = s_add(s.r0, s_mul(w0, x)); s.r1 = s_add(s.r1, s_mul(w1, x)); s.r2 = s_add(s.r2, s_mul(w2, x)); s.r3 = s_add(s.r3, s_mul(w3, x));
    } else if constexpr (V == 5) {   // ---- v_pk_add_f32 op_sel:[0,1] (src1 high -> low)
        const float y0 = s_mul(w0, x), y2 = s_mul(w2, x);
        f2 yy0 = {0.f, y0}, yy2 = {0.f, y2};
        asm volatile("" : "+v"(yy0), "+v"(yy2));
        asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(s.t01) : "v"(yy0));
        asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(s.t23) : "v"(yy2));
        s.r0 = s_add(s.r0, y0); s.r1 = s_add(s.r1, y0); s.r2 = s_add(s.r2, y2); s.r3 = s_add(s.r3, y2);
    } else if constexpr (V == 6) {   // ---- v_pk_mov_b32 op_sel:[1,0]: D.lo = src0's HIGH dword, D.hi = src1's low dword (this file's reading)
        const float y0 = s_fma(w0, x, s.r0), y1 = s_fma(w1, x, s.r1);
        f2 a = {-1.f, y0}, b = {y1, -2.f}, d;
        asm volatile("" : "+v"(a), "+v"(b));
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        s.t01 = d;
        s.r0 = y0; s.r1 = y1;
        s.t23 = f2{0.f, 0.f}; s.r2 = s.r3 = 0.f;
    } else if constexpr (V >= 7 && V <= 9) {   // ---- packed f16 VALU, src1 high half -> low lane
        const uint32_t a01 = pack16(w0, w1), a23 = pack16(w2, w3), bj = pack16(7.f, x), bb = pack16(x, x);
        uint32_t bjv = bj, bbv = bb;
        asm volatile("" : "+v"(bjv), "+v"(bbv));
        if (V == 7) {
            asm volatile("v_pk_fma_f16 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(s.th0) : "v"(a01), "v"(bjv));
            asm volatile("v_pk_fma_f16 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(s.th1) : "v"(a23), "v"(bjv));
            asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(s.rh0) : "v"(a01), "v"(bbv));
            asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(s.rh1) : "v"(a23), "v"(bbv));
        } else if (V == 8) {
            uint32_t m0, m1, n0, n1;
            asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(m0) : "v"(a01), "v"(bjv));
            asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(m1) : "v"(a23), "v"(bjv));
            asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(n0) : "v"(a01), "v"(bbv));
            asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(n1) : "v"(a23), "v"(bbv));
            asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(s.th0) : "v"(m0));
            asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(s.th1) : "v"(m1));
            asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(s.rh0) : "v"(n0));
            asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(s.rh1) : "v"(n1));
        } else {
            const uint32_t small = pack16(5.f, 0.0625f * x), small2 = pack16(0.0625f * x, 0.0625f * x);
            uint32_t sj = small, sb = small2;
            asm volatile("" : "+v"(sj), "+v"(sb));
            asm volatile("v_pk_add_f16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]" : "+v"(s.th0) : "v"(sj));
            asm volatile("v_pk_add_f16 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,1]" : "+v"(s.th1) : "v"(sj));
            asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(s.rh0) : "v"(sb));
            asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(s.rh1) : "v"(sb));
        }
    } else if constexpr (V == 10) {   // ---- the shipped epilogue's low-part split: v_fma_mixlo / mixhi_f16 with op_sel (net_epilogue.h wresid)
        const float a = s_fma(w0, x, s.r0), b = s_fma(w1, x, s.r1);
        s.r0 = a; s.r1 = b;
        const uint32_t hi = pack16(a, b);
        uint32_t lo;
        asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a));
        asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(b));
        // reference: convert the halves back (plain cvt), subtract in fp32 (exact), round once
        float ha, hb;
        asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(ha) : "v"(hi));
        const uint32_t hsh = hi >> 16;
        asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(hb) : "v"(hsh));
        const uint32_t ref = pack16(s_sub(a, ha), s_sub(b, hb));
        s.th0 ^= lo * 2654435761u + (s.th0 >> 3);
        s.rh0 ^= ref * 2654435761u + (s.rh0 >> 3);
    } else {   // V == 11 ---- the shipped epilogue's packed fp32 with neg_lo / neg_hi on PLAIN pairs (pk_sub, pk_fma_nc, pk_fma_na)
        f2 xx = {x, x};
        asm volatile("" : "+v"(xx));
        f2 d0, d1, d2;
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d0) : "v"(s.t01), "v"(w01));                    // t01 - w01
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d1) : "v"(w23), "v"(xx), "v"(d0));      // w23 * x - d0
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d2) : "v"(w01), "v"(xx), "v"(d1));      // d1 - w01 * x
        s.t01 = d2;
        const float e0 = s_sub(s.r0, w0), e1 = s_sub(s.r1, w1);
        const float g0 = s_fma(w2, x, -e0), g1 = s_fma(w3, x, -e1);
        s.r0 = s_fma(-w0, x, g0); s.r1 = s_fma(-w1, x, g1);
    }
}

template <int V>
__device__ __forceinline__ bool differs(const St& s) {
    if (V >= 7 && V <= 10) return s.th0 != s.rh0 || s.th1 != s.rh1;
    bool b = __float_as_uint(s.t01.x) != __float_as_uint(s.r0) || __float_as_uint(s.t01.y) != __float_as_uint(s.r1);
    if (V <= 5) b = b || __float_as_uint(s.t23.x) != __float_as_uint(s.r2) || __float_as_uint(s.t23.y) != __float_as_uint(s.r3);
    return b;
}

template <int V, int PARTNER>
__global__ __launch_bounds__(512) void k(const float* __restrict__ wt, const float* __restrict__ xin, unsigned* bad, int reps, float* sink) {
    __shared__ float xs[4][64];
    __shared__ volatile int done;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    if (wave >= 4) {   // partner waves: MFMAs into VGPR accumulators until the victims are done (bounded: 4 M iterations)
        if (PARTNER == 0) return;
        half8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
        f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        for (int it = 0; it < (1 << 22); ++it) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc1, 0, 0, 0);
            if ((it & 63) == 63 && done >= 4) break;
        }
        if (sink && acc0[0] + acc1[0] == 12345.678f) sink[0] = acc0[1];
        return;
    }
    unsigned nbad = 0;
    for (int rep = 0; rep < reps; ++rep) {
        xs[wave][lane] = xin[((blockIdx.x * 4 + wave + rep) & 1023) * 64 + lane];
        St s;
        s.t01 = s.t23 = f2{0.f, 0.f};
        s.r0 = s.r1 = s.r2 = s.r3 = 0.f;
        s.th0 = s.th1 = s.rh0 = s.rh1 = 0u;
#pragma unroll 1
        for (int i0 = 0; i0 < 64; i0 += 8) {
            float w[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) w[u][j] = wt[(size_t)(i0 + u) * 256 + lane + 64 * j];
#pragma unroll
            for (int u = 0; u < 8; ++u) step<V>(s, w[u][0], w[u][1], w[u][2], w[u][3], xs[wave][i0 + u]);
        }
        if (__ballot(differs<V>(s)) && lane == 0) ++nbad;
    }
    if (nbad && lane == 0) atomicAdd(bad, nbad);
    if (lane == 0) atomicAdd((int*)&done, 1);
}

static const char* NAMES[12] = {
    "v_pk_fma_f32 op_sel:[0,1,0]      (src1 high->low: CONTROL, known bad)",
    "v_pk_fma_f32 op_sel:[1,0,0]      (src0 high->low)",
    "v_pk_fma_f32 op_sel:[0,0,1]      (src2 high->low)",
    "v_pk_fma_f32 op_sel_hi:[0,1,1]   (src0 low->HIGH)",
    "v_pk_mul_f32 op_sel:[0,1]        (src1 high->low)",
    "v_pk_add_f32 op_sel:[0,1]        (src1 high->low)",
    "v_pk_mov_b32 op_sel:[1,0]        (src0 high->low)",
    "v_pk_fma_f16 op_sel:[0,1,0]      (src1 high half->low)",
    "v_pk_mul_f16 op_sel:[0,1]        (src1 high half->low)",
    "v_pk_add_f16 op_sel:[0,1]        (src1 high half->low)",
    "v_fma_mixlo/mixhi_f16 op_sel     (SHIPPED: net_epilogue.h wresid)",
    "v_pk_add/fma_f32 neg_lo/neg_hi   (SHIPPED: plain pairs, pk_sub / pk_fma_nc / pk_fma_na)",
};

template <int V>
static void run(const float* dw, const float* dx, unsigned* dbad, float* sink) {
    const int reps = 400;
    unsigned res[3];
    for (int i = 0; i < 3; ++i) {   // launch 0: no partner; launches 1, 2: MFMA partner
        hipMemset(dbad, 0, 4);
        if (i == 0) hipLaunchKernelGGL((k<V, 0>), dim3(256), dim3(512), 0, 0, dw, dx, dbad, reps, sink);
        else hipLaunchKernelGGL((k<V, 1>), dim3(256), dim3(512), 0, 0, dw, dx, dbad, reps, sink);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); exit(1); }
        hipMemcpy(&res[i], dbad, 4, hipMemcpyDeviceToHost);
    }
    const int total = 256 * 4 * reps;
    const char* verdict = res[0] ? "INCONCLUSIVE (differs alone: this file's reading of the form is wrong)"
                                 : ((res[1] || res[2]) ? "UNRELIABLE beside an MFMA-issuing wave" : "clean");
    printf("%-88s alone %7u | beside MFMA %7u, %7u of %d wave-results differ -> %s\n", NAMES[V], res[0], res[1], res[2], total, verdict);
    fflush(stdout);
}

int main() {
    std::vector<float> w(64 * 256), x(1024 * 64);
    srand(1);
    for (auto& v : w) v = (rand() / (float)RAND_MAX - 0.5f) * 0.25f;
    for (auto& v : x) v = rand() / (float)RAND_MAX;
    float *dw, *dx, *sink;
    unsigned* dbad;
    hipMalloc(&dw, w.size() * 4); hipMalloc(&dx, x.size() * 4); hipMalloc(&dbad, 4); hipMalloc(&sink, 4);
    hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    printf("victims = waves 0..3 of 256 eight-wave workgroups, 400 repetitions each (409600 wave-results per launch); partner waves 4..7 share their SIMDs\n");
    run<0>(dw, dx, dbad, sink); run<1>(dw, dx, dbad, sink); run<2>(dw, dx, dbad, sink); run<3>(dw, dx, dbad, sink);
    run<4>(dw, dx, dbad, sink); run<5>(dw, dx, dbad, sink); run<6>(dw, dx, dbad, sink); run<7>(dw, dx, dbad, sink);
    run<8>(dw, dx, dbad, sink); run<9>(dw, dx, dbad, sink); run<10>(dw, dx, dbad, sink); run<11>(dw, dx, dbad, sink);
    return 0;
}
