// What does a Winograd-trunk conv step -- three dependent v_mfma_f32_16x16x32_f16 into ONE accumulator plus the two
// ds_read_b128 of a later step -- cost per MFMA, and what changes it?  (k_trunk_w6 runs 26-27 cycles per MFMA, 19 with its LDS
// reads ablated, whatever the look-ahead: round 4, profiles/r04_w6_experiments.log.)  One workgroup per CU, W waves per SIMD,
// every wave runs `steps` steps over a ring of NACC accumulators; cycles by s_memtime.
//   variants: reads per step 0 / 1 / 2; accumulators in AGPRs (builtin) or VGPRs (asm, in place); chain order "3 in a row"
//   or two accumulators interleaved; ds_read_b128 or 2 x ds_read_b64 per operand; 1 or 2 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/probe_w6_step tools/probes/probe_w6_step.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

using half8 = _Float16 __attribute__((ext_vector_type(8)));
using half4 = _Float16 __attribute__((ext_vector_type(4)));
using f4 = float __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long clk() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define SB __builtin_amdgcn_sched_barrier(0)

// READS: ds_read_b128 per step (0, 1, 2); ASM: 0 = builtin MFMA (hipcc keeps the accumulators in VGPRs here), 1 = in-place asm
// MFMAs on VGPR accumulators, 2 = in-place asm MFMAs on AGPR accumulators; ILV: two accumulators interleaved;
// B64: each operand as two ds_read_b64; NACC accumulators in the ring
template <int READS, int ASM, bool ILV, bool B64, int NACC, int PAT = 0, int WOFF = 1280, int SSTR = 8192, int SMOD = 3, int LOOFF = 1024, int RB = 1>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int steps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 147456 / 16; i += blockDim.x) ((uint4*)lds)[i] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    __syncthreads();
    half8 wh, wl;
    for (int i = 0; i < 8; ++i) { wh[i] = (_Float16)(0.01f * (lane & 7)); wl[i] = (_Float16)(0.001f * i); }
    f4 acc[NACC];
    f4 acc2[RB == 2 ? NACC : 1];   // RB = 2: a second row block per step (a 32-channel wave: each operand pair feeds six MFMAs)
    half8 wh2 = wh, wl2 = wl;
    if (RB == 2) for (int i = 0; i < 8; ++i) { wh2[i] = (_Float16)(0.02f * (lane & 3)); wl2[i] = (_Float16)(0.003f * i); }
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        acc[i] = f4{0.f, 0.f, 0.f, 0.f};
        if (ASM == 2) asm volatile("" : "+a"(acc[i]));
        else asm volatile("" : "+v"(acc[i]));
        if (RB == 2) {
            acc2[i] = f4{0.f, 0.f, 0.f, 0.f};
            if (ASM == 2) asm volatile("" : "+a"(acc2[i]));
            else asm volatile("" : "+v"(acc2[i]));
        }
    }
    asm volatile("s_nop 4");
    // PAT 0: lane-linear 1 KB rows.  PAT 1: k_trunk_w6's round-3 image -- tile (position, row) pr = lane & 15 at pr * 3 KB, the
    // 16-byte slot of the lane's k-group XOR-swizzled with 2 * (pr & 7) (conflict-free by the lane-group model).  PAT 2: the
    // round-4 image [k-group g4][...][column c][16 B] with the g4 planes 12 KB apart, read with a row-tap shift of one column
    // (column 15's source lives in the next lane group's block, 48 KB away).
    const int c_ = lane & 15, g4_ = lane >> 4;
    const char* base = PAT == 0 ? lds + (wave & 3) * 32768 + lane * 16
                     : PAT == 1 ? lds + c_ * 3072 + ((g4_ << 4) ^ ((c_ & 7) << 5)) + (wave & 3) * 256
                     : PAT == 2 ? lds + g4_ * 12288 + (c_ == 15 ? 49152 : (c_ + 1) * 16) + (wave & 3) * 256
                     : PAT == 3 ? lds + g4_ * 12288 + c_ * 16 + (wave & 3) * 256              // planes 12 KB apart, no shift
                     : PAT == 4 ? lds + (wave & 3) * 32768 + lane * 16 + 16                    // linear, shifted by one slot
                     : PAT == 5 ? lds + g4_ * 1024 + c_ * 16 + (wave & 3) * 256                // planes 1 KB apart
                     : PAT == 6 ? lds + g4_ * 4096 + c_ * 16 + (wave & 3) * 256                // planes 4 KB apart
                     : PAT == 7 ? lds + (wave & 3) * 32768 + g4_ * 256 + (c_ == 15 ? 20480 : c_ * 16)   // linear, lane 15 far away
                     : PAT == 8 ? lds + (wave & 3) * 32768 + g4_ * 256 + (c_ + 1) * 16 - (c_ == 15 ? 256 : 0)  // rotate within the plane
                     : PAT == 9 ? lds + g4_ * 12288 + c_ * 16 + (wave & 3) * 3072              // planes 12 KB apart, waves 3 KB apart
                                : lds + g4_ * PAT + c_ * 16 + (wave & 3) * WOFF;            // PAT >= 100: k-group planes PAT bytes apart
    half8 xh[3], xl[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { xh[i] = *(const half8*)(base + i * 2048); xl[i] = *(const half8*)(base + i * 2048 + 1024); }
    auto mf = [&](half8 a, half8 b, f4& c) {
        if (ASM == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
        else if (ASM == 2) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
        else c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    };
    auto rd = [&](half8& dst, const char* p) {
        if (B64) {
            const half4 lo = *(const half4*)p, hi = *(const half4*)(p + 8);
            dst = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        } else {
            dst = *(const half8*)p;
        }
    };
    const unsigned long long t0 = clk();
#pragma unroll 1
    for (int s0 = 0; s0 < steps; s0 += NACC) {
#pragma unroll
        for (int i = 0; i < NACC; i += (ILV ? 2 : 1)) {
            const int sl = i % 3, nx = (i + 2) % 3;
            // (all step offsets are compile-time after unrolling: a run-time modulo here costs more than the reads)
            const char* p = base + (PAT == 0 || PAT == 4 || PAT == 7 || PAT == 8 ? (i & 7) * 2048 : PAT == 1 ? (i % 3) * 1024 : PAT == 9 ? (i % 5) * 512 : PAT >= 100 ? (i % SMOD) * SSTR : (i % 5) * 2048);
            if (!ILV) {
                SB; mf(wh, xl[sl], acc[i]); SB;
                if (READS >= 1) rd(xh[nx], p);
                SB; mf(wh, xh[sl], acc[i]); SB;
                if (READS >= 2) rd(xl[nx], PAT == 1 ? (const char*)((size_t)p ^ 128) : p + (PAT >= 100 ? LOOFF : 1024));
                SB; mf(wl, xh[sl], acc[i]); SB;
                if (RB == 2) {
                    SB; mf(wh2, xl[sl], acc2[i]); SB;
                    SB; mf(wh2, xh[sl], acc2[i]); SB;
                    SB; mf(wl2, xh[sl], acc2[i]); SB;
                }
            } else {
                const int sl2 = (i + 1) % 3, nx2 = i % 3;
                SB; mf(wh, xl[sl], acc[i]); SB;
                if (READS >= 1) rd(xh[nx], p);
                SB; mf(wh, xl[sl2], acc[i + 1]); SB;
                if (READS >= 2) rd(xl[nx], p + 1024);
                SB; mf(wh, xh[sl], acc[i]); SB;
                SB; mf(wh, xh[sl2], acc[i + 1]); SB;
                if (READS >= 1) rd(xh[nx2], p + 2048);
                SB; mf(wl, xh[sl], acc[i]); SB;
                if (READS >= 2) rd(xl[nx2], p + 3072);
                SB; mf(wl, xh[sl2], acc[i + 1]); SB;
            }
        }
    }
    const unsigned long long t1 = clk();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][3] + (RB == 2 ? acc2[i][1] : 0.f);
    if (r == 12345.678f) sink[0] = r;
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int READS, int ASM, bool ILV, bool B64, int NACC, int PAT = 0, int WOFF = 1280, int SSTR = 8192, int SMOD = 3, int LOOFF = 1024, int RB = 1>
static void run(const char* name, int waves_per_simd, unsigned long long* d, float* sink) {
    const int steps = 36 * 60, threads = 256 * waves_per_simd;
    hipFuncSetAttribute((const void*)k<READS, ASM, ILV, B64, NACC, PAT, WOFF, SSTR, SMOD, LOOFF, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(d, 0, 256 * 8 * 8);
        hipLaunchKernelGGL((k<READS, ASM, ILV, B64, NACC, PAT, WOFF, SSTR, SMOD, LOOFF, RB>), dim3(256), dim3(threads), 147456, 0, d, sink, steps);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    }
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0, mx = 0; int n = 0;
    for (auto v : h) if (v) { sum += (double)v; mx = mx > (double)v ? mx : (double)v; ++n; }
    const double mfmas_per_wave = 3.0 * RB * steps;
    printf("%-66s %d wave/SIMD: %6.2f cycles per MFMA per wave (mean), %6.2f per MFMA of the SIMD (slowest wave)\n", name, waves_per_simd,
           sum / n / mfmas_per_wave, mx / (mfmas_per_wave * waves_per_simd));
}

int main() {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 8 * 8); hipMalloc(&sink, 4);
    // the 8x8 trunk's question (round 4): today's wave (16 channels: 16 accumulators in VGPRs, two waves per SIMD, two
    // reads per three MFMAs) against a 32-channel wave (32 accumulators, one wave per SIMD, two reads per SIX MFMAs)
    run<2, 1, false, false, 16>("16-channel wave: asm VGPR acc, 16 acc, 2 reads / 3 MFMAs", 2, d, sink);
    run<2, 0, false, false, 16>("16-channel wave: builtin, 16 acc, 2 reads / 3 MFMAs", 2, d, sink);
    run<2, 0, false, false, 16, 0, 1280, 8192, 3, 1024, 2>("32-channel wave: builtin, 2 x 16 acc, 2 reads / 6 MFMAs", 1, d, sink);
    run<2, 2, false, false, 16, 0, 1280, 8192, 3, 1024, 2>("32-channel wave: asm AGPR acc, 2 x 16 acc, 2 reads / 6 MFMAs", 1, d, sink);
    run<2, 1, false, false, 16, 0, 1280, 8192, 3, 1024, 2>("32-channel wave: asm VGPR acc, 2 x 16 acc, 2 reads / 6 MFMAs", 1, d, sink);
    run<0, 0, false, false, 16, 0, 1280, 8192, 3, 1024, 2>("32-channel wave: builtin, 2 x 16 acc, no reads", 1, d, sink);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0, false, false, 36>("builtin (VGPR), 36 acc, no reads", w, d, sink);
        run<2, 0, false, false, 36>("builtin (VGPR), 36 acc, 2 ds_read_b128 / step, lane-linear", w, d, sink);
        run<0, 2, false, false, 36>("asm, AGPR accumulators, 36 acc, no reads", w, d, sink);
        run<2, 2, false, false, 36>("asm, AGPR acc, 2 ds_read_b128 / step, lane-linear image", w, d, sink);
        run<2, 2, true, false, 36>("asm, AGPR acc, 2 reads / step, two accumulators interleaved", w, d, sink);
        run<2, 2, false, true, 36>("asm, AGPR acc, operands as 2 x ds_read_b64", w, d, sink);
        run<2, 2, false, false, 36, 1>("  image: k_trunk_w6 round 3 (column x 3 KB, slot XOR 2 * (column & 7))", w, d, sink);
        run<2, 2, false, false, 36, 2>("  image: k-group planes 12 KB apart, columns shifted by one, column 15 far", w, d, sink);
        run<2, 2, false, false, 36, 3>("  image: k-group planes 12 KB apart", w, d, sink);
        run<2, 2, false, false, 36, 4>("  image: lane-linear, shifted by one 16-byte slot", w, d, sink);
        run<2, 2, false, false, 36, 5>("  image: k-group planes 1 KB apart", w, d, sink);
        run<2, 2, false, false, 36, 7>("  image: lane-linear, column 15 of each k-group far away", w, d, sink);
        run<2, 2, false, false, 36, 320>("  image: k-group planes 320 B apart", w, d, sink);
        run<2, 2, false, false, 36, 1088>("  image: k-group planes 1088 B apart", w, d, sink);
        run<0, 1, false, false, 18>("asm, VGPR accumulators, 18 acc, no reads", w, d, sink);
        run<2, 1, false, false, 18>("asm, VGPR accumulators, 18 acc, 2 ds_read_b128 / step, lane-linear", w, d, sink);
        run<2, 1, false, false, 18, 1>("asm, VGPR accumulators, 18 acc, round-3 image", w, d, sink);
    }
    return 0;
}
