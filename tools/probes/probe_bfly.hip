// probe_bfly.hip -- wave_bfly.h against the __shfl_xor butterflies it replaces: bit-exactness on random data (floats incl.
// negative zeros / infinities / denormals, doubles, ints) for every lane, and cycles per reduction for a lone wave.
// build: hipcc --offload-arch=gfx950 -O2 -I othello_reinforcement_learning_test_amd/csrc -o build/probe_bfly tools/probes/probe_bfly.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "wave_bfly.h"
using namespace oth;

__device__ __forceinline__ float s_sum(float v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ float s_max(float v) { for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ int s_isum(int v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ int s_imax(int v) { for (int o = 32; o; o >>= 1) v = max(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ double s_dmax(double v) { for (int o = 32; o; o >>= 1) v = fmax(v, __shfl_xor(v, o)); return v; }

__global__ void k_check(const float* f, const double* d, const int* i, int n_rows, unsigned long long* bad) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const float v = f[row * 64 + lane];
    const double w = d[row * 64 + lane];
    const int u = i[row * 64 + lane];
    unsigned long long b = 0;
    b += __float_as_uint(s_sum(v)) != __float_as_uint(bfly_sum_f32(v));
    const float m0 = s_max(v), m1 = bfly_max_f32(v);
    b += !(m0 == m1 || (m0 != m0 && m1 != m1));          // (max of +0 / -0: equal as numbers; the callers never look at the sign)
    b += s_isum(u) != bfly_sum_i32(u);
    b += s_imax(u) != bfly_max_i32(u);
    const double e0 = s_dmax(w), e1 = bfly_max_f64(w);
    b += !(e0 == e1);
    if (b) atomicAdd(bad, b);
}
template <int MODE> __global__ void k_time(const float* f, float* out, long long* cyc, int reps) {
    float v = f[threadIdx.x];
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0) v = s_sum(v) * 0.5f + 1.0f;
        else if (MODE == 1) v = bfly_sum_f32(v) * 0.5f + 1.0f;
        else if (MODE == 2) v = (float)s_dmax((double)v) * 0.5f + 1.0f;
        else v = (float)bfly_max_f64((double)v) * 0.5f + 1.0f;
    }
    const long long t1 = clock64();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    const int rows = 1 << 16, n = rows * 64;
    std::vector<float> f(n); std::vector<double> d(n); std::vector<int> iv(n);
    srand(7);
    for (int k = 0; k < n; ++k) {
        unsigned r = ((unsigned)rand() << 16) ^ (unsigned)rand();
        float x; memcpy(&x, &r, 4);
        if (x != x) x = (float)(r % 1000) - 500.f;         // no NaNs in the data (the heads never reduce NaNs)
        if (k % 97 == 0) x = -0.0f; if (k % 101 == 0) x = 0.0f; if (k % 9973 == 0) x = -INFINITY;
        f[k] = (k & 1) ? x : (float)((int)(r % 20001) - 10000) * 1e-3f;
        d[k] = (double)((int)(r % 2000001) - 1000000) * 1e-7 + ((k % 13 == 0) ? 0.0 : 1e-12 * (r & 255));
        iv[k] = (int)(r % 4001) - 2000;
    }
    float *df, *dout; double* dd; int* di; unsigned long long* dbad; long long* dcyc;
    hipMalloc(&df, n * 4); hipMalloc(&dd, n * 8); hipMalloc(&di, n * 4); hipMalloc(&dbad, 8); hipMalloc(&dout, 256); hipMalloc(&dcyc, 8);
    hipMemcpy(df, f.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dd, d.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(di, iv.data(), n * 4, hipMemcpyHostToDevice); hipMemset(dbad, 0, 8);
    hipLaunchKernelGGL(k_check, dim3(rows / 4), dim3(256), 0, 0, df, dd, di, rows, dbad);
    unsigned long long bad = 0; hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost);
    printf("bit-exactness: %d rows x 64 lanes x 5 reductions (f32 sum, f32 max, i32 sum, i32 max, f64 max): %llu mismatches\n", rows, bad);
    const char* names[4] = {"__shfl_xor f32 sum", "wave_bfly  f32 sum", "__shfl_xor f64 max", "wave_bfly  f64 max"};
    for (int mode = 0; mode < 4; ++mode) {
        const int reps = 10000; long long c = 0;
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(k_time<0>, dim3(1), dim3(64), 0, 0, df, dout, dcyc, reps);
            if (mode == 1) hipLaunchKernelGGL(k_time<1>, dim3(1), dim3(64), 0, 0, df, dout, dcyc, reps);
            if (mode == 2) hipLaunchKernelGGL(k_time<2>, dim3(1), dim3(64), 0, 0, df, dout, dcyc, reps);
            if (mode == 3) hipLaunchKernelGGL(k_time<3>, dim3(1), dim3(64), 0, 0, df, dout, dcyc, reps);
            hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
        }
        printf("%s: %.1f cycles per reduction (lone wave, dependent chain)\n", names[mode], (double)c / reps);
    }
    return bad ? 1 : 0;
}
