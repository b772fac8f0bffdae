// Developer microbenchmark: cycles per v_mfma_f32_32x32x2_f32 for dependent vs interleaved accumulator chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a0 = threadIdx.x * 0.001f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    float b0 = 0.5f, b1 = 0.25f, b2 = 0.125f, b3 = 0.0625f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 4 dependent MFMAs per accumulator, accumulators in sequence
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3, acc[i], 0, 0, 0);
            }
        } else if (MODE == 1) {   // interleaved: consecutive MFMAs hit different accumulators
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, b3, acc[i], 0, 0, 0);
        } else {                  // single accumulator, fully dependent
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
        }
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int blocks, int threads) {
    float *out; long long *cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, cyc, sizeof(long long) * 8, hipMemcpyDeviceToHost);
    double waves_per_simd = blocks * (threads / 64) / 1024.0;
    double mf = 16.0 * iters;
    double tf = (double)blocks * (threads / 64) * mf * 4096 / (ms * 1e-3) / 1e12;
    printf("%-28s blocks=%d thr=%d waves/SIMD=%.1f  cyc/MFMA/wave=%.1f  wall=%.3f ms  %.1f TFLOP/s\n", name, blocks, threads,
           waves_per_simd, (double)h[0] / mf, ms, tf);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("dep4 x 4acc", 256, 256);
    run<1>("interleaved 4acc", 256, 256);
    run<2>("single chain", 256, 256);
    run<0>("dep4 x 4acc 2w/SIMD", 512, 256);
    run<1>("interleaved 4acc 2w/SIMD", 512, 256);
    run<2>("single chain 2w/SIMD", 512, 256);
    return 0;
}
