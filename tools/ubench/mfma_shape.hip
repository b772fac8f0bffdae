// Developer microbenchmark: sustained (seconds-long) f32 MFMA loops fed from LDS with random data:
// v_mfma_f32_32x32x2_f32 (2x2 tiles per wave) vs v_mfma_f32_16x16x4_f32 (4x4 tiles per wave), same 64x64 wave tile.
// Reports TFLOP/s and the shader clock (s_memtime / s_memrealtime) of the last launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int STRIDE = 68, ROWS = 256;   // 64 k values per row (+4 pad): 8 k-blocks of 8

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const float *src, float *out, long long *stamps, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[ROWS * STRIDE];
    for (int i = threadIdx.x; i < ROWS * STRIDE; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int abase = (wave & 1) * 64, bbase = 128 + (wave >> 1) * 64;
    float s = 0;
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (SHAPE == 32) {
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int row = lane & 31, fk = 4 * (lane >> 5);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                float4 fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = *(const float4 *)&lds[(abase + row + 32 * i) * STRIDE + kb * 8 + fk];
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = *(const float4 *)&lds[(bbase + row + 32 * j) * STRIDE + kb * 8 + fk];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                    }
            }
            asm volatile("" ::: "memory");
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    } else {
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        const int row = lane & 15, fk = 4 * (lane >> 4);   // lane holds k = fk..fk+3 of a 16-k block
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {               // 16 k values per step
                float4 fa[4], fb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *(const float4 *)&lds[(abase + row + 16 * i) * STRIDE + kb * 16 + fk];
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = *(const float4 *)&lds[(bbase + row + 16 * j) * STRIDE + kb * 16 + fk];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                    }
            }
            asm volatile("" ::: "memory");
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

template <int SHAPE>
void run(const char *name, int blocks, const float *src, double seconds) {
    float *out; long long *st;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    (void)hipMalloc(&st, 16);
    const int iters = 2048;                                  // 2048 * 64 k * 64x64 per wave
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double flop = (double)blocks * 4 * iters * 2.0 * 64 * 64 * 64;
    float ms = 0; int n = 0; double total = 0;
    while (total < seconds * 1e3) {
        (void)hipEventRecord(e0);
        k<SHAPE><<<blocks, 256>>>(src, out, st, iters);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
        total += ms; ++n;
        if (n == 1 || n == 3) printf("   launch %d: %.3f ms %.1f TFLOP/s\n", n, ms, flop / (ms * 1e-3) / 1e12);
    }
    long long h[2]; (void)hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    printf("%-28s blocks=%d launches=%d last=%.3f ms  %.1f TFLOP/s  clock %.3f GHz\n", name, blocks, n, ms,
           flop / (ms * 1e-3) / 1e12, (double)h[0] / (double)h[1] * 0.1);
    (void)hipFree(out); (void)hipFree(st);
}

int main(int argc, char **argv) {
    double secs = argc > 1 ? atof(argv[1]) : 2.0;
    const int n = ROWS * STRIDE;
    float *h = (float *)malloc(n * 4), *d;
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    (void)hipMalloc(&d, n * 4);
    (void)hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    run<32>("32x32x2  2x2/wave 2blk/CU", 512, d, secs);
    run<16>("16x16x4  4x4/wave 2blk/CU", 512, d, secs);
    run<32>("32x32x2  2x2/wave 1blk/CU", 256, d, secs);
    run<16>("16x16x4  4x4/wave 1blk/CU", 256, d, secs);
    for (int i = 0; i < n; ++i) h[i] = 0.f;
    (void)hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    run<32>("32x32x2 zeros 2blk/CU", 512, d, secs);
    return 0;
}
