// Developer microbenchmark: the conv kernel's inner loop in isolation (ds_read_b128 fragments + 32x32x2 f32 MFMAs,
// no global loads, no barriers), random operands.  Variants: wave tile 2x2 / 1x2 / 2x4, waves per block, blocks per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WM, int WN, int WAVES, bool USE_LDS>
__global__ __launch_bounds__(64 * WAVES) void k(const float *src, float *out, long long *cyc, int iters) {
    constexpr int STRIDE = 36, ROWS = 256;
    __shared__ __attribute__((aligned(16))) float lds[ROWS * STRIDE];
    for (int i = threadIdx.x; i < ROWS * STRIDE; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[WM][WN];
    for (int i = 0; i < WM; ++i)
        for (int j = 0; j < WN; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int row = lane & 31, fk = 4 * (lane >> 5);
    const int arow = (wave & 1) * 32 * WM % 128 + row, brow = 128 + ((wave >> 1) * 32 * WN) % 128 + row;
    float4 ra[WM], rb[WN];
    for (int i = 0; i < WM; ++i) ra[i] = *(const float4 *)&lds[(arow + 32 * i) % ROWS * STRIDE + fk];
    for (int j = 0; j < WN; ++j) rb[j] = *(const float4 *)&lds[(brow + 32 * j) % ROWS * STRIDE + fk];
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float4 fa[WM], fb[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i)
                fa[i] = USE_LDS ? *(const float4 *)&lds[((arow + 32 * i) % ROWS) * STRIDE + kb * 8 + fk] : ra[i];
#pragma unroll
            for (int j = 0; j < WN; ++j)
                fb[j] = USE_LDS ? *(const float4 *)&lds[((brow + 32 * j) % ROWS) * STRIDE + kb * 8 + fk] : rb[j];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (!USE_LDS) {
            for (int i = 0; i < WM; ++i) asm volatile("" : "+v"(ra[i].x), "+v"(ra[i].y), "+v"(ra[i].z), "+v"(ra[i].w));
        } else {
            asm volatile("" ::: "memory");
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < WM; ++i)
        for (int j = 0; j < WN; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int WM, int WN, int WAVES, bool USE_LDS>
void run(const char *name, int blocks, const float *src) {
    float *out; long long *cyc;
    (void)hipMalloc(&out, sizeof(float) * blocks * 64 * WAVES);
    (void)hipMalloc(&cyc, sizeof(long long) * blocks);
    const int iters = 16384 / (WM * WN);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<WM, WN, WAVES, USE_LDS><<<blocks, 64 * WAVES>>>(src, out, cyc, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<WM, WN, WAVES, USE_LDS><<<blocks, 64 * WAVES>>>(src, out, cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[1]; (void)hipMemcpy(h, cyc, sizeof(long long), hipMemcpyDeviceToHost);
    double mf = 16.0 * WM * WN * iters;
    double tf = (double)blocks * WAVES * mf * 4096 / (ms * 1e-3) / 1e12;
    printf("%-34s blocks=%4d waves/blk=%d  waves/SIMD=%.1f  memtime/MFMA=%.1f  wall=%.3f ms  %.1f TFLOP/s  clk~%.2f GHz\n", name,
           blocks, WAVES, blocks * WAVES / 1024.0, (double)h[0] / mf, ms, tf, 0.0);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    const int n = 256 * 36;
    float *h = (float *)malloc(n * 4), *d;
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    (void)hipMalloc(&d, n * 4);
    (void)hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    run<2, 2, 4, false>("regs 2x2, 1 blk/CU", 256, d);
    run<2, 2, 4, false>("regs 2x2, 2 blk/CU", 512, d);
    run<2, 2, 4, true>("lds  2x2, 1 blk/CU", 256, d);
    run<2, 2, 4, true>("lds  2x2, 2 blk/CU", 512, d);
    run<2, 2, 4, true>("lds  2x2, 4 blk/CU (2 rounds)", 1024, d);
    run<1, 2, 8, true>("lds  1x2 8 waves, 2 blk/CU", 512, d);
    run<2, 4, 4, true>("lds  2x4, 1 blk/CU", 256, d);
    run<2, 4, 4, true>("lds  2x4, 2 blk/CU", 512, d);
    run<1, 1, 8, true>("lds  1x1 8 waves, 2 blk/CU", 512, d);
    return 0;
}
