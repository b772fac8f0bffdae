// Developer microbenchmark: what do other instructions issued between f32 MFMAs cost?  An LDS-fed 32x32x2 f32 loop
// (2x2 tiles per wave, 16 MFMAs per k-block) with NV extra VALU / NL extra LDS-store / NG extra global-load instructions
// per k-block, two workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int STRIDE = 68, ROWS = 256;

template <int NV, int NW, int NG, int ND = 0>
__global__ __launch_bounds__(256, 2) void k(const float *src, float *out, int iters, const float4 *g) {
    __shared__ __attribute__((aligned(16))) float lds[ROWS * STRIDE + 4096];
    for (int i = threadIdx.x; i < ROWS * STRIDE; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int abase = (wave & 1) * 64, bbase = 128 + (wave >> 1) * 64;
    float s = 0;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int row = lane & 31, fk = 4 * (lane >> 5);
    unsigned v0 = lane, v1 = wave;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)g, 0, 16384, 0x00020000);
    f4 gl = {0, 0, 0, 0};
    const unsigned saddr = (unsigned)(size_t)((float4 *)&lds[ROWS * STRIDE] + threadIdx.x);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            float4 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *(const float4 *)&lds[(abase + row + 32 * i) * STRIDE + kb * 8 + fk];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *(const float4 *)&lds[(bbase + row + 32 * j) * STRIDE + kb * 8 + fk];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][q], fb[j][q], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < NV / 4; ++e) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v0) : "v"(v1));
                if (q < NW) asm volatile("ds_write_b128 %0, %1" ::"v"(saddr), "v"(gl) : "memory");
                if (q < ND) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(lds + ROWS * STRIDE + 256 * wave), 16, (lane * 16 + v1 * 1024) & 16383, 0, 0, 0);
                }
                if (q < NG) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gl) : "v"(g + ((v1 * 64 + lane) & 1023)) : "memory");
            }
        }
        asm volatile("" ::: "memory");
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + v0 + gl.x;
}

template <int NV, int NW, int NG, int ND = 0>
void run(const float *src, const float4 *g, double seconds) {
    const int blocks = 512, iters = 1024;
    float *out;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double flop = (double)blocks * 4 * iters * 2.0 * 64 * 64 * 64;
    float ms = 0, best = 1e9; double total = 0;
    while (total < seconds * 1e3) {
        (void)hipEventRecord(e0);
        k<NV, NW, NG, ND><<<blocks, 256>>>(src, out, iters, g);
        (void)hipEventRecord(e1);
        (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
        total += ms; if (ms < best) best = ms;
    }
    printf("per 16 MFMAs: %2d VALU %d ds_write_b128 %d global_load_x4 %d buffer_load_lds_x4: last %.3f ms %.1f TFLOP/s (best %.1f)\n", NV, NW, NG, ND, ms,
           flop / (ms * 1e-3) / 1e12, flop / (best * 1e-3) / 1e12);
    (void)hipFree(out);
}

int main(int argc, char **argv) {
    double secs = argc > 1 ? atof(argv[1]) : 0.5;
    const int n = ROWS * STRIDE;
    float *h = (float *)malloc(n * 4), *d; float4 *g;
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    (void)hipMalloc(&d, n * 4); (void)hipMalloc(&g, 1024 * 16); (void)hipMemset(g, 0, 1024 * 16);
    (void)hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    run<0, 0, 0>(d, g, secs);
    run<4, 0, 0>(d, g, secs);
    run<8, 0, 0>(d, g, secs);
    run<16, 0, 0>(d, g, secs);
    run<32, 0, 0>(d, g, secs);
    run<0, 1, 0>(d, g, secs);
    run<0, 2, 0>(d, g, secs);
    run<0, 4, 0>(d, g, secs);
    run<0, 0, 1>(d, g, secs);
    run<0, 0, 2>(d, g, secs);
    run<0, 0, 4>(d, g, secs);
    run<12, 2, 2>(d, g, secs);
    run<0, 0, 0, 1>(d, g, secs);
    run<0, 0, 0, 2>(d, g, secs);
    run<0, 0, 0, 4>(d, g, secs);
    return 0;
}
