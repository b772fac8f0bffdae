// Training-side kernels of the only trainable layer (Dense 'connected_0'):
//   weight gradient dWt = dz^T . x on the exact-f32 matrix cores (the bias gradient rides along), and the Adam step.
// Replaces the Keras autodiff of /root/reference/models/model_tiny_yolov1.py:96 and
// Adam(lr=0.001) of /root/reference/train.py:59.
#include "common.h"
#include <cmath>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// GEMM view: M = units (u), N = K features (k), reduction over the batch (tiny: 32 per GPU).
// The reduction dimension is so short that both MFMA operands are read straight from global memory
// (dz row b: 32 consecutive u per half-wave, x row b: 32 consecutive k: both 128-byte segments);
// the kernel is bound by the 73.8 MB dWt store.  Block = 4 waves (2x2), wave = 2x2 tiles of 32x32.
// db[u] = sum_n dz[n][u]: the first column of blocks already holds every dz value of its 128 units in the A operands, so
// its k-side-0 waves add them up on the way (even and odd images in the two lane halves, joined by one shuffle).
__global__ __launch_bounds__(256) void dense_wgrad_kernel(const float *__restrict__ dz, const float *__restrict__ x,
                                                          float *__restrict__ dwt, float *__restrict__ db, int N, int K,
                                                          int units, int ldk, int ldz) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wu = wave >> 1, wk = wave & 1;
    const int u0 = blockIdx.y * 128 + wu * 64, k0 = blockIdx.x * 128 + wk * 64;
    const int i = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const bool uok0 = u0 + i < units, uok1 = u0 + 32 + i < units;
    const bool kok0 = k0 + i < K, kok1 = k0 + 32 + i < K;
    // Operand fetch through buffer resources: the lane's part of the address (column, image parity) is a constant VGPR, the
    // image pair is a SCALAR offset, rows past N and columns past units / K fall outside the resource and read as zero - no
    // address arithmetic and no selects in the vector ALU, which the exact-f32 MFMA shares (conv_gemm.hip: every vector
    // instruction between two MFMAs costs matrix cycles).  Matters when the loop is long: the gathered rows of a
    // data-parallel group (Trainer.exchange = 'gather', 64 ... 512 rows); one rank's 32 rows are bound by the 73.8 MB store.
    constexpr unsigned kOOB = 0xFFFFFFFFu;
    const __amdgpu_buffer_rsrc_t dz_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dz), 0, (int)((long long)N * ldz * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (int)((long long)N * K * 4), 0x00020000);
    const unsigned va0 = uok0 ? (unsigned)(h * ldz + u0 + i) * 4u : kOOB, va1 = uok1 ? (unsigned)(h * ldz + u0 + 32 + i) * 4u : kOOB;
    const unsigned vb0 = kok0 ? (unsigned)(h * K + k0 + i) * 4u : kOOB, vb1 = kok1 ? (unsigned)(h * K + k0 + 32 + i) * 4u : kOOB;
    float bsum0 = 0.f, bsum1 = 0.f;
    for (int b = 0; b < N; b += 2) {                 // images b + h in the two half-waves
        const int sa = b * ldz * 4, sb = b * K * 4;
        const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dz_rsrc, (int)va0, sa, 0));
        const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dz_rsrc, (int)va1, sa, 0));
        const float b0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, (int)vb0, sb, 0));
        const float b1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, (int)vb1, sb, 0));
        bsum0 += a0;                                 // (only the first column of workgroups uses them; a branch around the two adds measured slower)
        bsum1 += a1;
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (blockIdx.x == 0 && wk == 0) {
        bsum0 += __shfl_xor(bsum0, 32, 64);
        bsum1 += __shfl_xor(bsum1, 32, 64);
        if (h == 0 && uok0) db[u0 + i] = bsum0;
        if (h == 0 && uok1) db[u0 + 32 + i] = bsum1;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int k = k0 + 32 * b + i;
            if (k >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int u = u0 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (u < units) dwt[(size_t)u * ldk + k] = acc[a][b][r];
            }
        }
}

__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, long long n,
                                                   float lr_t, float b1, float b2, float eps) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4 *>(p)[i], gg = reinterpret_cast<const float4 *>(g)[i];
        float4 mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
#define TY1_ADAM(c)                                       \
    mm.c = b1 * mm.c + (1.f - b1) * gg.c;                 \
    vv.c = b2 * vv.c + (1.f - b2) * (gg.c * gg.c);        \
    pp.c = pp.c - lr_t * mm.c / (sqrtf(vv.c) + eps);
        TY1_ADAM(x) TY1_ADAM(y) TY1_ADAM(z) TY1_ADAM(w)
#undef TY1_ADAM
        reinterpret_cast<float4 *>(p)[i] = pp;
        reinterpret_cast<float4 *>(m)[i] = mm;
        reinterpret_cast<float4 *>(v)[i] = vv;
    }
    // tail (n not a multiple of 4)
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * (gi * gi);
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
    }
}

}  // namespace

extern "C" int ty1_dense_wgrad_f32(const float *dz, const float *x, float *dwt, float *db, int N, int K, int units,
                                   int ldk, int ldz, ty1_stream_t stream) {
    TY1_REQUIRE(N >= 0 && K > 0 && units > 0, "ty1_dense_wgrad: bad dimensions");
    TY1_REQUIRE(dz && x && dwt && db, "ty1_dense_wgrad: null pointer");
    TY1_REQUIRE(ldk >= K && ldz >= units, "ty1_dense_wgrad: leading dimensions too small");
    TY1_REQUIRE((long long)N * K * 4 < (1ll << 31) && (long long)N * ldz * 4 < (1ll << 31), "ty1_dense_wgrad: dz and x must stay below 2 GiB each (N=%d)", N);
    hipStream_t s = (hipStream_t)stream;
    ty1::ProfScope prof("dense_wgrad", s);
    dim3 grid((K + 127) / 128, (units + 127) / 128);
    dense_wgrad_kernel<<<grid, dim3(256), 0, s>>>(dz, x, dwt, db, N, K, units, ldk, ldz);
    return ty1::check_launch("ty1_dense_wgrad_f32");
}

extern "C" int ty1_adam_step_f32(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1,
                                 float beta2, float eps, int64_t t, ty1_stream_t stream) {
    TY1_REQUIRE(n >= 0 && t >= 1, "ty1_adam_step: bad n=%lld or t=%lld", (long long)n, (long long)t);
    if (n == 0) return 0;
    TY1_REQUIRE(p && g && m && v, "ty1_adam_step: null pointer");
    TY1_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "ty1_adam_step: pointers must be 16-byte aligned");
    // Keras: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
    const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)t)) / (1.0 - std::pow((double)beta1, (double)t));
    hipStream_t s = (hipStream_t)stream;
    long long blocks = (n / 4 + 255) / 256;
    // one workgroup per CU: alone the pass then takes 90 us instead of 78 (516 MB), but next to the MFMA-bound conv layers of
    // the next batch - where the trainer runs it - four waves per CU take fewer issue slots from them than sixteen
    // (training step at 32 images 1.016 -> 1.000 ms, tools/experiments/adam_grid_ab.sh; 128 per chip is too few: 145 us alone)
    if (blocks > ty1::num_cus()) blocks = ty1::num_cus();
    if (const char *e = ty1::dev_env("TY1_ADAM_BLOCKS")) blocks = atoll(e) > 0 && atoll(e) < blocks ? atoll(e) : blocks;     // developer sweep
    if (blocks < 1) blocks = 1;
    ty1::ProfScope prof("adam_step", s);
    adam_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(p, g, m, v, n, (float)lr_t, beta1, beta2, eps);
    return ty1::check_launch("ty1_adam_step_f32");
}
