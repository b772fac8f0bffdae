// Split-K Dense GEMM on the exact-f32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32):
//   part[s][n][u] = sum_{k in slice s} x[n][k] * wt[u][k]
// Replaces the matmul of Flatten + Dense(1470, 'linear', name='connected_0'):
// /root/reference/models/model_tiny_yolov1.py:89,96.  Bias add and the reduction over s are fused into
// ty1_head_activate_f32 (head.hip).
//
// Both operands are [row][k] with k contiguous (x is the HWC-flattened feature matrix, wt the transposed Dense
// kernel), so both LDS tiles are [row][BK] and every lane fetches four k values with one ds_read_b128 (row stride
// BK + 4 floats: conflict-free for the 16-lane groups).  A workgroup of 4 waves owns one BM x BN output tile and one
// K slice; M = batch is small (1..128), so the K split is what fills the chip (ops.pick_splits) and at N = 1 the
// kernel is the 73.8 MB weight stream.  Operands are fetched with raw buffer loads (out-of-range offset = 0: M / N
// tails and the drained pipeline need no branches); the loop body is the software pipeline of conv_persist_kernel
// (conv_gemm.hip): LDS store of the next chunk + re-issued fetch right after the barrier, fragments of k-block
// kb + 1 read while kb multiplies, 8 MFMAs kept back across the barrier.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct DenseArgs {
    const float *x;      // [M][K]
    const float *w;      // [units][ldk]
    float *part;         // [splits][M][ldz]
    int M, K, units, ldk, ldz;
    int chunks_total;    // K / BK
    int splits;          // gridDim.y: block (.., s) reduces chunks [chunks_total*s/splits, chunks_total*(s+1)/splits)
    int ntiles;
};

template <int WM, int WN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void dense_gemm_f32_kernel(const DenseArgs a) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    constexpr int BK = 32, NKB = BK / 8;
    constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
    constexpr int LDS_STRIDE = BK + 4;
    constexpr int CPR = BK / 4, RPP = 256 / CPR;
    constexpr int AP = (BM + RPP - 1) / RPP, BP = (BN + RPP - 1) / RPP;
    constexpr unsigned kOOB = 0xFFFFFFFFu;

    __shared__ __attribute__((aligned(16))) float As[2][BM * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDS_STRIDE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int mt = blockIdx.x / a.ntiles, nt = blockIdx.x - mt * a.ntiles;
    const int split = blockIdx.y;
    const int piece = tid % CPR, prow = tid / CPR;

    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x), 0, (int)((long long)a.M * a.K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.w), 0, (int)((long long)a.units * a.ldk * 4), 0x00020000);
    unsigned a_base[AP], b_base[BP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        const int row = p * RPP + prow, m = mt * BM + row;
        a_base[p] = (row < BM && m < a.M) ? (unsigned)((long long)m * a.K * 4) : kOOB;
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) {
        const int row = p * RPP + prow, n = nt * BN + row;
        b_base[p] = (row < BN && n < a.units) ? (unsigned)((long long)n * a.ldk * 4) : kOOB;
    }
    const int kc_begin = (int)((long long)a.chunks_total * split / a.splits);
    const int kc_end = (int)((long long)a.chunks_total * (split + 1) / a.splits);

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[AP], rb[BP];
    int ld_kc = kc_begin;
    auto load_chunk = [&]() {                 // past kc_end every lane goes out of range and nothing is fetched
        const bool live = ld_kc < kc_end;
        const unsigned koff = (unsigned)(ld_kc * BK + piece * 4) * 4u;
#pragma unroll
        for (int p = 0; p < AP; ++p)
            ra[p] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)((a_base[p] == kOOB || !live) ? kOOB : a_base[p] + koff), 0, 0);
#pragma unroll
        for (int p = 0; p < BP; ++p)
            rb[p] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (int)((b_base[p] == kOOB || !live) ? kOOB : b_base[p] + koff), 0, 0);
        ++ld_kc;
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const int row = p * RPP + prow;
            if (AP * RPP == BM || row < BM) *reinterpret_cast<u32x4 *>(&As[buf][row * LDS_STRIDE + piece * 4]) = ra[p];
        }
#pragma unroll
        for (int p = 0; p < BP; ++p) {
            const int row = p * RPP + prow;
            if (BP * RPP == BN || row < BN) *reinterpret_cast<u32x4 *>(&Bs[buf][row * LDS_STRIDE + piece * 4]) = rb[p];
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_row = lane & 31, frag_k = 4 * (lane >> 5);
    const int a_row0 = wave_m * (32 * WM) + frag_row, b_row0 = wave_n * (32 * WN) + frag_row;
    struct Frag {        // one k-block (8 k values: each lane holds 4 of them, the MFMA pairs the lane halves)
        float a[WM][4], b[WN][4];
    };
    auto read_frag = [&](Frag &f, int buf, int kb) {
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const float4 v = *reinterpret_cast<const float4 *>(
                __builtin_assume_aligned(&As[buf][(a_row0 + 32 * i) * LDS_STRIDE + kb * 8 + frag_k], 16));
            f.a[i][0] = v.x; f.a[i][1] = v.y; f.a[i][2] = v.z; f.a[i][3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(
                __builtin_assume_aligned(&Bs[buf][(b_row0 + 32 * j) * LDS_STRIDE + kb * 8 + frag_k], 16));
            f.b[j][0] = v.x; f.b[j][1] = v.y; f.b[j][2] = v.z; f.b[j][3] = v.w;
        }
    };
    auto mfma = [&](const Frag &f, int k0, int k1) {
#pragma unroll
        for (int k = k0; k < k1; ++k)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][k], f.b[j][k], acc[i][j], 0, 0, 0);
    };

    load_chunk();
    store_chunk(0);
    load_chunk();
    __syncthreads();
    // one basic block per chunk, identical for every chunk: the first pass multiplies a zero fragment, the last pass
    // stores / fetches nothing real (the fetch cursor is past kc_end)
    Frag f0, f1;
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) f1.b[j][k] = 0.f;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) f1.a[i][k] = 0.f;
    int buf = 0;
    read_frag(f0, 0, 0);
    for (int kc = kc_begin; kc < kc_end; ++kc) {
        store_chunk(buf ^ 1);                     // that buffer was last read before the previous barrier
        load_chunk();
        mfma(f1, 2, 4);
        read_frag(f1, buf, 1);
        mfma(f0, 0, 4);
        read_frag(f0, buf, 2);
        mfma(f1, 0, 4);
        read_frag(f1, buf, 3);
        mfma(f0, 0, 4);
        mfma(f1, 0, 2);
        static_assert(NKB == 4, "four k-blocks per chunk");
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        buf ^= 1;
        read_frag(f0, buf, 0);
    }
    mfma(f1, 2, 4);

    // lane holds unit (lane&31) of tile column j; register r is tile row (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int h = lane >> 5;
    float *out = a.part + (long long)split * a.M * a.ldz;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = nt * BN + wave_n * (32 * WN) + 32 * j + (lane & 31);
        if (n >= a.units) continue;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int m0 = mt * BM + wave_m * (32 * WM) + 32 * i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < a.M) out[(long long)m * a.ldz + n] = acc[i][j][r];
            }
        }
    }
}

template <int WM, int WN, int WAVES_M, int WAVES_N>
int launch_dense(const char *name, DenseArgs a, int mtiles, hipStream_t s) {
    const double flops = 2.0 * a.M * (double)a.units * a.K;
    const double bytes = 4.0 * ((double)a.M * a.K + (double)a.units * a.K + (double)a.splits * a.M * a.units);
    char tag[96];
    snprintf(tag, sizeof tag, "%s@1x%d>%d", name, a.K, a.units);
    ty1::ProfScope prof(tag, s, flops, bytes);
    dense_gemm_f32_kernel<WM, WN, WAVES_M, WAVES_N><<<dim3(mtiles * a.ntiles, a.splits), 256, 0, s>>>(a);
    return ty1::check_launch(name);
}

}  // namespace

extern "C" int ty1_dense_splitk_f32(const float *x, const float *wt, float *part, int N, int K, int units, int ldk,
                                    int ldz, int splits, ty1_stream_t stream) {
    TY1_REQUIRE(N >= 0 && K > 0 && units > 0 && splits > 0, "ty1_dense_splitk: bad dimensions");
    if (N == 0) return 0;
    TY1_REQUIRE(x && wt && part, "ty1_dense_splitk: null pointer");
    TY1_REQUIRE(ldk >= K && ldk % 4 == 0 && K % 4 == 0, "ty1_dense_splitk: ldk=%d must be >= K=%d, both multiples of 4", ldk, K);
    TY1_REQUIRE(ldz >= units, "ty1_dense_splitk: ldz=%d < units=%d", ldz, units);
    TY1_REQUIRE(K % 32 == 0 && splits <= K / 32, "ty1_dense_splitk: K=%d must be a multiple of 32 and splits=%d at most K/32", K, splits);
    TY1_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)wt & 15) == 0, "ty1_dense_splitk: x and wt must be 16-byte aligned");
    TY1_REQUIRE((long long)N * K * 4 < (1ll << 32) && (long long)units * ldk * 4 < (1ll << 32),
                "ty1_dense_splitk: operands must stay below 4 GiB each");
    DenseArgs a{};
    a.x = x; a.w = wt; a.part = part;
    a.M = N; a.K = K; a.units = units; a.ldk = ldk; a.ldz = ldz;
    a.chunks_total = K / 32;
    a.splits = splits;
    hipStream_t s = (hipStream_t)stream;
    if (N <= 32) {
        a.ntiles = (units + 127) / 128;
        return launch_dense<1, 1, 1, 4>("dense_f32_32x128", a, (N + 31) / 32, s);
    }
    a.ntiles = (units + 63) / 64;
    return launch_dense<1, 1, 2, 2>("dense_f32_64x64", a, (N + 63) / 64, s);
}
