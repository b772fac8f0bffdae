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
//
// ESZ = 2: the bf16 network (ty1_dense_splitk_bf16): x and wt are bf16, a K chunk is still 128 bytes of a row (64
// elements) and a 16-byte fragment is the operand of ONE v_mfma_f32_32x32x16_bf16; partial sums stay fp32.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct DenseArgs {
    const void *x;       // [M][K]       f32 or bf16
    const void *w;       // [units][ldk]
    float *part;         // [splits][M][ldz]
    int M, K, units, ldk, ldz;
    int chunks_total;    // K / (elements per 128-byte chunk)
    int splits;          // gridDim.y: block (.., s) reduces chunks [chunks_total*s/splits, chunks_total*(s+1)/splits)
    int ntiles;
};

template <int WM, int WN, int WAVES_M, int WAVES_N, int ESZ = 4>
__global__ __launch_bounds__(256) void dense_gemm_f32_kernel(const DenseArgs a) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    constexpr int BK = 32, NKB = BK / 8;      // BK in 4-byte units: a chunk is 128 bytes of a row
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
        const_cast<void *>(a.x), 0, (int)((long long)a.M * a.K * ESZ), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(a.w), 0, (int)((long long)a.units * a.ldk * ESZ), 0x00020000);
    // per-thread part of a fetch address (row, 16-byte piece of the chunk); the chunk itself is a SCALAR offset: no vector
    // arithmetic per load next to the MFMAs (conv_gemm.hip's rule; train.hip's weight-gradient loop gained 13-20 % from it)
    const unsigned a_bytes = (unsigned)((long long)a.M * a.K * ESZ), b_bytes = (unsigned)((long long)a.units * a.ldk * ESZ);
    unsigned a_base[AP], b_base[BP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        const int row = p * RPP + prow, m = mt * BM + row;
        a_base[p] = (row < BM && m < a.M) ? (unsigned)((long long)m * a.K * ESZ) + (unsigned)piece * 16u : kOOB;
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) {
        const int row = p * RPP + prow, n = nt * BN + row;
        b_base[p] = (row < BN && n < a.units) ? (unsigned)((long long)n * a.ldk * ESZ) + (unsigned)piece * 16u : kOOB;
    }
    const int kc_begin = (int)((long long)a.chunks_total * split / a.splits);
    const int kc_end = (int)((long long)a.chunks_total * (split + 1) / a.splits);

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[AP], rb[BP];
    int ld_kc = kc_begin;
    auto load_chunk = [&]() {                 // past kc_end the scalar offset is the size of the operand: every lane out of range, nothing fetched
        const bool live = ld_kc < kc_end;
        const unsigned koff = (unsigned)(ld_kc * BK) * 4u;
        const int sa = (int)(live ? koff : a_bytes), sb = (int)(live ? koff : b_bytes);
#pragma unroll
        for (int p = 0; p < AP; ++p) ra[p] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)a_base[p], sa, 0);
#pragma unroll
        for (int p = 0; p < BP; ++p) rb[p] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (int)b_base[p], sb, 0);
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
    // one k-block = 32 bytes of a row; each lane holds 16 of them: 4 floats (four MFMAs of k = 2, the lane halves pair
    // up) or 8 bf16 (one MFMA of k = 16)
    struct Frag {
        f32x4v a[WM], b[WN];
    };
    auto read_frag = [&](Frag &f, int buf, int kb) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
            f.a[i] = *reinterpret_cast<const f32x4v *>(
                __builtin_assume_aligned(&As[buf][(a_row0 + 32 * i) * LDS_STRIDE + kb * 8 + frag_k], 16));
#pragma unroll
        for (int j = 0; j < WN; ++j)
            f.b[j] = *reinterpret_cast<const f32x4v *>(
                __builtin_assume_aligned(&Bs[buf][(b_row0 + 32 * j) * LDS_STRIDE + kb * 8 + frag_k], 16));
    };
    // f32: k0..k1 of the four k pairs a fragment holds.  bf16: the fragment is one MFMA, issued with the first half.
    auto mfma = [&](const Frag &f, int k0, int k1) {
        if constexpr (ESZ == 4) {
#pragma unroll
            for (int k = k0; k < k1; ++k)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][k], f.b[j][k], acc[i][j], 0, 0, 0);
        } else if (k0 == 0) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[i]),
                                                                        __builtin_bit_cast(bf16x8, f.b[j]), acc[i][j], 0, 0, 0);
        }
    };

    load_chunk();
    store_chunk(0);
    load_chunk();
    __syncthreads();
    // one basic block per chunk, identical for every chunk: the first pass multiplies a zero fragment, the last pass
    // stores / fetches nothing real (the fetch cursor is past kc_end)
    Frag f0, f1;
#pragma unroll
    for (int j = 0; j < WN; ++j) f1.b[j] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < WM; ++i) f1.a[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};
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

template <int WM, int WN, int WAVES_M, int WAVES_N, int ESZ>
int launch_dense(const char *name, DenseArgs a, int mtiles, hipStream_t s) {
    const double flops = 2.0 * a.M * (double)a.units * a.K;
    const double bytes = ESZ * ((double)a.M * a.K + (double)a.units * a.K) + 4.0 * (double)a.splits * a.M * a.units;
    char tag[96];
    snprintf(tag, sizeof tag, "%s@1x%d>%d", name, a.K, a.units);
    ty1::ProfScope prof(tag, s, flops, bytes);
    dense_gemm_f32_kernel<WM, WN, WAVES_M, WAVES_N, ESZ><<<dim3(mtiles * a.ntiles, a.splits), 256, 0, s>>>(a);
    return ty1::check_launch(name);
}

template <int ESZ>
int dense_dispatch(const void *x, const void *wt, float *part, int N, int K, int units, int ldk, int ldz, int splits, hipStream_t s) {
    constexpr int CHUNK = 128 / ESZ;          // elements of K per chunk
    TY1_REQUIRE(N >= 0 && K > 0 && units > 0 && splits > 0, "ty1_dense_splitk: bad dimensions");
    if (N == 0) return 0;
    TY1_REQUIRE(x && wt && part, "ty1_dense_splitk: null pointer");
    TY1_REQUIRE(ldk >= K && (ldk * ESZ) % 16 == 0, "ty1_dense_splitk: ldk=%d must be >= K=%d with 16-byte aligned rows", ldk, K);
    TY1_REQUIRE(ldz >= units, "ty1_dense_splitk: ldz=%d < units=%d", ldz, units);
    TY1_REQUIRE(K % CHUNK == 0 && splits <= K / CHUNK, "ty1_dense_splitk: K=%d must be a multiple of %d and splits=%d at most K/%d",
                K, CHUNK, splits, CHUNK);
    TY1_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)wt & 15) == 0, "ty1_dense_splitk: x and wt must be 16-byte aligned");
    TY1_REQUIRE((long long)N * K * ESZ < (1ll << 32) && (long long)units * ldk * ESZ < (1ll << 32),
                "ty1_dense_splitk: operands must stay below 4 GiB each");
    DenseArgs a{};
    a.x = x; a.w = wt; a.part = part;
    a.M = N; a.K = K; a.units = units; a.ldk = ldk; a.ldz = ldz;
    a.chunks_total = K / CHUNK;
    a.splits = splits;
    if (N <= 32) {
        a.ntiles = (units + 127) / 128;
        return launch_dense<1, 1, 1, 4, ESZ>(ESZ == 4 ? "dense_f32_32x128" : "dense_bf16_32x128", a, (N + 31) / 32, s);
    }
    a.ntiles = (units + 63) / 64;
    // (bf16 above 64 rows on 128 x 64 tiles: GEMM + tail 39.3 -> 35.5 us alone, nothing on the whole network: not kept.  Round 3,
    //  inside the network at 128 images, tools/experiments/dense128_ab.sh: 64x64 21.9 us; one 128x64 tile per column panel - each
    //  weight byte fetched once - 21.3 us with 11 splits, 19.5 us with 22 but the tail behind it +1.9 us: the re-read traffic is
    //  not what bounds this kernel, its four MFMAs per wave and barrier are)
    // (four chunks in flight in registers instead of one, bf16: 39.6 vs 39.5 us - the two resident workgroups per CU already
    //  cover the fetch latency: not kept)
    return launch_dense<1, 1, 2, 2, ESZ>(ESZ == 4 ? "dense_f32_64x64" : "dense_bf16_64x64", a, (N + 63) / 64, s);
}

}  // namespace

extern "C" int ty1_dense_splitk_f32(const float *x, const float *wt, float *part, int N, int K, int units, int ldk,
                                    int ldz, int splits, ty1_stream_t stream) {
    return dense_dispatch<4>(x, wt, part, N, K, units, ldk, ldz, splits, (hipStream_t)stream);
}

extern "C" int ty1_dense_splitk_bf16(const void *x, const void *wt, float *part, int N, int K, int units, int ldk,
                                     int ldz, int splits, ty1_stream_t stream) {
    return dense_dispatch<2>(x, wt, part, N, K, units, ldk, ldz, splits, (hipStream_t)stream);
}
