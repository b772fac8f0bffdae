// Direct 3x3 convolution for a shallow input (Cin = 16: the second block of the backbone, K = 144), fused with
// folded BatchNorm + LeakyReLU + 2x2/2 max-pool, on v_mfma_f32_32x32x2_f32.
// Replaces /root/reference/models/model_tiny_yolov1.py:49-53 (convolution_1 block) for Cin = 16, Cout = 32 or 64.
//
// With K = 144 the implicit-GEMM pipeline (conv_gemm.hip) spends its time on per-tile overhead: a tile is nine short
// chunks.  Here the reduction is not tiled at all:
//   - the weights of the wave's 32 output channels live in REGISTERS for the whole kernel: MFMA B operand
//     B[k][col] with k = (tap, ci); lane (col = lane&31, h = lane>>5) holds w[col][tap][8g + 4h + j] (72 VGPRs per
//     32 channels), read once per workgroup;
//   - a persistent workgroup walks 8 x 32 pixel patches; the 10 x 34 pixel halo of the NEXT patch is fetched into
//     registers while the current one multiplies and stored to the other LDS buffer afterwards (coalesced 64-byte pixels),
//     pixel stride 20 floats and row stride = 32 mod 64 floats, which makes every ds_read_b128 of an A fragment
//     conflict-free for the 2-row x 16-column MFMA tile used below;
//   - an MFMA tile is 32 pixels = 8 pool windows (2 rows x 16 columns), rows ordered m = 4*window + (dy*2+dx), so the
//     four pixels of a window are the four consecutive accumulator registers of a lane (C/D row = (r&3) + 8(r>>2) +
//     4(lane>>5)) and the max-pool is three v_max in registers;
//   - one ds_read_b128 feeds four MFMAs (ci = 8g + 4h + j, j = 0..3); a wave owns 2 rows x 32 columns = two tiles;
//   - halo loads and output stores are buffer instructions (out-of-range offset = zero / dropped), so the patch loop
//     has no branches; with Cout = 32 the epilogue of a patch is issued between the MFMAs of the next one (second
//     accumulator set), and the pool runs before the activation (LeakyReLU with alpha >= 0 is non-decreasing).
// Two workgroups (2 x 29.4 KB of LDS each) share a CU.
#include "common.h"

#include <type_traits>

namespace {

// n / d for n < 2^31 by multiply-high and shift (round-up method)
struct FastDivU {
    unsigned M, l;
};
FastDivU make_fastdiv_u(unsigned d) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    return FastDivU{(unsigned)(((1ull << (32 + l)) / d) + 1ull - (1ull << 32)), l};
}
__device__ __forceinline__ unsigned fdiv_u(unsigned n, FastDivU f) { return (__umulhi(n, f.M) + n) >> f.l; }

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TH = 8, TW = 32;              // patch of pre-pool pixels per pipeline step
constexpr int CS = 20;                      // LDS floats per pixel (16 channels + 4 pad)
constexpr int RS = 736;                     // LDS floats per patch row: >= (TW+2)*CS = 680 and = 32 (mod 64)
constexpr int CIN = 16;
constexpr int STAGE_ITEMS = (TH + 2) * (TW + 2) * 4;            // 16-byte pieces of one halo
constexpr int STAGE_PER_THREAD = (STAGE_ITEMS + 255) / 256;     // 6

template <bool POOL, int NT>
__global__ __launch_bounds__(256) void conv3x3_c16_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ scale, const float *__restrict__ shift,
                                                          float *__restrict__ y, int NB, int H, int W, int Cout, float alpha,
                                                          int tiles_x, int tiles_y, int patches, FastDivU d_tx, FastDivU d_ty) {
    __shared__ __attribute__((aligned(16))) float tile[2][(TH + 2) * RS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;

    // ---- weights of this lane: wreg[nt][tap][g] = w[32nt + col][tap][8g + 4h .. +3]   (OHWI, 16 floats per tap)
    float4 wreg[NT][9][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int g = 0; g < 2; ++g)
                wreg[nt][t][g] = *reinterpret_cast<const float4 *>(w + ((size_t)(32 * nt + col) * 9 + t) * CIN + 8 * g + 4 * h);
    float sc[NT], sh[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        sc[nt] = scale[32 * nt + col];
        sh[nt] = shift[32 * nt + col];
    }

    // ---- halo staging through registers: fetch patch p+1 while patch p multiplies, store it after the MFMAs
    //      rows y0-1 .. y0+TH, pixels x0-1 .. x0+TW, 4 x 16-byte pieces per pixel.  The loads go through a buffer
    //      resource spanning ONE image: rows above / below it fall outside and read as zero, only the left / right
    //      border needs a test.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr unsigned kOOB = 0xFFFFFFFFu;
    u32x4 stage[STAGE_PER_THREAD];
    int lds_off[STAGE_PER_THREAD], st_pp[STAGE_PER_THREAD];
    unsigned st_off[STAGE_PER_THREAD];                       // byte offset of the piece in a patch at (y0, x0) = (1, 1)
#pragma unroll
    for (int k = 0; k < STAGE_PER_THREAD; ++k) {
        const int i = tid + 256 * k;
        const int piece = i & 3, pix = i >> 2;
        const int rr = pix / (TW + 2);
        st_pp[k] = i < STAGE_ITEMS ? pix - rr * (TW + 2) : -0x40000000;     // a dead piece never passes the border test
        lds_off[k] = i < STAGE_ITEMS ? rr * RS + (pix - rr * (TW + 2)) * CS + piece * 4 : -1;
        st_off[k] = (unsigned)(((rr * W + (pix - rr * (TW + 2))) * CIN + piece * 4) * 4);
    }
    auto decode = [&](int p, int &n, int &y0, int &x0) {
        p = __builtin_amdgcn_readfirstlane(p);               // block-uniform: keeps the resource words in SGPRs
        const unsigned r = fdiv_u((unsigned)p, d_tx), tx = (unsigned)p - r * (unsigned)tiles_x;
        const unsigned nn = fdiv_u(r, d_ty), ty = r - nn * (unsigned)tiles_y;
        n = __builtin_amdgcn_readfirstlane((int)nn);
        y0 = __builtin_amdgcn_readfirstlane((int)ty * TH);
        x0 = __builtin_amdgcn_readfirstlane((int)tx * TW);
    };
    auto fetch = [&](int p) {
        int n, y0, x0;
        decode(p, n, y0, x0);
        const bool live = p < patches;
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x) + (live ? (long long)n * H * W * CIN : 0ll), 0, live ? H * W * CIN * 4 : 0, 0x00020000);
        const unsigned patch_off = (unsigned)((((y0 - 1) * W + x0 - 1) * CIN) * 4);   // wraps for y0 = 0: outside the resource
#pragma unroll
        for (int k = 0; k < STAGE_PER_THREAD; ++k) {
            const bool in_row = (unsigned)(st_pp[k] + x0 - 1) < (unsigned)W;
            stage[k] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)(in_row ? st_off[k] + patch_off : kOOB), 0, 0);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int k = 0; k < STAGE_PER_THREAD; ++k)
            if (lds_off[k] >= 0) *reinterpret_cast<u32x4 *>(&tile[buf][lds_off[k]]) = stage[k];
    };

    // ---- this lane's pixel inside an MFMA tile: window wdw = (lane&31)>>2 of 8 along x, q = lane&3 inside the window
    const int q = lane & 3, wdw = (lane & 31) >> 2;
    const int lane_off = (q >> 1) * RS + (2 * wdw + (q & 1)) * CS + 4 * h;    // + tile origin + tap offset + 8g
    const int prow = 2 * wave;                               // wave owns pixel rows 2*wave, 2*wave+1 of the patch
    const int base0 = prow * RS + lane_off, base1 = base0 + 16 * CS;          // tile columns 0 and 1 (16 pixels apart)

    // ---- output: register group gq = r>>2 with lane half h is window wx = 8u + 2gq + h of the wave's row pair, r&3
    //      the position inside it.  Stores go through a buffer resource based at the patch's first output element:
    //      one 32-bit offset = lane part + compile-time piece part.
    const int Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
    const unsigned row_b = (unsigned)Wo * Cout * 4u, px_b = (unsigned)Cout * 4u;
    const unsigned out_lane = POOL ? (unsigned)wave * row_b + (unsigned)h * px_b + (unsigned)col * 4u
                                   : (unsigned)(2 * wave) * row_b + (unsigned)(2 * h) * px_b + (unsigned)col * 4u;
    const long long y_total = (long long)NB * Ho * Wo * Cout;
    // output descriptor and bounds of the patch whose epilogue is being issued: set once per patch (before the first patch
    // the descriptor has zero bytes and drops the stores); a patch inside the image needs no per-store test
    __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(y, 0, 0, 0x00020000);
    int rows_left = 0, cols_left = 0;
    bool out_full = true;
    auto set_out = [&](int n, int y0, int x0) {
        const int oy0 = POOL ? y0 >> 1 : y0, ox0 = POOL ? x0 >> 1 : x0;
        const long long o_org = (((long long)n * Ho + oy0) * Wo + ox0) * Cout;
        const long long o_left = (y_total - o_org) * 4;
        o_rsrc = __builtin_amdgcn_make_buffer_rsrc(y + o_org, 0, (int)(o_left > 0xFFFFFFFFll ? 0xFFFFFFFFll : o_left), 0x00020000);
        rows_left = Ho - oy0 - (POOL ? 1 : 2) * wave;
        cols_left = Wo - ox0 - (POOL ? 1 : 2) * h;
        out_full = y0 + TH <= H && x0 + TW <= W;
    };
    // piece idx = 4u + gq
    auto epi_piece = [&](auto full_c, int idx, const f32x16 (&acc)[2][NT]) {
        constexpr bool FULL = decltype(full_c)::value;
        const int u = idx >> 2, gq = idx & 3;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float z[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = acc[u][nt][4 * gq + r] * sc[nt] + sh[nt];
            if (POOL) {
                // LeakyReLU with alpha >= 0 is non-decreasing: pool first, activate once (same bits)
                const float m = fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3]));
                const float v = m > 0.f ? m : m * alpha;
                const bool ok = FULL || (0 < rows_left && 8 * u + 2 * gq < cols_left);
                const unsigned off = ok ? out_lane + (unsigned)(8 * u + 2 * gq) * px_b + 128u * nt : kOOB;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), o_rsrc, (int)off, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = z[r] > 0.f ? z[r] : z[r] * alpha;
                    const int dy = r >> 1, dx = 2 * (8 * u + 2 * gq) + (r & 1);
                    const bool ok = FULL || (dy < rows_left && dx < cols_left);
                    const unsigned off = ok ? out_lane + (unsigned)dy * row_b + (unsigned)dx * px_b + 128u * nt : kOOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), o_rsrc, (int)off, 0, 0);
                }
            }
        }
    };

    // With one channel tile per wave there are registers for a second accumulator set: the epilogue of a patch is
    // then issued between the MFMAs of the NEXT patch (one piece per tap) instead of after its own.
    constexpr bool PIPE = NT == 1;
    f32x16 acc[2][NT], prev[2][PIPE ? NT : 1];

    int p = blockIdx.x;
    fetch(p);
    store(0);
    __syncthreads();
    int buf = 0;
    for (; p < patches; p += gridDim.x) {
        fetch(p + gridDim.x);                               // lands during the MFMAs below (nothing past the end)
        int n, y0, x0;
        decode(p, n, y0, x0);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][nt][r] = 0.f;
        const float *tl = tile[buf];
        auto taps = [&](auto full_c) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int toff = (t / 3) * RS + (t % 3) * CS;   // halo origin is (y0-1, x0-1): tap (dy,dx) = (t/3-1, t%3-1)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const float4 a0 = *reinterpret_cast<const float4 *>(&tl[base0 + toff + 8 * g]);
                const float4 a1 = *reinterpret_cast<const float4 *>(&tl[base1 + toff + 8 * g]);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float4 wv = wreg[nt][t][g];
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, wv.x, acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, wv.x, acc[1][nt], 0, 0, 0);
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, wv.y, acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, wv.y, acc[1][nt], 0, 0, 0);
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, wv.z, acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, wv.z, acc[1][nt], 0, 0, 0);
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, wv.w, acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, wv.w, acc[1][nt], 0, 0, 0);
                }
            }
            if constexpr (PIPE) {
                if (t < 8) epi_piece(full_c, t, prev);
            }
        }
        };
        if (PIPE && !out_full) taps(std::false_type{});
        else taps(std::true_type{});
        store(buf ^ 1);                                     // that buffer was last read before the previous barrier
        if constexpr (!PIPE) {
            set_out(n, y0, x0);
            if (out_full) {
#pragma unroll
                for (int idx = 0; idx < 8; ++idx) epi_piece(std::true_type{}, idx, acc);
            } else {
#pragma unroll
                for (int idx = 0; idx < 8; ++idx) epi_piece(std::false_type{}, idx, acc);
            }
        }
        __syncthreads();
        buf ^= 1;
        if constexpr (PIPE) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) prev[u][nt] = acc[u][nt];
            set_out(n, y0, x0);
        }
    }
    if constexpr (PIPE) {
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) epi_piece(std::false_type{}, idx, prev);
    }
}


// ------------------------------------------------------------------------------------------------
// The same layer with bf16 activations and weights (fp32 accumulate, bf16 out) on v_mfma_f32_32x32x16_bf16:
// K = 16 of one MFMA is exactly the 16 input channels of one tap, so a 32-pixel tile is 9 MFMAs per 32 output
// channels; lane (pixel = lane&31, h = lane>>5) supplies channels 8h..8h+7 = one 16-byte LDS read, the weights
// of a lane are 9 x 16 bytes per 32 channels.  Halo in LDS: 32-byte pixels, row stride (TW+2)*32 + 16 bytes (an odd
// number of 16-byte slots: the two pixel rows of a tile then hit different banks).  With 16x fewer matrix cycles per
// byte than the f32 kernel this one is bound by its HBM traffic (155 MB at batch 64) and the LDS path.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BPX = 32;                                      // bytes per LDS pixel (16 bf16)
constexpr int BRS = (TW + 2) * BPX + 16;                     // bytes per LDS halo row
constexpr int BST_ITEMS = (TH + 2) * (TW + 2) * 2;           // 16-byte pieces of one halo
constexpr int BST_PER_THREAD = (BST_ITEMS + 255) / 256;      // 3

__device__ __forceinline__ unsigned short bf16_bits(float v) {     // round to nearest even (v_cvt_pk_bf16_f32)
    const __bf16 hv = (__bf16)v;
    return __builtin_bit_cast(unsigned short, hv);
}

template <bool POOL, int NT>
__global__ __launch_bounds__(256) void conv3x3_c16_bf16_kernel(const unsigned short *__restrict__ x, const unsigned short *__restrict__ w,
                                                               const float *__restrict__ scale, const float *__restrict__ shift,
                                                               unsigned short *__restrict__ y, int NB, int H, int W, int Cout,
                                                               float alpha, int tiles_x, int tiles_y, int patches,
                                                               FastDivU d_tx, FastDivU d_ty) {
    __shared__ __attribute__((aligned(16))) unsigned char tile[2][(TH + 2) * BRS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 31, h = lane >> 5;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr unsigned kOOB = 0xFFFFFFFFu;

    // ---- weights of this lane: wreg[nt][tap] = w[32nt + col][tap][8h .. 8h+7]   (OHWI bf16, 16 per tap)
    bf16x8 wreg[NT][9];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int t = 0; t < 9; ++t)
            wreg[nt][t] = *reinterpret_cast<const bf16x8 *>(w + ((size_t)(32 * nt + col) * 9 + t) * CIN + 8 * h);
    float sc[NT], sh[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        sc[nt] = scale[32 * nt + col];
        sh[nt] = shift[32 * nt + col];
    }

    // ---- halo staging through registers (see conv3x3_c16_kernel): 2 x 16-byte pieces per pixel
    u32x4 stage[BST_PER_THREAD];
    int lds_off[BST_PER_THREAD], st_pp[BST_PER_THREAD];
    unsigned st_off[BST_PER_THREAD];
#pragma unroll
    for (int k = 0; k < BST_PER_THREAD; ++k) {
        const int i = tid + 256 * k;
        const int half = i & 1, pix = i >> 1;
        const int rr = pix / (TW + 2), pp = pix - rr * (TW + 2);
        st_pp[k] = i < BST_ITEMS ? pp : -0x40000000;
        lds_off[k] = i < BST_ITEMS ? rr * BRS + pp * BPX + half * 16 : -1;
        st_off[k] = (unsigned)((rr * W + pp) * (CIN * 2) + half * 16);
    }
    // (a patch is only 18 MFMAs per wave: the scalar and vector bookkeeping around them is what this kernel's time is made
    //  of - 209 scalar + 128 vector instructions per patch before round 2c, most of them two integer divisions per decode,
    //  two decodes per patch and an output descriptor per epilogue piece; now magic-number division, one decode per patch
    //  carried over to the next iteration, one descriptor per patch, and no bounds tests for patches inside the image)
    auto decode = [&](int p, int &n, int &y0, int &x0) {
        p = __builtin_amdgcn_readfirstlane(p);
        const unsigned r = fdiv_u((unsigned)p, d_tx), tx = (unsigned)p - r * (unsigned)tiles_x;
        const unsigned nn = fdiv_u(r, d_ty), ty = r - nn * (unsigned)tiles_y;
        n = __builtin_amdgcn_readfirstlane((int)nn);
        y0 = __builtin_amdgcn_readfirstlane((int)ty * TH);
        x0 = __builtin_amdgcn_readfirstlane((int)tx * TW);
    };
    int fn = 0, fy0 = 0, fx0 = 0;                      // patch of the last fetch()
    auto fetch = [&](int p) {
        int n, y0, x0;
        decode(p < patches ? p : 0, n, y0, x0);
        fn = n; fy0 = y0; fx0 = x0;
        const bool live = p < patches;
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned short *>(x) + (live ? (long long)n * H * W * CIN : 0ll), 0, live ? H * W * CIN * 2 : 0, 0x00020000);
        const unsigned patch_off = (unsigned)(((y0 - 1) * W + x0 - 1) * (CIN * 2));   // wraps for y0 = 0: outside the resource
#pragma unroll
        for (int k = 0; k < BST_PER_THREAD; ++k) {
            const bool in_row = (unsigned)(st_pp[k] + x0 - 1) < (unsigned)W;
            stage[k] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, (int)(in_row ? st_off[k] + patch_off : kOOB), 0, 0);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int k = 0; k < BST_PER_THREAD; ++k)
            if (lds_off[k] >= 0) *reinterpret_cast<u32x4 *>(&tile[buf][lds_off[k]]) = stage[k];
    };

    const int q = lane & 3, wdw = (lane & 31) >> 2;
    const int lane_off = (q >> 1) * BRS + (2 * wdw + (q & 1)) * BPX + 16 * h;
    const int prow = 2 * wave;
    const int base0 = prow * BRS + lane_off, base1 = base0 + 16 * BPX;

    const int Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
    const unsigned row_b = (unsigned)Wo * Cout * 2u, px_b = (unsigned)Cout * 2u;
    const unsigned out_lane = POOL ? (unsigned)wave * row_b + (unsigned)h * px_b + (unsigned)col * 2u
                                   : (unsigned)(2 * wave) * row_b + (unsigned)(2 * h) * px_b + (unsigned)col * 2u;
    const long long y_total = (long long)NB * Ho * Wo * Cout;
    // output descriptor and bounds of the patch whose epilogue is in flight (set once per patch)
    __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(y, 0, 0, 0x00020000);
    int rows_left = 0, cols_left = 0;
    bool prev_full = true;
    auto set_prev = [&](int n, int y0, int x0) {
        const int oy0 = POOL ? y0 >> 1 : y0, ox0 = POOL ? x0 >> 1 : x0;
        const long long o_org = (((long long)n * Ho + oy0) * Wo + ox0) * Cout;
        const long long o_left = (y_total - o_org) * 2;
        o_rsrc = __builtin_amdgcn_make_buffer_rsrc(y + o_org, 0, (int)(o_left > 0xFFFFFFFFll ? 0xFFFFFFFFll : o_left), 0x00020000);
        rows_left = Ho - oy0 - (POOL ? 1 : 2) * wave;
        cols_left = Wo - ox0 - (POOL ? 1 : 2) * h;
        prev_full = y0 + TH <= H && x0 + TW <= W;
    };
    auto epi_piece = [&](auto full_c, int idx, const f32x16 (&acc)[2][NT]) {
        constexpr bool FULL = decltype(full_c)::value;
        const int u = idx >> 2, gq = idx & 3;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float z[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = acc[u][nt][4 * gq + r] * sc[nt] + sh[nt];
            if (POOL) {
                const float m = fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3]));
                const float v = m > 0.f ? m : m * alpha;
                const bool ok = FULL || (0 < rows_left && 8 * u + 2 * gq < cols_left);
                const unsigned off = ok ? out_lane + (unsigned)(8 * u + 2 * gq) * px_b + 64u * nt : kOOB;
                __builtin_amdgcn_raw_buffer_store_b16(bf16_bits(v), o_rsrc, (int)off, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = z[r] > 0.f ? z[r] : z[r] * alpha;
                    const int dy = r >> 1, dx = 2 * (8 * u + 2 * gq) + (r & 1);
                    const bool ok = FULL || (dy < rows_left && dx < cols_left);
                    const unsigned off = ok ? out_lane + (unsigned)dy * row_b + (unsigned)dx * px_b + 64u * nt : kOOB;
                    __builtin_amdgcn_raw_buffer_store_b16(bf16_bits(v), o_rsrc, (int)off, 0, 0);
                }
            }
        }
    };

    f32x16 acc[2][NT], prev[2][NT];
    int p = blockIdx.x;
    fetch(p);
    int n = fn, y0 = fy0, x0 = fx0;                    // the current patch (decoded by its fetch)
    store(0);
    __syncthreads();
    int buf = 0;
    // (before the first patch there is no epilogue in flight: the descriptor above has zero bytes and drops the stores)
    for (; p < patches; p += gridDim.x) {
        fetch(p + gridDim.x);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][nt][r] = 0.f;
        const unsigned char *tl = tile[buf];
        auto taps = [&](auto full_c) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int toff = (t / 3) * BRS + (t % 3) * BPX;
                const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(__builtin_assume_aligned(&tl[base0 + toff], 16));
                const bf16x8 a1 = *reinterpret_cast<const bf16x8 *>(__builtin_assume_aligned(&tl[base1 + toff], 16));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wreg[nt][t], acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wreg[nt][t], acc[1][nt], 0, 0, 0);
                }
                if (t < 8) epi_piece(full_c, t, prev);           // epilogue of the previous patch, one piece per tap
            }
        };
        if (prev_full) taps(std::true_type{});
        else taps(std::false_type{});
        store(buf ^ 1);
        __syncthreads();
        buf ^= 1;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) prev[u][nt] = acc[u][nt];
        set_prev(n, y0, x0);
        n = fn; y0 = fy0; x0 = fx0;
    }
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) epi_piece(std::false_type{}, idx, prev);
}


// ------------------------------------------------------------------------------------------------
// First layer (Cin = 3, K = 27 -> 28) as a persistent kernel on v_mfma_f32_16x16x4_f32: same arithmetic as
// conv3x3_c3_kernel (conv_gemm.hip: halo tile in LDS, 7 weight registers per 16 channels, one pool window per
// lane), but a workgroup walks 16 x 64 pixel patches and moves the NEXT patch's halo into the other LDS buffer by
// LDS-DMA while the current patch multiplies.  The layer is bound by its 406 MB of HBM traffic at batch 64 as much
// as by its MFMAs, so what matters is that loads are wide, always in flight, and that nothing drains the stores.
// Needs W % 4 == 0 and a 16-byte aligned input (an image row is then a whole number of float4).
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int C3_TH = 16, C3_TW = 64;
constexpr int C3_ROW = 208;                                  // LDS floats per halo row (200 used); 208 % 32 == 16
constexpr int C3_V4 = 50;                                    // float4 per halo row: floats 3*x0 - 4 .. 3*x0 + 195
constexpr int C3_TILE = (C3_TH + 2) * C3_ROW;                // floats of one halo buffer
constexpr int C3_ONES = 592;                                 // floats of 1.0 behind it (>= 2 * C3_ROW + 24 * 7 + 1), see FOLD
constexpr int C3_BUF = C3_TILE + C3_ONES;
// bf16 variant: which tap (k = 9 kh + 3 kw + ci) lane quarter kq supplies in slot e of the one K = 32 MFMA; -1 - offset: a zero-
// weight pad that reads the LDS float at `offset` from the pixel's first tap.  With k = 8 kq + e a 32-lane half of every
// A read hit 8 banks twice (SQ_LDS_BANK_CONFLICT 47 % of the LDS cycles, LDS instructions waiting 30 % of the wave cycles);
// this assignment, found by search over a model of the 32 banks that reproduces that figure, has no conflict (first layer of
// the bf16 network 117 -> 107 us alone, whole network +0.9 ... 3.6 %).  The same search for the f32 variant (26 -> 15 passes per
// 14 reads) made that kernel 1-2 % slower - there the LDS is busy 31 % of the time and the MFMA issue is the bound - not kept.
__device__ constexpr short C3_TAP_BF16[4][8] = {{22, 13, 0, 12, 20, 3, 16, 14},
                                                {19, 10, 8, 15, 26, 6, -1 - (C3_ROW + 1), 11},
                                                {21, 4, 17, 2, -1 - (2 * C3_ROW + 2), 18, 1, 25},
                                                {24, -1 - 4, 9, 5, 23, -1 - (C3_ROW + 8), 7, -1 - (2 * C3_ROW + 7)}};

template <bool POOL, int NT, bool OUT_BF16 = false>
__global__ __launch_bounds__(256) void conv3x3_c3p_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ scale, const float *__restrict__ shift,
                                                          void *__restrict__ y, int NB, int H, int W, int Cout, float alpha,
                                                          int tiles_x, int tiles_y, int patches, long long *stamp, int dbg,
                                                          FastDivU d_tx, FastDivU d_ty) {
    // developer knock-outs (dev build only; same instruction stream, no memory traffic): 1 drop the output stores, 2 drop the halo fetch
#ifdef TY1_DEV_STAMPS
#define C3_DBG(bit) ((dbg & (bit)) != 0)
#else
#define C3_DBG(bit) false
#endif
    __shared__ __attribute__((aligned(16))) float tile[2][C3_BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    // FOLD (f32): the folded BatchNorm rides in the MFMAs - the weights are multiplied by the channel's scale when they
    // are loaded, and the shift is the weight of the K pad slot k = 27, whose A operand is a 1.0 read from a block of
    // ones behind each halo buffer (every M-tile adds its own compile-time offset to the lane's address, so the block
    // covers the largest one).  The accumulator then IS scale * conv + shift and the epilogue loses its four v_fma per
    // tile: with exact-f32 MFMAs every vector instruction between two of them costs the matrix pipe ~8 cycles, and a
    // tile is only 7 MFMAs (measured per tile: 224 MFMA cycles + 99, of which the four v_fma were 34).  Results differ
    // from (conv * scale + shift) in the last bit or two, far inside the 1e-3 bar.  The bf16 variant keeps the separate
    // scale: its weights are rounded to bf16 and must be the oracle's rounded weights.
    constexpr bool FOLD = !OUT_BF16;

    // f32 (exact):  7 x v_mfma_f32_16x16x4_f32 per 16-pixel tile, lane quarter kq supplies k = 4*ks + kq of step ks.
    // OUT_BF16 (first layer of the bf16 network): image and weights are rounded to bf16 on the fly and a tile is
    //               ONE v_mfma_f32_16x16x32_bf16, lane quarter kq supplies the eight taps C3_TAP_BF16[kq][..]
    //               (five of the 32 slots are zero pads: weight 0, a valid LDS address).
    constexpr int KS = OUT_BF16 ? 8 : 7;
    // (bf16: a tap, or >= 27 for a pad; its LDS offset comes from the table)
    auto k_of = [&](int ks) { return OUT_BF16 ? (C3_TAP_BF16[kq][ks] >= 0 ? (int)C3_TAP_BF16[kq][ks] : 28) : 4 * ks + kq; };
    float wreg[NT][KS];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k = k_of(ks);
            if (FOLD) wreg[j][ks] = k < 27 ? w[(16 * j + i16) * 27 + k] * scale[16 * j + i16] : (k == 27 ? shift[16 * j + i16] : 0.f);
            else wreg[j][ks] = k < 27 ? w[(16 * j + i16) * 27 + k] : 0.f;
        }
    bf16x8 wb[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) wb[j][e] = (__bf16)(OUT_BF16 ? wreg[j][e < KS ? e : 0] : 0.f);
    // A-operand LDS offset of this lane for every k step, relative to the pixel's top-left tap
    int koff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = k_of(ks);
        const int kh = k / 9;
        koff[ks] = k < 27 ? kh * C3_ROW + (k - 9 * kh) : (OUT_BF16 ? -1 - (int)C3_TAP_BF16[kq][ks] : 0);
    }
    float sc[NT], sh[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        sc[j] = scale[16 * j + i16];
        sh[j] = shift[16 * j + i16];
    }

    // Halo staging by LDS-DMA (buffer_load_dwordx4 ... lds).  Halo row rr = image row y0 - 1 + rr, floats 3*x0 - 4 ...
    // 3*x0 + 195, is ONE wave instruction: lane c < 50 moves float4 c straight to tile[buf][rr * C3_ROW + 4c]; wave w takes
    // rows w, w + 4, ...  The loads go through a buffer resource that spans ONE image, so rows above and below the image
    // fall outside it and land as zeros; only the left / right border needs a test.  No staging registers, no ds_write,
    // and - the point - no compiler-placed wait: with register staging hipcc put `s_waitcnt vmcnt(3..0)` in front of the
    // four ds_writes at the end of a patch, which on this hardware also drains the 16 epilogue stores issued since the
    // loads (stores count in vmcnt), one exposed HBM write latency per patch.  Here a wave waits `vmcnt(stores per patch)`:
    // its DMAs, which are older than those stores, have landed; the stores stay in flight across the barrier.
    const int row_f = 3 * W;                                 // floats per image row
    const int c_g = lane < C3_V4 ? 4 * lane - 4 : -0x40000000;      // float index in the image row for x0 = 0 (a dead lane never passes the border test)
    typedef __attribute__((address_space(3))) void *lds_ptr;
    auto decode = [&](int p, int &n, int &y0, int &x0) {
        const unsigned r = fdiv_u((unsigned)p, d_tx), tx = (unsigned)p - r * (unsigned)tiles_x;
        const unsigned nn = fdiv_u(r, d_ty), ty = r - nn * (unsigned)tiles_y;
        n = (int)nn;
        y0 = (int)ty * C3_TH;
        x0 = (int)tx * C3_TW;
    };
    constexpr unsigned kOOB = 0xFFFFFFFFu;
    auto fetch = [&](int p, int buf) {
        p = __builtin_amdgcn_readfirstlane(p);               // block-uniform: keeps the resource words in SGPRs
        int n, y0, x0;
        decode(p, n, y0, x0);
        n = __builtin_amdgcn_readfirstlane(n); y0 = __builtin_amdgcn_readfirstlane(y0); x0 = __builtin_amdgcn_readfirstlane(x0);
        const bool live = p < patches;
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(x) + (live ? (long long)n * H * row_f : 0ll), 0, live && !C3_DBG(2) ? H * row_f * 4 : 0, 0x00020000);
        const unsigned patch_off = (unsigned)((y0 - 1) * row_f + 3 * x0 + c_g) * 4u;   // wraps for y0 = 0: outside the resource
        const bool in_row = (unsigned)(c_g + 3 * x0) < (unsigned)row_f;
        if (lane < C3_V4) {
#pragma unroll
            for (int k = 0; k < (C3_TH + 2 + 3) / 4; ++k) {
                const int rr = __builtin_amdgcn_readfirstlane(wave) + 4 * k;
                if (rr < C3_TH + 2)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_ptr)&tile[buf][rr * C3_ROW], 16,
                                                             (int)(in_row ? patch_off + (unsigned)(rr * row_f * 4) : kOOB), 0, 0, 0);
            }
        }
    };
    // stores a wave issues per patch (they follow the patch's DMAs in program order); vmcnt holds at most 63
    constexpr int kStoresPerPatch = 16 * NT * (POOL ? 1 : 4) < 63 ? 16 * NT * (POOL ? 1 : 4) : 63;
    auto wait_halo = [] { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kStoresPerPatch) : "memory"); };
    auto wg_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // pixel of this lane inside an M-tile: window (i16>>2) of 4, position q = i16&3 in the 2x2 window
    const int q = i16 & 3;
    const int lane_px = 2 * (i16 >> 2) + (q & 1), lane_py = q >> 1;
    // LDS float index of this lane's A element for k step ks in M-tile (pr = 0, wcol = 0) of buffer 0; halo float 0
    // of a row is image float 3*x0 - 4, so pixel x0 - 1 starts at 1.  Every other tile is a compile-time offset.
    int a_addr[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
        a_addr[ks] = (FOLD && k_of(ks) == 27) ? C3_TILE : (4 * wave + lane_py) * C3_ROW + lane_px * 3 + 1 + koff[ks];
    const float *lds = &tile[0][0];

    // Output: lane holds channel 16j + i16 of pool window kq (POOL) / of the 2x2 pixels of window kq, for every
    // M-tile t = (pr, wc): pooled row 2*wave + pr, windows 4*wc + kq.  Stores go through a buffer resource based at
    // the patch's first output element, so a store is one 32-bit offset: lane part + compile-time tile part.
    const int Ho = POOL ? H >> 1 : H, Wo = POOL ? W >> 1 : W;
    constexpr unsigned OSZ = OUT_BF16 ? 2u : 4u;            // bytes per output element (bf16 feeds the bf16 network)
    const unsigned row_b = (unsigned)Wo * Cout * OSZ, px_b = (unsigned)Cout * OSZ;
    const unsigned lane_off = POOL ? (unsigned)(2 * wave) * row_b + (unsigned)kq * px_b + (unsigned)i16 * OSZ
                                   : (unsigned)(4 * wave) * row_b + (unsigned)(2 * kq) * px_b + (unsigned)i16 * OSZ;
    const long long y_total = (long long)NB * Ho * Wo * Cout;

    // developer stamps (compiled in with -DTY1_DEV_STAMPS only: the fences change the schedule)
#ifdef TY1_DEV_STAMPS
#define C3_STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define C3_STAMP(var) do { } while (0)
#endif
    long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, acc_fetch = 0, acc_mma = 0, acc_store = 0, acc_bar = 0, n_done = 0;
    (void)ts0; (void)ts1; (void)ts2; (void)ts3; (void)ts4;
    if (stamp && tid == 0) {
        stamp[(long long)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memrealtime();
        stamp[(long long)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        stamp[(long long)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }

    const bool act_max = alpha > 0.f && alpha <= 1.f;
    if (FOLD)
        for (int i = tid; i < C3_ONES; i += 256) tile[0][C3_TILE + i] = tile[1][C3_TILE + i] = 1.f;
    int p = blockIdx.x;
    fetch(p, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    int buf = 0;
    for (; p < patches; p += gridDim.x) {
        C3_STAMP(ts0);
        fetch(p + gridDim.x, buf ^ 1);                       // lands during the MFMAs (zeros past the end); that buffer was last read before the previous barrier
        C3_STAMP(ts1);
        int n, y0, x0;
        decode(p, n, y0, x0);
        n = __builtin_amdgcn_readfirstlane(n); y0 = __builtin_amdgcn_readfirstlane(y0); x0 = __builtin_amdgcn_readfirstlane(x0);
        const int oy0 = POOL ? y0 >> 1 : y0, ox0 = POOL ? x0 >> 1 : x0;
        const long long o_org = (((long long)n * Ho + oy0) * Wo + ox0) * Cout;
        const long long o_left = (y_total - o_org) * OSZ;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            static_cast<char *>(y) + o_org * OSZ, 0, C3_DBG(1) ? 0 : (int)(o_left > 0xFFFFFFFFll ? 0xFFFFFFFFll : o_left), 0x00020000);
        auto put = [&](float v, unsigned off, int soff) {
            if constexpr (OUT_BF16) __builtin_amdgcn_raw_buffer_store_b16(bf16_bits(v), o_rsrc, (int)off, soff, 0);
            else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), o_rsrc, (int)off, soff, 0);
        };
        // a patch that lies wholly inside the image needs no per-store test: the tile part of the offset is a scalar
        // (the store's scalar offset), so a store costs no vector instruction besides itself - every one of those
        // costs the matrix pipe ~8 cycles, and this layer has one store per 7 MFMAs
        const bool full = y0 + C3_TH <= H && x0 + C3_TW <= W;
        // output rows / columns of the image left from this lane's first element (a tile is valid below them)
        const int rows_left = Ho - oy0 - (POOL ? 2 : 4) * wave;
        const int cols_left = Wo - ox0 - (POOL ? 1 : 2) * kq;
        int ab[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ab[ks] = a_addr[ks] + buf * C3_BUF;

        // Two M-tiles (independent MFMA chains) per step; the BN / LeakyReLU / pool / store of a step is written
        // after the MFMAs of the NEXT step so that the scheduler can run it under them.
        // A operands of a step (7 k-steps x 2 tiles) are read from LDS one whole step ahead of their MFMAs
        auto load_a = [&](int t2, float (&av)[KS][2]) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = t2 + u, pr = t >> 3, wc = t & 7;
#if defined(C3_KO) && (C3_KO & 2)
                    av[ks][u] = __int_as_float(ab[ks] + t);
#else
                    av[ks][u] = lds[ab[ks] + 2 * pr * C3_ROW + 24 * wc];   // window 4*wc -> pixel 8*wc -> float 24*wc
#endif
                }
        };
        auto compute = [&](const float (&av)[KS][2], f32x4 (&acc)[2][NT]) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[u][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (OUT_BF16) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    bf16x8 ab8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) ab8[e] = (__bf16)av[e < KS ? e : 0][u];
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[u][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab8, wb[j], acc[u][j], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks][0], wreg[j][ks], acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks][1], wreg[j][ks], acc[1][j], 0, 0, 0);
                    }
            }
        };
        auto finish = [&](auto full_c, auto act_c, int t2, const f32x4 (&acc)[2][NT]) {
            constexpr bool FULL = decltype(full_c)::value, ACT_MAX = decltype(act_c)::value;
#if defined(C3_KO) && (C3_KO & 1)
            for (int u = 0; u < 2; ++u) for (int j = 0; j < NT; ++j) asm volatile("" ::"v"(acc[u][j]));
            return;
#endif
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = t2 + u, pr = t >> 3, wc = t & 7;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float z[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) z[r] = FOLD ? acc[u][j][r] : acc[u][j][r] * sc[j] + sh[j];
                    if (POOL) {
                        // LeakyReLU with alpha >= 0 is non-decreasing: pool first, activate once (same bits)
                        // (v_max3 + v_max; for 0 < alpha <= 1 LeakyReLU(m) is max(m, alpha * m), bit for bit incl. -0, inf and
                        //  NaN: two VALU ops instead of three - every one of them is paid for in matrix-pipe cycles)
                        const float m = fmaxf(fmaxf(fmaxf(z[0], z[1]), z[2]), z[3]);
                        const float v = ACT_MAX ? fmaxf(m, m * alpha) : (m > 0.f ? m : m * alpha);
                        const int tile_off = pr * (int)row_b + 4 * wc * (int)px_b + 16 * (int)OSZ * j;
                        if constexpr (FULL) {
                            put(v, lane_off, tile_off);
                        } else {
                            const bool ok = pr < rows_left && 4 * wc < cols_left;
                            put(v, ok ? lane_off + (unsigned)tile_off : kOOB, 0);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float v = z[r] > 0.f ? z[r] : z[r] * alpha;
                            const int dy = 2 * pr + (r >> 1), dx = 8 * wc + (r & 1);
                            const int tile_off = dy * (int)row_b + dx * (int)px_b + 16 * (int)OSZ * j;
                            if constexpr (FULL) {
                                put(v, lane_off, tile_off);
                            } else {
                                const bool ok = dy < rows_left && dx < cols_left;
                                put(v, ok ? lane_off + (unsigned)tile_off : kOOB, 0);
                            }
                        }
                    }
                }
            }
        };
        auto patch_body = [&](auto full_c, auto act_c) {
            f32x4 accA[2][NT], accB[2][NT];
            float avA[KS][2], avB[KS][2];
            load_a(0, avA);
            load_a(2, avB);
            compute(avA, accA);
            // the fences pin "reads of the step after next | MFMAs of the next step + epilogue of this one" as stages
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                __builtin_amdgcn_sched_barrier(0);
                if (it < 3) load_a(4 * it + 4, avA);
                __builtin_amdgcn_sched_barrier(0);
                compute(avB, accB);
                finish(full_c, act_c, 4 * it, accA);
                __builtin_amdgcn_sched_barrier(0);
                if (it < 3) load_a(4 * it + 6, avB);
                __builtin_amdgcn_sched_barrier(0);
                if (it < 3) compute(avA, accA);
                finish(full_c, act_c, 4 * it + 2, accB);
            }
        };
        if (POOL && act_max) {
            if (full) patch_body(std::true_type{}, std::true_type{});
            else patch_body(std::false_type{}, std::true_type{});
        } else {
            if (full) patch_body(std::true_type{}, std::false_type{});
            else patch_body(std::false_type{}, std::false_type{});
        }
        C3_STAMP(ts2);
        wait_halo();
        C3_STAMP(ts3);
        wg_barrier();
        C3_STAMP(ts4);
        buf ^= 1;
        acc_fetch += ts1 - ts0; acc_mma += ts2 - ts1; acc_store += ts3 - ts2; acc_bar += ts4 - ts3; ++n_done;
    }
#undef C3_STAMP
#undef C3_DBG
    if (stamp && tid == 0) {
        stamp[(long long)blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
        stamp[(long long)blockIdx.x * 8 + 1] = acc_mma;
        stamp[(long long)blockIdx.x * 8 + 2] = acc_fetch;
        stamp[(long long)blockIdx.x * 8 + 6] = acc_store + (acc_bar << 32);
        stamp[(long long)blockIdx.x * 8 + 7] = n_done;
    }
}

}  // namespace

namespace ty1 {

namespace {
// workgroups of `kernel` (256 threads, static LDS) that one CU keeps resident, at most `cap`; `env` is a developer override
template <typename K>
int resident_blocks(K kernel, int cap, const char *env) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, 0) != hipSuccess || occ < 1) occ = 1;
    if (const char *e = ty1::dev_env(env)) occ = atoi(e) > 0 ? atoi(e) : occ;
    else if (occ > cap) occ = cap;
    if (ty1::dev_env("TY1_VERBOSE")) fprintf(stderr, "[ty1] %s: %d resident workgroups per CU\n", env, occ);
    return occ;
}
}  // namespace

// Cin == 16, Cout == 32 or 64.  Returns 0 on success (error text via ty1::fail).
int conv3x3_c16_direct(const float *x, const float *w, const float *scale, const float *shift, float *y, int N, int H,
                       int W, int Cout, int pool, float alpha, hipStream_t s) {
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const long long patches = (long long)N * tiles_y * tiles_x;
    TY1_REQUIRE(patches < (1ll << 31), "ty1_conv3x3(c16): too many patches");
    const double px = (double)N * H * W;
    ProfScope prof(pool ? "conv3x3_c16_pool" : "conv3x3_c16", s, 2.0 * px * Cout * 144,
                   4.0 * (px * 16 + 144.0 * Cout + px * Cout / (pool ? 4 : 1)));
    const int P = (int)patches;
    // persistent grid: the workgroups that stay resident (59 KB of LDS each: two per CU when the registers allow)
#define TY1_C16(PL, NTV)                                                                                              \
    do {                                                                                                              \
        static const int per_cu = resident_blocks(conv3x3_c16_kernel<PL, NTV>, 2, "TY1_C16_BLOCKS");                  \
        const long long want = (long long)per_cu * ty1::grid_cus();                                                        \
        conv3x3_c16_kernel<PL, NTV><<<dim3((unsigned)(patches < want ? patches : want)), dim3(256), 0, s>>>(          \
            x, w, scale, shift, y, N, H, W, Cout, alpha, tiles_x, tiles_y, P, make_fastdiv_u((unsigned)tiles_x),      \
            make_fastdiv_u((unsigned)tiles_y));                                                                       \
    } while (0)
    if (Cout == 32) {
        if (pool) TY1_C16(true, 1); else TY1_C16(false, 1);
    } else {
        if (pool) TY1_C16(true, 2); else TY1_C16(false, 2);
    }
#undef TY1_C16
    return check_launch("ty1_conv3x3(c16)");
}


// Cin == 16 in bf16 (x, w, y bf16; fp32 accumulate), Cout == 32 or 64.
int conv3x3_c16_direct_bf16(const void *x, const void *w, const float *scale, const float *shift, void *y, int N, int H,
                            int W, int Cout, int pool, float alpha, hipStream_t s) {
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const long long patches = (long long)N * tiles_y * tiles_x;
    TY1_REQUIRE(patches < (1ll << 31), "ty1_conv3x3(c16 bf16): too many patches");
    const double px = (double)N * H * W;
    ProfScope prof(pool ? "conv3x3_c16_bf16_pool" : "conv3x3_c16_bf16", s, 2.0 * px * Cout * 144,
                   2.0 * (px * 16 + 144.0 * Cout + px * Cout / (pool ? 4 : 1)));
    const int P = (int)patches;
    const unsigned short *xb = static_cast<const unsigned short *>(x), *wb = static_cast<const unsigned short *>(w);
    unsigned short *yb = static_cast<unsigned short *>(y);
#define TY1_C16B(PL, NTV)                                                                                             \
    do {                                                                                                              \
        static const int per_cu = resident_blocks(conv3x3_c16_bf16_kernel<PL, NTV>, 4, "TY1_C16B_BLOCKS");            \
        const long long want = (long long)per_cu * ty1::grid_cus();                                                        \
        conv3x3_c16_bf16_kernel<PL, NTV><<<dim3((unsigned)(patches < want ? patches : want)), dim3(256), 0, s>>>(     \
            xb, wb, scale, shift, yb, N, H, W, Cout, alpha, tiles_x, tiles_y, P, make_fastdiv_u((unsigned)tiles_x),  \
            make_fastdiv_u((unsigned)tiles_y));                                                                       \
    } while (0)
    if (Cout == 32) {
        if (pool) TY1_C16B(true, 1); else TY1_C16B(false, 1);
    } else {
        if (pool) TY1_C16B(true, 2); else TY1_C16B(false, 2);
    }
#undef TY1_C16B
    return check_launch("ty1_conv3x3(c16 bf16)");
}

// Cin == 3, Cout == 16 * NT (NT <= 4), W % 4 == 0, x 16-byte aligned.
// out_bf16: y is bf16 (the first layer of the bf16 network; the layer itself stays exact f32).
int conv3x3_c3_persistent(const float *x, const float *w, const float *scale, const float *shift, void *y, int N, int H,
                          int W, int Cout, int pool, float alpha, int out_bf16, hipStream_t s) {
    const int tiles_x = (W + C3_TW - 1) / C3_TW, tiles_y = (H + C3_TH - 1) / C3_TH;
    const long long patches = (long long)N * tiles_y * tiles_x;
    TY1_REQUIRE(patches < (1ll << 31), "ty1_conv3x3(c3): too many patches");
    TY1_REQUIRE((long long)H * W * 12 < (1ll << 31), "ty1_conv3x3(c3): one image must stay below 2 GiB");
    const int P = (int)patches, nt = Cout / 16;
    const char *dbg_env = ty1::dev_env("TY1_C3_DBG");
    const int dbg = dbg_env ? atoi(dbg_env) : 0;
    // persistent grid: as many workgroups as stay resident (35 KB of LDS, <= 128 registers: four per CU)
#define TY1_C3P(PL, NTV, OB)                                                                                          \
    do {                                                                                                              \
        static const int per_cu = resident_blocks(conv3x3_c3p_kernel<PL, NTV, OB>, 4, "TY1_C3_BLOCKS");               \
        const long long want = (long long)per_cu * ty1::grid_cus();                                                        \
        const long long nblk = patches < want ? patches : want;                                                       \
        conv3x3_c3p_kernel<PL, NTV, OB><<<dim3((unsigned)nblk), dim3(256), 0, s>>>(                                   \
            x, w, scale, shift, y, N, H, W, Cout, alpha, tiles_x, tiles_y, P, nblk <= g_stamp_cap ? g_stamp : nullptr, dbg, \
            make_fastdiv_u((unsigned)tiles_x), make_fastdiv_u((unsigned)tiles_y));                                   \
    } while (0)
#define TY1_C3P_NT(PL, OB)                                                                                            \
    do {                                                                                                              \
        if (nt == 1) TY1_C3P(PL, 1, OB); else if (nt == 2) TY1_C3P(PL, 2, OB);                                        \
        else if (nt == 3) TY1_C3P(PL, 3, OB); else TY1_C3P(PL, 4, OB);                                                \
    } while (0)
    if (out_bf16) {
        if (pool) TY1_C3P_NT(true, true); else TY1_C3P_NT(false, true);
    } else {
        if (pool) TY1_C3P_NT(true, false); else TY1_C3P_NT(false, false);
    }
#undef TY1_C3P_NT
#undef TY1_C3P
    return check_launch("ty1_conv3x3(c3)");
}

}  // namespace ty1
