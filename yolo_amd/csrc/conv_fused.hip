// The first two blocks of the bf16 network in ONE kernel:
//   conv 3x3 (3 -> 16) + BN + LeakyReLU + 2x2 max-pool  ->  conv 3x3 (16 -> 32) + BN + LeakyReLU + 2x2 max-pool
// Replaces /root/reference/models/model_tiny_yolov1.py:43-53 (convolution_0 and convolution_1 blocks) for the bf16
// configuration.  The (H/2) x (W/2) x 16 tensor between the two blocks (205 MB written and read back at 128 images, the two
// layers each within 1.3x of their own HBM floor) never reaches HBM: it lives in LDS.  fp32 image in, bf16 out; operands
// bf16, accumulation / BN / LeakyReLU / pool in fp32 - the rounding points of the two separate kernels.
//
// A workgroup (4 waves) walks patches of 16 x 32 block-1 pixels (= 8 x 16 output pixels x 32 channels).  Per patch:
//   stage   38 input rows x 216 floats (the patch's 3-pixel halo) are fetched into registers while the PREVIOUS patch
//           computes, then rounded to bf16 and written to LDS (`in`: 432 bytes per row).
//   block 0 the 18 x 34 pooled block-0 pixels the patch needs (1-pixel halo) x 16 channels, into LDS (`mid`: 32-byte pixels).
//           Cin = 3 gives a lane nothing contiguous to fetch per tap, so the GEMM is turned round: the WEIGHTS are the
//           expanded operand.  D[m][n] = sum_k T[m][k] * In[k][n] on v_mfma_f32_32x32x16_bf16 with
//             n = one pooled pixel of a pair (pooled row Y, pixel pair b): 32 of them per tile, 10 tiles per patch,
//             k = 16 CONTIGUOUS bf16 of one input row - a window that starts 8 * phase values into the 24 values (8 pixels)
//                 around pixel pair b - for each of the four input rows 2Y-1 .. 2Y+2: four MFMAs, each B fragment one
//                 16-byte LDS read (8-byte aligned), no per-tap gather, no conversion per use,
//             m = (dy, dx, channel c8) of the 2 x 2 pool window x 8 channels; T is the 3x3x3 filter laid out Toeplitz-
//                 fashion over the window (zero where a tap falls outside it), 16 VGPRs per wave for the whole kernel.
//           Wave w owns phase = w & 1 (which pixel of the pair) and channel half w >> 1 for all 10 tiles.  The C/D map
//           (row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) puts the four window positions of a channel in registers
//           r, r+4, r+8, r+12 of ONE lane: the pool is v_max3 + v_max, no lane exchange.
//           sign(BN scale) is folded into the bf16 weights (exact), so max-pool -> |scale| * m + shift -> LeakyReLU gives
//           the bits of BN -> LeakyReLU -> pool with a third of the vector work.
//           Pooled pixels outside the image are written as zeros: they are block 1's padding.
//   block 1 conv3x3_c16_bf16_kernel's scheme on `mid`: a tile is 2 rows x 16 pixels in pool-window order, nine MFMAs
//           (one per tap, K = 16 channels), weights of a lane in 36 VGPRs; wave w owns row pairs w and w + 4.
// MFMA cycles per patch and wave: 40 x 32 (block 0, 2.4x its useful work: the Toeplitz zeros) + 36 x 32 (block 1).
#include "common.h"

#include <cmath>

namespace {

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
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int F_PH = 16, F_PW = 32;                 // block-1 (pre-pool) pixels per patch
constexpr int F_RY = F_PH + 2, F_RX = F_PW + 2;     // pooled block-0 region of a patch: 18 x 34
constexpr int F_NB = F_RX / 2;                      // 17 pixel pairs per region row
constexpr int F_NPIX = F_RY * F_NB;                 // 306 columns of the block-0 GEMM
constexpr int F_NT = (F_NPIX + 31) / 32;            // 10 tiles
constexpr int F_INROWS = 2 * F_RY + 2;              // 38 input rows
constexpr int F_INV = 12 * (F_NB - 1) + 24;         // 216 values per staged input row
constexpr int F_INROWB = F_INV * 2;                 // bytes per staged row (bf16)
constexpr int F_CHUNKS = F_INV / 4;                 // 54 float4 per row
constexpr int F_STAGE = (F_INROWS * F_CHUNKS + 255) / 256;      // 9 float4 per thread
constexpr int F_PPX = 32;                           // bytes per `mid` pixel (16 bf16)
constexpr int F_PROWB = F_RX * F_PPX + 16;          // bytes per `mid` row: an odd number of 16-byte slots
constexpr int F_IN_BYTES = F_INROWS * F_INROWB, F_MID_BYTES = F_RY * F_PROWB;

struct FusedArgs {
    const float *x;                 // (N, H0, W0, 3) fp32
    const float *w0;                // (16, 3, 3, 3) fp32 OHWI, rounded to bf16 here
    const float *scale0, *shift0;   // folded BN of block 0 (16)
    const unsigned short *w1;       // (32, 3, 3, 16) bf16 OHWI
    const float *scale1, *shift1;   // folded BN of block 1 (32)
    unsigned short *y;              // (N, H0/4, W0/4, 32) bf16
    int NB, H0, W0;
    float alpha;
    int tiles_x, tiles_y, patches;
    FastDivU d_tx, d_ty;
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {          // v_cvt_pk_bf16_f32, round to nearest even
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned short bf16_bits(float v) {
    const __bf16 h = (__bf16)v;
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

__global__ __launch_bounds__(256, 2) void conv01_fused_bf16_kernel(const FusedArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_in[F_IN_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char lds_mid[F_MID_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int H1 = a.H0 >> 1, W1 = a.W0 >> 1, H2 = a.H0 >> 2, W2 = a.W0 >> 2;
    constexpr unsigned kOOB = 0xFFFFFFFFu;

    // ------------------------------------------------------------------ block 0: this wave's slice of the Toeplitz operand
    const int ph = wave & 1, hf = wave >> 1;
    bf16x8 tz[4];                                    // A fragments: row m = l31, k = 8h + j of input row rho
    {
        const int pos = l31 >> 3, dy = pos >> 1, dx = pos & 1, c = 8 * hf + (l31 & 7);
        const float sgn = a.scale0[c] < 0.f ? -1.f : 1.f;
#pragma unroll
        for (int rho = 0; rho < 4; ++rho)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int u = 8 * ph + 8 * h + j;              // value index in the pair's 24-value neighbourhood
                const int px = u / 3, ci = u - 3 * px;         // input pixel (0 = two left of the pair's first pre-pool pixel)
                const int kw = px - (2 + 2 * ph + dx) + 1, kh = rho - dy;
                float v = 0.f;
                if (kw >= 0 && kw < 3 && kh >= 0 && kh < 3) v = a.w0[(c * 9 + kh * 3 + kw) * 3 + ci];
                tz[rho][j] = (__bf16)(bf16_to_f32(bf16_bits(v)) * sgn);
            }
    }
    float asc0[4], sh0[4];                           // |scale|, shift of channels 8 hf + 4 h + r4
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        asc0[r4] = fabsf(a.scale0[8 * hf + 4 * h + r4]);
        sh0[r4] = a.shift0[8 * hf + 4 * h + r4];
    }
    // per tile: LDS address of this lane's B fragment / of its output pixel in `mid`, and (Y, X) of that pixel in the region
    int in_addr[F_NT], mid_addr[F_NT], yx[F_NT];
#pragma unroll
    for (int t = 0; t < F_NT; ++t) {
        const int n = 32 * t + l31, nn = n < F_NPIX ? n : F_NPIX - 1;
        const int yr = nn / F_NB, b = nn - yr * F_NB, xr = 2 * b + ph;
        in_addr[t] = 2 * yr * F_INROWB + 24 * b + 16 * ph + 16 * h;
        mid_addr[t] = n < F_NPIX ? yr * F_PROWB + xr * F_PPX + (8 * hf + 4 * h) * 2 : -1;
        yx[t] = yr | (xr << 8);
    }

    // ------------------------------------------------------------------ block 1: weights of this lane (sign of the BN scale folded in)
    bf16x8 w1r[9];
    float asc1, sh1;
    {
        const float s = a.scale1[l31];
        const unsigned short flip = s < 0.f ? 0x8000u : 0u;
        asc1 = fabsf(s);
        sh1 = a.shift1[l31];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            u32x4 raw = *reinterpret_cast<const u32x4 *>(a.w1 + ((size_t)l31 * 9 + t) * 16 + 8 * h);
            const unsigned f2 = (unsigned)flip | ((unsigned)flip << 16);
            raw ^= (u32x4){f2, f2, f2, f2};
            w1r[t] = __builtin_bit_cast(bf16x8, raw);
        }
    }
    // A fragment of tile (row pair rp, column half chf): pixel l31 in pool-window order
    const int q = l31 & 3, wdw = l31 >> 2;
    const int px_off = (q >> 1) * F_PROWB + (2 * wdw + (q & 1)) * F_PPX + 16 * h;

    // ------------------------------------------------------------------ input staging
    u32x4 stage[F_STAGE];
    int st_lds[F_STAGE], st_col[F_STAGE];
    unsigned st_off[F_STAGE];
#pragma unroll
    for (int k = 0; k < F_STAGE; ++k) {
        const int i = tid + 256 * k, row = i / F_CHUNKS, ch = i - row * F_CHUNKS;
        const bool live = i < F_INROWS * F_CHUNKS;
        st_lds[k] = live ? row * F_INROWB + ch * 8 : -1;
        st_col[k] = live ? 4 * ch : -0x40000000;                       // a dead piece never passes the column test
        st_off[k] = (unsigned)((row * a.W0 * 3 + 4 * ch) * 4);
    }
    auto decode = [&](int p, int &n, int &y1, int &x1) {
        p = __builtin_amdgcn_readfirstlane(p);
        const unsigned r = fdiv_u((unsigned)p, a.d_tx), tx = (unsigned)p - r * (unsigned)a.tiles_x;
        const unsigned nn = fdiv_u(r, a.d_ty), ty = r - nn * (unsigned)a.tiles_y;
        n = __builtin_amdgcn_readfirstlane((int)nn);
        y1 = __builtin_amdgcn_readfirstlane((int)ty * F_PH);
        x1 = __builtin_amdgcn_readfirstlane((int)tx * F_PW);
    };
    const int row_f = 3 * a.W0;
    auto fetch = [&](int p) {                        // rows 2 y1 - 3 ..., floats 6 x1 - 12 ... of image n; zeros outside it
        const bool live = p < a.patches;
        int n, y1, x1;
        decode(live ? p : 0, n, y1, x1);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(a.x) + (long long)n * a.H0 * row_f, 0, live ? a.H0 * row_f * 4 : 0, 0x00020000);
        const int c0 = 6 * x1 - 12;
        const unsigned org = (unsigned)(((2 * y1 - 3) * row_f + c0) * 4);      // wraps above the image: outside the resource
#pragma unroll
        for (int k = 0; k < F_STAGE; ++k) {
            const bool in_row = (unsigned)(st_col[k] + c0) < (unsigned)row_f;
            stage[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(in_row ? st_off[k] + org : kOOB), 0, 0);
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int k = 0; k < F_STAGE; ++k)
            if (st_lds[k] >= 0) {
                const u32x2 v = {pack_bf16(__uint_as_float(stage[k][0]), __uint_as_float(stage[k][1])),
                                 pack_bf16(__uint_as_float(stage[k][2]), __uint_as_float(stage[k][3]))};
                *reinterpret_cast<u32x2 *>(&lds_in[st_lds[k]]) = v;
            }
    };

    const float alpha = a.alpha;
    int p = blockIdx.x;
    fetch(p);
    store_stage();
    __syncthreads();
    for (; p < a.patches; p += gridDim.x) {
        fetch(p + gridDim.x);                        // lands while this patch computes
        int n, y1, x1;
        decode(p, n, y1, x1);

        // ---------------------------------------------------------------- block 0 -> mid
        {
            // region pixel (yr, xr) is pooled pixel (y1 - 1 + yr, x1 - 1 + xr): outside the image it is block 1's zero padding
            const bool interior = y1 > 0 && y1 + F_PH < H1 && x1 > 0 && x1 + F_PW < W1;
            auto frags = [&](int t, u32x4 (&bf)[4]) {
#pragma unroll
                for (int rho = 0; rho < 4; ++rho) {
                    const unsigned char *src = &lds_in[in_addr[t] + rho * F_INROWB];
                    const u32x2 lo = *reinterpret_cast<const u32x2 *>(__builtin_assume_aligned(src, 8));
                    const u32x2 hi = *reinterpret_cast<const u32x2 *>(__builtin_assume_aligned(src + 8, 8));
                    bf[rho] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
                }
            };
            u32x4 bcur[4], bnxt[4];
            frags(0, bcur);
#pragma unroll
            for (int t = 0; t < F_NT; ++t) {
                if (t + 1 < F_NT) frags(t + 1, bnxt);
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int rho = 0; rho < 4; ++rho)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tz[rho], __builtin_bit_cast(bf16x8, bcur[rho]), acc, 0, 0, 0);
                float v[4];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float m = fmaxf(fmaxf(fmaxf(acc[r4], acc[4 + r4]), acc[8 + r4]), acc[12 + r4]);
                    const float z = m * asc0[r4] + sh0[r4];
                    v[r4] = fmaxf(z, z * alpha);
                }
                u32x2 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
                if (!interior) {
                    const int yy = y1 - 1 + (yx[t] & 0xFF), xx = x1 - 1 + (yx[t] >> 8);
                    const bool inside = (unsigned)yy < (unsigned)H1 && (unsigned)xx < (unsigned)W1;
                    o[0] = inside ? o[0] : 0u;
                    o[1] = inside ? o[1] : 0u;
                }
                if (mid_addr[t] >= 0) *reinterpret_cast<u32x2 *>(&lds_mid[mid_addr[t]]) = o;
                if (t + 1 < F_NT) {
#pragma unroll
                    for (int rho = 0; rho < 4; ++rho) bcur[rho] = bnxt[rho];
                }
            }
        }
        __syncthreads();

        // ---------------------------------------------------------------- block 1: mid -> y
        {
            const int oy0 = y1 >> 1, ox0 = x1 >> 1;
            const long long o_org = (((long long)n * H2 + oy0) * W2 + ox0) * 32;
            const long long o_left = ((long long)a.NB * H2 * W2 * 32 - o_org) * 2;
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                a.y + o_org, 0, (int)(o_left > 0xFFFFFFFFll ? 0xFFFFFFFFll : o_left), 0x00020000);
            const unsigned row_b = (unsigned)W2 * 64u;
            const bool full = y1 + F_PH <= H1 && x1 + F_PW <= W1;
            const int rows_left = H2 - oy0, cols_left = W2 - ox0;
#pragma unroll
            for (int rpi = 0; rpi < 2; ++rpi) {
                const int rp = wave + 4 * rpi;                  // pixel rows 2 rp, 2 rp + 1 of the patch = output row rp
                f32x16 acc[2];
#pragma unroll
                for (int chf = 0; chf < 2; ++chf)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[chf][r] = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int toff = (t / 3) * F_PROWB + (t % 3) * F_PPX;        // mid origin is pixel (y1 - 1, x1 - 1)
#pragma unroll
                    for (int chf = 0; chf < 2; ++chf) {
                        const bf16x8 af = *reinterpret_cast<const bf16x8 *>(
                            __builtin_assume_aligned(&lds_mid[2 * rp * F_PROWB + 16 * chf * F_PPX + px_off + toff], 16));
                        acc[chf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, w1r[t], acc[chf], 0, 0, 0);
                    }
                }
                // lane holds channel l31 of windows 2 g + h (g = r >> 2) of each column half; r & 3 = position in the window
#pragma unroll
                for (int chf = 0; chf < 2; ++chf)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float m = fmaxf(fmaxf(fmaxf(acc[chf][4 * g], acc[chf][4 * g + 1]), acc[chf][4 * g + 2]), acc[chf][4 * g + 3]);
                        const float z = m * asc1 + sh1;
                        const float v = fmaxf(z, z * alpha);
                        const int oc = 8 * chf + 2 * g + h;     // output column inside the patch
                        const bool ok = full || (rp < rows_left && oc < cols_left);
                        const unsigned off = ok ? (unsigned)rp * row_b + (unsigned)oc * 64u + (unsigned)l31 * 2u : kOOB;
                        __builtin_amdgcn_raw_buffer_store_b16(bf16_bits(v), o_rsrc, (int)off, 0, 0);
                    }
            }
        }
        store_stage();                               // `in` was last read before the barrier above
        __syncthreads();
    }
}

}  // namespace

namespace ty1 {

// Geometry the fused kernel covers: H, W multiples of 4 (two pools), one image below 2 GiB; Cout 16 and 32 are fixed.
int conv01_fused_bf16(const float *x, const float *w0, const float *scale0, const float *shift0, const void *w1, const float *scale1,
                      const float *shift1, void *y, int N, int H, int W, float alpha, hipStream_t s) {
    FusedArgs a{};
    a.x = x; a.w0 = w0; a.scale0 = scale0; a.shift0 = shift0;
    a.w1 = static_cast<const unsigned short *>(w1); a.scale1 = scale1; a.shift1 = shift1;
    a.y = static_cast<unsigned short *>(y);
    a.NB = N; a.H0 = H; a.W0 = W; a.alpha = alpha;
    const int H1 = H / 2, W1 = W / 2;
    a.tiles_x = (W1 + F_PW - 1) / F_PW;
    a.tiles_y = (H1 + F_PH - 1) / F_PH;
    const long long patches = (long long)N * a.tiles_x * a.tiles_y;
    TY1_REQUIRE(patches < (1ll << 31), "ty1_conv3x3_x2_c3_bf16: too many patches");
    a.patches = (int)patches;
    a.d_tx = make_fastdiv_u((unsigned)a.tiles_x);
    a.d_ty = make_fastdiv_u((unsigned)a.tiles_y);
    static const int per_cu = [] {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, conv01_fused_bf16_kernel, 256, 0) != hipSuccess || occ < 1) occ = 2;
        if (const char *e = ty1::dev_env("TY1_FUSED_BLOCKS")) occ = atoi(e) > 0 ? atoi(e) : occ;
        return occ > 4 ? 4 : occ;
    }();
    const long long want = (long long)per_cu * ty1::grid_cus();
    const double px0 = (double)N * H * W, px1 = px0 / 4;
    ty1::ProfScope prof("conv01_fused_bf16", s, 2.0 * px0 * 16 * 27 + 2.0 * px1 * 32 * 144,
                        4.0 * px0 * 3 + 2.0 * (px1 / 4) * 32 + 4.0 * 27 * 16 + 2.0 * 144 * 32);
    conv01_fused_bf16_kernel<<<dim3((unsigned)(patches < want ? patches : want)), dim3(256), 0, s>>>(a);
    return check_launch("ty1_conv3x3_x2_c3_bf16");
}

}  // namespace ty1

extern "C" int ty1_conv3x3_x2_c3_bf16(const float *x, const float *w0, const float *scale0, const float *shift0, const void *w1,
                                      const float *scale1, const float *shift1, void *y, int N, int H, int W, float alpha,
                                      ty1_stream_t stream) {
    TY1_REQUIRE(N >= 0 && H > 0 && W > 0, "ty1_conv3x3_x2_c3_bf16: bad dimensions");
    if (N == 0) return 0;
    TY1_REQUIRE(x && w0 && scale0 && shift0 && w1 && scale1 && shift1 && y, "ty1_conv3x3_x2_c3_bf16: null pointer");
    TY1_REQUIRE(H % 4 == 0 && W % 4 == 0, "ty1_conv3x3_x2_c3_bf16: two 2x2 pools need H, W multiples of 4 (got %dx%d)", H, W);
    TY1_REQUIRE(alpha > 0.f && alpha <= 1.f, "ty1_conv3x3_x2_c3_bf16: the fused epilogue needs 0 < alpha <= 1 (got %g)", alpha);
    TY1_REQUIRE((long long)H * W * 12 < (1ll << 31), "ty1_conv3x3_x2_c3_bf16: one image must stay below 2 GiB");
    TY1_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w1 & 15) == 0 && ((uintptr_t)y & 1) == 0,
                "ty1_conv3x3_x2_c3_bf16: x and w1 must be 16-byte aligned");
    return ty1::conv01_fused_bf16(x, w0, scale0, shift0, w1, scale1, shift1, y, N, H, W, alpha, (hipStream_t)stream);
}
