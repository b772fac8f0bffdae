// The first two blocks of the bf16 network in ONE kernel:
//   conv 3x3 (3 -> 16) + BN + LeakyReLU + 2x2 max-pool  ->  conv 3x3 (16 -> 32) + BN + LeakyReLU + 2x2 max-pool
// Replaces /root/reference/models/model_tiny_yolov1.py:43-53 (convolution_0 and convolution_1 blocks) for the bf16
// configuration.  The (H/2) x (W/2) x 16 tensor between the two blocks (205 MB written and read back at 128 images, the two
// layers each within 1.3x of their own HBM floor) never reaches HBM: it lives in LDS.  fp32 image in, bf16 out; operands
// bf16, accumulation / BN / LeakyReLU / pool in fp32 - the rounding points of the two separate kernels.
//
// A workgroup (4 waves) walks patches of 16 x 32 block-1 pixels (= 8 x 16 output pixels x 32 channels).  Per patch:
//   stage   38 input rows x 216 floats (the patch's 3-pixel halo) are fetched into registers while the PREVIOUS patch
//           computes - in two halves of 20 registers, one under each block - then rounded to bf16 and written to LDS
//           (`in`: 432 bytes per row).
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
//           (one per tap, K = 16 channels), weights in LDS (9 KB, one conflict-free 16-byte read per MFMA: in registers
//           they cost 36 VGPRs and with them the third resident workgroup per CU); wave w owns row pairs w and w + 4.
// MFMA cycles per patch and wave: 40 x 32 (block 0, 2.4x its useful work: the Toeplitz zeros) + 36 x 32 (block 1).
#include "common.h"

#include <cmath>
#include <type_traits>

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
// bytes per staged row (bf16): 432 used; two rows = 4 dwords (mod 64 banks), which makes the block-0 fragment reads of a
// tile - 16 region rows x 2 pixel pairs per 32-lane group of a ds_read_b64 - hit 64 different banks
constexpr int F_INROWB = 520;
constexpr int F_CHUNKS = F_INV / 4;                 // 54 float4 per row
constexpr int F_STAGE = (F_INROWS * F_CHUNKS + 255) / 256;      // 9 float4 per thread
constexpr int F_PPX = 32;                           // bytes per `mid` pixel (16 bf16)
// bytes per `mid` row (1088 used): an odd number of 16-byte slots (the two pixel rows of a block-1 tile then use opposite
// halves of the 32-byte bank groups) and = 4 dwords (mod 32 banks): the 16 region rows a ds_write_b64 lane group stores
// to are then 2-way instead of 8-way conflicted
constexpr int F_PROWB = 1168;
static_assert(F_INROWB >= F_INV * 2 && F_PROWB >= F_RX * F_PPX && F_PROWB % 32 == 16, "LDS row strides");
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

// One ds_read_b64 (two LDS cycles, 64 banks).  Written as assembly because the optimiser pairs any two 8-byte LDS loads off
// one base register into a ds_read2_b64 (eight cycles, 32 banks) - the halves of a fragment, or the same half of two input
// rows.  The compiler does not count these reads in lgkmcnt: lds_wait<N>() below is their wait, and the "memory" clobber
// keeps its own LDS operations on their side of each read.
template <int OFF>
__device__ __forceinline__ u32x2 lds_read64(int addr) {
    u32x2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// at most N LDS operations of this wave still in flight; the fragments pass through so that their users cannot move above it
template <int N>
__device__ __forceinline__ void lds_wait(u32x4 (&bf)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]) : "n"(N) : "memory");
}

// developer knock-outs (tools/fused_variants.sh): 1 skip block 0, 2 skip block 1, 4 skip the global fetch, 8 skip the output stores
#ifndef F_KO
#define F_KO 0
#endif
#ifndef F_W1_LDS
#define F_W1_LDS 0             // block 1's weights: 0 in 36 VGPRs per lane, 1 in LDS (one more 16-byte read per MFMA)
#endif
#ifndef F_MIN_WAVES
#define F_MIN_WAVES 2          // waves per SIMD the register allocation aims at (= resident workgroups per CU)
#endif
__global__ __launch_bounds__(256, F_MIN_WAVES) void conv01_fused_bf16_kernel(const FusedArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_in[F_IN_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char lds_mid[F_MID_BYTES];
#if F_W1_LDS
    __shared__ __attribute__((aligned(16))) unsigned char lds_w1[9 * 1024];          // [tap][channel][k half] x 16 bytes
#endif
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
    // Tile -> (region row yr, pixel pair b) of lane l31.  Tiles 0..7: rows 0..15 x pairs 2t, 2t+1 (yr = l31 & 15, b = 2t +
    // (l31 >> 4)): a tile is then the lane's base address + t * constant - immediates - for both the fragment reads and the
    // `mid` stores, and the LDS lane groups (32 lanes of a ds_read_b64, 16 of a ds_write_b64) run down a COLUMN of the
    // region, which the row strides above spread over the banks.  The rest of the 18 x 17 region: tile 8 = pair 16 of rows
    // 0..15 and pairs 0..7 of rows 16, 17; tile 9 = pairs 8..16 of rows 16, 17 (18 lanes; the others repeat the last one
    // and store nothing).
    auto in_addr_of = [&](int yr, int b) { return 2 * yr * F_INROWB + 24 * b + 16 * ph + 16 * h; };
    auto mid_addr_of = [&](int yr, int b) { return yr * F_PROWB + (2 * b + ph) * F_PPX + (8 * hf + 4 * h) * 2; };
    const int yr_l = l31 & 15, b_l = l31 >> 4;
    const int yr_8 = l31 < 16 ? l31 : 16 + ((l31 - 16) >> 3), b_8 = l31 < 16 ? F_NB - 1 : (l31 - 16) & 7;
    const int i9 = l31 < 18 ? l31 : 17, yr_9 = 16 + i9 / 9, b_9 = 8 + i9 % 9;
    const bool live9 = l31 < 18;
    // (the two 8-byte halves of a fragment go through separate base registers so that they stay two ds_read_b64 - two LDS
    //  cycles each over 64 banks; merged into one ds_read2_b64 they take eight over 32)
    // (these feed ds_read_b64 written as assembly: absolute LDS addresses, the array's own offset included)
    typedef __attribute__((address_space(3))) unsigned char *lds_byte_ptr;
    const int in_org = (int)(unsigned)(uintptr_t)(lds_byte_ptr)&lds_in[0];
    int in_lane = in_org + in_addr_of(yr_l, b_l), in_lane_hi = in_lane + 8;
    int in_lane8 = in_org + in_addr_of(yr_8, b_8), in_lane8_hi = in_lane8 + 8, in_lane9 = in_org + in_addr_of(yr_9, b_9), in_lane9_hi = in_lane9 + 8;
    asm volatile("" : "+v"(in_lane_hi), "+v"(in_lane8_hi), "+v"(in_lane9_hi));
    const int mid_lane = mid_addr_of(yr_l, b_l), mid_lane8 = mid_addr_of(yr_8, b_8), mid_lane9 = mid_addr_of(yr_9, b_9);

    // ------------------------------------------------------------------ block 1: weights (sign of the BN scale folded in)
    const float asc1 = fabsf(a.scale1[l31]), sh1 = a.shift1[l31];
#if F_W1_LDS
    for (int i = tid; i < 9 * 64; i += 256) {
        const int tap = i >> 6, ch = (i >> 1) & 31, hh = i & 1;
        u32x4 raw = *reinterpret_cast<const u32x4 *>(a.w1 + ((size_t)ch * 9 + tap) * 16 + 8 * hh);
        const unsigned f2 = a.scale1[ch] < 0.f ? 0x80008000u : 0u;
        raw ^= (u32x4){f2, f2, f2, f2};
        *reinterpret_cast<u32x4 *>(&lds_w1[tap * 1024 + ch * 32 + hh * 16]) = raw;
    }
    const int w1_lane = l31 * 32 + h * 16;
    auto w1_frag = [&](int t) { return *reinterpret_cast<const bf16x8 *>(__builtin_assume_aligned(&lds_w1[w1_lane + t * 1024], 16)); };
#else
    bf16x8 w1r[9];
    {
        const unsigned f2 = a.scale1[l31] < 0.f ? 0x80008000u : 0u;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            u32x4 raw = *reinterpret_cast<const u32x4 *>(a.w1 + ((size_t)l31 * 9 + t) * 16 + 8 * h);
            raw ^= (u32x4){f2, f2, f2, f2};
            w1r[t] = __builtin_bit_cast(bf16x8, raw);
        }
    }
    auto w1_frag = [&](int t) { return w1r[t]; };
#endif
    // A fragment of tile (row pair rp, column half chf): pixel l31 in pool-window order
    const int q = l31 & 3, wdw = l31 >> 2;
    const int px_off = (q >> 1) * F_PROWB + (2 * wdw + (q & 1)) * F_PPX + 16 * h;

    // ------------------------------------------------------------------ input staging
    // Pass k moves input rows 4k .. 4k+3: thread (r4 = tid / 54, chunk = tid % 54) one float4 of row 4k + r4 (threads 216..255
    // rest).  Everything that depends on k is a scalar offset of the load / an immediate offset of the LDS store.
    constexpr int F_PASSES = (F_INROWS + 3) / 4, F_HALF = F_PASSES / 2;      // 10 passes, fetched five at a time
    static_assert(F_PASSES == 2 * F_HALF, "two equal halves");
    u32x4 stage[F_HALF];
    const int st_r4 = tid / F_CHUNKS, st_ch = tid - st_r4 * F_CHUNKS;
    const bool st_live = tid < 4 * F_CHUNKS;
    const int st_lds = st_r4 * F_INROWB + st_ch * 8;
    auto decode = [&](int p, int &n, int &y1, int &x1) {
        p = __builtin_amdgcn_readfirstlane(p);
        const unsigned r = fdiv_u((unsigned)p, a.d_tx), tx = (unsigned)p - r * (unsigned)a.tiles_x;
        const unsigned nn = fdiv_u(r, a.d_ty), ty = r - nn * (unsigned)a.tiles_y;
        n = __builtin_amdgcn_readfirstlane((int)nn);
        y1 = __builtin_amdgcn_readfirstlane((int)ty * F_PH);
        x1 = __builtin_amdgcn_readfirstlane((int)tx * F_PW);
    };
    const int row_f = 3 * a.W0;
    // rows 2 y1 - 3 ..., floats 6 x1 - 12 ... of image n; zeros outside it.  The lane's part of the offset is (r4 * row + chunk),
    // the pass's part (2 y1 - 3 + 4k rows) rides in the scalar offset, which the range check counts too: rows below the
    // image fall outside the resource.  Only pass 0 of the image's top patches has a negative row part: per-lane offsets.
    auto fetch_half = [&](int p, int half) {
        const bool live = p < a.patches;
        int n, y1, x1;
        decode(live ? p : 0, n, y1, x1);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(a.x) + (long long)n * a.H0 * row_f, 0, live ? a.H0 * row_f * 4 : 0, 0x00020000);
        const int c0 = 6 * x1 - 12;
        const bool in_row = st_live && (unsigned)(4 * st_ch + c0) < (unsigned)row_f;
        const unsigned lane_vo = in_row ? (unsigned)((st_r4 * row_f + c0 + 4 * st_ch) * 4) : kOOB;
#pragma unroll
        for (int kk = 0; kk < F_HALF; ++kk) {
            const int k = half * F_HALF + kk;
            const int rbase = 2 * y1 - 3 + 4 * k;               // >= 1 for k >= 1
            unsigned vo = lane_vo;
            int so = rbase * row_f * 4;
            if (k == 0) {
                const int row = rbase + st_r4;
                const unsigned neg_vo = in_row && row >= 0 ? (unsigned)((row * row_f + c0 + 4 * st_ch) * 4) : kOOB;
                vo = rbase < 0 ? neg_vo : lane_vo;
                so = rbase < 0 ? 0 : so;
            }
            if (4 * k + 3 >= F_INROWS) vo = 4 * k + st_r4 < F_INROWS ? vo : kOOB;
            if (F_KO & 4) stage[kk] = (u32x4){vo, vo, vo, vo} & 0x3f000000u;
            else stage[kk] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, so, 0);
        }
    };
    auto store_half = [&](int half) {
        if (st_live) {
#pragma unroll
            for (int kk = 0; kk < F_HALF; ++kk) {
                const int k = half * F_HALF + kk;
                if (4 * k + 3 < F_INROWS || 4 * k + st_r4 < F_INROWS) {
                    const u32x2 v = {pack_bf16(__uint_as_float(stage[kk][0]), __uint_as_float(stage[kk][1])),
                                     pack_bf16(__uint_as_float(stage[kk][2]), __uint_as_float(stage[kk][3]))};
                    *reinterpret_cast<u32x2 *>(&lds_in[st_lds + k * 4 * F_INROWB]) = v;
                }
            }
        }
    };

    const float alpha = a.alpha;
    int p = blockIdx.x;
    fetch_half(p, 0);
    store_half(0);
    fetch_half(p, 1);
    store_half(1);
    __syncthreads();
    for (; p < a.patches; p += gridDim.x) {
        fetch_half(p + gridDim.x, 0);                // the next patch's first 20 rows land while block 0 computes
        int n, y1, x1;
        decode(p, n, y1, x1);

        // ---------------------------------------------------------------- block 0 -> mid
        {
            // region pixel (yr, xr) is pooled pixel (y1 - 1 + yr, x1 - 1 + xr): outside the image it is block 1's zero padding
            const bool interior = y1 > 0 && y1 + F_PH < H1 && x1 > 0 && x1 + F_PW < W1;
            auto frags = [&](auto tc, u32x4 (&bf)[4]) {
                constexpr int t = decltype(tc)::value, toff = t < 8 ? t * 48 : 0;
                const int lo_base = t < 8 ? in_lane : (t == 8 ? in_lane8 : in_lane9);
                const int hi_base = t < 8 ? in_lane_hi : (t == 8 ? in_lane8_hi : in_lane9_hi);
                const u32x2 l0 = lds_read64<toff>(lo_base), h0 = lds_read64<toff>(hi_base);
                const u32x2 l1 = lds_read64<toff + F_INROWB>(lo_base), h1 = lds_read64<toff + F_INROWB>(hi_base);
                const u32x2 l2 = lds_read64<toff + 2 * F_INROWB>(lo_base), h2 = lds_read64<toff + 2 * F_INROWB>(hi_base);
                const u32x2 l3 = lds_read64<toff + 3 * F_INROWB>(lo_base), h3 = lds_read64<toff + 3 * F_INROWB>(hi_base);
                bf[0] = (u32x4){l0[0], l0[1], h0[0], h0[1]};
                bf[1] = (u32x4){l1[0], l1[1], h1[0], h1[1]};
                bf[2] = (u32x4){l2[0], l2[1], h2[0], h2[1]};
                bf[3] = (u32x4){l3[0], l3[1], h3[0], h3[1]};
            };
            // (wait: everything but the 8 reads of the tile fetched ahead has landed; the last tiles have nothing behind them)
            auto mma = [&](auto pending_c, u32x4 (&bf)[4], f32x16 &acc) {
                lds_wait<decltype(pending_c)::value>(bf);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int rho = 0; rho < 4; ++rho)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tz[rho], __builtin_bit_cast(bf16x8, bf[rho]), acc, 0, 0, 0);
            };
            auto finish = [&](auto interior_c, int t, const f32x16 &acc) {
                constexpr bool INTERIOR = decltype(interior_c)::value;
                float v[4];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float m = fmaxf(fmaxf(fmaxf(acc[r4], acc[4 + r4]), acc[8 + r4]), acc[12 + r4]);
                    const float z = m * asc0[r4] + sh0[r4];
                    v[r4] = fmaxf(z, z * alpha);
                }
                u32x2 o = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])};
                if constexpr (!INTERIOR) {
                    const int yr = t < 8 ? yr_l : (t == 8 ? yr_8 : yr_9), b = t < 8 ? 2 * t + b_l : (t == 8 ? b_8 : b_9);
                    const int yy = y1 - 1 + yr, xx = x1 - 1 + ph + 2 * b;
                    const bool inside = (unsigned)yy < (unsigned)H1 && (unsigned)xx < (unsigned)W1;
                    o[0] = inside ? o[0] : 0u;
                    o[1] = inside ? o[1] : 0u;
                }
                if (t < 8) *reinterpret_cast<u32x2 *>(&lds_mid[mid_lane + t * 4 * F_PPX]) = o;
                else if (t == 8) *reinterpret_cast<u32x2 *>(&lds_mid[mid_lane8]) = o;
                else if (live9) *reinterpret_cast<u32x2 *>(&lds_mid[mid_lane9]) = o;
            };
            // the MFMAs of tile t + 1 are issued before the epilogue of tile t (two accumulator sets): the vector work of
            // one runs under the matrix work of the other
            auto block0 = [&](auto interior_c) {
                u32x4 bA[4], bB[4];
                f32x16 accA, accB;
                using std::integral_constant;
                typedef integral_constant<int, 8> P8;
                typedef integral_constant<int, 0> P0;
                frags(integral_constant<int, 0>{}, bA);
                frags(integral_constant<int, 1>{}, bB);
                mma(P8{}, bA, accA);
#define F_PAIR(T)                                                                                   \
                frags(integral_constant<int, T + 2>{}, bA);                                          \
                mma(P8{}, bB, accB);                                                                 \
                finish(interior_c, T, accA);                                                         \
                frags(integral_constant<int, T + 3>{}, bB);                                          \
                mma(P8{}, bA, accA);                                                                 \
                finish(interior_c, T + 1, accB);
                F_PAIR(0) F_PAIR(2) F_PAIR(4) F_PAIR(6)
#undef F_PAIR
                mma(P0{}, bB, accB);
                finish(interior_c, 8, accA);
                finish(interior_c, 9, accB);
            };
            static_assert(F_NT == 10, "the tile schedule above is written out for ten tiles");
            if (!(F_KO & 1)) {
                if (interior) block0(std::true_type{});
                else block0(std::false_type{});
            }
        }
        __syncthreads();
        store_half(0);                               // `in` was last read before this barrier
        fetch_half(p + gridDim.x, 1);                // ... and its last 18 rows while block 1 computes

        // ---------------------------------------------------------------- block 1: mid -> y
        {
            const int oy0 = y1 >> 1, ox0 = x1 >> 1;
            const long long o_org = (((long long)n * H2 + oy0) * W2 + ox0) * 32;
            const long long o_left = ((long long)a.NB * H2 * W2 * 32 - o_org) * 2;
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                a.y + o_org, 0, (F_KO & 8) ? 0 : (int)(o_left > 0xFFFFFFFFll ? 0xFFFFFFFFll : o_left), 0x00020000);
            const unsigned row_b = (unsigned)W2 * 64u;
            const bool full = y1 + F_PH <= H1 && x1 + F_PW <= W1;
            const int rows_left = H2 - oy0, cols_left = W2 - ox0;
            auto block1 = [&](auto full_c) {
                constexpr bool FULL = decltype(full_c)::value;
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
                        const bf16x8 wf = w1_frag(t);
#pragma unroll
                        for (int chf = 0; chf < 2; ++chf) {
                            const bf16x8 af = *reinterpret_cast<const bf16x8 *>(
                                __builtin_assume_aligned(&lds_mid[2 * rp * F_PROWB + 16 * chf * F_PPX + px_off + toff], 16));
                            acc[chf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wf, acc[chf], 0, 0, 0);
                        }
                    }
                    // lane holds channel l31 of windows 2 g + h (g = r >> 2) of each column half; r & 3 = position in the window
                    const unsigned lane_o = (unsigned)rp * row_b + (unsigned)h * 64u + (unsigned)l31 * 2u;
#pragma unroll
                    for (int chf = 0; chf < 2; ++chf)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float m = fmaxf(fmaxf(fmaxf(acc[chf][4 * g], acc[chf][4 * g + 1]), acc[chf][4 * g + 2]), acc[chf][4 * g + 3]);
                            const float z = m * asc1 + sh1;
                            const float v = fmaxf(z, z * alpha);
                            const int oc = 8 * chf + 2 * g;         // output column inside the patch (+ h)
                            if constexpr (FULL) {
                                __builtin_amdgcn_raw_buffer_store_b16(bf16_bits(v), o_rsrc, (int)lane_o, oc * 64, 0);
                            } else {
                                const bool ok = rp < rows_left && oc + h < cols_left;
                                __builtin_amdgcn_raw_buffer_store_b16(bf16_bits(v), o_rsrc, (int)(ok ? lane_o + (unsigned)oc * 64u : kOOB), 0, 0);
                            }
                        }
                }
            };
            if (!(F_KO & 2)) {
                if (full) block1(std::true_type{});
                else block1(std::false_type{});
            }
        }
        store_half(1);
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
