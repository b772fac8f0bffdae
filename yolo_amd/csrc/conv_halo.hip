// 3x3 conv + folded BN + LeakyReLU (+ 2x2 max-pool) blocks of the bf16 network on an LDS-resident input halo
// (v_mfma_f32_16x16x32_bf16, 4 x 4 blocks per wave; /root/reference/models/model_tiny_yolov1.py:52-84, the blocks with
// Cin = 32 ... 1024).
//
// Why a second kernel next to conv_persist_kernel (conv_gemm.hip): that loop fetches a 128-byte row of A per output
// pixel, tap and K chunk - the im2col matrix, every activation nine times - and in bf16 the fetch, not the matrix pipe,
// bounds it: per 16 MFMAs (512 matrix cycles) a wave issues 8 LDS-DMA instructions at 60-185 cycles each.
// Here a workgroup owns a PH x PW patch of pre-pool pixels (GEMM rows in pool-window order, as there) - or, for a layer
// without a pool, NI whole images - and all of Cin:
//   * the (PH+2) x (PW+2) halo of a 64-byte or 128-byte K chunk is moved to LDS ONCE (LDS-DMA, zeros outside the image)
//     and serves all nine taps: a tap is a constant shift of the fragment address;
//   * the weights stream through a ring of four stages (one tap x BN channels x chunk), fetched three stages ahead
//     (counted vmcnt, raw s_barrier; fragments are read one 32-k step ahead of their MFMAs across the stage barriers), or
//     - when all of them fit (Cin = 32: 36 KB) - are loaded once and stay;
//   * the halo of the next chunk / next patch lands in the other halo buffer while this one multiplies.
// LDS-DMA bytes per FLOP drop to a third ... a half.  Bank conflicts: a pixel's (a weight row's) 16-byte K slots are
// XOR-swizzled with a key from its index (and, for pixels, its halo row) on the source side of the DMA; a fragment read is
// base(lane, tap) ^ (k-step << 6) (+ immediate); switching halo buffers adds +-(buffer size) to the 36 bases.
// Epilogue as in conv_persist_kernel: BN, 2x2 max over four accumulator registers, LeakyReLU once, bf16 store.
// Instantiations and what the dispatcher requires of a layer: conv3x3_halo_bf16_supported() at the end of this file.
#include "common.h"

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *lds_ptr;
constexpr unsigned kOOB = 0xFFFFFFFFu;
constexpr unsigned kNoStore = 0x80000000u;       // (+ a small lane offset stays outside every output resource)

struct HaloArgs {
    const void *x, *w;
    const float *scale, *shift;
    void *y;
    int NB, H, W, Cin, Cout;
    float alpha;
    int tiles_x, tiles_y, ntiles, items;         // patches per image row / column, channel tiles, work items
    float *part;                                 // K-split instantiations: fp32 partial tiles [item][rows of the tile][BN]
    int splits;                                  // ... an item is one of `splits` K slices of a (patch, channel tile)
    int key_a, key_c;                            // developer sweep of the halo swizzle key (dev build only; -1: the compiled-in pair)
    long long *stamp;                            // developer timeline (dev build only; 8 x i64 per wave, tools/stamp_trace.py --halo)
    int dbg;                                     // developer knock-outs (dev build only): 1 no halo fetch, 2 no weight fetch, 4 no barriers, 8 no epilogue, 16 DMA issue of the two waves of a SIMD staggered, 64 no fence between a step's fragment reads and its MFMAs
};
#ifdef TY1_DEV_STAMPS
#define HALO_DBG(bit) ((a.dbg & (bit)) != 0)
#define HALO_STAMP(var) do { if (do_stamp) { __builtin_amdgcn_sched_barrier(0); var = (long long)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define HALO_DBG(bit) false
#define HALO_STAMP(var) do { } while (0)
#endif

__device__ __forceinline__ unsigned short bf16_bits(float v) {     // round to nearest even
    const __bf16 hv = (__bf16)v;
    return __builtin_bit_cast(unsigned short, hv);
}

// CK: channels per K chunk (32 or 64).  BN: output channels per item.  PH x PW: patch (even, at most as many pixels as the tile has rows).
// RES_B: Cin == CK and Cout == BN - the whole weight tensor (9 x BN x CK) is loaded once and stays in LDS.
// NI > 1 (layers without a pool, whole maps): an item is NI consecutive images of PH x PW = H x W pixels, each with its own
// ring of zeros in LDS; GEMM rows in natural order (image, y, x), every accumulator is stored.
// FM x FN: accumulator blocks per wave.  M16: the blocks are 16 x 16 (v_mfma_f32_16x16x32_bf16, one fragment read = 16 rows x
// 32 k) instead of 32 x 32 (v_mfma_f32_32x32x16_bf16, 32 rows x 16 k): the same reads, bytes and matrix cycles per FLOP, but
// on random operands the chip holds a higher clock under the 16 x 16 shape (MI355X_MICROARCH.md, DVFS give-back item 7).
// NL: the first NL waves of the workgroup issue all LDS-DMA (default: every wave its share).
// KSPL (layers without a pool): an item reduces only Cin / splits channels and stores its raw fp32 accumulators as a partial
// tile; halo_fixup_kernel sums the slices in a fixed order and applies BN + LeakyReLU (deep K with too few tiles for the chip).
template <int CK, int BN, int WAVES_M, int WAVES_N, int PH, int PW, bool RES_B, int NI = 1, int FM = 2, int FN = 2, bool M16 = false,
          int NL = WAVES_M * WAVES_N, bool KSPL = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, RES_B ? 2 : 1) void conv_halo_bf16_kernel(const HaloArgs a) {
    constexpr int BS = M16 ? 16 : 32;                                  // rows / columns of an accumulator block
    constexpr int KS = M16 ? 32 : 16, KSH = M16 ? 6 : 5;               // k per MFMA; log2 of its bytes in a row
    constexpr int GW = M16 ? 1 : 4;                                    // pool windows of a block held by one lane (4 registers each)
    using Acc = typename std::conditional<M16, f32x4, f32x16>::type;
    constexpr bool POOL = NI == 1;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int CKB = CK * 2, NSLOT = CKB / 16, NS16 = CK / KS;      // bytes, 16-byte slots, MFMA k-steps per chunk row
    constexpr int HWD = PW + 2, HPI = (PH + 2) * HWD, HPIX = NI * HPI; // halo width, halo pixels per image, halo pixels
    constexpr int PIXPP = 1024 / CKB;                                  // pixels (rows) per 1 KiB DMA piece
    // swizzle key of row (pixel / weight row) idx: rows idx and idx + RPW start on the same bank, so the key changes every
    // RPW rows and 16 consecutive rows give 16 distinct (start bank, slot) pairs = all 64 banks for a ds_read_b128 group
    constexpr int RPW = 256 / CKB;
    auto key_of = [](int idx) { return (idx / RPW) & (NSLOT - 1); };
    // halo pixels: a 16-lane group of a fragment read covers pixels of two or three halo rows, and with a key from the pixel
    // index alone the rows collide (1.75-2.0 passes per group in a model of the bank mapping, SQ_LDS_BANK_CONFLICT agrees);
    // the halo row enters the key with a coefficient searched per patch shape (1.1-1.3 passes; 1.67 for 16 x 16 x 32 channels)
    constexpr int KA = (CK == 32 && PW == 16) ? 2 : 0;
    constexpr int KC = (CK == 32 && PW == 16) ? 1 : (CK == 32 && PW == 14) ? 2 : PW == 28 ? 5 : PW == 14 ? 4 : PW == 7 ? 7 : 0;
    auto key_px = [&](int hp) {
        const int row = hp / HWD;
#ifdef TY1_DEV_STAMPS
        if (a.key_a >= 0) return ((hp + a.key_a * row) / RPW + a.key_c * row) & (NSLOT - 1);
#endif
        return ((hp + KA * row) / RPW + KC * row) & (NSLOT - 1);
    };
    constexpr int HP = (HPIX + PIXPP - 1) / PIXPP;                     // halo pieces
    constexpr int HPW = (HP + NL - 1) / NL;                            // ... per loading wave
    constexpr int HALO1 = HP * 1024;                                   // byte distance of the two halo buffers
    constexpr int TS = RES_B ? 9 : 1, NSC = 9 / TS, NRING = RES_B ? 1 : 4;   // taps per stage, stages per chunk, ring
    constexpr int STAGE = TS * BN * CKB, BP = STAGE / 1024, PB = BP / NL;    // bytes, pieces, pieces per loading wave
    constexpr int BOFF = HALO1 + HP * 1024;
    constexpr int MPIX = NI * PH * PW, WINS = MPIX / 4;
    static_assert(BN == BS * FN * WAVES_N, "a wave owns (BS FM) x (BS FN) of the tile");
    static_assert(MPIX <= BS * FM * WAVES_M && (!POOL || (MPIX % 4 == 0 && PH % 2 == 0 && PW % 2 == 0)), "patch shape");
    static_assert(BP % NL == 0 && NL <= NW, "weight pieces divide among the loading waves");
    static_assert(RES_B || HPW <= 2 * NSC, "at most two halo pieces per wave and stage");
    static_assert(BOFF + NRING * STAGE <= 160 * 1024, "LDS");
    static_assert(PH + 2 < 256 && PW + 2 < 256, "halo coordinates are packed in bytes");
    __shared__ __attribute__((aligned(1024))) char lds[BOFF + NRING * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const bool is_loader = NL == NW || wave < NL;
    const int G = gridDim.x;
    const int nch = KSPL ? a.Cin / CK / a.splits : a.Cin / CK;       // K chunks of one item
    static_assert(!KSPL || (NI > 1 && !RES_B), "K split: layers without a pool, streamed weights");
    const int frag_row = lane & (BS - 1), h = lane / BS;               // row of a block, 16-byte k slot within an MFMA step
    const int Ho = a.H >> 1, Wo = a.W >> 1;

    // ---- fragment read bases.  A: pixel of GEMM row m (pool-window order) shifted by the tap, slot h of k-step 0.
    unsigned abase[FM][9];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        int m = wave_m * (BS * FM) + BS * i + frag_row;
        if (m >= MPIX) m = 0;                                          // padding rows: any valid pixel, never stored
        int py, px, sub = 0;
        if (POOL) {
            const int win = m >> 2, q = m & 3;
            py = 2 * (win / (PW / 2)) + (q >> 1);
            px = 2 * (win % (PW / 2)) + (q & 1);
        } else {
            sub = m / (PH * PW);
            const int rem = m - sub * (PH * PW);
            py = rem / PW;
            px = rem - py * PW;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hp = sub * HPI + (py + t / 3) * HWD + px + t % 3;
            abase[i][t] = (unsigned)(hp * CKB + ((h ^ key_px(hp)) << 4));
        }
    }
    unsigned bbase[NS16];                                              // B: weight row of this lane, slot of k-step s
    {
        const int n_l = wave_n * (BS * FN) + frag_row;
#pragma unroll
        for (int s = 0; s < NS16; ++s) bbase[s] = (unsigned)(BOFF + n_l * CKB + ((((KS / 8) * s + h) ^ key_of(n_l)) << 4));
    }

    // ---- DMA geometry of this lane: row dl_row of a piece, 16-byte slot dl_slot
    const int dl_row = lane / NSLOT, dl_slot = lane % NSLOT;
    unsigned voff_b[PB];                                               // weight pieces wave + NW * k of a stage
#pragma unroll
    for (int k = 0; k < PB; ++k) {
        const int r = (wave + NL * k) * PIXPP + dl_row, n = r % BN, t = r / BN;
        voff_b[k] = (unsigned)(((n * 9 + t) * a.Cin) * 2 + ((dl_slot ^ key_of(r)) << 4));
    }
    unsigned voff_h[HPW];                                              // halo pieces wave + NW * k
    int hyx[HPW];
#pragma unroll
    for (int k = 0; k < HPW; ++k) {
        const int hp = (wave + NL * k) * PIXPP + dl_row, sub = hp / HPI, hr = hp - sub * HPI, hy = hr / HWD, hx = hr - hy * HWD;
        voff_h[k] = (unsigned)((((sub * a.H + hy) * a.W + hx) * a.Cin) * 2 + ((dl_slot ^ key_px(hp)) << 4));
        hyx[k] = hp < HPIX && wave + NL * k < HP ? (hy | (hx << 8) | (sub << 16)) : -1;
    }
    const __amdgpu_buffer_rsrc_t b_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.w), 0, (int)((long long)a.Cout * 9 * a.Cin * 2), 0x00020000);

    // ---- work: item = (patch, channel tile), channel tile fastest; a step = one K chunk of an item
    struct Step {
        int img, y0, x0, nt, c, cg, item;         // c: chunk within the item, cg: chunk within Cin
        bool live;
    };
    auto step_of = [&](int u) {
        Step d;
        const int k = u / nch;
        const long long item = (long long)blockIdx.x + (long long)k * G;
        d.live = item < a.items;
        const int it_all = d.live ? (int)item : 0;
        const int sp = KSPL ? it_all % a.splits : 0, it = KSPL ? it_all / a.splits : it_all;      // K slice fastest
        d.c = u - k * nch;
        d.cg = __builtin_amdgcn_readfirstlane(sp * nch + d.c);
        d.item = __builtin_amdgcn_readfirstlane(it_all);
        d.nt = it % a.ntiles;
        const int patch = it / a.ntiles;
        const int tx = patch % a.tiles_x, r = patch / a.tiles_x;
        d.x0 = tx * PW;
        d.y0 = (r % a.tiles_y) * PH;
        d.img = (r / a.tiles_y) * NI;
        d.img = __builtin_amdgcn_readfirstlane(d.img); d.y0 = __builtin_amdgcn_readfirstlane(d.y0);
        d.x0 = __builtin_amdgcn_readfirstlane(d.x0); d.nt = __builtin_amdgcn_readfirstlane(d.nt);
        d.c = __builtin_amdgcn_readfirstlane(d.c);
        return d;
    };
    // halo piece k of this wave for step d into halo buffer `buf`.  The resource starts (W + 1) pixels before the image so
    // that the scalar offset of the patch origin (y0 - 1, x0 - 1) is never negative; lanes outside the image read as zero.
    const int a_shift = (a.W + 1) * a.Cin * 2;
    auto dma_halo = [&](const Step &d, int k, int buf) {
        const long long img_b = (long long)d.img * a.H * a.W * a.Cin * 2;
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(static_cast<const char *>(a.x)) + img_b - a_shift, 0,
            d.live ? NI * a.H * a.W * a.Cin * 2 + a_shift : 0, 0x00020000);
        const int soff = ((d.y0 * a.W + d.x0) * a.Cin + d.cg * CK) * 2;
        const int hy = hyx[k] & 0xFF, hx = (hyx[k] >> 8) & 0xFF, sub = hyx[k] >> 16;
        const bool ok = hyx[k] >= 0 && (unsigned)(d.y0 - 1 + hy) < (unsigned)a.H && (unsigned)(d.x0 - 1 + hx) < (unsigned)a.W &&
                        d.img + sub < a.NB;
        if (is_loader && wave + NL * k < HP && !HALO_DBG(1))
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_ptr)(lds + buf * HALO1 + (wave + NL * k) * 1024), 16,
                                                     (int)(ok ? voff_h[k] : kOOB), soff, 0, 0);
    };
    // the PB weight pieces of this wave for stage `st` of step d into ring slot `slot`
    auto dma_b = [&](const Step &d, int st, int slot) {
        const int soff = d.live ? ((d.nt * BN * 9 + st * TS) * a.Cin + d.cg * CK) * 2 : 0;
        if (HALO_DBG(2) || !is_loader) return;
#pragma unroll
        for (int k = 0; k < PB; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_ptr)(lds + BOFF + slot * STAGE + (wave + NL * k) * 1024), 16,
                                                     (int)voff_b[k], soff, 0, 0);
    };

    // ---- output geometry: lane holds channel (lane & (BS - 1)) of block column j; accumulator registers 4g .. 4g+3 of block
    //      i are the four pixels of pool window wave_m * (BS / 4) FM + (BS / 4) i + 2g + h (32 x 32 blocks) or ... + h (16 x 16)
    unsigned woff[FM][GW];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int g = 0; g < GW; ++g) {
            const int w = wave_m * (BS / 4 * FM) + BS / 4 * i + (M16 ? h : 2 * g + h);
            woff[i][g] = w < WINS ? (unsigned)(((w / (PW / 2)) * Wo + w % (PW / 2)) * a.Cout * 2) : kNoStore;
        }
    const long long y_bytes = (long long)a.NB * Ho * Wo * a.Cout * 2;

    Acc acc[FM][FN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4 * GW; ++r) acc[i][j][r] = 0.f;
    };
    // fragment ring: reads run AH k-steps ahead of their MFMAs (a 16 x 16 x 32 step is as long as two 32 x 32 x 16 steps; two
    // of them ahead measured 4 % fewer cycles per stage and +0.5 ... 1 % on the whole network, but needs 6-10 spilled registers)
    constexpr int RD = M16 ? 2 : 3, AH = RD - 1;
    struct Frag {
        bf16x8 a[FM], b[FN];
    };
    // k-step s of tap `tap`; the weights of that tap are rows [tb * BN, (tb + 1) * BN) of ring slot `slot`
    auto read_frag = [&](Frag &f, int tap, int s, int slot, int tb) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
            f.a[i] = *reinterpret_cast<const bf16x8 *>(__builtin_assume_aligned(&lds[abase[i][tap] ^ (unsigned)(s << KSH)], 16));
#pragma unroll
        for (int j = 0; j < FN; ++j)
            f.b[j] = *reinterpret_cast<const bf16x8 *>(
                __builtin_assume_aligned(&lds[bbase[s] + (unsigned)(slot * STAGE + (tb * BN + BS * j) * CKB)], 16));
    };
    auto mfma = [&](const Frag &f) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                if constexpr (M16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
            }
    };
    // (resident weights) all k-steps of a patch, fragments read two k-steps ahead of their MFMAs
    auto compute_stage = [&](int st) {
        constexpr int STEPS = TS * NS16;
        const int slot = RES_B ? 0 : st % 3;
        Frag f[RD];                                            // fragments run AH k-steps ahead of their MFMAs
        read_frag(f[0], st * TS, 0, slot, 0);
        if (AH > 1 && STEPS > 1) read_frag(f[1], st * TS + 1 / NS16, 1 % NS16, slot, 1 / NS16);
#pragma unroll
        for (int q = 0; q < STEPS; ++q) {
            if (q + AH < STEPS) read_frag(f[(q + AH) % RD], st * TS + (q + AH) / NS16, (q + AH) % NS16, slot, (q + AH) / NS16);
            if (M16 && !HALO_DBG(64)) __builtin_amdgcn_sched_barrier(0);
            mfma(f[q % RD]);
        }
    };
    auto wg_barrier = [&] {
        if (HALO_DBG(4)) return;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    // ---- streamed weights: the k-steps of a step form ONE stream - fragments are read two k-steps ahead of their MFMAs
    //      ACROSS the stage barriers (and across the step boundary), so no stage starts with an exposed LDS round trip.
    //      For that a stage must be published one barrier earlier than it is read: the weights are fetched three stages
    //      ahead into a ring of four, and the barrier that ends a stage waits only for the reads older than the eight
    //      (two k-steps) that already belong to the next stage.
    constexpr int Q = NSC * NS16;
    static_assert(RES_B || Q % RD == 0, "the fragment ring keeps its phase from step to step");
    Frag fr[RD];
    int gs = 0;                                 // stages so far: stage g of this workgroup lives in ring slot g & 3
    auto read_stream = [&](Frag &f, int tap, int s, unsigned boff, unsigned hd) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
            f.a[i] = *reinterpret_cast<const bf16x8 *>(__builtin_assume_aligned(&lds[(abase[i][tap] + hd) ^ (unsigned)(s << KSH)], 16));
#pragma unroll
        for (int j = 0; j < FN; ++j)
            f.b[j] = *reinterpret_cast<const bf16x8 *>(__builtin_assume_aligned(&lds[bbase[s] + boff + (unsigned)(BS * j * CKB)], 16));
    };
    auto stage_barrier = [&] {
        if (HALO_DBG(4)) return;
        // (the two k-steps read ahead are 2 (FM + FN) reads; the counter holds 15 at most, a stricter wait is harmless)
        constexpr int AHEAD = AH * (FM + FN) < 15 ? AH * (FM + FN) : 15;
        asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" ::"n"(AHEAD) : "memory");
    };

    // BN scale / shift of the item's channels: loaded when its first step starts, pinned behind that step's first wait
    float sc[FN], sh[FN];
    auto load_bn = [&](const Step &d) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = d.nt * BN + wave_n * (BS * FN) + BS * j + frag_row;
            sc[j] = a.scale[n];
            sh[j] = a.shift[n];
        }
    };
    auto landed = [&]() {
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(sc[j]), "+v"(sh[j]));
    };
    auto epilogue = [&](const Step &d) {
        if constexpr (KSPL) {
            // raw accumulators of this K slice -> partial tile `item` (every row of the tile, padding rows included)
            constexpr int TROWS = BS * FM * WAVES_M;
            const __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                a.part + (long long)d.item * TROWS * BN, 0, TROWS * BN * 4, 0x00020000);
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const unsigned vb = (unsigned)((wave_m * (BS * FM) + 4 * h) * BN + wave_n * (BS * FN) + BS * j + frag_row) * 4u;
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int r = 0; r < 4 * GW; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)acc[i][j][r]), p_rsrc, (int)vb,
                                                              (BS * i + (r & 3) + (M16 ? 0 : 8 * (r >> 2))) * BN * 4, 0);
            }
            return;
        }
        if constexpr (!POOL) {
            // rows of the item are consecutive rows of the output; the resource ends behind the last image that exists,
            // so padding rows and images past the batch are dropped by the range check
            const int imgs = a.NB - d.img < NI ? a.NB - d.img : NI;
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                static_cast<char *>(a.y) + (long long)d.img * (PH * PW) * a.Cout * 2, 0, imgs * (PH * PW) * a.Cout * 2, 0x00020000);
            const unsigned row_b = (unsigned)a.Cout * 2u;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const unsigned vb = (unsigned)(wave_m * (BS * FM) + 4 * h) * row_b + (unsigned)(d.nt * BN + wave_n * (BS * FN) + BS * j + frag_row) * 2u;
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int r = 0; r < 4 * GW; ++r) {
                        float t = acc[i][j][r] * sc[j] + sh[j];
                        t = t > 0.f ? t : t * a.alpha;
                        __builtin_amdgcn_raw_buffer_store_b16(bf16_bits(t), o_rsrc, (int)vb,
                                                              (BS * i + (r & 3) + (M16 ? 0 : 8 * (r >> 2))) * (int)row_b, 0);
                    }
            }
            return;
        }
        const long long o_org = (((long long)d.img * Ho + (d.y0 >> 1)) * Wo + (d.x0 >> 1)) * a.Cout * 2;
        const long long left = y_bytes - o_org;
        const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            static_cast<char *>(a.y) + o_org, 0, (int)(left > 0x7FFFFFFFll ? 0x7FFFFFFFll : left), 0x00020000);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = d.nt * BN + wave_n * (BS * FN) + BS * j + frag_row;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int g = 0; g < GW; ++g) {
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = acc[i][j][4 * g + q] * sc[j] + sh[j];
                    const float t = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));       // pool first (alpha >= 0), activate once
                    const float pooled = t > 0.f ? t : t * a.alpha;
                    __builtin_amdgcn_raw_buffer_store_b16(bf16_bits(pooled), o_rsrc, (int)(woff[i][g] + (unsigned)n * 2u), 0, 0);
                }
        }
    };

    // developer timeline of wave 0 (dev build): cycles between the stamps, summed over the stages of the workgroup
#ifdef TY1_DEV_STAMPS
    const bool do_stamp = a.stamp != nullptr;
#else
    constexpr bool do_stamp = false;
#endif
    long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0, s_mma = 0, s_vm = 0, s_bar = 0, s_epi = 0, n_st = 0;
    (void)tk0; (void)tk1; (void)tk2; (void)tk3; (void)tk4;
    if (do_stamp && lane == 0) a.stamp[((long long)blockIdx.x * NW + wave) * 8 + 0] = (long long)__builtin_amdgcn_s_memrealtime();

    // ---- prologue: first halo, first weight stages
    int u = 0;
    Step cur = step_of(0);
#pragma unroll
    for (int k = 0; k < HPW; ++k) dma_halo(cur, k, 0);
    dma_b(cur, 0, 0);
    if (!RES_B) {
        dma_b(cur, 1, 1);
        dma_b(cur, 2, 2);
    }
    load_bn(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    zero_acc();
    int hbuf = 0;
    if (!RES_B) {
        read_stream(fr[0], 0, 0, 0u, 0u);
        if (AH > 1) read_stream(fr[1], 1 / NS16, 1 % NS16, 0u, 0u);
    }

    // developer switch: the two waves of a SIMD (wave w and w + NW / 2) issue their LDS-DMA at different points of a stage
    // (-1 ... -1.5 % on a layer timed alone, +2 ... +3 % on the whole network: off)
    const bool late = HALO_DBG(16) && NW >= 8 && wave >= NW / 2;
    while (cur.live) {
        const Step nxt = step_of(u + 1);
        if (u > 0 && cur.c == 0) load_bn(cur);
        if (RES_B) {
            // the whole next halo goes out first and lands under the 9 x NS16 k-steps of this patch
#pragma unroll
            for (int k = 0; k < HPW; ++k) dma_halo(nxt, k, hbuf ^ 1);
            compute_stage(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the stores of the previous epilogue are a whole patch old)
            landed();
            wg_barrier();
        } else {
            const unsigned hd_next = hbuf ? (unsigned)-HALO1 : (unsigned)HALO1;     // the other halo buffer, relative to abase
#pragma unroll
            for (int st = 0; st < NSC; ++st) {
                // fetch: one halo piece of the next step, then the weights three stages ahead (ring slot last read in stage st - 1)
                auto fetch = [&] {
                    if (st < HPW) dma_halo(nxt, st, hbuf ^ 1);
                    if (st + NSC < HPW) dma_halo(nxt, st + NSC, hbuf ^ 1);
                    if (st + 3 < NSC) dma_b(cur, st + 3, (gs + 3) & 3);
                    else dma_b(nxt, st + 3 - NSC, (gs + 3) & 3);
                };
                HALO_STAMP(tk0);
                if (!late) fetch();
                const unsigned boff_cur = (unsigned)((gs & 3) * STAGE), boff_nxt = (unsigned)(((gs + 1) & 3) * STAGE);
#pragma unroll
                for (int ks = 0; ks < NS16; ++ks) {
                    if (late && ks == NS16 / 2) fetch();
                    const int q = st * NS16 + ks, q2 = q + AH, st2 = q2 / NS16, s2 = q2 % NS16;
                    if (q2 < Q) read_stream(fr[q2 % RD], st2, s2, st2 == st ? boff_cur : boff_nxt, 0u);
                    else read_stream(fr[q2 % RD], st2 - NSC, s2, boff_nxt, hd_next);      // first k-steps of the next step
                    // (without this fence hipcc lets the reads share registers with the fragments in use and issues them
                    //  behind ten of the sixteen MFMAs: the next step then starts with an exposed LDS round trip)
                    if (M16 && !HALO_DBG(64)) __builtin_amdgcn_sched_barrier(0);
                    mfma(fr[q % RD]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the PB newest DMAs of this wave (stage st + 3) may stay in flight; everything older - stage st + 2, the
                // halo piece, the stores of an epilogue a stage ago - has landed once the counter is down to PB
                HALO_STAMP(tk1);
                if (is_loader) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PB) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                HALO_STAMP(tk2);
                if (st == 0) landed();
                stage_barrier();
                HALO_STAMP(tk3);
                if (do_stamp) { s_mma += tk1 - tk0; s_vm += tk2 - tk1; s_bar += tk3 - tk2; ++n_st; }
                ++gs;
            }
        }
        if (cur.c == nch - 1) {
            HALO_STAMP(tk3);
            if (!HALO_DBG(8)) epilogue(cur);
            zero_acc();
            HALO_STAMP(tk4);
            if (do_stamp) s_epi += tk4 - tk3;
        }
        {
            const unsigned delta = hbuf ? (unsigned)-HALO1 : (unsigned)HALO1;
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int t = 0; t < 9; ++t) abase[i][t] += delta;
        }
        hbuf ^= 1;
        cur = nxt;
        ++u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // no LDS-DMA may outlive the workgroup
    if (do_stamp && lane == 0) {
        long long *o = a.stamp + ((long long)blockIdx.x * NW + wave) * 8;
        o[1] = s_mma; o[2] = s_vm; o[4] = s_bar; o[5] = s_epi; o[7] = n_st;
        o[3] = (long long)__builtin_amdgcn_s_memrealtime();
    }
}

// y = LeakyReLU(scale * sum over the K slices + shift) for the partial tiles of a K-split launch (NI images of P pixels per
// tile, rows in image order); slices are added in index order.  One thread per four channels of a row.
template <int TROWS, int BN>
__global__ __launch_bounds__(256) void halo_fixup_kernel(const float *__restrict__ part, const float *__restrict__ scale,
                                                         const float *__restrict__ shift, unsigned short *__restrict__ y, int ntiles,
                                                         int splits, int rows_per_tile, long long rows_total, int Cout, float alpha) {
    const int tile = blockIdx.x, nt = tile % ntiles, group = tile / ntiles;
    const int e = blockIdx.y * 256 + threadIdx.x, row = e / (BN / 4), c4 = (e - row * (BN / 4)) * 4;
    const long long orow = (long long)group * rows_per_tile + row;
    if (row >= rows_per_tile || orow >= rows_total) return;
    const float4 *p = reinterpret_cast<const float4 *>(part + ((long long)tile * splits * TROWS + row) * BN + c4);
    float4 v = p[0];
    for (int sp = 1; sp < splits; ++sp) {
        const float4 w = p[(long long)sp * TROWS * BN / 4];
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    const int n = nt * BN + c4;
    const float4 sc = *reinterpret_cast<const float4 *>(scale + n), sh = *reinterpret_cast<const float4 *>(shift + n);
    float t[4] = {v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w};
    unsigned short o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = bf16_bits(t[q] > 0.f ? t[q] : t[q] * alpha);
    *reinterpret_cast<uint2 *>(y + orow * Cout + n) = make_uint2(o[0] | ((unsigned)o[1] << 16), o[2] | ((unsigned)o[3] << 16));
}

template <int CK, int BN, int WAVES_M, int WAVES_N, int PH, int PW, bool RES_B, int NI = 1, int FM = 2, int FN = 2, bool M16 = false,
          int NL = WAVES_M * WAVES_N, bool KSPL = false>
int launch_halo(const char *name, HaloArgs a, int per_cu, hipStream_t s) {
    a.tiles_x = a.W / PW;
    a.tiles_y = a.H / PH;
    a.ntiles = a.Cout / BN;
    const long long items = (long long)((a.NB + NI - 1) / NI) * a.tiles_x * a.tiles_y * a.ntiles * (KSPL ? a.splits : 1);
    TY1_REQUIRE(items < (1ll << 31), "ty1_conv3x3_bf16(halo): too many work items");
    a.items = (int)items;
    const long long want = (long long)per_cu * ty1::grid_cus();
    const double px = (double)a.NB * a.H * a.W;
    {
        ty1::ProfScope prof(name, s, 2.0 * px * a.Cout * 9 * a.Cin, 2.0 * (px * a.Cin + 9.0 * a.Cin * a.Cout + px * a.Cout / (NI == 1 ? 4 : 1)));
        conv_halo_bf16_kernel<CK, BN, WAVES_M, WAVES_N, PH, PW, RES_B, NI, FM, FN, M16, NL, KSPL>
            <<<dim3((unsigned)(items < want ? items : want)), dim3(64 * WAVES_M * WAVES_N), 0, s>>>(a);
        if (int rc = ty1::check_launch(name)) return rc;
    }
    if constexpr (KSPL) {
        constexpr int TROWS = (M16 ? 16 : 32) * FM * WAVES_M;
        const int tiles = (int)(items / a.splits);
        ty1::ProfScope fix("conv_halo_fixup", s, 0.0, 4.0 * (double)items * TROWS * BN + 2.0 * px * a.Cout);
        halo_fixup_kernel<TROWS, BN><<<dim3((unsigned)tiles, (unsigned)((NI * PH * PW * (BN / 4) + 255) / 256)), 256, 0, s>>>(
            a.part, a.scale, a.shift, static_cast<unsigned short *>(a.y), a.ntiles, a.splits, NI * PH * PW, (long long)a.NB * PH * PW,
            a.Cout, a.alpha);
        return ty1::check_launch("conv_halo_fixup");
    }
    return 0;
}

// Which halo instantiation runs a bf16 layer, decided in ONE place: the support query, the workspace query and the
// launcher all switch on this plan (never on H / W / Cin again), so a layer can only reach the instantiation whose K depth
// and patch shape the acceptance rule - and the tests - mean.
enum HaloKind { HALO_NONE = 0, HALO_C32, HALO_8X28, HALO_14X14, HALO_4X7X7, HALO_4X7X7_KSPLIT };
struct HaloPlan {
    HaloKind kind;
    int slices;          // K slices of HALO_4X7X7_KSPLIT, else 1
};

HaloPlan halo_plan(int N, int H, int W, int Cin, int Cout, int pool) {
    const HaloPlan none{HALO_NONE, 1};
    if ((long long)4 * H * W * Cin * 2 + (long long)(W + 1) * Cin * 2 >= (1ll << 31)) return none;
    const long long cus = ty1::num_cus();
    const bool force = ty1::dev_env("TY1_HALO_FORCE") != nullptr;     // developer sweep: every shape the kernels can run
    if (!pool) {
        // 7x7 maps without a pool, four whole images per item: one item is Cin / 64 x 9 weight stages long, so only when
        // the items come in whole rounds of the chip (7x7x512->1024 at 128 images: 256 items)
        if (H == 7 && W == 7 && Cin % 64 == 0 && Cout % 128 == 0 && !ty1::dev_env("TY1_NO_HALO7")) {
            const long long items = (long long)((N + 3) / 4) * (Cout / 128);
            if (items >= cus) return (Cin <= 512 && (force || items % cus == 0)) ? HaloPlan{HALO_4X7X7, 1} : none;
            // too few tiles for the chip (7x7x1024->256 at 128 images: 64, 7x7x512->1024 at 64 images: 128): the items are K
            // slices whose raw accumulators go to a workspace (conv3x3_halo_bf16_ws_bytes), a second launch sums them
            if (Cin >= 512 && Cin <= 2048 && !ty1::dev_env("TY1_NO_HALO7K"))
                for (int sp = 2; sp <= 8; sp *= 2)
                    if ((Cin / 64) % sp == 0 && Cin / 64 / sp >= 2 && ((items * sp) % cus == 0 || (force && items * sp >= cus)))
                        return HaloPlan{HALO_4X7X7_KSPLIT, sp};
        }
        return none;
    }
    // item-count thresholds: tools/experiments/halo_batch_sweep.sh (16 ... 256 images, every shape forced vs the implicit GEMM)
    if (Cin == 32 && Cout == 64 && H % 16 == 0 && W % 16 == 0)
        return (force || (long long)N * (H / 16) * (W / 16) >= 3 * cus) ? HaloPlan{HALO_C32, 1} : none;
    if ((Cin == 64 || Cin == 128) && Cout % 128 == 0 && H % 8 == 0 && W % 28 == 0)      // one or two K chunks: what the tests cover
        return (force || (long long)N * (H / 8) * (W / 28) * (Cout / 128) * 4 >= 3 * cus) ? HaloPlan{HALO_8X28, 1} : none;
    // 28x28 / 14x14 maps: 14x14 patches (196 pixels of a 256-row tile: a quarter of the rows is padding, and the kernel is
    // still 5-27 % faster than the im2col loop) - but an item is 18-36 weight stages long, so only when the items come
    // in whole rounds of the chip (one round is enough) or there are many of them
    if ((Cin == 128 || Cin == 256) && Cout % 128 == 0 && H % 14 == 0 && W % 14 == 0 && !ty1::dev_env("TY1_NO_HALO14")) {
        const long long items = (long long)N * (H / 14) * (W / 14) * (Cout / 128);
        return (force || (items >= cus && (items % cus == 0 || items >= 8 * cus))) ? HaloPlan{HALO_14X14, 1} : none;
    }
    return none;
}

}  // namespace

namespace ty1 {

// 0: no halo instantiation takes this bf16 layer (the caller runs the implicit-GEMM kernel); 1: one does; >= 2: one does,
// with that many K slices (and a workspace, conv3x3_halo_bf16_ws_bytes).
int conv3x3_halo_bf16_supported(int N, int H, int W, int Cin, int Cout, int pool) {
    const HaloPlan p = halo_plan(N, H, W, Cin, Cout, pool);
    return p.kind == HALO_NONE ? 0 : (p.kind == HALO_4X7X7_KSPLIT ? p.slices : 1);
}

// workspace bytes of a layer conv3x3_halo_bf16_supported() accepts (0: none needed)
size_t conv3x3_halo_bf16_ws_bytes(int N, int H, int W, int Cin, int Cout, int pool) {
    const int sp = conv3x3_halo_bf16_supported(N, H, W, Cin, Cout, pool);
    if (pool || sp < 2) return 0;
    return (size_t)((N + 3) / 4) * (Cout / 128) * sp * 256 * 128 * sizeof(float);
}

int conv3x3_halo_bf16(const void *x, const void *w, const float *scale, const float *shift, void *y, int N, int H, int W,
                      int Cin, int Cout, int pool, float alpha, float *ws, hipStream_t s) {
    HaloArgs a{};
    a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.y = y;
    a.NB = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.alpha = alpha;
    if (const char *e = ty1::dev_env("TY1_HALO_DBG")) a.dbg = atoi(e);
    a.key_a = a.key_c = -1;
    if (const char *e = ty1::dev_env("TY1_HALO_KEY")) sscanf(e, "%d,%d", &a.key_a, &a.key_c);
#ifdef TY1_DEV_STAMPS
    a.stamp = ty1::g_stamp;
#endif
    const HaloPlan plan = halo_plan(N, H, W, Cin, Cout, pool);
    TY1_REQUIRE(plan.kind != HALO_NONE, "ty1_conv3x3_bf16(halo): no halo instantiation for N=%d %dx%d %d->%d pool=%d", N, H, W, Cin, Cout, pool);
    // every wave owns 64 x 64 of the tile as 4 x 4 blocks of v_mfma_f32_16x16x32_bf16 (in-call A/Bs against 2 x 2 blocks of
    // 32x32x16: -1.5 ... -3 % per layer timed alone; whole bf16 network +5.4, +4.4, +0.4 and -1.5 % images/s on four devices)
    if (ty1::dev_env("TY1_HALO_M32") && plan.kind != HALO_4X7X7_KSPLIT) {          // developer override: 2 x 2 blocks of 32x32x16
        switch (plan.kind) {
            case HALO_4X7X7: return launch_halo<64, 128, 4, 2, 7, 7, false, 4>("conv_halo_bf16_4x7x7", a, 1, s);
            case HALO_C32: return launch_halo<32, 64, 4, 1, 16, 16, true>("conv_halo_bf16_c32", a, 2, s);
            case HALO_14X14: return launch_halo<64, 128, 4, 2, 14, 14, false>("conv_halo_bf16_14x14", a, 1, s);
            default: return launch_halo<64, 128, 4, 2, 8, 28, false>("conv_halo_bf16_8x28", a, 1, s);
        }
    }
    switch (plan.kind) {
        case HALO_4X7X7_KSPLIT:
            a.splits = plan.slices;
            a.part = ws;
            TY1_REQUIRE(ws, "ty1_conv3x3_bf16(halo): K-sliced layer without workspace");
            return launch_halo<64, 128, 4, 2, 7, 7, false, 4, 4, 4, true, 8, true>("conv_halo_bf16_4x7x7_ksplit", a, 1, s);
        case HALO_4X7X7: return launch_halo<64, 128, 4, 2, 7, 7, false, 4, 4, 4, true>("conv_halo_bf16_4x7x7", a, 1, s);
        case HALO_C32: return launch_halo<32, 64, 4, 1, 16, 16, true, 1, 4, 4, true>("conv_halo_bf16_c32", a, 2, s);
        case HALO_14X14: return launch_halo<64, 128, 4, 2, 14, 14, false, 1, 4, 4, true>("conv_halo_bf16_14x14", a, 1, s);
        default: return launch_halo<64, 128, 4, 2, 8, 28, false, 1, 4, 4, true>("conv_halo_bf16_8x28", a, 1, s);
    }
}

}  // namespace ty1
