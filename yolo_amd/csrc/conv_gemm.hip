// Implicit-GEMM 3x3 convolution (+ folded BatchNorm + LeakyReLU + 2x2/2 max-pool epilogue) on the matrix cores of
// gfx950: exact-f32 v_mfma_f32_32x32x2_f32, or v_mfma_f32_32x32x16_bf16 for the bf16 network.
//
// Replaces /root/reference/models/model_tiny_yolov1.py:43-87 (conv blocks).  The Dense matmul is dense_gemm.hip, the
// shallow layers (Cin = 3, 16) are conv_direct.hip.
//
// GEMM view:  M = output pixels (N*H*W), N = Cout, K = 9*Cin (k = tap*Cin + ci).
//   A[m][k]  is gathered on the fly from the NHWC input (zero outside the image),
//   B[n][k]  is the OHWI weight matrix ([Cout][9*Cin]), so both LDS tiles are [row][k] with k
//            contiguous and every lane fetches 16 bytes of k with one ds_read_b128.
// For pooled layers the M index is ordered m = 4*window + (dy*2+dx): the four pixels of one
// 2x2 pool window are the four accumulator registers (reg&3) one lane already holds for one
// channel (MFMA 32x32 C/D layout row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)), so the max-pool is
// three v_max in registers and the pre-pool tensor never touches HBM.
//
// Block: 256 threads = 4 waves (WAVES_M x WAVES_N), each wave WM x WN tiles of 32x32.
//
// Grid quantisation.  Every tile of a layer costs the same, two to four tiles share a CU, and the tile
// count of this network is always 49 * 2^k: a plain grid leaves up to a quarter of the chip idle in
// its last round (784 tiles on 256 CUs = 3.06 rounds).  A layer is therefore run as
//   (1) the largest multiple of the CU count of whole tiles, fused epilogue, and in the same launch
//   (2) the remaining R < #CU tiles split s ways along K (R*s ~ a multiple of #CU short pieces that
//       write raw accumulator tiles to a caller-provided workspace), then
//   (3) a small fix-up kernel that sums the s partials and applies BN/LeakyReLU/max-pool.
// plan_tail() picks s with a cost model in units of one K chunk; without a workspace the layer runs
// as whole tiles only (same results up to summation order, slower tail).
#include "common.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#ifndef TY1_STATIC_SCRATCH_BYTES
#define TY1_STATIC_SCRATCH_BYTES (16u << 20)      // ty1_conv3x3_bn_leaky_static_f32: 16 MiB per GPU
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;

// one conv problem as the dispatcher hands it to a tile configuration (x / w / y hold bf16 data behind their float
// pointers in the bf16 network)
struct ConvArgs {
    const float *x;      // NHWC input
    const float *w;      // [Cout][9*Cin] weights, k contiguous
    const float *scale;  // folded BN, per output channel
    const float *shift;
    float *y;
    int NB, H, W, Cin, Cout;
    int M;               // GEMM rows = pre-pool pixels
    float alpha;
};

// LDS bytes up to which a tile configuration takes three stages instead of two.  Measured (batch 64, warm clocks): the
// third stage buys nothing (the fetch is not latency-exposed) and costs a resident workgroup on the small tiles
// (1-2 % slower), so the ring is two stages deep unless a build overrides this.
#ifndef TY1_NS3_LIMIT
#define TY1_NS3_LIMIT 0
#endif
#ifdef TY1_DEV_STAMPS
#define TY1_DBG(bit) (a.dbg & (bit))
#else
#define TY1_DBG(bit) false
#endif

using ty1::g_stamp;
using ty1::g_stamp_cap;

// Bijective XCD-aware remap: blocks b and b+8 share an XCD (private L2), so give every XCD a
// contiguous range of logical tile ids (neighbouring tiles share halo rows and weight panels).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// ------------------------------------------------------------------------------------------------
// Persistent conv kernel.  The grid is as many workgroups as stay resident (two to four per CU); workgroup b walks the work items
// slot(b), slot(b) + G, slot(b) + 2G ... (item = output tile x K-split piece) as ONE continuous stream of
// K chunks: the fetch cursor runs NS-1 chunks ahead of the MFMAs and simply crosses into the next item,
// so a new tile starts with its operands already in LDS - no per-tile prologue (address arithmetic +
// an exposed HBM round trip, 4-10 us per tile when every tile was its own workgroup) and no drain.
// Row geometry (image, y, x of every GEMM row of a tile, 3x3 validity mask) is recomputed per item with
// multiply-shift division.
// Software pipeline of one wave (the matrix pipe should never wait for this wave's own memory work).  Per chunk, right
// after the barrier that published it:
//   1. read the fragments of its first k-block
//   2. issue the LDS-DMA fetch of the NEXT chunk into the stage that was read last (between the first MFMA groups;
//      it costs the matrix pipe nothing and has the whole chunk to land)
//   3. the second half of the previous chunk's last k-block (8 MFMAs kept back across the barrier)
//   4. k-blocks 0..NKB-1 with the fragments of k-block kb+1 read while k-block kb multiplies
//   5. counted vmcnt wait + raw barrier with half of the last k-block still to issue; 6. cursor step (the only branch).
// The body is the same for every chunk (no peeling: peeled copies make the register allocator shuffle the 64
// accumulator registers around the loop); an item's first pass multiplies a zero fragment in step 3.
// ------------------------------------------------------------------------------------------------
struct FastDiv {         // q = n / d for 0 <= n < 2^31:  (mulhi(n, M) + n) >> l
    unsigned M, l;
};
FastDiv make_fastdiv(unsigned d) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    return FastDiv{(unsigned)(((1ull << (32 + l)) / d) + 1ull - (1ull << 32)), l};
}
__device__ __forceinline__ unsigned fdiv(unsigned n, FastDiv f) { return (__umulhi(n, f.M) + n) >> f.l; }

// Element type of the GEMM operands (activations and weights).  A K chunk is always BK*4 bytes of a row (8 or 4
// 16-byte pieces), so fetch and LDS layout do not depend on it; only the fragment -> MFMA step does:
// a 16-byte LDS read holds 4 floats = the operands of four v_mfma_f32_32x32x2_f32 (k = 2 each, lane half = k
// parity), or 8 bf16 = the operand of ONE v_mfma_f32_32x32x16_bf16 (lane half = k 8h..8h+7).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
struct OpF32 {
    static constexpr int ESZ = 4;
    typedef f32x4v vec;
};
struct OpBF16 {
    static constexpr int ESZ = 2;
    typedef bf16x8 vec;
};
__device__ __forceinline__ unsigned short bf16_bits(float v) {     // round to nearest even (v_cvt_pk_bf16_f32)
    const __bf16 h = (__bf16)v;
    return __builtin_bit_cast(unsigned short, h);
}

struct PersistArgs {
    const void *x, *w;   // activations NHWC and weights OHWI, f32 or bf16
    const float *scale, *shift;
    void *y;             // output tensor (f32 or bf16)
    float *part;         // partial tiles of the K-split tail: [split][tail tile][BM][BN]
    int NB, H, W, Cin, Cout, M;
    int chunks_total;
    float alpha;
    int ntiles, mtiles;  // tiles along N and M
    // Tile order.  An XCD walks a contiguous range of logical tiles and its resident tiles stream K in step, so
    // tiles that share an operand panel should be neighbours: nt fastest = the tiles of one A panel side by side
    // (each B panel is then shared by resident/ntiles of them).  mt fastest is kept as an option only.
    int mt_fastest;
    // work items: logical tiles [0, whole) with the fused epilogue, then the last `rem` tiles cut into `splits`
    // K pieces each (item whole + j: tile whole + j / splits, piece j % splits) that store raw partial tiles
    int whole, rem, splits;
    FastDiv d_img, d_row;   // pooled layers: (H/2)*(W/2), W/2; otherwise H*W, W
    FastDiv d_splits, d_ntiles, d_mtiles, d_cpt;   // splits, ntiles, mtiles, chunks per tap
    long long *stamp;
    int dbg;             // developer knock-outs (dev build only): 1 drop A fetches, 8 drop B fetches, 4 skip barriers, 16 coalesced (wrong) fetch pattern
};

// BK counts 4-byte units of a row's K chunk (BK * 4 bytes = BKE elements of type OP); OUT_BF16: the fused epilogue
// stores bf16 (raw partial tiles of the K-split tail are always f32).
template <int WM, int WN, int WAVES_M, int WAVES_N, int BK, bool POOL, typename OP = OpF32, bool OUT_BF16 = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void conv_persist_kernel(const PersistArgs a) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
    // LDS image of a stage: NKB planes (one per k-block of 32 bytes of K), a plane = [BM rows of A | BN rows of B] x 32
    // bytes + 64 bytes of padding.  A row's two 16-byte halves are swapped when bit 3 of the row index is set.  With
    // that, a fragment read is base + kb * PLANE + 1024 * i (pure immediates: the loop computes NO LDS address; every
    // vector instruction between two f32 MFMAs costs the matrix pipe its own issue cycles, measured 8.5 cycles per
    // VALU op, 13 per ds_write_b128, 17 per buffer load: tools/ubench/mfma_valu.hip), the ds_read_b128 16-lane groups
    // cover all 64 banks, and so do the 8-lane groups of the stores (plane stride = 16 banks mod 64).
    constexpr int PLANE = (BM + BN) * 32 + 64, STAGE = (BK / 8) * PLANE;
    constexpr int NKB = BK / 8;
    // Operand fetch is LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write, and - unlike a load
    // into VGPRs (17 cycles) or a ds_write_b128 (13) - no cost to the matrix pipe.  One wave-instruction fills 1 KiB =
    // 32 rows of one plane (lane l: row l/2, 16-byte half l&1).  The (BM/32 + BN/32) * NKB pieces of a chunk are dealt to
    // the four waves so that a wave keeps to its own row groups: piece q = wave + 4t -> row group q % G, plane q / G.
    constexpr int GA = BM / 32, GB = BN / 32;                          // row groups
    constexpr int NGA = GA >= 4 ? GA / 4 : 1, NGB = GB >= 4 ? GB / 4 : 1;   // row groups a wave touches
    constexpr int TA = (GA * NKB + 3) / 4, TB = (GB * NKB + 3) / 4;   // pieces per wave and chunk
    constexpr int ESZ = OP::ESZ, BKE = BK * 4 / ESZ;       // element size, elements per chunk
    constexpr int OSZ = OUT_BF16 ? 2 : 4;                  // bytes per output element
    constexpr unsigned kOOB = 0xFFFFFFFFu;
    static_assert(NKB == 2 || NKB == 4, "BK is 16 or 32");

    // NS stages: the fetch cursor runs NS-1 chunks ahead of the MFMAs (see TY1_NS3_LIMIT above).
    constexpr int NS = 3 * STAGE <= TY1_NS3_LIMIT ? 3 : 2;
    constexpr int PW = TA + TB;                 // DMA instructions per wave and chunk
    __shared__ __attribute__((aligned(16))) char lds[NS * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    // (dev knock-out 16: timing-only fetch pattern of 8 rows x 128 contiguous bytes per DMA instruction)
    const int dma_row = TY1_DBG(16) ? lane >> 3 : lane >> 1;
    const int dma_half = TY1_DBG(16) ? (lane & 7) : (lane & 1) ^ ((lane >> 4) & 1);   // logical half this lane fetches
    const int G = gridDim.x;
    const int items = a.whole + a.rem * a.splits;
    if (a.stamp && tid == 0) {
        a.stamp[(long long)blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memrealtime();
        a.stamp[(long long)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        a.stamp[(long long)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }

    const int chunks_per_tap = a.Cin / BKE;
    const long long x_total = (long long)a.NB * a.H * a.W * a.Cin;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void *>(a.w), 0, TY1_DBG(8) ? 0 : (int)((long long)a.Cout * 9 * a.Cin * ESZ), 0x00020000);

    // ---- item descriptors (block-uniform scalars)
    struct Desc {
        int mt, nt, kc0, kc1, tap0, ci0, pslot;
        bool live, raw;
    };
    auto decode = [&](int it) {
        Desc d;
        d.live = it < items;
        d.raw = d.live && it >= a.whole;
        const int j = d.raw ? it - a.whole : 0;
        const int tail_tile = (int)fdiv((unsigned)j, a.d_splits);
        const int split = j - tail_tile * a.splits;
        const int logical = d.raw ? a.whole + tail_tile : (d.live ? it : 0);
        if (a.mt_fastest) {                       // logical = nt * mtiles + mt
            d.nt = (int)fdiv((unsigned)logical, a.d_mtiles);
            d.mt = logical - d.nt * a.mtiles;
        } else {                                  // logical = mt * ntiles + nt
            d.mt = (int)fdiv((unsigned)logical, a.d_ntiles);
            d.nt = logical - d.mt * a.ntiles;
        }
        d.kc0 = d.raw ? a.chunks_total * split / a.splits : 0;
        d.kc1 = d.raw ? a.chunks_total * (split + 1) / a.splits : (d.live ? a.chunks_total : 0x7fffffff);
        d.tap0 = (int)fdiv((unsigned)d.kc0, a.d_cpt);
        d.ci0 = (d.kc0 - d.tap0 * chunks_per_tap) * BKE;
        d.pslot = split * a.rem + tail_tile;
        // block-uniform by construction: keep them in scalar registers
        d.mt = __builtin_amdgcn_readfirstlane(d.mt); d.nt = __builtin_amdgcn_readfirstlane(d.nt);
        d.kc0 = __builtin_amdgcn_readfirstlane(d.kc0); d.kc1 = __builtin_amdgcn_readfirstlane(d.kc1);
        d.tap0 = __builtin_amdgcn_readfirstlane(d.tap0); d.ci0 = __builtin_amdgcn_readfirstlane(d.ci0);
        d.pslot = __builtin_amdgcn_readfirstlane(d.pslot);
        return d;
    };
    // ---- per-item row geometry of this thread's fetch rows (buffer offsets relative to `origin`)
    //      A rows: byte offset of the pixel's channel 0 (+ this lane's 16-byte half) and the 9-bit set of taps that
    //      fall OUTSIDE the image; B rows: byte offset of the weight row (+ half), kOOB for rows past Cout.  The part of
    //      an address that changes from chunk to chunk (tap shift, channel / k position) is block-uniform and travels in
    //      the loads' scalar offset.
    struct Rows {
        unsigned a_off[NGA], a_inv[NGA], b_off[NGB];
    };
    auto make_rows = [&](const Desc &d, Rows &R, long long &origin) {
        const int m_first = d.mt * BM;
        {
            const unsigned img = __builtin_amdgcn_readfirstlane(fdiv(POOL ? (unsigned)m_first >> 2 : (unsigned)m_first, a.d_img));
            origin = (long long)img * a.H * a.W * a.Cin;
        }
#pragma unroll
        for (int p = 0; p < NGA; ++p) {
            const int m = m_first + 32 * (GA >= 4 ? wave + 4 * p : wave % GA) + dma_row;
            R.a_off[p] = kOOB;
            R.a_inv[p] = 0x1FFu;
            if (d.live && m < a.M) {
                unsigned n, yy, xx;
                if (POOL) {
                    const unsigned win = (unsigned)m >> 2, q = m & 3;
                    n = fdiv(win, a.d_img);
                    const unsigned rem = win - n * ((a.H >> 1) * (a.W >> 1));
                    const unsigned py = fdiv(rem, a.d_row), px = rem - py * (a.W >> 1);
                    yy = 2 * py + (q >> 1);
                    xx = 2 * px + (q & 1);
                } else {
                    n = fdiv((unsigned)m, a.d_img);
                    const unsigned rem = m - n * (a.H * a.W);
                    yy = fdiv(rem, a.d_row);
                    xx = rem - yy * a.W;
                }
                const long long el = (((long long)n * a.H + yy) * a.W + xx) * a.Cin - origin;
                R.a_off[p] = (unsigned)(el * ESZ) + (unsigned)dma_half * 16u;
                // tap t = 3*(dy+1) + (dx+1) is inside the image: product of a row condition and a column condition
                const unsigned rows = (yy > 0 ? 0x007u : 0u) | 0x038u | ((int)yy < a.H - 1 ? 0x1C0u : 0u);
                const unsigned cols = (xx > 0 ? 0x049u : 0u) | 0x092u | ((int)xx < a.W - 1 ? 0x124u : 0u);
                R.a_inv[p] = ~(rows & cols) & 0x1FFu;
            }
        }
#pragma unroll
        for (int p = 0; p < NGB; ++p) {
            const int n = d.nt * BN + 32 * (GB >= 4 ? wave + 4 * p : wave % GB) + dma_row;
            R.b_off[p] = (d.live && n < a.Cout) ? (unsigned)((long long)n * 9 * a.Cin * ESZ) + (unsigned)dma_half * 16u : kOOB;
        }
    };

    // Item of this block in round r.  Full rounds of whole tiles use the XCD-aware slot (each XCD walks a
    // contiguous range of tiles).  What is left - the last whole tiles, then the K pieces of the tail - is dealt
    // to blocks 0, 1, 2 ... in launch order (round after round), which the dispatcher spreads over all XCDs and
    // CUs: the blocks that get no whole tile in the partial round start on the tail pieces instead.
    const int slot = xcd_remap(blockIdx.x, G);
    const int full_rounds = a.whole / G, full_items = full_rounds * G;
    auto item_of = [&](int round) {
        if (round < full_rounds) return round * G + slot;
        const int v = full_items + (round - full_rounds) * G + (int)blockIdx.x;
        return v < items ? v : items;
    };
    Desc cur = decode(item_of(0)), nxt = decode(items);
    Rows lr;                                  // rows of the item the fetch cursor is in
    long long org_l;
    make_rows(cur, lr, org_l);

    // ---- BN scale / shift of an item's channels (lane holds channel lane&31 of each tile column).  They are
    //      fetched when the fetch cursor ENTERS the item, i.e. a chunk before its MFMAs start.
    float sc_cur[WN], sh_cur[WN], sc_nxt[WN], sh_nxt[WN];
    auto load_bn = [&](const Desc &d, float (&sc)[WN], float (&sh)[WN]) {
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = d.nt * BN + wave_n * (32 * WN) + 32 * j + (lane & 31);
            const bool ok = d.live && !d.raw && n < a.Cout;
            sc[j] = ok ? a.scale[n] : 0.f;
            sh[j] = ok ? a.shift[n] : 0.f;
        }
    };

    // (marker: tells the compiler that sc_nxt / sh_nxt are needed here, right after the per-chunk wait, where its own
    //  wait for them costs nothing; without it the wait would sit in the epilogue and drain the fetches in flight)
    auto landed = [&]() {
#pragma unroll
        for (int j = 0; j < WN; ++j) asm volatile("" : "+v"(sc_nxt[j]), "+v"(sh_nxt[j]));
    };

    // ---- fetch cursor: NS-1 chunks ahead of the MFMAs.  When it reaches the end of its item it moves on to this
    //      block's next item (a block-uniform branch taken once per item: new descriptor, new row geometry).
    int ld_round = 0, ld_kc = cur.kc0, ld_end = cur.kc1, ld_tap = cur.tap0, ld_ci0 = cur.ci0;
    // One chunk fetch = scalar set-up (SALU) + the tap mask of this wave's A rows (one v_bfe_i32 + one v_or per row
    // group: tap outside the image -> offset 0xFFFFFFFF -> the range check makes the DMA write zeros) + TA + TB DMA
    // instructions whose destination (stage, plane, row group) is a scalar in M0.
    // The A descriptor is based (W+1) pixels BEFORE the first image of the item so that the scalar offset of every tap
    // is non-negative; it is rebuilt when the cursor enters an item.
    const int a_shift = (a.W + 1) * a.Cin * ESZ;
    __amdgpu_buffer_rsrc_t a_rsrc;
    auto make_a_rsrc = [&]() {
        // (the range check of this hardware counts the scalar offset too - measured: a descriptor sized from the item
        // origin dropped the last rows of the tensor - so the record count runs from the shifted base to the tensor end)
        const long long left = (x_total - org_l) * ESZ + a_shift;
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(static_cast<const char *>(a.x)) + org_l * ESZ - a_shift, 0,
            TY1_DBG(1) ? 0 : (int)(left > 0xFFFFFFFFll ? 0xFFFFFFFFll : left), 0x00020000);
    };
    int ld_soff_a = 0, ld_soff_b = 0;
    unsigned a_moff[NGA];
    auto load_prep = [&]() {
        const int t3 = ld_tap / 3;
        const int dy = t3 - 1, dx = ld_tap - 3 * t3 - 1;
        ld_soff_a = a_shift + (((dy * a.W + dx) * a.Cin) + ld_ci0) * ESZ;
        ld_soff_b = ld_kc * (BK * 4);
#pragma unroll
        for (int p = 0; p < NGA; ++p)
            a_moff[p] = lr.a_off[p] | (unsigned)__builtin_amdgcn_sbfe((int)lr.a_inv[p], (unsigned)ld_tap, 1u);
    };
    typedef __attribute__((address_space(3))) void *lds_ptr;
    auto dma_A = [&](int stage_off, int t0, int t1) {
#pragma unroll
        for (int t = t0; t < t1; ++t)
            if (t < TA && (4 * t + 3 < GA * NKB || wave + 4 * t < GA * NKB)) {
                const int q = wave + 4 * t, g = q % GA, kb = q / GA;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr)(lds + stage_off + kb * PLANE + g * 1024), 16,
                                                         (int)a_moff[GA >= 4 ? t % NGA : 0], ld_soff_a + kb * 32, 0, 0);
            }
    };
    auto dma_B = [&](int stage_off, int t0, int t1) {
#pragma unroll
        for (int t = t0; t < t1; ++t)
            if (t < TB && (4 * t + 3 < GB * NKB || wave + 4 * t < GB * NKB)) {
                const int q = wave + 4 * t, g = q % GB, kb = q / GB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (lds_ptr)(lds + stage_off + kb * PLANE + (BM + 32 * g) * 32), 16,
                                                         (int)lr.b_off[GB >= 4 ? t % NGB : 0], ld_soff_b + kb * 32, 0, 0);
            }
    };
    // Cursor step, kept apart from the loads: its branch sits at the very end of the main loop body, so that body is
    // ONE basic block and the scheduler can spread the address arithmetic and the loads between the MFMAs.
    auto advance = [&]() {
        ++ld_kc;
        ld_ci0 += BKE;
        const bool wrap = ld_ci0 == a.Cin;
        ld_ci0 = wrap ? 0 : ld_ci0;
        ld_tap += wrap ? 1 : 0;
        if (ld_kc == ld_end) {
            ++ld_round;
            nxt = decode(item_of(ld_round));
            make_rows(nxt, lr, org_l);
            make_a_rsrc();
            load_bn(nxt, sc_nxt, sh_nxt);
            ld_kc = nxt.kc0;
            ld_end = nxt.kc1;
            ld_tap = nxt.tap0;
            ld_ci0 = nxt.ci0;
        }
    };
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_row = lane & 31, frag_h = lane >> 5;
    const int a_row0 = wave_m * (32 * WM) + frag_row;
    const int b_row0 = wave_n * (32 * WN) + frag_row;
    unsigned rd_a = (unsigned)(a_row0 * 32 + ((frag_h ^ ((a_row0 >> 3) & 1)) << 4));
    unsigned rd_b = (unsigned)(BM * 32 + b_row0 * 32 + ((frag_h ^ ((b_row0 >> 3) & 1)) << 4));
    typedef typename OP::vec fvec;
    struct Frag {
        fvec a[WM], b[WN];
    };
    auto read_frag = [&](Frag &f, int kb) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
            f.a[i] = *reinterpret_cast<const fvec *>(__builtin_assume_aligned(&lds[rd_a + kb * PLANE + 1024 * i], 16));
#pragma unroll
        for (int j = 0; j < WN; ++j)
            f.b[j] = *reinterpret_cast<const fvec *>(__builtin_assume_aligned(&lds[rd_b + kb * PLANE + 1024 * j], 16));
    };
    // f32: k0..k1 of the four k pairs a fragment holds.  bf16: the fragment is one MFMA, issued with the first half.
    auto mfma = [&](const Frag &f, int k0, int k1) {
        if constexpr (ESZ == 4) {
#pragma unroll
            for (int k = k0; k < k1; ++k)        // k outermost: consecutive MFMAs go to different accumulators
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][k], f.b[j][k], acc[i][j], 0, 0, 0);
        } else {
            if (k0 == 0) {
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i], f.b[j], acc[i][j], 0, 0, 0);
            }
        }
    };
    Frag f0, f1;
    auto SB = [] { __builtin_amdgcn_sched_barrier(0); };
    constexpr int AH = (TA + 1) / 2, BH = (TB + 1) / 2;   // the fetch of a chunk is issued in four pieces
    auto zero_f1 = [&]() {
        const f32x4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < WN; ++j) f1.b[j] = __builtin_bit_cast(fvec, z);
#pragma unroll
        for (int i = 0; i < WM; ++i) f1.a[i] = __builtin_bit_cast(fvec, z);
    };

    // per-lane epilogue constants that do not depend on the item (scale / shift depend on nt only)
    const int h = lane >> 5;

    load_bn(cur, sc_cur, sh_cur);
#pragma unroll
    for (int j = 0; j < WN; ++j) { sc_nxt[j] = 0.f; sh_nxt[j] = 0.f; }
    make_a_rsrc();
#pragma unroll
    for (int st = 0; st < NS - 1; ++st) {
        load_prep();
        dma_A(st * STAGE, 0, TA);
        dma_B(st * STAGE, 0, TB);
        advance();
    }
    // wait for the first chunk only: the PW * (NS - 2) newest DMAs of this wave may stay in flight
    // (a raw barrier: __syncthreads() would drain every DMA in flight - its fence waits vmcnt(0) - and there are no
    //  ds_writes to publish; lgkmcnt(0) retires this wave's reads of the stage that the others refill next)
    auto wg_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto wait_chunk = [] { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (NS - 2)) : "memory"); };
    wait_chunk();
    wg_barrier();
    zero_f1();
    int wr = (NS - 1) * STAGE;                // stage being filled (scalar byte offset)
    int rd_stage = 0;                         // stage being read
    read_frag(f0, 0);

    long long t_loop = 0, t_epi = 0, t_rows = 0, n_items = 0;      // developer stamps (only when a.stamp)
    while (cur.live) {
        long long t0 = 0, t1 = 0, t2 = 0;
        if (a.stamp) t0 = __builtin_amdgcn_s_memtime();
        for (int kc = cur.kc0; kc < cur.kc1; ++kc) {
            // The order below is pinned with scheduling barriers: left to itself the compiler sinks the fragment reads
            // behind the MFMAs that precede them in order to reuse their registers, which exposes one LDS round trip
            // per k-block.  The fetch of the NEXT chunk goes out first: the stage it fills was last read before the
            // barrier that ended the previous iteration, and it has this whole chunk to land.
            load_prep();
            SB();
            dma_A(wr, 0, AH);
            SB();
            mfma(f1, 2, 3);                       // second half of the previous chunk's last k-block (zeros at an item start)
            SB();
            dma_A(wr, AH, TA);
            SB();
            mfma(f1, 3, 4);
            SB();
            read_frag(f1, 1);
            SB();
            mfma(f0, 0, 1);
            SB();
            dma_B(wr, 0, BH);
            SB();
            mfma(f0, 1, 2);
            SB();
            dma_B(wr, BH, TB);
            SB();
            mfma(f0, 2, 3);
            SB();
            mfma(f0, 3, 4);
            SB();
            if (NKB == 4) {
                read_frag(f0, 2);
                SB();
                mfma(f1, 0, 1);
                SB();
                mfma(f1, 1, 2);
                SB();
                mfma(f1, 2, 3);
                SB();
                mfma(f1, 3, 4);
                SB();
                read_frag(f1, 3);
                SB();
                mfma(f0, 0, 1);
                SB();
                mfma(f0, 1, 2);
                SB();
                mfma(f0, 2, 3);
                SB();
                mfma(f0, 3, 4);
                SB();
            }
            mfma(f1, 0, 1);
            SB();
            mfma(f1, 1, 2);
            // every wave waits for its own DMA pieces of the next chunk, the barrier then publishes that stage
            wait_chunk();
            landed();
            __builtin_amdgcn_sched_barrier(0);
            if (!TY1_DBG(4)) wg_barrier();
            __builtin_amdgcn_sched_barrier(0);
            {                                     // ring step: scalar except the two fragment read bases
                wr = rd_stage * STAGE;            // the stage just read is the next one to fill
                const bool wrap = rd_stage == NS - 1;
                const int step = wrap ? -(NS - 1) * STAGE : STAGE;
                rd_stage = wrap ? 0 : rd_stage + 1;
                rd_a += step;
                rd_b += step;
            }
            read_frag(f0, 0);                     // first k-block of the next chunk (of the next item at an item end)
            advance();
        }
        mfma(f1, 2, 4);
        if (a.stamp) t1 = __builtin_amdgcn_s_memtime();

        // ---- epilogue of item `cur`.  Lane holds channel (lane&31) of tile column j; register r is tile row
        //      (r&3) + 8*(r>>2) + 4*(lane>>5).
        //      Stores go through a buffer resource based at the tile's first output row: one address VGPR per
        //      tile column, rows past the end of the tensor fall outside the resource and are dropped.
        if (cur.raw) {
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                a.part + (long long)cur.pslot * (BM * BN), 0, BM * BN * 4, 0x00020000);
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int col = wave_n * (32 * WN) + 32 * j + (lane & 31);
                const unsigned vb = (unsigned)(((wave_m * (32 * WM) + 4 * h) * BN + col) * 4);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[i][j][r];     // (bit_cast straight from the vector element stored element 4*(r/4))
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), o_rsrc,
                                                              (int)(vb + (unsigned)((32 * i + (r & 3) + 8 * (r >> 2)) * BN * 4)), 0, 0);
                    }
            }
        } else {
            const int rows_out = POOL ? a.M >> 2 : a.M;
            const int r0 = POOL ? cur.mt * (BM / 4) : cur.mt * BM;
            const long long obytes = (long long)(rows_out - r0) * a.Cout * OSZ;
            const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                static_cast<char *>(a.y) + (long long)r0 * a.Cout * OSZ, 0,
                (int)(obytes > 0xFFFFFFFFll ? 0xFFFFFFFFll : obytes), 0x00020000);
            const unsigned row_bytes = (unsigned)a.Cout * (unsigned)OSZ;
            const bool pool_first = a.alpha >= 0.f;
            auto put = [&](float v, unsigned off) {
                if constexpr (OUT_BF16) __builtin_amdgcn_raw_buffer_store_b16(bf16_bits(v), o_rsrc, (int)off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), o_rsrc, (int)off, 0, 0);
            };
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int n = cur.nt * BN + wave_n * (32 * WN) + 32 * j + (lane & 31);
                const bool n_ok = n < a.Cout;
                if (POOL) {
                    const unsigned vb = (unsigned)(wave_m * (8 * WM) + h) * row_bytes + (unsigned)n * (unsigned)OSZ;
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            // LeakyReLU with alpha >= 0 is non-decreasing, so it commutes with the window maximum:
                            // pool the four BN outputs first, activate once (bit-identical; alpha < 0 keeps the
                            // reference order).  Every VALU instruction here costs the co-resident workgroup
                            // matrix-pipe cycles.
                            float v[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] = acc[i][j][4 * g + q] * sc_cur[j] + sh_cur[j];
                            float pooled;
                            if (pool_first) {
                                const float t = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                                pooled = t > 0.f ? t : t * a.alpha;
                            } else {
#pragma unroll
                                for (int q = 0; q < 4; ++q) v[q] = v[q] > 0.f ? v[q] : v[q] * a.alpha;
                                pooled = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                            }
                            const unsigned off = n_ok ? vb + (unsigned)(8 * i + 2 * g) * row_bytes : kOOB;
                            put(pooled, off);
                        }
                } else {
                    const unsigned vb = (unsigned)(wave_m * (32 * WM) + 4 * h) * row_bytes + (unsigned)n * (unsigned)OSZ;
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float t = acc[i][j][r] * sc_cur[j] + sh_cur[j];
                            t = t > 0.f ? t : t * a.alpha;
                            const unsigned off = n_ok ? vb + (unsigned)(32 * i + (r & 3) + 8 * (r >> 2)) * row_bytes : kOOB;
                            put(t, off);
                        }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        zero_f1();
        if (a.stamp) t2 = __builtin_amdgcn_s_memtime();

        // ---- next item becomes current; the fetch cursor is already NS-1 chunks inside it
        cur = nxt;
#pragma unroll
        for (int j = 0; j < WN; ++j) { sc_cur[j] = sc_nxt[j]; sh_cur[j] = sh_nxt[j]; }
        if (a.stamp) {
            const long long t3 = __builtin_amdgcn_s_memtime();
            t_loop += t1 - t0; t_epi += t2 - t1; t_rows += t3 - t2; ++n_items;
            if (tid == 0 && blockIdx.x < 8 && n_items < 64) {     // per-item clock trace of the first blocks (rows 4096...)
                a.stamp[(4096 + blockIdx.x * 64 + n_items) * 8 + 0] = t3;
                a.stamp[(4096 + blockIdx.x * 64 + n_items) * 8 + 1] = __builtin_amdgcn_s_memrealtime();
            }
        }
    }
    if (a.stamp && tid == 0) {
        a.stamp[(long long)blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
        a.stamp[(long long)blockIdx.x * 8 + 1] = t_loop;
        a.stamp[(long long)blockIdx.x * 8 + 2] = t_epi;
        a.stamp[(long long)blockIdx.x * 8 + 6] = t_rows;
        a.stamp[(long long)blockIdx.x * 8 + 7] = n_items;
    }
}

// Fix-up of the K-split tail tiles: y = [pool](leaky(sum_s part[s] * scale + shift)).
// part is [splits][tiles][BM][BN] (tile-local rows, pooled layers in window-major order m = 4*window + q).
// One block per tile; a thread owns four consecutive channels of one output row (pool: of one window).
template <int BM, int BN, bool POOL, bool OUT_BF16 = false>
__global__ __launch_bounds__(256) void conv_fixup_kernel(const float *__restrict__ part, const float *__restrict__ scale,
                                                         const float *__restrict__ shift, void *__restrict__ yv,
                                                         int M, int Cout, int ntiles, int mtiles, int mt_fastest,
                                                         int tile_begin, int tiles, int splits, float alpha) {
    const int tile_local = blockIdx.x;
    const int logical = tile_begin + tile_local;
    const int mt = mt_fastest ? logical % mtiles : logical / ntiles;
    const int nt = mt_fastest ? logical / mtiles : logical % ntiles;
    constexpr int C4 = BN / 4;
    auto put4 = [&](size_t el, const float4 &o) {          // four consecutive channels starting at element `el`
        if constexpr (OUT_BF16) {
            uint2 pk;
            pk.x = (unsigned)bf16_bits(o.x) | ((unsigned)bf16_bits(o.y) << 16);
            pk.y = (unsigned)bf16_bits(o.z) | ((unsigned)bf16_bits(o.w) << 16);
            *reinterpret_cast<uint2 *>(static_cast<unsigned short *>(yv) + el) = pk;
        } else {
            *reinterpret_cast<float4 *>(static_cast<float *>(yv) + el) = o;
        }
    };
    const size_t split_stride = (size_t)tiles * BM * BN;
    const float *base = part + (size_t)tile_local * BM * BN;
    // a thread owns four consecutive channels of ONE GEMM row; the slabs are summed with four independent
    // accumulators so that the loads of a thread are all in flight together
    for (int idx = blockIdx.y * 256 + threadIdx.x; idx < BM * C4; idx += 256 * gridDim.y) {
        const int r = idx / C4, c = (idx - r * C4) * 4;
        const int n = nt * BN + c;
        const float *src = base + r * BN + c;
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        int s = 0;
        for (; s + 4 <= splits; s += 4) {
            const float4 p0 = *reinterpret_cast<const float4 *>(src + (size_t)s * split_stride);
            const float4 p1 = *reinterpret_cast<const float4 *>(src + (size_t)(s + 1) * split_stride);
            const float4 p2 = *reinterpret_cast<const float4 *>(src + (size_t)(s + 2) * split_stride);
            const float4 p3 = *reinterpret_cast<const float4 *>(src + (size_t)(s + 3) * split_stride);
            a0.x += p0.x; a0.y += p0.y; a0.z += p0.z; a0.w += p0.w;
            a1.x += p1.x; a1.y += p1.y; a1.z += p1.z; a1.w += p1.w;
            a2.x += p2.x; a2.y += p2.y; a2.z += p2.z; a2.w += p2.w;
            a3.x += p3.x; a3.y += p3.y; a3.z += p3.z; a3.w += p3.w;
        }
        for (; s < splits; ++s) {
            const float4 p0 = *reinterpret_cast<const float4 *>(src + (size_t)s * split_stride);
            a0.x += p0.x; a0.y += p0.y; a0.z += p0.z; a0.w += p0.w;
        }
        float4 v;
        v.x = (a0.x + a1.x) + (a2.x + a3.x); v.y = (a0.y + a1.y) + (a2.y + a3.y);
        v.z = (a0.z + a1.z) + (a2.z + a3.z); v.w = (a0.w + a1.w) + (a2.w + a3.w);
        if (n < Cout) {
            const float4 sc = *reinterpret_cast<const float4 *>(scale + n), sh = *reinterpret_cast<const float4 *>(shift + n);
            v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
            v.x = v.x > 0.f ? v.x : v.x * alpha; v.y = v.y > 0.f ? v.y : v.y * alpha;
            v.z = v.z > 0.f ? v.z : v.z * alpha; v.w = v.w > 0.f ? v.w : v.w * alpha;
        }
        if (POOL) {
            // rows 4w .. 4w+3 of a window sit in the four threads idx, idx + C4, idx + 2*C4, idx + 3*C4; C4 is 16 or 32
            // lanes, so with 256-thread blocks they are in different waves when C4 = 32 -> go through LDS
            __shared__ float4 red[256];
            red[threadIdx.x] = v;
            __syncthreads();
            // thread layout inside the block: t = rr * C4 + cc with rr = 256 / C4 consecutive rows (a multiple of 4)
            const int rr = threadIdx.x / C4;
            if ((rr & 3) == 0 && n < Cout) {
                const float4 v1 = red[threadIdx.x + C4], v2 = red[threadIdx.x + 2 * C4], v3 = red[threadIdx.x + 3 * C4];
                float4 o;
                o.x = fmaxf(fmaxf(v.x, v1.x), fmaxf(v2.x, v3.x)); o.y = fmaxf(fmaxf(v.y, v1.y), fmaxf(v2.y, v3.y));
                o.z = fmaxf(fmaxf(v.z, v1.z), fmaxf(v2.z, v3.z)); o.w = fmaxf(fmaxf(v.w, v1.w), fmaxf(v2.w, v3.w));
                const int win = mt * (BM / 4) + (r >> 2);
                if (win * 4 < M) put4((size_t)win * Cout + n, o);
            }
            __syncthreads();
        } else if (n < Cout) {
            const int m = mt * BM + r;
            if (m < M) put4((size_t)m * Cout + n, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// First layer (Cin = 3, K = 27 -> 28): direct convolution on v_mfma_f32_16x16x4_f32.
// A block owns a 16 x 64 pixel tile (pre-pool).  Its 18 x 66 x 3 input halo tile is staged ONCE in LDS
// with coalesced row loads (NHWC, C=3: a tile row is 198 contiguous floats) and all 27 taps of a pixel
// are read from it: for a fixed kh the 9 (kw,ci) values of a pixel are contiguous in LDS.  Each lane
// keeps its 7 weight values per 16-channel group in registers for the whole tile.  Each wave owns
// 4 rows x 64 px = 16 M-tiles of 16 pixels (= 4 pool windows x 4), two M-tiles in flight so the
// dependent-accumulator latency (40 cycles) hides behind the 32-cycle issue of the other chain.
// Operand maps (16x16x4): A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15];
// C/D: col = lane&15 (channel), row = 4*(lane>>4) + reg  ->  one lane holds one whole pool window.
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool POOL, int NT>
__global__ __launch_bounds__(kThreads) void conv3x3_c3_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ shift, float *__restrict__ y,
                                                              int NB, int H, int W, int Cout, float alpha) {
    constexpr int TH = 16, TW = 64;
    constexpr int ROWF = 3 * (TW + 2);      // 198 floats of input per tile row
    constexpr int ROW = 208;                // LDS row stride; 208 % 32 == 16 spreads the two pixel rows of a window
    __shared__ float tile[(TH + 2) * ROW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    int b = blockIdx.x;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int y0 = ty * TH, x0 = tx * TW;
    const int i16 = lane & 15, kq = lane >> 4;

    // weights of this lane: B[k = 4*ks + kq][channel 16*j + i16]; k = 27 is the zero pad
    float wreg[NT][7];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            const int k = 4 * ks + kq;
            wreg[j][ks] = k < 27 ? w[(16 * j + i16) * 27 + k] : 0.f;
        }
    // A-operand LDS offset of this lane for every k step, relative to the pixel's top-left tap
    int koff[7];
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
        const int k = 4 * ks + kq;
        const int kh = k / 9;
        koff[ks] = k < 27 ? kh * ROW + (k - 9 * kh) : 0;     // k = 27: any valid address (weight is 0)
    }

    // stage the halo tile: rows y0-1 .. y0+TH, float columns 3*(x0-1) .. 3*(x0+TW+1)-1, zero outside
    const float *img = x + (long long)n * H * W * 3;
    for (int i = tid; i < (TH + 2) * ROWF; i += kThreads) {
        const int rr = i / ROWF, cc = i - rr * ROWF;
        const int yy = y0 - 1 + rr, xx3 = (x0 - 1) * 3 + cc;
        float v = 0.f;
        if (yy >= 0 && yy < H && xx3 >= 0 && xx3 < W * 3) v = img[(long long)yy * W * 3 + xx3];
        tile[rr * ROW + cc] = v;
    }
    __syncthreads();

    float sc[NT], sh[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        sc[j] = scale[16 * j + i16];
        sh[j] = shift[16 * j + i16];
    }
    // pixel of this lane inside an M-tile: window (i16>>2) of 4, position q = i16&3 in the 2x2 window
    const int q = i16 & 3;
    const int lane_px = 2 * (i16 >> 2) + (q & 1), lane_py = q >> 1;

    for (int t2 = 0; t2 < 16; t2 += 2) {               // two M-tiles per iteration (independent chains)
        f32x4 acc[2][NT];
        int base[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = t2 + u;
            const int pr = t >> 3, wcol = (t & 7) * 4;           // pooled row in the wave's strip, first window
            const int py = 4 * wave + 2 * pr + lane_py, px = 2 * wcol + lane_px;
            base[u] = py * ROW + px * 3;
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[u][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < 7; ++ks) {
            const float a0 = tile[base[0] + koff[ks]];
            const float a1 = tile[base[1] + koff[ks]];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, wreg[j][ks], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, wreg[j][ks], acc[1][j], 0, 0, 0);
            }
        }
        // epilogue: this lane holds pool window (wcol + kq) of pooled row pr, channel 16j + i16
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = t2 + u;
            const int pr = t >> 3, win = (t & 7) * 4 + kq;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int co = 16 * j + i16;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = acc[u][j][r] * sc[j] + sh[j];
                    v[r] = z > 0.f ? z : z * alpha;
                }
                if (POOL) {
                    const int oy = (y0 >> 1) + 2 * wave + pr, ox = (x0 >> 1) + win;
                    if (oy < (H >> 1) && ox < (W >> 1))
                        y[(((long long)n * (H >> 1) + oy) * (W >> 1) + ox) * Cout + co] =
                            fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int oy = y0 + 4 * wave + 2 * pr + (r >> 1), ox = x0 + 2 * win + (r & 1);
                        if (oy < H && ox < W) y[(((long long)n * H + oy) * W + ox) * Cout + co] = v[r];
                    }
                }
            }
        }
    }
}

template <bool POOL>
void launch_c3(int nt, dim3 grid, hipStream_t s, const float *x, const float *w, const float *scale, const float *shift,
               float *y, int N, int H, int W, int Cout, float alpha) {
    dim3 block(kThreads);
    switch (nt) {
        case 1: conv3x3_c3_kernel<POOL, 1><<<grid, block, 0, s>>>(x, w, scale, shift, y, N, H, W, Cout, alpha); break;
        case 2: conv3x3_c3_kernel<POOL, 2><<<grid, block, 0, s>>>(x, w, scale, shift, y, N, H, W, Cout, alpha); break;
        case 3: conv3x3_c3_kernel<POOL, 3><<<grid, block, 0, s>>>(x, w, scale, shift, y, N, H, W, Cout, alpha); break;
        default: conv3x3_c3_kernel<POOL, 4><<<grid, block, 0, s>>>(x, w, scale, shift, y, N, H, W, Cout, alpha); break;
    }
}

using ty1::num_cus;

// How to run T equal tiles of C chunks each on `slots` CUs (see the header comment).  Costs are in chunk-times
// of one CU: a tile costs C, every block pays a fixed `ovh` (prologue, first fetch, epilogue), the two extra
// launches and the partial round trip cost `fix`.
struct TailPlan {
    int whole, rem, splits;
};
TailPlan plan_tail(int T, int C, int slots) {
    TailPlan p{T, 0, 1};
    static const char *off = ty1::dev_env("TY1_NO_TAIL_SPLIT");
    if (off || T % slots == 0 || C < 2) return p;
    // (measured with per-CU end-time stamps, 7x7x512->1024 at batch 64: 96 pieces left 160 CUs idle for the last 5 % of the
    //  kernel; the piece count that gives every CU the same share wins even though it is the largest split)
    const double ovh = 3.0, fix = 4.0, fix_per_split = 0.1;
    double best = std::ceil((double)T / slots) * (C + ovh);
    const int rem = T % slots, whole = T - rem;
    static const char *hyst = ty1::dev_env("TY1_TAIL_HYST");        // developer override: required gain per step (default 0.5 %)
    const double need = hyst ? atof(hyst) : 0.995;
    for (int s = 2; s <= 32 && C / s >= 3; ++s) {        // a piece keeps >= 3 chunks (fetch cursor of the persistent kernel)
        const int per = (C + s - 1) / s;
        const double t = (double)whole / slots * (C + ovh) + std::ceil((double)rem * s / slots) * (per + ovh) + fix + fix_per_split * s;
        if (t < need * best) {
            best = t;
            p = TailPlan{whole, rem, s};
        }
    }
    return p;
}

// One conv layer with tile configuration <WM, WN, WAVES_M, WAVES_N, BK>: one persistent launch over the whole
// tiles and the K pieces of the tail, then the fix-up of the tail tiles.
// ws_need != nullptr: size query only (returns the workspace bytes the plan wants).
// OP / OUT_BF16: element type of x and w / of y (c.x, c.w, c.y then carry bf16 data behind their float pointers).
template <int WM, int WN, int WAVES_M, int WAVES_N, int BK, bool POOL, typename OP = OpF32, bool OUT_BF16 = false>
int run_conv(const char *name, const ConvArgs &c, float *ws, size_t ws_bytes, size_t *ws_need, hipStream_t s) {
    constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
    constexpr int BKE = BK * 4 / OP::ESZ;                    // elements of K per chunk
    PersistArgs a{};
    a.x = c.x; a.w = c.w; a.scale = c.scale; a.shift = c.shift; a.y = c.y; a.part = ws;
    a.NB = c.NB; a.H = c.H; a.W = c.W; a.Cin = c.Cin; a.Cout = c.Cout; a.M = c.M;
    a.alpha = c.alpha;
    a.chunks_total = 9 * c.Cin / BKE;
    a.ntiles = (c.Cout + BN - 1) / BN;
    a.d_img = make_fastdiv(POOL ? (c.H / 2) * (c.W / 2) : c.H * c.W);
    a.d_row = make_fastdiv(POOL ? c.W / 2 : c.W);
    a.d_ntiles = make_fastdiv(a.ntiles);
    a.mtiles = (c.M + BM - 1) / BM;
    a.d_mtiles = make_fastdiv(a.mtiles);
    a.mt_fastest = 0;     // measured (rocprofv3 FETCH_SIZE, batch 64): mt-fastest raised L2 misses 1.5-5x on the deep layers
    a.d_cpt = make_fastdiv(c.Cin / BKE);
    const int T = ((c.M + BM - 1) / BM) * a.ntiles;
    TailPlan p = plan_tail(T, a.chunks_total, num_cus());
    const size_t need = (size_t)p.rem * p.splits * BM * BN * sizeof(float);
    if (ws_need) {
        *ws_need = p.rem ? need : 0;
        return 0;
    }
    if (p.rem && (!ws || ws_bytes < need)) p = TailPlan{T, 0, 1};     // no workspace: whole tiles only
    a.whole = p.whole; a.rem = p.rem; a.splits = p.splits;
    a.d_splits = make_fastdiv(p.splits);
    const int items = p.whole + p.rem * p.splits;
    // persistent grid: as many workgroups as stay resident together (LDS / register limited), at least 2 per CU
    static const int occ = [] {
        int n = 0;
        static const char *cap = ty1::dev_env("TY1_MAX_BLOCKS_PER_CU");
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_persist_kernel<WM, WN, WAVES_M, WAVES_N, BK, POOL, OP, OUT_BF16>,
                                                         64 * WAVES_M * WAVES_N, 0) != hipSuccess || n < 1)
            n = 2;
        if (cap && atoi(cap) > 0 && n > atoi(cap)) n = atoi(cap);
        return n > 4 ? 4 : n;
    }();
    const int grid = items < occ * ty1::grid_cus() ? items : occ * ty1::grid_cus();
    a.stamp = (g_stamp && grid <= g_stamp_cap) ? g_stamp : nullptr;
    static const char *dbg = ty1::dev_env("TY1_DBG");
    a.dbg = dbg ? atoi(dbg) : 0;
    // algorithmic work: 2*M*N*K flops; input + weights read once, output written once
    const double K = 9.0 * c.Cin, out_rows = POOL ? c.M / 4.0 : (double)c.M;
    char tag[96];
    {
        snprintf(tag, sizeof tag, "conv_%s_%s%s@%dx%d>%d", OP::ESZ == 4 ? "f32" : "bf16", name, POOL ? "_pool" : "", c.H, c.Cin, c.Cout);
        ty1::ProfScope prof(tag, s, 2.0 * c.M * c.Cout * K,
                            OP::ESZ * ((double)c.M * c.Cin + c.Cout * K) + (OUT_BF16 ? 2.0 : 4.0) * out_rows * c.Cout);
        conv_persist_kernel<WM, WN, WAVES_M, WAVES_N, BK, POOL, OP, OUT_BF16><<<grid, 64 * WAVES_M * WAVES_N, 0, s>>>(a);
        if (int rc = ty1::check_launch(name)) return rc;
    }
    if (p.rem) {
        snprintf(tag, sizeof tag, "conv_fixup@%dx%d>%d", c.H, c.Cin, c.Cout);
        ty1::ProfScope prof(tag, s, 0.0, (double)need + 4.0 * p.rem * BM * BN / (POOL ? 4 : 1));
        conv_fixup_kernel<BM, BN, POOL, OUT_BF16><<<dim3(p.rem, BM * BN / 4 / 256 < 32 ? BM * BN / 4 / 256 : 32), 256, 0, s>>>(ws, c.scale, c.shift, c.y, c.M, c.Cout, a.ntiles, a.mtiles, a.mt_fastest, p.whole,
                                                                       p.rem, p.splits, c.alpha);
        return ty1::check_launch("conv_fixup");
    }
    return 0;
}

template <int WM, int WN, int WAVES_M, int WAVES_N, int BK>
int run_conv_p(const char *name, bool pool, const ConvArgs &a, float *ws, size_t ws_bytes, size_t *ws_need, hipStream_t s) {
    return pool ? run_conv<WM, WN, WAVES_M, WAVES_N, BK, true>(name, a, ws, ws_bytes, ws_need, s)
                : run_conv<WM, WN, WAVES_M, WAVES_N, BK, false>(name, a, ws, ws_bytes, ws_need, s);
}

// Shared by the launch and the workspace query.  ws_need != nullptr: query only.
int conv3x3_dispatch(const float *x, const float *w, const float *scale, const float *shift, float *y, int N, int H, int W,
                     int Cin, int Cout, int pool, float alpha, float *ws, size_t ws_bytes, size_t *ws_need, hipStream_t s) {
    TY1_REQUIRE(N >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "ty1_conv3x3: bad dimensions");
    if (ws_need) *ws_need = 0;
    if (N == 0) return 0;
    TY1_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0), "ty1_conv3x3: pool needs even H, W (got %dx%d)", H, W);
    TY1_REQUIRE(Cout % 16 == 0, "ty1_conv3x3: Cout=%d must be a multiple of 16", Cout);
    TY1_REQUIRE((long long)N * H * W < (1ll << 31) - 256, "ty1_conv3x3: N*H*W must be below 2^31");
    if (!ws_need) TY1_REQUIRE(x && w && scale && shift && y, "ty1_conv3x3: null pointer");

    if (Cin == 3) {
        TY1_REQUIRE(Cout <= 64, "ty1_conv3x3: Cin=3 path supports Cout <= 64 (got %d)", Cout);
        if (ws_need) return 0;
        const int tiles = ((W + 63) / 64) * ((H + 15) / 16);
        TY1_REQUIRE((long long)N * tiles < (1ll << 31), "ty1_conv3x3: grid too large");
        dim3 grid((unsigned)(N * tiles));
        const double px = (double)N * H * W;
        ty1::ProfScope prof(pool ? "conv3x3_c3_pool" : "conv3x3_c3", s, 2.0 * px * Cout * 27,
                            4.0 * (px * 3 + 27.0 * Cout + px * Cout / (pool ? 4 : 1)));
        static const char *old_c3 = ty1::dev_env("TY1_OLD_C3");             // developer override
        // (the persistent kernel pools before it activates, which needs a non-decreasing LeakyReLU: alpha >= 0)
        if (W % 4 == 0 && ((uintptr_t)x & 15) == 0 && alpha >= 0.f && !old_c3)
            return ty1::conv3x3_c3_persistent(x, w, scale, shift, y, N, H, W, Cout, pool, alpha, 0, s);
        if (pool)
            launch_c3<true>(Cout / 16, grid, s, x, w, scale, shift, y, N, H, W, Cout, alpha);
        else
            launch_c3<false>(Cout / 16, grid, s, x, w, scale, shift, y, N, H, W, Cout, alpha);
        return ty1::check_launch("ty1_conv3x3(c3)");
    }

    TY1_REQUIRE(Cin % 16 == 0, "ty1_conv3x3: Cin=%d must be 3 or a multiple of 16", Cin);
    // 32-bit buffer offsets: a tile's rows span at most two images plus one halo row
    TY1_REQUIRE((2ll * H * W + 2ll * (W + 1)) * Cin * 4 < (1ll << 31), "ty1_conv3x3: one image (%dx%dx%d) is too large for 32-bit offsets", H, W, Cin);
    if (!ws_need)
        TY1_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)ws & 15) == 0,
                    "ty1_conv3x3: x, w and workspace must be 16-byte aligned");
    ConvArgs a{};
    a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.y = y;
    a.NB = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.M = N * H * W;
    a.alpha = alpha;
    const bool p = pool != 0;

    static const char *no_direct = ty1::dev_env("TY1_NO_DIRECT16");          // developer override
    // shallow K: direct kernel, weights in registers (it pools before it activates: needs alpha >= 0; one image < 2 GiB)
    if (Cin == 16 && (Cout == 32 || Cout == 64) && alpha >= 0.f && (long long)H * W * 64 < (1ll << 31) && !no_direct) {
        if (ws_need) return 0;
        return ty1::conv3x3_c16_direct(x, w, scale, shift, y, N, H, W, Cout, pool, alpha, s);
    }
    // tile choice: keep BN <= Cout; big tiles once they fill the chip, 64x64 for the 7x7 layers
    if (Cin % 32 != 0) {              // Cin == 16 (48, 80 ...): BK = 16
        if (Cout % 64 == 0) return run_conv_p<2, 1, 2, 2, 16>("bk16_128x64", p, a, ws, ws_bytes, ws_need, s);
        if (Cout % 32 == 0) return run_conv_p<2, 1, 4, 1, 16>("bk16_256x32", p, a, ws, ws_bytes, ws_need, s);
        return ty1::fail("ty1_conv3x3: Cin=%d needs Cout %% 32 == 0 (got %d)", Cin, Cout);
    }
    const long long mt128 = ((long long)a.M + 127) / 128;
    static const char *force = ty1::dev_env("TY1_TILE");          // developer override
    if (force && !strcmp(force, "128x128") && Cout % 128 == 0) return run_conv_p<2, 2, 2, 2, 32>("128x128", p, a, ws, ws_bytes, ws_need, s);
    if (force && !strcmp(force, "128x64") && Cout % 64 == 0) return run_conv_p<2, 1, 2, 2, 32>("128x64", p, a, ws, ws_bytes, ws_need, s);
    if (force && !strcmp(force, "64x64") && Cout % 64 == 0) return run_conv_p<1, 1, 2, 2, 32>("64x64", p, a, ws, ws_bytes, ws_need, s);
    if (Cout % 128 == 0 && mt128 * (Cout / 128) >= 2 * num_cus())
        return run_conv_p<2, 2, 2, 2, 32>("128x128", p, a, ws, ws_bytes, ws_need, s);
    if (Cout % 64 == 0) {
        if (mt128 * (Cout / 64) >= 2 * num_cus() || a.M >= 4096 * 2)
            return run_conv_p<2, 1, 2, 2, 32>("128x64", p, a, ws, ws_bytes, ws_need, s);
        // too few 64x64 tiles to give every CU one (7x7x1024->256 at batch 64: 196): the layer is all K-split tail either
        // way, and large tiles fetch half the bytes per MFMA (measured 127 -> 121 us; at M < 2048 the small tile still wins)
        const long long mt64 = ((long long)a.M + 63) / 64;
        if (Cout % 128 == 0 && Cin >= 512 && mt64 * (Cout / 64) < num_cus() && a.M >= 2048)
            return run_conv_p<2, 2, 2, 2, 32>("128x128", p, a, ws, ws_bytes, ws_need, s);
        return run_conv_p<1, 1, 2, 2, 32>("64x64", p, a, ws, ws_bytes, ws_need, s);
    }
    if (Cout % 32 == 0) return run_conv_p<2, 1, 4, 1, 32>("256x32", p, a, ws, ws_bytes, ws_need, s);
    return ty1::fail("ty1_conv3x3: Cout=%d must be a multiple of 32 when Cin != 3", Cout);
}

// bf16 network: activations and weights bf16, fp32 accumulate, folded BN in fp32.  out_f32 (pool == 0 only): the
// layer stores fp32 (the last conv layer feeds the fp32 Dense).  Cin == 3: x and w are passed in fp32 (the image
// enters in fp32) and rounded to bf16 inside the kernel, y is bf16.
template <int WM, int WN, int WAVES_M, int WAVES_N, int BK>
int run_conv_bf16(const char *name, bool pool, bool out_f32, const ConvArgs &a, float *ws, size_t ws_bytes, size_t *ws_need,
                  hipStream_t s) {
    if (pool) return run_conv<WM, WN, WAVES_M, WAVES_N, BK, true, OpBF16, true>(name, a, ws, ws_bytes, ws_need, s);
    if (out_f32) return run_conv<WM, WN, WAVES_M, WAVES_N, BK, false, OpBF16, false>(name, a, ws, ws_bytes, ws_need, s);
    return run_conv<WM, WN, WAVES_M, WAVES_N, BK, false, OpBF16, true>(name, a, ws, ws_bytes, ws_need, s);
}

int conv3x3_dispatch_bf16(const void *x, const void *w, const float *scale, const float *shift, void *y, int N, int H, int W,
                          int Cin, int Cout, int pool, float alpha, int out_f32, float *ws, size_t ws_bytes, size_t *ws_need,
                          hipStream_t s) {
    TY1_REQUIRE(N >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "ty1_conv3x3_bf16: bad dimensions");
    if (ws_need) *ws_need = 0;
    if (N == 0) return 0;
    TY1_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0), "ty1_conv3x3_bf16: pool needs even H, W (got %dx%d)", H, W);
    TY1_REQUIRE(Cout % 16 == 0, "ty1_conv3x3_bf16: Cout=%d must be a multiple of 16", Cout);
    TY1_REQUIRE((long long)N * H * W < (1ll << 31) - 256, "ty1_conv3x3_bf16: N*H*W must be below 2^31");
    TY1_REQUIRE(alpha >= 0.f, "ty1_conv3x3_bf16: alpha must be >= 0");
    TY1_REQUIRE(!out_f32 || !pool, "ty1_conv3x3_bf16: fp32 output is supported for pool == 0 only");
    if (!ws_need) {
        TY1_REQUIRE(x && w && scale && shift && y, "ty1_conv3x3_bf16: null pointer");
        TY1_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)ws & 15) == 0 && ((uintptr_t)y & 7) == 0,
                    "ty1_conv3x3_bf16: x, w and workspace must be 16-byte aligned, y 8-byte aligned");
    }
    if (Cin == 3) {
        TY1_REQUIRE(Cout <= 64 && W % 4 == 0 && (long long)H * W * 12 < (1ll << 31) && !out_f32,
                    "ty1_conv3x3_bf16: Cin=3 needs Cout <= 64, W %% 4 == 0, bf16 output");
        if (ws_need) return 0;
        const double px = (double)N * H * W;
        ty1::ProfScope prof(pool ? "conv3x3_c3_pool_bf16out" : "conv3x3_c3_bf16out", s, 2.0 * px * Cout * 27,
                            4.0 * (px * 3 + 27.0 * Cout) + 2.0 * px * Cout / (pool ? 4 : 1));
        return ty1::conv3x3_c3_persistent(static_cast<const float *>(x), static_cast<const float *>(w), scale, shift, y, N, H, W,
                                          Cout, pool, alpha, 1, s);
    }
    if (Cin == 16) {
        TY1_REQUIRE((Cout == 32 || Cout == 64) && (long long)H * W * 32 < (1ll << 31) && !out_f32,
                    "ty1_conv3x3_bf16: Cin=16 needs Cout 32 or 64, bf16 output");
        if (ws_need) return 0;
        return ty1::conv3x3_c16_direct_bf16(x, w, scale, shift, y, N, H, W, Cout, pool, alpha, s);
    }
    TY1_REQUIRE(Cin % 32 == 0, "ty1_conv3x3_bf16: Cin=%d must be 3, 16 or a multiple of 32", Cin);
    TY1_REQUIRE((2ll * H * W + 2ll * (W + 1)) * Cin * 2 < (1ll << 31), "ty1_conv3x3_bf16: one image (%dx%dx%d) is too large for 32-bit offsets", H, W, Cin);
    // layers with enough whole work items: input halo in LDS instead of im2col rows (conv_halo.hip); only its K-sliced form
    // (deep K, few tiles) wants a workspace
    if (!out_f32 && ty1::conv3x3_halo_bf16_supported(N, H, W, Cin, Cout, pool) && !ty1::dev_env("TY1_NO_HALO")) {
        const size_t need = ty1::conv3x3_halo_bf16_ws_bytes(N, H, W, Cin, Cout, pool);
        if (ws_need) {
            *ws_need = need;
            return 0;
        }
        if (need == 0 || (ws && ws_bytes >= need))        // (a K-sliced layer without its workspace runs on the implicit GEMM)
            return ty1::conv3x3_halo_bf16(x, w, scale, shift, y, N, H, W, Cin, Cout, pool, alpha, ws, s);
    }
    ConvArgs a{};
    a.x = static_cast<const float *>(x); a.w = static_cast<const float *>(w); a.scale = scale; a.shift = shift;
    a.y = static_cast<float *>(y);
    a.NB = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
    a.M = N * H * W;
    a.alpha = alpha;
    const bool p = pool != 0, of = out_f32 != 0;
    static const char *force = ty1::dev_env("TY1_TILE");          // developer override
    if (Cin % 64 != 0) {               // 64-byte K chunks (32 bf16)
        TY1_REQUIRE(Cout % 64 == 0, "ty1_conv3x3_bf16: Cin=%d needs Cout %% 64 == 0 (got %d)", Cin, Cout);
        if (force && !strcmp(force, "256x64")) return run_conv_bf16<4, 1, 2, 2, 16>("bk16_256x64", p, of, a, ws, ws_bytes, ws_need, s);
        if (force && !strcmp(force, "128x64")) return run_conv_bf16<2, 1, 2, 2, 16>("bk16_128x64", p, of, a, ws, ws_bytes, ws_need, s);
        // a long M (112x112x32->64 at 128 images): 256-row tiles read each weight chunk once per 256 rows (116 -> 109 us)
        if (((long long)a.M + 255) / 256 >= 2 * num_cus() && !ty1::dev_env("TY1_OLD_RULES"))
            return run_conv_bf16<4, 1, 2, 2, 16>("bk16_256x64", p, of, a, ws, ws_bytes, ws_need, s);
        return run_conv_bf16<2, 1, 2, 2, 16>("bk16_128x64", p, of, a, ws, ws_bytes, ws_need, s);
    }
    const long long mt128 = ((long long)a.M + 127) / 128;
#ifdef TY1_DEV_STAMPS       // tile shapes that only the sweeps of the dev build instantiate
    if (force && !strcmp(force, "128x256") && Cout % 256 == 0) return run_conv_bf16<2, 4, 2, 2, 32>("128x256", p, of, a, ws, ws_bytes, ws_need, s);
    if (force && !strcmp(force, "256x64") && Cout % 64 == 0) return run_conv_bf16<4, 1, 2, 2, 32>("256x64", p, of, a, ws, ws_bytes, ws_need, s);
#endif
    if (force && !strcmp(force, "256x128") && Cout % 128 == 0) return run_conv_bf16<4, 2, 2, 2, 32>("256x128", p, of, a, ws, ws_bytes, ws_need, s);
    if (force && !strcmp(force, "128x128") && Cout % 128 == 0) return run_conv_bf16<2, 2, 2, 2, 32>("128x128", p, of, a, ws, ws_bytes, ws_need, s);
    if (force && !strcmp(force, "128x64") && Cout % 64 == 0) return run_conv_bf16<2, 1, 2, 2, 32>("128x64", p, of, a, ws, ws_bytes, ws_need, s);
    if (force && !strcmp(force, "64x64") && Cout % 64 == 0) return run_conv_bf16<1, 1, 2, 2, 32>("64x64", p, of, a, ws, ws_bytes, ws_need, s);
    // 128x128 once it fills the chip, and also when even 128x64 tiles would not give two rounds (7x7x1024->256 at 128 images:
    // 196 tiles): such a layer is K-split either way and the bf16 kernel is bound by fetched bytes per MFMA, which the large
    // tile halves (measured 103 -> 50 us against the 64x64 tile, 71 us with 128x64)
    if (Cout % 128 == 0 && (mt128 * (Cout / 128) >= 2 * num_cus() || (Cin >= 512 && mt128 * (Cout / 64) < 2 * num_cus())))
        return run_conv_bf16<2, 2, 2, 2, 32>("128x128", p, of, a, ws, ws_bytes, ws_need, s);
    TY1_REQUIRE(Cout % 64 == 0, "ty1_conv3x3_bf16: Cout=%d must be a multiple of 64 when Cin >= 32", Cout);
    // deep K, too few 128x64 tiles for four rounds (7x7x512->1024 at 128 images: 784): one 256x128 workgroup per CU with
    // 128x64 per wave, everything K-split (measured 93 -> 84 us; the same tile loses 10-20 % on the layers with more tiles)
    if (Cout % 128 == 0 && Cin >= 512 && a.M >= 4096 && mt128 * (Cout / 64) < 4 * num_cus() && !ty1::dev_env("TY1_OLD_RULES"))
        return run_conv_bf16<4, 2, 2, 2, 32>("256x128", p, of, a, ws, ws_bytes, ws_need, s);
    if (mt128 * (Cout / 64) >= 2 * num_cus() || a.M >= 4096 * 2)
        return run_conv_bf16<2, 1, 2, 2, 32>("128x64", p, of, a, ws, ws_bytes, ws_need, s);
    return run_conv_bf16<1, 1, 2, 2, 32>("64x64", p, of, a, ws, ws_bytes, ws_need, s);
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float *__restrict__ x, unsigned short *__restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += 256ll * gridDim.x) y[i] = bf16_bits(x[i]);
}

}  // namespace

extern "C" int ty1_conv3x3_workspace_bytes_bf16(int N, int H, int W, int Cin, int Cout, int pool, size_t *bytes) {
    TY1_REQUIRE(bytes, "ty1_conv3x3_workspace_bytes_bf16: null result pointer");
    return conv3x3_dispatch_bf16(nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, Cout, pool, 0.f, 0, nullptr, 0, bytes, nullptr);
}

extern "C" int ty1_conv3x3_bn_leaky_bf16(const void *x, const void *w, const float *scale, const float *shift, void *y, int N,
                                         int H, int W, int Cin, int Cout, int pool, float alpha, int out_f32, void *workspace,
                                         size_t workspace_bytes, ty1_stream_t stream) {
    return conv3x3_dispatch_bf16(x, w, scale, shift, y, N, H, W, Cin, Cout, pool, alpha, out_f32, (float *)workspace,
                                 workspace_bytes, nullptr, (hipStream_t)stream);
}

extern "C" int ty1_cast_f32_to_bf16(const float *x, void *y, long long n, ty1_stream_t stream) {
    TY1_REQUIRE(n >= 0, "ty1_cast_f32_to_bf16: bad length");
    if (n == 0) return 0;
    TY1_REQUIRE(x && y, "ty1_cast_f32_to_bf16: null pointer");
    const long long blocks = (n + 255) / 256;
    cast_f32_bf16_kernel<<<(unsigned)(blocks < 4096 ? blocks : 4096), 256, 0, (hipStream_t)stream>>>(x, static_cast<unsigned short *>(y), n);
    return ty1::check_launch("ty1_cast_f32_to_bf16");
}

extern "C" int ty1_debug_stamp_buffer(long long *buf, long long cap_blocks) {
    g_stamp = buf;
    g_stamp_cap = buf ? cap_blocks : 0;
    return 0;
}

extern "C" int ty1_conv3x3_workspace_bytes(int N, int H, int W, int Cin, int Cout, int pool, size_t *bytes) {
    TY1_REQUIRE(bytes, "ty1_conv3x3_workspace_bytes: null result pointer");
    return conv3x3_dispatch(nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, Cout, pool, 0.f, nullptr, 0, bytes, nullptr);
}

extern "C" int ty1_conv3x3_bn_leaky_ws_f32(const float *x, const float *w, const float *scale, const float *shift,
                                           float *y, int N, int H, int W, int Cin, int Cout, int pool, float alpha,
                                           void *workspace, size_t workspace_bytes, ty1_stream_t stream) {
    return conv3x3_dispatch(x, w, scale, shift, y, N, H, W, Cin, Cout, pool, alpha, (float *)workspace, workspace_bytes,
                            nullptr, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// Entry point that allocates nothing (include/ty1.h: ty1_conv3x3_bn_leaky_static_f32).  The partial tiles of the K-split
// tail go to a scratch array that is part of this code object: every GPU the library runs on holds its own copy until
// the process ends.  In host code the array's name is only a handle; its device address on the current GPU comes from
// hipGetSymbolAddress, once per device, and only that pointer is ever passed to a kernel.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr size_t kStaticScratchBytes = TY1_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
void *g_scratch_ptr[64] = {};          // device address of `scratch`, per device id
std::mutex g_scratch_mutex;

int static_scratch_ptr(void **out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return ty1::fail("ty1_conv3x3_bn_leaky_static_f32: hipGetDevice failed: %s", hipGetErrorString(e));
    TY1_REQUIRE(dev >= 0 && dev < 64, "ty1_conv3x3_bn_leaky_static_f32: device id %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return ty1::fail("ty1_conv3x3_bn_leaky_static_f32: hipGetSymbolAddress(scratch) failed on device %d: %s", dev,
                             hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}
}  // namespace

extern "C" size_t ty1_static_scratch_bytes(void) { return kStaticScratchBytes; }

extern "C" int ty1_static_scratch_init(void) {
    void *ptr = nullptr;
    return static_scratch_ptr(&ptr);
}

extern "C" int ty1_conv3x3_bn_leaky_static_f32(const float *x, const float *w, const float *scale, const float *shift,
                                               float *y, int N, int H, int W, int Cin, int Cout, int pool, float alpha,
                                               ty1_stream_t stream) {
    size_t need = 0;
    if (int rc = conv3x3_dispatch(nullptr, nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, Cout, pool, 0.f, nullptr, 0, &need, nullptr))
        return rc;
    TY1_REQUIRE(need <= kStaticScratchBytes,
                "ty1_conv3x3_bn_leaky_static_f32: this problem (N=%d %dx%d %d->%d) wants %zu bytes of scratch for its K-split tail, "
                "the static buffer holds %zu: pass a workspace to ty1_conv3x3_bn_leaky_ws_f32 instead",
                N, H, W, Cin, Cout, need, kStaticScratchBytes);
    void *ptr = nullptr;
    if (need) {
        if (int rc = static_scratch_ptr(&ptr)) return rc;
    }
    return conv3x3_dispatch(x, w, scale, shift, y, N, H, W, Cin, Cout, pool, alpha, (float *)ptr, need ? kStaticScratchBytes : 0,
                            nullptr, (hipStream_t)stream);
}

extern "C" int ty1_conv3x3_bn_leaky_f32(const float *x, const float *w, const float *scale, const float *shift,
                                        float *y, int N, int H, int W, int Cin, int Cout, int pool, float alpha,
                                        ty1_stream_t stream) {
    return conv3x3_dispatch(x, w, scale, shift, y, N, H, W, Cin, Cout, pool, alpha, nullptr, 0, nullptr, (hipStream_t)stream);
}
