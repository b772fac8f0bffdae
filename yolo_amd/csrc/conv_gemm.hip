// Implicit-GEMM 3x3 convolution (+ folded BatchNorm + LeakyReLU + 2x2/2 max-pool epilogue) and the
// split-K Dense GEMM, on the exact-f32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces /root/reference/models/model_tiny_yolov1.py:43-87 (conv blocks) and :96 (Dense matmul).
//
// GEMM view:  M = output pixels (N*H*W), N = Cout, K = TAPS*Cin (k = tap*Cin + ci).
//   A[m][k]  is gathered on the fly from the NHWC input (zero outside the image),
//   B[n][k]  is the OHWI weight matrix ([Cout][9*Cin]), so both LDS tiles are [row][k] with k
//            contiguous and every lane fetches four k values with one ds_read_b128.
// For pooled layers the M index is ordered m = 4*window + (dy*2+dx): the four pixels of one
// 2x2 pool window are the four accumulator registers (reg&3) one lane already holds for one
// channel (MFMA 32x32 C/D layout row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)), so the max-pool is
// three v_max in registers and the pre-pool tensor never touches HBM.
//
// Block: 256 threads = 4 waves (WAVES_M x WAVES_N), each wave WM x WN tiles of 32x32.
// K loop: chunks of BK floats, double-buffered in LDS (row stride BK+4 floats: conflict-free for
// the ds_read_b128 16-lane groups), next chunk prefetched into registers during the MFMAs.
#include "common.h"

#include <cstdlib>
#include <cstring>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;

struct ConvArgs {
    const float *x;      // A source: NHWC input (conv) or [M][K] row-major (dense)
    const float *w;      // [Nvalid rows][ldw] weights, k contiguous
    const float *scale;  // per output channel (conv epilogue)
    const float *shift;
    float *y;            // output
    int NB, H, W, Cin;   // input geometry (dense: H=W=1, Cin=K)
    int Cout;            // GEMM N (valid rows of w)
    int ldw;             // row stride of w in floats
    int ldy;             // dense: row stride of the output partials
    long long M;         // GEMM rows (pre-pool pixels or dense batch rows)
    int chunks_per_split;
    float alpha;
    int mtiles, ntiles;
};

// Bijective XCD-aware remap: blocks b and b+8 share an XCD (private L2), so give every XCD a
// contiguous range of logical tile ids (neighbouring tiles share halo rows and weight panels).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

template <int WM, int WN, int WAVES_M, int WAVES_N, int BK, int TAPS, bool POOL, bool RAW>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N) void conv_gemm_f32_kernel(const ConvArgs a) {
    static_assert(WAVES_M * WAVES_N == 4 || WAVES_M * WAVES_N == 8, "4 or 8 waves per block");
    constexpr int NTHREADS = 64 * WAVES_M * WAVES_N;
    constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
    constexpr int LDS_STRIDE = BK + 4;
    constexpr int CPR = BK / 4;               // 16-byte pieces per tile row
    constexpr int RPP = NTHREADS / CPR;       // tile rows covered per pass
    constexpr int AP = (BM + RPP - 1) / RPP;  // passes for the A tile
    constexpr int BP = (BN + RPP - 1) / RPP;  // passes for the B tile

    __shared__ __attribute__((aligned(16))) float As[2][BM * LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDS_STRIDE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;

    // Two blocks share a CU, i.e. two waves share each SIMD's matrix pipe.  Running the same program at
    // equal priority they fall into lockstep and stall together at every barrier / LDS-store phase.
    // Give the wave in the odd hardware wave slot a higher static priority: it issues whenever it can,
    // the partner fills its gaps, and the two drift apart.  (HW_REG_HW_ID[3:0] = wave slot on the SIMD.)
    if (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1) __builtin_amdgcn_s_setprio(1);

    const int nwg = a.mtiles * a.ntiles;
    const int logical = xcd_remap(blockIdx.x, nwg);
    const int mt = logical / a.ntiles, nt = logical - mt * a.ntiles;
    const int split = blockIdx.y;

    const int piece = tid % CPR, prow = tid / CPR;

    // ---- operands are fetched with raw buffer loads: an out-of-range offset returns 0, which gives the
    //      zero padding of the 3x3 halo, the M / N tails and a branch-free K loop (one basic block that
    //      the scheduler can interleave with the MFMAs).  Offsets are 32-bit bytes relative to the first
    //      image this block touches, so tensors larger than 4 GiB still work.
    constexpr unsigned kOOB = 0xFFFFFFFFu;
    const long long m_first = (long long)mt * BM;
    long long a_origin;                       // element offset of the block's buffer base inside x
    if (TAPS == 1) {
        a_origin = m_first * a.Cin;
    } else {
        const long long pix = POOL ? (m_first >> 2) / ((a.H >> 1) * (a.W >> 1)) : m_first / ((long long)a.H * a.W);
        a_origin = pix * a.H * a.W * a.Cin;
    }
    const long long a_total = (TAPS == 1 ? a.M : (long long)a.NB * a.H * a.W) * a.Cin;
    const long long a_left = (a_total - a_origin) * 4;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.x + a_origin), 0, (int)(a_left > 0xFFFFFFFFll ? 0xFFFFFFFFll : a_left), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(a.w), 0, (int)((long long)a.Cout * a.ldw * 4), 0x00020000);

    unsigned a_base[AP];                      // byte offset of the row's centre pixel (or row start)
    unsigned a_mask[AP];                      // bit t: tap t lies inside the image
#pragma unroll
    for (int p = 0; p < AP; ++p) {
        const int row = p * RPP + prow;
        const long long m = m_first + row;
        a_base[p] = 0;
        a_mask[p] = 0;
        if (row < BM && m < a.M) {
            if (TAPS == 1) {
                a_base[p] = (unsigned)((long long)row * a.Cin * 4);
                a_mask[p] = 1;
            } else {
                long long n;
                int yy, xx;
                if (POOL) {
                    const int Hp = a.H >> 1, Wp = a.W >> 1;
                    const long long win = m >> 2;
                    const int q = (int)(m & 3);
                    n = win / (Hp * Wp);
                    const int rem = (int)(win - n * (Hp * Wp));
                    const int py = rem / Wp, px = rem - py * Wp;
                    yy = 2 * py + (q >> 1);
                    xx = 2 * px + (q & 1);
                } else {
                    n = m / (a.H * a.W);
                    const int rem = (int)(m - n * (a.H * a.W));
                    yy = rem / a.W;
                    xx = rem - yy * a.W;
                }
                const long long el = ((n * a.H + yy) * a.W + xx) * a.Cin - a_origin;
                a_base[p] = (unsigned)(el * 4);
                unsigned mask = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int y2 = yy + t / 3 - 1, x2 = xx + t % 3 - 1;
                    if (y2 >= 0 && y2 < a.H && x2 >= 0 && x2 < a.W) mask |= 1u << t;
                }
                a_mask[p] = mask;
            }
        }
    }
    unsigned b_base[BP];
#pragma unroll
    for (int p = 0; p < BP; ++p) {
        const int row = p * RPP + prow;
        const int n = nt * BN + row;
        b_base[p] = (row < BN && n < a.Cout) ? (unsigned)((long long)n * a.ldw * 4) : kOOB;
    }

    const int chunks_per_tap = a.Cin / BK;
    const int kc_begin = split * a.chunks_per_split;
    const int kc_end = kc_begin + a.chunks_per_split;

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[AP], rb[BP];
    auto load_chunk = [&](int kc) {
        int tap, ci0;
        if (TAPS == 1) {
            tap = 0;
            ci0 = kc * BK;
        } else {
            tap = kc / chunks_per_tap;
            ci0 = (kc - tap * chunks_per_tap) * BK;
        }
        const int dy = TAPS == 1 ? 0 : tap / 3 - 1, dx = TAPS == 1 ? 0 : tap % 3 - 1;
        const int delta = (((dy * a.W + dx) * a.Cin) + ci0 + piece * 4) * 4;      // bytes, may be negative
#pragma unroll
        for (int p = 0; p < AP; ++p) {
            const unsigned off = ((a_mask[p] >> tap) & 1u) ? a_base[p] + (unsigned)delta : kOOB;
            ra[p] = __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, (int)off, 0, 0);
        }
        const unsigned koff = (unsigned)(kc * BK + piece * 4) * 4u;
#pragma unroll
        for (int p = 0; p < BP; ++p) {
            const unsigned off = b_base[p] == kOOB ? kOOB : b_base[p] + koff;
            rb[p] = __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, (int)off, 0, 0);
        }
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
    const int a_row0 = wave_m * (32 * WM) + frag_row;
    const int b_row0 = wave_n * (32 * WN) + frag_row;

    auto compute = [&](int buf, int kb) {
        float4 fa[WM], fb[WN];
#pragma unroll
        for (int i = 0; i < WM; ++i)
            fa[i] = *reinterpret_cast<const float4 *>(&As[buf][(a_row0 + 32 * i) * LDS_STRIDE + kb * 8 + frag_k]);
#pragma unroll
        for (int j = 0; j < WN; ++j)
            fb[j] = *reinterpret_cast<const float4 *>(&Bs[buf][(b_row0 + 32 * j) * LDS_STRIDE + kb * 8 + frag_k]);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
            }
    };

    load_chunk(kc_begin);
    store_chunk(0);
    __syncthreads();

    int buf = 0;
    for (int kc = kc_begin; kc + 1 < kc_end; ++kc) {
        load_chunk(kc + 1);                       // lands while the MFMAs of this chunk run
        __builtin_amdgcn_sched_barrier(0);        // keep the loads up here: the scheduler otherwise sinks them to their use
#pragma unroll
        for (int kb = 0; kb < BK / 8 - 1; ++kb) compute(buf, kb);
        __builtin_amdgcn_sched_barrier(0);
        // the other buffer was last read before the barrier that ended the previous iteration
        store_chunk(buf ^ 1);
        compute(buf, BK / 8 - 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) compute(buf, kb);

    // ---- epilogue.  Lane holds channel (lane&31) of tile column j; register r is tile row
    //      (r&3) + 8*(r>>2) + 4*(lane>>5).
    const int h = lane >> 5;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = nt * BN + wave_n * (32 * WN) + 32 * j + (lane & 31);
        if (n >= a.Cout) continue;
        float sc = 1.f, sh = 0.f;
        if (!RAW) {
            sc = a.scale[n];
            sh = a.shift[n];
        }
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const long long m0 = (long long)mt * BM + wave_m * (32 * WM) + 32 * i;
            if (RAW) {
                float *out = a.y + (long long)split * a.M * a.ldy;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < a.M) out[m * a.ldy + n] = acc[i][j][r];
                }
            } else if (POOL) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float t = acc[i][j][4 * g + q] * sc + sh;
                        v[q] = t > 0.f ? t : t * a.alpha;
                    }
                    const float pooled = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                    const long long win = (m0 >> 2) + 2 * g + h;
                    if (win * 4 < a.M) a.y[win * a.Cout + n] = pooled;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long m = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    float t = acc[i][j][r] * sc + sh;
                    t = t > 0.f ? t : t * a.alpha;
                    if (m < a.M) a.y[m * a.Cout + n] = t;
                }
            }
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

template <int WM, int WN, int WAVES_M, int WAVES_N, int BK, int TAPS, bool POOL, bool RAW>
int launch(const char *name, ConvArgs a, int splits, hipStream_t s) {
    constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
    a.mtiles = (int)((a.M + BM - 1) / BM);
    a.ntiles = (a.Cout + BN - 1) / BN;
    dim3 grid(a.mtiles * a.ntiles, splits), block(64 * WAVES_M * WAVES_N);
    // algorithmic work: 2*M*N*K flops; input + weights read once, output written once
    const double K = (double)TAPS * a.Cin;
    const double out_rows = POOL ? (double)a.M / 4 : (double)a.M;
    char tag[96];
    snprintf(tag, sizeof tag, "%s@%dx%d>%d", name, a.H, a.Cin, a.Cout);
    ty1::ProfScope prof(tag, s, 2.0 * (double)a.M * a.Cout * K,
                        4.0 * ((double)a.M * a.Cin + (double)a.Cout * K + out_rows * a.Cout * (RAW ? splits : 1)));
    conv_gemm_f32_kernel<WM, WN, WAVES_M, WAVES_N, BK, TAPS, POOL, RAW><<<grid, block, 0, s>>>(a);
    return ty1::check_launch(name);
}

}  // namespace

extern "C" int ty1_conv3x3_bn_leaky_f32(const float *x, const float *w, const float *scale, const float *shift,
                                        float *y, int N, int H, int W, int Cin, int Cout, int pool, float alpha,
                                        ty1_stream_t stream) {
    TY1_REQUIRE(N >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "ty1_conv3x3: bad dimensions");
    if (N == 0) return 0;
    TY1_REQUIRE(x && w && scale && shift && y, "ty1_conv3x3: null pointer");
    TY1_REQUIRE(!pool || (H % 2 == 0 && W % 2 == 0), "ty1_conv3x3: pool needs even H, W (got %dx%d)", H, W);
    TY1_REQUIRE(Cout % 16 == 0, "ty1_conv3x3: Cout=%d must be a multiple of 16", Cout);
    TY1_REQUIRE((long long)N * H * W * Cout < (1ll << 40), "ty1_conv3x3: tensor too large");
    hipStream_t s = (hipStream_t)stream;

    if (Cin == 3) {
        TY1_REQUIRE(Cout <= 64, "ty1_conv3x3: Cin=3 path supports Cout <= 64 (got %d)", Cout);
        const int tiles = ((W + 63) / 64) * ((H + 15) / 16);
        TY1_REQUIRE((long long)N * tiles < (1ll << 31), "ty1_conv3x3: grid too large");
        dim3 grid((unsigned)(N * tiles));
        const double px = (double)N * H * W;
        ty1::ProfScope prof(pool ? "conv3x3_c3_pool" : "conv3x3_c3", s, 2.0 * px * Cout * 27,
                            4.0 * (px * 3 + 27.0 * Cout + px * Cout / (pool ? 4 : 1)));
        if (pool)
            launch_c3<true>(Cout / 16, grid, s, x, w, scale, shift, y, N, H, W, Cout, alpha);
        else
            launch_c3<false>(Cout / 16, grid, s, x, w, scale, shift, y, N, H, W, Cout, alpha);
        return ty1::check_launch("ty1_conv3x3(c3)");
    }

    TY1_REQUIRE(Cin % 16 == 0, "ty1_conv3x3: Cin=%d must be 3 or a multiple of 16", Cin);
    TY1_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, "ty1_conv3x3: x and w must be 16-byte aligned");
    ConvArgs a{};
    a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.y = y;
    a.NB = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ldw = 9 * Cin; a.ldy = Cout;
    a.M = (long long)N * H * W;
    a.alpha = alpha;
    const bool bk32 = Cin % 32 == 0;
    a.chunks_per_split = 9 * Cin / (bk32 ? 32 : 16);

    // tile choice: keep BN <= Cout, and prefer >= 2 blocks per CU
    if (!bk32) {                      // Cin == 16 (48, 80 ...): BK = 16
        if (Cout % 64 == 0)
            return pool ? launch<2, 1, 2, 2, 16, 9, true, false>("conv_f32_bk16_128x64_pool", a, 1, s)
                        : launch<2, 1, 2, 2, 16, 9, false, false>("conv_f32_bk16_128x64", a, 1, s);
        if (Cout % 32 == 0)
            return pool ? launch<2, 1, 4, 1, 16, 9, true, false>("conv_f32_bk16_256x32_pool", a, 1, s)
                        : launch<2, 1, 4, 1, 16, 9, false, false>("conv_f32_bk16_256x32", a, 1, s);
        return ty1::fail("ty1_conv3x3: Cin=%d needs Cout %% 32 == 0 (got %d)", Cin, Cout);
    }
    const long long tiles128 = ((a.M + 127) / 128) * ((Cout + 127) / 128);
    static const char *exp_cfg = getenv("TY1_EXP");
    if (exp_cfg && Cout % 128 == 0) {
        if (!strcmp(exp_cfg, "w8"))
            return pool ? launch<2, 1, 2, 4, 32, 9, true, false>("conv_f32_128x128w8_pool", a, 1, s)
                        : launch<2, 1, 2, 4, 32, 9, false, false>("conv_f32_128x128w8", a, 1, s);
        if (!strcmp(exp_cfg, "bk16"))
            return pool ? launch<2, 2, 2, 2, 16, 9, true, false>("conv_f32_128x128bk16_pool", a, 1, s)
                        : launch<2, 2, 2, 2, 16, 9, false, false>("conv_f32_128x128bk16", a, 1, s);
        if (!strcmp(exp_cfg, "w8b"))
            return pool ? launch<1, 2, 4, 2, 32, 9, true, false>("conv_f32_128x128w8b_pool", a, 1, s)
                        : launch<1, 2, 4, 2, 32, 9, false, false>("conv_f32_128x128w8b", a, 1, s);
    }
    if (Cout % 128 == 0 && tiles128 >= 512)
        return pool ? launch<2, 2, 2, 2, 32, 9, true, false>("conv_f32_128x128_pool", a, 1, s)
                    : launch<2, 2, 2, 2, 32, 9, false, false>("conv_f32_128x128", a, 1, s);
    if (Cout % 64 == 0) {
        const long long tiles = ((a.M + 127) / 128) * (Cout / 64);
        if (tiles >= 512 || a.M >= 4096)
            return pool ? launch<2, 1, 2, 2, 32, 9, true, false>("conv_f32_128x64_pool", a, 1, s)
                        : launch<2, 1, 2, 2, 32, 9, false, false>("conv_f32_128x64", a, 1, s);
        return pool ? launch<1, 1, 2, 2, 32, 9, true, false>("conv_f32_64x64_pool", a, 1, s)
                    : launch<1, 1, 2, 2, 32, 9, false, false>("conv_f32_64x64", a, 1, s);
    }
    if (Cout % 32 == 0)
        return pool ? launch<2, 1, 4, 1, 32, 9, true, false>("conv_f32_256x32_pool", a, 1, s)
                    : launch<2, 1, 4, 1, 32, 9, false, false>("conv_f32_256x32", a, 1, s);
    return ty1::fail("ty1_conv3x3: Cout=%d must be a multiple of 32 when Cin != 3", Cout);
}

extern "C" int ty1_dense_splitk_f32(const float *x, const float *wt, float *part, int N, int K, int units, int ldk,
                                    int ldz, int splits, ty1_stream_t stream) {
    TY1_REQUIRE(N >= 0 && K > 0 && units > 0 && splits > 0, "ty1_dense_splitk: bad dimensions");
    if (N == 0) return 0;
    TY1_REQUIRE(x && wt && part, "ty1_dense_splitk: null pointer");
    TY1_REQUIRE(ldk >= K && ldk % 4 == 0 && K % 4 == 0, "ty1_dense_splitk: ldk=%d must be >= K=%d, both multiples of 4", ldk, K);
    TY1_REQUIRE(ldz >= units, "ty1_dense_splitk: ldz=%d < units=%d", ldz, units);
    TY1_REQUIRE(K % (32 * splits) == 0, "ty1_dense_splitk: K=%d must be a multiple of 32*splits (splits=%d)", K, splits);
    TY1_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)wt & 15) == 0, "ty1_dense_splitk: x and wt must be 16-byte aligned");
    ConvArgs a{};
    a.x = x; a.w = wt; a.y = part;
    a.NB = N; a.H = 1; a.W = 1; a.Cin = K; a.Cout = units; a.ldw = ldk; a.ldy = ldz;
    a.M = N;
    a.alpha = 0.f;
    a.chunks_per_split = K / 32 / splits;
    hipStream_t s = (hipStream_t)stream;
    if (N <= 32) return launch<1, 1, 1, 4, 32, 1, false, true>("dense_f32_32x128", a, splits, s);
    return launch<1, 1, 2, 2, 32, 1, false, true>("dense_f32_64x64", a, splits, s);
}
