// Decode + adaptive threshold + class-agnostic greedy NMS: one 64-lane wavefront per image,
// one lane per grid cell (S*S <= 64), B boxes per lane in registers.
//
// Mirrors /root/reference/tiny_yolov1.py:82-140 (C=1 twin faces/testFace.py:51-110) bit for bit:
// scores/confidences in float32, box geometry in float64, ">= best" responsible-box
// ties, "max/2" adaptive threshold, per-round confidence ties all kept with the LAST tied slot in
// raster order as the suppressor (Q4), IoU > thr strict so NaN never suppresses.
//
// The image's C+5B head rows are staged through LDS with coalesced loads; every reduction is a
// 64-lane butterfly (no block barrier inside the greedy loop); the suppressor box is broadcast
// with readlane-style shuffles.  HBM-bound by construction: 4*(C+5B)*S*S bytes in, 41*S*S*B out.
//
// Strict IEEE arithmetic is required for bit-exact parity: this file is compiled with
// -ffp-contract=off and uses no fast-math.
#include "common.h"
#include "head_math.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float np_maxf(float a, float b) {      // np.max: NaN propagates
    if (a != a) return a;
    if (b != b) return b;
    return a >= b ? a : b;
}

// 64-lane all-reduce on the DPP crossbar (a ds_bpermute butterfly costs an LDS round trip per step): quad swaps,
// half-row and row mirrors leave every lane of a 16-lane row with the row's result, the four rows are then combined
// from lanes 15 / 31 / 47 / 63.  Both operators are commutative and associative (NaN propagates through np_maxf
// whatever the order), so the result does not depend on the reduction tree.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(dpp_i<CTRL>(__float_as_int(v)));
}
constexpr int kQuadXor1 = 0xB1, kQuadXor2 = 0x4E, kRowHalfMirror = 0x141, kRowMirror = 0x140;

__device__ __forceinline__ float wave_np_max(float v) {
    v = np_maxf(v, dpp_f<kQuadXor1>(v));
    v = np_maxf(v, dpp_f<kQuadXor2>(v));
    v = np_maxf(v, dpp_f<kRowHalfMirror>(v));
    v = np_maxf(v, dpp_f<kRowMirror>(v));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 47));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    return np_maxf(np_maxf(r0, r1), np_maxf(r2, r3));
}

__device__ __forceinline__ int wave_max_i(int v) {
    int o;
    o = dpp_i<kQuadXor1>(v); v = o > v ? o : v;
    o = dpp_i<kQuadXor2>(v); v = o > v ? o : v;
    o = dpp_i<kRowHalfMirror>(v); v = o > v ? o : v;
    o = dpp_i<kRowMirror>(v); v = o > v ? o : v;
    const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31);
    const int r2 = __builtin_amdgcn_readlane(v, 47), r3 = __builtin_amdgcn_readlane(v, 63);
    const int a = r0 > r1 ? r0 : r1, b = r2 > r3 ? r2 : r3;
    return a > b ? a : b;
}

// value of lane `src` (wave-uniform index) in every lane
__device__ __forceinline__ double wave_bcast_f64(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), src);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// tiny_yolov1.py:58-72 in float64; np.maximum/np.minimum propagate NaN.
__device__ __forceinline__ double iou_f64(const double amin[2], const double amax[2], const double bmin[2],
                                          const double bmax[2]) {
    double wh[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        double hi = amax[d] < bmax[d] ? amax[d] : bmax[d];
        double lo = amin[d] > bmin[d] ? amin[d] : bmin[d];
        double v = hi - lo;
        bool nan = (amin[d] != amin[d]) || (bmin[d] != bmin[d]) || (amax[d] != amax[d]) || (bmax[d] != bmax[d]);
        if (nan) v = __builtin_nan("");
        else if (!(v > 0.)) v = 0.;
        wh[d] = v;
    }
    double inter = wh[0] * wh[1];
    double pa = (amax[0] - amin[0]) * (amax[1] - amin[1]);
    double ta = (bmax[0] - bmin[0]) * (bmax[1] - bmin[1]);
    double uni = pa + ta - inter;
    return inter / uni;
}

// One wavefront, one image: `img` is the image's head tensor (S,S,C+5B) in LDS; lane = grid cell.
template <int B>
__device__ __forceinline__ void decode_nms_wave(const float *img, int n, bool active, int lane, int S, int C, int img_size,
                                                double iou_thr, double *__restrict__ boxes, float *__restrict__ score,
                                                int32_t *__restrict__ cls, uint8_t *__restrict__ keep) {
    const int cells = S * S, D = C + 5 * B, slots = cells * B;
    const bool valid = active && lane < cells;
    const float *pc = img + (valid ? lane : 0) * D;
    const float NEG_INF = -__builtin_inff();

    float bcs[B], trust[B];
    int bcl[B];
    bool filt[B], kept[B];
    double mn[B][2], mx[B][2];

    // :90-93 scores = class * trust, first argmax and max over classes (NumPy argmax NaN rule)
    float cell_best = NEG_INF;
#pragma unroll
    for (int k = 0; k < B; ++k) {
        float mp = NEG_INF;
        int mi = 0;
        if (valid) {
            float t = pc[C + k];
            mp = pc[0] * t;
            if (mp == mp) {
                for (int m = 1; m < C; ++m) {
                    float v = pc[m] * t;
                    if (!(v <= mp)) {
                        mp = v;
                        mi = m;
                        if (mp != mp) break;
                    }
                }
            }
        }
        bcs[k] = mp;
        bcl[k] = mi;
        cell_best = (k == 0) ? mp : np_maxf(cell_best, mp);
        kept[k] = false;
    }
    // :94-99 responsible box(es) of the cell and the adaptive threshold max/2
    const float gmax = wave_np_max(cell_best);
    const float thr = gmax / 2;
    const int r = lane / S, c = lane - r * S;
#pragma unroll
    for (int k = 0; k < B; ++k) {
        filt[k] = valid && (bcs[k] >= thr) && (bcs[k] >= cell_best);
        // :103-112 zero what is filtered, decode (YOLO_head xy in f64, wh in f32; min/max in f64)
        const float f = filt[k] ? 1.0f : 0.0f;
        const float *pb = pc + C + B + 4 * k;
        float bx = pb[0] * f, by = pb[1] * f, bw = pb[2] * f, bh = pb[3] * f;
        double x = ((double)bx + (double)c) / (double)S * (double)img_size;
        double y = ((double)by + (double)r) / (double)S * (double)img_size;
        float w = bw * (float)img_size, h = bh * (float)img_size;
        float hw = w / 2, hh = h / 2;
        mn[k][0] = x - (double)hw;
        mn[k][1] = y - (double)hh;
        mx[k][0] = x + (double)hw;
        mx[k][1] = y + (double)hh;
        trust[k] = valid ? pc[C + k] * f : NEG_INF;
        bcl[k] = filt[k] ? bcl[k] : 0;
        if (valid) {
            score[(size_t)n * slots + lane * B + k] = trust[k];
            cls[(size_t)n * slots + lane * B + k] = bcl[k];
        }
    }

    // :113-137 greedy loop.  Every round clears at least one slot, so it ends within `slots` rounds.
    float lmax = trust[0];
#pragma unroll
    for (int k = 1; k < B; ++k) lmax = np_maxf(lmax, trust[k]);
    float tmax = wave_np_max(lmax);
    for (int round = 0; round < slots && tmax > 0; ++round) {
        int my_last = -1;
#pragma unroll
        for (int k = 0; k < B; ++k) {
            if (trust[k] == tmax) {
                kept[k] = true;
                filt[k] = false;
                my_last = lane * B + k;
            }
        }
        const int last = wave_max_i(my_last);          // last tied slot in raster order
        const int owner = last / B, kl = last - owner * B;
        double smin[2], smax[2], sel[4];
        sel[0] = mn[0][0]; sel[1] = mn[0][1]; sel[2] = mx[0][0]; sel[3] = mx[0][1];
#pragma unroll
        for (int k = 1; k < B; ++k)
            if (kl == k) { sel[0] = mn[k][0]; sel[1] = mn[k][1]; sel[2] = mx[k][0]; sel[3] = mx[k][1]; }
        smin[0] = wave_bcast_f64(sel[0], owner);
        smin[1] = wave_bcast_f64(sel[1], owner);
        smax[0] = wave_bcast_f64(sel[2], owner);
        smax[1] = wave_bcast_f64(sel[3], owner);
        lmax = NEG_INF;
#pragma unroll
        for (int k = 0; k < B; ++k) {
            if (filt[k]) {
                double v = iou_f64(smin, smax, mn[k], mx[k]);
                if (v > iou_thr) filt[k] = false;
            }
            if (valid) trust[k] = trust[k] * (filt[k] ? 1.0f : 0.0f);
            lmax = (k == 0) ? trust[k] : np_maxf(lmax, trust[k]);
        }
        tmax = wave_np_max(lmax);
    }

    // :139-140
    if (valid) {
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const double m = kept[k] ? 1.0 : 0.0;
            double *o = boxes + ((size_t)n * slots + lane * B + k) * 4;
            double2 lo = make_double2(mn[k][0] * m, mn[k][1] * m);
            double2 hi = make_double2(mx[k][0] * m, mx[k][1] * m);
            reinterpret_cast<double2 *>(o)[0] = lo;
            reinterpret_cast<double2 *>(o)[1] = hi;
            keep[(size_t)n * slots + lane * B + k] = kept[k] ? 1 : 0;
        }
    }
}

template <int B>
__global__ __launch_bounds__(256) void decode_nms_kernel(const float *__restrict__ pred, int N, int S, int C,
                                                          int img_size, double iou_thr, int img_stride, int vec2,
                                                          double *__restrict__ boxes, float *__restrict__ score,
                                                          int32_t *__restrict__ cls, uint8_t *__restrict__ keep) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    const bool active = n < N;
    const int total = S * S * (C + 5 * B);
    float *img = lds + wave * img_stride;

    if (active) {
        const float *src = pred + (size_t)n * total;
        // 8-byte loads when every image base is 8-byte aligned (checked on the host)
        if (vec2) {
            const float2 *s2 = reinterpret_cast<const float2 *>(src);
            float2 *d2 = reinterpret_cast<float2 *>(img);
            for (int i = lane; i < total / 2; i += 64) d2[i] = s2[i];
        } else {
            for (int i = lane; i < total; i += 64) img[i] = src[i];
        }
    }
    __syncthreads();
    decode_nms_wave<B>(img, n, active, lane, S, C, img_size, iou_thr, boxes, score, cls, keep);
}

// The whole inference tail behind the Dense GEMM in ONE launch: split-K reduction + bias, ReshapeYOLO.call, decode,
// adaptive threshold, NMS.  One 1024-thread workgroup per image: all sixteen waves reduce and activate the image's
// S*S*(C+5B) logits into LDS (the arithmetic of ty1_head_activate_f32: head_math.h), then wave 0 runs the decode + NMS
// of ty1_decode_nms on that LDS image.  The head tensor goes to HBM only if the caller asks for it (y != nullptr).
constexpr int kTailThreads = 1024;
template <int B>
__global__ __launch_bounds__(kTailThreads) void head_decode_nms_kernel(const float *__restrict__ part, const float *__restrict__ bias,
                                                               float *__restrict__ y, int N, int S, int C, int ldz, int splits,
                                                               int img_size, double iou_thr, double *__restrict__ boxes,
                                                               float *__restrict__ score, int32_t *__restrict__ cls,
                                                               uint8_t *__restrict__ keep) {
    extern __shared__ float lds[];
    const int n = blockIdx.x, cells = S * S, D = C + 5 * B, units = cells * D;
    float *z = lds, *img = lds + ((units + 1) & ~1), *smx = img + ((units + 1) & ~1), *ssum = smx + 64;
    const size_t slab = (size_t)N * ldz;
    const float *pn = part + (size_t)n * ldz;
    for (int t = threadIdx.x; t < units; t += kTailThreads) {
        const int cell = t / D, d = t - cell * D;
        z[t] = ty1::reduce_unit(pn, slab, ty1::planar_unit(cell, d, cells, B, C), bias, splits);
    }
    __syncthreads();
    // softmax maximum and denominator once per cell, in activate_channel's order (same bits), instead of once per channel
    if ((int)threadIdx.x < cells) {
        const float *zc = z + threadIdx.x * D;
        float mx = zc[0];
        for (int m = 1; m < C; ++m) mx = fmaxf(mx, zc[m]);
        float sum = 0.f;
        for (int m = 0; m < C; ++m) sum += expf(zc[m] - mx);
        smx[threadIdx.x] = mx;
        ssum[threadIdx.x] = sum;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < units; t += kTailThreads) {
        const int cell = t / D, d = t - cell * D;
        const float v = d < C ? expf(z[t] - smx[cell]) / ssum[cell] : ty1::sigmoid_f32(z[t]);
        img[t] = v;
        if (y) y[(size_t)n * units + t] = v;
    }
    __syncthreads();
    if (threadIdx.x < 64) decode_nms_wave<B>(img, n, true, (int)threadIdx.x, S, C, img_size, iou_thr, boxes, score, cls, keep);
}

// NumPy twins of the reference (tiny_yolov1.py:36-72): YOLO_head with xy in float64 and wh in float32,
// iou in float64.  feats has no batch axis there; a leading batch axis is accepted here.
__global__ void yolo_head_np_kernel(const float *__restrict__ feats, double *__restrict__ xy, float *__restrict__ wh,
                                    long long nbox, int S, int B, int img_size) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbox) return;
    const long long cell = i / B;
    const int c = (int)(cell % S), r = (int)((cell / S) % S);
    const float4 f = reinterpret_cast<const float4 *>(feats)[i];
    xy[2 * i + 0] = ((double)f.x + (double)c) / (double)S * (double)img_size;
    xy[2 * i + 1] = ((double)f.y + (double)r) / (double)S * (double)img_size;
    wh[2 * i + 0] = f.z * (float)img_size;
    wh[2 * i + 1] = f.w * (float)img_size;
}

__global__ void iou_f64_kernel(const double *__restrict__ pmin, const double *__restrict__ pmax,
                               const double *__restrict__ tmin, const double *__restrict__ tmax,
                               double *__restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = iou_f64(pmin + 2 * i, pmax + 2 * i, tmin + 2 * i, tmax + 2 * i);
}

}  // namespace

extern "C" int ty1_yolo_head_np_f64(const float *feats, double *xy, float *wh, int N, int S, int B, int img_size,
                                    ty1_stream_t stream) {
    TY1_REQUIRE(N >= 0 && S > 0 && B > 0 && img_size > 0, "ty1_yolo_head_np: bad dimensions");
    const long long nbox = (long long)N * S * S * B;
    if (nbox == 0) return 0;
    TY1_REQUIRE(feats && xy && wh, "ty1_yolo_head_np: null pointer");
    TY1_REQUIRE(((uintptr_t)feats & 15) == 0, "ty1_yolo_head_np: feats must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ty1::ProfScope prof("yolo_head_np", s);
    yolo_head_np_kernel<<<dim3((unsigned)((nbox + 255) / 256)), dim3(256), 0, s>>>(feats, xy, wh, nbox, S, B, img_size);
    return ty1::check_launch("ty1_yolo_head_np_f64");
}

extern "C" int ty1_iou_f64(const double *pred_mins, const double *pred_maxes, const double *true_mins,
                           const double *true_maxes, double *out, int64_t n, ty1_stream_t stream) {
    TY1_REQUIRE(n >= 0, "ty1_iou_f64: negative count");
    if (n == 0) return 0;
    TY1_REQUIRE(pred_mins && pred_maxes && true_mins && true_maxes && out, "ty1_iou_f64: null pointer");
    hipStream_t s = (hipStream_t)stream;
    ty1::ProfScope prof("iou_f64", s);
    iou_f64_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(pred_mins, pred_maxes, true_mins, true_maxes, out, n);
    return ty1::check_launch("ty1_iou_f64");
}

// Fixed-size detection records for the multi-GPU gather: rec[i] = (x0, y0, x1, y1, score, class or -1 if the slot is
// not kept), all float64 (box corners stay bit-exact).  One thread per slot, one 48-byte record each.
__global__ __launch_bounds__(256) void pack_detections_kernel(const double *__restrict__ boxes, const float *__restrict__ score,
                                                               const int32_t *__restrict__ cls, const uint8_t *__restrict__ keep,
                                                               double *__restrict__ rec, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double2 lo = *reinterpret_cast<const double2 *>(boxes + 4 * i), hi = *reinterpret_cast<const double2 *>(boxes + 4 * i + 2);
    double2 *o = reinterpret_cast<double2 *>(rec + 6 * i);
    o[0] = lo;
    o[1] = hi;
    o[2] = make_double2((double)score[i], keep[i] ? (double)cls[i] : -1.0);
}

extern "C" int ty1_decode_nms(const float *pred, int N, int S, int B, int C, int img_size, double iou_thr,
                              double *boxes, float *score, int32_t *cls, uint8_t *keep, ty1_stream_t stream) {
    TY1_REQUIRE(N >= 0 && S > 0 && C > 0 && img_size > 0, "ty1_decode_nms: bad dimensions N=%d S=%d C=%d", N, S, C);
    TY1_REQUIRE(S * S <= 64, "ty1_decode_nms: S*S=%d exceeds one wavefront (64 cells)", S * S);
    TY1_REQUIRE(B >= 1 && B <= 4, "ty1_decode_nms: B=%d not in 1..4", B);
    if (N == 0) return 0;
    TY1_REQUIRE(pred && boxes && score && cls && keep, "ty1_decode_nms: null pointer");
    const int total = S * S * (C + 5 * B);
    const int img_stride = (total + 1) & ~1;
    const size_t lds = (size_t)4 * img_stride * sizeof(float);
    TY1_REQUIRE(lds <= 64 * 1024, "ty1_decode_nms: C=%d needs %zu bytes of LDS (max 65536)", C, lds);
    const int vec2 = ((total & 1) == 0 && ((uintptr_t)pred & 7) == 0) ? 1 : 0;
    TY1_REQUIRE(((uintptr_t)boxes & 15) == 0, "ty1_decode_nms: boxes must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((N + 3) / 4), block(256);
    // algorithmic bytes: the head tensor in, (4 f64 + f32 + i32 + u8) per slot out
    ty1::ProfScope prof("decode_nms", s, 0.0, (double)N * (4.0 * total + 41.0 * S * S * B));
    switch (B) {
        case 1: decode_nms_kernel<1><<<grid, block, lds, s>>>(pred, N, S, C, img_size, iou_thr, img_stride, vec2, boxes, score, cls, keep); break;
        case 2: decode_nms_kernel<2><<<grid, block, lds, s>>>(pred, N, S, C, img_size, iou_thr, img_stride, vec2, boxes, score, cls, keep); break;
        case 3: decode_nms_kernel<3><<<grid, block, lds, s>>>(pred, N, S, C, img_size, iou_thr, img_stride, vec2, boxes, score, cls, keep); break;
        default: decode_nms_kernel<4><<<grid, block, lds, s>>>(pred, N, S, C, img_size, iou_thr, img_stride, vec2, boxes, score, cls, keep); break;
    }
    return ty1::check_launch("ty1_decode_nms");
}

extern "C" int ty1_head_decode_nms_f32(const float *part, const float *bias, float *y, int N, int S, int B, int C, int ldz,
                                       int splits, int img_size, double iou_thr, double *boxes, float *score, int32_t *cls,
                                       uint8_t *keep, ty1_stream_t stream) {
    TY1_REQUIRE(N >= 0 && S > 0 && C > 0 && img_size > 0 && splits > 0, "ty1_head_decode_nms: bad dimensions N=%d S=%d C=%d splits=%d", N, S, C, splits);
    TY1_REQUIRE(S * S <= 64, "ty1_head_decode_nms: S*S=%d exceeds one wavefront (64 cells)", S * S);
    TY1_REQUIRE(B >= 1 && B <= 4, "ty1_head_decode_nms: B=%d not in 1..4", B);
    if (N == 0) return 0;
    const int units = S * S * (C + 5 * B);
    TY1_REQUIRE(part && boxes && score && cls && keep, "ty1_head_decode_nms: null pointer");
    TY1_REQUIRE(ldz >= units, "ty1_head_decode_nms: ldz=%d < units=%d", ldz, units);
    TY1_REQUIRE(((uintptr_t)boxes & 15) == 0, "ty1_head_decode_nms: boxes must be 16-byte aligned");
    const size_t lds = ((size_t)2 * ((units + 1) & ~1) + 128) * sizeof(float);
    TY1_REQUIRE(lds <= 64 * 1024, "ty1_head_decode_nms: C=%d needs %zu bytes of LDS (max 65536)", C, lds);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(N), block(kTailThreads);
    ty1::ProfScope prof("head_decode_nms", s, 0.0, (double)N * (4.0 * units * (splits + (y ? 1.0 : 0.0)) + 41.0 * S * S * B));
    switch (B) {
        case 1: head_decode_nms_kernel<1><<<grid, block, lds, s>>>(part, bias, y, N, S, C, ldz, splits, img_size, iou_thr, boxes, score, cls, keep); break;
        case 2: head_decode_nms_kernel<2><<<grid, block, lds, s>>>(part, bias, y, N, S, C, ldz, splits, img_size, iou_thr, boxes, score, cls, keep); break;
        case 3: head_decode_nms_kernel<3><<<grid, block, lds, s>>>(part, bias, y, N, S, C, ldz, splits, img_size, iou_thr, boxes, score, cls, keep); break;
        default: head_decode_nms_kernel<4><<<grid, block, lds, s>>>(part, bias, y, N, S, C, ldz, splits, img_size, iou_thr, boxes, score, cls, keep); break;
    }
    return ty1::check_launch("ty1_head_decode_nms_f32");
}

extern "C" int ty1_pack_detections(const double *boxes, const float *score, const int32_t *cls, const uint8_t *keep,
                                   double *rec, long long slots, ty1_stream_t stream) {
    TY1_REQUIRE(slots >= 0, "ty1_pack_detections: bad slot count");
    if (slots == 0) return 0;
    TY1_REQUIRE(boxes && score && cls && keep && rec, "ty1_pack_detections: null pointer");
    TY1_REQUIRE(((uintptr_t)boxes & 15) == 0 && ((uintptr_t)rec & 15) == 0, "ty1_pack_detections: boxes and rec must be 16-byte aligned");
    TY1_REQUIRE(slots < (1ll << 31) * 256, "ty1_pack_detections: too many slots");
    ty1::ProfScope prof("pack_detections", (hipStream_t)stream, 0.0, 89.0 * (double)slots);
    pack_detections_kernel<<<(unsigned)((slots + 255) / 256), 256, 0, (hipStream_t)stream>>>(boxes, score, cls, keep, rec, slots);
    return ty1::check_launch("ty1_pack_detections");
}
