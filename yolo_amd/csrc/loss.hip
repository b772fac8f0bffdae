// yolo_head / iou (Keras-backend twins, float32) and the fused yolo_loss forward + backward.
// Replaces /root/reference/yolo/yolo.py:6-107 (C=1 twin /root/reference/faces/trainFace.py:66-126).
//
// Loss kernel: one 64-lane wavefront per image, one lane per grid cell.  The forward follows the
// reference's op order in float32; the backward is the closed form of SURVEY 8a/A8 (the responsible-
// box mask carries no gradient).  Per-image sums use a fixed butterfly, the batch SUM (Q12: not a
// mean) is a second single-block kernel with a fixed order, so results are run-to-run deterministic.
#include "common.h"
#include "head_math.h"
#include "loss_math.h"

#pragma clang fp contract(off)

namespace {

template <int B>
__global__ __launch_bounds__(256) void yolo_loss_kernel(const float *__restrict__ y_true,
                                                        const float *__restrict__ y_pred, int N, int S, int C,
                                                        float img, float *__restrict__ loss_per_image,
                                                        float *__restrict__ grad) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;                                   // wave-uniform, no block barrier below
    const int cells = S * S, Dt = C + 5, Dp = C + 5 * B;
    float cell_loss = 0.f;
    if (lane < cells) {
        const float *t = y_true + ((size_t)n * cells + lane) * Dt;
        const float *p = y_pred + ((size_t)n * cells + lane) * Dp;
        float *g = grad ? grad + ((size_t)n * cells + lane) * Dp : nullptr;
        cell_loss = ty1::loss_cell<B>(t, p, g, lane, S, C, img);
    }
    const float total = ty1::wave_sum(cell_loss);
    if (lane == 0) loss_per_image[n] = total;
}

__global__ __launch_bounds__(256) void sum_fixed_order_kernel(const float *__restrict__ v, int n, float *__restrict__ out) {
    __shared__ float part[256];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += v[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0];
}

// The training tail behind the Dense GEMM in one launch (ty1_train_tail_f32): ty1_head_activate_f32, ty1_yolo_loss_fwd_bwd's
// per-image part and ty1_head_activate_bwd_f32 with the head tensor and its gradient kept in LDS.  One 1024-thread
// workgroup per image: all sixteen waves reduce the split-K partials and activate the logits (head_math.h), wave 0 runs
// the loss cell code (loss_math.h), all waves apply the head Jacobian and write the planar dz row.  Same functions, same
// order of operations as the three separate kernels: same bits.
constexpr int kTrainTailThreads = 1024;
template <int B>
__global__ __launch_bounds__(kTrainTailThreads) void train_tail_kernel(const float *__restrict__ part, const float *__restrict__ bias,
                                                                       const float *__restrict__ y_true, float *__restrict__ y,
                                                                       float *__restrict__ dz, int N, int S, int C, int ldz_part,
                                                                       int splits, float img, int ldz, float *__restrict__ loss_per_image) {
    extern __shared__ float lds[];
    const int n = blockIdx.x, cells = S * S, D = C + 5 * B, units = cells * D;
    float *z = lds, *yl = lds + ((units + 1) & ~1), *gl = yl + ((units + 1) & ~1), *smx = gl + ((units + 1) & ~1), *ssum = smx + 64;
    const size_t slab = (size_t)N * ldz_part;
    const float *pn = part + (size_t)n * ldz_part;
    for (int t = threadIdx.x; t < units; t += kTrainTailThreads) {
        const int cell = t / D, d = t - cell * D;
        z[t] = ty1::reduce_unit(pn, slab, ty1::planar_unit(cell, d, cells, B, C), bias, splits);
    }
    __syncthreads();
    // softmax maximum and denominator once per cell, in activate_channel's order (same bits)
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
    for (int t = threadIdx.x; t < units; t += kTrainTailThreads) {
        const int cell = t / D, d = t - cell * D;
        const float v = d < C ? expf(z[t] - smx[cell]) / ssum[cell] : ty1::sigmoid_f32(z[t]);
        yl[t] = v;
        if (y) y[(size_t)n * units + t] = v;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        float cell_loss = 0.f;
        if (lane < cells)
            cell_loss = ty1::loss_cell<B>(y_true + ((size_t)n * cells + lane) * (C + 5), yl + lane * D, gl + lane * D, lane, S, C, img);
        const float total = ty1::wave_sum(cell_loss);
        if (lane == 0) loss_per_image[n] = total;
    }
    __syncthreads();
    ty1::head_bwd_image(yl, gl, dz + (size_t)n * ldz, cells, B, C, ldz, (int)threadIdx.x, kTrainTailThreads);
}

__global__ void yolo_head_kernel(const float *__restrict__ feats, float *__restrict__ xy, float *__restrict__ wh,
                                 long long nbox, int S, int B, float img) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbox) return;
    const long long cell = i / B;
    const int c = (int)(cell % S), r = (int)((cell / S) % S);
    const float4 f = reinterpret_cast<const float4 *>(feats)[i];
    const float fS = (float)S;
    reinterpret_cast<float2 *>(xy)[i] = make_float2((f.x + (float)c) / fS * img, (f.y + (float)r) / fS * img);
    reinterpret_cast<float2 *>(wh)[i] = make_float2(f.z * img, f.w * img);
}

__global__ void iou_kernel(const float2 *__restrict__ pmin, const float2 *__restrict__ pmax,
                           const float2 *__restrict__ tmin, const float2 *__restrict__ tmax, float *__restrict__ out,
                           long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = ty1::iou_f32(pmin[i].x, pmin[i].y, pmax[i].x, pmax[i].y, tmin[i].x, tmin[i].y, tmax[i].x, tmax[i].y);
}

}  // namespace

extern "C" int ty1_yolo_loss_fwd_bwd(const float *y_true, const float *y_pred, int N, int S, int B, int C, int img_size,
                                     float *loss_per_image, float *loss, float *grad, ty1_stream_t stream) {
    TY1_REQUIRE(N >= 0 && S > 0 && C > 0 && img_size > 0, "ty1_yolo_loss: bad dimensions");
    TY1_REQUIRE(S * S <= 64, "ty1_yolo_loss: S*S=%d exceeds one wavefront", S * S);
    TY1_REQUIRE(B >= 1 && B <= 4, "ty1_yolo_loss: B=%d not in 1..4", B);
    TY1_REQUIRE(loss, "ty1_yolo_loss: null loss pointer");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) return hipMemsetAsync(loss, 0, sizeof(float), s) == hipSuccess ? 0 : ty1::fail("ty1_yolo_loss: memset failed");
    TY1_REQUIRE(y_true && y_pred && loss_per_image, "ty1_yolo_loss: null pointer");
    dim3 grid((N + 3) / 4), block(256);
    {
        ty1::ProfScope prof("yolo_loss", s);
        switch (B) {
            case 1: yolo_loss_kernel<1><<<grid, block, 0, s>>>(y_true, y_pred, N, S, C, (float)img_size, loss_per_image, grad); break;
            case 2: yolo_loss_kernel<2><<<grid, block, 0, s>>>(y_true, y_pred, N, S, C, (float)img_size, loss_per_image, grad); break;
            case 3: yolo_loss_kernel<3><<<grid, block, 0, s>>>(y_true, y_pred, N, S, C, (float)img_size, loss_per_image, grad); break;
            default: yolo_loss_kernel<4><<<grid, block, 0, s>>>(y_true, y_pred, N, S, C, (float)img_size, loss_per_image, grad); break;
        }
    }
    if (int rc = ty1::check_launch("ty1_yolo_loss_fwd_bwd")) return rc;
    {
        ty1::ProfScope prof("loss_sum", s);
        sum_fixed_order_kernel<<<1, 256, 0, s>>>(loss_per_image, N, loss);
    }
    return ty1::check_launch("ty1_yolo_loss_fwd_bwd(sum)");
}

extern "C" int ty1_train_tail_f32(const float *part, const float *bias, const float *y_true, float *y, float *dz, int N, int S,
                                  int B, int C, int ldz_part, int splits, int img_size, int ldz, float *loss_per_image,
                                  float *loss, ty1_stream_t stream) {
    TY1_REQUIRE(N >= 0 && S > 0 && C > 0 && img_size > 0 && splits > 0, "ty1_train_tail: bad dimensions");
    TY1_REQUIRE(S * S <= 64, "ty1_train_tail: S*S=%d exceeds one wavefront", S * S);
    TY1_REQUIRE(B >= 1 && B <= 4, "ty1_train_tail: B=%d not in 1..4", B);
    TY1_REQUIRE(loss, "ty1_train_tail: null loss pointer");
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) return hipMemsetAsync(loss, 0, sizeof(float), s) == hipSuccess ? 0 : ty1::fail("ty1_train_tail: memset failed");
    const int units = S * S * (C + 5 * B);
    TY1_REQUIRE(part && y_true && dz && loss_per_image, "ty1_train_tail: null pointer");
    TY1_REQUIRE(ldz_part >= units && ldz >= units, "ty1_train_tail: leading dimensions (%d, %d) < units=%d", ldz_part, ldz, units);
    const size_t lds = (size_t)(3 * ((units + 1) & ~1) + 128) * 4;
    TY1_REQUIRE(lds <= 64 * 1024, "ty1_train_tail: C=%d needs %zu bytes of LDS (max 65536)", C, lds);
    dim3 grid(N), block(kTrainTailThreads);
    {
        ty1::ProfScope prof("train_tail", s, 0.0, (double)N * 4.0 * (units * (splits + 1.0) + S * S * (C + 5) + ldz));
#define TY1_TT(BV) train_tail_kernel<BV><<<grid, block, lds, s>>>(part, bias, y_true, y, dz, N, S, C, ldz_part, splits, (float)img_size, ldz, loss_per_image)
        switch (B) {
            case 1: TY1_TT(1); break;
            case 2: TY1_TT(2); break;
            case 3: TY1_TT(3); break;
            default: TY1_TT(4); break;
        }
#undef TY1_TT
    }
    if (int rc = ty1::check_launch("ty1_train_tail_f32")) return rc;
    {
        ty1::ProfScope prof("loss_sum", s);
        sum_fixed_order_kernel<<<1, 256, 0, s>>>(loss_per_image, N, loss);
    }
    return ty1::check_launch("ty1_train_tail_f32(sum)");
}

extern "C" int ty1_yolo_head_f32(const float *feats, float *xy, float *wh, int N, int S, int B, int img_size,
                                 ty1_stream_t stream) {
    TY1_REQUIRE(N >= 0 && S > 0 && B > 0 && img_size > 0, "ty1_yolo_head: bad dimensions");
    const long long nbox = (long long)N * S * S * B;
    if (nbox == 0) return 0;
    TY1_REQUIRE(feats && xy && wh, "ty1_yolo_head: null pointer");
    TY1_REQUIRE(((uintptr_t)feats & 15) == 0 && ((uintptr_t)xy & 7) == 0 && ((uintptr_t)wh & 7) == 0,
                "ty1_yolo_head: misaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    ty1::ProfScope prof("yolo_head", s);
    yolo_head_kernel<<<dim3((unsigned)((nbox + 255) / 256)), dim3(256), 0, s>>>(feats, xy, wh, nbox, S, B, (float)img_size);
    return ty1::check_launch("ty1_yolo_head_f32");
}

extern "C" int ty1_iou_f32(const float *pred_mins, const float *pred_maxes, const float *true_mins,
                           const float *true_maxes, float *out, int64_t n, ty1_stream_t stream) {
    TY1_REQUIRE(n >= 0, "ty1_iou: negative count");
    if (n == 0) return 0;
    TY1_REQUIRE(pred_mins && pred_maxes && true_mins && true_maxes && out, "ty1_iou: null pointer");
    TY1_REQUIRE((((uintptr_t)pred_mins | (uintptr_t)pred_maxes | (uintptr_t)true_mins | (uintptr_t)true_maxes) & 7) == 0,
                "ty1_iou: misaligned pointer");
    hipStream_t s = (hipStream_t)stream;
    ty1::ProfScope prof("iou", s);
    iou_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(
        (const float2 *)pred_mins, (const float2 *)pred_maxes, (const float2 *)true_mins, (const float2 *)true_maxes, out, n);
    return ty1::check_launch("ty1_iou_f32");
}
