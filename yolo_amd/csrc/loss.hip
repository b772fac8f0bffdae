// yolo_head / iou (Keras-backend twins, float32) and the fused yolo_loss forward + backward.
// Replaces /root/reference/yolo/yolo.py:6-107 (C=1 twin /root/reference/faces/trainFace.py:66-126).
//
// Loss kernel: one 64-lane wavefront per image, one lane per grid cell.  The forward follows the
// reference's op order in float32; the backward is the closed form of SURVEY 8a/A8 (the responsible-
// box mask carries no gradient).  Per-image sums use a fixed butterfly, the batch SUM (Q12: not a
// mean) is a second single-block kernel with a fixed order, so results are run-to-run deterministic.
#include "common.h"

#pragma clang fp contract(off)

namespace {

// K.max (tf.reduce_max; torch.amax in the golden generator's backend mapping): a NaN operand gives NaN
__device__ __forceinline__ float max_nan(float a, float b) {
    if (a != a) return a;
    if (b != b) return b;
    return a >= b ? a : b;
}

// d/dw of 5*MR*((sqrt(clip(img*lw, 0)) - sqrt(clip(img*w, 0))) / img)^2 as autodiff delivers it (K.sqrt clips its argument
// at 0 first): the clip passes no gradient for w < 0; at w == 0 the square root's derivative is infinite, so the chain
// rule gives 0 * inf = NaN even where the box is not responsible (MR == 0).
__device__ __forceinline__ float dwh(float MR, float l, float b, float img) {
    if (b < 0.f) return 0.f;
    if (MR > 0.f) return 5.f * MR * (1.f - sqrtf(fmaxf(l, 0.f) / b)) / img;
    return b == 0.f ? __builtin_nanf("") : 0.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// yolo/yolo.py:6-20 on one pair of boxes (float32, K.maximum(x, 0.) clamps)
__device__ __forceinline__ float iou_f32(float pminx, float pminy, float pmaxx, float pmaxy, float tminx, float tminy,
                                         float tmaxx, float tmaxy) {
    float iw = fminf(pmaxx, tmaxx) - fmaxf(pminx, tminx);
    float ih = fminf(pmaxy, tmaxy) - fmaxf(pminy, tminy);
    iw = fmaxf(iw, 0.f);
    ih = fmaxf(ih, 0.f);
    const float inter = iw * ih;
    const float pa = (pmaxx - pminx) * (pmaxy - pminy);
    const float ta = (tmaxx - tminx) * (tmaxy - tminy);
    return inter / (pa + ta - inter);
}

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
        const int r = lane / S, c = lane - r * S;
        const float fS = (float)S;
        const float R = t[C + 4];
        // label box -> pixels (yolo_head :42-43, X_Y_W_H_To_Min_Max)
        const float lx = t[C + 0], ly = t[C + 1], lw = t[C + 2], lh = t[C + 3];
        const float lxy0 = (lx + (float)c) / fS * img, lxy1 = (ly + (float)r) / fS * img;
        const float lwp = lw * img, lhp = lh * img;
        const float lminx = lxy0 - lwp / 2, lminy = lxy1 - lhp / 2, lmaxx = lxy0 + lwp / 2, lmaxy = lxy1 + lhp / 2;

        float ious[B], px[B], py[B], pw[B], ph[B], bx[B], by[B], bw[B], bh[B];
        float best = 0.f;
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const float *pb = p + C + B + 4 * k;
            bx[k] = pb[0]; by[k] = pb[1]; bw[k] = pb[2]; bh[k] = pb[3];
            px[k] = (bx[k] + (float)c) / fS * img;
            py[k] = (by[k] + (float)r) / fS * img;
            pw[k] = bw[k] * img;
            ph[k] = bh[k] * img;
            ious[k] = iou_f32(px[k] - pw[k] / 2, py[k] - ph[k] / 2, px[k] + pw[k] / 2, py[k] + ph[k] / 2, lminx, lminy,
                              lmaxx, lmaxy);
            // K.max :79-80 propagates a NaN IoU (0/0: zero-area label and zero-area box); `>=` below is then False for
            // every box of the cell and none is responsible
            best = (k == 0) ? ious[k] : max_nan(best, ious[k]);
        }
        // class loss :89
        float acc = 0.f;
        for (int m = 0; m < C; ++m) {
            const float d = t[m] - p[m];
            acc += R * (d * d);
            if (g) g[m] = 2.f * R * (p[m] - t[m]);
        }
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const float Mk = (ious[k] >= best) ? 1.f : 0.f;          // :82 (no gradient)
            const float MR = Mk * R;
            const float tk = p[C + k];
            // confidence :84-86
            acc += 0.5f * (1.f - MR) * ((0.f - tk) * (0.f - tk)) + MR * ((1.f - tk) * (1.f - tk));
            // box :101-102
            const float dx = (lxy0 - px[k]) / img, dy = (lxy1 - py[k]) / img;
            const float sw = (sqrtf(fmaxf(lwp, 0.f)) - sqrtf(fmaxf(pw[k], 0.f))) / img;
            const float sh = (sqrtf(fmaxf(lhp, 0.f)) - sqrtf(fmaxf(ph[k], 0.f))) / img;
            acc += 5.f * MR * (dx * dx) + 5.f * MR * (dy * dy) + 5.f * MR * (sw * sw) + 5.f * MR * (sh * sh);
            if (g) {
                g[C + k] = (1.f - MR) * tk - 2.f * MR * (1.f - tk);
                float *gb = g + C + B + 4 * k;
                gb[0] = -10.f * MR * (lx - bx[k]) / (fS * fS);
                gb[1] = -10.f * MR * (ly - by[k]) / (fS * fS);
                gb[2] = dwh(MR, lw, bw[k], img);
                gb[3] = dwh(MR, lh, bh[k], img);
            }
        }
        cell_loss = acc;
    }
    const float total = wave_sum(cell_loss);
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
    out[i] = iou_f32(pmin[i].x, pmin[i].y, pmax[i].x, pmax[i].y, tmin[i].x, tmin[i].y, tmax[i].x, tmax[i].y);
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
