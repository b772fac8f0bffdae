// Arithmetic of yolo_loss shared by ty1_yolo_loss_fwd_bwd (loss.hip) and the fused training tail ty1_train_tail_f32
// (loss.hip as well; both run exactly these functions, compiled with -ffp-contract=off: same bits).
// Follows /root/reference/yolo/yolo.py:6-107 (C=1 twin /root/reference/faces/trainFace.py:66-126).
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace ty1 {

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

// One grid cell (= one lane of the image's wavefront): forward in the reference's op order, closed-form backward
// (the responsible-box mask carries no gradient).  t: the cell's C+5 label values, p: its C+5B predictions,
// g: its C+5B gradient slots or nullptr.  -> the cell's loss.
template <int B>
__device__ __forceinline__ float loss_cell(const float *t, const float *p, float *g, int lane, int S, int C, float img) {
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
    return acc;
}

}  // namespace ty1
