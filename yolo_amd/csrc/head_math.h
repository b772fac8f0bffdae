// Arithmetic of the Dense tail shared by ty1_head_activate_f32 (head.hip) and the fused inference tail
// ty1_head_decode_nms_f32 (decode_nms.hip): split-K reduction + bias, and ReshapeYOLO.call
// (/root/reference/models/model_tiny_yolov1.py:14-34).  Both translation units are compiled with -ffp-contract=off and
// run exactly these functions, so the two entry points produce the same bits.
#pragma once
#include <hip/hip_runtime.h>

namespace ty1 {

// Both branches keep exp() finite: 1 / (1 + exp(-v)) alone returns exactly 0 from v < -88.7 on (exp overflows), while the
// true value stays a positive fp32 number down to v = -103.9 - and a box width of exactly 0 is where yolo_loss' sqrt has
// no finite derivative (csrc/loss.hip: dwh).
__device__ __forceinline__ float sigmoid_f32(float v) {
    if (v >= 0.f) return 1.0f / (1.0f + expf(-v));
    const float e = expf(v);
    return e / (1.0f + e);
}

// planar Dense unit of (cell, channel d) of the cell-major head tensor
__device__ __forceinline__ int planar_unit(int cell, int d, int cells, int B, int C) {
    const int i1 = cells * C, i2 = i1 + cells * B;
    return d < C ? cell * C + d : (d < C + B ? i1 + cell * B + (d - C) : i2 + cell * 4 * B + (d - C - B));
}

// z[u] = bias[u] + sum over the split-K partial slabs, four independent chains (the loads of a thread are in flight together)
__device__ __forceinline__ float reduce_unit(const float *__restrict__ pn, size_t slab, int u, const float *__restrict__ bias,
                                             int splits) {
    float a0 = bias ? bias[u] : 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int s = 0;
    for (; s + 4 <= splits; s += 4) {
        a0 += pn[(size_t)s * slab + u];
        a1 += pn[(size_t)(s + 1) * slab + u];
        a2 += pn[(size_t)(s + 2) * slab + u];
        a3 += pn[(size_t)(s + 3) * slab + u];
    }
    for (; s < splits; ++s) a0 += pn[(size_t)s * slab + u];
    return (a0 + a1) + (a2 + a3);
}

// channel d of one cell from the cell's D logits zc: softmax over the C classes (max-subtracted), sigmoid for the rest
__device__ __forceinline__ float activate_channel(const float *zc, int d, int C) {
    if (d < C) {
        float mx = zc[0];
        for (int m = 1; m < C; ++m) mx = fmaxf(mx, zc[m]);
        float sum = 0.f;
        for (int m = 0; m < C; ++m) sum += expf(zc[m] - mx);
        return expf(zc[d] - mx) / sum;
    }
    return sigmoid_f32(zc[d]);
}

// Backward of ReshapeYOLO.call for one image: softmax Jacobian for the classes, s(1 - s) for the sigmoids; yn / gn are
// the image's activated tensor and its gradient (cell-major, D = C + 5B per cell, global memory or LDS), dn the image's
// planar dz row of ldz floats (zero-filled behind the units).  Thread tid of nthreads.
__device__ __forceinline__ void head_bwd_image(const float *yn, const float *gn, float *dn, int cells, int B, int C, int ldz,
                                               int tid, int nthreads) {
    const int D = C + 5 * B, units = cells * D;
    const int i1 = cells * C, i2 = i1 + cells * B;
    for (int cell = tid; cell < cells; cell += nthreads) {
        float dot = 0.f;
        for (int m = 0; m < C; ++m) dot += gn[cell * D + m] * yn[cell * D + m];
        for (int m = 0; m < C; ++m) dn[cell * C + m] = yn[cell * D + m] * (gn[cell * D + m] - dot);
    }
    for (int i = tid; i < cells * B; i += nthreads) {
        const int cell = i / B, k = i - cell * B;
        const float s = yn[cell * D + C + k];
        dn[i1 + i] = gn[cell * D + C + k] * s * (1.f - s);
    }
    for (int i = tid; i < cells * 4 * B; i += nthreads) {
        const int cell = i / (4 * B), k = i - cell * (4 * B);
        const float s = yn[cell * D + C + B + k];
        dn[i2 + i] = gn[cell * D + C + B + k] * s * (1.f - s);
    }
    for (int u = units + tid; u < ldz; u += nthreads) dn[u] = 0.f;
}

}  // namespace ty1
