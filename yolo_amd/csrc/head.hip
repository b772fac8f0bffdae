// Head activations (ReshapeYOLO.call) fused with the Dense split-K reduction and bias add, and the
// matching backward (softmax / sigmoid Jacobians).
// Replaces /root/reference/models/model_tiny_yolov1.py:14-34 (C=1 twin faces/trainFace.py:38-58).
//
// Layouts: Dense output is PLANAR: z[0 : S*S*C] -> (S,S,C), z[S*S*C : +S*S*B] -> (S,S,B),
// rest -> (S,S,4B); the activated tensor is cell-major (N,S,S,C+5B) = [cls C | conf B | box 4B].
// z is reduced over the split-K partials into LDS once.
#include "common.h"

namespace {

// Both branches keep exp() finite: 1 / (1 + exp(-v)) alone returns exactly 0 from v < -88.7 on (exp overflows), while the
// true value stays a positive fp32 number down to v = -103.9 - and a box width of exactly 0 is where yolo_loss' sqrt has
// no finite derivative (csrc/loss.hip: dwh).
__device__ __forceinline__ float sigmoidf_(float v) {
    if (v >= 0.f) return 1.0f / (1.0f + expf(-v));
    const float e = expf(v);
    return e / (1.0f + e);
}

// grid (N, ceil(cells / kCellsPerBlock)): a block owns a few cells of one image, so even a small batch
// spreads over the chip and the split-K reduction (tens of partial slabs) runs as independent loads.
constexpr int kCellsPerBlock = 7;

__global__ __launch_bounds__(256) void head_activate_kernel(const float *__restrict__ part,
                                                             const float *__restrict__ bias, float *__restrict__ y,
                                                             int N, int S, int B, int C, int ldz, int splits) {
    extern __shared__ float z[];                       // [cells_here][D]  cell-major, already in the output layout
    const int n = blockIdx.x, cells = S * S, D = C + 5 * B, units = cells * D;
    const int c0 = blockIdx.y * kCellsPerBlock, nc = min(kCellsPerBlock, cells - c0);
    const int i1 = cells * C, i2 = i1 + cells * B;
    const size_t slab = (size_t)N * ldz;
    const float *pn = part + (size_t)n * ldz;
    for (int t = threadIdx.x; t < nc * D; t += blockDim.x) {
        const int lc = t / D, d = t - lc * D, cell = c0 + lc;
        // planar Dense unit of (cell, channel d)
        const int u = d < C ? cell * C + d : (d < C + B ? i1 + cell * B + (d - C) : i2 + cell * 4 * B + (d - C - B));
        float a0 = bias ? bias[u] : 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int s = 0;
        for (; s + 4 <= splits; s += 4) {
            a0 += pn[(size_t)s * slab + u];
            a1 += pn[(size_t)(s + 1) * slab + u];
            a2 += pn[(size_t)(s + 2) * slab + u];
            a3 += pn[(size_t)(s + 3) * slab + u];
        }
        for (; s < splits; ++s) a0 += pn[(size_t)s * slab + u];
        z[t] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    float *out = y + (size_t)n * units + (size_t)c0 * D;
    for (int t = threadIdx.x; t < nc * D; t += blockDim.x) {
        const int lc = t / D, d = t - lc * D;
        const float *zc = z + lc * D;
        float r;
        if (d < C) {                                   // softmax over the C classes of the cell
            float mx = zc[0];
            for (int m = 1; m < C; ++m) mx = fmaxf(mx, zc[m]);
            float sum = 0.f;
            for (int m = 0; m < C; ++m) sum += expf(zc[m] - mx);
            r = expf(zc[d] - mx) / sum;
        } else {
            r = sigmoidf_(zc[d]);
        }
        out[t] = r;
    }
}

__global__ __launch_bounds__(256) void head_activate_bwd_kernel(const float *__restrict__ y,
                                                                 const float *__restrict__ gy, float *__restrict__ dz,
                                                                 int N, int S, int B, int C, int ldz) {
    const int n = blockIdx.x, cells = S * S, D = C + 5 * B, units = cells * D;
    const float *yn = y + (size_t)n * units, *gn = gy + (size_t)n * units;
    float *dn = dz + (size_t)n * ldz;
    const int i1 = cells * C, i2 = i1 + cells * B;
    for (int cell = threadIdx.x; cell < cells; cell += blockDim.x) {
        float dot = 0.f;
        for (int m = 0; m < C; ++m) dot += gn[cell * D + m] * yn[cell * D + m];
        for (int m = 0; m < C; ++m) dn[cell * C + m] = yn[cell * D + m] * (gn[cell * D + m] - dot);
    }
    for (int i = threadIdx.x; i < cells * B; i += blockDim.x) {
        const int cell = i / B, k = i - cell * B;
        const float s = yn[cell * D + C + k];
        dn[i1 + i] = gn[cell * D + C + k] * s * (1.f - s);
    }
    for (int i = threadIdx.x; i < cells * 4 * B; i += blockDim.x) {
        const int cell = i / (4 * B), k = i - cell * (4 * B);
        const float s = yn[cell * D + C + B + k];
        dn[i2 + i] = gn[cell * D + C + B + k] * s * (1.f - s);
    }
    for (int u = units + threadIdx.x; u < ldz; u += blockDim.x) dn[u] = 0.f;
}

}  // namespace

extern "C" int ty1_head_activate_f32(const float *part, const float *bias, float *y, int N, int S, int B, int C,
                                     int ldz, int splits, ty1_stream_t stream) {
    TY1_REQUIRE(N >= 0 && S > 0 && B > 0 && C > 0 && splits > 0, "ty1_head_activate: bad dimensions");
    if (N == 0) return 0;
    const int units = S * S * (C + 5 * B);
    TY1_REQUIRE(part && y, "ty1_head_activate: null pointer");
    TY1_REQUIRE(ldz >= units, "ty1_head_activate: ldz=%d < units=%d", ldz, units);
    const int D = C + 5 * B;
    TY1_REQUIRE((size_t)kCellsPerBlock * D * 4 <= 64 * 1024, "ty1_head_activate: %d channels per cell exceed the LDS budget", D);
    hipStream_t s = (hipStream_t)stream;
    ty1::ProfScope prof("head_activate", s, 0.0, 4.0 * N * units * (splits + 1.0));
    head_activate_kernel<<<dim3(N, (S * S + kCellsPerBlock - 1) / kCellsPerBlock), dim3(256), (size_t)kCellsPerBlock * D * 4, s>>>(
        part, bias, y, N, S, B, C, ldz, splits);
    return ty1::check_launch("ty1_head_activate_f32");
}

extern "C" int ty1_head_activate_bwd_f32(const float *y, const float *grad_y, float *dz, int N, int S, int B, int C,
                                         int ldz, ty1_stream_t stream) {
    TY1_REQUIRE(N >= 0 && S > 0 && B > 0 && C > 0, "ty1_head_activate_bwd: bad dimensions");
    if (N == 0) return 0;
    TY1_REQUIRE(y && grad_y && dz, "ty1_head_activate_bwd: null pointer");
    TY1_REQUIRE(ldz >= S * S * (C + 5 * B), "ty1_head_activate_bwd: ldz too small");
    hipStream_t s = (hipStream_t)stream;
    ty1::ProfScope prof("head_activate_bwd", s);
    head_activate_bwd_kernel<<<dim3(N), dim3(256), 0, s>>>(y, grad_y, dz, N, S, B, C, ldz);
    return ty1::check_launch("ty1_head_activate_bwd_f32");
}
