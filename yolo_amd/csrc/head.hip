// Head activations (ReshapeYOLO.call) fused with the Dense split-K reduction and bias add, and the
// matching backward (softmax / sigmoid Jacobians).
// Replaces /root/reference/models/model_tiny_yolov1.py:14-34 (C=1 twin faces/trainFace.py:38-58).
//
// Layouts: Dense output is PLANAR: z[0 : S*S*C] -> (S,S,C), z[S*S*C : +S*S*B] -> (S,S,B),
// rest -> (S,S,4B); the activated tensor is cell-major (N,S,S,C+5B) = [cls C | conf B | box 4B].
// z is reduced over the split-K partials into LDS once.
#include "common.h"
#include "head_math.h"

namespace {

// grid (N, ceil(cells / kCellsPerBlock)): a block owns a few cells of one image, so even a small batch
// spreads over the chip and the split-K reduction (tens of partial slabs) runs as independent loads.
constexpr int kCellsPerBlock = 7;

__global__ __launch_bounds__(256) void head_activate_kernel(const float *__restrict__ part,
                                                             const float *__restrict__ bias, float *__restrict__ y,
                                                             int N, int S, int B, int C, int ldz, int splits) {
    extern __shared__ float z[];                       // [cells_here][D]  cell-major, already in the output layout
    const int n = blockIdx.x, cells = S * S, D = C + 5 * B, units = cells * D;
    const int c0 = blockIdx.y * kCellsPerBlock, nc = min(kCellsPerBlock, cells - c0);
    const size_t slab = (size_t)N * ldz;
    const float *pn = part + (size_t)n * ldz;
    for (int t = threadIdx.x; t < nc * D; t += blockDim.x) {
        const int lc = t / D, d = t - lc * D, cell = c0 + lc;
        z[t] = ty1::reduce_unit(pn, slab, ty1::planar_unit(cell, d, cells, B, C), bias, splits);
    }
    __syncthreads();
    float *out = y + (size_t)n * units + (size_t)c0 * D;
    for (int t = threadIdx.x; t < nc * D; t += blockDim.x) {
        const int lc = t / D, d = t - lc * D;
        out[t] = ty1::activate_channel(z + lc * D, d, C);
    }
}

__global__ __launch_bounds__(256) void head_activate_bwd_kernel(const float *__restrict__ y,
                                                                 const float *__restrict__ gy, float *__restrict__ dz,
                                                                 int N, int S, int B, int C, int ldz) {
    const int n = blockIdx.x, cells = S * S, units = cells * (C + 5 * B);
    ty1::head_bwd_image(y + (size_t)n * units, gy + (size_t)n * units, dz + (size_t)n * ldz, cells, B, C, ldz, (int)threadIdx.x,
                        (int)blockDim.x);
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
