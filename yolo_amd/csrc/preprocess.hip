// Image preprocessing on the GPU: uint8 HWC image -> bilinear resize -> [BGR->RGB] -> /255 -> fp32 NHWC slot.
// Replaces the resize + normalise half of /root/reference/utils.py:5-16 (cv2.cvtColor BGR2RGB :11, cv2.resize :13,
// image / 255. :15); decoding the file stays on the host.  SURVEY 8f row 2: at > 10 k images/s the host resize is the
// bottleneck in front of the network.
//
// PARITY UNPINNED: OpenCV is not installed, so this follows the published definition of cv2.resize INTER_LINEAR on
// 8-bit images (half-pixel centres: src = (dst + 0.5) * in/out - 0.5, edge clamp, coefficients rounded to 11 bits,
// 22-bit product rounded half up) exactly as yolo_amd/utils.py:resize_bilinear_u8 restates it; the tests compare
// this kernel with that host restatement bit for bit.
// Compiled with -ffp-contract=off: the coordinate arithmetic is float64 in NumPy's operation order.
#include "common.h"

namespace {

struct Tap {
    int i0, i1, w0, w1;
};
__device__ __forceinline__ Tap tap_of(int o, int n_out, int n_in) {
    const double f = ((double)o + 0.5) * ((double)n_in / (double)n_out) - 0.5;
    const double fl = floor(f);
    long long i0 = (long long)fl;
    double frac = f - fl;
    if (i0 < 0) frac = 0.0;
    if (i0 >= n_in - 1) frac = 0.0;
    const long long c0 = i0 < 0 ? 0 : (i0 > n_in - 1 ? n_in - 1 : i0);
    const long long c1 = i0 + 1 < 0 ? 0 : (i0 + 1 > n_in - 1 ? n_in - 1 : i0 + 1);
    const int w1 = (int)rint(frac * 2048.0);
    return Tap{(int)c0, (int)c1, 2048 - w1, w1};
}

// one thread per output pixel (3 channels): HBM-bound, 12 source bytes in and 12 bytes out per thread
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char *__restrict__ img, int H0, int W0, int swap_rb,
                                                           float *__restrict__ out, int OH, int OW) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= OH * OW) return;
    const int oy = idx / OW, ox = idx - oy * OW;
    const Tap ty = tap_of(oy, OH, H0), tx = tap_of(ox, OW, W0);
    const unsigned char *r0 = img + (size_t)ty.i0 * W0 * 3, *r1 = img + (size_t)ty.i1 * W0 * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int sc = swap_rb ? 2 - c : c;
        const long long a = (long long)r0[tx.i0 * 3 + sc] * tx.w0 + (long long)r0[tx.i1 * 3 + sc] * tx.w1;
        const long long b = (long long)r1[tx.i0 * 3 + sc] * tx.w0 + (long long)r1[tx.i1 * 3 + sc] * tx.w1;
        const long long v = (a * ty.w0 + b * ty.w1 + (1ll << 21)) >> 22;
        out[(size_t)idx * 3 + c] = (float)((double)v / 255.0);      // image / 255. in float64, then Keras' cast to fp32
    }
}

}  // namespace

extern "C" int ty1_preprocess_u8(const unsigned char *img, int H0, int W0, int swap_rb, float *out, int out_h, int out_w,
                                 ty1_stream_t stream) {
    TY1_REQUIRE(H0 > 0 && W0 > 0 && out_h > 0 && out_w > 0, "ty1_preprocess_u8: bad dimensions");
    TY1_REQUIRE(img && out, "ty1_preprocess_u8: null pointer");
    TY1_REQUIRE((long long)out_h * out_w < (1ll << 31) && (long long)H0 * W0 * 3 < (1ll << 40), "ty1_preprocess_u8: image too large");
    hipStream_t s = (hipStream_t)stream;
    ty1::ProfScope prof("preprocess_u8", s, 0.0, 12.0 * out_h * out_w + 3.0 * H0 * W0);
    preprocess_u8_kernel<<<(unsigned)(((long long)out_h * out_w + 255) / 256), 256, 0, s>>>(img, H0, W0, swap_rb, out, out_h, out_w);
    return ty1::check_launch("ty1_preprocess_u8");
}
