// Image preprocessing on the GPU: uint8 HWC image -> bilinear resize -> [BGR->RGB] -> /255 -> fp32 NHWC slot.
// Replaces the resize + normalise half of /root/reference/utils.py:5-16 (cv2.cvtColor BGR2RGB :11, cv2.resize :13,
// image / 255. :15); decoding the file stays on the host.  SURVEY 8f row 2: at > 10 k images/s the host resize is the
// bottleneck in front of the network.
//
// PARITY UNPINNED: OpenCV is not installed and the reference holds no resized image.  The kernel restates OpenCV's
// generic 8-bit INTER_LINEAR path (the same restatement as oracle/resize_ref.py, which the tests compare it with bit
// for bit, and yolo_amd/utils.py:resize_bilinear_u8):
//   coordinates in float32 with half-pixel centres, s = floor(f), frac = f - s; the x axis snaps taps outside the row to
//   the border pixel (frac = 0), the y axis clamps the two rows and keeps frac;
//   11-bit coefficients short(rint((1 - frac) * 2048)), short(rint(frac * 2048));
//   horizontal pass R = S[x0] * a0 + S[x1] * a1 in int32; vertical pass
//   D = (((b0 * (R0 >> 4)) >> 16) + ((b1 * (R1 >> 4)) >> 16) + 2) >> 2;
//   an exact 2x downscale is OpenCV's 2x2 area average (a + b + c + d + 2) >> 2.
// Compiled with -ffp-contract=off: the coordinate arithmetic must round as NumPy's does.
#include "common.h"

namespace {

struct Tap {
    int i0, i1, w0, w1;
};
// ratio = (double)n_in / (double)n_out, divided once on the host (IEEE division: the same double as on the device): per
// thread the float64 division was most of this kernel's arithmetic
__device__ __forceinline__ Tap tap_of(int o, double ratio, int n_in, bool x_axis) {
    const float f = (float)(((double)o + 0.5) * ratio - 0.5);
    const float fl = floorf(f);
    float frac = f - fl;
    const int s = (int)fl;
    if (x_axis && (s < 0 || s >= n_in - 1)) frac = 0.f;
    const int c0 = s < 0 ? 0 : (s > n_in - 1 ? n_in - 1 : s);
    const int c1 = s + 1 < 0 ? 0 : (s + 1 > n_in - 1 ? n_in - 1 : s + 1);
    return Tap{c0, c1, (int)rintf((1.f - frac) * 2048.f), (int)rintf(frac * 2048.f)};
}

// one thread per output pixel (3 channels): HBM-bound, 12 source bytes in and 12 bytes out per thread
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char *__restrict__ img, int H0, int W0, int swap_rb,
                                                           float *__restrict__ out, int OH, int OW, double ry, double rx) {
    // uchar -> image / 255. in float64 -> Keras' cast to fp32: 256 possible results, computed once per workgroup with that very
    // expression (a float64 division per channel and pixel was half of this kernel's time)
    __shared__ float unit[256];
    unit[threadIdx.x] = (float)((double)threadIdx.x / 255.0);
    __syncthreads();
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= OH * OW) return;
    img += (size_t)blockIdx.y * H0 * W0 * 3;               // blockIdx.y = frame of a batch of equal-sized frames
    out += (size_t)blockIdx.y * OH * OW * 3;
    const int oy = idx / OW, ox = idx - oy * OW;
    const bool area2 = H0 == 2 * OH && W0 == 2 * OW;
    Tap ty, tx;
    if (area2) {
        ty = Tap{2 * oy, 2 * oy + 1, 0, 0};
        tx = Tap{2 * ox, 2 * ox + 1, 0, 0};
    } else {
        ty = tap_of(oy, ry, H0, false);
        tx = tap_of(ox, rx, W0, true);
    }
    const unsigned char *r0 = img + (size_t)ty.i0 * W0 * 3, *r1 = img + (size_t)ty.i1 * W0 * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int sc = swap_rb ? 2 - c : c;
        const int p00 = r0[tx.i0 * 3 + sc], p01 = r0[tx.i1 * 3 + sc], p10 = r1[tx.i0 * 3 + sc], p11 = r1[tx.i1 * 3 + sc];
        int v;
        if (area2) {
            v = (p00 + p01 + p10 + p11 + 2) >> 2;
        } else {
            const int a = (p00 * tx.w0 + p01 * tx.w1) >> 4, b = (p10 * tx.w0 + p11 * tx.w1) >> 4;
            v = (((ty.w0 * a) >> 16) + ((ty.w1 * b) >> 16) + 2) >> 2;
        }
        out[(size_t)idx * 3 + c] = unit[v & 255];
    }
}

}  // namespace

extern "C" int ty1_preprocess_batch_u8(const unsigned char *imgs, int n, int H0, int W0, int swap_rb, float *out, int out_h,
                                       int out_w, ty1_stream_t stream) {
    TY1_REQUIRE(n >= 0 && n <= 65535, "ty1_preprocess_batch_u8: n=%d not in 0..65535", n);
    TY1_REQUIRE(H0 > 0 && W0 > 0 && out_h > 0 && out_w > 0, "ty1_preprocess_batch_u8: bad dimensions");
    if (n == 0) return 0;
    TY1_REQUIRE(imgs && out, "ty1_preprocess_batch_u8: null pointer");
    TY1_REQUIRE((long long)out_h * out_w < (1ll << 31) && (long long)H0 * W0 * 3 < (1ll << 40), "ty1_preprocess_batch_u8: image too large");
    hipStream_t s = (hipStream_t)stream;
    ty1::ProfScope prof("preprocess_batch_u8", s, 0.0, (double)n * (12.0 * out_h * out_w + 3.0 * H0 * W0));
    preprocess_u8_kernel<<<dim3((unsigned)(((long long)out_h * out_w + 255) / 256), (unsigned)n), 256, 0, s>>>(imgs, H0, W0, swap_rb, out,
                                                                                                            out_h, out_w, (double)H0 / (double)out_h, (double)W0 / (double)out_w);
    return ty1::check_launch("ty1_preprocess_batch_u8");
}

extern "C" int ty1_preprocess_u8(const unsigned char *img, int H0, int W0, int swap_rb, float *out, int out_h, int out_w,
                                 ty1_stream_t stream) {
    TY1_REQUIRE(H0 > 0 && W0 > 0 && out_h > 0 && out_w > 0, "ty1_preprocess_u8: bad dimensions");
    TY1_REQUIRE(img && out, "ty1_preprocess_u8: null pointer");
    TY1_REQUIRE((long long)out_h * out_w < (1ll << 31) && (long long)H0 * W0 * 3 < (1ll << 40), "ty1_preprocess_u8: image too large");
    hipStream_t s = (hipStream_t)stream;
    ty1::ProfScope prof("preprocess_u8", s, 0.0, 12.0 * out_h * out_w + 3.0 * H0 * W0);
    preprocess_u8_kernel<<<(unsigned)(((long long)out_h * out_w + 255) / 256), 256, 0, s>>>(img, H0, W0, swap_rb, out, out_h, out_w, (double)H0 / (double)out_h,
                                                                                            (double)W0 / (double)out_w);
    return ty1::check_launch("ty1_preprocess_u8");
}
