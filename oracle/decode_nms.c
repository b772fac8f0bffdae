/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or called by the product
 * path (yolo_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may use it, and there only as the checker / the reported CPU baseline.
 *
 * Plain-C restatement of the reference's decode + adaptive threshold + greedy NMS
 * ("yolo_eval"), one image at a time, statement by statement:
 *   /root/reference/tiny_yolov1.py:82-140   (C=20, img_size 448)
 *   /root/reference/faces/testFace.py:51-110 (C=1 twin, img_size 224)
 *   /root/reference/tiny_yolov1.py:36-55    YOLO_head  (xy in float64, wh in float32: SURVEY Q5)
 *   /root/reference/tiny_yolov1.py:58-72    iou        (float64)
 *   /root/reference/utils.py:19-23          X_Y_W_H_To_Min_Max
 * Pinned by tests/golden/nms_c20.npz and nms_c1.npz, which were produced by running the
 * reference's own _main (tests/golden/make_golden.py): kept boxes in raster order with
 * int()-truncated corners, class ids / confidence strings, and the exact sequence of
 * float64 IoU values the reference computed.
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -fno-fast-math (oracle/Makefile).
 * Float arithmetic must stay strict IEEE (no FMA contraction) to mirror NumPy.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#define MAX_SLOTS 1024

/* np.max over floats: NaN propagates. */
static float np_maxf(float a, float b) {
    if (isnan(a)) return a;
    if (isnan(b)) return b;
    return a >= b ? a : b;
}

/* tiny_yolov1.py:58-72, float64 */
static double iou_f64(const double *amin, const double *amax, const double *bmin, const double *bmax) {
    double iw = fmin(amax[0], bmax[0]) - fmax(amin[0], bmin[0]);
    double ih = fmin(amax[1], bmax[1]) - fmax(amin[1], bmin[1]);
    /* np.maximum / np.minimum propagate NaN; fmin/fmax do not, so handle it explicitly */
    if (isnan(amin[0]) || isnan(bmin[0]) || isnan(amax[0]) || isnan(bmax[0])) iw = NAN;
    if (isnan(amin[1]) || isnan(bmin[1]) || isnan(amax[1]) || isnan(bmax[1])) ih = NAN;
    if (!isnan(iw) && !(iw > 0.)) iw = 0.;   /* np.maximum(x, 0.) */
    if (!isnan(ih) && !(ih > 0.)) ih = 0.;
    double inter = iw * ih;
    double pa = (amax[0] - amin[0]) * (amax[1] - amin[1]);
    double ta = (bmax[0] - bmin[0]) * (bmax[1] - bmin[1]);
    double uni = pa + ta - inter;
    return inter / uni;                      /* 0/0 -> NaN, as NumPy */
}

/*
 * pred   : (N, S, S, C + 5B) float32, layout [cls C | conf B | box 4B]   (model_tiny_yolov1.py:14-34)
 * boxes  : (N, S*S*B, 4) float64  x0,y0,x1,y1 in pixels, multiplied by the keep flag (:139-140)
 * score  : (N, S*S*B)   float32   predict_trust after the first filter multiply (:112)
 * cls    : (N, S*S*B)   int32     box_classes * filter_mask (:107)
 * keep   : (N, S*S*B)   uint8     nms_mask
 * trace  : optional float64 buffer receiving every IoU evaluated, in reference order
 * Returns 0, or -1 on bad arguments.
 */
int oracle_decode_nms(const float *pred, int N, int S, int B, int C, int img_size, double iou_thr,
                      double *boxes, float *score, int32_t *cls, uint8_t *keep,
                      double *trace, int64_t trace_cap, int64_t *trace_len) {
    const int cells = S * S, slots = cells * B, D = C + 5 * B;
    if (slots > MAX_SLOTS || N < 0 || S <= 0 || B <= 0 || C <= 0) return -1;
    int64_t tl = 0;
    float bcs[MAX_SLOTS], trust[MAX_SLOTS];
    int filt[MAX_SLOTS], nms[MAX_SLOTS], bcl[MAX_SLOTS];
    double mins[MAX_SLOTS][2], maxs[MAX_SLOTS][2];

    for (int n = 0; n < N; ++n) {
        const float *p = pred + (size_t)n * cells * D;
        /* :90-93  scores = class * trust (f32); argmax (first) and max over classes */
        for (int cell = 0; cell < cells; ++cell) {
            const float *pc = p + (size_t)cell * D;
            for (int k = 0; k < B; ++k) {
                float t = pc[C + k];
                float mp = pc[0] * t;
                int mi = 0;
                if (!isnan(mp)) {
                    for (int m = 1; m < C; ++m) {
                        float v = pc[m] * t;
                        if (!(v <= mp)) { mp = v; mi = m; if (isnan(mp)) break; }
                    }
                }
                bcs[cell * B + k] = mp;
                bcl[cell * B + k] = mi;
            }
        }
        /* :94-99 responsible box per cell, global max / 2 */
        float gmax = bcs[0];
        for (int s = 1; s < slots; ++s) gmax = np_maxf(gmax, bcs[s]);
        float thr = gmax / 2;
        for (int cell = 0; cell < cells; ++cell) {
            float best = bcs[cell * B];
            for (int k = 1; k < B; ++k) best = np_maxf(best, bcs[cell * B + k]);
            for (int k = 0; k < B; ++k) {
                int s = cell * B + k;
                filt[s] = (bcs[s] >= thr) && (bcs[s] >= best);
            }
        }
        /* :103-112 zero what is filtered; decode boxes (YOLO_head + min/max) */
        for (int cell = 0; cell < cells; ++cell) {
            const float *pc = p + (size_t)cell * D;
            int r = cell / S, c = cell % S;
            for (int k = 0; k < B; ++k) {
                int s = cell * B + k;
                float f = (float)filt[s];
                const float *pb = pc + C + B + 4 * k;
                float bx = pb[0] * f, by = pb[1] * f, bw = pb[2] * f, bh = pb[3] * f;
                bcl[s] *= filt[s];
                double x = ((double)bx + (double)c) / (double)S * (double)img_size;
                double y = ((double)by + (double)r) / (double)S * (double)img_size;
                float w = bw * (float)img_size, h = bh * (float)img_size;
                float hw = w / 2, hh = h / 2;
                mins[s][0] = x - (double)hw; mins[s][1] = y - (double)hh;
                maxs[s][0] = x + (double)hw; maxs[s][1] = y + (double)hh;
                trust[s] = pc[C + k] * f;
                nms[s] = 0;
                score[(size_t)n * slots + s] = trust[s];
                cls[(size_t)n * slots + s] = bcl[s];
            }
        }
        /* :113-137 greedy class-agnostic NMS with the tie rule */
        float tmax = trust[0];
        for (int s = 1; s < slots; ++s) tmax = np_maxf(tmax, trust[s]);
        int last = 0;
        while (tmax > 0) {
            for (int s = 0; s < slots; ++s)
                if (trust[s] == tmax) { nms[s] = 1; filt[s] = 0; last = s; }
            for (int s = 0; s < slots; ++s)
                if (filt[s] == 1) {
                    double v = iou_f64(mins[last], maxs[last], mins[s], maxs[s]);
                    if (trace && tl < trace_cap) trace[tl] = v;
                    ++tl;
                    if (v > iou_thr) filt[s] = 0;
                }
            for (int s = 0; s < slots; ++s) trust[s] = trust[s] * (float)filt[s];
            tmax = trust[0];
            for (int s = 1; s < slots; ++s) tmax = np_maxf(tmax, trust[s]);
        }
        /* :139-140 */
        for (int s = 0; s < slots; ++s) {
            double *o = boxes + ((size_t)n * slots + s) * 4;
            double m = (double)nms[s];
            o[0] = mins[s][0] * m; o[1] = mins[s][1] * m;
            o[2] = maxs[s][0] * m; o[3] = maxs[s][1] * m;
            keep[(size_t)n * slots + s] = (uint8_t)nms[s];
        }
    }
    if (trace_len) *trace_len = tl;
    return 0;
}
