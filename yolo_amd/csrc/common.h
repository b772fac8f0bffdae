// Shared host-side helpers of libty1.so: error reporting, launch checks, event profiling.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/ty1.h"

namespace ty1 {

std::string &last_error();
int fail(const char *fmt, ...);

// Event bracket around one kernel launch (only when profiling is enabled on this thread).
struct ProfScope {
    // flops / bytes: ALGORITHMIC work of this launch (for the roofline figure), not measured traffic
    ProfScope(const char *name, hipStream_t s, double flops = 0.0, double bytes = 0.0);
    ~ProfScope();
    std::string name;
    hipStream_t stream;
    double flops, bytes;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool on;
};

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

// Developer overrides (TY1_TILE, TY1_DBG, TY1_NO_TAIL_SPLIT ...) exist in the dev build only (tools/build_dev.sh,
// -DTY1_DEV_STAMPS -> libty1_dev.so): the shipping library reads no environment variable.
#ifdef TY1_DEV_STAMPS
inline const char *dev_env(const char *name) { return getenv(name); }
#else
inline const char *dev_env(const char *) { return nullptr; }
#endif

// CU count of the CURRENT device, cached per device id (one process may drive several GPUs; safe from several host
// threads).  Plans (tile choice, K-split pieces, Dense splits) are made for this count.
int num_cus();
// CUs a persistent grid may fill: num_cus() minus the count set with ty1_set_reserved_cus (at least 8).  Only the
// NUMBER OF WORKGROUPS of a launch follows it; the work items and their order of summation do not, so results are
// bit-identical for every setting.
int grid_cus();

// developer timeline buffer (8 x i64 per workgroup), set by ty1_debug_stamp_buffer; see tools/stamp_trace.py
extern long long *g_stamp;
extern long long g_stamp_cap;

// conv_direct.hip
int conv3x3_c3_persistent(const float *x, const float *w, const float *scale, const float *shift, void *y, int N, int H,
                          int W, int Cout, int pool, float alpha, int out_bf16, hipStream_t s);
int conv3x3_c16_direct_bf16(const void *x, const void *w, const float *scale, const float *shift, void *y, int N, int H,
                            int W, int Cout, int pool, float alpha, hipStream_t s);
int conv3x3_c16_direct(const float *x, const float *w, const float *scale, const float *shift, float *y, int N, int H,
                       int W, int Cout, int pool, float alpha, hipStream_t s);

// conv_halo.hip
int conv3x3_halo_bf16_supported(int N, int H, int W, int Cin, int Cout, int pool);     // 0 no, 1 yes, >= 2: yes with that many K slices
size_t conv3x3_halo_bf16_ws_bytes(int N, int H, int W, int Cin, int Cout, int pool);   // workspace of a K-sliced layer (else 0)
int conv3x3_halo_bf16(const void *x, const void *w, const float *scale, const float *shift, void *y, int N, int H, int W,
                      int Cin, int Cout, int pool, float alpha, float *ws, hipStream_t s);

}  // namespace ty1

#define TY1_REQUIRE(cond, ...)                       \
    do {                                             \
        if (!(cond)) return ty1::fail(__VA_ARGS__);  \
    } while (0)
