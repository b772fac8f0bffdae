// libty1.so: error state, BN folding (host), per-kernel event profiling.
#include "common.h"

#include <atomic>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

namespace ty1 {

long long *g_stamp = nullptr;
long long g_stamp_cap = 0;

std::string &last_error() {
    static thread_local std::string e;
    return e;
}

int fail(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    last_error() = buf;
    return 1;
}

int num_cus() {
    static std::atomic<int> cache[64];
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess) return cus;
    if (dev >= 0 && dev < 64 && cache[dev].load(std::memory_order_relaxed) > 0) return cache[dev].load(std::memory_order_relaxed);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    if (dev >= 0 && dev < 64) cache[dev].store(cus, std::memory_order_relaxed);
    return cus;
}

static std::atomic<int> g_reserved_cus{0};

int grid_cus() {
    const int n = num_cus() - g_reserved_cus.load(std::memory_order_relaxed);
    return n < 8 ? (num_cus() < 8 ? num_cus() : 8) : n;
}

struct ProfEntry {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    int launches = 0;
    double ms = 0.0, flops = 0.0, bytes = 0.0;
};
static thread_local bool g_prof_on = false;
static thread_local std::string g_prof_filter;              // non-empty: only launches whose name starts with it
static thread_local int g_prof_stride = 1, g_prof_seq = 0;     // bracket every g_prof_stride-th launch that passes the filter
static thread_local std::map<std::string, ProfEntry> g_prof;
static thread_local std::vector<hipEvent_t> g_event_pool;   // events are reused: creating two per launch is slow

static hipEvent_t event_get() {
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}

ProfScope::ProfScope(const char *n, hipStream_t s, double fl, double by)
    : name(n), stream(s), flops(fl), bytes(by), on(g_prof_on) {
    if (!on) return;
    if (!g_prof_filter.empty() && name.compare(0, g_prof_filter.size(), g_prof_filter) != 0) {
        on = false;
        return;
    }
    if (g_prof_stride > 1 && (g_prof_seq++ % g_prof_stride) != 0) {
        on = false;
        return;
    }
    e0 = event_get();
    e1 = event_get();
    if (!e0 || !e1) {
        on = false;
        return;
    }
    (void)hipEventRecord(e0, stream);
}

ProfScope::~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e1, stream);
    ProfEntry &e = g_prof[name];
    e.pending.emplace_back(e0, e1);
    e.flops += flops;
    e.bytes += bytes;
}

static void prof_collect() {
    for (auto &kv : g_prof) {
        for (auto &p : kv.second.pending) {
            float ms = 0.f;
            if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
                kv.second.ms += ms;
                kv.second.launches += 1;
            }
            g_event_pool.push_back(p.first);
            g_event_pool.push_back(p.second);
        }
        kv.second.pending.clear();
    }
}

}  // namespace ty1

extern "C" {

int ty1_version(void) { return 1; }

const char *ty1_last_error(void) { return ty1::last_error().c_str(); }

int ty1_fold_bn(const float *gamma, const float *beta, const float *mean, const float *var, float eps, int C,
                float *scale, float *shift) {
    TY1_REQUIRE(gamma && beta && mean && var && scale && shift && C > 0, "ty1_fold_bn: bad arguments");
    for (int c = 0; c < C; ++c) {
        float s = gamma[c] / std::sqrt(var[c] + eps);
        scale[c] = s;
        shift[c] = beta[c] - mean[c] * s;
    }
    return 0;
}

int ty1_set_reserved_cus(int k) {
    TY1_REQUIRE(k >= 0, "ty1_set_reserved_cus: k=%d must be >= 0", k);
    ty1::g_reserved_cus.store(k, std::memory_order_relaxed);
    return 0;
}

int ty1_get_reserved_cus(int *k) {
    TY1_REQUIRE(k, "ty1_get_reserved_cus: null result pointer");
    *k = ty1::g_reserved_cus.load(std::memory_order_relaxed);
    return 0;
}

int ty1_prof_enable(int on) {
    ty1::g_prof_on = on != 0;
    return 0;
}

int ty1_prof_filter(const char *prefix) {
    ty1::g_prof_filter = prefix ? prefix : "";
    return 0;
}

int ty1_prof_stride(int n) {
    ty1::g_prof_stride = n > 1 ? n : 1;
    ty1::g_prof_seq = 0;
    return 0;
}

int ty1_prof_reset(void) {
    ty1::prof_collect();
    ty1::g_prof.clear();
    return 0;
}

int ty1_prof_query(const char *prefix, int *launches, double *total_ms) {
    TY1_REQUIRE(prefix && launches && total_ms, "ty1_prof_query: null argument");
    ty1::prof_collect();
    int n = 0;
    double ms = 0;
    size_t len = strlen(prefix);
    for (auto &kv : ty1::g_prof)
        if (kv.first.compare(0, len, prefix) == 0) {
            n += kv.second.launches;
            ms += kv.second.ms;
        }
    *launches = n;
    *total_ms = ms;
    return 0;
}

int ty1_prof_dump(char *buf, size_t cap) {
    TY1_REQUIRE(buf && cap > 0, "ty1_prof_dump: bad buffer");
    ty1::prof_collect();
    size_t off = 0;
    buf[0] = 0;
    for (auto &kv : ty1::g_prof) {
        int w = snprintf(buf + off, cap - off, "%s %d %.6f %.6e %.6e\n", kv.first.c_str(), kv.second.launches,
                         kv.second.ms, kv.second.flops, kv.second.bytes);
        if (w < 0 || (size_t)w >= cap - off) break;
        off += (size_t)w;
    }
    return 0;
}

}  // extern "C"
