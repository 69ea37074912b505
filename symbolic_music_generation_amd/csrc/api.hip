// ABI version, error strings, and the per-kernel timing hook.
#include <mutex>
#include <vector>
#include "common.h"
#include "musicxl_internal.h"

extern "C" int mxl_abi_version(void) { return MXL_ABI_VERSION; }

extern "C" const char* mxl_error_string(int code) {
    if (code == MXL_OK) return "ok";
    if (code == MXL_EINVAL) return "libmusicxl: invalid argument (shape / alignment / null pointer)";
    if (code == MXL_EUNSUPPORTED) return "libmusicxl: unsupported configuration";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "libmusicxl: unknown error";
}

// ---- mxl_ktime_*: hipEvent pairs around individual kernels of the attention path (bench.py's per-kernel roofline)
namespace mxl_kt {
int g_on = 0;
namespace {
struct Rec { int id; hipEvent_t a, b; };
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
std::vector<Rec> g_open;      // begun, not yet ended (one per id at most; launches do not nest)
hipEvent_t take() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace
void begin(int id, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r{id, take(), take()};
    (void)hipEventRecord(r.a, s);
    g_open.push_back(r);
}
void end(int id, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = g_open.size(); i-- > 0;) {
        if (g_open[i].id != id) continue;
        Rec r = g_open[i];
        g_open.erase(g_open.begin() + i);
        (void)hipEventRecord(r.b, s);
        g_recs.push_back(r);
        return;
    }
}
}  // namespace mxl_kt

extern "C" int mxl_ktime_enable(int on) {
    mxl_kt::g_on = on ? 1 : 0;
    return MXL_OK;
}

extern "C" int mxl_ktime_collect(float* ms_sum, int* launches, int n) {
    MXL_CHECK_ARG(ms_sum && launches && n >= MXL_KT_COUNT);
    std::lock_guard<std::mutex> lk(mxl_kt::g_mu);
    for (int i = 0; i < n; i++) { ms_sum[i] = 0.f; launches[i] = 0; }
    for (auto& r : mxl_kt::g_recs) {
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e != hipSuccess) return (int)e;
        if (r.id >= 0 && r.id < n) { ms_sum[r.id] += ms; launches[r.id] += 1; }
        mxl_kt::g_pool.push_back(r.a);
        mxl_kt::g_pool.push_back(r.b);
    }
    mxl_kt::g_recs.clear();
    return MXL_OK;
}
