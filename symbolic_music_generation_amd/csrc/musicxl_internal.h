// Internal glue: pulls in the public C ABI so kernels and the header cannot drift apart.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/musicxl.h"

// per-kernel event brackets of mxl_ktime_enable (api.hip)
namespace mxl_kt {
extern int g_on;
void begin(int id, hipStream_t s);
void end(int id, hipStream_t s);
struct Scope {
    int id; hipStream_t s; bool on;
    Scope(int id_, hipStream_t s_) : id(id_), s(s_), on(g_on != 0) { if (on) begin(id, s); }
    ~Scope() { if (on) end(id, s); }
};
}  // namespace mxl_kt
