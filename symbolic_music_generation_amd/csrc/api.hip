// ABI version + error strings.
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
