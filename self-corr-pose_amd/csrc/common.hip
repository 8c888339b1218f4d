// self-corr-pose_amd/csrc/common.hip -- ABI version + last-error string of libscp_hip.so.
#include <cstdio>

#include "scp_common.h"
#include "scp_hip.h"

namespace {
thread_local char g_err[512] = "";
}

namespace scp {
int fail(int code, const char* what) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(static_cast<hipError_t>(code)));
    return code;
}
int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    return fail(static_cast<int>(e), what);
}
}  // namespace scp

extern "C" int scp_abi_version(void) { return SCP_ABI_VERSION; }
extern "C" const char* scp_last_error(void) { return g_err; }

extern "C" int scp_stream_create(void** stream) {
    if (!stream) return scp::fail(hipErrorInvalidValue, "scp_stream_create: NULL out pointer");
    hipStream_t s = nullptr;
    const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) return scp::fail(static_cast<int>(e), "scp_stream_create");
    *stream = static_cast<void*>(s);
    return 0;
}
extern "C" int scp_stream_destroy(void* stream) {
    const hipError_t e = hipStreamDestroy(static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : scp::fail(static_cast<int>(e), "scp_stream_destroy");
}
