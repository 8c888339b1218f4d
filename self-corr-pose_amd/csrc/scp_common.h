// self-corr-pose_amd/csrc/scp_common.h -- error plumbing shared by the kernels' host launchers.
#pragma once
#include <hip/hip_runtime.h>

namespace scp {
// records `what` (+ the HIP error text) for scp_last_error() and returns `code`
int fail(int code, const char* what);
// hipGetLastError() after a launch; 0 when clean
int check_launch(const char* what);
}  // namespace scp
