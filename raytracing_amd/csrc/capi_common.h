// Shared by the two halves of the C ABI (host helpers / device objects).
#pragma once
#include <string>
#include "../../include/raytracing_amd.h"

namespace rta {
// Stores msg as the calling thread's last error and returns code.
int fail(int code, const std::string &msg);
}  // namespace rta
