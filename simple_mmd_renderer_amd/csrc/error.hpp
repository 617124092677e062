// error.hpp -- the per-thread "last error" text behind mmdx_last_error_string().
#pragma once

#include <string>

#include "../../include/mmdx.h"

namespace mmdx {

// Remember `msg` as the calling thread's last error and return `st` (so callers can `return fail(...)`).
mmdx_status fail(mmdx_status st, const std::string &msg);

}  // namespace mmdx
