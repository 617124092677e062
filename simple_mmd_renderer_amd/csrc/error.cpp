// error.cpp -- see error.hpp.  No exceptions cross the C ABI: every entry point returns a status
// code and leaves its explanation here (the reference's loaders throw mmd::exception instead,
// L/util/dwarf.inl:46-59; its Deform has no error path at all).
#include "error.hpp"

namespace {
thread_local std::string g_last_error;
}

namespace mmdx {

mmdx_status fail(mmdx_status st, const std::string &msg) {
    g_last_error = msg;
    return st;
}

}  // namespace mmdx

extern "C" const char *mmdx_last_error_string(void) { return g_last_error.c_str(); }
