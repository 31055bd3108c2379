// gbnf_internal.h -- shared between the translation units of libgbnf_hip.so (not part of the C ABI).
#pragma once

namespace gbnf {

// Records the message behind gbnf_last_error() (thread-local) and returns `code`.
int fail(int code, const char* fmt, ...);

}  // namespace gbnf
