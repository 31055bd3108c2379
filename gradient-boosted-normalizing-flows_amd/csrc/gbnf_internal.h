// gbnf_internal.h -- shared between the translation units of libgbnf_hip.so (not part of the C ABI).
#pragma once

namespace gbnf {

// Records the message behind gbnf_last_error() (thread-local) and returns `code`.
int fail(int code, const char* fmt, ...);

// Kernel-variant key only (not a descriptor value): the activation differs between the steps / nets of a component
// (`--coupling_network random` in the reference); the kernel reads it per step and net from the step header.
constexpr int GBNF_ACT_PER_STEP = 3;

}  // namespace gbnf
