// gbnf_internal.h -- shared between the translation units of libgbnf_hip.so (not part of the C ABI).
#pragma once

namespace gbnf {

// Records the message behind gbnf_last_error() (thread-local) and returns `code`.
int fail(int code, const char* fmt, ...);

// The CURRENT device's counter of waves that stored a split-f16 operand beyond +-65504 (it saturates there); null if it
// could not be allocated.  Call it at launch time, with the launch's device current.  Read and reset by
// gbnf_saturation_count().
unsigned* saturation_counter();

// Kernel-variant key only (not a descriptor value): the activation differs between the steps / nets of a component
// (`--coupling_network random` in the reference); the kernel reads it per step and net from the step header.
constexpr int GBNF_ACT_PER_STEP = 3;

}  // namespace gbnf
