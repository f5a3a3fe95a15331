// Thread-local last-error text behind mbpe_last_error() (include/mbpe.h).
#include "mbpe.h"
#include "mbpe_host.h"

#include <string>

namespace {
thread_local std::string g_last_error;
}

namespace mbpe_host {
void set_last_error(const std::string &msg) { g_last_error = msg; }
const char *last_error() { return g_last_error.c_str(); }
}  // namespace mbpe_host

extern "C" {
const char *mbpe_last_error(void) { return mbpe_host::last_error(); }
// MBPE_SOURCE_HASH: sha256 prefix of the library's sources, passed by the build (__graft_entry__.build_lib), so that a
// binary can be told from a stale one: build() recompiles unless the hash in the .so matches the tree.
#ifndef MBPE_SOURCE_HASH
#define MBPE_SOURCE_HASH "unknown"
#endif
const char *mbpe_version(void) { return "mbpe-amd 0.3 (gfx950) mbpe-src:" MBPE_SOURCE_HASH; }
}
