// test_hooks.h -- switches read from the environment by tests and A/B tools (queue capacities that force overflow paths, issue orders, builder limits, timing
// print-outs).  They exist only in a build with -DINCR_TEST_HOOKS: the Makefile's default build (what __graft_entry__.build() and the test suite use) sets it,
// `make RELEASE=1` does not -- a release library reads nothing but INCR_ROCTX / ROCP_TOOL_LIBRARIES (the profiler-range switch, api.hip) from its environment.
#pragma once
#include <cstdlib>
namespace incr {
#ifdef INCR_TEST_HOOKS
inline const char* test_hook_env(const char* name) { return getenv(name); }
#else
inline const char* test_hook_env(const char*) { return nullptr; }
#endif
}
