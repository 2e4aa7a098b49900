// Thread-local last-error slot behind rt_last_error().  The reference reports failures by
// printing and exiting (CR/Renderer_CUDA.cpp:3-6, CR/RayTracer.cu:111-114); this library
// returns a code and keeps the message.
#include "rt_amd.h"
#include "rt_internal.h"

static thread_local std::string g_last_error;

int rt_fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

extern "C" const char* rt_last_error(void) { return g_last_error.c_str(); }

// Identity of this build: a hash of the library's sources and compiler flags, passed in by the build script.  Measurements
// taken with one build (profiler counters kept in files) can then be told apart from a library built from other sources.
#ifndef RT_BUILD_ID
#define RT_BUILD_ID "unidentified"
#endif
extern "C" const char* rt_build_id(void) { return RT_BUILD_ID; }
