// Internal helpers shared by the translation units of librt_amd.so (not part of the C ABI).
#ifndef RT_INTERNAL_H
#define RT_INTERNAL_H
#include <string>

// Record `msg` as this thread's last error and return `code`.
int rt_fail(int code, const std::string& msg);

#endif
