// Error reporting and device probing for libsequoia_hip.
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/sequoia_hip.h"
#include "sq_common.h"

static thread_local char g_err[512] = "";

void sq_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* sq_last_error(void) { return g_err; }

extern "C" int sq_version(void) { return 100; }

extern "C" int sq_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return 0; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
