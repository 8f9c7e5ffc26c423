// svx_api.hip -- version / error strings of the libsvx.so C ABI (include/svx.h).
#include "../../include/svx.h"

extern "C" int svx_version(void) { return SVX_VERSION; }

extern "C" const char* svx_strerror(int code)
{
    switch (code) {
    case SVX_OK:        return "ok";
    case SVX_EINVAL:    return "invalid argument";
    case SVX_ECAPACITY: return "output capacity too small";
    case SVX_ELAUNCH:   return "HIP launch error";
    default:            return "unknown svx error";
    }
}
