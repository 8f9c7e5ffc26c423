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

// crc32c (Castagnoli) of a host buffer, slicing-by-8: the per-tensor / per-block checksums of the -m checkpoint
// (tensor bundle: crc32c::Mask(crc32c::Value(bytes))) are verified on read; a byte loop in Python would take minutes on
// the 228 MB of AlexNet weights.
extern "C" uint32_t svx_crc32c(const void* data, size_t n)
{
    struct Tables {
        uint32_t t[8][256];
        Tables()
        {
            for (uint32_t i = 0; i < 256; ++i) {
                uint32_t c = i;
                for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
                t[0][i] = c;
            }
            for (uint32_t i = 0; i < 256; ++i)
                for (int k = 1; k < 8; ++k) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xFF];
        }
    };
    static const Tables tables;                   // function-local static: initialised once, thread-safe (C++11)
    const uint32_t (*tbl)[256] = tables.t;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t lo, hi;
        __builtin_memcpy(&lo, p, 4);
        __builtin_memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = tbl[7][lo & 0xFF] ^ tbl[6][(lo >> 8) & 0xFF] ^ tbl[5][(lo >> 16) & 0xFF] ^ tbl[4][lo >> 24]
          ^ tbl[3][hi & 0xFF] ^ tbl[2][(hi >> 8) & 0xFF] ^ tbl[1][(hi >> 16) & 0xFF] ^ tbl[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = tbl[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
