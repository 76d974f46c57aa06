// Shared helpers for the gfx950 kernels behind include/nlsh_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nlsh_hip.h"

namespace nlsh {

void set_error(const char *fmt, ...);

#define NLSH_CHECK_HIP(expr)                                                                          \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            nlsh::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return NLSH_E_HIP;                                                                        \
        }                                                                                             \
    } while (0)

#define NLSH_REQUIRE(cond, code, ...)   \
    do {                                \
        if (!(cond)) {                  \
            nlsh::set_error(__VA_ARGS__); \
            return (code);              \
        }                               \
    } while (0)

constexpr int WAVE = 64;

// fp32 -> uint32 whose unsigned order equals the float order (-inf < ... < -0 < +0 < ... < +inf < NaN).
__host__ __device__ inline uint32_t mono_from_float(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float float_from_mono(uint32_t m) {
    uint32_t u = (m & 0x80000000u) ? (m & 0x7FFFFFFFu) : ~m;
    return __builtin_bit_cast(float, u);
}
// sort key of a candidate: (distance, global row id) ascending.
__host__ __device__ inline uint64_t make_key(float dist, int32_t id) {
    return ((uint64_t)mono_from_float(dist) << 32) | (uint32_t)id;
}
constexpr uint64_t KEY_NONE = ~0ull;

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace nlsh
