// TEST INFRASTRUCTURE ONLY — CPU restatement ("oracle") of the vszip hot path.
//
// Nothing under oracle/ is part of the shipped product. Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library, and only as the checker / the CPU baseline that is timed beside the
// GPU path. The product (vapoursynth-zip_amd/) never links or calls it.
//
// Every function cites the reference file:line (relative to the vszip v19.0.0
// tree) whose arithmetic it restates. The restatement is scalar C++ compiled
// with -ffp-contract=off so that f32 operation order is exactly the order
// written here; fused multiply-adds appear only where the reference writes
// @mulAdd, via fmaf().
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

// dtype codes shared with include/vszip_hip.h
enum { VSZO_U8 = 0, VSZO_U16 = 1, VSZO_F16 = 2, VSZO_F32 = 3, VSZO_U32 = 4 /* PlaneAverage only, helper.zig:78 */ };

// IEEE binary16 storage type with explicit conversions (g++ 11 has no
// _Float16 on x86). f32 -> f16 is round-to-nearest-even like Zig's @floatCast.
struct half_t {
    uint16_t bits;
};

static inline float half_to_float(half_t h) {
    const uint32_t s = (uint32_t)(h.bits & 0x8000u) << 16;
    uint32_t e = (h.bits >> 10) & 0x1Fu;
    uint32_t m = h.bits & 0x3FFu;
    uint32_t out;
    if (e == 0) {
        if (m == 0) {
            out = s;
        } else {
            // subnormal: normalise
            int sh = 0;
            while (!(m & 0x400u)) {
                m <<= 1;
                ++sh;
            }
            m &= 0x3FFu;
            out = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
        }
    } else if (e == 31) {
        out = s | 0x7F800000u | (m << 13);
    } else {
        out = s | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    std::memcpy(&f, &out, 4);
    return f;
}

static inline half_t float_to_half(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t s = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    half_t h;
    if (x >= 0x7F800000u) {  // inf / nan
        h.bits = (uint16_t)(s | 0x7C00u | ((x > 0x7F800000u) ? 0x200u : 0u));
        return h;
    }
    if (x >= 0x477FF000u) {  // rounds to >= 65520 -> inf
        h.bits = (uint16_t)(s | 0x7C00u);
        return h;
    }
    if (x < 0x38800000u) {  // subnormal or zero in f16
        if (x < 0x33000000u) {  // < 2^-25 -> 0
            h.bits = (uint16_t)s;
            return h;
        }
        const int e = (int)(x >> 23);
        uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
        const int shift = 126 - e;  // 14..24
        const uint32_t lsb = 1u << shift;
        const uint32_t rnd = (lsb >> 1) - 1 + ((m >> shift) & 1u);
        m = (m + rnd) >> shift;
        h.bits = (uint16_t)(s | m);
        return h;
    }
    const uint32_t rnd = 0xFFFu + ((x >> 13) & 1u);
    x += rnd;
    h.bits = (uint16_t)(s | ((x - 0x38000000u) >> 13));
    return h;
}

template <typename T>
struct px_traits;
template <>
struct px_traits<uint8_t> {
    static constexpr bool is_int = true;
    static inline float to_f32(uint8_t v) { return (float)v; }
};
template <>
struct px_traits<uint16_t> {
    static constexpr bool is_int = true;
    static inline float to_f32(uint16_t v) { return (float)v; }
};
template <>
struct px_traits<uint32_t> {
    static constexpr bool is_int = true;
    static inline float to_f32(uint32_t v) { return (float)v; }
};
template <>
struct px_traits<float> {
    static constexpr bool is_int = false;
    static inline float to_f32(float v) { return v; }
    static inline float from_f32(float v) { return v; }
};
template <>
struct px_traits<half_t> {
    static constexpr bool is_int = false;
    static inline float to_f32(half_t v) { return half_to_float(v); }
    static inline half_t from_f32(float v) { return float_to_half(v); }
};

#define VSZO_API extern "C" __attribute__((visibility("default")))
