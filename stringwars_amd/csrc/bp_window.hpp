// bp_window.hpp -- clamped tape windows shared by the bit-parallel kernels (bitparallel.hip, prepass.hip).
#pragma once
#include "common.hpp"

namespace swh {

__device__ __forceinline__ int bp_med3i(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// Window of a tape around one string: dword reads are clamped into [lo, hi] (indices relative to
// `base`), so nothing outside the tape's data buffer is ever touched.
struct ByteWindow {
    const uint8_t *base;
    int lo, hi, avail;
    bool tiny;
    __device__ __forceinline__ void init(const uint8_t *data, uint64_t start, uint64_t total) {
        auto c31 = [](int64_t v) { return (int)(v < -0x40000000ll ? -0x40000000ll : (v > 0x40000000ll ? 0x40000000ll : v)); };
        base = data + start;
        lo = c31(-(int64_t)start);
        hi = c31((int64_t)total - (int64_t)start - 4);
        avail = c31((int64_t)total - (int64_t)start);
        tiny = total < 4;
    }
    // bytes idx..idx+3 (little-endian); bytes outside the tape come back as garbage
    __device__ __forceinline__ uint32_t fetch4(int idx) const {
        if (!tiny) {
            int c = bp_med3i(idx, lo, hi);
            uint32_t dw;
            __builtin_memcpy(&dw, base + c, 4);
            int d = bp_med3i(idx - c, -3, 3);
            return d >= 0 ? dw >> (8 * d) : dw << (-8 * d);
        }
        uint32_t dw = 0;
        for (int u = 0; u < 4; ++u) {
            int pos = idx + u;
            if (pos >= lo && pos < avail) dw |= (uint32_t)base[pos] << (8 * u);
        }
        return dw;
    }
};


}  // namespace swh
