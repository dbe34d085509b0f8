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
    // Split form of fetch4_wide for software pipelining: issue the load now, realign the word where it is consumed
    // (the realignment reads the loaded register, so doing it next to the load would expose the memory latency).
    __device__ __forceinline__ uint32_t fetch4_raw(int idx, int &shift) const {
        if (tiny) { shift = 24; return fetch4_tiny(idx); }
        int c = bp_med3i(idx, lo, hi);
        uint32_t dw;
        __builtin_memcpy(&dw, base + c, 4);
        shift = 8 * bp_med3i(idx - c, -3, 3) + 24;
        return dw;
    }
    static __device__ __forceinline__ uint32_t realign(uint32_t dw, int shift) {
        return (uint32_t)((((uint64_t)dw) << 24) >> (uint32_t)shift);
    }
    // same as fetch4 for tapes of at least 4 bytes: one unconditional (clamped) dword load, no branches
    __device__ __forceinline__ uint32_t fetch4_wide(int idx) const {
        int c = bp_med3i(idx, lo, hi);
        uint32_t dw;
        __builtin_memcpy(&dw, base + c, 4);
        // shift by d = idx - c bytes in [-3, 3] without a branch (a divergent one would put a wait after every load)
        int d = bp_med3i(idx - c, -3, 3);
        return (uint32_t)((((uint64_t)dw) << 24) >> (uint32_t)(8 * d + 24));
    }
    // Sixteen bytes idx..idx+15 with ONE (unaligned, clamped) 128-bit load: the texture-address unit handles a
    // wave's scattered loads one instruction x one cache line at a time, so a lane that wants 16 consecutive bytes
    // should ask for them once rather than as four dwords. Only for tapes of at least 16 bytes. Returns how far the
    // clamp moved the window (0 for every window that lies inside the tape); `fix16` repairs the rare moved ones.
    __device__ __forceinline__ int fetch16_raw(int idx, uint32_t (&out)[4]) const {
        int hi16 = avail - 16;
        int c = bp_med3i(idx, lo, hi16);
        uint4 v;
        __builtin_memcpy(&v, base + c, 16);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
        return idx - c;
    }
    __device__ __forceinline__ void fix16(int idx, int moved, uint32_t (&out)[4]) const {
        if (__builtin_expect(moved != 0, 0)) {
#pragma unroll
            for (int q = 0; q < 4; ++q) out[q] = fetch4_wide(idx + q * 4);
        }
    }
    // tapes shorter than one dword: byte reads
    __device__ __forceinline__ uint32_t fetch4_tiny(int idx) const {
        uint32_t dw = 0;
        for (int u = 0; u < 4; ++u) {
            int pos = idx + u;
            if (pos >= lo && pos < avail) dw |= (uint32_t)base[pos] << (8 * u);
        }
        return dw;
    }
    // bytes idx..idx+3 (little-endian); bytes outside the tape come back as garbage
    __device__ __forceinline__ uint32_t fetch4(int idx) const { return tiny ? fetch4_tiny(idx) : fetch4_wide(idx); }
};

// Byte alphabet, address arithmetic done by hand: the LDS byte address of EqLo[v][lane] is tbase + (v << 8) and of
// EqHi[v][lane] tbase + 4096 + (v << 8), where tbase (the wave's table + 4 * lane) has zeros in bits 8..11 -- the
// per-wave tables are 8 KB apart and start on an 8 KB boundary (dynamic LDS from 0, or an aligned static array). So a nibble that has been shifted
// to bits 8..11 goes in with ONE v_bitop3 ((x & 0xF00) | tbase; 2.8 cycles, against shift + and + add at 4.4 each
// the compiler emits for the indexed form), and the +4096 rides in the ds instruction's offset field.
typedef __attribute__((address_space(3))) uint32_t lds_u32;
struct NibbleTables {
    uint32_t tbase, mask;
    __device__ __forceinline__ void init(uint32_t *table, int lane) {
        tbase = (uint32_t)(uintptr_t)(lds_u32 *)(table + lane);
        if (tbase & 0xF00u) __builtin_trap();   // the layout assumption above: fail loudly, never score with wrong tables
        mask = 0xF00u;
        asm volatile("" : "+v"(mask));   // keep it in a register: three distinct registers is the fast form of v_bitop3
    }
    // x holds the symbol in byte U
    template <int U> __device__ __forceinline__ uint32_t lo_addr(uint32_t x) const {
        uint32_t s = x;
        if constexpr (U == 0) s = x << 8;
        else if constexpr (U > 1) s = x >> (8 * U - 8);
        return (uint32_t)__builtin_amdgcn_bitop3_b32((int)s, (int)mask, (int)tbase, 0xEA);
    }
    template <int U> __device__ __forceinline__ uint32_t hi_addr(uint32_t x) const {
        uint32_t s;
        if constexpr (U == 0) s = x << 4;
        else s = x >> (8 * U - 4);
        return (uint32_t)__builtin_amdgcn_bitop3_b32((int)s, (int)mask, (int)tbase, 0xEA);
    }
    template <int U> __device__ __forceinline__ uint32_t lookup(uint32_t x) const {
        return *(const lds_u32 *)(uintptr_t)lo_addr<U>(x) & *(const lds_u32 *)(uintptr_t)(hi_addr<U>(x) + 4096);
    }
    template <int U> __device__ __forceinline__ void insert(uint32_t x, uint32_t bit) const {
        __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)lo_addr<U>(x), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_or((lds_u32 *)(uintptr_t)(hi_addr<U>(x) + 4096), bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

}  // namespace swh
