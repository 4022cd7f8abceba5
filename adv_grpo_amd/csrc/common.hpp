// common.hpp -- shared device helpers for the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>

#include "../../include/advgrpo.h"

namespace advgrpo {

// ---- error reporting (thread-local message, negative return codes)
void set_error(const char* fmt, ...);
#define ADVGRPO_CHECK(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            advgrpo::set_error(__VA_ARGS__); \
            return -1;                      \
        }                                   \
    } while (0)
#define ADVGRPO_LAUNCH_CHECK()                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            advgrpo::set_error("%s:%d launch failed: %s", __FILE__, __LINE__,         \
                               hipGetErrorString(e__));                               \
            return -2;                                                                \
        }                                                                             \
    } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved), bit-identical to torch's cast
__host__ __device__ inline float bf2f(uint16_t h) {
    union { uint32_t u; float f; } c; c.u = (uint32_t)h << 16; return c.f;
}
__host__ __device__ inline uint16_t f2bf(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    // gfx950 converts in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even): one instruction per PAIR of values
    // instead of ~7 integer operations each; identical results for every non-NaN input
    return __builtin_bit_cast(uint16_t, (__bf16)f);
#endif
    union { uint32_t u; float f; } c; c.f = f;
    uint32_t u = c.u;
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ inline float round_bf16(float f) { return bf2f(f2bf(f)); }

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;

// ---- wave / block reductions (64-lane wave)
// All-reduce on the VALU: quad permutes and row mirrors (DPP) inside each 16-lane row, then the gfx950 row / half-wave
// swaps -- no LDS round trips (the __shfl_xor butterfly is six ds_bpermute, each ~100 cycles of latency).
template <class Op>
__device__ __forceinline__ float wave_allreduce(float v, Op op) {
    auto dpp = [](float x, auto ctrl) __attribute__((always_inline)) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value,
                                                                     0xf, 0xf, true));
    };
    v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));    // quad_perm [1,0,3,2]: lane ^ 1
    v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));    // quad_perm [2,3,0,1]: lane ^ 2
    v = op(v, dpp(v, std::integral_constant<int, 0x141>{}));   // row_half_mirror: the other quad of the 8-lane half
    v = op(v, dpp(v, std::integral_constant<int, 0x140>{}));   // row_mirror: the other half of the 16-lane row
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // rows 0<->1, 2<->3
    v = op(a, b);
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));   // halves of the wave
    return op(a, b);
}
// sum over each aligned group of 8 lanes, result in all 8 (same partners and order as three __shfl_xor 1 / 2 / 4 steps, on the
// VALU's DPP path instead of three dependent ds_bpermute round trips)
__device__ __forceinline__ float group8_sum(float v) {
    auto dpp = [](float x, auto ctrl) __attribute__((always_inline)) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value,
                                                                     0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});    // lane ^ 1
    v += dpp(v, std::integral_constant<int, 0x4E>{});    // lane ^ 2
    v += dpp(v, std::integral_constant<int, 0x141>{});   // the other quad of the 8-lane half (every lane of a quad holds its sum)
    return v;
}
__device__ inline float wave_sum(float v) {
    return wave_allreduce(v, [](float x, float y) { return x + y; });
}
__device__ inline float wave_max(float v) {
    return wave_allreduce(v, [](float x, float y) { return fmaxf(x, y); });
}
// sum over a block of NW waves; result valid in every thread. smem: >= NW floats.
template <int NW>
__device__ inline float block_sum(float v, float* smem) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[w] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) r += smem[i];
    return r;
}

// bijective XCD remap: consecutive remapped ids live on one XCD (observed placement b % 8)
__device__ inline int xcd_remap(int bid, int nwg) {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
// 1-D grid of nx * H * B workgroups -> (x, h, b) with x fastest in the REMAPPED id: all nx blocks of one (b, h), and runs
// of consecutive heads, execute on one XCD and share that head's K / V (or Q / dO) in the XCD's private 4 MB L2.
// With a plain 3-D grid the hardware deals consecutive x round-robin over the 8 XCDs and every L2 fetches the head itself.
__device__ __forceinline__ void xcd_local_bh(int nx, int H, int nwg, bool local, int& x, int& h, int& b) {
    const int id = local ? xcd_remap(blockIdx.x, nwg) : (int)blockIdx.x;
    x = id % nx;
    const int bh = id / nx;
    h = bh % H;
    b = bh / H;
}

inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
inline int dtype_size(int dt) { return dt == ADVGRPO_BF16 ? 2 : (dt == ADVGRPO_F64 ? 8 : 4); }

// ---- Philox4x32-10
struct Philox {
    uint32_t k0, k1;
    __device__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
    __device__ inline void operator()(uint64_t ctr, uint32_t out[4]) const {
        uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0u, c3 = 0u;
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
            const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
            const uint32_t n0 = hi1 ^ c1 ^ a, n2 = hi0 ^ c3 ^ b;
            c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
            a += 0x9E3779B9u; b += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }
};
// four standard normals from one Philox block (Box-Muller)
__device__ inline void philox_normal4(const Philox& ph, uint64_t ctr, float z[4]) {
    uint32_t r[4];
    ph(ctr, r);
    const float k = 2.3283064365386963e-10f;  // 2^-32
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float u1 = ((float)r[2 * p] + 0.5f) * k;
        const float u2 = ((float)r[2 * p + 1] + 0.5f) * k;
        const float rad = sqrtf(-2.0f * logf(fminf(fmaxf(u1, 1e-30f), 1.0f)));
        float s, c;
        sincosf(6.283185307179586f * u2, &s, &c);
        z[2 * p] = rad * c;
        z[2 * p + 1] = rad * s;
    }
}

}  // namespace advgrpo
