// gemm_device.hpp -- device pieces shared by the bf16 GEMM kernel files (gemm.hip, gemm8p.hip): activations, the
// XCD-grouped tile order and the fused epilogues.  See gemm.hip for the kernel family's description.
#pragma once
#include <type_traits>
#include <utility>

#include "gemm.hpp"

namespace advgrpo {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

// gelu_tanh(x) = 0.5 x (1 + tanh u), u = sqrt(2/pi) (x + 0.044715 x^3)  ==  x * sigmoid(2u)  ==  x / (1 + 2^(x (C1 + C2 x^2)))
// with C1 = -2 sqrt(2/pi) log2(e), C2 = 0.044715 C1: one v_exp + one v_rcp, and five other operations written so that the
// scalar form below and the two-column form of the eight-phase epilogue (gelu_tanh_pk: v_pk_mul / v_pk_fma / v_pk_add on
// register pairs) execute the SAME IEEE operations per element -- the tile variants stay bit-identical.
#define ADVGRPO_GELU_C1 (-2.3022081981f)
#define ADVGRPO_GELU_C2 (-0.10294324f)
__device__ __forceinline__ float gelu_tanh_f32(float x) {
#pragma clang fp contract(off)
    float t = x * x;
    t = __builtin_fmaf(t, ADVGRPO_GELU_C2, ADVGRPO_GELU_C1);
    t = t * x;
    const float d = __builtin_amdgcn_exp2f(t) + 1.0f;
    return x * __builtin_amdgcn_rcpf(d);
}
typedef float gelu_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gelu_f32x2 gelu_tanh_pk(gelu_f32x2 x) {
#pragma clang fp contract(off)
    gelu_f32x2 t = x * x;
    t = __builtin_elementwise_fma(t, gelu_f32x2{ADVGRPO_GELU_C2, ADVGRPO_GELU_C2}, gelu_f32x2{ADVGRPO_GELU_C1, ADVGRPO_GELU_C1});
    t = t * x;
    gelu_f32x2 d = gelu_f32x2{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
    return x * gelu_f32x2{__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}

// d/du gelu_tanh(u), from the same sigmoid r = 1 / (1 + 2^(u (C1 + C2 u^2))) = (1 + tanh z) / 2 the forward uses:
//   gelu' = (1 + th) / 2 + u (1 - th^2) / 2 * z'   with th = 2 r - 1, 1 - th^2 = 4 r (1 - r), z' = c (1 + 3 a u^2)
//         = r (1 + (1 - r) u (2c + 6ac u^2))
// one v_exp + one v_rcp + 8 full-rate operations (libm's tanhf is ~40 instructions and dominated the dX = (dY W) * gelu'
// epilogue of the G-step).  Scalar and packed forms execute the same operations per element.
#define ADVGRPO_DGELU_K1 1.5957691216057308f     /* 2 sqrt(2/pi) */
#define ADVGRPO_DGELU_K2 0.21406444881780073f    /* 6 * 0.044715 * sqrt(2/pi) */
__device__ __forceinline__ float dgelu_tanh_f32(float u) {
#pragma clang fp contract(off)
    const float u2 = u * u;
    float t = __builtin_fmaf(u2, ADVGRPO_GELU_C2, ADVGRPO_GELU_C1);
    t = t * u;
    const float r = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(t) + 1.0f);
    float q = __builtin_fmaf(u2, ADVGRPO_DGELU_K2, ADVGRPO_DGELU_K1);
    q = q * u;
    q = q * (1.0f - r);
    return __builtin_fmaf(q, r, r);
}
// y[k] *= gelu'(z[k]) on four column pairs, stage by stage (see gelu_tanh_pk4)
__device__ __forceinline__ void dgelu_tanh_mul_pk4(gelu_f32x2 (&y)[4], const gelu_f32x2 (&z)[4]) {
#pragma clang fp contract(off)
    gelu_f32x2 u2[4], t[4], q[4];
    const gelu_f32x2 c1{ADVGRPO_GELU_C1, ADVGRPO_GELU_C1}, c2{ADVGRPO_GELU_C2, ADVGRPO_GELU_C2};
    const gelu_f32x2 k1{ADVGRPO_DGELU_K1, ADVGRPO_DGELU_K1}, k2{ADVGRPO_DGELU_K2, ADVGRPO_DGELU_K2};
#pragma unroll
    for (int k = 0; k < 4; ++k) u2[k] = z[k] * z[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = __builtin_elementwise_fma(u2[k], c2, c1);
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = t[k] * z[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = gelu_f32x2{__builtin_amdgcn_exp2f(t[k].x), __builtin_amdgcn_exp2f(t[k].y)};
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = t[k] + 1.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = gelu_f32x2{__builtin_amdgcn_rcpf(t[k].x), __builtin_amdgcn_rcpf(t[k].y)};   // r
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = __builtin_elementwise_fma(u2[k], k2, k1);
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = q[k] * z[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = q[k] * (1.0f - t[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = __builtin_elementwise_fma(q[k], t[k], t[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = y[k] * q[k];
}

// four column pairs at once, stage by stage: the same operations per element as gelu_tanh_pk, but every instruction's
// operands were produced four instructions earlier.  Chain by chain (one pair after the other through one temporary) the
// compiler had to put an s_nop between every two dependent packed operations: 333 of the 2311 instructions of the
// bias + GELU epilogue of a tile were s_nop.
__device__ __forceinline__ void gelu_tanh_pk4(gelu_f32x2 (&v)[4]) {
#pragma clang fp contract(off)
    gelu_f32x2 t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = v[k] * v[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = __builtin_elementwise_fma(t[k], gelu_f32x2{ADVGRPO_GELU_C2, ADVGRPO_GELU_C2}, gelu_f32x2{ADVGRPO_GELU_C1, ADVGRPO_GELU_C1});
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = t[k] * v[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = gelu_f32x2{__builtin_amdgcn_exp2f(t[k].x), __builtin_amdgcn_exp2f(t[k].y)};
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = t[k] + 1.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = gelu_f32x2{__builtin_amdgcn_rcpf(t[k].x), __builtin_amdgcn_rcpf(t[k].y)};
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = v[k] * t[k];
}

__device__ inline float act_fn(float x, int act) {
    switch (act) {
        case ACT_GELU_TANH: return gelu_tanh_f32(x);
        case ACT_GELU_ERF: return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
        case ACT_SILU: return x / (1.0f + __expf(-x));
        case ACT_QUICK_GELU: return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
        default: return x;
    }
}
// derivative of the activation at pre-activation u
__device__ inline float dact_fn(float u, int act) {
    if (act == ACT_MUL_AUX) return u;
    if (act == ACT_DGELU_TANH) return dgelu_tanh_f32(u);
    // exact GELU: Phi(u) + u phi(u)
    return 0.5f * (1.0f + erff(u * 0.7071067811865476f)) + u * 0.3989422804014327f * __expf(-0.5f * u * u);
}

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(<N-1>)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}


// Grouped tile order inside an XCD's contiguous id range: ids walk down group_m (4, measured best of 1..16) row-tiles before moving to the next
// column-tile, so the ~32-64 tiles an XCD runs at once form a near-square patch of C and share their A / W panels
// in that XCD's 4 MB L2 (a row-major order gives a 1 x 48 strip for the wide QKV / FF1 outputs: every W panel of
// the layer streams through every XCD for each row of tiles).
__device__ __forceinline__ void tile_coords(int id, int tiles_m, int tiles_n, int group_m, int& tm, int& tn) {
    const int per_group = group_m * tiles_n;
    const int g = id / per_group, r = id - g * per_group;
    const int first_m = g * group_m;
    const int gsz = min(tiles_m - first_m, group_m);
    tn = r / gsz;
    tm = first_m + (r - tn * gsz);
}

// ---- fused epilogue shared by the kernel variants
// Row-coalesced path.  In the MFMA accumulator layout a lane owns 4 consecutive columns of ONE row and the 16
// lanes of a quarter-wave own 16 DIFFERENT rows, so storing straight from the accumulators issues 64 separate
// 8-byte writes per instruction (measured: 0.8 TB/s for the whole epilogue).  Instead each wave bounces one
// 16-row slab of its tile at a time through a private LDS scratch (f32, rows padded by 16 B) and comes back with
// 8 consecutive columns per lane and TN/8 consecutive lanes per row: every bias / gate / residual / aux read and
// every output store is a 16-byte access and a row's lanes cover whole 128-byte lines.
// The arithmetic and its order are the same as in the fragment-layout path below (bit-identical results).
template <int FM, int FN, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_rows(const GemmParams& p, f32x4 (&acc)[FM][FN], int m0, int n0, int wm,
                                                   int wn, int bz, int lane, char* scratch) {
    constexpr int RS = TN * 4 + 16;            // scratch row stride in bytes
    constexpr int LPR = TN / 8;                // lanes per row on the way out
    constexpr int RPP = 64 / LPR;              // rows per pass (a 48-wide wave tile uses 60 of the 64 lanes)
    constexpr int PASSES = (16 + RPP - 1) / RPP;
    static_assert(RPP >= 1 && LPR >= 1 && TN % 8 == 0, "wave tile too wide for the row epilogue");
    const int mrow = lane & 15, ncol = (lane >> 4) * 4;
    const int orow_l = lane / LPR, c8 = (lane % LPR) * 8;
    auto unpack8 = [](const uint4& q, float (&f)[8]) __attribute__((always_inline)) {
        f[0] = bf2f((bf16_t)(q.x & 0xffffu)); f[1] = bf2f((bf16_t)(q.x >> 16));
        f[2] = bf2f((bf16_t)(q.y & 0xffffu)); f[3] = bf2f((bf16_t)(q.y >> 16));
        f[4] = bf2f((bf16_t)(q.z & 0xffffu)); f[5] = bf2f((bf16_t)(q.z >> 16));
        f[6] = bf2f((bf16_t)(q.w & 0xffffu)); f[7] = bf2f((bf16_t)(q.w >> 16));
    };
    auto pack8 = [](const float (&v)[8]) __attribute__((always_inline)) {
        uint4 pk;
        pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        pk.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
        pk.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
        return pk;
    };
    const int n = n0 + wn * TN + c8;
    float bias8[8];
    if (p.bias && n < p.N) unpack8(*reinterpret_cast<const uint4*>(p.bias + n), bias8);
    // The residual rows of the whole wave tile are requested before the first slab is bounced: left inside the slab loop
    // each of these (L2 / HBM latency) loads was waited for on the spot, a dozen vmcnt(0) per tile.  (The gate vectors
    // stay in the loop: a few KB per launch, L1-resident; prefetching them too would spill at the 128-VGPR bound.)
    uint4 res_q[FM][PASSES];
    static_for<FM>([&](auto idx) {
        constexpr int i = decltype(idx)::value;
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int row = ps * RPP + orow_l;
            const int m = m0 + wm * TM + i * 16 + row;
            res_q[i][ps] = uint4{0u, 0u, 0u, 0u};
            if (!p.residual || (64 % LPR != 0 && lane >= LPR * RPP) || (16 % RPP != 0 && row >= 16) || m >= p.M || n >= p.N) continue;
            int64_t orow = m;
            if (p.seg_rows > 0) {
                const int bidx = m / p.seg_rows;
                orow = (int64_t)bidx * p.seg_stride + p.seg_off + (m - bidx * p.seg_rows);
            }
            res_q[i][ps] = *reinterpret_cast<const uint4*>(p.residual + (int64_t)bz * p.strideR + orow * p.ldr + n);
        }
    });
    static_for<FM>([&](auto idx) {
        constexpr int i = decltype(idx)::value;
#pragma unroll
        for (int j = 0; j < FN; ++j)
            *reinterpret_cast<f32x4*>(scratch + mrow * RS + (j * 16 + ncol) * 4) = acc[i][j];
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int row = ps * RPP + orow_l;
            if ((64 % LPR != 0 && lane >= LPR * RPP) || (16 % RPP != 0 && row >= 16)) continue;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(scratch + row * RS + c8 * 4);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(scratch + row * RS + c8 * 4 + 16);
            const int m = m0 + wm * TM + i * 16 + row;
            if (m >= p.M || n >= p.N) continue;
            int64_t orow = m;
            if (p.seg_rows > 0) {
                const int bidx = m / p.seg_rows;
                orow = (int64_t)bidx * p.seg_stride + p.seg_off + (m - bidx * p.seg_rows);
            }
            float v[8] = {lo[0] * p.alpha, lo[1] * p.alpha, lo[2] * p.alpha, lo[3] * p.alpha,
                          hi[0] * p.alpha, hi[1] * p.alpha, hi[2] * p.alpha, hi[3] * p.alpha};
            if (p.bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bias8[e];
            }
            if constexpr (TN == 64) {
                if (p.rms_w) {   // QK-norm: the wave tile's 64 columns are one head, its row sits in 8 adjacent lanes
                    const int hh = n >> 6;
                    float sq = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        v[e] = round_bf16(v[e]);          // the Linear's bf16 output is what gets normalised
                        sq += v[e] * v[e];
                    }
                    sq = group8_sum(sq);
                    if (hh < p.rms_nheads) {
                        const float rs = rsqrtf(sq * (1.0f / 64.0f) + p.rms_eps);
                        if (p.rms_rs_out && (lane & 7) == 0) p.rms_rs_out[orow * p.rms_nheads + hh] = rs;
                        float w8[8];
                        unpack8(*reinterpret_cast<const uint4*>(p.rms_w + (hh / p.rms_hpw) * 64 + c8), w8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = round_bf16(v[e] * rs) * w8[e];
                    }
                }
            }
            if (p.aux_out)
                *reinterpret_cast<uint4*>(p.aux_out + (int64_t)bz * p.strideC + orow * p.ld_aux + n) = pack8(v);
            if (p.act >= ACT_DGELU_TANH) {
                float z[8];
                unpack8(*reinterpret_cast<const uint4*>(p.aux_in + (int64_t)bz * p.strideC + orow * p.ld_aux + n), z);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dact_fn(z[e], p.act);
            } else if (p.act != ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = act_fn(v[e], p.act);
            }
            if (p.gate) {
                const int gb = p.gate_rows > 0 ? m / p.gate_rows : 0;
                float g[8];
                unpack8(*reinterpret_cast<const uint4*>(p.gate + (int64_t)bz * p.gate_batch_stride +
                                                        (int64_t)gb * p.gate_stride + n), g);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= g[e];
            }
            if (p.residual) {
                float r[8];
                unpack8(res_q[i][ps], r);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r[e];
            }
            const int64_t o = (int64_t)bz * p.strideC + orow * p.ldc + n;
            if (p.out_dtype == ADVGRPO_BF16) {
                *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + o) = pack8(v);
            } else {
                float* c = reinterpret_cast<float*>(p.C) + o;
                *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
    });
}

// true when every operand of the epilogue can be accessed as aligned 16-byte row segments
__device__ __forceinline__ bool epilogue_rows_ok(const GemmParams& p) {
    auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    return p.splitk == 1 && (p.N & 7) == 0 && ((p.ldc | p.ldr | p.gate_stride | p.ld_aux | p.strideC | p.strideR |
                                                p.gate_batch_stride) & 7) == 0 &&
           a16(p.C) && a16(p.bias) && a16(p.gate) && a16(p.residual) && a16(p.aux_out) && a16(p.aux_in);
}

template <int FM, int FN, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_frag(const GemmParams& p, f32x4 (&acc)[FM][FN], int m0, int n0, int wm,
                                                   int wn, int bz, int lane) {
    // lane holds C[m][n..n+3], m = frag row (lane&15), n = (lane>>4)*4.
    // Full, 8-byte-aligned quads take the vector path (bf16x4 loads of bias / gate / residual, one bf16x4 or
    // float4 store); the ragged N edge falls back to predicated scalars.  Fragments are visited with
    // compile-time indices (static_for) so the accumulators never leave the register file.
    const int mrow = lane & 15, ncol = (lane >> 4) * 4;
    const bool vec_ok = ((p.ldc | p.ldr | p.gate_stride) & 3) == 0 && (p.N & 3) == 0;
    static_for<FM * FN>([&](auto idx) {
        constexpr int i = decltype(idx)::value / FN, j = decltype(idx)::value % FN;
        const int m = m0 + wm * TM + i * 16 + mrow;
        const int n = n0 + wn * TN + j * 16 + ncol;
        if (m >= p.M || n >= p.N) return;
        int64_t orow = m;
        if (p.seg_rows > 0) {
            const int bidx = m / p.seg_rows;
            orow = (int64_t)bidx * p.seg_stride + p.seg_off + (m - bidx * p.seg_rows);
        }
        const int gb = p.gate_rows > 0 ? m / p.gate_rows : 0;
        const bf16_t* grow = p.gate ? p.gate + (int64_t)bz * p.gate_batch_stride + (int64_t)gb * p.gate_stride : nullptr;
        const bf16_t* rrow = p.residual ? p.residual + (int64_t)bz * p.strideR + orow * p.ldr : nullptr;
        const int64_t o = (int64_t)bz * p.strideC + orow * p.ldc + n;
        const f32x4 a4 = acc[i][j];
        float v0 = a4[0] * p.alpha, v1 = a4[1] * p.alpha, v2 = a4[2] * p.alpha, v3 = a4[3] * p.alpha;
        if (p.splitk > 1) {   // partial tile: f32 atomic accumulation (gradient buffers), no other epilogue
            float* c = reinterpret_cast<float*>(p.C) + o;
            unsafeAtomicAdd(c, v0);
            if (n + 1 < p.N) unsafeAtomicAdd(c + 1, v1);
            if (n + 2 < p.N) unsafeAtomicAdd(c + 2, v2);
            if (n + 3 < p.N) unsafeAtomicAdd(c + 3, v3);
            return;
        }
        if (vec_ok) {
            if (p.bias) {
                const uint2 q = *reinterpret_cast<const uint2*>(p.bias + n);
                v0 += bf2f((bf16_t)(q.x & 0xffffu)); v1 += bf2f((bf16_t)(q.x >> 16));
                v2 += bf2f((bf16_t)(q.y & 0xffffu)); v3 += bf2f((bf16_t)(q.y >> 16));
            }
            if (p.aux_out) {
                uint2 pk;
                pk.x = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
                pk.y = (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16);
                *reinterpret_cast<uint2*>(p.aux_out + (int64_t)bz * p.strideC + orow * p.ld_aux + n) = pk;
            }
            if (p.act >= ACT_DGELU_TANH) {
                const uint2 q = *reinterpret_cast<const uint2*>(p.aux_in + (int64_t)bz * p.strideC + orow * p.ld_aux + n);
                v0 *= dact_fn(bf2f((bf16_t)(q.x & 0xffffu)), p.act); v1 *= dact_fn(bf2f((bf16_t)(q.x >> 16)), p.act);
                v2 *= dact_fn(bf2f((bf16_t)(q.y & 0xffffu)), p.act); v3 *= dact_fn(bf2f((bf16_t)(q.y >> 16)), p.act);
            } else if (p.act != ACT_NONE) {
                v0 = act_fn(v0, p.act); v1 = act_fn(v1, p.act); v2 = act_fn(v2, p.act); v3 = act_fn(v3, p.act);
            }
            if (grow) {
                const uint2 q = *reinterpret_cast<const uint2*>(grow + n);
                v0 *= bf2f((bf16_t)(q.x & 0xffffu)); v1 *= bf2f((bf16_t)(q.x >> 16));
                v2 *= bf2f((bf16_t)(q.y & 0xffffu)); v3 *= bf2f((bf16_t)(q.y >> 16));
            }
            if (rrow) {
                const uint2 q = *reinterpret_cast<const uint2*>(rrow + n);
                v0 += bf2f((bf16_t)(q.x & 0xffffu)); v1 += bf2f((bf16_t)(q.x >> 16));
                v2 += bf2f((bf16_t)(q.y & 0xffffu)); v3 += bf2f((bf16_t)(q.y >> 16));
            }
            if (p.out_dtype == ADVGRPO_BF16) {
                uint2 pk;
                pk.x = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
                pk.y = (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16);
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.C) + o) = pk;
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + o) = make_float4(v0, v1, v2, v3);
            }
        } else {
            auto put = [&](int r, float y) {
                if (n + r >= p.N) return;
                if (p.bias) y += bf2f(p.bias[n + r]);
                y = act_fn(y, p.act);
                if (grow) y *= bf2f(grow[n + r]);
                if (rrow) y += bf2f(rrow[n + r]);
                if (p.out_dtype == ADVGRPO_BF16) reinterpret_cast<bf16_t*>(p.C)[o + r] = f2bf(y);
                else reinterpret_cast<float*>(p.C)[o + r] = y;
            };
            put(0, v0); put(1, v1); put(2, v2); put(3, v3);
        }
    });
}

// f32 in / f32 out epilogue of the split-bf16 VAE convolutions (GemmParams::f32_io): y = act(acc + bias_f32[n]) +
// residual_f32[m, n], f32 store, straight from the accumulator layout (a lane's 4 consecutive columns are one float4;
// the 4 lanes of a row cover 64 contiguous bytes).  Only the convolution kernels instantiate it.
// STATS: st[i][j][0 / 1] = sum / sum of squares of the lane's four stored values of fragment (i, j)
template <int FM, int FN, int TM, int TN, bool STATS = false>
__device__ __forceinline__ void gemm_epilogue_f32io(const GemmParams& p, f32x4 (&acc)[FM][FN], int m0, int n0, int wm,
                                                    int wn, int lane, float (*st)[FN][2] = nullptr) {
    const int mrow = lane & 15, ncol = (lane >> 4) * 4;
    const float* bias = reinterpret_cast<const float*>(p.bias);
    const float* res = reinterpret_cast<const float*>(p.residual);
    float* out = reinterpret_cast<float*>(p.C);
    const bool vec_ok = ((p.ldc | p.ldr) & 3) == 0 && (p.N & 3) == 0;
    static_for<FM * FN>([&](auto idx) {
        constexpr int i = decltype(idx)::value / FN, j = decltype(idx)::value % FN;
        const int m = m0 + wm * TM + i * 16 + mrow;
        const int n = n0 + wn * TN + j * 16 + ncol;
        if (m >= p.M || n >= p.N) return;
        const f32x4 a4 = acc[i][j];
        float v[4] = {a4[0] * p.alpha, a4[1] * p.alpha, a4[2] * p.alpha, a4[3] * p.alpha};    // (alpha = 1 everywhere but the f16x2 convolutions)
        if (vec_ok) {
            if (bias) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias + n);
                v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fn(v[e], p.act);
            if (res) {
                const float4 r4 = *reinterpret_cast<const float4*>(res + (int64_t)m * p.ldr + n);
                v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
            }
            if constexpr (STATS) {
                const float sv = (v[0] + v[1]) + (v[2] + v[3]);
                const float qv = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                st[i][j][0] = sv;
                st[i][j][1] = qv;
            }
            if (p.pair_out) {     // (uniform) the next convolution's operand rows directly: hi = f16(s y), lo = f16(s y - hi), as split8_f16 (vae.hip)
                uint32_t hw[2], lw[2];
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    const float a = fminf(fmaxf(v[e] * p.pair_prescale, -65504.f), 65504.f);        // (saturating, as split8_f16)
                    const float b = fminf(fmaxf(v[e + 1] * p.pair_prescale, -65504.f), 65504.f);
                    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
                    const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
                    hw[e >> 1] = (uint32_t)__builtin_bit_cast(uint16_t, ha) | ((uint32_t)__builtin_bit_cast(uint16_t, hb) << 16);
                    lw[e >> 1] = (uint32_t)__builtin_bit_cast(uint16_t, la) | ((uint32_t)__builtin_bit_cast(uint16_t, lb) << 16);
                }
                uint16_t* row = reinterpret_cast<uint16_t*>(p.pair_out) + (int64_t)m * 3 * p.N + n;
                *reinterpret_cast<uint2*>(row) = uint2{hw[0], hw[1]};
                *reinterpret_cast<uint2*>(row + 2 * p.N) = uint2{lw[0], lw[1]};
            } else
            // streaming store: up to 1 GiB of output that the next kernel reads only after it has left every cache
            __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(out + (int64_t)m * p.ldc + n));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e >= p.N) continue;
                float y = v[e];
                if (bias) y += bias[n + e];
                y = act_fn(y, p.act);
                if (res) y += res[(int64_t)m * p.ldr + n + e];
                out[(int64_t)m * p.ldc + n + e] = y;
            }
        }
    });
}

// all waves of the workgroup must be past their last main-loop LDS read when this is called (it syncs itself)
template <int FM, int FN, int TM, int TN, bool F32IO = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[FM][FN], int m0, int n0, int wm, int wn,
                                              int bz, int lane, char* smem, int wave) {
    if constexpr (F32IO) {
        if (p.f32_io) {
            gemm_epilogue_f32io<FM, FN, TM, TN>(p, acc, m0, n0, wm, wn, lane);
            return;
        }
    }
    if (epilogue_rows_ok(p) && !(p.debug & 16)) {
        __syncthreads();
        gemm_epilogue_rows<FM, FN, TM, TN>(p, acc, m0, n0, wm, wn, bz, lane, smem + wave * (16 * (TN * 4 + 16)));
    } else {
        gemm_epilogue_frag<FM, FN, TM, TN>(p, acc, m0, n0, wm, wn, bz, lane);
    }
}

// two independent problems served by one launch (same tile variant): workgroups [0, tiles_a) run problem a, the rest
// problem b.  Used to run the short text-stream Linear of a joint MMDiT block in the tail of its image-stream twin
// instead of as a second, badly filled launch (M = 16 x 205 rows against 256 CUs x 2 workgroups).
struct GemmPair { GemmParams a, b; int tiles_a; };

// split-bf16 3x3 convolution of the fp32-equivalent VAE decode (conv_x3.hip); fp16-pair activations x fp16 weights
int conv3x3_x3_launch(const GemmParams& p, hipStream_t s);
int conv3x3_f16x2_launch(const GemmParams& p, hipStream_t s, int form = 0 /* 0: f16x2, 1: bf16x2, 2: f16x1 */);

// 256x256 eight-phase kernel (gemm8p.hip)
bool gemm8p_ok(const GemmParams& p);
int gemm8p_launch(const GemmParams& p, hipStream_t s);
int gemm8p_launch_pair(const GemmParams& a, const GemmParams& b, hipStream_t s);
#ifdef ADVGRPO_EXPERIMENTS
// experiment (experiments/gemm4w.hip): the same wave tile in 4-wave workgroups, two per CU; b may be null
int gemm4w_launch_pair(const GemmParams* a, const GemmParams* b, hipStream_t s);
#endif

}  // namespace advgrpo
