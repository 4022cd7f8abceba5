// attention_bwd.hip -- backward of the fused attention for gfx950 (bf16 in/out, f32 math, head dim 64).
//
// Needed by the G-step: autograd of the transformer call in compute_log_prob
// (scripts/train_sd3_fast_pickscore.py:233-267) as reached from loss.backward() (:1165); diffusers runs
// F.scaled_dot_product_attention, whose backward this replaces.
//
// Three launches, no atomics, probabilities recomputed from the forward's base-2 log-sum-exp:
//   delta   D[q] = sum_d O[q,d] dO[q,d]                                       (HBM bound row kernel)
//   dq      per 128-query workgroup, loop over 64-key tiles (the forward's structure):
//           S^T = K Q^T, P^T = exp2(S^T c - L[q]), dP^T = V dO^T, dS^T = P^T (dP^T - D[q]),
//           dQ^T += K^T dS^T        -- lane = one query, so L and D are per-lane scalars
//   dkdv    per 128-key workgroup (wave = 32 keys), loop over 64-query tiles:
//           S = Q K^T, P = exp2(S c - L[q]), dV^T += dO^T P, dP = dO V^T, dS = P (dP - D[q]), dK^T += Q^T dS
// In both kernels the f32 score fragment is re-used in place as the MFMA B operand of the following product
// (with the same consistent permutation of the contraction index as the forward kernel), the "transposed"
// operands (K^T, dO^T, Q^T) come from row-major LDS tiles through ds_read_b64_tr_b16, and the accumulators are
// kept transposed so each lane ends with 4 consecutive d of one row: 8-byte stores.
#include <stdlib.h>

#include "common.hpp"
#include "gemm.hpp"

namespace advgrpo {

struct AttnBwdParams {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* o; const bf16_t* d_o;
    const float* lse; float* delta;
    bf16_t* dq; bf16_t* dk; bf16_t* dv;
    int64_t ldq, ldk, ldv, ldo, lddo, lddq;
    int64_t bsq, bsk, bsv, bso, bsdo, bsdq;
    int H, Sq, Skv;
    float scale, scale_log2e;
    int xcd_local;                // XCD-local block order (common.hpp xcd_local_bh)
};

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
constexpr int BW_PITCH = 72;   // LDS row pitch (elements) of 64-wide tiles: 144 B
constexpr int BW_TILE = 64 * BW_PITCH;

// A operand = transpose of a row-major LDS tile: rows (row0 + g*4 + j) and (+16), columns col0 + t
__device__ inline bf16x8_t tr_frag(const bf16_t* tile, int row0, int col0, int g, int t) {
    const bf16_t* a0 = tile + (row0 + g * 4 + (t >> 2)) * BW_PITCH + col0 + (t & 3) * 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(a0 + 16 * BW_PITCH));
    const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, both);
}
__device__ inline bf16x8_t pack_pair(const f32x4& a, const f32x4& b) {
    bf16x8_t f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        f[r] = (__bf16)a[r];
        f[4 + r] = (__bf16)b[r];
    }
    return f;
}

// ---------------------------------------------------------------- delta[b,h,q] = sum_d O dO
__global__ __launch_bounds__(256) void attn_bwd_delta_kernel(const AttnBwdParams p, int B) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (b, q)
    if (row >= (int64_t)B * p.Sq) return;
    const int b = row / p.Sq, q = row % p.Sq;
    const bf16_t* o = p.o + (int64_t)b * p.bso + (int64_t)q * p.ldo;
    const bf16_t* d = p.d_o + (int64_t)b * p.bsdo + (int64_t)q * p.lddo;
    const int sub = lane & 7;
    for (int h0 = 0; h0 < p.H; h0 += 8) {
        const int h = h0 + (lane >> 3);
        float s = 0.f;
        if (h < p.H) {
            const uint4 a = *reinterpret_cast<const uint4*>(o + h * 64 + sub * 8);
            const uint4 c = *reinterpret_cast<const uint4*>(d + h * 64 + sub * 8);
            const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                s += bf2f((bf16_t)(aw[k] & 0xffffu)) * bf2f((bf16_t)(cw[k] & 0xffffu)) +
                     bf2f((bf16_t)(aw[k] >> 16)) * bf2f((bf16_t)(cw[k] >> 16));
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (h < p.H && sub == 0) p.delta[((int64_t)b * p.H + h) * p.Sq + q] = s;
    }
}

// ---------------------------------------------------------------- dQ
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(const AttnBwdParams p) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * BW_TILE];  // K0 K1 V0 V1
    bf16_t* Ks = smem;
    bf16_t* Vs = smem + 2 * BW_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, t = lane & 15;
    int qblk, h, b;
    xcd_local_bh((p.Sq + 127) / 128, p.H, (int)gridDim.x, p.xcd_local, qblk, h, b);
    const int q0 = qblk * 128 + wave * 32;
    const bf16_t* qp = p.q + (int64_t)b * p.bsq + h * 64;
    const bf16_t* kp = p.k + (int64_t)b * p.bsk + h * 64;
    const bf16_t* vp = p.v + (int64_t)b * p.bsv + h * 64;
    const bf16_t* dop = p.d_o + (int64_t)b * p.bsdo + h * 64;

    bf16x8_t qf[2][2], dof[2][2];
    float L[2], Dl[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int qr = q0 + qb * 16 + t;
        qr = qr < p.Sq ? qr : p.Sq - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            qf[qb][ks] = *reinterpret_cast<const bf16x8_t*>(qp + (int64_t)qr * p.ldq + ks * 32 + g * 8);
            dof[qb][ks] = *reinterpret_cast<const bf16x8_t*>(dop + (int64_t)qr * p.lddo + ks * 32 + g * 8);
        }
        L[qb] = p.lse[((int64_t)b * p.H + h) * p.Sq + qr];
        Dl[qb] = p.delta[((int64_t)b * p.H + h) * p.Sq + qr];
    }
    uint4 kreg0, kreg1, vreg0, vreg1;
    const int srow0 = tid >> 3, scol = (tid & 7) * 8;
    auto issue = [&](int kv0) __attribute__((always_inline)) {
        int r0 = kv0 + srow0, r1 = r0 + 32;
        r0 = r0 < p.Skv ? r0 : p.Skv - 1;
        r1 = r1 < p.Skv ? r1 : p.Skv - 1;
        kreg0 = *reinterpret_cast<const uint4*>(kp + (int64_t)r0 * p.ldk + scol);
        kreg1 = *reinterpret_cast<const uint4*>(kp + (int64_t)r1 * p.ldk + scol);
        vreg0 = *reinterpret_cast<const uint4*>(vp + (int64_t)r0 * p.ldv + scol);
        vreg1 = *reinterpret_cast<const uint4*>(vp + (int64_t)r1 * p.ldv + scol);
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
        const int off0 = srow0 * BW_PITCH + scol, off1 = off0 + 32 * BW_PITCH;
        *reinterpret_cast<uint4*>(Ks + buf * BW_TILE + off0) = kreg0;
        *reinterpret_cast<uint4*>(Ks + buf * BW_TILE + off1) = kreg1;
        *reinterpret_cast<uint4*>(Vs + buf * BW_TILE + off0) = vreg0;
        *reinterpret_cast<uint4*>(Vs + buf * BW_TILE + off1) = vreg1;
    };
    f32x4 dq[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) dq[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = (p.Skv + 63) / 64;
    issue(0);
    commit(0);
    __syncthreads();
    for (int it = 0; it < nt; ++it) {
        const int buf = it & 1, kv0 = it * 64;
        if (it + 1 < nt) issue(kv0 + 64);
        const bf16_t* Kt = Ks + buf * BW_TILE;
        const bf16_t* Vt = Vs + buf * BW_TILE;
        f32x4 s[4][2], dp[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) { s[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Kt + (kb * 16 + t) * BW_PITCH + ks * 32 + g * 8);
                const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(Vt + (kb * 16 + t) * BW_PITCH + ks * 32 + g * 8);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qb][ks], s[kb][qb], 0, 0, 0);
                    dp[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[qb][ks], dp[kb][qb], 0, 0, 0);
                }
            }
        }
        // lane: key = kv0 + kb*16 + g*4 + r, query = q0 + qb*16 + t
        bf16x8_t dsf[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kv0 + kb * 16 + g * 4 + r;
                    float pr = __builtin_amdgcn_exp2f(s[kb][qb][r] * p.scale_log2e - L[qb]);
                    if (key >= p.Skv) pr = 0.f;
                    s[kb][qb][r] = pr * (dp[kb][qb][r] - Dl[qb]);
                }
            dsf[qb][0] = pack_pair(s[0][qb], s[1][qb]);
            dsf[qb][1] = pack_pair(s[2][qb], s[3][qb]);
        }
        // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
        for (int kpair = 0; kpair < 2; ++kpair)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const bf16x8_t ktf = tr_frag(Kt, 2 * kpair * 16, db * 16, g, t);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    dq[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf, dsf[qb][kpair], dq[db][qb], 0, 0, 0);
            }
        if (it + 1 < nt) commit(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        const int qi = q0 + qb * 16 + t;
        if (qi >= p.Sq) continue;
        bf16_t* op = p.dq + (int64_t)b * p.bsdq + (int64_t)qi * p.lddq + h * 64 + g * 4;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const f32x4 v = dq[db][qb] * p.scale;
            uint2 pk;
            pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
            pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
            *reinterpret_cast<uint2*>(op + db * 16) = pk;
        }
    }
}

// ---------------------------------------------------------------- dK, dV
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(const AttnBwdParams p) {
    // Q0 Q1 dO0 dO1 tiles + L / D vectors for the two buffers
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * BW_TILE];
    __shared__ float lds_L[2][64], lds_D[2][64];
    bf16_t* Qs = smem;
    bf16_t* Os = smem + 2 * BW_TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, t = lane & 15;
    int kblk, h, b;
    xcd_local_bh((p.Skv + 127) / 128, p.H, (int)gridDim.x, p.xcd_local, kblk, h, b);
    const int k0 = kblk * 128 + wave * 32;                 // this wave's 32 keys
    const bf16_t* qp = p.q + (int64_t)b * p.bsq + h * 64;
    const bf16_t* kp = p.k + (int64_t)b * p.bsk + h * 64;
    const bf16_t* vp = p.v + (int64_t)b * p.bsv + h * 64;
    const bf16_t* dop = p.d_o + (int64_t)b * p.bsdo + h * 64;
    const float* lsep = p.lse + ((int64_t)b * p.H + h) * p.Sq;
    const float* delp = p.delta + ((int64_t)b * p.H + h) * p.Sq;

    // K, V fragments as B operands: lane = key t of block kb, 8 consecutive d
    bf16x8_t kf[2][2], vf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        int kr = k0 + kb * 16 + t;
        kr = kr < p.Skv ? kr : p.Skv - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kf[kb][ks] = *reinterpret_cast<const bf16x8_t*>(kp + (int64_t)kr * p.ldk + ks * 32 + g * 8);
            vf[kb][ks] = *reinterpret_cast<const bf16x8_t*>(vp + (int64_t)kr * p.ldv + ks * 32 + g * 8);
        }
    }
    uint4 qreg0, qreg1, oreg0, oreg1;
    float lreg = 0.f, dreg = 0.f;
    const int srow0 = tid >> 3, scol = (tid & 7) * 8;
    auto issue = [&](int qs0) __attribute__((always_inline)) {
        int r0 = qs0 + srow0, r1 = r0 + 32;
        r0 = r0 < p.Sq ? r0 : p.Sq - 1;
        r1 = r1 < p.Sq ? r1 : p.Sq - 1;
        qreg0 = *reinterpret_cast<const uint4*>(qp + (int64_t)r0 * p.ldq + scol);
        qreg1 = *reinterpret_cast<const uint4*>(qp + (int64_t)r1 * p.ldq + scol);
        oreg0 = *reinterpret_cast<const uint4*>(dop + (int64_t)r0 * p.lddo + scol);
        oreg1 = *reinterpret_cast<const uint4*>(dop + (int64_t)r1 * p.lddo + scol);
        if (tid < 64) {
            const int r = qs0 + tid;
            lreg = r < p.Sq ? lsep[r] : INFINITY;    // +inf => P = exp2(-inf) = 0 for padded queries
            dreg = r < p.Sq ? delp[r] : 0.f;
        }
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
        const int off0 = srow0 * BW_PITCH + scol, off1 = off0 + 32 * BW_PITCH;
        *reinterpret_cast<uint4*>(Qs + buf * BW_TILE + off0) = qreg0;
        *reinterpret_cast<uint4*>(Qs + buf * BW_TILE + off1) = qreg1;
        *reinterpret_cast<uint4*>(Os + buf * BW_TILE + off0) = oreg0;
        *reinterpret_cast<uint4*>(Os + buf * BW_TILE + off1) = oreg1;
        if (tid < 64) { lds_L[buf][tid] = lreg; lds_D[buf][tid] = dreg; }
    };
    f32x4 dk[4][2], dv[4][2];   // [d block][key block]: lane = key t, 4 consecutive d
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { dk[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    const int nt = (p.Sq + 63) / 64;
    issue(0);
    commit(0);
    __syncthreads();
    for (int it = 0; it < nt; ++it) {
        const int buf = it & 1;
        if (it + 1 < nt) issue((it + 1) * 64);
        const bf16_t* Qt = Qs + buf * BW_TILE;
        const bf16_t* Ot = Os + buf * BW_TILE;
        // S[q][key] = Q K^T and dP[q][key] = dO V^T : A from LDS rows (queries), B = this wave's K / V fragments
        f32x4 s[4][2], dp[4][2];    // [query block][key block]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) { s[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int qb = 0; qb < 4; ++qb) {
                const bf16x8_t qa = *reinterpret_cast<const bf16x8_t*>(Qt + (qb * 16 + t) * BW_PITCH + ks * 32 + g * 8);
                const bf16x8_t oa = *reinterpret_cast<const bf16x8_t*>(Ot + (qb * 16 + t) * BW_PITCH + ks * 32 + g * 8);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    s[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kf[kb][ks], s[qb][kb], 0, 0, 0);
                    dp[qb][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa, vf[kb][ks], dp[qb][kb], 0, 0, 0);
                }
            }
        // lane: query = qb*16 + g*4 + r (tile local), key = k0 + kb*16 + t
        bf16x8_t pf[2][2], dsf[2][2];   // [key block][query pair]
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const bool key_ok = (k0 + kb * 16 + t) < p.Skv;
#pragma unroll
            for (int qb = 0; qb < 4; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql = qb * 16 + g * 4 + r;
                    float pr = __builtin_amdgcn_exp2f(s[qb][kb][r] * p.scale_log2e - lds_L[buf][ql]);
                    if (!key_ok) pr = 0.f;
                    s[qb][kb][r] = pr;
                    dp[qb][kb][r] = pr * (dp[qb][kb][r] - lds_D[buf][ql]);
                }
            pf[kb][0] = pack_pair(s[0][kb], s[1][kb]);
            pf[kb][1] = pack_pair(s[2][kb], s[3][kb]);
            dsf[kb][0] = pack_pair(dp[0][kb], dp[1][kb]);
            dsf[kb][1] = pack_pair(dp[2][kb], dp[3][kb]);
        }
        // dV^T[d][key] += dO^T[d][q] P[q][key] ;  dK^T[d][key] += Q^T[d][q] dS[q][key]
#pragma unroll
        for (int qpair = 0; qpair < 2; ++qpair)
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const bf16x8_t otf = tr_frag(Ot, 2 * qpair * 16, db * 16, g, t);
                const bf16x8_t qtf = tr_frag(Qt, 2 * qpair * 16, db * 16, g, t);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    dv[db][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(otf, pf[kb][qpair], dv[db][kb], 0, 0, 0);
                    dk[db][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsf[kb][qpair], dk[db][kb], 0, 0, 0);
                }
            }
        if (it + 1 < nt) commit(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int ki = k0 + kb * 16 + t;
        if (ki >= p.Skv) continue;
        bf16_t* okp = p.dk + (int64_t)b * p.bsdq + (int64_t)ki * p.lddq + h * 64 + g * 4;
        bf16_t* ovp = p.dv + (int64_t)b * p.bsdq + (int64_t)ki * p.lddq + h * 64 + g * 4;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const f32x4 a = dk[db][kb] * p.scale, c = dv[db][kb];
            uint2 pk;
            pk.x = (uint32_t)f2bf(a[0]) | ((uint32_t)f2bf(a[1]) << 16);
            pk.y = (uint32_t)f2bf(a[2]) | ((uint32_t)f2bf(a[3]) << 16);
            *reinterpret_cast<uint2*>(okp + db * 16) = pk;
            pk.x = (uint32_t)f2bf(c[0]) | ((uint32_t)f2bf(c[1]) << 16);
            pk.y = (uint32_t)f2bf(c[2]) | ((uint32_t)f2bf(c[3]) << 16);
            *reinterpret_cast<uint2*>(ovp + db * 16) = pk;
        }
    }
}

}  // namespace advgrpo

using namespace advgrpo;

extern "C" int advgrpo_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                     const float* lse, float* delta, void* dq, void* dk, void* dv, int64_t ldq,
                                     int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq, int64_t bsq,
                                     int64_t bsk, int64_t bsv, int64_t bso, int64_t bsdo, int64_t bsdq, int B, int H,
                                     int Sq, int Skv, int head_dim, float scale, void* stream) {
    ADVGRPO_CHECK(head_dim == 64, "attention_bwd: head_dim %d not supported (64)", head_dim);
    ADVGRPO_CHECK(q && k && v && o && d_o && lse && delta && dq && dk && dv, "attention_bwd: null pointer");
    ADVGRPO_CHECK(B > 0 && H > 0 && Sq > 0 && Skv > 0, "attention_bwd: bad shape");
    ADVGRPO_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && lddq % 4 == 0,
                  "attention_bwd: row pitches must keep 16-byte (inputs) / 8-byte (outputs) alignment");
    AttnBwdParams p{};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (const bf16_t*)o;
    p.d_o = (const bf16_t*)d_o; p.lse = lse; p.delta = delta;
    p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq;
    p.bsq = bsq; p.bsk = bsk; p.bsv = bsv; p.bso = bso; p.bsdo = bsdo; p.bsdq = bsdq;
    p.H = H; p.Sq = Sq; p.Skv = Skv; p.scale = scale; p.scale_log2e = scale * 1.4426950408889634f;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(attn_bwd_delta_kernel, dim3((unsigned)(((int64_t)B * Sq + 3) / 4)), dim3(256), 0, s, p, B);
    ADVGRPO_LAUNCH_CHECK();
    int xcd_local = 1;
#ifdef ADVGRPO_EXPERIMENTS
    { const char* e = getenv("ADVGRPO_ATTN_NO_XCD"); if (e && atoi(e)) xcd_local = 0; }
#endif
    p.xcd_local = xcd_local;
    ADVGRPO_CHECK((int64_t)((Sq + 127) / 128) * H * B < (1ll << 31) && (int64_t)((Skv + 127) / 128) * H * B < (1ll << 31),
                  "attention_bwd: grid too large");
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3((unsigned)((Sq + 127) / 128) * H * B), dim3(256), 0, s, p);
    ADVGRPO_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel, dim3((unsigned)((Skv + 127) / 128) * H * B), dim3(256), 0, s, p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}
