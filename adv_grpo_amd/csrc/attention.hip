// attention.hip -- fused softmax(Q K^T * scale) V for gfx950 (flash-style, bf16 in/out, f32 math).
//
// Serves the joint image+text attention of the MMDiT blocks (1229 tokens at 512^2, 24 heads x 64;
// reference call sites sd3_pipeline_with_logprob_fast.py:630-637 / train_sd3_fast_pickscore.py:235-255
// via diffusers' JointAttnProcessor2_0 -> F.scaled_dot_product_attention) and the ViT reward
// towers (DINOv2-B 1370 tokens, 12 heads x 64; CLIP text 77 tokens causal).
//
// CDNA4 mapping (wave64, v_mfma_f32_16x16x32_bf16):
//   * workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32 queries.
//   * S^T = K Q^T is computed with K as the MFMA A operand, so a lane holds scores of ONE query
//     (column lane&15) for 16 keys: row max / row sum are in-lane plus two cross-lane steps, and
//     the probabilities are already in the B-operand layout of the P V product -- no LDS round
//     trip and no permutes between the two MFMAs (the key order inside a 32-deep MFMA step is
//     permuted consistently on the P and the V side, which a sum over keys does not see).
//   * V^T fragments come from the row-major V tile through ds_read_b64_tr_b16 (hardware
//     transpose), K fragments through ds_read_b128; both tiles use a 144-byte row pitch, which
//     spreads the 16 rows of a fragment over all 16-byte bank slots.
//   * K/V tiles are register staged: the next tile's global loads are issued before the current
//     tile's MFMAs and written to the other LDS buffer afterwards (one barrier per tile).
//   * O is kept transposed in the accumulators (lane = one query, 4 consecutive d): the online
//     softmax rescale is a per-lane scalar and the epilogue is an 8-byte bf16x4 store per lane.
#include <stdlib.h>

#include "attention.hpp"

namespace advgrpo {


// HD = 64 (MMDiT, DINOv2, CLIP text) or 80 (CLIP ViT-H vision: 1280 / 16 heads).  For HD = 80 the
// QK^T contraction is padded to 96 with zero columns in LDS (K) and zero Q fragments.
template <int HD>
__global__ __launch_bounds__(256) void attention_fwd_kernel(const AttnParams p) {
    static_assert(HD == 64 || HD == 80, "head dim 64 or 80");
    constexpr int KS = (HD + 31) / 32;          // 32-deep MFMA steps over d
    constexpr int DB = HD / 16;                 // 16-wide output blocks over d
    constexpr int ATT_PITCH = KS * 32 + 8;      // LDS row pitch in elements: 144 B / 208 B, odd multiple of 16 B
    constexpr int CPR = HD / 8;                 // 16-byte chunks per row
    constexpr int NCH = (ATT_KB * CPR + 255) / 256;  // staging chunks per thread
    constexpr int TILE = ATT_KB * ATT_PITCH;  // elements per K or V tile
    __shared__ __attribute__((aligned(16))) bf16_t smem[4 * TILE];  // K0 K1 V0 V1
    bf16_t* Ks = smem;
    bf16_t* Vs = smem + 2 * TILE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, t = lane & 15;
    int qblk, h, b;
    xcd_local_bh(p.nqb, p.H, p.nwg, p.xcd_local, qblk, h, b);
    const int q0 = qblk * ATT_QB + wave * 32;

    const bf16_t* qp = p.q + (int64_t)b * p.bsq + h * HD;
    const bf16_t* kp = p.k + (int64_t)b * p.bsk + h * HD;
    const bf16_t* vp = p.v + (int64_t)b * p.bsv + h * HD;

    // ---- Q fragments (B operand: lane = query t, 8 consecutive d at (ks*32 + g*8))
    bf16x8_t qf[2][KS];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int qr = q0 + qb * 16 + t;
        qr = qr < p.Sq ? qr : p.Sq - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks * 32 + g * 8 < HD)
                qf[qb][ks] = *reinterpret_cast<const bf16x8_t*>(qp + (int64_t)qr * p.ldq + ks * 32 + g * 8);
            else
                qf[qb][ks] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }

    // ---- staging: ATT_KB * CPR 16-byte chunks per tile, NCH per thread for K and for V
    uint4 kreg[NCH], vreg[NCH];
    auto issue = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * 256;
            if (c < ATT_KB * CPR) {
                int r = kv0 + c / CPR;
                r = r < p.Skv ? r : p.Skv - 1;
                kreg[i] = *reinterpret_cast<const uint4*>(kp + (int64_t)r * p.ldk + (c % CPR) * 8);
                vreg[i] = *reinterpret_cast<const uint4*>(vp + (int64_t)r * p.ldv + (c % CPR) * 8);
            }
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + i * 256;
            if (c < ATT_KB * CPR) {
                const int off = (c / CPR) * ATT_PITCH + (c % CPR) * 8;
                *reinterpret_cast<uint4*>(Ks + buf * TILE + off) = kreg[i];
                *reinterpret_cast<uint4*>(Vs + buf * TILE + off) = vreg[i];
            }
        }
    };
    if constexpr (KS * 32 != HD) {  // zero the padded K columns of both buffers once
        for (int i = tid; i < 2 * ATT_KB * (KS * 32 - HD) / 8; i += 256) {
            const int row = i / ((KS * 32 - HD) / 8), cc = i % ((KS * 32 - HD) / 8);
            *reinterpret_cast<uint4*>(Ks + row * ATT_PITCH + HD + cc * 8) = uint4{0, 0, 0, 0};
        }
    }

    f32x4 o[DB][2];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) o[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

    int kv_end = p.Skv;
    if (p.causal) {  // keys <= last query of the workgroup
        const int last_q = min(qblk * ATT_QB + ATT_QB, p.Sq) - 1;
        kv_end = min(kv_end, last_q + 1);
    }
    const int nt = (kv_end + ATT_KB - 1) / ATT_KB;

    issue(0);
    commit(0);
    __syncthreads();
    for (int it = 0; it < nt; ++it) {
        const int buf = it & 1, kv0 = it * ATT_KB;
        if (it + 1 < nt) issue(kv0 + ATT_KB);
        const bf16_t* Kt = Ks + buf * TILE;
        const bf16_t* Vt = Vs + buf * TILE;

        // ---- S^T[key][q] = K Q^T
        f32x4 s[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) s[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8_t kf[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
                kf[kb] = *reinterpret_cast<const bf16x8_t*>(Kt + (kb * 16 + t) * ATT_PITCH + ks * 32 + g * 8);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb], qf[qb][ks], s[kb][qb], 0, 0, 0);
        }
        // lane holds S[key = kv0 + kb*16 + g*4 + r][query = q0 + qb*16 + t]
        const bool edge = (kv0 + ATT_KB > p.Skv) || p.causal;
        if (edge) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kv0 + kb * 16 + g * 4 + r;
                        const int qi = q0 + qb * 16 + t;
                        if (key >= p.Skv || (p.causal && key > qi)) s[kb][qb][r] = -INFINITY;
                    }
        }
        // ---- online softmax (base-2), per query block
        bf16x8_t pf[2][2];  // [qb][key pair]
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][qb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run[qb], mx * p.scale_log2e);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // fully masked so far
            const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_use);
            m_run[qb] = m_new;
            float psum = 0.f;
            float pv[4][4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(s[kb][qb][r] * p.scale_log2e - m_use);
                    pv[kb][r] = e;
                    psum += e;
                }
            l_run[qb] = l_run[qb] * alpha + psum;
#pragma unroll
            for (int db = 0; db < DB; ++db) o[db][qb] *= alpha;
#pragma unroll
            for (int kpair = 0; kpair < 2; ++kpair) {
                bf16x8_t f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f[r] = (__bf16)pv[2 * kpair][r];
                    f[4 + r] = (__bf16)pv[2 * kpair + 1][r];
                }
                pf[qb][kpair] = f;
            }
        }
        // ---- O^T[d][q] += V^T P^T ; A operand = V^T via transpose reads
#pragma unroll
        for (int kpair = 0; kpair < 2; ++kpair) {
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const bf16_t* a0 = Vt + ((2 * kpair) * 16 + g * 4 + (t >> 2)) * ATT_PITCH + db * 16 + (t & 3) * 4;
                const bf16_t* a1 = a0 + 16 * ATT_PITCH;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(a0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(a1));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, both);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb][kpair], o[db][qb], 0, 0, 0);
            }
        }
        if (it + 1 < nt) commit(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: normalise and store bf16x4 (query t, d = db*16 + g*4 .. +3)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float l = l_run[qb];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        const int qi = q0 + qb * 16 + t;
        if (qi >= p.Sq) continue;
        if (p.lse && g == 0) p.lse[((int64_t)b * p.H + h) * p.Sq + qi] = m_run[qb] + __builtin_amdgcn_logf(l);
        bf16_t* op = p.o + (int64_t)b * p.bso + (int64_t)qi * p.ldo + h * HD + g * 4;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const f32x4 v = o[db][qb] * inv;
            uint2 pk;
            pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
            pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
            *reinterpret_cast<uint2*>(op + db * 16) = pk;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Short sequences with the whole K / V of a (batch, head) resident in LDS: the CLIP ViT-H vision tower of the PickScore
// scorer (257 tokens, 16 heads x 80; pickscore_scorer.py:40-44 -> transformers' CLIPAttention).  The tiled kernel above walks
// 5 key tiles with a global load -> register -> LDS round trip and a barrier per tile while 1-2 workgroups per CU have
// nothing to hide that latency behind (26 us for 2.7 GFLOP: profiles/r6_vit_tower.txt).  Here a workgroup requests every K / V
// row (and its Q fragments) at once, waits once, and its waves then run their 16-query chunks over the resident tiles with
// no further barrier (8 waves: the softmax of one wave is VALU work beside the other wave's MFMAs on the same SIMD).  Two workgroups per (batch, head) -- the two halves of the query chunks, neighbours in the XCD-local
// order so the second one finds K / V in L2 -- fill the 256 CUs once at 8 images x 16 heads.  K rows sit at 256 bytes
// with 16-byte chunk c of row r at c ^ (r & 15): a ds_read_b128 service group mixes lanes of two neighbouring chunks
// (MI355X_MICROARCH: {0-3, 12-15, 20-27}, ...), and the tiled kernel's 208-byte pitch puts (chunk c, row r + 5) on the slot of
// (chunk c + 1, row r) -- five two-way conflicts per group, the bank-conflict share of round 5's PMC; V rows are packed at
// 160 bytes: the 8 rows x 32 bytes a ds_read_b64_tr_b16 lane group touches tile the 256 bytes of the banks exactly
// (0,160,64,224,128,32,192,96).
// Same arithmetic in the same order as attention_fwd_kernel<80> (key blocks past the last key are skipped: they add zeros).
constexpr int RES_MAXT = 5;        // resident key tiles of ATT_KB rows (Skv <= 320)
constexpr int RES_MAXC = 2;        // query chunks per wave
constexpr int RES_WAVES = 8;       // waves per workgroup: two per SIMD, so one wave's softmax (VALU) runs beside the other's MFMAs
constexpr int RES_NQB = 1;         // 16-query blocks per chunk: 257 rows = 17 chunks over 2 x 8 waves (one wave takes two)
constexpr int RES_THREADS = RES_WAVES * 64;
constexpr int RES_MAX_SQ = 2 * RES_WAVES * RES_MAXC * RES_NQB * 16;   // 512

template <int HD>
__global__ __launch_bounds__(RES_THREADS) void attention_fwd_resident_kernel(const AttnParams p) {
    static_assert(HD % 16 == 0, "head dim must be a multiple of 16");
    constexpr int KS = (HD + 31) / 32, DB = HD / 16, CPR = HD / 8;
    constexpr int KP = 128, VP = HD;                               // LDS row pitches in elements: K 256 bytes with chunk ^ (row & 15), V packed
    static_assert(KS * 4 <= 16, "the K swizzle holds 16 chunks of 16 bytes per row");
    constexpr int PADC = (KS * 32 - HD) / 8;                       // zero chunks behind a K row
    constexpr int NQB = RES_NQB, QC = NQB * 16;                    // queries per chunk
    constexpr int NCH = (RES_MAXT * ATT_KB * CPR + RES_THREADS - 1) / RES_THREADS;     // staging chunks per thread
    extern __shared__ __attribute__((aligned(16))) char res_smem[];
    const int nt = (p.Skv + ATT_KB - 1) / ATT_KB;
    bf16_t* Ks = reinterpret_cast<bf16_t*>(res_smem);
    bf16_t* Vs = Ks + RES_MAXT * ATT_KB * KP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, t = lane & 15;
    int half, h, b;
    xcd_local_bh(2, p.H, p.nwg, p.xcd_local, half, h, b);
    const int nchunks = (p.Sq + QC - 1) / QC;
    const int c_lo = half ? nchunks / 2 : 0, c_hi = half ? nchunks : nchunks / 2;

    const bf16_t* qp = p.q + (int64_t)b * p.bsq + h * HD;
    const bf16_t* kp = p.k + (int64_t)b * p.bsk + h * HD;
    const bf16_t* vp = p.v + (int64_t)b * p.bsv + h * HD;

    // ---- Q fragments of this wave's chunks first (B operand: lane = query t, 8 consecutive d at ks*32 + g*8) ...
    bf16x8_t qf[RES_MAXC][NQB][KS];
#pragma unroll
    for (int c = 0; c < RES_MAXC; ++c) {
        const int chunk = c_lo + wave + c * RES_WAVES;
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            int qr = chunk * QC + qb * 16 + t;
            qr = qr < p.Sq ? qr : p.Sq - 1;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks * 32 + g * 8 < HD && chunk < c_hi)
                    qf[c][qb][ks] = *reinterpret_cast<const bf16x8_t*>(qp + (int64_t)qr * p.ldq + ks * 32 + g * 8);
                else
                    qf[c][qb][ks] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    }
    // ---- ... then every K / V row of the head: all requests in flight together, one wait, one barrier.  The chunk count is the
    // compile-time maximum (rows past Skv re-read the last row and are masked or never read): predicated staging registers
    // end up in scratch with a wait per chunk.
    {
        constexpr int TOTAL = RES_MAXT * ATT_KB * CPR;
        static_assert(NCH == 7, "the staging below is written out for 7 chunks per thread (5 tiles x 64 rows x 10 chunks / 512 threads)");
        auto src = [&](const bf16_t* base, int64_t ld, int i) {
            const int c = min(tid + i * RES_THREADS, TOTAL - 1);
            int r = c / CPR;
            r = r < p.Skv ? r : p.Skv - 1;
            return *reinterpret_cast<const uint4*>(base + (int64_t)r * ld + (c % CPR) * 8);
        };
        auto dst = [&](bf16_t* base, int pitch, int i, const uint4& v, bool swz) {
            const int c = min(tid + i * RES_THREADS, TOTAL - 1);       // (the clamped tail writes the last chunk again: same bytes)
            const int r = c / CPR, cc = c % CPR;
            *reinterpret_cast<uint4*>(base + r * pitch + (swz ? (cc ^ (r & 15)) : cc) * 8) = v;
        };
        // (named values, not arrays: with a scheduling fence between the requests and the LDS writes the compiler keeps staging ARRAYS in scratch)
#define RES_LD(i) const uint4 k##i = src(kp, p.ldk, i), v##i = src(vp, p.ldv, i);
#define RES_ST(i) dst(Ks, KP, i, k##i, true); dst(Vs, VP, i, v##i, false);
        RES_LD(0) RES_LD(1) RES_LD(2) RES_LD(3) RES_LD(4) RES_LD(5) RES_LD(6)
        __builtin_amdgcn_sched_barrier(0);   // every request issued before the first LDS write (else: load, wait, write, 7 times over)
        if constexpr (PADC > 0) {
            for (int i = tid; i < RES_MAXT * ATT_KB * PADC; i += RES_THREADS)
                *reinterpret_cast<uint4*>(Ks + (i / PADC) * KP + ((CPR + i % PADC) ^ ((i / PADC) & 15)) * 8) = uint4{0, 0, 0, 0};
        }
        RES_ST(0) RES_ST(1) RES_ST(2) RES_ST(3) RES_ST(4) RES_ST(5) RES_ST(6)
#undef RES_LD
#undef RES_ST
    }
    __syncthreads();

#pragma unroll
    for (int c = 0; c < RES_MAXC; ++c) {
        const int chunk = c_lo + wave + c * RES_WAVES;
        if (chunk >= c_hi) continue;                 // wave-uniform
        const int q0 = chunk * QC;
        f32x4 o[DB][NQB];
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int j = 0; j < NQB; ++j) o[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        float m_run[NQB], l_run[NQB];
#pragma unroll
        for (int j = 0; j < NQB; ++j) { m_run[j] = -INFINITY; l_run[j] = 0.f; }

        for (int it = 0; it < nt; ++it) {
            const int kv0 = it * ATT_KB;
            const bf16_t* Kt = Ks + kv0 * KP;
            const bf16_t* Vt = Vs + kv0 * VP;
            const int nkb = min(4, (p.Skv - kv0 + 15) >> 4);     // 16-key blocks of this tile that hold a key (wave-uniform)

            // ---- S^T[key][q] = K Q^T
            f32x4 s[4][NQB];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NQB; ++j) s[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    if (kb < nkb) {
                        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Kt + (kb * 16 + t) * KP + ((ks * 4 + g) ^ t) * 8);
#pragma unroll
                        for (int qb = 0; qb < NQB; ++qb)
                            s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[c][qb][ks], s[kb][qb], 0, 0, 0);
                    }
                }
            }
            // lane holds S[key = kv0 + kb*16 + g*4 + r][query = q0 + qb*16 + t]
            if (kv0 + ATT_KB > p.Skv) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (kv0 + kb * 16 + g * 4 + r >= p.Skv) s[kb][qb][r] = -INFINITY;
            }
            // ---- online softmax (base-2), per query block
            bf16x8_t pf[NQB][2];  // [qb][key pair]
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) {
                float mx = -INFINITY;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][qb][r]);
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run[qb], mx * p.scale_log2e);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_use);
                m_run[qb] = m_new;
                float psum = 0.f;
                float pv[4][4];
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float e = __builtin_amdgcn_exp2f(s[kb][qb][r] * p.scale_log2e - m_use);
                        pv[kb][r] = e;
                        psum += e;
                    }
                l_run[qb] = l_run[qb] * alpha + psum;
#pragma unroll
                for (int db = 0; db < DB; ++db) o[db][qb] *= alpha;
#pragma unroll
                for (int kpair = 0; kpair < 2; ++kpair) {
                    bf16x8_t f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        f[r] = (__bf16)pv[2 * kpair][r];
                        f[4 + r] = (__bf16)pv[2 * kpair + 1][r];
                    }
                    pf[qb][kpair] = f;
                }
            }
            // ---- O^T[d][q] += V^T P^T ; A operand = V^T via transpose reads
#pragma unroll
            for (int kpair = 0; kpair < 2; ++kpair) {
                if (2 * kpair < nkb) {
#pragma unroll
                    for (int db = 0; db < DB; ++db) {
                        const bf16_t* a0 = Vt + ((2 * kpair) * 16 + g * 4 + (t >> 2)) * VP + db * 16 + (t & 3) * 4;
                        const bf16_t* a1 = a0 + 16 * VP;
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a1));
                        typedef __attribute__((ext_vector_type(8))) short s16x8;
                        const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, both);
#pragma unroll
                        for (int qb = 0; qb < NQB; ++qb)
                            o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb][kpair], o[db][qb], 0, 0, 0);
                    }
                }
            }
        }

        // ---- epilogue: normalise and store bf16x4 (query t, d = db*16 + g*4 .. +3)
#pragma unroll
        for (int qb = 0; qb < NQB; ++qb) {
            float l = l_run[qb];
            l += __shfl_xor(l, 16, 64);
            l += __shfl_xor(l, 32, 64);
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            const int qi = q0 + qb * 16 + t;
            if (qi >= p.Sq) continue;
            if (p.lse && g == 0) p.lse[((int64_t)b * p.H + h) * p.Sq + qi] = m_run[qb] + __builtin_amdgcn_logf(l);
            bf16_t* op = p.o + (int64_t)b * p.bso + (int64_t)qi * p.ldo + h * HD + g * 4;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const f32x4 v = o[db][qb] * inv;
                uint2 pk;
                pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                *reinterpret_cast<uint2*>(op + db * 16) = pk;
            }
        }
    }
}

static int attention_fwd_resident_launch(AttnParams p, int B, hipStream_t s) {
    constexpr int HD = 80, KP = 128, VP = 80;
    const size_t lds = (size_t)RES_MAXT * ATT_KB * (KP + VP) * sizeof(bf16_t);      // 130 KiB: one workgroup per CU
    static bool attr_set = false;
    if (!attr_set) {
        ADVGRPO_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attention_fwd_resident_kernel<HD>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          RES_MAXT * ATT_KB * (KP + VP) * (int)sizeof(bf16_t)) == hipSuccess,
                      "attention: cannot raise the dynamic LDS limit of the resident kernel");
        attr_set = true;
    }
    p.nqb = 2;
    p.nwg = 2 * p.H * B;
    hipLaunchKernelGGL(attention_fwd_resident_kernel<HD>, dim3((unsigned)p.nwg), dim3(RES_THREADS), lds, s, p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// Head dim 64, LDS-DMA variant: K/V tiles go HBM -> LDS with global_load_lds_dwordx4 into a 3-slot ring and stay
// in flight across the workgroup barrier (counted s_waitcnt vmcnt + raw s_barrier), so a tile is requested two
// iterations before it is consumed and no VGPRs are spent on staging.  The padded pitch of the register-staged
// kernel is impossible with DMA (lane-linear 1 KiB destinations), so bank conflicts are removed by swizzling the
// per-lane SOURCE chunk and un-swizzling on the read:
//   K (ds_read_b128 fragments): 16-byte chunk c of row r sits at slot c ^ (r & 7)       (as in gemm.hip)
//   V (ds_read_b64_tr_b16):     32-byte segment s of row r sits at slot s ^ ((r >> 1) & 3): the 8 rows a 32-lane
//                               service group touches land on 8 disjoint 8-bank ranges.
#ifndef ATT_NS
#define ATT_NS 3
#define ATT_WGS 3
#endif
template <bool BIAS>
__global__ __launch_bounds__(256, ATT_WGS) void attention_fwd_glds_kernel(const AttnParams p) {
    constexpr int HD = 64, NS = ATT_NS, TILE_B = ATT_KB * 128;   // 8 KiB per K or V tile
    constexpr int LOADS = 4;                                 // DMA instructions per wave per tile (2 K + 2 V)
    __shared__ __attribute__((aligned(16))) char smem[NS * 2 * TILE_B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, t = lane & 15;
    int qblk, h, b;
    xcd_local_bh(p.nqb, p.H, p.nwg, p.xcd_local, qblk, h, b);
    const int q0 = qblk * ATT_QB + wave * 32;
    const bf16_t* qp = p.q + (int64_t)b * p.bsq + h * HD;
    const bf16_t* kp = p.k + (int64_t)b * p.bsk + h * HD;
    const bf16_t* vp = p.v + (int64_t)b * p.bsv + h * HD;

    bf16x8_t qf[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        int qr = q0 + qb * 16 + t;
        qr = qr < p.Sq ? qr : p.Sq - 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[qb][ks] = *reinterpret_cast<const bf16x8_t*>(qp + (int64_t)qr * p.ldq + ks * 32 + g * 8);
    }
    // DMA: instruction j (0..7) of a tile covers rows 8j..8j+7; this wave issues j = wave and wave + 4
    const int lrow = lane >> 3, pch = lane & 7;
    const int k_src_chunk = pch ^ lrow;                                   // K: chunk ^ (row & 7)
    // per-lane source pointers of tile 0 (rows clamped to the sequence); interior tiles advance them by a uniform
    // stride, only a ragged last tile recomputes the clamped rows (64-bit integer multiplies are quarter rate)
    const bf16_t* k_src[2];
    const bf16_t* v_src[2];
    int v_chunk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rl = (wave + i * 4) * 8 + lrow;
        v_chunk[i] = ((((pch >> 1) ^ ((rl >> 1) & 3)) << 1) | (pch & 1)) * 8;
        const int r = rl < p.Skv ? rl : p.Skv - 1;
        k_src[i] = kp + (int64_t)r * p.ldk + k_src_chunk * 8;
        v_src[i] = vp + (int64_t)r * p.ldv + v_chunk[i];
    }
    const int64_t k_step = (int64_t)ATT_KB * p.ldk, v_step = (int64_t)ATT_KB * p.ldv;
    auto stage = [&](int slot, int kv0) __attribute__((always_inline)) {   // called with kv0 = 0, KB, 2 KB, ... in order
        char* base = smem + slot * 2 * TILE_B;
        const bool ragged = kv0 + ATT_KB > p.Skv;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = wave + i * 4;
            const bf16_t* ks = k_src[i];
            const bf16_t* vs = v_src[i];
            if (ragged) {
                asm volatile("; ragged tile" ::: "memory");
                int r = kv0 + j * 8 + lrow;
                r = r < p.Skv ? r : p.Skv - 1;
                ks = kp + (int64_t)r * p.ldk + k_src_chunk * 8;
                vs = vp + (int64_t)r * p.ldv + v_chunk[i];
            }
            att_dma16(ks, base + j * 1024);
            att_dma16(vs, base + TILE_B + j * 1024);
            k_src[i] += k_step;
            v_src[i] += v_step;
        }
    };
    const float sc = BIAS ? 1.0f : p.scale_log2e;
    f32x4 o[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) o[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

    int kv_end = p.Skv;
    if (p.causal) kv_end = min(kv_end, min(qblk * ATT_QB + ATT_QB, p.Sq));
    const int nt = (kv_end + ATT_KB - 1) / ATT_KB;
    // fragment read offsets (bytes)
    int k_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) k_off[ks] = t * 128 + (((ks * 4 + g) ^ (t & 7)) << 4);
    const int v_row = g * 4 + (t >> 2);                   // + (2*kpair[+1])*16
    const int v_sw = (g * 2 + (t >> 3)) & 3;              // ((row >> 1) & 3) for that row (16 | row base)
    const int v_base = v_row * 128 + (t & 3) * 8;

    // the Q fragments must have ARRIVED, in the compiler's own bookkeeping, before the first DMA is issued: otherwise it keeps
    // "s_waitcnt vmcnt(3) ... vmcnt(0)" for them in front of the loop's first MFMAs, and with the (to it invisible) DMA
    // instructions in flight behind them those waits drain the ring in every iteration
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) asm volatile("" ::"v"(qf[qb][ks]));
    stage(0, 0);
    if (NS > 2 && nt > 1) stage(1, ATT_KB);
    int slot = 0;
    for (int it = 0; it < nt; ++it) {
        const int kv0 = it * ATT_KB;
        if (NS > 2 && it + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int fill = slot == 0 ? NS - 1 : slot - 1;   // slot of tile it-1: free after this barrier
        if (it + NS - 1 < nt) stage(fill, kv0 + (NS - 1) * ATT_KB);
        const char* Kt = smem + slot * 2 * TILE_B;
        const char* Vt = Kt + TILE_B;
        slot = slot == NS - 1 ? 0 : slot + 1;

        f32x4 s[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) s[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t kf[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) kf[kb] = *reinterpret_cast<const bf16x8_t*>(Kt + kb * 2048 + k_off[ks]);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    s[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kb], qf[qb][ks], s[kb][qb], 0, 0, 0);
        }
        if constexpr (BIAS) {   // scores <- scores * scale + bias, in base-2 units (the softmax below then uses scale 1)
            asm volatile("; biased tile" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb) {
                    const int qi = min(q0 + qb * 16 + t, p.Sq - 1);
                    const float* brow = p.bias + ((int64_t)h * p.Sq + qi) * p.Skv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = min(kv0 + kb * 16 + g * 4 + r, p.Skv - 1);
                        s[kb][qb][r] = s[kb][qb][r] * p.scale_log2e + brow[key] * 1.4426950408889634f;
                    }
                }
        }
        const bool edge = (kv0 + ATT_KB > p.Skv) || p.causal;
        if (edge) {   // ragged last tile / causal only: a real (wave-uniform) branch, interior tiles pay nothing
            asm volatile("; masked tile" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kv0 + kb * 16 + g * 4 + r;
                        const int qi = q0 + qb * 16 + t;
                        if (key >= p.Skv || (p.causal && key > qi)) s[kb][qb][r] = -INFINITY;
                    }
        }
        bf16x8_t pf[2][2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            float mx = -INFINITY;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kb][qb][r]);
            mx = xor32_max(xor16_max(mx));
            const float m_new = fmaxf(m_run[qb], mx * sc);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_use);
            m_run[qb] = m_new;
            // exp2(s * scale - m) two scores per v_pk_fma_f32; row sums on v_pk_add_f32
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 sc2 = {sc, sc}, nm2 = {-m_use, -m_use};
            f32x2 ps2 = {0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    f32x2 a = {s[kb][qb][r], s[kb][qb][r + 1]};
                    a = __builtin_elementwise_fma(a, sc2, nm2);
                    a[0] = __builtin_amdgcn_exp2f(a[0]);
                    a[1] = __builtin_amdgcn_exp2f(a[1]);
                    s[kb][qb][r] = a[0];
                    s[kb][qb][r + 1] = a[1];
                    ps2 += a;
                }
            l_run[qb] = l_run[qb] * alpha + (ps2[0] + ps2[1]);
            // the running maximum settles after the first few tiles: skip the rescale of O when no lane's moved
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int db = 0; db < 4; ++db) o[db][qb] *= alpha;
            }
#pragma unroll
            for (int kpair = 0; kpair < 2; ++kpair) {
                bf16x8_t f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f[r] = (__bf16)s[2 * kpair][qb][r];
                    f[4 + r] = (__bf16)s[2 * kpair + 1][qb][r];
                }
                pf[qb][kpair] = f;
            }
        }
#pragma unroll
        for (int kpair = 0; kpair < 2; ++kpair) {
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const char* a0 = Vt + (2 * kpair) * 2048 + v_base + ((db ^ v_sw) << 5);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (__attribute__((address_space(3))) s16x4*)(a0 + 2048));
                typedef __attribute__((ext_vector_type(8))) short s16x8;
                const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                const bf16x8_t vf = __builtin_bit_cast(bf16x8_t, both);
#pragma unroll
                for (int qb = 0; qb < 2; ++qb)
                    o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb][kpair], o[db][qb], 0, 0, 0);
            }
        }
    }
    // Output: in the accumulator layout a lane owns 4 dims of one row and a quarter-wave spans 16 rows, i.e. 64
    // scattered 8-byte stores per instruction.  Each wave bounces its 32 x 64 bf16 tile through LDS (the ring slot
    // of tile nt-2: every wave passed the last barrier, so nobody reads it and no DMA targets it any more) and
    // stores whole 128-byte rows, 16 bytes per lane.
    char* ob = smem + (slot == 0 ? NS - 2 : (slot == 1 ? NS - 1 : slot - 2)) * 2 * TILE_B + wave * 4096;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
        float l = l_run[qb];
        l = xor32_add(xor16_add(l));
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        const int qi = q0 + qb * 16 + t;
        if (p.lse && g == 0 && qi < p.Sq) p.lse[((int64_t)b * p.H + h) * p.Sq + qi] = m_run[qb] + __builtin_amdgcn_logf(l);
        const int r = qb * 16 + t;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
            const f32x4 v = o[db][qb] * inv;
            uint2 pk;
            pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
            pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
            *reinterpret_cast<uint2*>(ob + r * 128 + (((db * 2 + (g >> 1)) ^ (r & 7)) << 4) + (g & 1) * 8) = pk;
        }
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int r = ps * 8 + (lane >> 3), c = lane & 7;
        const uint4 q = *reinterpret_cast<const uint4*>(ob + r * 128 + ((c ^ (r & 7)) << 4));
        const int qi = q0 + r;
        if (qi < p.Sq) *reinterpret_cast<uint4*>(p.o + (int64_t)b * p.bso + (int64_t)qi * p.ldo + h * HD + c * 8) = q;
    }
}

int attention_fwd(const AttnParams& p_in, int B, int head_dim, hipStream_t s) {
    AttnParams p = p_in;
    ADVGRPO_CHECK(head_dim == 64 || head_dim == 80 || head_dim == 128, "attention: head_dim %d not supported (64, 80, 128)", head_dim);
    ADVGRPO_CHECK(p.q && p.k && p.v && p.o, "attention: null pointer");
    ADVGRPO_CHECK(p.Sq > 0 && p.Skv > 0 && p.H > 0 && B > 0, "attention: bad shape");
    ADVGRPO_CHECK(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldv % 8 == 0 && p.ldo % 4 == 0,
                  "attention: row pitches must keep 16-byte (q,k,v) / 8-byte (o) alignment");
    ADVGRPO_CHECK(!p.bias || head_dim == 64, "attention: the score bias is implemented for head dim 64");
    int use_glds = 1, xcd_local = 1, use_pipe = 1, use_resident = 1;
#ifdef ADVGRPO_EXPERIMENTS   // A/B knobs of the experiments build only (the product library reads no environment)
    { const char* e = getenv("ADVGRPO_ATTN_REGSTAGE"); if (e && atoi(e)) use_glds = 0; }
    { const char* e = getenv("ADVGRPO_ATTN_NO_XCD"); if (e && atoi(e)) xcd_local = 0; }
    { const char* e = getenv("ADVGRPO_ATTN_NO_PIPE"); if (e && atoi(e)) use_pipe = 0; }
    { const char* e = getenv("ADVGRPO_ATTN_NO_RESIDENT"); if (e && atoi(e)) use_resident = 0; }
#endif
    p.nqb = (p.Sq + ATT_QB - 1) / ATT_QB;
    const int64_t nwg = (int64_t)p.nqb * p.H * B;
    ADVGRPO_CHECK(nwg < (1ll << 31), "attention: grid too large");
    p.nwg = (int)nwg;
    p.xcd_local = xcd_local;
    dim3 grid((unsigned)nwg);
    const bool o16 = p.ldo % 8 == 0 && p.bso % 8 == 0 && (reinterpret_cast<uintptr_t>(p.o) & 15) == 0;   // 16-byte row stores
    if (head_dim == 128) {      // the Qwen-Image MMDiT's joint attention (attention_d128.hip)
        ADVGRPO_CHECK(o16 && !p.bias && !p.causal, "attention: head dim 128 is implemented without mask / bias and for 16-byte aligned output rows");
        return attention_fwd_d128_launch(p, B, s);
    }
    ADVGRPO_CHECK(!p.bias || (use_glds && o16), "attention: the score bias needs the LDS-DMA kernel (16-byte aligned output rows)");
    // plain head-dim-64 attention (MMDiT joint / second attention, DINOv2): the software-pipelined kernel of attention_pipe.hip
    if (head_dim == 64 && use_glds && o16 && !p.bias && !p.causal && use_pipe) return attention_fwd_pipe_launch(p, s);
    if (head_dim == 64 && use_glds && o16 && p.bias) hipLaunchKernelGGL(attention_fwd_glds_kernel<true>, grid, dim3(256), 0, s, p);
    else if (head_dim == 64 && use_glds && o16) hipLaunchKernelGGL(attention_fwd_glds_kernel<false>, grid, dim3(256), 0, s, p);
    else if (head_dim == 64) hipLaunchKernelGGL(attention_fwd_kernel<64>, grid, dim3(256), 0, s, p);
    else if (use_resident && !p.causal && !p.bias && p.Skv <= RES_MAXT * ATT_KB && p.Sq <= RES_MAX_SQ && (int64_t)2 * p.H * B < (1ll << 31))
        return attention_fwd_resident_launch(p, B, s);       // the CLIP ViT-H vision tower (257 tokens): K / V of a head resident in LDS
    else hipLaunchKernelGGL(attention_fwd_kernel<80>, grid, dim3(256), 0, s, p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

}  // namespace advgrpo

using namespace advgrpo;

extern "C" int advgrpo_attention_fwd(const void* q, const void* k, const void* v, void* o, int64_t ldq, int64_t ldk,
                                     int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk, int64_t bsv, int64_t bso,
                                     int B, int H, int Sq, int Skv, int head_dim, float scale, int causal,
                                     float* lse, void* stream) {
    AttnParams p{};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.bsq = bsq; p.bsk = bsk; p.bsv = bsv; p.bso = bso;
    p.H = H; p.Sq = Sq; p.Skv = Skv;
    p.scale_log2e = scale * 1.4426950408889634f;
    p.causal = causal;
    p.lse = lse;
    return attention_fwd(p, B, head_dim, as_stream(stream));
}

extern "C" int advgrpo_attention_fallback_count(long long* count_host, int reset) {
    ADVGRPO_CHECK(count_host, "attention_fallback_count: null pointer");
    unsigned long long a = 0, b = 0;
    ADVGRPO_CHECK(hipDeviceSynchronize() == hipSuccess && attention_pipe_fallbacks(&a, reset) == 0 && attention_d128_fallbacks(&b, reset) == 0,
                  "attention_fallback_count: cannot read the device counters");
    *count_host = (long long)(a + b);
    return 0;
}

/* same + additive score bias [H,Sq,Skv] f32 shared by the batch: softmax(q k^T * scale + bias) v (T5's relative position
 * bias; T5 itself uses scale = 1).  Text encoders of encode_prompt, train_dreambooth_lora_sd3.py:98-144. */
extern "C" int advgrpo_attention_fwd_bias(const void* q, const void* k, const void* v, void* o, int64_t ldq, int64_t ldk,
                                          int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk, int64_t bsv, int64_t bso,
                                          int B, int H, int Sq, int Skv, int head_dim, float scale, int causal,
                                          const float* bias, void* stream) {
    ADVGRPO_CHECK(bias, "attention_fwd_bias: null bias");
    AttnParams p{};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.bsq = bsq; p.bsk = bsk; p.bsv = bsv; p.bso = bso;
    p.H = H; p.Sq = Sq; p.Skv = Skv;
    p.scale_log2e = scale * 1.4426950408889634f;
    p.causal = causal;
    p.bias = bias;
    return attention_fwd(p, B, head_dim, as_stream(stream));
}
