// attention_d128.hip -- flash attention forward for head dim 128 on gfx950 (bf16 in/out, f32 softmax).
//
// Serves the joint text+image attention of the Qwen-Image MMDiT (BASELINE config 5: 24 heads x 128, 4096 image + text
// tokens at 1024^2; the reference names the model only as a to-do, README.md:75 / config/grpo.py:324,330 -- the call site
// it would replace is the transformer call of sd3_pipeline_with_logprob_fast.py:630-637 -> F.scaled_dot_product_attention).
//
// At head dim 128 a 32-query x 64-key tile costs 32 v_mfma_f32_32x32x16_bf16 (16 for K Q^T, 16 for P V) against the same
// ~75 VALU instructions of softmax the head-dim-64 kernel pays per 16 MFMAs (attention_pipe.hip), so the structure is
// built around the matrix pipe, the way MI355X_MICROARCH.md "Two waves per SIMD" describes:
//   * workgroup = 8 waves = 256 queries of one (batch, head), ONE workgroup per CU: a 64-key K or V tile is 16 KiB and is
//     shared by eight waves (0.004 L2->LDS bytes per flop, half of what 4-wave workgroups would move);
//   * a tile is 32 MFMA slots per wave and ONE barrier:
//       slots  0..15  the P V products of tile j - 1, with the softmax of tile j on the VALU in their shadow
//       slots 16..31  the K Q^T products of tile j + 1, with the four LDS-DMA requests of the wave in their shadow
//     one score buffer (the second half overwrites what the first consumed), two probability buffers by name.  (Built and
//     measured as well, D128_VAR bit 1: the two wave groups -- waves w and w + 4 share a SIMD -- ONE HALF APART, a barrier per
//     half, so that a wave in its VALU-heavy half sits beside a partner in the MFMA-only one: 2 % slower than lock-step,
//     3.31 vs 3.25 ms at B = 16, 24 heads, S = 4224; as were s_setprio around the MFMA-only half, static priority for the second
//     group, operand reads 3 or 4 slots ahead and an unpinned first half: all within 1 % -- LABNOTES.md section 6, round 4.)
//   * everything that addresses LDS is a compile-time constant: the tile loop is unrolled over the ring period (4), so a
//     fragment read is "per-lane base + immediate" and the DMA destinations are literals (the first version computed the
//     ring slot at run time: 150 VALU + 61 SALU instructions per 32 MFMAs, 47 spilled SGPRs; the instruction stream of the
//     two waves of a SIMD, not the matrix pipe, set the pace);
//   * the LDS operand of slot s + 2 is read in slot s, across the half boundary (the K tile landed before the tile began);
//   * K and V tiles arrive by hand-written LDS-DMA (SGPR base + 32-bit lane offset) into two 4-slot rings; the bundle
//     {K(t+1), V(t)} is requested in the second half of tile t - 2 and waited for (counted vmcnt, then the barrier) at
//     the end of tile t - 1;
//   * layouts as in attention_pipe.hip: S^T = K Q^T so a lane holds 32 scores of ONE query; P stays in registers as the
//     B operand of P V; V^T through ds_read_b64_tr_b16; softmax scale in f32; probabilities relative to the row maximum
//     of the first tile, never rescaled, with the same overflow-detecting fallback (running maximum per tile);
//   * LDS images (256-byte rows): K chunk c of row r at slot c ^ (r & 15) -- the 16 rows of a ds_read_b128 service
//     group cover all sixteen 16-byte slots; V 64-byte quarter q of row r at q ^ (r & 3) -- the 4 key rows of a
//     transposed read land on the four quarters of the bank row (PMC: 0.8 % bank-conflict cycles).
#include "attention.hpp"
#include "gemm_device.hpp"

// timing experiments (scripts/ablate_d128.sh builds one library per value; the product build has 0): 1 = wave groups one half
// apart (a barrier per half), 2 = s_setprio 1 around the K Q^T half, 4 = static priority for the second wave group,
// 8 = first half left to the compiler's scheduler (no slot pins), 128 = DMA requests in a clump at the top of the tile;
// results WRONG with: 16 = no DMA, 32 = no LDS fragment reads
#ifndef D128_VAR
#define D128_VAR 0
#endif
#ifndef D128_AHEAD
#define D128_AHEAD 2          // the LDS operand of MFMA slot s + D128_AHEAD is read in slot s
#endif

namespace advgrpo {

__device__ unsigned long long g_d128_fallbacks = 0;     // workgroups that took the running-maximum fallback (see attention_pipe.hip)

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int D128_QB = 256;                  // queries per workgroup
constexpr int D128_TILE = ATT_KB * 256;       // 16 KiB per K or V tile
constexpr int D128_SLOTS = 4;
constexpr int D128_VRING = D128_SLOTS * D128_TILE;          // byte offset of the V ring
constexpr int D128_LDS = 2 * D128_SLOTS * D128_TILE + 64;   // 128 KiB + the fallback flags
constexpr bool D128_STAGGER = (D128_VAR & 1) != 0;

__device__ __forceinline__ float d128_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// (through the compiler, not inline asm: see attention_pipe.hip -- the hazard recogniser only pads instructions it knows)
__device__ __forceinline__ uint32_t d128_cvt_pk(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
#if (D128_VAR & 8)
#define D128_SB()
#define D128_PIN(x)
#else
#define D128_SB() __builtin_amdgcn_sched_barrier(0)
#define D128_PIN(x) asm volatile("" : "+v"(x))
#endif
#define D128_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

template <int N>
using d128_int = std::integral_constant<int, N>;

}  // namespace

__global__ __launch_bounds__(512, 2) void attention_fwd_d128_kernel(const AttnParams p) {
    constexpr int HD = 128;
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // K ring (4 x 16 KiB) | V ring (4 x 16 KiB) | 8 flag words
    int* const wg_flag = reinterpret_cast<int*>(smem + 2 * D128_SLOTS * D128_TILE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                                         // wave group (experiment D128_VAR & 1: group 1 runs one half behind)
    const int ql = lane & 31, hi = lane >> 5;
    int qblk, h, b;
    xcd_local_bh(p.nqb, p.H, p.nwg, p.xcd_local, qblk, h, b);
    const int q0 = qblk * D128_QB + wave * 32;
    const bf16_t* qp = p.q + (int64_t)b * p.bsq + h * HD;
    const bf16_t* kp = p.k + (int64_t)b * p.bsk + h * HD;
    const bf16_t* vp = p.v + (int64_t)b * p.bsv + h * HD;
    const int nt = (p.Skv + ATT_KB - 1) / ATT_KB;

    // ---- Q fragments (B operand of K Q^T: lane = query, 8 consecutive d at ks*16 + hi*8)
    bf16x8_t qf[8];
    {
        int qr = q0 + ql;
        qr = qr < p.Sq ? qr : p.Sq - 1;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + (int64_t)qr * p.ldq + ks * 16 + hi * 8);
    }

    // ---- DMA sources.  Instruction jj (0..15) of a tile fills LDS rows 4 jj .. 4 jj + 3 (1 KiB); this wave issues jj = wave and
    // wave + 8.  Per-lane byte offsets inside a tile: interior tiles, and the (possibly ragged) LAST tile, whose rows past the
    // end of the sequence are clamped to its last row (their scores are masked) -- the only ragged tile there is, so its
    // offsets are known up front and a request picks between the two sets with a wave-uniform select.
    const int lrow = lane >> 4, slot16 = lane & 15;
    uint32_t k_lo[2], v_lo[2], k_last[2], v_last[2];
    {
        const int rmax = p.Skv - 1 - (nt - 1) * ATT_KB;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int R = (wave + 8 * i) * 4 + lrow;
            const int r = min(R, rmax);
            const uint32_t kc = (uint32_t)((slot16 ^ (R & 15)) * 16);
            const uint32_t vc = (uint32_t)((((((slot16 >> 2) ^ (R & 3))) << 2) | (slot16 & 3)) * 16);
            k_lo[i] = (uint32_t)R * (uint32_t)(p.ldk * 2) + kc;
            v_lo[i] = (uint32_t)R * (uint32_t)(p.ldv * 2) + vc;
            k_last[i] = (uint32_t)r * (uint32_t)(p.ldk * 2) + kc;
            v_last[i] = (uint32_t)r * (uint32_t)(p.ldv * 2) + vc;
        }
    }
    const int64_t k_step = (int64_t)ATT_KB * p.ldk * 2, v_step = (int64_t)ATT_KB * p.ldv * 2;
    const uint32_t lds0 = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)(smem)) + wave * 1024;
    auto dma = [&](const char* base, uint32_t off, uint32_t lds) __attribute__((always_inline)) {
        if constexpr ((D128_VAR & 16) != 0) return;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory");
    };
    // piece i (0, 1) of K tile t into ring slot `slot` / of V tile t into its own slot; slot: compile-time
    auto dma_k = [&](int t, auto slot_tag, auto i_tag) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_tag)::value & (D128_SLOTS - 1), i = decltype(i_tag)::value;
        const char* base = reinterpret_cast<const char*>(kp) + (int64_t)t * k_step;
        dma(base, t == nt - 1 ? k_last[i] : k_lo[i], lds0 + slot * D128_TILE + i * 8192);
    };
    auto dma_v = [&](int t, auto slot_tag, auto i_tag) __attribute__((always_inline)) {
        constexpr int slot = decltype(slot_tag)::value & (D128_SLOTS - 1), i = decltype(i_tag)::value;
        const char* base = reinterpret_cast<const char*>(vp) + (int64_t)t * v_step;
        dma(base, t == nt - 1 ? v_last[i] : v_lo[i], lds0 + D128_VRING + slot * D128_TILE + i * 8192);
    };
    // bundle t = {K(t+1), V(t)}, t = T4 (mod 4): always 4 DMA instructions per wave, so the counted waits are constants -- the
    // bundle of the last tile requests that tile's K rows once more, into the slot K(nt) would have used (dead), never read.
    // piece = 0..3: K piece 0, K piece 1, V piece 0, V piece 1
    auto bundle_piece = [&](int t, auto t4_tag, auto piece_tag) __attribute__((always_inline)) {
        constexpr int T4 = decltype(t4_tag)::value & 3, piece = decltype(piece_tag)::value;
        if constexpr (piece < 2) dma_k(min(t + 1, nt - 1), d128_int<T4 + 1>{}, d128_int<piece>{});
        else dma_v(t, d128_int<T4>{}, d128_int<piece - 2>{});
    };
    auto bundle = [&](int t, auto t4_tag) __attribute__((always_inline)) {
        bundle_piece(t, t4_tag, d128_int<0>{}); bundle_piece(t, t4_tag, d128_int<1>{});
        bundle_piece(t, t4_tag, d128_int<2>{}); bundle_piece(t, t4_tag, d128_int<3>{});
    };

    // ---- fragment read bases (bytes inside a tile; the ring slot and the block inside the tile are immediates)
    //   K: key row kb*32 + ql, 16-byte chunk 2 ks + hi at slot ^ (row & 15)
    //   V^T: 16-lane group g16 reads the 4 (keys 4 hi ..) x 16 (d) block of d half g16 & 1; lane l16 supplies key row l16 >> 2,
    //        4 consecutive d at (l16 & 3) * 4; the 64-byte quarter db of a row sits at db ^ (row & 3)
    int k_off[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) k_off[ks] = ql * 256 + (((2 * ks + hi) ^ (ql & 15)) << 4);
    const int vrow = 4 * hi + ((lane & 15) >> 2);
    int v_off[4];
#pragma unroll
    for (int db = 0; db < 4; ++db)      // (the V ring's base is part of the per-lane offset: the immediates then stay below the 64 KiB of a DS offset field)
        v_off[db] = D128_VRING + vrow * 256 + ((db ^ (vrow & 3)) << 6) + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

    typedef __attribute__((ext_vector_type(8))) short s16x8;
    auto kfrag = [&](int imm, int ks) __attribute__((always_inline)) {          // imm = slot * TILE + kb * 8192
        if constexpr ((D128_VAR & 32) != 0) return qf[ks];
        return *reinterpret_cast<const bf16x8_t*>(smem + k_off[ks] + imm);
    };
    auto vfrag = [&](int imm, int db) __attribute__((always_inline)) {          // imm = slot * TILE + kk * 4096
        if constexpr ((D128_VAR & 32) != 0) return qf[db];
        const char* a0 = smem + v_off[db] + imm;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
        const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 2048));
        const s16x8 both = __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8_t, both);
    };

    f32x16 o[4];          // O^T accumulators: d block db, lane (query = lane & 31, hi): d = db*32 + 8*(r>>2) + 4*hi + (r&3)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int db = 0; db < 4; ++db) o[db] = zero16;
    f32x2 lsum[2] = {{0.f, 0.f}, {0.f, 0.f}};
    float m_ref = 0.f;
    const f32x2 c2 = {p.scale_log2e, p.scale_log2e};
    f32x2 nm2 = {0.f, 0.f};

    // the Q fragments must have arrived, in the compiler's own bookkeeping, before the first DMA is issued
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" ::"v"(qf[ks]));

    // ---- prologue: K(0) | bundle 0 = {K(1), V(0)} | bundle 1 = {K(2), V(1)} ; S(0) = K(0) Q^T ; reference maximum
    dma_k(0, d128_int<0>{}, d128_int<0>{});
    dma_k(0, d128_int<0>{}, d128_int<1>{});
    bundle(0, d128_int<0>{});
    if (nt > 1) bundle(1, d128_int<1>{});
    if (nt > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x16 s[2];
    u32x4 pA[4], pB[4];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) s[kb] = D128_MFMA(kfrag(kb * 8192, ks), qf[ks], ks == 0 ? zero16 : s[kb]);
    auto row_max = [&](const f32x16 (&sc)[2]) __attribute__((always_inline)) {
        float m0 = d128_max3(sc[0][0], sc[0][1], sc[0][2]);
        float m1 = d128_max3(sc[0][3], sc[0][4], sc[0][5]);
        float m2 = d128_max3(sc[1][0], sc[1][1], sc[1][2]);
        float m3 = d128_max3(sc[1][3], sc[1][4], sc[1][5]);
        m0 = d128_max3(m0, sc[0][6], sc[0][7]);
        m1 = d128_max3(m1, sc[0][8], sc[0][9]);
        m2 = d128_max3(m2, sc[1][6], sc[1][7]);
        m3 = d128_max3(m3, sc[1][8], sc[1][9]);
        m0 = d128_max3(m0, sc[0][10], sc[0][11]);
        m1 = d128_max3(m1, sc[0][12], sc[0][13]);
        m2 = d128_max3(m2, sc[1][10], sc[1][11]);
        m3 = d128_max3(m3, sc[1][12], sc[1][13]);
        m0 = d128_max3(m0, sc[0][14], sc[0][15]);
        m2 = d128_max3(m2, sc[1][14], sc[1][15]);
        m0 = d128_max3(m0, m1, m2);
        float a = d128_max3(m0, m3, m3), bb = a;
        ADVGRPO_SWAP32(a, bb);
        return d128_max3(a, bb, bb);
    };
    auto mask_tail = [&](f32x16 (&sc)[2], int kv0) __attribute__((always_inline)) {
        if (kv0 + ATT_KB > p.Skv) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kb * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
                    if (key >= p.Skv) sc[kb][r] = -INFINITY;
                }
        }
    };
    if (nt == 1) mask_tail(s, 0);
    m_ref = row_max(s) * p.scale_log2e;       // (tile 0 always holds a real key: finite)
    nm2 = f32x2{-m_ref, -m_ref};
    // bundle 0 (read from the next half on) must have landed before the barrier that closes this "second half of tile -1"
    if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (D128_STAGGER && grp == 1) __builtin_amdgcn_s_barrier();      // stagger: group 1 runs one half behind
    if ((D128_VAR & 4) != 0 && grp == 1) __builtin_amdgcn_s_setprio(1);

    auto pair_x = [&](int i) __attribute__((always_inline)) {
        const f32x2 s2 = {s[i >> 3][(2 * i) & 15], s[i >> 3][((2 * i) & 15) + 1]};
        return __builtin_elementwise_fma(s2, c2, nm2);
    };
    auto pair_exp = [&](f32x2 x) __attribute__((always_inline)) {
        f32x2 e;
        e[0] = __builtin_amdgcn_exp2f(x[0]);
        e[1] = __builtin_amdgcn_exp2f(x[1]);
        return e;
    };
    // One tile j = J4 (mod 4).  Slots 0..15: P V of tile j - 1 (probabilities pp, V ring slot J4 - 1: key step sl >> 2, d block
    // sl & 3) beside the softmax of tile j (scores s -> pn; pair i: multiply-add in slot i, exponentials in slot i + 1, bf16 pack
    // + row sum in slot i + 2).  Slots 16..31: K Q^T of tile j + 1 (K ring slot J4 + 1: key block sl & 1, d step (sl - 16) >> 1)
    // beside the four DMA requests of bundle j + 2 (after slots 18, 22, 26, 30).  The LDS operand of slot sl + 2 is read in slot sl.
    auto tile = [&](auto j4_tag, auto first_tag, int j, const u32x4 (&pp)[4], u32x4 (&pn)[4]) __attribute__((always_inline)) {
        constexpr int J4 = decltype(j4_tag)::value & 3;
        constexpr bool HAVE_PV = !decltype(first_tag)::value;
        constexpr int VRD = ((J4 + 3) & 3) * D128_TILE, KRD = ((J4 + 1) & 3) * D128_TILE;
        const bool more_dma = j + 2 < nt;       // (the last tile still multiplies a K tile: bundle nt - 1's copy of K(nt - 1), result unused)
        if ((D128_VAR & 128) != 0 && more_dma) bundle(j + 2, d128_int<J4 + 2>{});
        if (j == nt - 1) mask_tail(s, j * ATT_KB);
        auto operand = [&](auto s_tag) __attribute__((always_inline)) {
            constexpr int sl = decltype(s_tag)::value;
            if constexpr (sl < 16) return vfrag(VRD + (sl >> 2) * 4096, sl & 3);
            else return kfrag(KRD + (sl & 1) * 8192, (sl - 16) >> 1);
        };
        auto has_operand = [](int sl) constexpr { return (sl >= 16 && sl < 32) || (HAVE_PV && sl >= 0 && sl < 16); };
        constexpr int AH = D128_AHEAD, NA = D128_AHEAD + 1;
        bf16x8_t a[NA];
        static_for<AH>([&](auto i_tag) __attribute__((always_inline)) {
            constexpr int i = decltype(i_tag)::value;
            if constexpr (has_operand(i)) a[i] = operand(d128_int<i>{});
        });
        f32x2 x[2], e[2];
        auto slot = [&](auto s_tag) __attribute__((always_inline)) {
            constexpr int sl = decltype(s_tag)::value;
            if constexpr (sl == 16) {                                   // ---- end of the first half
                D128_SB();
                if constexpr (D128_STAGGER) {
                    __builtin_amdgcn_s_barrier();
                    // (staggered groups: the other group confirms the arrival of ITS pieces of this K tile only at this barrier,
                    // so the first two operands cannot be read ahead of it)
                    static_for<AH>([&](auto i_tag) __attribute__((always_inline)) {
                        constexpr int i = 16 + decltype(i_tag)::value;
                        a[i % NA] = operand(d128_int<i>{});
                    });
                }
                if constexpr ((D128_VAR & 2) != 0) __builtin_amdgcn_s_setprio(1);
            }
            D128_SB();
            if constexpr (sl < 16) {
                if constexpr (HAVE_PV) o[sl & 3] = D128_MFMA(a[sl % NA], __builtin_bit_cast(bf16x8_t, pp[sl >> 2]), o[sl & 3]);
            } else if constexpr (sl < 32) {
                s[sl & 1] = D128_MFMA(a[sl % NA], qf[(sl - 16) >> 1], sl < 18 ? zero16 : s[sl & 1]);
            }
            if constexpr (has_operand(sl + AH) && !(D128_STAGGER && sl < 16 && sl + AH >= 16)) {
                a[(sl + AH) % NA] = operand(d128_int<(sl + AH) % 32>{});
            }
            if constexpr (sl >= 2 && sl - 2 < 16) {
                pn[(sl - 2) >> 2][(sl - 2) & 3] = d128_cvt_pk(e[sl & 1][0], e[sl & 1][1]);
                D128_PIN(pn[(sl - 2) >> 2][(sl - 2) & 3]);
                lsum[sl & 1] += e[sl & 1];
                D128_PIN(lsum[sl & 1]);
            }
            if constexpr (sl >= 1 && sl - 1 < 16) { e[(sl - 1) & 1] = pair_exp(x[(sl - 1) & 1]); D128_PIN(e[(sl - 1) & 1]); }
            if constexpr (sl < 16) { x[sl & 1] = pair_x(sl); D128_PIN(x[sl & 1]); }
            if constexpr ((D128_VAR & 128) == 0 && sl >= 18 && sl < 32 && ((sl - 18) & 3) == 0) {
                if (more_dma) bundle_piece(j + 2, d128_int<J4 + 2>{}, d128_int<(sl - 18) / 4>{});
            }
        };
        static_for<32>(slot);
        if constexpr ((D128_VAR & 2) != 0) __builtin_amdgcn_s_setprio(0);
        // bundle j + 1 (read from the next half on) must have landed: everything but the bundle requested in this tile
        if (more_dma) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        D128_SB();
        __builtin_amdgcn_s_barrier();                                  // ---- end of the tile
    };
    auto tile_pv = [&](int imm, const u32x4 (&pp)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8_t pb = __builtin_bit_cast(bf16x8_t, pp[kk]);
#pragma unroll
            for (int db = 0; db < 4; ++db) o[db] = D128_MFMA(vfrag(imm + kk * 4096, db), pb, o[db]);
        }
    };
    typedef std::integral_constant<bool, true> first_t;
    typedef std::integral_constant<bool, false> steady_t;
    // P(j) lives in pA for even j, pB for odd j; the body is unrolled over the ring period
    tile(d128_int<0>{}, first_t{}, 0, pB, pA);
    int j = 1;
    for (; j + 4 <= nt; j += 4) {
        tile(d128_int<1>{}, steady_t{}, j, pA, pB);
        tile(d128_int<2>{}, steady_t{}, j + 1, pB, pA);
        tile(d128_int<3>{}, steady_t{}, j + 2, pA, pB);
        tile(d128_int<0>{}, steady_t{}, j + 3, pB, pA);
    }
    if (j < nt) { tile(d128_int<1>{}, steady_t{}, j, pA, pB); ++j; }
    if (j < nt) { tile(d128_int<2>{}, steady_t{}, j, pB, pA); ++j; }
    if (j < nt) { tile(d128_int<3>{}, steady_t{}, j, pA, pB); ++j; }
    // the last tile's P V (its probabilities: pA for an even tile index, pB for an odd one; V ring slot (nt - 1) & 3)
    switch ((nt - 1) & 3) {
        case 0: tile_pv(0 * D128_TILE, pA); break;
        case 1: tile_pv(1 * D128_TILE, pB); break;
        case 2: tile_pv(2 * D128_TILE, pA); break;
        default: tile_pv(3 * D128_TILE, pB); break;
    }
    if (D128_STAGGER && grp == 0) __builtin_amdgcn_s_barrier();      // matches group 1's stagger barrier
    if ((D128_VAR & 4) != 0) __builtin_amdgcn_s_setprio(0);

    // ---- the window check (attention_pipe.hip): a row sum that is zero, huge or not finite sends the workgroup through
    // the classic running-maximum loop
    float l = xor32_add((lsum[0][0] + lsum[0][1]) + (lsum[1][0] + lsum[1][1]));
    {
        const bool bad = !(l > 1e-30f && l < 1e30f);
        const bool wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
        if (lane == 0) wg_flag[wave] = wave_bad ? 1 : 0;
        __syncthreads();                             // (also: every wave is past its last ring read)
        int any_bad = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) any_bad |= wg_flag[w];
        if (any_bad) {
            asm volatile("; fallback: running maximum per tile" ::: "memory");
            if (tid == 0) atomicAdd(&g_d128_fallbacks, 1ull);
#pragma unroll
            for (int db = 0; db < 4; ++db) o[db] = zero16;
            float m_run = -INFINITY, l_run = 0.f;
            for (int t = 0; t < nt; ++t) {
                __builtin_amdgcn_s_barrier();
                dma_k(t, d128_int<0>{}, d128_int<0>{}); dma_k(t, d128_int<0>{}, d128_int<1>{});
                dma_v(t, d128_int<0>{}, d128_int<0>{}); dma_v(t, d128_int<0>{}, d128_int<1>{});
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) s[kb] = D128_MFMA(kfrag(kb * 8192, ks), qf[ks], ks == 0 ? zero16 : s[kb]);
                mask_tail(s, t * ATT_KB);
                const float m_new = fmaxf(m_run, row_max(s) * p.scale_log2e);
                const float f = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                nm2 = f32x2{-m_new, -m_new};
#pragma unroll
                for (int db = 0; db < 4; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[db][r] *= f;
                l_run *= f;
                lsum[0] = f32x2{0.f, 0.f}; lsum[1] = f32x2{0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x2 e = pair_exp(pair_x(i));
                    pA[i >> 2][i & 3] = d128_cvt_pk(e[0], e[1]);
                    lsum[i & 1] += e;
                }
                tile_pv(0, pA);
                l_run += xor32_add((lsum[0][0] + lsum[0][1]) + (lsum[1][0] + lsum[1][1]));
            }
            m_ref = m_run;
            l = l_run;
            __builtin_amdgcn_s_barrier();            // the epilogue reuses the K ring
        }
    }

    // ---- epilogue: normalise, bounce the wave's 32 x 128 bf16 tile through its private 8 KiB of the (dead) K ring, store
    // whole 256-byte rows, 16 bytes per lane
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    const int qi = q0 + ql;
    if (p.lse && hi == 0 && qi < p.Sq) p.lse[((int64_t)b * p.H + h) * p.Sq + qi] = m_ref + __builtin_amdgcn_logf(l);
    char* ob = smem + wave * 8192;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 pk;
            pk.x = d128_cvt_pk(o[db][4 * i] * inv, o[db][4 * i + 1] * inv);
            pk.y = d128_cvt_pk(o[db][4 * i + 2] * inv, o[db][4 * i + 3] * inv);
            // d = db*32 + 8 i + 4 hi .. + 3  ->  16-byte chunk db*4 + i, 8-byte half hi
            *reinterpret_cast<uint2*>(ob + ql * 256 + (((db * 4 + i) ^ (ql & 15)) << 4) + hi * 8) = pk;
        }
    // (each wave reads back only what it wrote itself: the LDS accesses of one wave are ordered)
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        const int r = ps * 4 + (lane >> 4), c = lane & 15;
        const uint4 v = *reinterpret_cast<const uint4*>(ob + r * 256 + ((c ^ (r & 15)) << 4));
        const int qo = q0 + r;
        if (qo < p.Sq) *reinterpret_cast<uint4*>(p.o + (int64_t)b * p.bso + (int64_t)qo * p.ldo + h * HD + c * 8) = v;
    }
}

int attention_d128_fallbacks(unsigned long long* out, int reset) {
    unsigned long long v = 0, z = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_d128_fallbacks), sizeof(v)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_d128_fallbacks), &z, sizeof(z)) != hipSuccess) return -1;
    *out = v;
    return 0;
}

int attention_fwd_d128_launch(const AttnParams& p_in, int B, hipStream_t s) {
    AttnParams p = p_in;
    p.nqb = (p.Sq + D128_QB - 1) / D128_QB;
    const int64_t nwg = (int64_t)p.nqb * p.H * B;
    ADVGRPO_CHECK(nwg < (1ll << 31), "attention: grid too large");
    p.nwg = (int)nwg;
    // 32-bit lane offsets inside a tile: 64 rows x pitch
    ADVGRPO_CHECK((int64_t)ATT_KB * p.ldk * 2 < (1ll << 31) && (int64_t)ATT_KB * p.ldv * 2 < (1ll << 31), "attention: row pitch too large");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_fwd_d128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, D128_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL(attention_fwd_d128_kernel, dim3((unsigned)p.nwg), dim3(512), D128_LDS, s, p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

}  // namespace advgrpo
