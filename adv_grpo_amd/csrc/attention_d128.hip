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
//   * a tile is two PHASES per wave, one barrier each:
//       P1(j)  softmax of tile j on the VALU, in the shadow of the 16 P V MFMAs of tile j - 1
//       P2(j)  the 16 K Q^T MFMAs of tile j + 1 (nothing else)
//     and the two wave groups (waves w and w + 4 share a SIMD) run ONE PHASE APART: while a wave is in its VALU-heavy
//     P1 its SIMD partner is in the MFMA-only P2, so each SIMD's matrix pipe always has 32 MFMAs to issue per phase slot
//     and the softmax never stands alone.  One score buffer (P2(j) overwrites what P1(j) consumed), two probability
//     buffers by name (the loop body is unrolled twice);
//   * K and V tiles arrive by hand-written LDS-DMA (SGPR base + 32-bit lane offset) into two 4-slot rings; the bundle
//     {K(t+1), V(t)} is requested two tiles ahead and waited for (counted vmcnt, then the phase barrier) one phase before
//     its first reader -- with the groups a phase apart that is the latest point that covers both;
//   * layouts as in attention_pipe.hip: S^T = K Q^T so a lane holds 32 scores of ONE query; P stays in registers as the
//     B operand of P V; V^T through ds_read_b64_tr_b16; softmax scale in f32; probabilities relative to the row maximum
//     of the first tile, never rescaled, with the same overflow-detecting fallback (running maximum per tile);
//   * LDS images (256-byte rows): K chunk c of row r at slot c ^ (r & 15) -- the 16 rows of a ds_read_b128 service
//     group cover all sixteen 16-byte slots; V 64-byte quarter q of row r at q ^ (r & 3) -- the 4 key rows of a
//     transposed read land on the four quarters of the bank row.
#include "attention.hpp"
#include "gemm_device.hpp"

namespace advgrpo {

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int D128_QB = 256;                  // queries per workgroup
constexpr int D128_TILE = ATT_KB * 256;       // 16 KiB per K or V tile
constexpr int D128_SLOTS = 4;
constexpr int D128_LDS = 2 * D128_SLOTS * D128_TILE + 64;   // 128 KiB + the fallback flags

__device__ __forceinline__ float d128_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// (through the compiler, not inline asm: see attention_pipe.hip -- the hazard recogniser only pads instructions it knows)
__device__ __forceinline__ uint32_t d128_cvt_pk(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
#define D128_SB() __builtin_amdgcn_sched_barrier(0)
#define D128_PIN(x) asm volatile("" : "+v"(x))
#define D128_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

}  // namespace

__global__ __launch_bounds__(512, 2) void attention_fwd_d128_kernel(const AttnParams p) {
    constexpr int HD = 128;
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // K ring (4 x 16 KiB) | V ring (4 x 16 KiB) | 8 flag words
    int* const wg_flag = reinterpret_cast<int*>(smem + 2 * D128_SLOTS * D128_TILE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                                         // wave group: group 1 runs one phase behind
    const int ql = lane & 31, hi = lane >> 5;
    int qblk, h, b;
    xcd_local_bh(p.nqb, p.H, p.nwg, p.xcd_local, qblk, h, b);
    const int q0 = qblk * D128_QB + wave * 32;
    const bf16_t* qp = p.q + (int64_t)b * p.bsq + h * HD;
    const bf16_t* kp = p.k + (int64_t)b * p.bsk + h * HD;
    const bf16_t* vp = p.v + (int64_t)b * p.bsv + h * HD;

    // ---- Q fragments (B operand of K Q^T: lane = query, 8 consecutive d at ks*16 + hi*8)
    bf16x8_t qf[8];
    {
        int qr = q0 + ql;
        qr = qr < p.Sq ? qr : p.Sq - 1;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + (int64_t)qr * p.ldq + ks * 16 + hi * 8);
    }

    // ---- DMA sources.  Instruction jj (0..15) of a tile fills LDS rows 4 jj .. 4 jj + 3 (1 KiB); this wave issues jj = wave, wave + 8.
    const int lrow = lane >> 4, slot16 = lane & 15;
    uint32_t k_lo[2], v_lo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int R = (wave + 8 * i) * 4 + lrow;
        k_lo[i] = (uint32_t)R * (uint32_t)(p.ldk * 2) + (uint32_t)((slot16 ^ (R & 15)) * 16);
        v_lo[i] = (uint32_t)R * (uint32_t)(p.ldv * 2) + (uint32_t)((((((slot16 >> 2) ^ (R & 3))) << 2) | (slot16 & 3)) * 16);
    }
    const int64_t k_step = (int64_t)ATT_KB * p.ldk * 2, v_step = (int64_t)ATT_KB * p.ldv * 2;
    const uint32_t k_lds = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)(smem)) + wave * 1024;
    const uint32_t v_lds = k_lds + D128_SLOTS * D128_TILE;
    auto dma = [&](const char* base, uint32_t off, uint32_t lds) __attribute__((always_inline)) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory");
    };
    // a ragged last tile clamps the rows past the end of the sequence to its last row (their scores are masked)
    auto stage_k = [&](int t, int slot) __attribute__((always_inline)) {
        const char* base = reinterpret_cast<const char*>(kp) + (int64_t)t * k_step;
        const uint32_t lds = k_lds + (slot & (D128_SLOTS - 1)) * D128_TILE;
        if ((t + 1) * ATT_KB > p.Skv) {
            asm volatile("; ragged K tile" ::: "memory");
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int R = (wave + 8 * i) * 4 + lrow;
                const int r = min(R, p.Skv - 1 - t * ATT_KB);
                dma(base, (uint32_t)r * (uint32_t)(p.ldk * 2) + (uint32_t)((slot16 ^ (R & 15)) * 16), lds + i * 8192);
            }
        } else {
            dma(base, k_lo[0], lds);
            dma(base, k_lo[1], lds + 8192);
        }
    };
    auto stage_v = [&](int t) __attribute__((always_inline)) {
        const char* base = reinterpret_cast<const char*>(vp) + (int64_t)t * v_step;
        const uint32_t lds = v_lds + (t & (D128_SLOTS - 1)) * D128_TILE;
        if ((t + 1) * ATT_KB > p.Skv) {
            asm volatile("; ragged V tile" ::: "memory");
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int R = (wave + 8 * i) * 4 + lrow;
                const int r = min(R, p.Skv - 1 - t * ATT_KB);
                dma(base, (uint32_t)r * (uint32_t)(p.ldv * 2) + (uint32_t)((((((slot16 >> 2) ^ (R & 3))) << 2) | (slot16 & 3)) * 16),
                    lds + i * 8192);
            }
        } else {
            dma(base, v_lo[0], lds);
            dma(base, v_lo[1], lds + 8192);
        }
    };
    const int nt = (p.Skv + ATT_KB - 1) / ATT_KB;
    // bundle t = {K(t+1), V(t)}: always 4 DMA instructions per wave, so the counted waits are constants: the bundle of the last
    // tile requests that tile's K rows once more, into the slot K(nt) would have used (dead: K(nt-4) was last read five phases
    // ago) -- never read
    auto stage_bundle = [&](int t) __attribute__((always_inline)) {
        stage_k(min(t + 1, nt - 1), t + 1);
        stage_v(t);
    };

    // ---- fragment read offsets (bytes inside a tile)
    //   K: key row kb*32 + ql, 16-byte chunk 2 ks + hi at slot ^ (row & 15)
    //   V^T: 16-lane group g16 reads the 4 (keys 4 hi ..) x 16 (d) block of d half g16 & 1; lane l16 supplies key row l16 >> 2,
    //        4 consecutive d at (l16 & 3) * 4; the 64-byte quarter db of a row sits at db ^ (row & 3)
    int k_off[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) k_off[ks] = ql * 256 + (((2 * ks + hi) ^ (ql & 15)) << 4);
    const int vrow = 4 * hi + ((lane & 15) >> 2);
    int v_off[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) v_off[db] = vrow * 256 + ((db ^ (vrow & 3)) << 6) + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const char* const Kr = smem;
    const char* const Vr = smem + D128_SLOTS * D128_TILE;
    auto kfrag = [&](const char* tile, int kb, int ks) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8_t*>(tile + kb * 8192 + k_off[ks]);
    };
    auto vfrag = [&](const char* tile, int kk, int db) __attribute__((always_inline)) {
        const char* a0 = tile + kk * 4096 + v_off[db];
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
    stage_k(0, 0);
    stage_bundle(0);
    if (nt > 1) stage_bundle(1);
    if (nt > 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x16 s[2];
    u32x4 pA[4], pB[4];
    auto qk_tile = [&](const char* tile) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) s[kb] = D128_MFMA(kfrag(tile, kb, ks), qf[ks], ks == 0 ? zero16 : s[kb]);
    };
    qk_tile(Kr);
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
    // the bundle that P1(1) / P2(0) read must have landed before the barrier that closes "P2(-1)"
    if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();      // stagger: group 1 runs one phase behind

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
    // One tile j.  P1: probabilities of tile j (scores s -> pn) beside the P V products of tile j - 1 (probabilities pp,
    // V(j-1)); P2: the scores of tile j + 1.  Slot sl of P1: MFMA (key step sl >> 2, d block sl & 3); score pair i: multiply-add
    // in slot i, exponentials in slot i + 1, bf16 pack + row sum in slot i + 2; the LDS operand of slot sl + 2 is read in slot sl.
    auto tile = [&](auto first_tag, int j, const u32x4 (&pp)[4], u32x4 (&pn)[4]) __attribute__((always_inline)) {
        constexpr bool HAVE_PV = !decltype(first_tag)::value;
        if (j + 2 < nt) stage_bundle(j + 2);
        if (j == nt - 1) mask_tail(s, j * ATT_KB);
        const char* vt = Vr + ((j - 1) & (D128_SLOTS - 1)) * D128_TILE;
        bf16x8_t a[3];
        if constexpr (HAVE_PV) {
            a[0] = vfrag(vt, 0, 0);
            a[1] = vfrag(vt, 0, 1);
        }
        f32x2 x[2], e[2];
        auto slot = [&](auto s_tag) __attribute__((always_inline)) {
            constexpr int sl = decltype(s_tag)::value;
            D128_SB();
            if constexpr (HAVE_PV && sl < 16) {
                o[sl & 3] = D128_MFMA(a[sl % 3], __builtin_bit_cast(bf16x8_t, pp[sl >> 2]), o[sl & 3]);
                if constexpr (sl + 2 < 16) a[(sl + 2) % 3] = vfrag(vt, (sl + 2) >> 2, (sl + 2) & 3);
            }
            if constexpr (sl >= 2 && sl - 2 < 16) {
                pn[(sl - 2) >> 2][(sl - 2) & 3] = d128_cvt_pk(e[sl & 1][0], e[sl & 1][1]);
                D128_PIN(pn[(sl - 2) >> 2][(sl - 2) & 3]);
                lsum[sl & 1] += e[sl & 1];
                D128_PIN(lsum[sl & 1]);
            }
            if constexpr (sl >= 1 && sl - 1 < 16) { e[(sl - 1) & 1] = pair_exp(x[(sl - 1) & 1]); D128_PIN(e[(sl - 1) & 1]); }
            if constexpr (sl < 16) { x[sl & 1] = pair_x(sl); D128_PIN(x[sl & 1]); }
        };
        static_for<18>(slot);
        D128_SB();
        __builtin_amdgcn_s_barrier();                                  // ---- end of P1(j)
        if (j + 1 < nt) qk_tile(Kr + ((j + 1) & (D128_SLOTS - 1)) * D128_TILE);
        // bundle j + 1 (read from the next phase on) must have landed: everything but the bundle requested in this tile
        if (j + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        D128_SB();
        __builtin_amdgcn_s_barrier();                                  // ---- end of P2(j)
    };
    auto tile_pv = [&](const char* vt, const u32x4 (&pp)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8_t pb = __builtin_bit_cast(bf16x8_t, pp[kk]);
#pragma unroll
            for (int db = 0; db < 4; ++db) o[db] = D128_MFMA(vfrag(vt, kk, db), pb, o[db]);
        }
    };
    typedef std::integral_constant<bool, true> first_t;
    typedef std::integral_constant<bool, false> steady_t;
    // P(j) lives in pA for even j, pB for odd j
    tile(first_t{}, 0, pB, pA);
    int j = 1;
    for (; j + 2 <= nt; j += 2) {
        tile(steady_t{}, j, pA, pB);
        tile(steady_t{}, j + 1, pB, pA);
    }
    if (j < nt) {
        tile(steady_t{}, j, pA, pB);
        tile_pv(Vr + ((nt - 1) & (D128_SLOTS - 1)) * D128_TILE, pB);
    } else {
        tile_pv(Vr + ((nt - 1) & (D128_SLOTS - 1)) * D128_TILE, pA);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();      // matches group 1's stagger barrier

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
#pragma unroll
            for (int db = 0; db < 4; ++db) o[db] = zero16;
            float m_run = -INFINITY, l_run = 0.f;
            for (int t = 0; t < nt; ++t) {
                __builtin_amdgcn_s_barrier();
                stage_k(t, t);
                stage_v(t);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                qk_tile(Kr + (t & (D128_SLOTS - 1)) * D128_TILE);
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
                tile_pv(Vr + (t & (D128_SLOTS - 1)) * D128_TILE, pA);
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
