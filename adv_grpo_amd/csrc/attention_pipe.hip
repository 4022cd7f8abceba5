// attention_pipe.hip -- software-pipelined flash attention forward for head dim 64 on gfx950 (bf16 in/out, f32 math).
//
// Serves the joint image+text attention and the image-only second attention of the MMDiT blocks (1229 / 1024 tokens at
// 512^2, 24 heads x 64; reference call sites sd3_pipeline_with_logprob_fast.py:630-637 and
// train_sd3_fast_pickscore.py:235-255 via F.scaled_dot_product_attention) and the DINOv2 reward tower (1370 tokens).
//
// Why a second kernel: at head dim 64 the softmax costs ~100 VALU instructions (32 of them quarter-rate v_exp_f32) per
// 16 MFMAs of a 32-query x 64-key tile -- the kernel is VALU-issue bound, not MFMA bound.  Measured on the chip
// (scripts/probes/valu_rates.hip): a wave that alternates "all MFMAs, then all softmax" leaves the matrix pipe idle during
// the softmax, and a partner wave on the same SIMD does NOT fill it (a VALU-only wave beside an MFMA-only wave runs at
// half its rate); what does work is the SAME wave issuing ~6-8 VALU instructions in the shadow of each of its own
// 32-cycle MFMAs, two such waves per SIMD (16 MFMA + 128 VALU per wave: 672 cycles per wave-tile against 512 of pure
// MFMA time).  So this kernel is pipelined IN the wave: while the VALU runs the softmax of tile j, the matrix pipe
// runs Q K^T of tile j + 1 and P V of tile j - 1, one MFMA per "slot" with a fixed share of the softmax in its shadow
// (the slots are pinned with sched_barrier; the compiler orders inside a slot only).
//
// What keeps the VALU share small:
//   * v_mfma_f32_32x32x16_bf16 with S^T = K Q^T: a lane holds 32 scores of ONE query (query = lane & 31), the other
//     half-wave the other 32 keys: row max = 16 v_max3 + one half-wave swap, no per-16-row bookkeeping twice.
//   * the softmax scale is applied in f32: score pair -> s * (scale * log2e) - m as ONE packed multiply-add in front of the two
//     v_exp_f32 (a first version folded scale * log2e into a bf16 copy of Q: 3.9e-2 max error on a 6x-scaled-keys case);
//   * NO per-tile maximum and NO rescale: m is fixed to the row maximum of tile 0 (the first 64 keys) and every probability
//     of every later tile is taken relative to it.  Mathematically the same softmax (any reference cancels in O / l), and
//     numerically the same to the bf16 rounding of P as long as no 2^(s - m) leaves the f32 exponent range.  What detects
//     that it did: the final row sum -- a row whose l is zero, larger than 1e30 or not finite (scores more than ~100 octaves
//     above or below tile 0's maximum) makes the WHOLE WORKGROUP redo its 128 queries with the classic per-tile running
//     maximum (the "fallback" loop at the end of the kernel: one tile at a time, no overlap -- several times slower, and
//     taken silently; tests/test_gpu_attention.py::test_attention_scores_far_outside_the_first_tiles_window forces it and
//     checks o and lse against the reference, DESIGN.md deviation 10 states how often trained-model-like scores take it);
//   * P stays in registers as the B operand of P V (the key order inside a 16-deep MFMA step is permuted consistently
//     on the P side and on the V side, which a sum over keys does not see); V^T fragments by ds_read_b64_tr_b16.
//
// Layout: workgroup = 4 waves = 128 queries of one (batch, head), three workgroups per CU (3 waves per SIMD, <= 168 VGPRs);
// K and V tiles of 64 keys arrive by LDS-DMA into two 3-slot rings (K runs two tiles ahead of V: iteration j multiplies
// K[j+1] and V[j-1]), requested two iterations ahead, counted s_waitcnt + ONE barrier per tile.  Swizzles (source side of
// the DMA, undone on the read): K chunk c of row r at LDS row r ^ ((r >> 4) & 1), slot c ^ (r & 7) -- the 16 rows of a
// ds_read_b128 service group then cover all sixteen 16-byte bank slots; V 64-byte half h of row r at h ^ ((r >> 1) & 1).
#include "attention.hpp"

// timing-only ablations (scripts/ablate_attention.sh; results are WRONG with any bit set): 1 = no wait + barrier, 2 = v_exp ->
// v_mul, 4 = no LDS fragment reads in the slots, 8 = no DMA, 16 = (unused), 32 = clock probe into lse[0..1], 64 = no row sums, 128 = never take the fallback,
// 256 = seven s_memrealtime stamps + HW_ID of every workgroup's wave 0 into lse (as 8 x u64 per workgroup; scripts/probes/attn_stamps.py)
#ifndef ATT_ABL
#define ATT_ABL 0
#endif

namespace advgrpo {

// workgroups that took the running-maximum fallback since the last reset (advgrpo_attention_fallback_count): the slow path is
// otherwise silent
__device__ unsigned long long g_pipe_fallbacks = 0;

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// (through the compiler, which fuses the pair into v_max3_f32: as an asm statement its reads of the score accumulators were
// not padded against the MFMAs that had just written them, and the reference maximum came out of half-written registers)
__device__ __forceinline__ float att_max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// two f32 -> packed bf16 (v_cvt_pk_bf16_f32).  Through the compiler, NOT inline asm: the packed probabilities are written
// into registers that MFMAs issued a slot earlier may still be reading as their B operand, and the hazard recogniser only
// pads instructions it knows (an asm statement here produced wrong probabilities for half of the query lanes).
__device__ __forceinline__ uint32_t att_cvt_pk(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float att_exp2(float x) {
    if constexpr ((ATT_ABL & 2) != 0) return x * 0.75f;
    return __builtin_amdgcn_exp2f(x);
}

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

#define ATT_SB() __builtin_amdgcn_sched_barrier(0)
// keeps a value (and the instructions that made it) in the slot where it was written: without it the row-sum adds are sunk
// out of the iteration (their result is only read at the very end) and all 32 probabilities stay live across it
#define ATT_PIN(x) asm volatile("" : "+v"(x))
#define ATT_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

}  // namespace

// (three workgroups per CU since round 4: 168 VGPRs + 32 bytes of scratch outside the main loop against 180 / 0 at two -- the 3840
// workgroups of the joint attention (16 x 24 x 1229) are exactly five rounds of 768 instead of seven and a half of 512: 219 vs 226 - 234 us,
// S = 1370: 73 - 75 vs 77 - 79, S = 4301: 1.92 vs 1.98 ms, S = 1024 unchanged; 3 x 48 KiB of LDS)
__global__ __launch_bounds__(256, 3) void attention_fwd_pipe_kernel(const AttnParams p) {
    constexpr int HD = 64, TILE_B = ATT_KB * 128;             // 8 KiB per K or V tile
    __shared__ __attribute__((aligned(1024))) char smem[6 * TILE_B];
    __shared__ int wg_flag[4];
    char* const Kr = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hi = lane >> 5;
    [[maybe_unused]] unsigned long long clk0 = 0, rt0 = 0;
    if constexpr ((ATT_ABL & 32) != 0) { clk0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    [[maybe_unused]] unsigned long long stamp[7];
#ifndef ATT_FENCE
#define ATT_FENCE 0
#endif
#define ATT_STAMP(i) if constexpr ((ATT_ABL & 256) != 0) { asm volatile("" ::: "memory"); stamp[i] = __builtin_amdgcn_s_memrealtime(); asm volatile("" ::: "memory"); } \
                     else if constexpr (((ATT_FENCE >> (i)) & 1) != 0) { asm volatile("" ::: "memory"); }
    ATT_STAMP(0)
    int qblk, h, b;
    xcd_local_bh(p.nqb, p.H, p.nwg, p.xcd_local, qblk, h, b);
    const int q0 = qblk * ATT_QB + wave * 32;
    const bf16_t* qp = p.q + (int64_t)b * p.bsq + h * HD;
    const bf16_t* kp = p.k + (int64_t)b * p.bsk + h * HD;
    const bf16_t* vp = p.v + (int64_t)b * p.bsv + h * HD;

    bf16x8_t qf[4];          // Q fragments: loaded in the prologue, behind the first DMA requests
    // ---- DMA sources.  Instruction jj (0..7) of a tile fills LDS rows 8 jj .. 8 jj + 7; this wave issues jj = wave, wave + 4.
    // Uniform (SGPR) tile base + 32-bit per-lane byte offset: the builtin form keeps a 64-bit pointer per lane and instruction.
    const int lrow = lane >> 3, pch = lane & 7;
    // byte offset of this lane's 16 bytes inside an interior tile for instruction jj = wave; instruction wave + 4 is 32 LDS rows = 32 tile
    // rows further on with the same swizzle term (R + 32 keeps bit 4, bits 0..2 and bit 1 of R), so it uses the SAME per-lane offset with
    // the uniform base moved by 32 rows: one register each for K and V (with two, the pair was spilled and reloaded -- scratch_load +
    // s_waitcnt vmcnt(0), i.e. a wait for the tile DMA just issued -- in every ring step)
    uint32_t k_lo, v_lo;
    {
        const int R = wave * 8 + lrow;                        // LDS row
        const int rk = R ^ ((R >> 4) & 1);                    // tile row held by that LDS row (K)
        k_lo = (uint32_t)rk * (uint32_t)(p.ldk * 2) + (uint32_t)((pch ^ (rk & 7)) * 16);
        v_lo = (uint32_t)R * (uint32_t)(p.ldv * 2) + (uint32_t)((pch ^ (((R >> 1) & 1) << 2)) * 16);
    }
    const int64_t k_half = (int64_t)32 * p.ldk * 2, v_half = (int64_t)32 * p.ldv * 2;
    const int64_t k_step = (int64_t)ATT_KB * p.ldk * 2, v_step = (int64_t)ATT_KB * p.ldv * 2;      // bytes per tile
    const uint32_t k_lds = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)(Kr)) + wave * 1024;
    const uint32_t v_lds = k_lds + 3 * TILE_B;
    auto dma = [&](const char* base, uint32_t off, uint32_t lds) __attribute__((always_inline)) {
        if constexpr ((ATT_ABL & 8) != 0) return;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory");
    };
    // a ragged last tile clamps the rows past the end of the sequence to its last row (their scores are masked)
    auto stage_k = [&](int slot, int t) __attribute__((always_inline)) {
        const char* base = reinterpret_cast<const char*>(kp) + (int64_t)t * k_step;
        const uint32_t lds = k_lds + slot * TILE_B;
        if ((t + 1) * ATT_KB > p.Skv) {
            asm volatile("; ragged K tile" ::: "memory");
            int rl;                           // (fresh lane id: see the epilogue)
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(rl));
            const int lrow = rl >> 3, pch = rl & 7;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int R = (wave + 4 * i) * 8 + lrow;
                const int rk = R ^ ((R >> 4) & 1);
                const int r = min(rk, p.Skv - 1 - t * ATT_KB);
                dma(base, (uint32_t)r * (uint32_t)(p.ldk * 2) + (uint32_t)((pch ^ (rk & 7)) * 16), lds + i * 4096);
            }
        } else {
            dma(base, k_lo, lds);
            dma(base + k_half, k_lo, lds + 4096);
        }
    };
    auto stage_v = [&](int slot, int t) __attribute__((always_inline)) {
        const char* base = reinterpret_cast<const char*>(vp) + (int64_t)t * v_step;
        const uint32_t lds = v_lds + slot * TILE_B;
        if ((t + 1) * ATT_KB > p.Skv) {
            asm volatile("; ragged V tile" ::: "memory");
            int rl;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(rl));
            const int lrow = rl >> 3, pch = rl & 7;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int R = (wave + 4 * i) * 8 + lrow;
                const int r = min(R, p.Skv - 1 - t * ATT_KB);
                dma(base, (uint32_t)r * (uint32_t)(p.ldv * 2) + (uint32_t)((pch ^ (((R >> 1) & 1) << 2)) * 16), lds + i * 4096);
            }
        } else {
            dma(base, v_lo, lds);
            dma(base + v_half, v_lo, lds + 4096);
        }
    };

    // ---- fragment read offsets (bytes inside a tile)
    const int rowpos = ql ^ ((ql >> 4) & 1);
    int k_off[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) k_off[ks] = rowpos * 128 + ((((2 * ks + hi) ^ (lane & 7))) << 4);
    const int vb = (4 * hi + ((lane & 15) >> 2)) * 128 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
    const int b3 = (lane >> 3) & 1;
    int v_off[2];
    v_off[0] = vb + (b3 ? 64 : 0);
    v_off[1] = vb + (b3 ? 0 : 64);

    typedef __attribute__((ext_vector_type(8))) short s16x8;
    // fragment reads at byte offset `off` (slot base + block offset: an immediate when the slot is a compile-time constant)
    auto kfrag = [&](int off, int ks) __attribute__((always_inline)) {
        if constexpr ((ATT_ABL & 4) != 0) return qf[ks];
        return *reinterpret_cast<const bf16x8_t*>(smem + k_off[ks] + off);
    };
    auto vfrag = [&](int off, int db) __attribute__((always_inline)) {
        if constexpr ((ATT_ABL & 4) != 0) return qf[db];
        const char* a0 = smem + v_off[db] + off;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
        const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 1024));
        const s16x8 both = __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8_t, both);
    };

    const int nt = (p.Skv + ATT_KB - 1) / ATT_KB;
    f32x16 o[2];          // O^T accumulators: d block db, lane (query = lane & 31, hi): d = db*32 + 8*(r>>2) + 4*hi + (r&3)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    o[0] = zero16; o[1] = zero16;
    f32x2 lsum[2] = {{0.f, 0.f}, {0.f, 0.f}};    // this lane's share of the row sum (two packed accumulators)
    // (the reference maximum m_ref -- log2 units, the row maximum of tile 0 -- lives only as nm2 = {-m_ref, -m_ref}: a separate register
    //  for it, live across the main loop for the sake of the log-sum-exp output, was the 169th and cost the kernel its only scratch
    //  spills: 6 scratch instructions, 1.2 % of the launch at 16 x 24 x 1229)
    const f32x2 c2 = {p.scale_log2e, p.scale_log2e};
    f32x2 nm2 = {0.f, 0.f};              // -m_ref twice: addend of the packed multiply-add in front of the exponentials

    // ---- a wave whose 32 queries all lie past the end of the sequence (the last query block of a (batch, head) when Sq is not a
    // multiple of 128: wave 3 of every tenth workgroup at Sq = 1229) only keeps the workgroup's protocol -- its two DMA instructions per
    // tile and every barrier -- and leaves its SIMD's matrix and vector issue slots to the two other workgroups resident there
    if (q0 >= p.Sq) {
        stage_k(0, 0);
        if (nt > 1) stage_k(1, 1);
        if (nt > 2) stage_k(2, 2);
        stage_v(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int j = 0; j + 1 < nt; ++j) {
            if (j + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (j + 3 < nt) stage_k(j % 3, j + 3);
            if (j + 1 < nt) stage_v((j + 1) % 3, j + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                            // the tail's barrier
        if (lane == 0) wg_flag[wave] = 0;
        __syncthreads();
        if ((ATT_ABL & 128) == 0 && (wg_flag[0] | wg_flag[1] | wg_flag[2] | wg_flag[3])) {      // the others take the fallback loop
            for (int t = 0; t < nt; ++t) {
                __builtin_amdgcn_s_barrier();
                stage_k(0, t);
                stage_v(0, t);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ---- prologue: K0 | K1 | K2, V0 ; S_0 = K0 Q^T ; reference maximum = row maximum of tile 0
    stage_k(0, 0);
    if (nt > 1) stage_k(1, 1);
    if (nt > 2) stage_k(2, 2);
    stage_v(0, 0);
    ATT_STAMP(1)
    // ---- Q fragments (B operand of K Q^T: lane = query, 8 consecutive d at ks*16 + hi*8), unscaled: the softmax scale is applied in f32.
    // Requested BEHIND the first tiles' DMA (round 5; before: loaded and waited for in front of it, two memory latencies in a row -- 1.5 +
    // 0.95 us of a workgroup's 31 us at 16 x 24 x 1229, scripts/probes/attn_stamps.py): memory operations of a wave complete in order, so
    // the compiler's own wait for these four loads -- placed right here by the empty asm uses: left to itself it would sit in front of
    // the first MFMA of some later slot and drain the ring there -- covers the older DMA requests as well.
    {
        int qr = q0 + ql;
        qr = qr < p.Sq ? qr : p.Sq - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8_t raw = *reinterpret_cast<const bf16x8_t*>(qp + (int64_t)qr * p.ldq + ks * 16 + hi * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[ks][e] = raw[e];
        }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[ks]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    ATT_STAMP(2)
    f32x16 sA[2], sB[2];
    u32x4 pA[4], pB[4];
    sA[0] = zero16; sA[1] = zero16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) sA[kb] = ATT_MFMA32(kfrag(kb * 4096, ks), qf[ks], sA[kb]);

    auto row_max = [&](const f32x16 (&sc)[2]) __attribute__((always_inline)) {
        float m0 = att_max3(sc[0][0], sc[0][1], sc[0][2]);
        float m1 = att_max3(sc[0][3], sc[0][4], sc[0][5]);
        float m2 = att_max3(sc[1][0], sc[1][1], sc[1][2]);
        float m3 = att_max3(sc[1][3], sc[1][4], sc[1][5]);
        m0 = att_max3(m0, sc[0][6], sc[0][7]);
        m1 = att_max3(m1, sc[0][8], sc[0][9]);
        m2 = att_max3(m2, sc[1][6], sc[1][7]);
        m3 = att_max3(m3, sc[1][8], sc[1][9]);
        m0 = att_max3(m0, sc[0][10], sc[0][11]);
        m1 = att_max3(m1, sc[0][12], sc[0][13]);
        m2 = att_max3(m2, sc[1][10], sc[1][11]);
        m3 = att_max3(m3, sc[1][12], sc[1][13]);
        m0 = att_max3(m0, sc[0][14], sc[0][15]);
        m2 = att_max3(m2, sc[1][14], sc[1][15]);
        m0 = att_max3(m0, m1, m2);
        float a = att_max3(m0, m3, m3), b = a;
        ADVGRPO_SWAP32(a, b);
        return att_max3(a, b, b);
    };
    // score pair i (0..15) of a tile = elements 2i, 2i+1 of the lane's 32 scores, in log2 units relative to the reference max
    auto pair_x = [&](const f32x16 (&sc)[2], int i) __attribute__((always_inline)) {
        const f32x2 s2 = {sc[i >> 3][(2 * i) & 15], sc[i >> 3][((2 * i) & 15) + 1]};
        return __builtin_elementwise_fma(s2, c2, nm2);
    };
    auto pair_exp = [&](f32x2 x) __attribute__((always_inline)) {
        f32x2 e;
        e[0] = att_exp2(x[0]);
        e[1] = att_exp2(x[1]);
        return e;
    };
    // probabilities of a whole tile without overlap (tail, fallback): raw scores -> packed bf16 B operands
    auto tile_probs = [&](const f32x16 (&sc)[2], u32x4 (&pn)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f32x2 e = pair_exp(pair_x(sc, i));
            pn[i >> 2][i & 3] = att_cvt_pk(e[0], e[1]);
            lsum[i & 1] += e;
        }
    };
    // O += P V for one tile whose V sits at byte offset vt (8 MFMAs, no overlap: tail, fallback)
    auto tile_pv = [&](int vt, const u32x4 (&pp)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const bf16x8_t pb = __builtin_bit_cast(bf16x8_t, pp[kk]);
            o[0] = ATT_MFMA32(vfrag(vt + kk * 2048, 0), pb, o[0]);
            o[1] = ATT_MFMA32(vfrag(vt + kk * 2048, 1), pb, o[1]);
        }
    };
    auto mask_tail = [&](f32x16 (&sc)[2], int kv0) __attribute__((always_inline)) {
        if (kv0 + ATT_KB > p.Skv) {       // keys past the end of the sequence
            int ml;                           // (fresh lane id: `hi` kept live across the main loop for this one use was spilled)
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ml));
            const int mhi = ml >> 5;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + kb * 32 + 8 * (r >> 2) + 4 * mhi + (r & 3);
                    if (key >= p.Skv) sc[kb][r] = -INFINITY;
                }
        }
    };
    {
        const float m_ref = row_max(sA) * p.scale_log2e;      // (a row of tile 0 always holds a real key: finite)
        nm2 = f32x2{-m_ref, -m_ref};
    }
    ATT_STAMP(3)

    // One steady-state iteration j (0 <= j <= nt - 2), j = PH (mod 3): probabilities of tile j (raw scores sc -> pn), Q K^T
    // of tile j + 1 into sn, P V of tile j - 1 (probabilities pp).  16 MFMA slots: 0..7 = Q K^T (key blocks alternate, so no MFMA
    // waits for the one just issued), 8..15 = P V (16-key step kk, d-block alternating); the A operand of slot s + 2 is read
    // from LDS in slot s; score pair i: multiply-add in slot i, exponentials in slot i + 1, bf16 pack + row sum in slot i + 2.  Ring slots: K[j+1] in (PH+1)%3, V[j-1] in (PH+2)%3, the requests for K[j+3] and
    // V[j+1] go to PH and (PH+1)%3.
    auto iteration = [&](const int PH, auto first_tag, int j, const f32x16 (&sc)[2], f32x16 (&sn)[2], const u32x4 (&pp)[4],
                         u32x4 (&pn)[4]) __attribute__((always_inline)) {
        // (PH is a literal in the unrolled main loop: every slot offset below folds to an immediate of the LDS reads)
        constexpr bool HAVE_PV = !decltype(first_tag)::value;         // j == 0: no previous tile to multiply
        const int KRD = ((PH + 1) % 3) * TILE_B, VRD = (3 + (PH + 2) % 3) * TILE_B;
        // tiles needed now: K[j+1] and V[j-1] (requested two iterations ago); still in flight: K[j+2], V[j]
        if constexpr ((ATT_ABL & 1) == 0) {
            if (j + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (j + 3 < nt) stage_k(PH, j + 3);                           // the slot of K[j], read in iteration j - 1
        if (j + 1 < nt) stage_v((PH + 1) % 3, j + 1);                 // the slot of V[j-2]
        // LDS operand of MFMA slot s
        auto needs_operand = [](int s) constexpr { return s < 8 || (HAVE_PV && s < 16); };
        auto operand = [&](auto s_tag) __attribute__((always_inline)) {
            constexpr int s = decltype(s_tag)::value;
            if constexpr (s < 8) return kfrag(KRD + (s & 1) * 4096, s >> 1);
            else return vfrag(VRD + ((s - 8) >> 1) * 2048, (s - 8) & 1);
        };
        bf16x8_t a[3];
        a[0] = operand(std::integral_constant<int, 0>{});
        a[1] = operand(std::integral_constant<int, 1>{});
        f32x2 x[2], e[2];
        auto slot = [&](auto s_tag) __attribute__((always_inline)) {
            constexpr int s = decltype(s_tag)::value;
            ATT_SB();
            if constexpr (s < 8) {
                if constexpr ((s >> 1) == 0) sn[s & 1] = ATT_MFMA32(a[s % 3], qf[0], zero16);
                else sn[s & 1] = ATT_MFMA32(a[s % 3], qf[s >> 1], sn[s & 1]);
            } else if constexpr (HAVE_PV && s < 16) {
                o[(s - 8) & 1] = ATT_MFMA32(a[s % 3], __builtin_bit_cast(bf16x8_t, pp[(s - 8) >> 1]), o[(s - 8) & 1]);
            }
            if constexpr (needs_operand(s + 2)) a[(s + 2) % 3] = operand(std::integral_constant<int, (s + 2) % 32>{});
            // (the empty asm statements keep each piece in its slot: without them the whole softmax is scheduled in front of
            // the first MFMA -- sched_barrier alone did not hold it)
            if constexpr (s >= 2 && s - 2 < 16) {
                pn[(s - 2) >> 2][(s - 2) & 3] = att_cvt_pk(e[s & 1][0], e[s & 1][1]);
                ATT_PIN(pn[(s - 2) >> 2][(s - 2) & 3]);
                if constexpr ((ATT_ABL & 64) == 0) { lsum[s & 1] += e[s & 1]; ATT_PIN(lsum[s & 1]); }
            }
            if constexpr (s >= 1 && s - 1 < 16) { e[(s - 1) & 1] = pair_exp(x[(s - 1) & 1]); ATT_PIN(e[(s - 1) & 1]); }
            if constexpr (s < 16) { x[s & 1] = pair_x(sc, s); ATT_PIN(x[s & 1]); }
        };
        slot(std::integral_constant<int, 0>{}); slot(std::integral_constant<int, 1>{});
        slot(std::integral_constant<int, 2>{}); slot(std::integral_constant<int, 3>{});
        slot(std::integral_constant<int, 4>{}); slot(std::integral_constant<int, 5>{});
        slot(std::integral_constant<int, 6>{}); slot(std::integral_constant<int, 7>{});
        slot(std::integral_constant<int, 8>{}); slot(std::integral_constant<int, 9>{});
        slot(std::integral_constant<int, 10>{}); slot(std::integral_constant<int, 11>{});
        slot(std::integral_constant<int, 12>{}); slot(std::integral_constant<int, 13>{});
        slot(std::integral_constant<int, 14>{}); slot(std::integral_constant<int, 15>{});
        slot(std::integral_constant<int, 16>{}); slot(std::integral_constant<int, 17>{});     // (the last two packs, no MFMA)
        ATT_SB();
    };

    // ---- main loop: iterations 0 .. nt - 2.  The ring phase has period 3 and the score / probability buffers alternate by
    // name, so the body is unrolled six times (no exit inside: an exit edge per copy made the register allocator spill ~900
    // values); the up to five left-over iterations run one at a time with the ring phase in a register and the buffers
    // copied back to their names.  Invariant between iterations (nt >= 2): current tile's scores in sB, previous tile's
    // probabilities in pA.
    typedef std::integral_constant<bool, true> first_t;
    typedef std::integral_constant<bool, false> steady_t;
#ifndef ATT_PRIO
#define ATT_PRIO 0      // experiment (scripts/probes/r6_job31.sh): 1 / 2 = s_setprio 1 / 3 over the main loop, 3 = static priority by workgroup parity: no gain, see LABNOTES 9
#endif
    if constexpr (ATT_PRIO == 1) __builtin_amdgcn_s_setprio(1);
    if constexpr (ATT_PRIO == 2) __builtin_amdgcn_s_setprio(3);
    if constexpr (ATT_PRIO == 3) { if (blockIdx.x & 1) __builtin_amdgcn_s_setprio(1); }
    if (nt >= 2) {
        iteration(0, first_t{}, 0, sA, sB, pB, pA);
        int j = 1;
        const int nfull = nt - 1;
        for (; j + 6 <= nfull; j += 6) {
            iteration(1, steady_t{}, j, sB, sA, pA, pB);
            iteration(2, steady_t{}, j + 1, sA, sB, pB, pA);
            iteration(0, steady_t{}, j + 2, sB, sA, pA, pB);
            iteration(1, steady_t{}, j + 3, sA, sB, pB, pA);
            iteration(2, steady_t{}, j + 4, sB, sA, pA, pB);
            iteration(0, steady_t{}, j + 5, sA, sB, pB, pA);
        }
        for (; j < nfull; ++j) {
            iteration(j % 3, steady_t{}, j, sB, sA, pA, pB);
            sB[0] = sA[0]; sB[1] = sA[1];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) pA[kk] = pB[kk];
        }
    } else {
        sB[0] = sA[0]; sB[1] = sA[1];
    }
    // now: the last tile's raw scores in sB, the probabilities of tile nt - 2 (if any) in pA

    // ---- tail: P V of tile nt - 2, probabilities of the (possibly ragged) last tile, its P V
    ATT_STAMP(4)
    if constexpr (ATT_PRIO != 0) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (nt > 1) tile_pv((3 + (nt - 2) % 3) * TILE_B, pA);
    mask_tail(sB, (nt - 1) * ATT_KB);
    {
        // a ragged last tile only pays for the 16-key steps that hold real keys (13 of 64 at Skv = 1229: one step -- 4 of the 16 score
        // pairs, 2 of the 8 products; the masked scores would have come out as exact zeros: same sums, same bits)
        const int steps = (p.Skv - (nt - 1) * ATT_KB + 15) >> 4;         // 1 .. 4, uniform
        const int vt = (3 + (nt - 1) % 3) * TILE_B;
        if (steps >= 4) {
            tile_probs(sB, pB);
            tile_pv(vt, pB);
        } else {
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                if (kk < steps) {
#pragma unroll
                    for (int i = 4 * kk; i < 4 * kk + 4; ++i) {
                        const f32x2 e = pair_exp(pair_x(sB, i));
                        pB[kk][i & 3] = att_cvt_pk(e[0], e[1]);
                        lsum[i & 1] += e;
                    }
                    const bf16x8_t pb = __builtin_bit_cast(bf16x8_t, pB[kk]);
                    o[0] = ATT_MFMA32(vfrag(vt + kk * 2048, 0), pb, o[0]);
                    o[1] = ATT_MFMA32(vfrag(vt + kk * 2048, 1), pb, o[1]);
                }
            }
        }
    }
    ATT_STAMP(5)

    // ---- the window check.  Every probability was taken relative to the row maximum of tile 0 and nothing was rescaled
    // on the way: exact as long as no 2^(s - m_ref) left the f32 / bf16 exponent range.  A row sum that is zero, huge or
    // not finite says it did (scores more than ~100 octaves away from tile 0's); then the whole workgroup redoes its
    // block with the classic per-tile running maximum -- slow, and practically never taken.
    float l = xor32_add((lsum[0][0] + lsum[0][1]) + (lsum[1][0] + lsum[1][1]));       // both key halves of the query
    {
        const bool bad = !(l > 1e-30f && l < 1e30f);
        // workgroup-wide OR through four LDS words (__syncthreads_or came back non-zero at random here: workgroups then took the
        // fallback, whose rounding differs in the last bit, and two launches on the same data were not bitwise equal)
        const bool wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
        if (lane == 0) wg_flag[wave] = wave_bad ? 1 : 0;
        __syncthreads();
        const int any_bad = wg_flag[0] | wg_flag[1] | wg_flag[2] | wg_flag[3];
        if ((ATT_ABL & 128) == 0 && any_bad) {
            asm volatile("; fallback: running maximum per tile" ::: "memory");
            if (tid == 0) atomicAdd(&g_pipe_fallbacks, 1ull);
            o[0] = zero16; o[1] = zero16;
            float m_run = -INFINITY, l_run = 0.f;
            for (int t = 0; t < nt; ++t) {
                __builtin_amdgcn_s_barrier();                    // everyone is done with slot 0
                stage_k(0, t);
                stage_v(0, t);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                sA[0] = zero16; sA[1] = zero16;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) sA[kb] = ATT_MFMA32(kfrag(kb * 4096, ks), qf[ks], sA[kb]);
                mask_tail(sA, t * ATT_KB);
                const float m_new = fmaxf(m_run, row_max(sA) * p.scale_log2e);
                const float f = att_exp2(m_run - m_new);         // first tile: exp2(-inf) = 0 on zeros
                m_run = m_new;
                nm2 = f32x2{-m_new, -m_new};
#pragma unroll
                for (int r = 0; r < 16; ++r) { o[0][r] *= f; o[1][r] *= f; }
                l_run *= f;
                lsum[0] = f32x2{0.f, 0.f}; lsum[1] = f32x2{0.f, 0.f};
                tile_probs(sA, pA);
                tile_pv(3 * TILE_B, pA);
                l_run += xor32_add((lsum[0][0] + lsum[0][1]) + (lsum[1][0] + lsum[1][1]));
            }
            l = l_run;                                       // (nm2 = -m_run already)
            __builtin_amdgcn_s_barrier();                        // the epilogue reuses the K ring
        }
    }

    // ---- epilogue: normalise, bounce the wave's 32 x 64 bf16 tile through LDS (K ring: every wave is past the tail's barrier
    // and the tail only reads V slots), store whole 128-byte rows, 16 bytes per lane
    if constexpr ((ATT_ABL & 32) != 0) {      // experiment: shader clock cycles and 100 MHz ticks of one wave's lifetime -> lse[0..3]
        const unsigned long long clk1 = __builtin_readcyclecounter(), rt1 = __builtin_amdgcn_s_memrealtime();
        if (blockIdx.x == 1000 && tid == 0 && p.lse) {
            ((unsigned long long*)p.lse)[0] = clk1 - clk0;
            ((unsigned long long*)p.lse)[1] = rt1 - rt0;
        }
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    // the lane id re-derived (two VALU instructions) from here on: lane / ql / q0 + ql kept live across the main loop for the sake of these
    // few address computations were what the 168-register budget spilled (4 scratch instructions and their waits per workgroup)
    int el;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(el));
    const int eql = el & 31, ehi = el >> 5;
    const int q0s = __builtin_amdgcn_readfirstlane(q0);
    const int qi = q0s + eql;
    if ((ATT_ABL & (32 | 256)) == 0 && p.lse && ehi == 0 && qi < p.Sq) p.lse[((int64_t)b * p.H + h) * p.Sq + qi] = __builtin_amdgcn_logf(l) - nm2[0];
    char* ob = Kr + wave * 4096;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 pk;
            pk.x = att_cvt_pk(o[db][4 * i] * inv, o[db][4 * i + 1] * inv);
            pk.y = att_cvt_pk(o[db][4 * i + 2] * inv, o[db][4 * i + 3] * inv);
            // d = db*32 + 8 i + 4 hi .. + 3  ->  16-byte chunk db*4 + i, 8-byte half hi
            *reinterpret_cast<uint2*>(ob + eql * 128 + (((db * 4 + i) ^ (eql & 7)) << 4) + ehi * 8) = pk;
        }
    // (each wave reads back only what it wrote itself: no barrier, the LDS accesses of one wave are ordered)
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int r = ps * 8 + (el >> 3), c = el & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(ob + r * 128 + ((c ^ (r & 7)) << 4));
        const int qo = q0s + r;
        if (qo < p.Sq) *reinterpret_cast<uint4*>(p.o + (int64_t)b * p.bso + (int64_t)qo * p.ldo + h * HD + c * 8) = v;
    }
    if constexpr ((ATT_ABL & 256) != 0) {
        ATT_STAMP(6)
        if (tid == 0 && p.lse) {
            unsigned long long* dst = (unsigned long long*)p.lse + (size_t)blockIdx.x * 8;
            for (int i = 0; i < 7; ++i) dst[i] = stamp[i];
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            dst[7] = ((unsigned long long)xcc << 32) | hw;
        }
    }
}

int attention_pipe_fallbacks(unsigned long long* out, int reset) {
    unsigned long long v = 0, z = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_pipe_fallbacks), sizeof(v)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_pipe_fallbacks), &z, sizeof(z)) != hipSuccess) return -1;
    *out = v;
    return 0;
}

int attention_fwd_pipe_launch(const AttnParams& p, hipStream_t s) {
    hipLaunchKernelGGL(attention_fwd_pipe_kernel, dim3((unsigned)p.nwg), dim3(256), 0, s, p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

}  // namespace advgrpo
