// attention_bwd_pipe.hip -- software-pipelined attention backward for head dim 64 on gfx950 (bf16 in/out, f32 math).
//
// Needed by the G-step: autograd of the transformer call in compute_log_prob (scripts/train_sd3_fast_pickscore.py:233-267) as
// reached from loss.backward() (:1165); diffusers runs F.scaled_dot_product_attention, whose backward this replaces.
//
// ONE kernel template, two instantiations, the same arithmetic as the first version (probabilities recomputed from the forward's
// base-2 log-sum-exp, no atomics, deterministic):
//   dQ    (DKDV = false): a workgroup OWNS 128 queries (wave = 32, lane = one query) and STREAMS the keys in tiles of 32:
//           S^T = K Q^T, P^T = exp2(S^T c - L[q]), dP^T - D = V dO^T - D[q], dS^T = P^T (dP^T - D), dQ^T += K^T dS^T
//   dK/dV (DKDV = true):  a workgroup owns 128 keys (lane = one key) and streams the queries:
//           S - L/c = Q K^T - L[q]/c, P = exp2(c (S - L/c)), dP - D = dO V^T - D[q], dS = P (dP - D),
//           dV^T += dO^T P, dK^T += Q^T dS
// The first version of both ran "all MFMAs of a tile, then all exponentials": 1.06 cycles per (query, key) pair and SIMD against 0.5
// of pure matrix time (dK/dV), i.e. matrix pipe and VALU strictly in turn.  This one is pipelined IN the wave like the forward
// (attention_pipe.hip): step g issues, one v_mfma_f32_32x32x16_bf16 per "slot",
//   slots 0 .. NA-1   the accumulating products of tile g-1 (A = transposed rows of the streamed tile by ds_read_b64_tr_b16,
//                     B = the packed bf16 P / dS of tile g-1, which never leave registers),
//   slots NA .. NA+7  the two score products of tile g+1 (A = rows of the streamed tile, B = the wave's own Q,dO / K,V fragments),
// and the VALU works on tile g in their shadows (pair i: scale in slot i+VX, two v_exp_f32 in the next, multiply and bf16 packs in
// the one after: 6 - 7 VALU per pair, 4 per MFMA slot -- the forward needs 8).  What keeps the VALU share small:
//   * the "- D[q]" (and, for dK/dV, the "- L[q]") never touch the VALU: they are the INITIAL VALUE of the MFMA accumulator (dQ: 16
//     registers holding -D of the lane's query; dK/dV: the vectors -L/c and -D, written by the delta kernel in blocks of 32 queries,
//     arrive in LDS beside the tile and are read straight into the accumulator registers);
//   * keys / queries past the end are handled outside the loop (dQ: the last tile's scores are set to -inf before its step; dK/dV:
//     the padding queries of the L vector are -inf, so their probabilities are exactly 0).
// Streamed tiles (32 rows x 128 bytes, ONE copy serves both access patterns) arrive by LDS-DMA in an 8-slot ring, requested four
// steps ahead, counted s_waitcnt + one barrier per step; the body is unrolled eight times so every LDS offset is an immediate.
// Swizzle (source side of the DMA): 16-byte chunk c of row r sits at chunk c ^ f(r), f(r) = 4*((r >> 1) & 1) | ((r >> 2) & 3):
//   - ds_read_b128 row fragments: the sixteen lanes of a service group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, MI355X_MICROARCH.md
//     LDS table) hold rows whose (r & 1, f(r)) are all different -> sixteen different 16-byte bank slots;
//   - ds_read_b64_tr_b16: a 32-lane group reads 4 rows x 64 bytes; rows r and r + 2 differ in bit 2 of f, r and r + 1 in the row
//     parity (128-byte pitch) -> again sixteen different slots.
// Layout: workgroup = 4 waves, two workgroups per CU (<= 256 VGPRs), 66 KiB of LDS each.
#include <type_traits>
#include <utility>

#include "attention_bwd.hpp"

// timing-only ablations (results are WRONG with any bit set): 1 = no wait + barrier, 2 = v_exp -> v_mul, 4 = no LDS fragment reads,
// 8 = no DMA
#ifndef BP_ABL
#define BP_ABL 0
#endif
// LDS operand reads run this many MFMA slots ahead of their use (4 staging registers: 2 or 3)
#ifndef BP_AHEAD
#define BP_AHEAD 3
#endif
// waves per workgroup: 4 (128 own rows, two workgroups per CU) or 8 (256 own rows, one workgroup per CU: half the L2 -> LDS
// traffic per MFMA, every wave requests ONE of the two operands' quarter tiles)
#ifndef BP_WAVES
#define BP_WAVES 4
#endif
// MFMA slot in whose shadow the step's DMA requests are issued (-1: right after the barrier, where every wave of the CU issues
// its requests at the same moment)
#ifndef BP_DMA_SLOT
#define BP_DMA_SLOT -1
#endif

namespace advgrpo {

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int BP_ROWS = 32;                 // streamed rows per tile
constexpr int BP_TILE = BP_ROWS * 128;      // bytes
constexpr int BP_RING = 8;                  // tiles per ring
constexpr int BP_RING1 = BP_RING * BP_TILE; // byte offset of the second operand's ring
constexpr int BP_VEC = 2 * BP_RING1;        // byte offset of the (-L/c | -D) vectors, 256 bytes per slot
constexpr int BP_LDS_DQ = BP_VEC, BP_LDS_DKDV = BP_VEC + BP_RING * 256;
constexpr int BP_OWN = BP_WAVES * 32;      // own rows per workgroup

#define BP_SB() __builtin_amdgcn_sched_barrier(0)
#define BP_PIN(x) asm volatile("" : "+v"(x))
#define BP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

__device__ __forceinline__ uint32_t bp_cvt_pk(float lo, float hi) {      // (through the compiler: see attention_pipe.hip)
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bp_exp2(float x) {
    if constexpr ((BP_ABL & 2) != 0) return x * 0.75f;
    return __builtin_amdgcn_exp2f(x);
}
template <int N>
__device__ __forceinline__ void bp_wait_vm() {
    static_assert(N == 0 || N == 1 || N == 2 || N == 3 || N == 4 || N == 6 || N == 9, "add the literal");
    if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
}
template <int N>
using int_c = std::integral_constant<int, N>;

// The VALU work of one tile as a list of single instructions in issue order, dealt evenly over the MFMA slots of a step:
// element k (0..15) of the lane's 16 scores: X scale (0), E exponential (1), M multiply by dP - D (2); pair i: C pack dS (3),
// CP pack P (4, dK/dV only).  Round r holds X(r), E(r-1), M(r-2) and the packs whose inputs are two rounds old, so every
// instruction has two or three independent ones between itself and its producer.
struct BpOp { int kind, idx; };
template <bool DKDV>
struct BpSched { BpOp ops[96]; int n; };
template <bool DKDV>
constexpr BpSched<DKDV> bp_make_sched() {
    BpSched<DKDV> s{};
    int n = 0;
    for (int r = 0; r < 19; ++r) {
        if (r < 16) s.ops[n++] = BpOp{0, r};
        if (r >= 1 && r - 1 < 16) s.ops[n++] = BpOp{1, r - 1};
        if (r >= 2 && r - 2 < 16) s.ops[n++] = BpOp{2, r - 2};
        if (DKDV && r >= 3 && ((r - 3) & 1) == 0 && (r - 3) / 2 < 8) s.ops[n++] = BpOp{4, (r - 3) / 2};
        if (r >= 4 && ((r - 4) & 1) == 0 && (r - 4) / 2 < 8) s.ops[n++] = BpOp{3, (r - 4) / 2};
    }
    s.n = n;
    return s;
}
template <bool DKDV>
inline constexpr BpSched<DKDV> bp_sched = bp_make_sched<DKDV>();
template <int N0, class F, int... I>
__device__ __forceinline__ void bp_for(F&& f, std::integer_sequence<int, I...>) {
    (f(int_c<N0 + I>{}), ...);
}

template <bool DKDV>
__global__ __launch_bounds__(BP_WAVES * 64, BP_WAVES == 4 ? 2 : 1) void attn_bwd_pipe_kernel(const AttnBwdParams p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int NA = DKDV ? 8 : 4;        // accumulating MFMA slots per step
    constexpr int NS = NA + 8;              // MFMA slots per step
    constexpr int NB = (BP_WAVES == 4 ? 2 : 1) + (DKDV ? 1 : 0);        // DMA instructions per wave and tile
    constexpr int VS0 = 1;                  // first MFMA slot with VALU work in its shadow
    constexpr int NOPS = bp_sched<DKDV>.n;
    static_assert(NS % 4 == 0 && BP_AHEAD >= 2 && BP_AHEAD <= 3, "operand staging wraps around the step");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hi = lane >> 5;
    const int n_own = DKDV ? p.Skv : p.Sq, n_str = DKDV ? p.Sq : p.Skv;
    int blk, h, b;
    xcd_local_bh((n_own + BP_OWN - 1) / BP_OWN, p.H, (int)gridDim.x, p.xcd_local, blk, h, b);
    const int own0 = blk * BP_OWN + wave * 32;
    const int64_t bh = (int64_t)b * p.H + h;
    // own side (B operands: lane = own row) and streamed side (A operands through LDS)
    const bf16_t* b0p = (DKDV ? p.k + (int64_t)b * p.bsk : p.q + (int64_t)b * p.bsq) + h * 64;
    const bf16_t* b1p = (DKDV ? p.v + (int64_t)b * p.bsv : p.d_o + (int64_t)b * p.bsdo) + h * 64;
    const int64_t ld_b0 = DKDV ? p.ldk : p.ldq, ld_b1 = DKDV ? p.ldv : p.lddo;
    const bf16_t* x0p = (DKDV ? p.q + (int64_t)b * p.bsq : p.k + (int64_t)b * p.bsk) + h * 64;
    const bf16_t* x1p = (DKDV ? p.d_o + (int64_t)b * p.bsdo : p.v + (int64_t)b * p.bsv) + h * 64;
    const uint32_t ld0b = (uint32_t)((DKDV ? p.ldq : p.ldk) * 2), ld1b = (uint32_t)((DKDV ? p.lddo : p.ldv) * 2);   // row pitch, bytes
    const float* vecp = p.vec + bh * p.nb32 * 64;

    bf16x8_t bf0[4], bf1[4];
    const int own_r = min(own0 + ql, n_own - 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        bf0[ks] = *reinterpret_cast<const bf16x8_t*>(b0p + (int64_t)own_r * ld_b0 + ks * 16 + hi * 8);
        bf1[ks] = *reinterpret_cast<const bf16x8_t*>(b1p + (int64_t)own_r * ld_b1 + ks * 16 + hi * 8);
    }
    // dQ: L and D of the lane's query
    float nl = 0.f;
    f32x16 negD16;
    {
        float nd = 0.f;
        if constexpr (!DKDV) {
            nl = -p.lse[bh * p.Sq + own_r];
            nd = vecp[(own_r >> 5) * 64 + 32 + (own_r & 31)];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) negD16[r] = nd;
    }
    const float c = p.scale_log2e;

    // ---- DMA: one instruction per wave and operand fills LDS rows 8 wave .. 8 wave + 7 of a tile
    const int R = (wave & 3) * 8 + (lane >> 3), pch = lane & 7;
    const uint32_t sw = (uint32_t)((pch ^ ((((R >> 1) & 1) << 2) | ((R >> 2) & 3))) << 4);      // source chunk of LDS chunk pch
    const uint32_t x0_lo = (uint32_t)R * ld0b + sw, x1_lo = (uint32_t)R * ld1b + sw;
    const uint32_t lds0 = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)(smem)) + (wave & 3) * 1024;
    const uint32_t ldsv = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)(smem)) + BP_VEC;
    const uint32_t lane4 = lane * 4;
    auto dma16 = [&](const char* base, uint32_t off, uint32_t lds) __attribute__((always_inline)) {
        if constexpr ((BP_ABL & 8) != 0) return;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory");
    };
    auto dma4 = [&](const char* base, uint32_t off, uint32_t lds) __attribute__((always_inline)) {
        if constexpr ((BP_ABL & 8) != 0) return;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory");
    };
    const int nt = (n_str + BP_ROWS - 1) / BP_ROWS;
    // the next tile to request -> ring slot `slot`; the source bases advance by one tile per call (scalar adds only).  A ragged
    // last tile clamps the rows past the end to the last row (kept on a branch of its own: the asm comment stops if-conversion).
    const char* nxt0 = reinterpret_cast<const char*>(x0p);
    const char* nxt1 = reinterpret_cast<const char*>(x1p);
    const char* nxtv = reinterpret_cast<const char*>(vecp);
    const int64_t step0 = (int64_t)BP_ROWS * ld0b, step1 = (int64_t)BP_ROWS * ld1b;
    const bool ragged = nt * BP_ROWS > n_str;
    auto stage = [&](int t, int slot) __attribute__((always_inline)) {
        if constexpr (BP_WAVES == 4) {
            if (ragged && t == nt - 1) {
                asm volatile("; ragged tile" ::: "memory");
                const uint32_t r = (uint32_t)min(R, n_str - 1 - t * BP_ROWS);
                dma16(nxt0, r * ld0b + sw, lds0 + slot * BP_TILE);
                dma16(nxt1, r * ld1b + sw, lds0 + BP_RING1 + slot * BP_TILE);
            } else {
                dma16(nxt0, x0_lo, lds0 + slot * BP_TILE);
                dma16(nxt1, x1_lo, lds0 + BP_RING1 + slot * BP_TILE);
            }
        } else {                                   // waves 0-3 bring the first operand's tile, waves 4-7 the second's
            const char* src = wave < 4 ? nxt0 : nxt1;
            const uint32_t ldb = wave < 4 ? ld0b : ld1b, lds = lds0 + (wave < 4 ? 0 : BP_RING1) + slot * BP_TILE;
            if (ragged && t == nt - 1) {
                asm volatile("; ragged tile" ::: "memory");
                dma16(src, (uint32_t)min(R, n_str - 1 - t * BP_ROWS) * ldb + sw, lds);
            } else {
                dma16(src, wave < 4 ? x0_lo : x1_lo, lds);
            }
        }
        if constexpr (DKDV) dma4(nxtv, lane4, ldsv + slot * 256);
        nxt0 += step0;
        nxt1 += step1;
        nxtv += 256;
    };

    // ---- fragment read offsets (bytes inside a tile)
    const int fq = (((ql >> 1) & 1) << 2) | ((ql >> 2) & 3);
    int r_off[4];                 // row fragment: row ql, chunk 2 ks + hi
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) r_off[ks] = ql * 128 + (((2 * ks + hi) ^ fq) << 4);
    // transposed fragment (16 streamed rows kk, 32 d of block db): this lane addresses row 4 hi + j (+ 8 for the second read),
    // 8 bytes at logical byte db*64 + ((lane >> 4) & 1)*32 + (lane & 3)*8
    int t_off[2][2];
    {
        const int j = (lane & 15) >> 2, u = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
        const int f1 = (((j >> 1) & 1) << 2) | hi;              // f(4 hi + j) ; f(4 hi + j + 8) = f1 ^ 2
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            t_off[db][0] = (4 * hi + j) * 128 + ((((db * 4 + u) ^ f1)) << 4) + (lane & 1) * 8;
            t_off[db][1] = (4 * hi + j + 8) * 128 + ((((db * 4 + u) ^ f1 ^ 2)) << 4) + (lane & 1) * 8;
        }
    }
    const int v_off = BP_VEC + 16 * hi;
    auto rfrag = [&](int off, int ks) __attribute__((always_inline)) {
        if constexpr ((BP_ABL & 4) != 0) return bf0[ks];
        return *reinterpret_cast<const bf16x8_t*>(smem + r_off[ks] + off);
    };
    auto tfrag = [&](int off, int db) __attribute__((always_inline)) {
        if constexpr ((BP_ABL & 4) != 0) return bf0[db];
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + t_off[db][0] + off));
        const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + t_off[db][1] + off));
        const s16x8 both = __builtin_shufflevector(lo, up, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8_t, both);
    };
    // the -L/c (which = 0) or -D (1) values of the 32 streamed queries of a tile, in the accumulator's row order
    auto vec_init = [&](f32x16& dst, int voff, int which) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(smem + v_off + voff + which * 128 + 32 * i);
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[4 * i + e] = v[e];
        }
    };

    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x16 sc[2], dp[2];          // score sets: lane = own row, register r = streamed row 8 (r >> 2) + 4 hi + (r & 3)
    u32x4 dsf[2][2], pf[2][2];    // packed bf16 dS (and P) of a tile: [set][16-row step kk]
    f32x16 acc0[2], acc1[2];      // dQ^T / dK^T and dV^T: d = db*32 + 8 (r >> 2) + 4 hi + (r & 3)
    acc0[0] = zero16; acc0[1] = zero16; acc1[0] = zero16; acc1[1] = zero16;
    bf16x8_t a[4];                // staged A operands
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = bf0[i];

    // ---- prologue: tiles 0 .. 3 requested while the own fragments are still on their way, ONE wait for everything (the
    // compiler's own wait for the fragments would be a vmcnt(0) anyway: it does not see the DMA instructions), scores of tile 0
    stage(0, 0);
    if (nt > 1) stage(1, 1);
    if (nt > 2) stage(2, 2);
    if (nt > 3) stage(3, 3);
    bp_wait_vm<0>();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { asm volatile("" ::"v"(bf0[ks])); asm volatile("" ::"v"(bf1[ks])); }
    asm volatile("" ::"v"(nl), "v"(negD16));
    __builtin_amdgcn_s_barrier();
    if constexpr (DKDV) { vec_init(sc[0], 0, 0); vec_init(dp[0], 0, 1); }
    else { sc[0] = zero16; dp[0] = negD16; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        sc[0] = BP_MFMA(rfrag(0, ks), bf0[ks], sc[0]);
        dp[0] = BP_MFMA(rfrag(BP_RING1, ks), bf1[ks], dp[0]);
    }

    // ---- one step: g = SL (mod 8); VALU on tile g (score set SL & 1), accumulating products of tile g - 1, scores of tile g + 1
    auto step = [&](auto sl_tag, auto first_tag, auto last_tag, int g) __attribute__((always_inline)) {
        constexpr int SL = decltype(sl_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;      // LAST: no tile g + 1
        constexpr int CUR = SL & 1, NXT = CUR ^ 1;
        constexpr int T_PREV = ((SL + 7) & 7) * BP_TILE, T_NEXT = ((SL + 1) & 7) * BP_TILE, T_CUR = SL * BP_TILE;
        constexpr int V_NEXT = ((SL + 1) & 7) * 256;
        if constexpr ((BP_ABL & 1) == 0) {
            // needed now: tile g + 1 (requested in step g - 3); still in flight: tiles g + 2, g + 3
            if (g + 3 < nt) bp_wait_vm<2 * NB>();
            else if (g + 2 < nt) bp_wait_vm<NB>();
            else bp_wait_vm<0>();
            __builtin_amdgcn_s_barrier();
        }
        if constexpr (BP_DMA_SLOT < 0) {
            if (g + 4 < nt) stage(g + 4, (SL + 4) & 7);        // the slot of tile g - 4
        }
        if constexpr (!DKDV) {
            if (ragged && g == nt - 1) {                       // keys past the end of the sequence: P = 0
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (g * BP_ROWS + 8 * (r >> 2) + 4 * hi + (r & 3) >= n_str) sc[CUR][r] = -INFINITY;
            }
        }
        // LDS operand of MFMA slot s of THIS step (s < NS) or of the next one (s >= NS: its first slots read tile g)
        auto operand = [&](auto s_tag) __attribute__((always_inline)) {
            constexpr int s = decltype(s_tag)::value % NS;
            constexpr int tp = decltype(s_tag)::value >= NS ? T_CUR : T_PREV;
            if constexpr (s < NA) {
                constexpr int kk = (s & 3) >> 1, db = s & 1;
                constexpr int ring = (DKDV && s < 4) ? BP_RING1 : 0;        // dK/dV: slots 0-3 dV (A = dO^T), 4-7 dK (A = Q^T)
                return tfrag(ring + tp + kk * 2048, db);
            } else {
                constexpr int ks = (s - NA) >> 1, which = (s - NA) & 1;
                return rfrag(which * BP_RING1 + T_NEXT, ks);
            }
        };
        auto needs_operand = [](int s) constexpr { return s < NA ? !FIRST : !LAST; };
        if constexpr (FIRST) {           // (later steps find their first operands staged by the step before)
            if constexpr (NA - BP_AHEAD <= 0) a[0] = operand(int_c<NA>{});
        }
        float xv[16], ev[16];
        auto valu = [&](auto n_tag) __attribute__((always_inline)) {
            constexpr BpOp op = bp_sched<DKDV>.ops[decltype(n_tag)::value];
            constexpr int k = op.idx;
            if constexpr (op.kind == 0) {
                if constexpr (DKDV) xv[k] = sc[CUR][k] * c;
                else xv[k] = __builtin_fmaf(sc[CUR][k], c, nl);
                BP_PIN(xv[k]);
            } else if constexpr (op.kind == 1) {
                ev[k] = bp_exp2(xv[k]);
                BP_PIN(ev[k]);
            } else if constexpr (op.kind == 2) {
                xv[k] = ev[k] * dp[CUR][k];
                BP_PIN(xv[k]);
            } else if constexpr (op.kind == 3) {
                dsf[CUR][k >> 2][k & 3] = bp_cvt_pk(xv[2 * k], xv[2 * k + 1]);
                BP_PIN(dsf[CUR][k >> 2][k & 3]);
            } else {
                pf[CUR][k >> 2][k & 3] = bp_cvt_pk(ev[2 * k], ev[2 * k + 1]);
                BP_PIN(pf[CUR][k >> 2][k & 3]);
            }
        };
        auto slot = [&](auto s_tag) __attribute__((always_inline)) {
            constexpr int s = decltype(s_tag)::value;
            BP_SB();
            if constexpr (s < NA) {
                if constexpr (!FIRST) {
                    constexpr int kk = (s & 3) >> 1, db = s & 1;
                    if constexpr (DKDV && s < 4) acc1[db] = BP_MFMA(a[s & 3], __builtin_bit_cast(bf16x8_t, pf[NXT][kk]), acc1[db]);
                    else acc0[db] = BP_MFMA(a[s & 3], __builtin_bit_cast(bf16x8_t, dsf[NXT][kk]), acc0[db]);
                }
            } else if constexpr (s < NS && !LAST) {
                constexpr int ks = (s - NA) >> 1, which = (s - NA) & 1;
                if constexpr (which == 0) {
                    if constexpr (ks == 0 && !DKDV) sc[NXT] = BP_MFMA(a[s & 3], bf0[0], zero16);
                    else sc[NXT] = BP_MFMA(a[s & 3], bf0[ks], sc[NXT]);
                } else {
                    if constexpr (ks == 0 && !DKDV) dp[NXT] = BP_MFMA(a[s & 3], bf1[0], negD16);
                    else dp[NXT] = BP_MFMA(a[s & 3], bf1[ks], dp[NXT]);
                }
            }
            if constexpr (s < NS && needs_operand(s + BP_AHEAD)) a[(s + BP_AHEAD) & 3] = operand(int_c<s + BP_AHEAD>{});
            if constexpr (BP_DMA_SLOT >= 0 && s == BP_DMA_SLOT) {
                if (g + 4 < nt) stage(g + 4, (SL + 4) & 7);
            }
            if constexpr (DKDV && !LAST) {        // accumulator start values of tile g + 1
                if constexpr (s == 0) vec_init(sc[NXT], V_NEXT, 0);
                if constexpr (s == 1) vec_init(dp[NXT], V_NEXT, 1);
            }
            // this slot's share of the VALU list (the empty asm statements keep each instruction in its slot, see attention_pipe.hip)
            if constexpr (s >= VS0) {
                constexpr int n0 = (s - VS0) * NOPS / (NS - VS0), n1 = (s - VS0 + 1) * NOPS / (NS - VS0);
                bp_for<n0>(valu, std::make_integer_sequence<int, n1 - n0>{});
            }
        };
        slot(int_c<0>{}); slot(int_c<1>{}); slot(int_c<2>{}); slot(int_c<3>{});
        slot(int_c<4>{}); slot(int_c<5>{}); slot(int_c<6>{}); slot(int_c<7>{});
        slot(int_c<8>{}); slot(int_c<9>{}); slot(int_c<10>{}); slot(int_c<11>{});
        if constexpr (NS > 12) { slot(int_c<12>{}); slot(int_c<13>{}); slot(int_c<14>{}); slot(int_c<15>{}); }
        BP_SB();
    };

    typedef std::integral_constant<bool, true> yes_t;
    typedef std::integral_constant<bool, false> no_t;
    // (a LAST instantiation of the step without the score products of a tile that does not exist costs registers on every path:
    // 246 / 256 VGPRs + 344 bytes of scratch with the eight extra copies, measured 13 % slower; the last step multiplies a stale
    // ring slot instead and nobody reads the result)
    step(int_c<0>{}, yes_t{}, no_t{}, 0);
    int g = 1;
    for (; g + 8 <= nt; g += 8) {
        step(int_c<1>{}, no_t{}, no_t{}, g);     step(int_c<2>{}, no_t{}, no_t{}, g + 1);
        step(int_c<3>{}, no_t{}, no_t{}, g + 2); step(int_c<4>{}, no_t{}, no_t{}, g + 3);
        step(int_c<5>{}, no_t{}, no_t{}, g + 4); step(int_c<6>{}, no_t{}, no_t{}, g + 5);
        step(int_c<7>{}, no_t{}, no_t{}, g + 6); step(int_c<0>{}, no_t{}, no_t{}, g + 7);
    }
    // (g = 1 (mod 8) here) the up to seven left-over steps
    if (g < nt) { step(int_c<1>{}, no_t{}, no_t{}, g); ++g; }
    if (g < nt) { step(int_c<2>{}, no_t{}, no_t{}, g); ++g; }
    if (g < nt) { step(int_c<3>{}, no_t{}, no_t{}, g); ++g; }
    if (g < nt) { step(int_c<4>{}, no_t{}, no_t{}, g); ++g; }
    if (g < nt) { step(int_c<5>{}, no_t{}, no_t{}, g); ++g; }
    if (g < nt) { step(int_c<6>{}, no_t{}, no_t{}, g); ++g; }
    if (g < nt) { step(int_c<7>{}, no_t{}, no_t{}, g); ++g; }

    // ---- the accumulating products of the last tile
    {
        const int toff = ((nt - 1) & 7) * BP_TILE;
        auto last = [&](const u32x4 (&ds_)[2], const u32x4 (&p_)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    acc0[db] = BP_MFMA(tfrag(toff + kk * 2048, db), __builtin_bit_cast(bf16x8_t, ds_[kk]), acc0[db]);
                    if constexpr (DKDV)
                        acc1[db] = BP_MFMA(tfrag(BP_RING1 + toff + kk * 2048, db), __builtin_bit_cast(bf16x8_t, p_[kk]), acc1[db]);
                }
        };
        if ((nt - 1) & 1) last(dsf[1], pf[1]);
        else last(dsf[0], pf[0]);
    }

    // ---- epilogue: scale, bounce each wave's 32 x 64 bf16 tile through LDS (the rings are dead after the barrier), store whole
    // 128-byte rows, 16 bytes per lane
    __syncthreads();
    auto store = [&](const f32x16 (&acc)[2], float scl, char* ob, bf16_t* out) __attribute__((always_inline)) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint2 pk;
                pk.x = bp_cvt_pk(acc[db][4 * i] * scl, acc[db][4 * i + 1] * scl);
                pk.y = bp_cvt_pk(acc[db][4 * i + 2] * scl, acc[db][4 * i + 3] * scl);
                *reinterpret_cast<uint2*>(ob + ql * 128 + (((db * 4 + i) ^ (ql & 7)) << 4) + hi * 8) = pk;
            }
        // (each wave reads back only what it wrote itself: no barrier, the LDS accesses of one wave are ordered)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int r = ps * 8 + (lane >> 3), cc = lane & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(ob + r * 128 + ((cc ^ (r & 7)) << 4));
            const int row = own0 + r;
            if (row < n_own) *reinterpret_cast<uint4*>(out + (int64_t)b * p.bsdq + (int64_t)row * p.lddq + h * 64 + cc * 8) = v;
        }
    };
    if constexpr (DKDV) {
        store(acc0, p.scale, smem + wave * 4096, p.dk);
        store(acc1, 1.0f, smem + BP_RING1 + wave * 4096, p.dv);
    } else {
        store(acc0, p.scale, smem + wave * 4096, p.dq);
    }
}

}  // namespace

int attention_bwd_pipe_launch(const AttnBwdParams& p, int B, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_pipe_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, BP_LDS_DQ);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_pipe_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, BP_LDS_DKDV);
        attr_set = true;
    }
    const int64_t nq = (int64_t)((p.Sq + BP_OWN - 1) / BP_OWN) * p.H * B, nk = (int64_t)((p.Skv + BP_OWN - 1) / BP_OWN) * p.H * B;
    ADVGRPO_CHECK(nq < (1ll << 31) && nk < (1ll << 31), "attention_bwd: grid too large");
    hipLaunchKernelGGL(attn_bwd_pipe_kernel<false>, dim3((unsigned)nq), dim3(BP_WAVES * 64), BP_LDS_DQ, s, p);
    ADVGRPO_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_bwd_pipe_kernel<true>, dim3((unsigned)nk), dim3(BP_WAVES * 64), BP_LDS_DKDV, s, p);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

}  // namespace advgrpo
