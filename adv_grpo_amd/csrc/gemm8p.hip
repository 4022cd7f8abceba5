// gemm8p.hip -- 256 x 256 x 64 eight-phase bf16 MFMA GEMM for gfx950 (the wide Linears of the MMDiT: M = 16384 image rows
// (+ 3280 text rows), N, K in {1536, 4608, 6144}; reference call sites sd3_pipeline_with_logprob_fast.py:630-637,
// train_sd3_fast_pickscore.py:235-255 -> diffusers' SD3Transformer2DModel Linears).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )      same operands, fragment layout and fused epilogue as gemm.hip
//
// Why a second kernel: the two-workgroup tiles of gemm.hip (192x128, 128x128) need 0.036 LDS bytes per flop and wait for
// the slowest L2 miss of the next k-tile once per 64-deep step; they converge at ~1 PFLOP/s.  This one is the structure
// the CDNA4 guide measures at 1.3-1.5 PFLOP/s on random data:
//   * one 512-thread workgroup per CU, tile 256 x 256, wave grid 2 (M) x 4 (N): each wave owns 128 x 64 of C = 128
//     accumulator VGPRs, i.e. 0.023 LDS bytes per flop;
//   * the two wave groups (wr = 0 / 1; waves w and w + 4 share a SIMD) run ONE BARRIER APART: while a group is in a
//     16-MFMA compute segment the other is in its LDS-read + DMA-issue segment, so on every SIMD the matrix pipe and
//     the LDS / VMEM paths are busy at the same time (s_setprio 1 around the MFMAs);
//   * a k-tile (64 deep) is four phases, one 64 x 32 quadrant of the wave tile x K = 64 each; fragments stay in
//     registers across phases (80 VGPRs), 4 / 4 / 8 / 8 ds_read_b128 per phase;
//   * LDS = a ring of 8 item slots x 16 KiB (2 k-tile buffers x {A sub 0, A sub 1, W sub 0, W sub 1}); an item is what
//     every wave reads in ONE phase (A sub s = rows s*64..+63 of both wave rows, W sub s = columns s*32..+31 of the four
//     wave columns), so a slot is dead right after that phase.  global_load_lds_dwordx4 fills one item per phase (2
//     instructions per wave, lane-linear 8-row x 128-byte pieces, XOR swizzle on the per-lane SOURCE chunk and on the
//     read: conflict-free b128 lane groups), TWO phases after the slot's last read and SIX phases before its first:
//     five items (80 KiB per CU) are in flight at every counted s_waitcnt vmcnt(10); raw s_barrier.
//     Hazards are separated by construction, not by timing (groups run one barrier apart):
//       RAW  an item is waited for (vmcnt, by every issuing wave) before the barrier that closes the load segment of
//            phase y and first read in phase y + 1;
//       WAR  reads of phase q complete (lgkmcnt(0)) right after the barrier that closes its load segment; the slot is
//            re-filled in the load segment of phase q + 2, i.e. after two more barriers.
//   * persistent: grid = min(tiles, CUs); a workgroup walks tiles id, id + grid, ... in the XCD-grouped order of
//     gemm.hip (each XCD works on a near-square patch of C per round); two problems can share the launch (GemmPair).
#include <stdlib.h>

#include "gemm_device.hpp"

namespace advgrpo {

namespace {

constexpr int P8_BM = 256, P8_BN = 256, P8_BK = 64;
constexpr int P8_HALF = 128 * P8_BK * 2;          // one half-tile image: 128 rows x 128 B
constexpr int P8_BUF = 4 * P8_HALF;               // A0 A1 W0 W1
constexpr int P8_RING = 2 * P8_BUF;               // 128 KiB
constexpr int P8_SCRATCH = 4 * 16 * 64 * 4;       // per-wave epilogue scratch: 4 slabs of 16 rows x 64 f32, XOR-swizzled; aliases the ring
constexpr int P8_LDS = P8_RING;

// One LDS-DMA instruction: 64 lanes x 16 bytes from (uniform base + per-lane 32-bit offset) to LDS [m0 .. m0 + 1 KiB).
// Hand-written because the builtin form keeps every per-lane source as a 64-bit VGPR pair and re-adds the k offset on the
// VALU: 24 VGPRs and 8 v_lshl_add_u64 per k-tile more than the SGPR-base form, enough to spill inside the main loop.
__device__ __forceinline__ void p8_dma16(const char* base, uint32_t off, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ uint32_t lds_addr(const char* p) {
    return (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)(p));
}

// Fused epilogue, compact form.  Same arithmetic, in the same order, as gemm_epilogue_rows (gemm_device.hpp) -- results are
// bit-identical to the other tile variants -- but the eight 16-row slabs of the 128 x 64 wave tile are walked by a real
// loop (the fully unrolled form is ~100k instructions for this tile: the instruction fetch alone cost 29 us per tile).
// Slabs are bounced through the wave's private 16 KiB of the (by then dead) DMA ring, four at a time: f32 rows of 256 B,
// 16-byte chunk c of row r at slot c ^ (r & 7) -- conflict-free for the ds_write_b128 of the accumulator layout (8 lanes =
// 8 rows, one chunk) and for the ds_read_b128 of the row layout.
__device__ __forceinline__ void p8_epilogue(const GemmParams& p, f32x4 (&acc)[8][4], int mw0, int nw0, int lane, char* scratch) {
    const int mrow = lane & 15, q = lane >> 4;
    const int orow_l = lane >> 3, c8 = (lane & 7) * 8;
    const int n = nw0 + c8;
    auto unpack8 = [](const uint4& v, float (&f)[8]) __attribute__((always_inline)) {
        f[0] = bf2f((bf16_t)(v.x & 0xffffu)); f[1] = bf2f((bf16_t)(v.x >> 16));
        f[2] = bf2f((bf16_t)(v.y & 0xffffu)); f[3] = bf2f((bf16_t)(v.y >> 16));
        f[4] = bf2f((bf16_t)(v.z & 0xffffu)); f[5] = bf2f((bf16_t)(v.z >> 16));
        f[6] = bf2f((bf16_t)(v.w & 0xffffu)); f[7] = bf2f((bf16_t)(v.w >> 16));
    };
    auto pack8 = [](const float (&v)[8]) __attribute__((always_inline)) {
        uint4 pk;
        pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        pk.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
        pk.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
        return pk;
    };
    auto out_row = [&](int m) __attribute__((always_inline)) -> int64_t {
        if (p.seg_rows <= 0) return m;
        const int bidx = m / p.seg_rows;
        return (int64_t)bidx * p.seg_stride + p.seg_off + (m - bidx * p.seg_rows);
    };
    float bias8[8];
    if (p.bias && n < p.N) unpack8(*reinterpret_cast<const uint4*>(p.bias + n), bias8);
    // residual rows: requested one slab ahead of their use
    auto load_res = [&](int i, uint4 (&r)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int m = mw0 + i * 16 + ps * 8 + orow_l;
            r[ps] = uint4{0u, 0u, 0u, 0u};
            if (p.residual && m < p.M && n < p.N) r[ps] = *reinterpret_cast<const uint4*>(p.residual + out_row(m) * p.ldr + n);
        }
    };
    uint4 res_cur[2], res_nxt[2];
    load_res(0, res_cur);
    const int w_off = mrow * 256, w_sw = mrow & 7;
    // Accumulators may only be indexed statically (a runtime-indexed accumulator array goes to scratch memory), so each
    // half of the wave tile (4 slabs) is written to LDS with static indices and the fused arithmetic then runs as a runtime
    // loop over the slabs it finds there.
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(scratch + sl * 4096 + w_off + (((j * 4 + q) ^ w_sw) << 4)) = acc[half * 4 + sl][j];
#pragma unroll 1
    for (int sl = 0; sl < 4; ++sl) {
        const int i = half * 4 + sl;
        const char* slab = scratch + sl * 4096;
        if (i + 1 < 8) load_res(i + 1, res_nxt);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int row = ps * 8 + orow_l;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + row * 256 + ((((lane & 7) * 2) ^ (row & 7)) << 4));
            const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + row * 256 + ((((lane & 7) * 2 + 1) ^ (row & 7)) << 4));
            const int m = mw0 + i * 16 + row;
            if (m >= p.M || n >= p.N) continue;
            const int64_t orow = out_row(m);
            float v[8] = {lo[0] * p.alpha, lo[1] * p.alpha, lo[2] * p.alpha, lo[3] * p.alpha,
                          hi[0] * p.alpha, hi[1] * p.alpha, hi[2] * p.alpha, hi[3] * p.alpha};
            if (p.bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bias8[e];
            }
            if (p.rms_w) {   // QK-norm: the wave tile's 64 columns are one head, its row sits in 8 adjacent lanes
                const int hh = n >> 6;
                float sq = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = round_bf16(v[e]);
                    sq += v[e] * v[e];
                }
                sq += __shfl_xor(sq, 1, 64);
                sq += __shfl_xor(sq, 2, 64);
                sq += __shfl_xor(sq, 4, 64);
                if (hh < p.rms_nheads) {
                    const float rs = rsqrtf(sq * (1.0f / 64.0f) + p.rms_eps);
                    if (p.rms_rs_out && (lane & 7) == 0) p.rms_rs_out[orow * p.rms_nheads + hh] = rs;
                    float w8[8];
                    unpack8(*reinterpret_cast<const uint4*>(p.rms_w + (hh / p.rms_hpw) * 64 + c8), w8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = round_bf16(v[e] * rs) * w8[e];
                }
            }
            if (p.aux_out) *reinterpret_cast<uint4*>(p.aux_out + orow * p.ld_aux + n) = pack8(v);
            if (p.act >= ACT_DGELU_TANH) {
                float z[8];
                unpack8(*reinterpret_cast<const uint4*>(p.aux_in + orow * p.ld_aux + n), z);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dact_fn(z[e], p.act);
            } else if (p.act != ACT_NONE) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = act_fn(v[e], p.act);
            }
            if (p.gate) {
                const int gb = p.gate_rows > 0 ? m / p.gate_rows : 0;
                float g[8];
                unpack8(*reinterpret_cast<const uint4*>(p.gate + (int64_t)gb * p.gate_stride + n), g);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= g[e];
            }
            if (p.residual) {
                float r[8];
                unpack8(res_cur[ps], r);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r[e];
            }
            const int64_t o = orow * p.ldc + n;
            if (p.out_dtype == ADVGRPO_BF16) {
                *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.C) + o) = pack8(v);
            } else {
                float* c = reinterpret_cast<float*>(p.C) + o;
                *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
        res_cur[0] = res_nxt[0];
        res_cur[1] = res_nxt[1];
    }
    }
}

}  // namespace

struct P8Sched { int tiles_a, tiles_total; };

template <bool PAIR>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const GemmPair pp, const P8Sched sc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;      // wave row (group) / wave column

    // per-lane pieces that do not depend on the tile
    const int lrow = lane >> 3;                   // row inside an 8-row DMA instruction
    const int schunk = (lane & 7) ^ lrow;         // pre-swizzled source chunk of this lane's LDS slot
    int frag_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) frag_off[ks] = (lane & 15) * 128 + (((ks * 4 + (lane >> 4)) ^ (lane & 7)) << 4);
    // this wave's fragment bases inside a k-tile buffer
    const int a_base = wr * 64 * 128;                     // + sub * P8_HALF + i * 16 * 128
    const int b_base = 2 * P8_HALF + wc * 32 * 128;       // + sub * P8_HALF + j * 16 * 128

    const int nwg = gridDim.x;
    for (int tile = blockIdx.x; tile < sc.tiles_total; tile += nwg) {
        // ---- which problem / which tile: every round of nwg tiles is dealt XCD-contiguously
        const int round0 = tile - (int)blockIdx.x;
        const int in_round = min(nwg, sc.tiles_total - round0);
        int id = round0 + xcd_remap(blockIdx.x, in_round);
        const bool second = PAIR && id >= sc.tiles_a;
        const GemmParams& p = second ? pp.b : pp.a;
        if (second) id -= sc.tiles_a;
        const int tiles_n = (p.N + P8_BN - 1) / P8_BN, tiles_m = (p.M + P8_BM - 1) / P8_BM;
        int tile_m, tile_n;
        tile_coords(id, tiles_m, tiles_n, 4, tile_m, tile_n);
        const int m0 = tile_m * P8_BM, n0 = tile_n * P8_BN;

        // ---- DMA sources.  A ring item is what every wave reads in ONE phase: item "A sub s" = rows {s*64 .. s*64+63} of
        // both 128-row wave-row halves, item "W sub s" = columns {s*32 .. s*32+31} of the four 64-wide wave columns;
        // slot row r of the 128-row item image belongs to wave row r >> 6 (A) / wave column r >> 5 (W).
        // (32-bit byte offsets from the uniform operand bases: the DMA takes "SGPR base + VGPR offset", 8 VGPRs instead of
        // 16 for the eight per-lane sources -- with 64-bit pointers the main loop spilled one and drained vmcnt to reload it)
        uint32_t a_off[2][2], b_off[2][2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int sr = (wave + it * 8) * 8 + lrow;                    // slot row of this lane's 16 bytes
                int r = m0 + (sr >> 6) * 128 + sub * 64 + (sr & 63);
                r = r < p.M ? r : p.M - 1;
                int64_t ar = r;
                if (p.a_seg_rows > 0) {
                    const int bi = r / p.a_seg_rows;
                    ar = (int64_t)bi * p.a_seg_stride + p.a_seg_off + (r - bi * p.a_seg_rows);
                }
                a_off[sub][it] = (uint32_t)((ar * p.lda + schunk * 8) * 2);
                int n = n0 + (sr >> 5) * 64 + sub * 32 + (sr & 31);
                n = n < p.N ? n : p.N - 1;
                b_off[sub][it] = (uint32_t)(((int64_t)n * p.ldw + schunk * 8) * 2);
            }
        const char* a_bytes = reinterpret_cast<const char*>(p.A);
        const char* w_bytes = reinterpret_cast<const char*>(p.W);
        const int nk = p.K / P8_BK;
        const int n_items = 4 * nk;                       // half-tiles of this output tile, in issue (= first-read) order
        // which = 0,1: A sub 0 / 1; 2,3: W sub 0 / 1
        auto stage = [&](int buf, int which, int kt) __attribute__((always_inline)) {
            if (kt >= nk) return;
            char* base = smem + buf * P8_BUF + which * P8_HALF;
            const char* src = (which < 2 ? a_bytes : w_bytes) + kt * (P8_BK * 2);   // uniform
            const uint32_t* off = which < 2 ? a_off[which] : b_off[which - 2];
#pragma unroll
            for (int it = 0; it < 2; ++it) p8_dma16(src, off[it], lds_addr(base + (wave + it * 8) * 1024));
        };
        // wait until at most `allowed` half-tiles (2 DMA instructions each) of this wave are still in flight
        auto wait_inflight = [&](int allowed) __attribute__((always_inline)) {
            if (allowed >= 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (allowed == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (allowed == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (allowed == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (allowed == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };

        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- prologue: the seven half-tiles the steady state would have issued before phase 0 of k-tile 0, in its
        // order (A0 W0 W1 A1 of k-tile 0, A0 W0 W1 of k-tile 1); the first two must have landed for the first reads
        stage(0, 0, 0); stage(0, 2, 0); stage(0, 3, 0); stage(0, 1, 0);
        stage(1, 0, 1); stage(1, 2, 1); stage(1, 3, 1);
        wait_inflight(min(7, n_items) - 2);
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();       // stagger: group 1 runs one barrier behind

        bf16x8_t af0[2][4], af1[2][4], b0[2][2], b1[2][2];
        auto read_a = [&](const char* buf, int sub, bf16x8_t (&a)[2][4]) __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a[ks][i] = *reinterpret_cast<const bf16x8_t*>(buf + a_base + sub * P8_HALF + i * 16 * 128 + frag_off[ks]);
        };
        auto read_b = [&](const char* buf, int sub, bf16x8_t (&b)[2][2]) __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    b[ks][j] = *reinterpret_cast<const bf16x8_t*>(buf + b_base + sub * P8_HALF + j * 16 * 128 + frag_off[ks]);
        };
        // end of a load segment at global phase g (= 4 kt + ph): everything first read in phase g + 1 must have landed
        // (items 0 .. g + 2 of the issue order); 8 + g items have been issued (capped by n_items)
        auto close_load = [&](int g, bool steady) __attribute__((always_inline)) {
            if (steady) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else wait_inflight(min(8 + g, n_items) - min(g + 3, n_items));
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
#define P8_COMPUTE(MH, NH, AF, BF)                                                                                      \
        do {                                                                                                            \
            __builtin_amdgcn_s_setprio(1);                                                                              \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                            \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                           \
                    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
                        acc[(MH) * 4 + i][(NH) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                      \
                            BF[ks][j], AF[ks][i], acc[(MH) * 4 + i][(NH) * 2 + j], 0, 0, 0);                            \
            __builtin_amdgcn_s_setprio(0);                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
            __builtin_amdgcn_s_barrier();                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
        } while (0)

        // Ring of 8 half-tile slots (2 k-tile buffers x {A0 A1 W0 W1}), one slot re-filled per phase, TWO phases after
        // its last read; every half-tile is issued six phases before its first read, so five half-tiles (80 KiB per CU)
        // are in flight at each counted wait.  Phases of k-tile kt (buffer cur), fragments kept in registers:
        //   ph0  read W sub 0            fill A1(kt+1)   compute (A0, W0)        [A sub 0 was read in ph3 of kt-1]
        //   ph1  read W sub 1            fill A0(kt+2)   compute (A0, W1)
        //   ph2  read A sub 1            fill W0(kt+2)   compute (A1, W1)
        //   ph3  read A sub 0 of kt+1    fill W1(kt+2)   compute (A1, W0)
        auto ktile = [&](int cur, int kt, bool steady) __attribute__((always_inline)) {
            const char* buf = smem + cur * P8_BUF;
            const char* nxt = smem + (cur ^ 1) * P8_BUF;
            const int g = 4 * kt;
            read_b(buf, 0, b0);
            stage(cur ^ 1, 1, kt + 1);
            close_load(g, steady);
            P8_COMPUTE(0, 0, af0, b0);
            read_b(buf, 1, b1);
            stage(cur, 0, kt + 2);
            close_load(g + 1, steady);
            P8_COMPUTE(0, 1, af0, b1);
            read_a(buf, 1, af1);
            stage(cur, 2, kt + 2);
            close_load(g + 2, steady);
            P8_COMPUTE(1, 1, af1, b1);
            if (kt + 1 < nk) read_a(nxt, 0, af0);
            stage(cur, 3, kt + 2);
            close_load(g + 3, steady);
            P8_COMPUTE(1, 0, af1, b0);
        };
        read_a(smem, 0, af0);
        int kt = 0;
        for (; kt + 3 < nk; kt += 2) {                   // both k-tiles issue all of their re-fills: constant waits
            ktile(0, kt, true);
            ktile(1, kt + 1, true);
        }
        for (; kt < nk; kt += 2) {
            ktile(0, kt, false);
            if (kt + 1 < nk) ktile(1, kt + 1, false);
        }
#undef P8_COMPUTE
        if (wr == 0) __builtin_amdgcn_s_barrier();       // matches group 1's stagger barrier
        p8_epilogue(p, acc, m0 + wr * 128, n0 + wc * 64, lane, smem + wave * P8_SCRATCH);
        __syncthreads();                                  // (ring re-use by the next tile's prologue: every wave is out of its main loop)
    }
}

// every operand of the fused epilogue can be accessed as aligned 16-byte row segments (the dispatcher asks before choosing)
bool gemm8p_ok(const GemmParams& p) {
    auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    return p.batch == 1 && p.splitk == 1 && !p.conv && (p.N & 7) == 0 &&
           ((p.ldc | p.ldr | p.gate_stride | p.ld_aux) & 7) == 0 && a16(p.C) && a16(p.bias) && a16(p.gate) &&
           a16(p.residual) && a16(p.aux_out) && a16(p.aux_in) && a16(p.rms_w);
}

namespace {

int prepare(GemmParams& p) {
    ADVGRPO_CHECK(p.A && p.W && p.C, "gemm8p: null operand");
    ADVGRPO_CHECK(p.M > 0 && p.N > 0 && p.K > 0 && p.K % 64 == 0, "gemm8p: need M,N>0 and K %% 64 == 0 (M=%d N=%d K=%d)", p.M,
                  p.N, p.K);
    ADVGRPO_CHECK(p.lda % 8 == 0 && p.ldw % 8 == 0, "gemm8p: lda/ldw must be multiples of 8 elements");
    ADVGRPO_CHECK(p.batch == 1 && p.splitk == 1 && !p.conv, "gemm8p: plain (unbatched, unsplit) problems only");
    {
        const int64_t a_rows = p.a_seg_rows > 0 ? ((int64_t)((p.M - 1) / p.a_seg_rows) * p.a_seg_stride + p.a_seg_off + p.a_seg_rows) : p.M;
        ADVGRPO_CHECK(a_rows * p.lda * 2 < (1ll << 32) && (int64_t)p.N * p.ldw * 2 < (1ll << 32),
                      "gemm8p: operands must span less than 4 GiB (32-bit DMA offsets)");
    }
    ADVGRPO_CHECK(gemm8p_ok(p), "gemm8p: the row epilogue needs N %% 8 == 0 and 16-byte aligned rows of every operand");
    return 0;
}

template <bool PAIR>
int launch8p(const GemmPair& pp, const P8Sched& sc, hipStream_t s) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
            set_error("gemm8p: cannot query the device");
            return -2;
        }
        cus = prop.multiProcessorCount;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8p_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  P8_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8p_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  P8_LDS);
    }
    const int grid = sc.tiles_total < cus ? sc.tiles_total : cus;
    hipLaunchKernelGGL((gemm8p_kernel<PAIR>), dim3(grid), dim3(512), P8_LDS, s, pp, sc);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

int tiles_of(const GemmParams& p) { return ((p.M + P8_BM - 1) / P8_BM) * ((p.N + P8_BN - 1) / P8_BN); }

}  // namespace

int gemm8p_launch(const GemmParams& p_in, hipStream_t s) {
    GemmPair pp{};
    pp.a = p_in;
    if (prepare(pp.a)) return -1;
    pp.tiles_a = tiles_of(pp.a);
    return launch8p<false>(pp, P8Sched{pp.tiles_a, pp.tiles_a}, s);
}

int gemm8p_launch_pair(const GemmParams& a, const GemmParams& b, hipStream_t s) {
    GemmPair pp{};
    pp.a = a; pp.b = b;
    if (prepare(pp.a) || prepare(pp.b)) return -1;
    pp.tiles_a = tiles_of(pp.a);
    return launch8p<true>(pp, P8Sched{pp.tiles_a, pp.tiles_a + tiles_of(pp.b)}, s);
}

}  // namespace advgrpo
