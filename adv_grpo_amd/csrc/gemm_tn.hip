// gemm_tn.hip -- token-contracted ("TN") bf16 GEMM for the LoRA weight gradients of the G-step.
//
//   C[n1, n2] += alpha * sum_m P[row_p(m), n1] * Q[row_q(m), n2]          (f32 atomic accumulation)
//
// replaces, for every adapted Linear, `loss.backward()`'s weight-gradient products of PEFT's LoRA layer
// (scripts/train_sd3_fast_pickscore.py:490-511 adapter setup, :1165 backward):
//   dB[out, r] = s * dY^T (X A^T)      -> P = dY (wide), Q = X A^T (64 wide)
//   dA[r, in]  = s * (dY B)^T X        -> P = X  (wide), Q = dY B  (64 wide), written transposed.
// Both operands are stored token-major ([M, features]) and the contraction runs over the tokens, i.e. over the ROW
// index of both.  The K-major MFMA fragments are produced by transposing on the LDS read
// (ds_read_b64_tr_b16, the same read the attention kernel uses for V): no transposed copies of the 50-150 MB
// activation / gradient matrices are ever written (the first version of the G-step spent 7 % of its time in a
// transpose kernel and ran these products as split-K NT GEMMs on the copies).
//
// Tile: 128 (n1) x 64 (n2) per workgroup, 4 waves of 32 x 64, 64 tokens per LDS stage; LDS-DMA into a 3-slot ring that
// stays in flight across the workgroup barrier (counted s_waitcnt vmcnt + raw s_barrier, as in the attention kernel),
// two workgroups per CU.  grid.y slices the token range; each slice writes its partial 128 x 64 tile to a workspace
// and a second tiny kernel adds the slices into the gradient accumulator: f32 atomics from ~40 slices onto the same
// 8192 addresses serialise in the L2 (measured 13 us per million atomic adds, 4x the time of the whole pass over P).
// HBM-bound: one pass over P (M x N1 bf16) per call; the MFMA work is 2*M*N1*64 flop.
#include "common.hpp"
#include "gemm.hpp"

namespace advgrpo {

typedef __attribute__((ext_vector_type(4))) float tn_f32x4;
typedef __attribute__((ext_vector_type(4))) short tn_s16x4;
typedef __attribute__((ext_vector_type(8))) short tn_s16x8;
typedef __attribute__((address_space(3))) void* tn_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* tn_gptr_t;

struct TnParams {
    const bf16_t* P; int64_t ldp; int p_seg_rows; int64_t p_seg_stride, p_seg_off;
    const bf16_t* Q; int64_t ldq; int q_seg_rows; int64_t q_seg_stride, q_seg_off;
    float* ws;                                      // [slices][N1][64] partial tiles
    int M, N1; int chunks_per_block;                // 64-token chunks per grid.y slice
};

// row of token m through the (seg_rows, seg_stride, seg_off) map of the GEMM family
__device__ __forceinline__ int64_t tn_row(int m, int seg_rows, int64_t seg_stride, int64_t seg_off) {
    if (seg_rows <= 0) return m;
    const int b = m / seg_rows;
    return (int64_t)b * seg_stride + seg_off + (m - b * seg_rows);
}

// per-lane source cursor of one DMA instruction: pointer of the current token's row + position inside its segment
struct TnCursor {
    const bf16_t* ptr; int rem;
    __device__ __forceinline__ void init(const bf16_t* base, int64_t ld, int m, int seg_rows, int64_t seg_stride,
                                         int64_t seg_off) {
        ptr = base + tn_row(m, seg_rows, seg_stride, seg_off) * ld;
        rem = seg_rows > 0 ? m % seg_rows : 0;
    }
    // advance by 64 tokens without a division (segments are at least a few tokens long: loop, normally 0 or 1 trips)
    __device__ __forceinline__ void advance(int64_t ld, int seg_rows, int64_t seg_stride) {
        ptr += 64 * ld;
        if (seg_rows > 0) {
            rem += 64;
            while (rem >= seg_rows) { rem -= seg_rows; ptr += (seg_stride - seg_rows) * ld; }
        }
    }
};

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TnParams p) {
    constexpr int MC = 64, NS = 3;               // tokens per stage, ring depth
    constexpr int P_BYTES = MC * 256, Q_BYTES = MC * 128, STAGE = P_BYTES + Q_BYTES;   // 16 KiB + 8 KiB
    constexpr int LOADS = 6;                     // DMA instructions per wave per stage
    __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, t = lane & 15;
    const int n1_0 = blockIdx.x * 128;
    const int chunk0 = blockIdx.y * p.chunks_per_block;
    const int nchunks_total = (p.M + MC - 1) / MC;
    const int nch = min(p.chunks_per_block, nchunks_total - chunk0);
    if (nch <= 0) return;

    // DMA: P instruction j (0..15) covers tokens 4j..4j+3 (256-byte rows), this wave issues j = wave + 4*i;
    //      Q instruction j (0..7) covers tokens 8j..8j+7 (128-byte rows), this wave issues j = wave + 4*i.
    TnCursor pc[4], qc[2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        pc[i].init(p.P + n1_0 + (lane & 15) * 8, p.ldp, min(chunk0 * MC + (wave + i * 4) * 4 + (lane >> 4), p.M - 1),
                   p.p_seg_rows, p.p_seg_stride, p.p_seg_off);
#pragma unroll
    for (int i = 0; i < 2; ++i)
        qc[i].init(p.Q + (lane & 7) * 8, p.ldq, min(chunk0 * MC + (wave + i * 4) * 8 + (lane >> 3), p.M - 1), p.q_seg_rows,
                   p.q_seg_stride, p.q_seg_off);
    auto stage = [&](int slot, int chunk) __attribute__((always_inline)) {   // chunks are staged in increasing order
        char* base = smem + slot * STAGE;
        const bool ragged = (chunk + 1) * MC > p.M;                           // only the last chunk of the matrix
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16_t* src = pc[i].ptr;
            if (ragged) {
                const int m = min(chunk * MC + (wave + i * 4) * 4 + (lane >> 4), p.M - 1);
                src = p.P + n1_0 + (lane & 15) * 8 + tn_row(m, p.p_seg_rows, p.p_seg_stride, p.p_seg_off) * p.ldp;
            }
            __builtin_amdgcn_global_load_lds((tn_gptr_t)src, (tn_lds_ptr_t)(base + (wave + i * 4) * 1024), 16, 0, 0);
            pc[i].advance(p.ldp, p.p_seg_rows, p.p_seg_stride);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16_t* src = qc[i].ptr;
            if (ragged) {
                const int m = min(chunk * MC + (wave + i * 4) * 8 + (lane >> 3), p.M - 1);
                src = p.Q + (lane & 7) * 8 + tn_row(m, p.q_seg_rows, p.q_seg_stride, p.q_seg_off) * p.ldq;
            }
            __builtin_amdgcn_global_load_lds((tn_gptr_t)src, (tn_lds_ptr_t)(base + P_BYTES + (wave + i * 4) * 1024), 16, 0, 0);
            qc[i].advance(p.ldq, p.q_seg_rows, p.q_seg_stride);
        }
    };

    tn_f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = tn_f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed fragment reads: inside a 16-lane group, lane t supplies the address of 4 contiguous features of token
    // (g*4 + t/4) and receives feature t of tokens g*4 .. g*4+3; a second read 16 tokens further completes the 8
    // contraction values of the lane.  P and Q use the same token permutation, so the products pair up correctly.
    const int tok = g * 4 + (t >> 2);
    const int p_off = tok * 256 + (wave * 32) * 2 + (t & 3) * 8;
    const int q_off = tok * 128 + (t & 3) * 8;

    stage(0, chunk0);
    if (nch > 1) stage(1, chunk0 + 1);
    int slot = 0;
    for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int fill = slot == 0 ? NS - 1 : slot - 1;      // slot of chunk c-1: free once every wave is past the barrier
        if (c + 2 < nch) stage(fill, chunk0 + c + 2);
        const char* pb = smem + slot * STAGE;
        const char* qb = pb + P_BYTES;
        slot = slot == NS - 1 ? 0 : slot + 1;
        const int m0 = (chunk0 + c) * MC;
        const bool ragged = m0 + MC > p.M;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            bf16x8_t pf[2], qf[4];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const char* a = pb + kc * 32 * 256 + p_off + nb * 32;
                const tn_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(a));
                const tn_s16x4 hi =
                    __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(a + 16 * 256));
                tn_s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                if (ragged) {   // tokens past M were clamped on the way in: zero their contribution (P side suffices)
                    asm volatile("; ragged chunk" ::: "memory");
                    const int mb = m0 + kc * 32 + g * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (mb + e >= p.M) both[e] = 0;
                        if (mb + 16 + e >= p.M) both[4 + e] = 0;
                    }
                }
                pf[nb] = __builtin_bit_cast(bf16x8_t, both);
            }
#pragma unroll
            for (int qn = 0; qn < 4; ++qn) {
                const char* a = qb + kc * 32 * 128 + q_off + qn * 32;
                const tn_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(a));
                const tn_s16x4 hi =
                    __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(a + 16 * 128));
                qf[qn] = __builtin_bit_cast(bf16x8_t, (tn_s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int qn = 0; qn < 4; ++qn)
                    acc[nb][qn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[qn], pf[nb], acc[nb][qn], 0, 0, 0);
        }
    }
    // lane holds partial C[n1 = n1_0 + wave*32 + nb*16 + t][n2 = qn*16 + g*4 .. +3]
    float* ws = p.ws + (int64_t)blockIdx.y * p.N1 * 64;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int n1 = n1_0 + wave * 32 + nb * 16 + t;
#pragma unroll
        for (int qn = 0; qn < 4; ++qn)
            *reinterpret_cast<tn_f32x4*>(ws + (int64_t)n1 * 64 + qn * 16 + g * 4) = acc[nb][qn];
    }
}

// C (+)= alpha * sum over slices of the partial tiles
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float* __restrict__ ws, int slices, int N1,
                                                              float* __restrict__ C, int64_t ldc, int transpose_out,
                                                              float alpha) {
    const int i = blockIdx.x * 256 + threadIdx.x;        // element (n1, n2) of the [N1, 64] tile
    if (i >= N1 * 64) return;
    float s = 0.f;
    for (int k = 0; k < slices; ++k) s += ws[(int64_t)k * N1 * 64 + i];
    const int n1 = i >> 6, n2 = i & 63;
    float* c = transpose_out ? C + (int64_t)n2 * ldc + n1 : C + (int64_t)n1 * ldc + n2;
    *c += s * alpha;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Grouped form (round 5): every token-contracted product of one adapter group -- the dB products of up to six adapters, or the dA
// products of the adapters that share an input -- in ONE launch, and no reduce kernel.
//   * a launch carries up to TN_MAX problems (advgrpo_tn_desc); a workgroup finds its (problem, 128-column tile, token slice) from
//     the problems' workgroup prefix (wave-uniform scan over at most TN_MAX entries);
//   * the dA products of the adapters that share an input X (the three of a fused q | k | v projection) are three problems of the SAME
//     launch: their workgroups run side by side and X comes from HBM once, from L2 / the Infinity Cache twice.  (A 192-wide Q -- one
//     pass over X against [u_q | u_k | u_v], three outputs -- was built and measured: 110 us for the two dA problems of a group at
//     config 2 beside 68 us for its six dB problems, against 118 us for all TWELVE as 64-wide problems of one launch; scripts/bench_tn.py.
//     Its 40 KiB stages allow fewer workgroups per CU and its 96 KiB partial tiles make the last workgroup's sum the long pole.)
//   * slices of the token range meet in the workspace, and the LAST workgroup to arrive at a tile (agent-scope counter) adds the
//     slices in slice order into C: a fixed summation order whoever arrives last -- bitwise reproducible, no float atomics -- and the
//     separate reduce launch is gone.  XCD L2s are not coherent with each other: partial tiles are written and read with agent-scope
//     (sc1) accesses, the counter is bumped after the stores have completed (vmcnt(0) + workgroup barrier), and the last workgroup
//     leaves the counter at zero for the next launch.
constexpr int TN_MAX = 12;
struct TnProblem {
    const bf16_t* P; const bf16_t* Q; float* C;
    int64_t ldp, p_seg_stride, p_seg_off, ldq, q_seg_stride, q_seg_off, ldc, ws_off;
    int p_seg_rows, q_seg_rows, transpose_out, M, N1, tiles, slices, chunks_per_block, wg0, cnt_off;
    float alpha;
};
struct TnGroup { TnProblem pr[TN_MAX]; int n; float* ws; int* counters; };

typedef unsigned long long tn_u64;
typedef __attribute__((ext_vector_type(2))) float tn_f32x2;
__device__ __forceinline__ void tn_st16_sc1(float* p, const tn_f32x4& v) {
    tn_u64* q = reinterpret_cast<tn_u64*>(p);
    const tn_f32x2 lo{v[0], v[1]}, hi{v[2], v[3]};
    __hip_atomic_store(q, __builtin_bit_cast(tn_u64, lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, __builtin_bit_cast(tn_u64, hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ tn_f32x4 tn_ld16_sc1(const float* p) {
    tn_u64* q = reinterpret_cast<tn_u64*>(const_cast<float*>(p));
    const tn_f32x2 lo = __builtin_bit_cast(tn_f32x2, __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const tn_f32x2 hi = __builtin_bit_cast(tn_f32x2, __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    return tn_f32x4{lo[0], lo[1], hi[0], hi[1]};
}

// One DMA instruction of the grouped kernel: 64 lanes x 16 bytes from (uniform base + per-lane 32-bit byte offset) to LDS [lds, lds + 1 KiB).
// Hand-written (as in attention.hpp / gemm8p_kernel.hpp): issued through the builtin, the compiler treats the DMA as a possible alias of
// every later LDS read and waits with vmcnt(0) in front of the first fragment read of the SAME iteration -- the ring then hides nothing
// (the per-adapter kernel above: 1.4 TB/s).
__device__ __forceinline__ void tn_dma16(const void* base, uint32_t off, uint32_t lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory");
}
// A lane's source cursor: byte offset of its token's row (+ column) from the operand base, and the token's position in its row segment.
// Advancing by MC tokens is division-free.
struct TnCur {
    uint32_t off; int rem;
};
__device__ __forceinline__ void tn_cur_init(TnCur& c, int m, int64_t ld, int col, int seg_rows, int64_t seg_stride, int64_t seg_off) {
    c.off = (uint32_t)((tn_row(m, seg_rows, seg_stride, seg_off) * ld + col) * 2);
    c.rem = seg_rows > 0 ? m % seg_rows : 0;
}
__device__ __forceinline__ void tn_cur_advance(TnCur& c, int step, uint32_t step_bytes, int seg_rows, uint32_t jump_bytes) {
    c.off += step_bytes;
    c.rem += step;
    while (c.rem >= seg_rows) {           // (0 or 1 trips for the hot path's 205- / 1024-token segments; short segments wrap more than once)
        c.off += jump_bytes;
        c.rem -= seg_rows;
    }
}

__global__ __launch_bounds__(256, 2) void gemm_tn_grouped_kernel(const TnGroup grp) {
    constexpr int NQB = 1;                       // Q is 64 columns wide
    constexpr int MC = 64, NS = 3, NQ = 64 * NQB, KC = MC / 32;      // 64 tokens (24 KiB) per stage, three stages, two workgroups per CU
    constexpr int P_BYTES = MC * 256, Q_ROW = 128 * NQB, Q_BYTES = MC * Q_ROW, STAGE = P_BYTES + Q_BYTES;
    constexpr int PI = P_BYTES / 4096, QI = Q_BYTES / 4096;            // DMA instructions per wave per stage (1 KiB each, 4 waves)
    constexpr int LOADS = PI + QI;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, t = lane & 15;
    // ---- which problem, tile, slice
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < grp.n; ++i)
        if ((int)blockIdx.x >= grp.pr[i].wg0) pi = i;
    pi = __builtin_amdgcn_readfirstlane(pi);
    const TnProblem& p = grp.pr[pi];
    const int local = (int)blockIdx.x - p.wg0;
    const int tile = local % p.tiles, slice = local / p.tiles;
    const int n1_0 = tile * 128;
    const int M = p.M;
    const int m_begin = slice * p.chunks_per_block * 64;               // (chunks_per_block counts 64-token chunks whatever MC is)
    const int m_end = min(M, m_begin + p.chunks_per_block * 64);
    const int nch = (m_end - m_begin + MC - 1) / MC;                   // >= 1 by construction of `slices`

    // ---- DMA sources.  The stage image is token-major: P rows of 256 bytes (this tile's 128 columns), Q rows of Q_ROW bytes.
    // Instruction j of an operand fills LDS bytes [1024 j, 1024 j + 1024): lane l -> byte 1024 j + 16 l -> token byte / row, column.
    // This wave issues j = wave + 4 i.  A token past the end of the matrix reads the matrix's last row (its products are zeroed).
    const int p_sr = p.p_seg_rows > 0 ? p.p_seg_rows : 0x7fffffff, q_sr = p.q_seg_rows > 0 ? p.q_seg_rows : 0x7fffffff;
    const uint32_t p_step = (uint32_t)(MC * p.ldp * 2), q_step = (uint32_t)(MC * p.ldq * 2);
    const uint32_t p_jump = p.p_seg_rows > 0 ? (uint32_t)((p.p_seg_stride - p.p_seg_rows) * p.ldp * 2) : 0u;
    const uint32_t q_jump = p.q_seg_rows > 0 ? (uint32_t)((p.q_seg_stride - p.q_seg_rows) * p.ldq * 2) : 0u;
    TnCur pc[PI], qc[QI];
    int p_tok[PI], q_tok[QI];
    uint32_t p_last[PI], q_last[QI];                                   // the same column of the matrix's last row
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const int byte = 1024 * (wave + 4 * i) + 16 * lane;
        p_tok[i] = byte >> 8;
        const int col = n1_0 + ((byte & 255) >> 1);
        tn_cur_init(pc[i], min(m_begin + p_tok[i], M - 1), p.ldp, col, p.p_seg_rows, p.p_seg_stride, p.p_seg_off);
        p_last[i] = (uint32_t)((tn_row(M - 1, p.p_seg_rows, p.p_seg_stride, p.p_seg_off) * p.ldp + col) * 2);
    }
#pragma unroll
    for (int i = 0; i < QI; ++i) {
        const int byte = 1024 * (wave + 4 * i) + 16 * lane;
        q_tok[i] = byte / Q_ROW;
        const int col = (byte - q_tok[i] * Q_ROW) >> 1;
        tn_cur_init(qc[i], min(m_begin + q_tok[i], M - 1), p.ldq, col, p.q_seg_rows, p.q_seg_stride, p.q_seg_off);
        q_last[i] = (uint32_t)((tn_row(M - 1, p.q_seg_rows, p.q_seg_stride, p.q_seg_off) * p.ldq + col) * 2);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)(smem)) + wave * 1024;
    auto stage = [&](int slot, int m0) __attribute__((always_inline)) {      // stages are requested in increasing token order
        const uint32_t base = lds0 + slot * STAGE;
        const bool ragged = m0 + MC > M;                                      // (uniform; only the last stage of the matrix)
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const uint32_t off = ragged && m0 + p_tok[i] >= M ? p_last[i] : pc[i].off;
            tn_dma16(p.P, off, base + i * 4096);
            tn_cur_advance(pc[i], MC, p_step, p_sr, p_jump);
        }
#pragma unroll
        for (int i = 0; i < QI; ++i) {
            const uint32_t off = ragged && m0 + q_tok[i] >= M ? q_last[i] : qc[i].off;
            tn_dma16(p.Q, off, base + P_BYTES + i * 4096);
            tn_cur_advance(qc[i], MC, q_step, q_sr, q_jump);
        }
    };

    tn_f32x4 acc[2][4 * NQB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4 * NQB; ++j) acc[i][j] = tn_f32x4{0.f, 0.f, 0.f, 0.f};
    // transposed fragment reads: inside a 16-lane group, lane t supplies the address of 4 contiguous features of token
    // (g*4 + t/4) and receives feature t of tokens g*4 .. g*4+3; a second read 16 tokens further completes the 8
    // contraction values of the lane.  P and Q use the same token permutation, so the products pair up correctly.
    const int tok = g * 4 + (t >> 2);
    const int p_off = tok * 256 + (wave * 32) * 2 + (t & 3) * 8;
    const int q_off = tok * Q_ROW + (t & 3) * 8;

    stage(0, m_begin);
    if (nch > 1) stage(1, m_begin + MC);
    int slot = 0;
    for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int fill = slot == 0 ? NS - 1 : slot - 1;      // slot of stage c-1: free once every wave is past the barrier
        if (c + 2 < nch) stage(fill, m_begin + (c + 2) * MC);
        const char* pb = smem + slot * STAGE;
        const char* qb = pb + P_BYTES;
        slot = slot == NS - 1 ? 0 : slot + 1;
        const int m0 = m_begin + c * MC;
        const bool ragged = m0 + MC > m_end;                 // tokens of the next slice / past the matrix: zero their contribution
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            bf16x8_t pf[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const char* a = pb + kc * 32 * 256 + p_off + nb * 32;
                const tn_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(a));
                const tn_s16x4 hi =
                    __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(a + 16 * 256));
                tn_s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                if (ragged) {
                    asm volatile("; ragged stage" ::: "memory");
                    const int mb = m0 + kc * 32 + g * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (mb + e >= m_end) both[e] = 0;
                        if (mb + 16 + e >= m_end) both[4 + e] = 0;
                    }
                }
                pf[nb] = __builtin_bit_cast(bf16x8_t, both);
            }
#pragma unroll
            for (int qn = 0; qn < 4 * NQB; ++qn) {
                const char* a = qb + kc * 32 * Q_ROW + q_off + qn * 32;
                const tn_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(a));
                const tn_s16x4 hi =
                    __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(a + 16 * Q_ROW));
                const bf16x8_t qf = __builtin_bit_cast(bf16x8_t, (tn_s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[nb][qn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, pf[nb], acc[nb][qn], 0, 0, 0);
            }
        }
    }
    int* const s_last_p = reinterpret_cast<int*>(smem);           // (the ring is idle from here on; every wave is past its last read
    __syncthreads();                                              //  once it has passed this barrier)
#define s_last (*s_last_p)
    // ---- partial tile -> workspace (agent scope), then the arrival counter of the tile.  lane holds
    //      partial C[n1 = n1_0 + wave*32 + nb*16 + t][n2 = qn*16 + g*4 .. +3]
    float* ws = grp.ws + p.ws_off;
    float* mine = ws + (int64_t)slice * p.N1 * NQ;
    if (p.slices > 1) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int n1 = n1_0 + wave * 32 + nb * 16 + t;
#pragma unroll
            for (int qn = 0; qn < 4 * NQB; ++qn) tn_st16_sc1(mine + (int64_t)n1 * NQ + qn * 16 + g * 4, acc[nb][qn]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's stores have completed
        __syncthreads();                                                // ... every wave's
        if (tid == 0) {
            int* cnt = grp.counters + p.cnt_off + tile;
            // release / acquire in the memory model's own terms as well: the sc1 stores above already leave the XCD's L2 and
            // the sc1 loads below bypass this CU's L1 (the guide's "16-B sc1 stores AND sc1 loads" hand-off), but the ordering of the
            // counter against them must not rest on instruction forms alone
            const int before = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            s_last = before == p.slices - 1;
            if (s_last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // zero again for the next launch
            }
        }
        __syncthreads();
        if (!s_last) return;
    }
    // ---- the last workgroup at this tile: C (+)= alpha * (slice 0 + slice 1 + ...), in slice order.  Slice by slice, with the lane's
    // 8 NQB chunk loads of a slice issued together (a first version walked chunk by chunk, slice by slice inside: 8 NQB x slices
    // dependent round trips of an agent-scope load each -- 300 - 440 us per launch, most of it this loop in 24 workgroups)
    if (p.slices > 1) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int qn = 0; qn < 4 * NQB; ++qn)
                acc[nb][qn] = tn_ld16_sc1(ws + (int64_t)(n1_0 + wave * 32 + nb * 16 + t) * NQ + qn * 16 + g * 4);
        for (int k = 1; k < p.slices; ++k) {
            const float* wk = ws + (int64_t)k * p.N1 * NQ;
            tn_f32x4 v[2][4 * NQB];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int qn = 0; qn < 4 * NQB; ++qn)
                    v[nb][qn] = tn_ld16_sc1(wk + (int64_t)(n1_0 + wave * 32 + nb * 16 + t) * NQ + qn * 16 + g * 4);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int qn = 0; qn < 4 * NQB; ++qn) acc[nb][qn] += v[nb][qn];
        }
    }
    // C += alpha * sum: every load of C first, then every store (written as `C[..] += ..` per element the compiler has to assume that a
    // store aliases the next load and serialises 8 - 96 memory round trips per lane: 120 of the 130 us of the 192-wide launch)
    tn_f32x4 cur[2][4 * NQB];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int n1 = n1_0 + wave * 32 + nb * 16 + t;
#pragma unroll
        for (int qn = 0; qn < 4 * NQB; ++qn) {
            const float* C = p.C;
            const int n2 = (qn & 3) * 16 + g * 4;
            if (p.transpose_out) {
#pragma unroll
                for (int e = 0; e < 4; ++e) cur[nb][qn][e] = C[(int64_t)(n2 + e) * p.ldc + n1];
            } else {
                cur[nb][qn] = *reinterpret_cast<const tn_f32x4*>(C + (int64_t)n1 * p.ldc + n2);
            }
        }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int n1 = n1_0 + wave * 32 + nb * 16 + t;
#pragma unroll
        for (int qn = 0; qn < 4 * NQB; ++qn) {
            const tn_f32x4 out = cur[nb][qn] + acc[nb][qn] * p.alpha;
            float* C = p.C;
            const int n2 = (qn & 3) * 16 + g * 4;
            if (p.transpose_out) {
#pragma unroll
                for (int e = 0; e < 4; ++e) C[(int64_t)(n2 + e) * p.ldc + n1] = out[e];
            } else {
                *reinterpret_cast<tn_f32x4*>(C + (int64_t)n1 * p.ldc + n2) = out;
            }
        }
    }
}

#undef s_last

}  // namespace advgrpo

using namespace advgrpo;

extern "C" int64_t advgrpo_gemm_tn_workspace_bytes(int M, int N1) {
    const int nchunks = (M + 63) / 64;
    return (int64_t)(nchunks < 1 ? 1 : nchunks) * N1 * 64 * 4;   // upper bound: one slice per chunk
}

extern "C" int advgrpo_gemm_tn_f32acc(const void* P, int64_t ldp, int p_seg_rows, int64_t p_seg_stride, int64_t p_seg_off,
                                      const void* Q, int64_t ldq, int q_seg_rows, int64_t q_seg_stride, int64_t q_seg_off,
                                      float* C, int64_t ldc, int transpose_out, int M, int N1, int N2, float alpha,
                                      void* workspace, void* stream) {
    ADVGRPO_CHECK(P && Q && C && workspace && M > 0, "gemm_tn: bad argument");
    ADVGRPO_CHECK(N2 == 64 && N1 > 0 && N1 % 128 == 0, "gemm_tn: needs N2 == 64 and N1 %% 128 == 0 (N1=%d N2=%d)", N1, N2);
    ADVGRPO_CHECK(ldp % 8 == 0 && ldq % 8 == 0 && (reinterpret_cast<uintptr_t>(P) & 15) == 0 &&
                      (reinterpret_cast<uintptr_t>(Q) & 15) == 0 && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
                  "gemm_tn: operands must be 16-byte aligned with pitches that are multiples of 8");
    ADVGRPO_CHECK((p_seg_rows == 0 || p_seg_rows >= 8) && (q_seg_rows == 0 || q_seg_rows >= 8), "gemm_tn: bad row segments");
    TnParams p{};
    p.P = (const bf16_t*)P; p.ldp = ldp; p.p_seg_rows = p_seg_rows; p.p_seg_stride = p_seg_stride; p.p_seg_off = p_seg_off;
    p.Q = (const bf16_t*)Q; p.ldq = ldq; p.q_seg_rows = q_seg_rows; p.q_seg_stride = q_seg_stride; p.q_seg_off = q_seg_off;
    p.ws = (float*)workspace; p.M = M; p.N1 = N1;
    const int nchunks = (M + 63) / 64, tiles = N1 / 128;
    // ~1 workgroup per CU (measured best of 128..1024: more slices cost more partial-tile traffic than they hide latency)
    int target = 256;
#ifdef ADVGRPO_EXPERIMENTS
    { static int knob = -1; if (knob < 0) { const char* e = getenv("ADVGRPO_TN_BLOCKS"); knob = e ? atoi(e) : 0; } if (knob > 0) target = knob; }
#endif
    int slices = (target + tiles - 1) / tiles;
    if (slices > (nchunks + 1) / 2) slices = (nchunks + 1) / 2;
    if (slices < 1) slices = 1;
    p.chunks_per_block = (nchunks + slices - 1) / slices;
    slices = (nchunks + p.chunks_per_block - 1) / p.chunks_per_block;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, slices), dim3(256), 0, s, p);
    ADVGRPO_LAUNCH_CHECK();
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((N1 * 64 + 255) / 256), dim3(256), 0, s, (const float*)workspace, slices,
                       N1, C, ldc, transpose_out, alpha);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}

// one advgrpo_tn_desc (include/advgrpo.h) -> TnProblem; returns the workgroups it takes
constexpr int TN_MAX_SLICES = 8;               // token slices per problem: what the last workgroup at a tile has to add up
constexpr int64_t TN_CNT_BYTES = 4096;          // arrival counters at the front of the workspace (1024 tiles per launch)

extern "C" int64_t advgrpo_gemm_tn_grouped_workspace_bytes(const advgrpo_tn_desc* d, int n) {
    int64_t f = 0;
    for (int i = 0; i < n; ++i) {
        const int nchunks = (d[i].M + 63) / 64;
        const int slices = nchunks < 64 ? (nchunks < 1 ? 1 : nchunks) : 64;     // (64: room for the EXPERIMENTS build's slice knob)        // upper bound of what the launcher picks
        f += (int64_t)slices * d[i].N1 * 64;
    }
    return f * 4 + TN_CNT_BYTES;
}

extern "C" int advgrpo_gemm_tn_grouped(const advgrpo_tn_desc* d, int n, void* workspace, int64_t workspace_bytes,
                                       int workspace_is_zeroed, void* stream) {
    ADVGRPO_CHECK(d && n > 0 && n <= TN_MAX && workspace, "gemm_tn_grouped: 1..%d problems and a workspace", TN_MAX);
    constexpr int NQ = 64;
    TnGroup g{};
    g.n = n;
    int64_t chunks_total = 0;
    for (int i = 0; i < n; ++i) {
        const advgrpo_tn_desc& a = d[i];
        ADVGRPO_CHECK(a.P && a.Q && a.C && a.M > 0, "gemm_tn_grouped: problem %d: bad argument", i);
        ADVGRPO_CHECK(a.N1 > 0 && a.N1 % 128 == 0, "gemm_tn_grouped: N1 %% 128 == 0 (N1=%d)", a.N1);
        ADVGRPO_CHECK(a.ldp % 8 == 0 && a.ldq % 8 == 0 && (reinterpret_cast<uintptr_t>(a.P) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.Q) & 15) == 0,
                      "gemm_tn_grouped: operands must be 16-byte aligned with pitches that are multiples of 8");
        ADVGRPO_CHECK((a.p_seg_rows == 0 || a.p_seg_rows >= 8) && (a.q_seg_rows == 0 || a.q_seg_rows >= 8), "gemm_tn_grouped: bad row segments");
        {   // 32-bit byte offsets from the operand bases
            auto last_row = [](int M, int sr, int64_t ss, int64_t so) { return sr > 0 ? (int64_t)((M - 1) / sr) * ss + so + (M - 1) % sr : (int64_t)M - 1; };
            ADVGRPO_CHECK((last_row(a.M, a.p_seg_rows, a.p_seg_stride, a.p_seg_off) + 1) * a.ldp * 2 < (1ll << 32) &&
                              (last_row(a.M, a.q_seg_rows, a.q_seg_stride, a.q_seg_off) + 1) * a.ldq * 2 < (1ll << 32),
                          "gemm_tn_grouped: problem %d: an operand spans 4 GiB or more (32-bit byte offsets)", i);
        }
        ADVGRPO_CHECK(a.transpose_out || (a.ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0), "gemm_tn_grouped: C rows must be 16-byte aligned");
        chunks_total += (int64_t)(a.M + 63) / 64 * (a.N1 / 128);
    }
    // ~768 workgroups per launch (NQB = 1: two per CU), at most TN_MAX_SLICES token slices per problem and at least 8 chunks (512 tokens)
    // per slice
    // (same-box sweep at the config-2 shapes, scripts/bench_tn.py: 4 / 8 / 16 / 32 slices -> 77 / 70 / 100 / 104 us for the six dB problems of a
    //  q | k | v group; 384 .. 1536 target workgroups within 5 %)
    int64_t target = 768;
    int max_slices = TN_MAX_SLICES;
#ifdef ADVGRPO_EXPERIMENTS
    { const char* e = getenv("ADVGRPO_TNG_TARGET"); if (e && atoi(e) > 0) target = atoi(e); }
    { const char* e = getenv("ADVGRPO_TNG_SLICES"); if (e && atoi(e) > 0) max_slices = atoi(e); }
#endif
    int cpb = (int)((chunks_total + target - 1) / target);
    if (cpb < 8) cpb = 8;
    int wg = 0, cnt = 0;
    int64_t off = 0;
    for (int i = 0; i < n; ++i) {
        const advgrpo_tn_desc& a = d[i];
        TnProblem& p = g.pr[i];
        const int nchunks = (a.M + 63) / 64;
        int c = cpb;
        if ((nchunks + c - 1) / c > max_slices) c = (nchunks + max_slices - 1) / max_slices;
        p.chunks_per_block = c;
        p.slices = (nchunks + c - 1) / c;
        p.tiles = a.N1 / 128;
        p.P = (const bf16_t*)a.P; p.Q = (const bf16_t*)a.Q;
        p.C = a.C;
        p.ldp = a.ldp; p.p_seg_rows = a.p_seg_rows; p.p_seg_stride = a.p_seg_stride; p.p_seg_off = a.p_seg_off;
        p.ldq = a.ldq; p.q_seg_rows = a.q_seg_rows; p.q_seg_stride = a.q_seg_stride; p.q_seg_off = a.q_seg_off;
        p.ldc = a.ldc; p.transpose_out = a.transpose_out; p.M = a.M; p.N1 = a.N1; p.alpha = a.alpha;
        p.wg0 = wg; p.cnt_off = cnt; p.ws_off = off;
        wg += p.tiles * p.slices;
        cnt += p.tiles;
        off += (int64_t)p.slices * a.N1 * NQ;
    }
    // layout: [arrival counters of every tile: zero between launches, at a fixed place so that launches with different problem sets
    // share the zeroed words | partial tiles]
    ADVGRPO_CHECK((int64_t)cnt * 4 <= TN_CNT_BYTES, "gemm_tn_grouped: too many tiles in one launch (%d)", cnt);
    ADVGRPO_CHECK(off * 4 + TN_CNT_BYTES <= workspace_bytes, "gemm_tn_grouped: workspace too small (%lld bytes needed)", (long long)(off * 4 + TN_CNT_BYTES));
    ADVGRPO_CHECK((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "gemm_tn_grouped: workspace must be 256-byte aligned");
    g.counters = reinterpret_cast<int*>(workspace);
    g.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + TN_CNT_BYTES);
    hipStream_t s = as_stream(stream);
    if (!workspace_is_zeroed) {
        if (hipMemsetAsync(workspace, 0, TN_CNT_BYTES, s) != hipSuccess) { set_error("gemm_tn_grouped: memset failed"); return -2; }
    }
    constexpr int LDS = 3 * (64 * 256 + 64 * 128);
    static bool attr1 = false;
    if (!attr1) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_grouped_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr1 = true; }
    hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(wg), dim3(256), LDS, s, g);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}
