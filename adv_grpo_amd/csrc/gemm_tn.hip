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
