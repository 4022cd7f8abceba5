// lora_merge.hip -- every adapted projection's working weights from the flat LoRA parameter vector, ONE launch for the model (gfx950).
//
// Reference site: the policy the reference trains is PEFT's `base_layer(x) + scaling * lora_B(lora_A(x))` on the attention projections
// (scripts/train_sd3_fast_pickscore.py:490-511, get_peft_model); after every optimizer step (TP:1166-1171) this package re-merges
//   W_eff = W + (alpha / r) B A        (deviation 2: the rollout and the replay then read ONE weight per projection)
// and the data-gradient GEMMs of the G-step read W_eff^T.  Rounds 3 - 4 issued, per adapter, two 64-deep GEMMs (W_eff and W_eff^T), two
// transposes and the copies that build the stacked A / block-diagonal B^T operands of the adapter-gradient GEMMs: ~950 launches of
// ~8 us each per optimizer step for SD3.5-medium's 190 adapters (8 - 12 ms, all of it launch latency: the pass moves 3.6 GB).
//
// One workgroup per 64 x 64 tile of one adapter's W_eff:
//   acc[n, k] = sum_r B[n, r] A[r, k]      f32 fused multiply-adds, r ascending over the padded rank (bf16 products are exact in f32)
//   v = bf16(acc * alpha + base[n, k])     multiply, then add (two roundings, as the GEMM epilogues do), one rounding to bf16
//   w[n, k] = v  and  wT[k, n] = v         the transpose is the SAME bits by construction (rounds 3 - 4 computed it with a second GEMM)
// and, from the operand tiles it holds anyway, the workgroups of an adapter's first tile column / row write the adapter's rows of the
// group's stacked A (a_cat) and its diagonal block of the group's block-diagonal B^T (b_bd).  HBM-bound: base in, two tiles out.
#include "common.hpp"

#include "../../include/advgrpo.h"

namespace advgrpo {
namespace {

constexpr int MR = 64;                       // padded rank (RPAD of mmdit_train.py)

__global__ __launch_bounds__(256) void lora_merge_kernel(const advgrpo_lora_merge_item* __restrict__ items, float alpha) {
    __shared__ __attribute__((aligned(16))) bf16_t As[MR][64];        // A tile   [r][k]
    __shared__ __attribute__((aligned(16))) bf16_t Bst[MR][64];       // B tile^T [r][n]
    __shared__ __attribute__((aligned(16))) bf16_t T[64][72];         // the output tile [n][k] (row pitch 144 B: 16-byte aligned rows)
    const advgrpo_lora_merge_item it = items[blockIdx.z];
    const int n0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
    if (n0 >= it.N || k0 >= it.K) return;
    const int tid = threadIdx.x;
    const bf16_t* A = (const bf16_t*)it.A;
    const bf16_t* B = (const bf16_t*)it.B;
    // operand tiles: 512 16-byte pieces each, two per thread
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int piece = tid + h * 256, row = piece >> 3, c8 = (piece & 7) * 8;
        *reinterpret_cast<uint4*>(&As[row][c8]) = *reinterpret_cast<const uint4*>(A + (int64_t)row * it.K + k0 + c8);      // row = r
        const uint4 bv = *reinterpret_cast<const uint4*>(B + (int64_t)(n0 + row) * MR + c8);                            // row = n, c8 = r
        const uint32_t wv[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            Bst[c8 + 2 * q][row] = (bf16_t)(wv[q] & 0xffffu);
            Bst[c8 + 2 * q + 1][row] = (bf16_t)(wv[q] >> 16);
        }
    }
    __syncthreads();
    const int tn = tid >> 4, tk = tid & 15;      // a 4 (n) x 4 (k) patch per thread
    float acc[4][4] = {};
#pragma unroll 8
    for (int r = 0; r < MR; ++r) {
        const uint2 bw = *reinterpret_cast<const uint2*>(&Bst[r][tn * 4]);
        const uint2 aw = *reinterpret_cast<const uint2*>(&As[r][tk * 4]);
        const float b[4] = {bf2f((bf16_t)(bw.x & 0xffffu)), bf2f((bf16_t)(bw.x >> 16)), bf2f((bf16_t)(bw.y & 0xffffu)), bf2f((bf16_t)(bw.y >> 16))};
        const float a[4] = {bf2f((bf16_t)(aw.x & 0xffffu)), bf2f((bf16_t)(aw.x >> 16)), bf2f((bf16_t)(aw.y & 0xffffu)), bf2f((bf16_t)(aw.y >> 16))};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fmaf(b[i], a[j], acc[i][j]);
    }
    const bf16_t* base = (const bf16_t*)it.base;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = tn * 4 + i;
        const uint2 rw = *reinterpret_cast<const uint2*>(base + (int64_t)(n0 + n) * it.ld_base + k0 + tk * 4);
        const float res[4] = {bf2f((bf16_t)(rw.x & 0xffffu)), bf2f((bf16_t)(rw.x >> 16)), bf2f((bf16_t)(rw.y & 0xffffu)), bf2f((bf16_t)(rw.y >> 16))};
        bf16_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma clang fp contract(off)
            const float scaled = acc[i][j] * alpha;
            o[j] = f2bf(scaled + res[j]);
        }
        const uint2 packed = uint2{(uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16)};
        if (it.w) *reinterpret_cast<uint2*>((bf16_t*)it.w + (int64_t)(n0 + n) * it.ld_w + k0 + tk * 4) = packed;
        *reinterpret_cast<uint2*>(&T[n][tk * 4]) = packed;
    }
    __syncthreads();
    // transposed tile and the operand copies: thread -> (row, 16-element segment), 32 contiguous bytes each
    const int row = tid >> 2, seg = (tid & 3) * 16;
    {
        bf16_t o[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) o[q] = T[seg + q][row];                   // row = k
        uint4* dst = reinterpret_cast<uint4*>((bf16_t*)it.wT + (int64_t)(k0 + row) * it.ld_wT + n0 + seg);
        dst[0] = *reinterpret_cast<const uint4*>(&o[0]);
        dst[1] = *reinterpret_cast<const uint4*>(&o[8]);
    }
    if (it.b_bd && blockIdx.x == 0) {                                          // B^T rows r of this tile's 64 outputs n
        uint4* dst = reinterpret_cast<uint4*>((bf16_t*)it.b_bd + (int64_t)row * it.ld_bd + n0 + seg);
        dst[0] = *reinterpret_cast<const uint4*>(&Bst[row][seg]);
        dst[1] = *reinterpret_cast<const uint4*>(&Bst[row][seg + 8]);
    }
    if (it.a_cat && blockIdx.y == 0) {                                         // A rows r, this tile's 64 inputs k
        uint4* dst = reinterpret_cast<uint4*>((bf16_t*)it.a_cat + (int64_t)row * it.K + k0 + seg);
        dst[0] = *reinterpret_cast<const uint4*>(&As[row][seg]);
        dst[1] = *reinterpret_cast<const uint4*>(&As[row][seg + 8]);
    }
}

}  // namespace
}  // namespace advgrpo

using namespace advgrpo;

extern "C" int advgrpo_lora_merge(const advgrpo_lora_merge_item* items_device, int n_items, int max_N, int max_K, int rank_padded,
                                  float alpha, void* stream) {
    ADVGRPO_CHECK(items_device && n_items > 0 && n_items <= 65535, "lora_merge: need 1 <= n_items <= 65535 and a device table");
    ADVGRPO_CHECK(rank_padded == MR, "lora_merge: the padded rank must be %d (got %d)", MR, rank_padded);
    ADVGRPO_CHECK(max_N > 0 && max_K > 0 && max_N % 64 == 0 && max_K % 64 == 0 && max_N / 64 <= 65535,
                  "lora_merge: N and K of every adapter must be multiples of 64 (max_N=%d max_K=%d)", max_N, max_K);
    hipLaunchKernelGGL(lora_merge_kernel, dim3(max_K / 64, max_N / 64, n_items), dim3(256), 0, as_stream(stream), items_device, alpha);
    ADVGRPO_LAUNCH_CHECK();
    return 0;
}
