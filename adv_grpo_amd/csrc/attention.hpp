// attention.hpp -- parameter block and LDS-DMA / lane-swap helpers shared by the attention forward kernels.
#pragma once
#include "common.hpp"
#include "gemm.hpp"

namespace advgrpo {

struct AttnParams {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o;
    int64_t ldq, ldk, ldv, ldo;   // row pitch (elements)
    int64_t bsq, bsk, bsv, bso;   // batch pitch (elements)
    int H, Sq, Skv;
    float scale_log2e;            // softmax scale * log2(e)
    int causal;
    float* lse;                   // optional [B,H,Sq]: base-2 log-sum-exp of the scaled scores (for backward)
    const float* bias;            // optional additive score bias [H,Sq,Skv] f32 (T5 relative position bias), head dim 64 only
    int nqb, nwg, xcd_local;      // query blocks per (b,h); workgroups in the 1-D grid; XCD-local block order (common.hpp)
};

typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr int ATT_QB = 128;   // queries per workgroup
constexpr int ATT_KB = 64;    // keys per tile

// xor-16 / xor-32 lane exchanges on the VALU (gfx950 v_permlane{16,32}_swap) instead of ds_bpermute.
// v_permlane32_swap d, s: d[32..63] <-> s[0..31]; with d = s = v the pair (d, s) afterwards holds, in every lane,
// the lane's value and its xor-32 partner's (16: odd 16-lane rows of d <-> even rows of s).
// (inline asm with two read-write operands: given the same value twice the builtin form is folded onto ONE
// register and returns the swap of a register with itself -- checked with scripts/probes/permlane_probe.hip)
#define ADVGRPO_SWAP16(a, b) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b))
#define ADVGRPO_SWAP32(a, b) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b))
__device__ __forceinline__ float xor16_max(float v) { float a = v, b = v; ADVGRPO_SWAP16(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xor32_max(float v) { float a = v, b = v; ADVGRPO_SWAP32(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float xor16_add(float v) { float a = v, b = v; ADVGRPO_SWAP16(a, b); return a + b; }
__device__ __forceinline__ float xor32_add(float v) { float a = v, b = v; ADVGRPO_SWAP32(a, b); return a + b; }

// One LDS-DMA instruction (64 lanes x 16 bytes -> LDS [lds, lds + 1 KiB)), hand-written: issued through the builtin the
// compiler treats the DMA as a possible alias of every later ds_read and puts s_waitcnt vmcnt(0) in front of the first
// fragment read of the SAME iteration.  Ring hazards are covered by the caller's counted wait + barrier.
__device__ __forceinline__ void att_dma16(const void* src, const char* lds) {
    const uint32_t l = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)(lds));
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(l) : "memory");
}

// the software-pipelined head-dim-64 forward (attention_pipe.hip): non-causal, no score bias, 16-byte aligned output rows
int attention_fwd_pipe_launch(const AttnParams& p, hipStream_t s);
// head dim 128 (attention_d128.hip): 8-wave workgroups of 256 queries, non-causal, no score bias, 16-byte aligned output rows
int attention_fwd_d128_launch(const AttnParams& p, int B, hipStream_t s);
// workgroups of the two kernels above that took their running-maximum fallback since the last reset (host call, synchronises)
int attention_pipe_fallbacks(unsigned long long* out, int reset);
int attention_d128_fallbacks(unsigned long long* out, int reset);

}  // namespace advgrpo
