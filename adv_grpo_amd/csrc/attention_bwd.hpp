// attention_bwd.hpp -- parameter block shared by the attention backward kernels (attention_bwd.hip, attention_bwd_pipe.hip).
#pragma once
#include "attention.hpp"

namespace advgrpo {

struct AttnBwdParams {
    const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* o; const bf16_t* d_o;
    const float* lse;
    // per-query vectors written by the delta kernel, in blocks of 32 queries padded to whole blocks:
    //   vec[((b*H + h)*nb32 + q/32)*64 + q%32]      = -lse[q] / (scale * log2 e)   (-inf for the padding queries)
    //   vec[((b*H + h)*nb32 + q/32)*64 + 32 + q%32] = -sum_d o[q,d] d_o[q,d]        (0 for the padding queries)
    float* vec;
    bf16_t* dq; bf16_t* dk; bf16_t* dv;
    int64_t ldq, ldk, ldv, ldo, lddo, lddq;
    int64_t bsq, bsk, bsv, bso, bsdo, bsdq;
    int H, Sq, Skv, nb32;
    float scale, scale_log2e;
    int xcd_local;                // XCD-local block order (common.hpp xcd_local_bh)
};

// dQ (DKDV = false) and dK / dV (true): software-pipelined kernels of attention_bwd_pipe.hip
int attention_bwd_pipe_launch(const AttnBwdParams& p, int B, hipStream_t s);
// head dim 128 (attention_bwd_d128.hip): its own delta kernel writes D[b,h,q] = sum_d O dO as a plain [B,H,Sq] array into p.vec
int attention_bwd_d128_launch(const AttnBwdParams& p, int B, hipStream_t s);

}  // namespace advgrpo
