// attention_bwd.hip -- C entry of the attention backward for gfx950 (bf16 in/out, f32 math, head dim 64) and its row kernel.
//
// Needed by the G-step: autograd of the transformer call in compute_log_prob
// (scripts/train_sd3_fast_pickscore.py:233-267) as reached from loss.backward() (:1165); diffusers runs
// F.scaled_dot_product_attention, whose backward this replaces.
//
// Three launches, no atomics, probabilities recomputed from the forward's base-2 log-sum-exp:
//   delta   D[q] = sum_d O[q,d] dO[q,d]  -> the per-query vectors (-L/c | -D) in blocks of 32 queries   (HBM bound row kernel)
//   dq, dkdv   the two instantiations of the software-pipelined kernel in attention_bwd_pipe.hip
#include <stdlib.h>

#include "attention_bwd.hpp"

namespace advgrpo {

// ---------------------------------------------------------------- per-query vectors: -L/c and -D = -sum_d O dO
// One wave per (b, q) row, q running over whole blocks of 32 (the padding queries get -inf / 0: their probabilities are then
// exactly 0 in the dK/dV kernel, which never tests a query index).
__global__ __launch_bounds__(256) void attn_bwd_delta_kernel(const AttnBwdParams p, int B) {
    const int lane = threadIdx.x & 63;
    const int Sp = p.nb32 * 32;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (b, q)
    if (row >= (int64_t)B * Sp) return;
    const int b = row / Sp, q = row % Sp;
    const bool real = q < p.Sq;
    const bf16_t* o = p.o + (int64_t)b * p.bso + (int64_t)q * p.ldo;
    const bf16_t* d = p.d_o + (int64_t)b * p.bsdo + (int64_t)q * p.lddo;
    const int sub = lane & 7;
    const float inv_c = 1.0f / p.scale_log2e;
    for (int h0 = 0; h0 < p.H; h0 += 8) {
        const int h = h0 + (lane >> 3);
        float s = 0.f;
        if (h < p.H && real) {
            const uint4 a = *reinterpret_cast<const uint4*>(o + h * 64 + sub * 8);
            const uint4 c = *reinterpret_cast<const uint4*>(d + h * 64 + sub * 8);
            const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, cw[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                s += bf2f((bf16_t)(aw[k] & 0xffffu)) * bf2f((bf16_t)(cw[k] & 0xffffu)) +
                     bf2f((bf16_t)(aw[k] >> 16)) * bf2f((bf16_t)(cw[k] >> 16));
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (h < p.H && sub == 0) {
            float* v = p.vec + ((((int64_t)b * p.H + h) * p.nb32 + (q >> 5)) * 64) + (q & 31);
            v[0] = real ? -p.lse[((int64_t)b * p.H + h) * p.Sq + q] * inv_c : -INFINITY;
            v[32] = real ? -s : 0.f;
        }
    }
}

}  // namespace advgrpo

using namespace advgrpo;

extern "C" int advgrpo_attention_bwd(const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                     const float* lse, float* work, void* dq, void* dk, void* dv, int64_t ldq,
                                     int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddo, int64_t lddq, int64_t bsq,
                                     int64_t bsk, int64_t bsv, int64_t bso, int64_t bsdo, int64_t bsdq, int B, int H,
                                     int Sq, int Skv, int head_dim, float scale, void* stream) {
    ADVGRPO_CHECK(head_dim == 64 || head_dim == 128, "attention_bwd: head_dim %d not supported (64, 128)", head_dim);
    ADVGRPO_CHECK(q && k && v && o && d_o && lse && work && dq && dk && dv, "attention_bwd: null pointer");
    ADVGRPO_CHECK(B > 0 && H > 0 && Sq > 0 && Skv > 0, "attention_bwd: bad shape");
    ADVGRPO_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && lddq % 8 == 0 &&
                      bsq % 8 == 0 && bsk % 8 == 0 && bsv % 8 == 0 && bsdo % 8 == 0 && bsdq % 8 == 0,
                  "attention_bwd: row and batch pitches must keep 16-byte alignment");
    ADVGRPO_CHECK(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dq | (uintptr_t)dk |
                   (uintptr_t)dv) % 16 == 0, "attention_bwd: pointers must be 16-byte aligned");
    ADVGRPO_CHECK(32 * ldq * 2 < (1ll << 31) && 32 * ldk * 2 < (1ll << 31) && 32 * ldv * 2 < (1ll << 31) && 32 * lddo * 2 < (1ll << 31),
                  "attention_bwd: row pitch too large");
    AttnBwdParams p{};
    p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (const bf16_t*)o;
    p.d_o = (const bf16_t*)d_o; p.lse = lse; p.vec = work;
    p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.lddo = lddo; p.lddq = lddq;
    p.bsq = bsq; p.bsk = bsk; p.bsv = bsv; p.bso = bso; p.bsdo = bsdo; p.bsdq = bsdq;
    p.H = H; p.Sq = Sq; p.Skv = Skv; p.nb32 = (Sq + 31) / 32; p.scale = scale; p.scale_log2e = scale * 1.4426950408889634f;
    hipStream_t s = as_stream(stream);
    if (head_dim == 128) {
        p.xcd_local = 1;
        return attention_bwd_d128_launch(p, B, s);
    }
    hipLaunchKernelGGL(attn_bwd_delta_kernel, dim3((unsigned)(((int64_t)B * p.nb32 * 32 + 3) / 4)), dim3(256), 0, s, p, B);
    ADVGRPO_LAUNCH_CHECK();
    int xcd_local = 1;
#ifdef ADVGRPO_EXPERIMENTS
    { const char* e = getenv("ADVGRPO_ATTN_NO_XCD"); if (e && atoi(e)) xcd_local = 0; }
#endif
    p.xcd_local = xcd_local;
    if (int rc = attention_bwd_pipe_launch(p, B, s)) return rc;
    return 0;
}
