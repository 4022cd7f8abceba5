// gemm8p_fp8.hip -- the eight-phase GEMM (gemm8p_kernel.hpp) on fp8 (OCP e4m3) operands: the rollout's four epilogue classes and
// the training forward's FF1 class, with the per-token x per-output-channel scale step, instantiated in their own translation unit.
//
//   C[M,N] = epilogue( (a_scale[m] * w_scale[n]) * sum_k A8[m,k] * W8[n,k] )        A8 [M,K], W8 [N,K] one byte per element
//
// BASELINE config 5 names an "fp8 MFMA path" for the MMDiT Linears; the reference has no code for it, so
// the numerics are this build's: symmetric e4m3 quantisation, one f32 scale per token row of the activation (quantize.hip)
// and per output channel of the weight, f32 accumulation, scales applied to the accumulator before the bias.  The k loop is
// the bf16 kernel's byte for byte (a k-tile is 128 bytes of a row = 128 fp8); a phase issues 8
// v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales instead of 16 v_mfma_f32_16x16x32_bf16 -- the same matrix-pipe
// time for twice the contraction depth.
#include "gemm8p_kernel.hpp"

namespace advgrpo {

int gemm8p_launch_fp8_class(int epi, const GemmPair& pp, const P8Sched& sc, hipStream_t s) {
    switch (epi) {
        case EPI_BIAS | F_SCALE: return launch8p<EPI_BIAS | F_SCALE, true>(pp, sc, s);
        case EPI_BIAS_RMS | F_SCALE: return launch8p<EPI_BIAS_RMS | F_SCALE, true>(pp, sc, s);
        case EPI_BIAS_GELU | F_SCALE: return launch8p<EPI_BIAS_GELU | F_SCALE, true>(pp, sc, s);
        case EPI_BIAS_GATE_RES | F_SCALE: return launch8p<EPI_BIAS_GATE_RES | F_SCALE, true>(pp, sc, s);
        case EPI_BIAS_GELU_AUX | F_SCALE: return launch8p<EPI_BIAS_GELU_AUX | F_SCALE, true>(pp, sc, s);   // training forward of FF1
    }
    set_error("gemm8p: class %d is not an fp8 class", epi);
    return -1;
}

}  // namespace advgrpo
