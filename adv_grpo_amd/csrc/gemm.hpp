// gemm.hpp -- parameter block of the bf16 MFMA GEMM family (internal C++ interface).
#pragma once
#include "common.hpp"

namespace advgrpo {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

enum { ACT_NONE = 0, ACT_GELU_TANH = 1, ACT_GELU_ERF = 2, ACT_SILU = 3, ACT_QUICK_GELU = 4 /* x sigmoid(1.702 x), CLIP-L */,
       ACT_DGELU_TANH = 5 /* y *= gelu_tanh'(aux_in[m,n]) */, ACT_DGELU_ERF = 6,
       ACT_MUL_AUX = 7 /* y *= aux_in[m,n]: the gate of T5's gated-GELU feed-forward */ };

struct GemmParams {
    const bf16_t* A; const bf16_t* W; void* C;
    int64_t lda, ldw, ldc;
    int out_dtype;
    int M, N, K;
    // epilogue: y = act(alpha*acc + bias[n]); y *= gate[gb, n]; y += residual[orow, n]
    const bf16_t* bias; int act; float alpha;
    const bf16_t* gate; int64_t gate_stride; int gate_rows; int64_t gate_batch_stride;  // gb = m / gate_rows
    const bf16_t* residual; int64_t ldr; int64_t strideR;
    // output row map: orow = (m / seg_rows) * seg_stride + seg_off + m % seg_rows   (seg_rows = 0: identity)
    int seg_rows; int64_t seg_stride, seg_off;
    // input row map for A (same formula), e.g. reading the image rows out of a joint image+text buffer
    int a_seg_rows; int64_t a_seg_stride, a_seg_off;
    int batch; int64_t strideA, strideW, strideC;
    // implicit-GEMM 3x3 conv (stride 1, pad 1, optional nearest x2 upsample of the input), NHWC:
    // A = input [B, Hout>>ups, Wout>>ups, Cin], row m = output pixel (b, y, x), k = (ky*3+kx)*Cin + c
    int conv; int Hout, Wout, Cin, ups; const bf16_t* zero_page;
    // training extras: aux_out[orow, n] = pre-activation (bf16, pitch ldc); aux_in[orow, n] feeds the d-activation
    // epilogues; splitk > 1: the K range is split over grid.y and partial tiles are atomically added into f32 C
    bf16_t* aux_out; const bf16_t* aux_in; int64_t ld_aux; int splitk;
    // fused per-head RMSNorm on the first rms_nheads 64-wide column groups of the output (QK-norm of a fused QKV
    // projection: q heads then k heads, V untouched): y = bf16(bf16(x) * rsqrt(mean(x^2) + eps)) * w[(head / hpw), :];
    // rms_rs_out[orow, head] (f32, optional) keeps 1/rms for the backward.  Needs a 64-wide wave tile.
    const bf16_t* rms_w; int rms_nheads; int rms_hpw; float rms_eps; float* rms_rs_out;
    // fp8 (OCP e4m3) operands on the eight-phase kernel: A [M, K] and W [N, K] one BYTE per element (lda / ldw in elements =
    // bytes), a_scale [M] / w_scale [N] f32 multiply the accumulators in the epilogue (per-token x per-output-channel scaling)
    int fp8; const float* a_scale; const float* w_scale;
    int f32_io;  // convolutions only: bias / residual / C are f32 (the split-bf16 VAE mode keeps f32 between kernels)
    // f16x2 convolution only: sums for the GroupNorm that follows -- gn_partial[(m / 16) * (N / 4) + n / 4] = {sum, sum of squares}
    // (f32 pairs) over 16 consecutive output pixels x 4 consecutive output channels.  16 divides every image's pixel count, so a
    // block never straddles two images and an image's sums do not depend on its position in the batch
    float* gn_partial;
    // f16x2 convolution only: INSTEAD of the f32 output, the fp16-pair rows [hi | unwritten | lo] of pair_prescale * y (the operand form of
    // the next f16x2 convolution when nothing else reads y: a resnet's conv2 in front of an upsampler) -- pair_out [M, 3 N] 16-bit
    void* pair_out; float pair_prescale;
    int debug;   // experiments only (ADVGRPO_GEMM_DEBUG): bit0 = skip steady-state DMA, bit1 = skip LDS fragment reads
};

int gemm_bf16(const GemmParams& p, hipStream_t stream);
// both problems in one launch when a pair-capable tile variant fits, else two launches
int gemm_bf16_pair(const GemmParams& a, const GemmParams& b, hipStream_t stream);

}  // namespace advgrpo
