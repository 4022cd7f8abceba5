// gemm8p.hip -- 256 x 256 x 64 eight-phase persistent bf16 MFMA GEMM for gfx950 (the wide Linears of the MMDiT: M = 16384
// image rows (+ 3280 text rows), N, K in {1536, 4608, 6144}; reference call sites sd3_pipeline_with_logprob_fast.py:630-637,
// train_sd3_fast_pickscore.py:235-255 -> diffusers' SD3Transformer2DModel Linears).
//
//   C[M,N] = epilogue( A[M,K] . W[N,K]^T )      same operands, fragment layout and fused epilogue as gemm.hip
//
// Why a second kernel: the two-workgroup tiles of gemm.hip (192x128, 128x128) need 0.036 LDS bytes per flop and wait for
// the slowest L2 miss of the next k-tile once per 64-deep step; they converge at ~1 PFLOP/s.  This structure runs its main
// loop at 1.45 PFLOP/s on random data (4096 x 4096 x 32768; 88 % of the MFMA rate at the clock the chip sustains):
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
//     The next tile's first k-tile is requested BEFORE the epilogue of the current one (the epilogue bounces its slabs
//     through the other k-tile buffer), so a tile boundary costs the epilogue but no pipeline refill.
#include "gemm8p_kernel.hpp"

namespace advgrpo {

unsigned long long* g_p8_stamps = nullptr;
int gemm8p_launch_train_class(int epi, const GemmPair& pp, const P8Sched& sc, hipStream_t s);   // gemm8p_train.hip
int gemm8p_launch_fp8_class(int epi, const GemmPair& pp, const P8Sched& sc, hipStream_t s);     // gemm8p_fp8.hip

// every operand of the fused epilogue can be accessed as aligned 16-byte row segments (the dispatcher asks before choosing)
bool gemm8p_ok(const GemmParams& p) {
    auto a16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    // (the epilogue lets a 16-row slab straddle at most one segment / gate-vector boundary)
    if ((p.seg_rows > 0 && p.seg_rows < 16) || (p.gate && p.gate_rows > 0 && p.gate_rows < 16)) return false;
    return p.batch == 1 && p.splitk == 1 && !p.conv && (p.N & 7) == 0 &&
           ((p.ldc | p.ldr | p.gate_stride | p.ld_aux) & 7) == 0 && a16(p.C) && a16(p.bias) && a16(p.gate) &&
           a16(p.residual) && a16(p.aux_out) && a16(p.aux_in) && a16(p.rms_w);
}

namespace {


int prepare(GemmParams& p) {
    ADVGRPO_CHECK(p.A && p.W && p.C, "gemm8p: null operand");
    const int es = p.fp8 ? 1 : 2;                 // bytes per operand element; a k-tile is 128 bytes of a row
    ADVGRPO_CHECK(p.M > 0 && p.N > 0 && p.K > 0 && (p.K * es) % 128 == 0, "gemm8p: need M,N>0 and K %% %d == 0 (M=%d N=%d K=%d)",
                  128 / es, p.M, p.N, p.K);
    ADVGRPO_CHECK((p.lda * es) % 16 == 0 && (p.ldw * es) % 16 == 0, "gemm8p: operand rows must be multiples of 16 bytes");
    ADVGRPO_CHECK(!p.fp8 || (p.a_scale && p.w_scale && p.a_seg_rows == 0 && (reinterpret_cast<uintptr_t>(p.w_scale) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.a_scale) & 3) == 0),
                  "gemm8p: fp8 operands need a_scale [M], a 16-byte aligned w_scale [N] and a contiguous row range of A");
    ADVGRPO_CHECK(p.batch == 1 && p.splitk == 1 && !p.conv, "gemm8p: plain (unbatched, unsplit) problems only");
    {
        const int64_t a_rows = p.a_seg_rows > 0 ? ((int64_t)((p.M - 1) / p.a_seg_rows) * p.a_seg_stride + p.a_seg_off + p.a_seg_rows) : p.M;
        ADVGRPO_CHECK(a_rows * p.lda * es < (1ll << 32) && (int64_t)p.N * p.ldw * es < (1ll << 32),
                      "gemm8p: operands must span less than 4 GiB (32-bit DMA offsets)");
    }
    {   // the epilogue addresses its operands with 24-bit row x 24-bit pitch multiplies and 32-bit byte offsets
        const int64_t o_rows = p.seg_rows > 0 ? ((int64_t)((p.M - 1) / p.seg_rows) * p.seg_stride + p.seg_off + p.seg_rows) : p.M;
        const int64_t ld_max = std::max(std::max(p.ldc, p.ldr), p.ld_aux);
        const int64_t g_rows = p.gate_rows > 0 ? (p.M - 1) / p.gate_rows + 1 : 1;
        ADVGRPO_CHECK(o_rows < (1 << 24) && ld_max < (1 << 24) && o_rows * ld_max * 4 < (1ll << 32) && p.seg_stride < (1ll << 31) &&
                          p.gate_stride < (1 << 24) && g_rows < (1 << 24) && g_rows * p.gate_stride * 2 < (1ll << 32),
                      "gemm8p: output too large for the 32-bit epilogue offsets (%lld rows x pitch %lld; %lld gate vectors x pitch %lld)",
                      (long long)o_rows, (long long)ld_max, (long long)g_rows, (long long)p.gate_stride);
    }
    ADVGRPO_CHECK(gemm8p_ok(p), "gemm8p: the row epilogue needs N %% 8 == 0 and 16-byte aligned rows of every operand");
    return 0;
}

// the epilogue class a problem can run with (the specialised classes assume bf16 output, a bias, no aux / d-activation)
int epi_class(const GemmParams& p) {
    if (p.out_dtype != ADVGRPO_BF16) return EPI_GENERIC;
    int f = 0;
    if (p.bias) f |= F_BIAS;
    if (p.rms_w) f |= F_RMS;
    if (p.act == ACT_GELU_TANH) f |= F_GELU;
    else if (p.act == ACT_DGELU_TANH && p.aux_in) f |= F_DGELU;
    else if (p.act != ACT_NONE) return EPI_GENERIC;
    if (p.aux_in && !(f & F_DGELU)) return EPI_GENERIC;
    if (p.gate && p.residual) f |= F_GATE_RES;
    else if (p.gate || p.residual) return EPI_GENERIC;
    if (p.aux_out) f |= F_AUX_OUT;
    if (p.fp8) {     // fp8 operands: the four rollout classes, each with the scale step
        switch (f) {
            case EPI_BIAS: case EPI_BIAS_RMS: case EPI_BIAS_GELU: case EPI_BIAS_GATE_RES: case EPI_BIAS_GELU_AUX: return f | F_SCALE;
        }
        return -2;   // (no generic fp8 class)
    }
    switch (f) {     // the instantiated classes
        case EPI_PLAIN: case EPI_BIAS: case EPI_BIAS_RMS: case EPI_BIAS_GELU: case EPI_BIAS_GATE_RES: case EPI_BIAS_GELU_AUX: case EPI_DGELU:
            return f;
    }
    return EPI_GENERIC;
}

int launch8p_any(int epi, const GemmPair& pp, const P8Sched& sc, hipStream_t s) {
    if (epi == -2) { set_error("gemm8p: fp8 operands need a bf16 output and one of the epilogues bias / bias+QK-norm / bias+GELU (+pre-activation) / bias+gate+residual"); return -1; }
    if (epi >= 0 && (epi & F_SCALE)) return gemm8p_launch_fp8_class(epi, pp, sc, s);
    switch (epi) {
        case EPI_BIAS: return launch8p<EPI_BIAS>(pp, sc, s);
        case EPI_BIAS_RMS: return launch8p<EPI_BIAS_RMS>(pp, sc, s);
        case EPI_BIAS_GELU: return launch8p<EPI_BIAS_GELU>(pp, sc, s);
        case EPI_BIAS_GATE_RES: return launch8p<EPI_BIAS_GATE_RES>(pp, sc, s);
        case EPI_PLAIN: case EPI_DGELU: case EPI_BIAS_GELU_AUX: return gemm8p_launch_train_class(epi, pp, sc, s);
        default: return launch8p<EPI_GENERIC>(pp, sc, s);
    }
}

int tiles_of(const GemmParams& p) { return ((p.M + P8_BM - 1) / P8_BM) * ((p.N + P8_BN - 1) / P8_BN); }

}  // namespace

int gemm8p_launch(const GemmParams& p_in, hipStream_t s) {
    GemmPair pp{};
    pp.a = p_in;
    if (prepare(pp.a)) return -1;
    pp.b = pp.a;                                  // never reached: tiles_total == tiles_a
    pp.tiles_a = tiles_of(pp.a);
    return launch8p_any(epi_class(pp.a), pp, P8Sched{pp.tiles_a, pp.tiles_a, nullptr}, s);
}

int gemm8p_launch_pair(const GemmParams& a, const GemmParams& b, hipStream_t s) {
    GemmPair pp{};
    pp.a = a; pp.b = b;
    if (prepare(pp.a) || prepare(pp.b)) return -1;
    pp.tiles_a = tiles_of(pp.a);
    const int ea = epi_class(pp.a), eb = epi_class(pp.b);
    ADVGRPO_CHECK(pp.a.fp8 == pp.b.fp8 && (!pp.a.fp8 || ea == eb), "gemm8p: a paired fp8 launch needs two fp8 problems with the same epilogue");
    return launch8p_any(ea == eb ? ea : (int)EPI_GENERIC, pp, P8Sched{pp.tiles_a, pp.tiles_a + tiles_of(pp.b), nullptr}, s);
}

}  // namespace advgrpo

#ifdef ADVGRPO_EXPERIMENTS
/* experiment only (not in include/advgrpo.h): copy the s_memtime stamps of the last gemm8p launch to the host */
extern "C" int advgrpo_dbg_p8_stamps(unsigned long long* host, int count) {
    if (!advgrpo::g_p8_stamps) return -1;
    return hipMemcpy(host, advgrpo::g_p8_stamps, (size_t)count * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif
