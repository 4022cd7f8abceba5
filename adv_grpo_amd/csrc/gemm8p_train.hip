// gemm8p_train.hip -- the G-step's epilogue classes of the eight-phase GEMM (gemm8p_kernel.hpp), instantiated in their own
// translation unit so that they compile in parallel with the rollout classes of gemm8p.hip:
//   EPI_PLAIN          data-gradient GEMMs (dX = dY W), LoRA products
//   EPI_DGELU          dX = (dY W2) * gelu_tanh'(pre-activation)              (backward of FF2 -> FF1 seam)
//   EPI_BIAS_GELU_AUX  gelu_tanh(x W1 + b) keeping the pre-activation          (training forward of FF1)
// Reference site: loss.backward() through the MMDiT Linears, scripts/train_sd3_fast_pickscore.py:1165.
#include "gemm8p_kernel.hpp"

namespace advgrpo {

int gemm8p_launch_train_class(int epi, const GemmPair& pp, const P8Sched& sc, hipStream_t s) {
    switch (epi) {
        case EPI_PLAIN: return launch8p<EPI_PLAIN>(pp, sc, s);
        case EPI_DGELU: return launch8p<EPI_DGELU>(pp, sc, s);
        case EPI_BIAS_GELU_AUX: return launch8p<EPI_BIAS_GELU_AUX>(pp, sc, s);
    }
    set_error("gemm8p: class %d is not a training class", epi);
    return -1;
}

}  // namespace advgrpo
