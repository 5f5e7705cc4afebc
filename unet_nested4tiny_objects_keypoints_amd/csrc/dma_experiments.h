// The ablation switches of gemm_bf16_dma.hip in ONE place (timing experiments only: tools/bf16_dma_ablation.sh builds the
// source with -DUNETPP_DMA_EXP_<NAME>; every variant but the normal build computes wrong results).  The kernel reads them
// as `if constexpr (dma_exp::kNoMfma) ...`: its body holds no preprocessor branches.  tools/ablation_audit.py (run by
// tests/test_isa_hazards.py) checks that a variant that is meant to keep the matrix work still has the normal build's
// v_mfma count (wino_experiments.h tells why).
#pragma once

namespace unetpp {
namespace dma_exp {

#ifdef UNETPP_DMA_EXP_NO_INDMA  // no input LDS-DMA
constexpr bool kNoInDma = true;
#else
constexpr bool kNoInDma = false;
#endif
#ifdef UNETPP_DMA_EXP_NO_WDMA  // no weight LDS-DMA (what the weight bytes cost a CU's memory path)
constexpr bool kNoWDma = true;
#else
constexpr bool kNoWDma = false;
#endif
#ifdef UNETPP_DMA_EXP_HALF_WDMA  // every second chunk's weights only
constexpr bool kHalfWDma = true;
#else
constexpr bool kHalfWDma = false;
#endif
#ifdef UNETPP_DMA_EXP_NO_STORE  // the whole epilogue (accumulators reset, values packed) without its stores
constexpr bool kNoStore = true;
#else
constexpr bool kNoStore = false;
#endif
#ifdef UNETPP_DMA_EXP_STORE_LINEAR  // every store instruction writes 1 KB contiguous (wrong layout)
constexpr bool kStoreLinear = true;
#else
constexpr bool kStoreLinear = false;
#endif
#ifdef UNETPP_DMA_EXP_NO_MFMA  // no matrix work
constexpr bool kNoMfma = true;
#else
constexpr bool kNoMfma = false;
#endif
#ifdef UNETPP_DMA_EXP_NO_EPI  // the accumulators are consumed by an empty asm instead of the epilogue
constexpr bool kNoEpi = true;
#else
constexpr bool kNoEpi = false;
#endif

}  // namespace dma_exp
}  // namespace unetpp
