// The ablation switches of gemm_wino.hip in ONE place (timing experiments only: tools/wino_ablation.sh builds the
// source with -DUNETPP_WINO_EXP_<NAME>, every variant but the normal build computes wrong results).  The kernel reads
// them as `if constexpr (!wino_exp::kNoStore) ...`, so its body holds no preprocessor branches.  tools/ablation_audit.py
// (run by tests/test_isa_hazards.py) compiles every switch and checks that a variant that is meant to keep the matrix
// work still has the normal build's v_mfma count -- an ablation that removes the only consumer of the accumulators lets
// hipcc delete the MFMAs too, which is how rounds 3-4 measured a "no epilogue" build of gemm_bf16_dma.hip that had no
// matrix work left.
#pragma once

namespace unetpp {
namespace wino_exp {

#ifdef UNETPP_WINO_EXP_NO_STAGING  // nothing of the next chunk is staged (the cursor still moves)
constexpr bool kNoStaging = true;
#else
constexpr bool kNoStaging = false;
#endif
#ifdef UNETPP_WINO_EXP_NO_STORE  // the prefetched inputs are not written to LDS
constexpr bool kNoStore = true;
#else
constexpr bool kNoStore = false;
#endif
#ifdef UNETPP_WINO_EXP_NO_DMA  // no weight LDS-DMA
constexpr bool kNoDma = true;
#else
constexpr bool kNoDma = false;
#endif
#ifdef UNETPP_WINO_EXP_NO_LOADS  // no input loads
constexpr bool kNoLoads = true;
#else
constexpr bool kNoLoads = false;
#endif
#ifdef UNETPP_WINO_EXP_NO_BARRIER  // no chunk barrier (and no wait for the weight DMA)
constexpr bool kNoBarrier = true;
#else
constexpr bool kNoBarrier = false;
#endif
#ifdef UNETPP_WINO_EXP_NO_EPILOGUE  // the accumulators are consumed by an empty asm instead of the epilogue
constexpr bool kNoEpilogue = true;
#else
constexpr bool kNoEpilogue = false;
#endif

#ifdef UNETPP_WINO_EXP_NO_OUT_STORE  // the epilogue runs (transform, bias, ReLU) but its plain 16-byte stores are not issued
constexpr bool kNoOutStore = true;
#else
constexpr bool kNoOutStore = false;
#endif

#ifdef UNETPP_WINO_EXP_LAX_WAIT  // the chunk barrier behind an epilogue does not wait for the epilogue's stores (NOR for
constexpr bool kLaxWait = true;  // the weight DMA issued after them: wrong results, an upper bound of what the wait costs)
#else
constexpr bool kLaxWait = false;
#endif

}  // namespace wino_exp
}  // namespace unetpp
