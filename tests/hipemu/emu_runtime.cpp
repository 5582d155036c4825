// storage of the emulator's per-thread launch context (see include/hip/hip_runtime.h)
#include <hip/hip_runtime.h>
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
namespace hipemu {
thread_local Launch* cur = nullptr;
thread_local Wave* wave = nullptr;
thread_local int tid_flat = 0;
}  // namespace hipemu

// rnnt_fused.hip hands its GEMMs to gemm_bf16.hip (MFMA / LDS-DMA: not emulated): calls fail loudly
#include "../../include/nsp_hip.h"
int nsp_gemm_bf16_launch(const nsp_gemm_params&, hipStream_t) { return NSP_EUNSUPPORTED; }
