// storage of the emulator's per-thread launch context (see include/hip/hip_runtime.h)
#include <hip/hip_runtime.h>
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
namespace hipemu {
thread_local Launch* cur = nullptr;
thread_local Wave* wave = nullptr;
thread_local int tid_flat = 0;
}  // namespace hipemu

