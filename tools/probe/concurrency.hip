// Do two HIP streams ever run kernels concurrently on this box?  Each kernel spins ~200 us on 8 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void spin(long long cycles, int* sink) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 9999) *sink = 1;
}
int main() {
  hipStream_t a, b;
  hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
  const long long cyc = 20000;  // wall_clock64 runs at 100 MHz -> 200 us
  for (int rep = 0; rep < 2; ++rep) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, a, cyc, nullptr);
    hipDeviceSynchronize();
    auto t1 = std::chrono::steady_clock::now();
    for (int i = 0; i < 20; ++i) {
      hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, a, cyc, nullptr);
      hipLaunchKernelGGL(spin, dim3(8), dim3(64), 0, b, cyc, nullptr);
    }
    hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    printf("one stream x20: %.2f ms | two streams x20 each: %.2f ms (concurrent if ~equal)\n",
           std::chrono::duration<double, std::milli>(t1 - t0).count(),
           std::chrono::duration<double, std::milli>(t2 - t1).count());
  }
  return 0;
}
