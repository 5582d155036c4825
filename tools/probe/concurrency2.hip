// Big full-chip kernels on stream a + a dependent chain of small kernels on stream b: does the chain hide?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void spin(long long cycles) {
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
}
static double ms(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
  return std::chrono::duration<double, std::milli>(b - a).count();
}
int main(int argc, char** argv) {
  int prio_lo, prio_hi;
  hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  printf("priority range: least %d greatest %d\n", prio_lo, prio_hi);
  for (int mode = 0; mode < 3; ++mode) {
    hipStream_t a, b;
    if (mode == 0) { a = 0; hipStreamCreateWithFlags(&b, hipStreamNonBlocking); }
    if (mode == 1) { hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking); }
    if (mode == 2) { a = 0; hipStreamCreateWithPriority(&b, hipStreamNonBlocking, prio_hi); }
    // big: 2048 WGs x 256 threads, each WG spins 10 us -> ~2 waves of WGs over 256 CUs x 8 slots; 60 launches
    // chain: 400 kernels of 256 WGs x 256 threads spinning 3 us
    for (int rep = 0; rep < 2; ++rep) {
      hipDeviceSynchronize();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < 60; ++i) hipLaunchKernelGGL(spin, dim3(8192), dim3(256), 0, a, 1000);
      hipDeviceSynchronize();
      auto t1 = std::chrono::steady_clock::now();
      for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, b, 300);
      hipDeviceSynchronize();
      auto t2 = std::chrono::steady_clock::now();
      for (int i = 0; i < 400; ++i) {
        hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, b, 300);
        if (i % 7 == 0 && i / 7 < 60) hipLaunchKernelGGL(spin, dim3(8192), dim3(256), 0, a, 1000);
      }
      hipDeviceSynchronize();
      auto t3 = std::chrono::steady_clock::now();
      if (rep) printf("mode %d (%s): big alone %.2f ms | chain alone %.2f ms | both %.2f ms\n", mode,
             mode == 0 ? "null + nonblocking" : mode == 1 ? "two nonblocking" : "null + high-priority", ms(t0, t1), ms(t1, t2), ms(t2, t3));
    }
  }
  return 0;
}
