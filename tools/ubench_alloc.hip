// hipMalloc / hipFree cost against size on this box (the first MSM of a process pays for ~65 GB of workspace at 2^26)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipFree(0);
  for (size_t gb : {1, 4, 16, 32, 64}) {
    void* p = nullptr;
    double t0 = now();
    hipError_t e = hipMalloc(&p, gb << 30);
    double t1 = now();
    if (e != hipSuccess) { printf("%zu GB: %s\n", gb, hipGetErrorString(e)); continue; }
    hipMemset(p, 0, 256); hipDeviceSynchronize();
    double t2 = now();
    hipFree(p);
    double t3 = now();
    printf("%3zu GB: hipMalloc %8.1f ms (%.1f ms/GB)   hipFree %8.1f ms\n", gb, t1 - t0, (t1 - t0) / gb, t3 - t2);
  }
  // many medium buffers, as the workspace makes them
  double t0 = now();
  void* q[64];
  for (int i = 0; i < 64; i++) hipMalloc(&q[i], 1ull << 30);
  double t1 = now();
  for (int i = 0; i < 64; i++) hipFree(q[i]);
  printf("64 x 1 GB: hipMalloc %.1f ms in all, hipFree %.1f ms\n", t1 - t0, now() - t1);
  return 0;
}
