#include <omp.h>
#include <stdio.h>
#include <stdint.h>
int main() {
  for (int nt = 1; nt <= 256; nt *= 2) {
    double t0 = omp_get_wtime();
    uint64_t tot = 0;
#pragma omp parallel num_threads(nt) reduction(+ : tot)
    {
      uint64_t x = omp_get_thread_num() + 1;
      for (long i = 0; i < 400000000L; i++) x = x * 6364136223846793005ULL + 1442695040888963407ULL;
      tot += x;
    }
    double dt = omp_get_wtime() - t0;
    printf("threads %3d  %.3f s  aggregate %.2f G iter/s  (%llu)\n", nt, dt, nt * 0.4 / dt, (unsigned long long)tot);
  }
  return 0;
}
