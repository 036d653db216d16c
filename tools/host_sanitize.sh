#!/bin/bash
# AddressSanitizer + UBSan over the HOST code (GPU sanitizers are not available on this pool): the C port of the oracle on
# seeded inputs with nested OpenMP teams, and the library's host arithmetic (Horner step, binary-Euclid inversion, partial sums)
# through the CPU tests that call msm_combine_curve / msm_combine_groups.  Run from the repo root on a box without a GPU.
set -e
T=$(mktemp -d)
python3 - "$T" <<'PY'
import sys
sys.path.insert(0, ".")
from oracle import msm_oracle as O
n = 1 << 16
base, _ = O.random_points_bls377("sanitize", 256)
open(sys.argv[1] + "/points.bin", "wb").write(O.points_to_bytes(base, 48) * (n // 256))
open(sys.argv[1] + "/scalars.bin", "wb").write(O.scalars_to_bytes(O.prng_ints("sanitize/s", n, O.BLS12_377.q)))
PY
cat > $T/drv.c <<'C'
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
int oracle_msm_bls377(const uint8_t*, const uint8_t*, uint64_t, int, uint8_t*, int*, int*);
int main(int argc, char** argv) {
  char path[512];
  for (int a = 2; a < argc; a++) {
    uint64_t n = strtoull(argv[a], 0, 10);
    uint8_t *pts = malloc(96 * n), *sc = malloc(32 * n);
    snprintf(path, sizeof path, "%s/points.bin", argv[1]); FILE* f = fopen(path, "rb"); if (fread(pts, 96, n, f) != n) return 2; fclose(f);
    snprintf(path, sizeof path, "%s/scalars.bin", argv[1]); f = fopen(path, "rb"); if (fread(sc, 32, n, f) != n) return 2; fclose(f);
    for (int c = 0; c <= 13; c += (c == 0 ? 3 : 5)) {
      uint8_t out[96]; int inf = 0, thr = 0;
      printf("n=%llu c=%d rc=%d threads=%d\n", (unsigned long long)n, c, oracle_msm_bls377(pts, sc, n, c, out, &inf, &thr), thr);
    }
    free(pts); free(sc);
  }
  return 0;
}
C
gcc -O1 -g -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer oracle/msm_oracle.c $T/drv.c -o $T/drv
OMP_NUM_THREADS=16 $T/drv $T 1 2 33 1000 65536
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SAN="-O1 -g -pthread -std=c++17 --offload-arch=gfx950 -fPIC -Iinclude -Imontgomery_amd/csrc -fsanitize=address,undefined -fno-omit-frame-pointer"
for c in CvBls377 CvBls381 CvPallas; do $HIPCC $SAN -DMSM_CURVE_TU=$c -c montgomery_amd/csrc/kernels_curve.hip -o $T/k_$c.o & done
$HIPCC $SAN -c montgomery_amd/csrc/msm_api.hip -o $T/api.o; wait
$HIPCC --offload-arch=gfx950 -shared -pthread -fsanitize=address,undefined $T/*.o -o $T/libmsm_asan.so
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
MSM_HIP_LIB=$T/libmsm_asan.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 python3 -m pytest tests/test_distributed.py tests/test_abi.py -x -q -k "combine or exports or opts"
rm -rf $T
