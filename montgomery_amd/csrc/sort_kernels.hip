// The curve-independent kernels (scans, counting sort, radix split, tail descriptors, finish ordering): their one definition.
#include <hip/hip_runtime.h>
#define MSM_SORT_TU 1
#include "sort_kernels.h"
#include "tree_kernels.h"
