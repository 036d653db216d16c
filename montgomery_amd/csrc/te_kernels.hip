// The twisted Edwards kernels (Ed-on-BLS12-377, extended coordinates): their one definition.
#include <hip/hip_runtime.h>
#define MSM_TE_TU 1
#include "te_kernels.h"
