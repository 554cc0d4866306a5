// Second translation unit of mlp_bwd_split.hip: only the double-staged instantiation mlp_bwd_split_kernel<3, true> (the
// BASELINE net, K0 <= 36) and its launcher, so that it can be compiled with its own scheduling strategy (build.py EXTRA).
#define PSDF_SPLIT_TU_DOUBLE 1
#include "mlp_bwd_split.hip"
