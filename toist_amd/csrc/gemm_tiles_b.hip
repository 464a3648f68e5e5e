// Second translation unit of gemm.hip: the 64x64x64 tile family (see "translation units" in gemm.hip).
#define GEMM_UNIT 1
#include "gemm.hip"
