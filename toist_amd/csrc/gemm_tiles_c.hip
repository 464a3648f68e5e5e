// Third translation unit of gemm.hip: the 128x64x64 and 64x128x64 tile families (see "translation units" in gemm.hip).
#define GEMM_UNIT 2
#include "gemm.hip"
