// PARAFAC2 slab kernels (functions/cmtf_fun_AOADMM.m:157-250, :509-589) -- see par2.h.
#include "common.h"
namespace aoadmm {
}  // namespace aoadmm
