// Hessian-kernel instantiations of the collocation engine for one registry entry (EstimateInitialConditionOCP); see ctd_hess_kernels.hpp.
#include "ctd_hess_step.hpp"
namespace ctd {
CTD_INSTANTIATE_HESS(EstimateInitialConditionOCP)
CTD_INSTANTIATE_HESS_STEP(EstimateInitialConditionOCP)
}
