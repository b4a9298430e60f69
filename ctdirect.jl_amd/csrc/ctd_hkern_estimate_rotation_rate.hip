// Hessian-kernel instantiations of the collocation engine for one registry entry (EstimateRotationRateOCP); see ctd_hess_kernels.hpp.
#include "ctd_hess_step.hpp"
namespace ctd {
CTD_INSTANTIATE_HESS(EstimateRotationRateOCP)
CTD_INSTANTIATE_HESS_STEP(EstimateRotationRateOCP)
}
