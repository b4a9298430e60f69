// Hessian-kernel instantiations of the collocation engine for one registry entry (DoubleIntegratorPathOCP); see ctd_hess_kernels.hpp.
#include "ctd_hess_step.hpp"
namespace ctd {
CTD_INSTANTIATE_HESS(DoubleIntegratorPathOCP)
CTD_INSTANTIATE_HESS_STEP(DoubleIntegratorPathOCP)
}
