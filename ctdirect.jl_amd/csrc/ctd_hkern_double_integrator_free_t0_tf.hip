// Hessian-kernel instantiations of the collocation engine for one registry entry (DoubleIntegratorFreeT0TfOCP); see ctd_hess_kernels.hpp.
#include "ctd_hess_step.hpp"
namespace ctd {
CTD_INSTANTIATE_HESS(DoubleIntegratorFreeT0TfOCP)
CTD_INSTANTIATE_HESS_STEP(DoubleIntegratorFreeT0TfOCP)
}
