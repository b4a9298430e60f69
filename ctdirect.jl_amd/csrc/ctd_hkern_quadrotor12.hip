// Hessian-kernel instantiations of the collocation engine for one registry entry (Quadrotor12OCP); see ctd_hess_kernels.hpp.
#include "ctd_hess_step.hpp"
namespace ctd {
CTD_INSTANTIATE_HESS(Quadrotor12OCP)
CTD_INSTANTIATE_HESS_STEP(Quadrotor12OCP)
}
