// Kernel instantiations of the collocation engine for one registry entry (LeastSquaresConstraintOCP); see ctd_kernels.hpp.
#include "ctd_kernels.hpp"
namespace ctd {
CTD_INSTANTIATE_LAUNCHERS(LeastSquaresConstraintOCP)
}
