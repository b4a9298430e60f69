// Kernel instantiations of the collocation engine for one registry entry (Quadrotor12OCP); see ctd_kernels.hpp.
#include "ctd_kernels.hpp"
namespace ctd {
CTD_INSTANTIATE_LAUNCHERS(Quadrotor12OCP)
}
