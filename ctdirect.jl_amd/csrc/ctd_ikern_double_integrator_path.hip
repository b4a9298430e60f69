// Fused iteration kernels (ctd_iter_kernels.hpp) of one registry entry (DoubleIntegratorPathOCP).
#include "ctd_iter_kernels.hpp"
namespace ctd {
CTD_INSTANTIATE_ITER(DoubleIntegratorPathOCP)
}
