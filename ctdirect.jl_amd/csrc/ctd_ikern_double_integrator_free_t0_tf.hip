// Fused iteration kernels (ctd_iter_kernels.hpp) of one registry entry (DoubleIntegratorFreeT0TfOCP).
#include "ctd_iter_kernels.hpp"
namespace ctd {
CTD_INSTANTIATE_ITER(DoubleIntegratorFreeT0TfOCP)
}
