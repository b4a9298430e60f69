// store_pattern.hip -- how fast can the Jacobian values of the 12-state quadrotor (cfg 5: 20 000 steps x 2904 entries x 8 B
// = 465 MB, + c and the V column) be WRITTEN, in the order the emit phase writes them?  No arithmetic, no LDS: the store
// stream alone, for (a) the order of cons_jac_kernel's long-period emit loop (a lane owns positions k, k + 256, ... of the
// period and walks the T steps of the tile: consecutive stores of a lane are one period = 23 KB apart), (b) the linear order
// (the tile's contiguous T * 2904 entries front to back), (c) linear with 16-byte stores.
//   hipcc --offload-arch=gfx950 -O3 bench/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>

__global__ void __launch_bounds__(256) k_period_major(double* out, int Ls, int T, long nsteps) {
    const long a = (long)blockIdx.x * T;
    const int nreg = (int)std::min<long>(T, nsteps - a);
    double* o = out + a * Ls;
    for (int k = threadIdx.x; k < Ls; k += blockDim.x)
        for (int s = 0; s < nreg; ++s) o[(long)s * Ls + k] = 1.0 + k;
}
__global__ void __launch_bounds__(256) k_linear(double* out, int Ls, int T, long nsteps) {
    const long a = (long)blockIdx.x * T;
    const int nreg = (int)std::min<long>(T, nsteps - a);
    double* o = out + a * Ls;
    const int E = nreg * Ls;
    for (int e = threadIdx.x; e < E; e += blockDim.x) o[e] = 1.0 + e;
}
__global__ void __launch_bounds__(256) k_linear16(double* out, int Ls, int T, long nsteps) {
    const long a = (long)blockIdx.x * T;
    const int nreg = (int)std::min<long>(T, nsteps - a);
    double2* o = reinterpret_cast<double2*>(out + a * Ls);
    const int E = nreg * Ls / 2;
    for (int e = threadIdx.x; e < E; e += blockDim.x) o[e] = make_double2(1.0 + e, 2.0);
}

template <class F> static double time_us(F launch, hipStream_t st) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch();
    (void)hipStreamSynchronize(st);
    const int K = 200;
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < K; ++i) launch();
    (void)hipEventRecord(e1, st);
    (void)hipStreamSynchronize(st);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / K * 1e3;
}

int main() {
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const long N = 20000; const int Ls = 2904;
    double* out; if (hipMalloc(&out, (size_t)N * Ls * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const double bytes = (double)N * Ls * 8;
    for (int T : {4, 8, 16, 32}) {
        const int grid = (int)((N + T - 1) / T);
        const double a = time_us([&] { k_period_major<<<grid, 256, 0, st>>>(out, Ls, T, N); }, st);
        const double b = time_us([&] { k_linear<<<grid, 256, 0, st>>>(out, Ls, T, N); }, st);
        const double c = time_us([&] { k_linear16<<<grid, 256, 0, st>>>(out, Ls, T, N); }, st);
        printf("T=%2d grid %5d: period-major %.1f us (%.2f TB/s)   linear %.1f us (%.2f TB/s)   linear 16-byte %.1f us (%.2f TB/s)\n", T, grid,
               a, bytes / a / 1e6, b, bytes / b / 1e6, c, bytes / c / 1e6);
    }
    return 0;
}
