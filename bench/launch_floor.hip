// launch_floor.hip -- what one launch of the cfg-2 geometry costs on this GPU before any arithmetic:
//   (a) an empty kernel, (b) a kernel that only streams the evaluation's output bytes (9.6 MB of 8-byte stores, the
//   workgroup geometry of cons_jac_kernel), (c) the same after one dependent global load per lane (a memory round trip in
//   front of the stores, as reading x is).  Timed like bench.py: K back-to-back launches between two events / K, and per
//   dispatch with hipExtLaunchKernelGGL events.  Build: hipcc --offload-arch=gfx950 -O3 bench/launch_floor.hip -o gpurun_out/launch_floor
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Args { double* out; const double* in; long n_out; int per_block; int mode; char pad[600]; };

__global__ void k_empty(Args a) {}

__global__ void k_store(Args a) {
    extern __shared__ double lds[];
    const long base = (long)blockIdx.x * a.per_block;
    double v = 1.0;
    if (a.mode >= 1) v = a.in[(blockIdx.x * 64 + (threadIdx.x & 63)) % 4096];     // one memory round trip before the stores
    if (a.mode >= 2) {                                                           // + a second, dependent round trip
        const long j = (long)(v * 0.0) + threadIdx.x;
        v += a.in[j % 4096];
    }
    for (int e = threadIdx.x; e < a.per_block; e += blockDim.x)
        if (base + e < a.n_out) a.out[base + e] = v + e;
}

template <class F> static void time_it(const char* name, F launch, hipStream_t st) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 200; ++i) launch(nullptr, nullptr);
    hipStreamSynchronize(st);
    const int K = 2000;
    hipEventRecord(e0, st);
    for (int i = 0; i < K; ++i) launch(nullptr, nullptr);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> per;
    std::vector<hipEvent_t> ev(64);
    for (auto& e : ev) hipEventCreate(&e);
    for (int rep = 0; rep < 8; ++rep) {
        for (int i = 0; i < 32; ++i) launch(ev[2 * i], ev[2 * i + 1]);
        hipStreamSynchronize(st);
        for (int i = 0; i < 32; ++i) { float t; hipEventElapsedTime(&t, ev[2 * i], ev[2 * i + 1]); per.push_back(t); }
    }
    std::sort(per.begin(), per.end());
    printf("%-44s back-to-back %.2f us/launch   per-dispatch events: median %.2f us, min %.2f us\n", name, ms / K * 1e3,
           per[per.size() / 2] * 1e3, per[0] * 1e3);
}

struct Small { double* out; long n; };
__global__ void k_empty_small(Small a) {}

int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    {   // what the launch floor depends on: kernel-argument bytes, dynamic LDS, launch API
        Args a{nullptr, nullptr, 0, 0, 0, {}};
        Small sm{nullptr, 0};
        time_it("empty 640 B args, 14.9 KB LDS, ext launch", [&](hipEvent_t x, hipEvent_t y) { hipExtLaunchKernelGGL(k_empty, dim3(478), dim3(320), 14896, st, x, y, 0, a); }, st);
        time_it("empty 640 B args, no LDS, ext launch", [&](hipEvent_t x, hipEvent_t y) { hipExtLaunchKernelGGL(k_empty, dim3(478), dim3(320), 0, st, x, y, 0, a); }, st);
        time_it("empty 16 B args, no LDS, ext launch", [&](hipEvent_t x, hipEvent_t y) { hipExtLaunchKernelGGL(k_empty_small, dim3(478), dim3(320), 0, st, x, y, 0, sm); }, st);
        time_it("empty 16 B args, no LDS, 1 workgroup", [&](hipEvent_t x, hipEvent_t y) { hipExtLaunchKernelGGL(k_empty_small, dim3(1), dim3(64), 0, st, x, y, 0, sm); }, st);
        time_it("empty 640 B args, 14.9 KB LDS, <<<>>> (events unused)", [&](hipEvent_t, hipEvent_t) { k_empty<<<478, 320, 14896, st>>>(a); }, st);
        time_it("empty 16 B args, no LDS, <<<>>> (events unused)", [&](hipEvent_t, hipEvent_t) { k_empty_small<<<478, 320, 0, st>>>(sm); }, st);
        time_it("empty 16 B args, 1 workgroup, <<<>>> (events unused)", [&](hipEvent_t, hipEvent_t) { k_empty_small<<<1, 64, 0, st>>>(sm); }, st);
    }
    const long n_out = 1200032;             // doubles written by one cfg-2 evaluation (c rows + Jacobian values)
    double *out, *in;
    CK(hipMalloc(&out, n_out * 8)); CK(hipMalloc(&in, 4096 * 8)); CK(hipMemset(in, 0, 4096 * 8));
    for (int grid : {478}) {
        for (int block : {320}) {
            Args a{out, in, n_out, (int)((n_out + grid - 1) / grid), 0, {}};
            const size_t lds = 14896;
            char nm[128];
            snprintf(nm, sizeof nm, "empty          grid %d block %d", grid, block);
            time_it(nm, [&](hipEvent_t x, hipEvent_t y) { hipExtLaunchKernelGGL(k_empty, dim3(grid), dim3(block), lds, st, x, y, 0, a); }, st);
            for (int mode = 0; mode < 3; ++mode) {
                a.mode = mode;
                snprintf(nm, sizeof nm, "store 9.6 MB +%d round trips grid %d block %d", mode, grid, block);
                time_it(nm, [&](hipEvent_t x, hipEvent_t y) { hipExtLaunchKernelGGL(k_store, dim3(grid), dim3(block), lds, st, x, y, 0, a); }, st);
            }
        }
    }
    return 0;
}
