// Tuning probe: what a dependent kernel boundary inside a hipGraph costs on this box, by launch shape -- grid size, kernel
// argument bytes, static LDS, launch bounds.   hipcc --offload-arch=gfx950 -O3 -o tools/boundary_probe tools/boundary_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { char b[600]; };
__global__ void k_plain(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void k_args(Big a, Big b, int* p) { if (p && threadIdx.x == 9999) p[0] = a.b[5] + b.b[7]; }
__global__ void k_lds(int* p) {
    __shared__ float s[4352];
    if (p && threadIdx.x == 9999) { s[threadIdx.x] = 1.f; p[0] = (int)s[3]; }
}
__global__ __launch_bounds__(256, 7) void k_lb(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void k_touch(Big a, int* p) {      // reads its arguments (scalar loads) like a real kernel
    __shared__ float s[4352];
    int v = 0;
    for (int i = 0; i < 600; i += 64) v += a.b[i];
    if (p && v == 123456) { s[threadIdx.x] = v; p[0] = (int)s[1]; }
}
template <typename F>
float run(F launch, int n) {
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) launch(st);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms;
    }
    return best * 1e3f / n;
}
int main() {
    int* d; hipMalloc(&d, 64);
    Big a{}, b{};
    const int n = 200;
    for (int grid : {1, 64, 256, 1178}) {
        printf("grid %4d x 256:  plain %.2f us  | 1.2 KB of arguments %.2f | 17 KB static LDS %.2f | launch_bounds(256,7) %.2f | args read + LDS %.2f\n", grid,
               run([&](hipStream_t s) { hipLaunchKernelGGL(k_plain, dim3(grid), dim3(256), 0, s, (int*)nullptr); }, n),
               run([&](hipStream_t s) { hipLaunchKernelGGL(k_args, dim3(grid), dim3(256), 0, s, a, b, (int*)nullptr); }, n),
               run([&](hipStream_t s) { hipLaunchKernelGGL(k_lds, dim3(grid), dim3(256), 0, s, (int*)nullptr); }, n),
               run([&](hipStream_t s) { hipLaunchKernelGGL(k_lb, dim3(grid), dim3(256), 0, s, (int*)nullptr); }, n),
               run([&](hipStream_t s) { hipLaunchKernelGGL(k_touch, dim3(grid), dim3(256), 0, s, a, (int*)nullptr); }, n));
    }
    // the same on a CU-masked stream is not probed here (tools/cumask_probe.hip)
    return 0;
}
