// What clock does ONE resident workgroup get on an otherwise idle MI355X?  (round 6: the sweep server of sfm_solve_pnp_ransac is one
// workgroup; its 28 x 6 butterfly + one point per lane took ~15 us per request — the shader clock, not the instruction count.)
// A workgroup of 1024 lanes runs a fixed chain of dependent fp64 FMAs; s_memtime (shader clock) against s_memrealtime (100 MHz)
// gives the clock it ran at.  Cases: after 200 ms of idle; the same launch repeated back-to-back; right after 30 ms of a chip-wide load.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
__global__ __launch_bounds__(1024) void chain_kernel(long long* out, int n, double seed) {
    const long long r0 = wall_clock64(), c0 = clock64();
    double x = seed + threadIdx.x;
    for (int i = 0; i < n; ++i) x = __builtin_fma(x, 1.0000001, 0.5);
    const long long c1 = clock64(), r1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = r1 - r0; out[1] = c1 - c0; }
    if (x == 12345.678) out[2] = 1;
}
__global__ void load_kernel(double* sink, int n) {
    double x = threadIdx.x;
    for (int i = 0; i < n; ++i) x = __builtin_fma(x, 1.0000001, 0.5);
    if (x == 12345.678) sink[0] = x;
}
int main() {
    long long *d, h[3];
    double* sink;
    hipMalloc(&d, 64); hipMalloc(&sink, 64);
    auto run = [&](const char* what, int n) {
        hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(1024), 0, 0, d, n, 1.0);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-44s n=%6d  %8.2f us  %10lld shader cycles  => %7.1f MHz  (%.1f cycles per dependent FMA)\n", what, n, h[0] / 100.0, h[1], h[1] / (h[0] / 100.0), (double)h[1] / n);
    };
    run("first launch (cold)", 2000);
    for (int rep = 0; rep < 3; ++rep) {
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
        run("after 200 ms idle", 2000);
        run("  again, back-to-back", 2000);
        run("  again, back-to-back", 2000);
        run("  longer chain", 200000);
        run("  again, back-to-back", 2000);
    }
    hipLaunchKernelGGL(load_kernel, dim3(4096), dim3(256), 0, 0, sink, 3000000);
    hipDeviceSynchronize();
    run("right after ~30+ ms of chip-wide load", 2000);
    run("  again", 2000);
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
    run("  5 ms later", 2000);
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    run("  50 ms later", 2000);
    return 0;
}
