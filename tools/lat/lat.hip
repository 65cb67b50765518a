// Serial-latency probe for lane-serial kernels (the JPEG entropy decoder): ns and shader cycles per step of a dependent chain of
// (a) integer ALU ops, (b) LDS reads, (c) global loads that hit L2 / L1 (pointer chase over 64 KB), run by ONE 16-lane workgroup
// and by 16 workgroups.   hipcc --offload-arch=gfx950 -O3 tools/lat/lat.hip -o tools/lat/lat && tools/lat/lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_alu(int n, unsigned* out, long long* cyc) {
    unsigned x = threadIdx.x + 1;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) x = x * 1664525u + 1013904223u;
    cyc[blockIdx.x] = clock64() - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void k_lds(int n, unsigned* out, long long* cyc) {
    __shared__ unsigned tab[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = (i * 97 + 13) & 4095;
    __syncthreads();
    unsigned x = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) x = tab[x];
    cyc[blockIdx.x] = clock64() - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void k_glb(int n, const unsigned* tab, unsigned* out, long long* cyc) {
    unsigned x = threadIdx.x * 64 % 16384;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) x = tab[x];
    cyc[blockIdx.x] = clock64() - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main() {
    const int n = 200000;
    unsigned *out, *tab; long long* cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 4096); hipMalloc(&tab, 16384 * 4);
    std::vector<unsigned> h(16384);
    for (int i = 0; i < 16384; ++i) h[i] = (i * 4099 + 17) % 16384;
    hipMemcpy(tab, h.data(), 16384 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {1, 16, 256}) {
        for (int kind = 0; kind < 3; ++kind) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k_alu, dim3(blocks), dim3(16), 0, 0, n, out, cyc);
                if (kind == 1) hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(16), 0, 0, n, out, cyc);
                if (kind == 2) hipLaunchKernelGGL(k_glb, dim3(blocks), dim3(16), 0, 0, n, tab, out, cyc);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("blocks %3d %s: %7.1f ns/step, %6.1f cycles/step (clock64), kernel %.2f ms -> %.0f MHz\n", blocks,
                   kind == 0 ? "alu chain (mul+add)" : kind == 1 ? "lds chain          " : "global chain (L2)  ", ms * 1e6 / n, (double)c / n, ms,
                   c / (ms * 1e3));
        }
    }
    return 0;
}
