// LDS-DMA issue-path microbenchmark (gfx950): bytes per clock per CU of `buffer_load_dwordx4 ... lds` as a function of the
// shape of one instruction's 64 x 16 B = 1 KB: R rows of 1024/R contiguous bytes at a row stride of S bytes.
// A GEMM operand tile [rows][64 k] bf16 is the R = 8 case (8 rows x 128 B).  Source windows stay L2-resident.
//   hipcc --offload-arch=gfx950 -O3 -o dma_bench dma_bench.hip && ./dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void dma_kernel(const char* __restrict__ src, size_t src_bytes, int rows_per_instr,
                                                         int row_stride, int iters, int window_rows, uint64_t* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (uint32_t)src_bytes, 0x00020000);
    const int run = 1024 / rows_per_instr;                 // contiguous bytes per row
    const int lanes_per_row = run / 16;
    const int r = lane / lanes_per_row, c = lane % lanes_per_row;
    // each block owns a window of `window_rows` rows (L2 resident after the first pass), each wave a slice of it
    const uint32_t base = (uint32_t)(((size_t)blockIdx.x * window_rows) % (src_bytes / row_stride - window_rows)) * (uint32_t)row_stride;
    uint32_t voff = base + (uint32_t)r * row_stride + c * 16;
    char* dst = smem + wid * 8192;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
    int row = wid * rows_per_instr;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(dst + u * 1024), 16, voff + (uint32_t)row * row_stride, 0, 0, 0);
            row += WAVES * rows_per_instr;
            if (row + rows_per_instr > window_rows) row = wid * rows_per_instr;
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint64_t t1 = __builtin_readcyclecounter();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    const size_t bytes = 512ull << 20;
    char* src;
    uint64_t* cyc;
    hipMalloc(&src, bytes);
    hipMemset(src, 1, bytes);
    hipMalloc(&cyc, 1024 * sizeof(uint64_t));
    const int blocks = 256, iters = 200;
    printf("%-8s %-10s %-8s %-8s %12s\n", "waves", "rows/instr", "stride", "window", "B/clk/CU");
    for (int waves : {4, 8}) {
        for (int stride : {1536, 6144, 1024, 128}) {
            for (int R : {1, 2, 4, 8, 16}) {
                if (1024 / R > stride && stride != 128) continue;      // rows would overlap
                if (stride == 128 && R != 8) continue;                  // dense 128-B rows: the tile-major (pre-packed) layout
                const int window_rows = 512;                             // 512 rows of the matrix per block
                const size_t need = (size_t)(blocks + 1) * window_rows * stride;
                if (need > bytes) continue;
                auto launch = [&]() {
                    if (waves == 4) hipLaunchKernelGGL(dma_kernel<4>, dim3(blocks), dim3(256), 65536, 0, src, bytes, R, stride, iters, window_rows, cyc);
                    else hipLaunchKernelGGL(dma_kernel<8>, dim3(blocks), dim3(512), 65536, 0, src, bytes, R, stride, iters, window_rows, cyc);
                };
                launch(); launch();
                hipDeviceSynchronize();
                std::vector<uint64_t> h(blocks);
                hipMemcpy(h.data(), cyc, blocks * sizeof(uint64_t), hipMemcpyDeviceToHost);
                double avg = 0;
                for (auto v : h) avg += (double)v;
                avg /= blocks;
                const double bpc = (double)waves * iters * 8 * 1024 / avg;
                printf("%-8d %-10d %-8d %-8d %12.1f\n", waves, R, stride, window_rows, bpc);
            }
        }
    }
    return 0;
}
