// Microbenchmark: global -> LDS DMA (global_load_lds_dwordx4) throughput per CU as a function of the
// shape of one 1-KiB piece (what 64 lanes x 16 B fetch): how many distinct 128-B lines, how long the
// contiguous runs.  One 512-thread block per CU, every wave issues PIECES pieces per "stage" and
// waits with a counted vmcnt, as conv_fwd256_kernel does.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

// mode: run = bytes of each contiguous run inside a piece (16, 32, 64, 128, 1024); rows of a piece are
// `row_stride` bytes apart (the tensor's pixel pitch).  Each block walks its own 256-row x K window.
template <int PIECES>
__global__ __launch_bounds__(512) void dma_kernel(const unsigned char *__restrict__ src, long block_stride,
                                                  int run, long row_stride, int nstage, int wrap, int *sink) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * PIECES * 8 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char *base = src + (long)blockIdx.x * block_stride;
    const int lanes_per_run = run / 16;                    // lanes covering one contiguous run
    const int runs_per_piece = 64 / lanes_per_run;         // rows per piece
    const int r_in_piece = lane / lanes_per_run, off_in_run = (lane % lanes_per_run) * 16;
    for (int s = 0; s < nstage; ++s) {
        unsigned char *dst = smem + (s % 3) * (PIECES * 8 * 1024) + wave * (PIECES * 1024);
#pragma unroll
        for (int pc = 0; pc < PIECES; ++pc) {
            // piece pc of wave `wave` in stage s: rows (wave*PIECES+pc)*runs_per_piece + r, k offset s*run
            const long row = (long)(wave * PIECES + pc) * runs_per_piece + r_in_piece;
            const unsigned char *g = base + row * row_stride + (long)(s % wrap) * run + off_in_run;
            __builtin_amdgcn_global_load_lds((glb_void *)g, (lds_void *)(dst + pc * 1024), 16, 0, 0);
        }
        if (s >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (smem[threadIdx.x] == 123 && sink) sink[0] = 1;
}

int main() {
    const int ncu = 256;
    const long block_stride = 8L << 20;                    // 8 MiB window per block (L2 / MALL resident mix)
    unsigned char *src;
    int *sink;
    hipMalloc(&src, block_stride * ncu + (64 << 20));
    hipMalloc(&sink, 4);
    hipMemset(src, 1, block_stride * ncu + (64 << 20));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct Case { const char *name; int run; long row_stride; };
    const Case cases[] = {
        {"32 rows x 32 B, pitch 512 B (conv A, C=256)", 32, 512},
        {"32 rows x 32 B, pitch 128 B (C=64)", 32, 128},
        {"32 rows x 32 B, pitch 4608 B (weights K=2304)", 32, 4608},
        {"16 rows x 64 B, pitch 512 B", 64, 512},
        {"8 rows x 128 B, pitch 512 B", 128, 512},
        {"8 rows x 128 B, pitch 4608 B", 128, 4608},
        {"1 x 1024 B contiguous (pitch 1024)", 1024, 1024},
        {"64 rows x 16 B, pitch 512 B", 16, 512},
    };
    const int nstage = 2048;
    for (int l2 = 0; l2 < 3; ++l2) {
    printf("---- %s\n", l2 == 0 ? "streaming: 8 MiB window per block, no re-read" :
                         l2 == 1 ? "own window per block, re-read every 16 stages (L2-hot, like conv A over taps)" :
                                   "ONE window shared by all blocks, re-read (L2-hot, like conv weights)");
    for (const Case &c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(dma_kernel<4>, dim3(ncu), dim3(512), 0, 0, src, l2 == 2 ? 0L : block_stride, c.run,
                               c.row_stride, nstage, l2 == 0 ? 1 << 30 : 16, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep == 1) {
                const double bytes = (double)ncu * nstage * 8 * 4 * 1024;
                const double clk = ms * 1e-3 * 2.4e9;
                printf("%-48s %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU  %5.1f clk/piece/CU\n", c.name, ms, bytes / ms / 1e9,
                       bytes / ncu / clk, clk / (nstage * 32.0));
            }
        }
    }
    }
    printf("---- waves issuing per CU (shared L2-hot window), 4 pieces per wave and stage\n");
    for (int nw = 1; nw <= 8; nw *= 2) {
        for (int ci : {0, 3, 4, 6}) {
            const Case &c = cases[ci];
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(dma_kernel<4>, dim3(ncu), dim3(64 * nw), 0, 0, src, 0L, c.run, c.row_stride, nstage, 16, sink);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double clk = ms * 1e-3 * 2.4e9;
            printf("waves %d  %-44s %6.1f clk/piece/CU  %6.1f clk/piece/wave\n", nw, c.name, clk / (nstage * 4.0 * nw),
                   clk / (nstage * 4.0));
        }
    }
    return 0;
}
