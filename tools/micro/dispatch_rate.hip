// How fast can the chip start workgroups?  Empty / tiny kernels with the launch shape of conv_fwd_kernel
// (256 threads, 48 KB static LDS, a ~350-byte by-value parameter block), many workgroups; and a store-only
// kernel that writes what the 128x128 tile's epilogue writes (64 KB per workgroup) to bound the store path.
//   hipcc --offload-arch=gfx950 -O3 -o dispatch_rate dispatch_rate.hip && ./dispatch_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

struct Params { float *out; int n; int pad[84]; };

template <int LDS_KB, int REGS>
__global__ __launch_bounds__(256) void empty_kernel(const Params p) {
    __shared__ unsigned char smem[LDS_KB * 1024];
    if (p.n < 0) { smem[threadIdx.x] = 1; __syncthreads(); p.out[threadIdx.x] = smem[(threadIdx.x + 1) & 255]; }
}

// each workgroup writes a 128 x 128 tile of 4-byte elements as 16-B stores (rows of 512 B inside a row-major
// matrix with row_elems columns) -- the parts-only epilogue's traffic without anything else
__global__ __launch_bounds__(256) void store_tile_kernel(float *out, int gn, long row_elems) {
    const int bid = blockIdx.x;
    const long m0 = (long)(bid / gn) * 128, n0 = (long)(bid % gn) * 128;
    const int t = threadIdx.x;
    const float4 v = make_float4((float)bid, 1.f, 2.f, 3.f);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int row = (t >> 5) + 8 * q, col = 4 * (t & 31);
        *(float4 *)(out + (m0 + row) * row_elems + n0 + col) = v;
    }
}

// persistent form: 256 * blocks_per_cu workgroups walk the tile list
__global__ __launch_bounds__(256) void store_tile_persistent(float *out, int gn, long row_elems, int ntiles) {
    const int t = threadIdx.x;
    for (int bid = blockIdx.x; bid < ntiles; bid += gridDim.x) {
        const long m0 = (long)(bid / gn) * 128, n0 = (long)(bid % gn) * 128;
        const float4 v = make_float4((float)bid, 1.f, 2.f, 3.f);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (t >> 5) + 8 * q, col = 4 * (t & 31);
            *(float4 *)(out + (m0 + row) * row_elems + n0 + col) = v;
        }
    }
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F>
static float time_ms(F f, int iters = 20) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < iters; ++i) f();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / iters;
}

int main() {
    Params p;
    const long M = 1048576, C = 256;
    CHECK(hipMalloc(&p.out, M * C * 4));
    p.n = 1;
    for (int nwg : {256, 2048, 16384, 65536}) {
        const float t0 = time_ms([&] { hipLaunchKernelGGL((empty_kernel<1, 0>), dim3(nwg), dim3(256), 0, 0, p); });
        const float t1 = time_ms([&] { hipLaunchKernelGGL((empty_kernel<48, 0>), dim3(nwg), dim3(256), 0, 0, p); });
        printf("empty kernel, %6d workgroups of 256 threads: 1 KB LDS %.4f ms (%.1f ns per WG), 48 KB LDS %.4f ms (%.1f ns per WG)\n",
               nwg, t0, t0 * 1e6 / nwg, t1, t1 * 1e6 / nwg);
    }
    const int gn = (int)(C / 128), ntiles = (int)(M / 128) * gn;
    const double gb = (double)M * C * 4 / 1e9;
    const float ts = time_ms([&] { hipLaunchKernelGGL(store_tile_kernel, dim3(ntiles), dim3(256), 0, 0, p.out, gn, C); });
    printf("store-only, one 128x128 tile (64 KB) per workgroup, %d workgroups: %.4f ms = %.2f TB/s\n", ntiles, ts, gb / ts);
    for (int per_cu : {1, 2, 4, 8}) {
        const float tp = time_ms([&] {
            hipLaunchKernelGGL(store_tile_persistent, dim3(256 * per_cu), dim3(256), 0, 0, p.out, gn, C, ntiles);
        });
        printf("store-only, persistent, %d workgroups per CU: %.4f ms = %.2f TB/s\n", per_cu, tp, gb / tp);
    }
    return 0;
}
