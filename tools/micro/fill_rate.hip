// Zero-fill rate of a 1.4 GB region (the four gradient maps of the pyramid RoIAlign backward at 16 x 1024^2):
// hipMemsetAsync against 16-byte-store kernels of several shapes.
//   hipcc --offload-arch=gfx950 -O3 -o fill_rate fill_rate.hip && ./fill_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(256) void fill_loop(float4 *p, long n) {         // grid-stride
    const float4 z = make_float4(0, 0, 0, 0);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = z;
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void fill_once(float4 *p, long n) {         // block = U x 4 KB, no loop
    const float4 z = make_float4(0, 0, 0, 0);
    const long base = (long)blockIdx.x * 256 * U + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long i = base + u * 256;
        if (i < n) {
            if (NT) { typedef float v4 __attribute__((ext_vector_type(4))); __builtin_nontemporal_store((v4){0, 0, 0, 0}, (v4 *)(p + i)); } else p[i] = z;
        }
    }
}

template <int U>
__global__ __launch_bounds__(256) void fill_chunk(float4 *p, long n, long per) {   // block owns one contiguous chunk
    const float4 z = make_float4(0, 0, 0, 0);
    long i = (long)blockIdx.x * per + threadIdx.x;
    const long e = min(n, (long)(blockIdx.x + 1) * per);
    for (; i + (U - 1) * 256 < e; i += U * 256)
#pragma unroll
        for (int u = 0; u < U; ++u) p[i + u * 256] = z;
    for (; i < e; i += 256) p[i] = z;
}

template <class F> static void run(const char *name, size_t bytes, F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a, 0);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    printf("%-36s %.4f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6);
}

int main() {
    const size_t bytes = 1426063360ull;
    float4 *p; hipMalloc(&p, bytes);
    const long n = bytes / 16;
    run("hipMemsetAsync", bytes, [&] { hipMemsetAsync(p, 0, bytes, 0); });
    for (int g : {2048, 4096, 8192, 16384, 65536}) {
        char nm[64]; snprintf(nm, 64, "grid-stride, %d blocks", g);
        run(nm, bytes, [&] { hipLaunchKernelGGL(fill_loop, dim3(g), dim3(256), 0, 0, p, n); });
    }
    run("once U=1", bytes, [&] { hipLaunchKernelGGL((fill_once<1, false>), dim3((n + 255) / 256), dim3(256), 0, 0, p, n); });
    run("once U=4", bytes, [&] { hipLaunchKernelGGL((fill_once<4, false>), dim3((n + 1023) / 1024), dim3(256), 0, 0, p, n); });
    run("once U=8", bytes, [&] { hipLaunchKernelGGL((fill_once<8, false>), dim3((n + 2047) / 2048), dim3(256), 0, 0, p, n); });
    run("once U=16", bytes, [&] { hipLaunchKernelGGL((fill_once<16, false>), dim3((n + 4095) / 4096), dim3(256), 0, 0, p, n); });
    run("once U=4 nontemporal", bytes, [&] { hipLaunchKernelGGL((fill_once<4, true>), dim3((n + 1023) / 1024), dim3(256), 0, 0, p, n); });
    run("once U=16 nontemporal", bytes, [&] { hipLaunchKernelGGL((fill_once<16, true>), dim3((n + 4095) / 4096), dim3(256), 0, 0, p, n); });
    for (int g : {2048, 4096, 16384}) {
        char nm[64]; snprintf(nm, 64, "chunk U=4, %d blocks", g);
        const long per = ((n + g - 1) / g + 255) / 256 * 256;
        run(nm, bytes, [&] { hipLaunchKernelGGL((fill_chunk<4>), dim3(g), dim3(256), 0, 0, p, n, per); });
    }
    return 0;
}
