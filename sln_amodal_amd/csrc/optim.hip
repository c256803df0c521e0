// Global-norm clip + momentum SGD over the whole parameter set in three launches (gfx950).
//
// The reference clips with torch.nn.utils.clip_grad_norm(parameters, 5.0) and then steps
// torch.optim.SGD(momentum 0.9, weight decay 1e-4 on the non-'bn' group) (model.py:352-358, 441-444):
// one norm kernel per tensor, a host-side sum, one scaling kernel per tensor, and 3-4 elementwise
// kernels per tensor for the update.  Here the tensors are cut into fixed-size chunks described by
// device-resident tables, so a launch covers every tensor:
//   sqnorm_partial_kernel   sum of squares per chunk, float64 accumulation
//   sqnorm_final_kernel     ordered sum of the partials -> squared total norm (one block)
//   sgd_clip_kernel         g' = g * min(1, max_norm / (norm + 1e-6));  d = g' + wd * p;
//                           buf = momentum * buf + d;  p = p - lr * buf
//                           (nothing at all when the norm is inf / NaN: the step is skipped and counted)
// HBM-bound: 4 B read in pass 1; 12 B read + 8 B written per element in pass 3.  The clipped
// gradient is not written back (nothing reads it after the step).  Rounds like the eager sequence
// (compiled with -ffp-contract=off: one rounding per operation).
#include "common.h"

#define OPT_THREADS 256

__global__ __launch_bounds__(OPT_THREADS) void sqnorm_partial_kernel(const float *const *__restrict__ grads,
                                                                     const int64_t *__restrict__ numel,
                                                                     const int32_t *__restrict__ chunk_tensor,
                                                                     const int64_t *__restrict__ chunk_offset,
                                                                     int chunk_elems, double *__restrict__ partial) {
    __shared__ double s_red[OPT_THREADS / SLN_WAVE];
    const int t = threadIdx.x;
    const int ti = chunk_tensor[blockIdx.x];
    const int64_t off = chunk_offset[blockIdx.x];
    const float *g = grads[ti] + off;
    const int64_t n = min((int64_t)chunk_elems, numel[ti] - off);
    double acc = 0.0;
    if ((((uintptr_t)g) & 15) == 0) {
        const int64_t n4 = n >> 2;
        const float4 *g4 = (const float4 *)g;
        for (int64_t i = t; i < n4; i += OPT_THREADS) {
            const float4 v = g4[i];
            acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
        for (int64_t i = (n4 << 2) + t; i < n; i += OPT_THREADS) acc += (double)g[i] * g[i];
    } else {
        for (int64_t i = t; i < n; i += OPT_THREADS) acc += (double)g[i] * g[i];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((t & 63) == 0) s_red[t >> 6] = acc;
    __syncthreads();
    if (t == 0) partial[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ __launch_bounds__(1024) void sqnorm_final_kernel(const double *__restrict__ partial, int n,
                                                            double *__restrict__ sqnorm) {
    __shared__ double s_red[16];
    const int t = threadIdx.x;
    double acc = 0.0;
    for (int i = t; i < n; i += 1024) acc += partial[i];       // fixed assignment: reproducible
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((t & 63) == 0) s_red[t >> 6] = acc;
    __syncthreads();
    if (t == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += s_red[w];
        sqnorm[0] = s;
    }
}

__device__ __forceinline__ void sgd_one(float &p, float g, float &b, float cf, float wd, float momentum,
                                        float lr) {
    const float gc = g * cf;
    const float d = gc + wd * p;
    b = momentum * b + d;
    p = p - lr * b;
}

__global__ __launch_bounds__(OPT_THREADS) void sgd_clip_kernel(float *const *__restrict__ params,
                                                               const float *const *__restrict__ grads,
                                                               float *const *__restrict__ bufs,
                                                               const int64_t *__restrict__ numel,
                                                               const float *__restrict__ weight_decay,
                                                               const int32_t *__restrict__ chunk_tensor,
                                                               const int64_t *__restrict__ chunk_offset,
                                                               int chunk_elems, const double *__restrict__ sqnorm,
                                                               float max_norm, float lr, float momentum,
                                                               int32_t *__restrict__ skipped) {
    const int t = threadIdx.x;
    // A non-finite gradient norm (an inf / NaN anywhere in the step's gradients) skips the whole update --
    // weights and momentum buffers untouched, counted on the device; no host round trip decides it.  This is a
    // GUARD OF THIS BUILD, not reference behaviour: the reference would apply the NaN (its `continue`s at
    // model.py:416-418, 433-434 skip batches without ground truth, not non-finite gradients).  The count is
    // therefore surfaced everywhere a step is driven -- ClippedSGD.skipped_steps(), asserted zero by the GPU
    // training tests and by bench.py, logged per epoch by MaskRCNN.train_epoch -- so that a kernel bug which
    // produces a NaN gradient reads as an error, not as slow learning.
    const double sq = sqnorm[0];
    if (!(sq >= 0.0 && sq < INFINITY)) {
        if (skipped && blockIdx.x == 0 && t == 0) atomicAdd(skipped, 1);
        return;
    }
    const int ti = chunk_tensor[blockIdx.x];
    const int64_t off = chunk_offset[blockIdx.x];
    float *p = params[ti] + off, *b = bufs[ti] + off;
    const float *g = grads[ti] + off;
    const int64_t n = min((int64_t)chunk_elems, numel[ti] - off);
    const float wd = weight_decay[ti];
    // clip_grad_norm: clip_coef = max_norm / (total_norm + 1e-6), applied only when < 1
    const double coef = (double)max_norm / (sqrt(sq) + 1e-6);
    const float cf = coef < 1.0 ? (float)coef : 1.0f;
    if (((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)b)) & 15) == 0) {
        const int64_t n4 = n >> 2;
        float4 *p4 = (float4 *)p, *b4 = (float4 *)b;
        const float4 *g4 = (const float4 *)g;
        for (int64_t i = t; i < n4; i += OPT_THREADS) {
            float4 pv = p4[i], bv = b4[i];
            const float4 gv = g4[i];
            sgd_one(pv.x, gv.x, bv.x, cf, wd, momentum, lr);
            sgd_one(pv.y, gv.y, bv.y, cf, wd, momentum, lr);
            sgd_one(pv.z, gv.z, bv.z, cf, wd, momentum, lr);
            sgd_one(pv.w, gv.w, bv.w, cf, wd, momentum, lr);
            p4[i] = pv;
            b4[i] = bv;
        }
        for (int64_t i = (n4 << 2) + t; i < n; i += OPT_THREADS) sgd_one(p[i], g[i], b[i], cf, wd, momentum, lr);
    } else {
        for (int64_t i = t; i < n; i += OPT_THREADS) sgd_one(p[i], g[i], b[i], cf, wd, momentum, lr);
    }
}

extern "C" int sln_grad_sqnorm_f32(const float *const *grads, const int64_t *numel, const int32_t *chunk_tensor,
                                   const int64_t *chunk_offset, int n_chunks, int chunk_elems, double *partial,
                                   double *sqnorm, sln_stream_t stream) {
    if (n_chunks < 0 || chunk_elems < 1 || !sqnorm) return SLN_ERR_INVALID_ARG;
    if (n_chunks > 0 && (!grads || !numel || !chunk_tensor || !chunk_offset || !partial)) return SLN_ERR_INVALID_ARG;
    sln_enter();
    hipStream_t st = (hipStream_t)stream;
    if (n_chunks > 0)
        hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(n_chunks), dim3(OPT_THREADS), 0, st, grads, numel,
                           chunk_tensor, chunk_offset, chunk_elems, partial);
    hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(1024), 0, st, partial, n_chunks, sqnorm);
    return sln_launch_status();
}

extern "C" int sln_sgd_clip_step_f32(float *const *params, const float *const *grads, float *const *bufs,
                                     const int64_t *numel, const float *weight_decay, const int32_t *chunk_tensor,
                                     const int64_t *chunk_offset, int n_chunks, int chunk_elems,
                                     const double *sqnorm, float max_norm, float lr, float momentum,
                                     int32_t *skipped, sln_stream_t stream) {
    if (n_chunks < 0 || chunk_elems < 1) return SLN_ERR_INVALID_ARG;
    if (n_chunks == 0) return SLN_OK;
    if (!params || !grads || !bufs || !numel || !weight_decay || !chunk_tensor || !chunk_offset || !sqnorm)
        return SLN_ERR_INVALID_ARG;
    sln_enter();
    hipLaunchKernelGGL(sgd_clip_kernel, dim3(n_chunks), dim3(OPT_THREADS), 0, (hipStream_t)stream, params, grads,
                       bufs, numel, weight_decay, chunk_tensor, chunk_offset, chunk_elems, sqnorm, max_norm, lr,
                       momentum, skipped);
    return sln_launch_status();
}
