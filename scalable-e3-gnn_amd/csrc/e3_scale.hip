// Power-of-two operand scales for the fp16-split MFMA kernels (e3_tp_mfma_core.h: split2_f16).
//
// The split keeps 22 significant bits of an fp32 operand while its lo half is a normal fp16 number; the callers
// therefore scale a tensor so that its largest magnitude lands at 2^target (features: 2^10, leaving headroom for the
// CG / spherical-harmonic factors of the per-row features; weights: 2^13 at pack time).  A scale is a device-resident
// pair {s, 1/s}: nothing crosses to the host, the consuming kernel reads it in its prologue.
#include "e3_common.h"

#include <algorithm>

namespace e3 {

#include "e3_tp_mfma_core.h"

struct ScaleArgs {
  const float* base[4];
  int64_t rows[4], ld[4];
  int cols[4];
  int n;
};

// out4: [0] s, [1] 1/s, [2] float bits of the running max |x| (zeroed on the stream before this kernel), [3] unused
// A segment whose rows are whole float4s (cols and ld multiples of 4, 16-byte aligned base) is read 16 bytes per lane:
// dense segments as one flat range, strided ones row by row (a workgroup per row step, no per-element division).
__global__ __launch_bounds__(256) void absmax_kernel(ScaleArgs a, uint32_t* bits) {
  float m = 0.f;
  for (int t = 0; t < a.n; ++t) {
    const int64_t rows = a.rows[t], ld = a.ld[t];
    const int cols = a.cols[t];
    const float* base = a.base[t];
    const bool vec = ((cols | ld) & 3) == 0 && (reinterpret_cast<uintptr_t>(base) & 15) == 0;
    if (vec && ld == cols) {
      const float4* b4 = reinterpret_cast<const float4*>(base);
      const int64_t n4 = rows * cols / 4;
      for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = b4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
    } else if (vec) {
      const int c4 = cols / 4;
      for (int64_t r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {  // a wave per row
        const float4* b4 = reinterpret_cast<const float4*>(base + r * ld);
        for (int c = threadIdx.x & 63; c < c4; c += 64) {
          const float4 v = b4[c];
          m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
      }
    } else {
      for (int64_t r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4)
        for (int c = threadIdx.x & 63; c < cols; c += 64) m = fmaxf(m, fabsf(base[r * ld + c]));
    }
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  // NaN / inf inputs leave the scale at the finite maximum seen so far (the product then propagates them itself)
  if ((threadIdx.x & 63) == 0 && m > 0.f && m < INFINITY) atomicMax(bits, __builtin_bit_cast(uint32_t, m));
}

__global__ void scale_finalize_kernel(float* out4, int target) {
  const float s = pow2_scale_from_bits(reinterpret_cast<const uint32_t*>(out4)[2], target);
  out4[0] = s;
  out4[1] = 1.0f / s;
}

// h_new = h + u (row-major, same shape) and the running max |h_new| in one pass: the residual update of a SEGNN layer
// produces the next layer's operand scale for free
__global__ __launch_bounds__(256) void add_absmax_kernel(const float4* __restrict__ h, const float4* __restrict__ u,
                                                         float4* __restrict__ o, int64_t n4, uint32_t* bits) {
  float m = 0.f;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = h[i], b = u[i];
    const float4 r = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    o[i] = r;
    m = fmaxf(fmaxf(m, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
  }
  for (int o2 = 32; o2 > 0; o2 >>= 1) m = fmaxf(m, __shfl_xor(m, o2));
  if ((threadIdx.x & 63) == 0 && m > 0.f && m < INFINITY) atomicMax(bits, __builtin_bit_cast(uint32_t, m));
}

int scale_finalize(float* out4, int target_log2, hipStream_t s) {
  hipLaunchKernelGGL(scale_finalize_kernel, dim3(1), dim3(1), 0, s, out4, target_log2);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

}  // namespace e3

using namespace e3;

extern "C" {

int e3_pow2_scale(const e3_tp_segment* segs, const int64_t* nrows, int nseg, int target_log2, float* out4,
                  void* stream) {
  if (!segs || !nrows || nseg < 1 || nseg > 4 || !out4 || target_log2 < -20 || target_log2 > 14) return E3_ERR_INVALID_ARG;
  ScaleArgs a;
  int64_t total = 0;
  for (int i = 0; i < 4; ++i) { a.base[i] = nullptr; a.rows[i] = 0; a.ld[i] = 0; a.cols[i] = 0; }
  for (int i = 0; i < nseg; ++i) {
    if (nrows[i] < 0 || segs[i].ncols <= 0 || segs[i].ld < segs[i].ncols || (nrows[i] > 0 && !segs[i].base))
      return E3_ERR_INVALID_ARG;
    a.base[i] = (const float*)segs[i].base;
    a.rows[i] = nrows[i];
    a.ld[i] = segs[i].ld;
    a.cols[i] = segs[i].ncols;
    total += nrows[i] * segs[i].ncols;
  }
  a.n = nseg;
  hipStream_t s = (hipStream_t)stream;
  E3_HIP_CHECK(hipMemsetAsync(out4, 0, 16, s));
  if (total > 0) {
    const int grid = (int)std::min<int64_t>((total + 1023) / 1024, 2048);
    hipLaunchKernelGGL(absmax_kernel, dim3(grid), dim3(256), 0, s, a, reinterpret_cast<uint32_t*>(out4) + 2);
  }
  hipLaunchKernelGGL(scale_finalize_kernel, dim3(1), dim3(1), 0, s, out4, target_log2);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

int e3_add_pow2_scale(const float* h, const float* u, float* out, int64_t n, int target_log2, float* out4,
                      void* stream) {
  if (n < 0 || !out4 || (n > 0 && (!h || !u || !out)) || (n & 3) || target_log2 < -20 || target_log2 > 14)
    return E3_ERR_INVALID_ARG;
  if (((uintptr_t)h | (uintptr_t)u | (uintptr_t)out) & 15) return E3_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  E3_HIP_CHECK(hipMemsetAsync(out4, 0, 16, s));
  if (n > 0) {
    const int64_t n4 = n / 4;
    const int grid = (int)std::min<int64_t>((n4 + 255) / 256, 2048);
    hipLaunchKernelGGL(add_absmax_kernel, dim3(grid), dim3(256), 0, s, (const float4*)h, (const float4*)u, (float4*)out,
                       n4, reinterpret_cast<uint32_t*>(out4) + 2);
  }
  hipLaunchKernelGGL(scale_finalize_kernel, dim3(1), dim3(1), 0, s, out4, target_log2);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

}  // extern "C"
