// Edge / node stages around the tensor product (builder-defined; spec in include/e3gnn.h).
// All of them are HBM-bound streaming kernels: one wave per CSR row (dst node) where a row reduction
// or a per-row broadcast is involved, 256-B coalesced row segments everywhere.
#include "e3_common.h"

#include <algorithm>

namespace e3 {

constexpr float kSqrt3 = 1.7320508075688772f;

// one wave per dst node: lanes over the row's edges; wave-reduce the mean of Y1
__global__ __launch_bounds__(256) void edge_geometry_kernel(const float4* __restrict__ pos4,
                                                            const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ src, int64_t N,
                                                            float4* __restrict__ edge_y, float* __restrict__ edge_d,
                                                            float4* __restrict__ node_a) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave0; i < N; i += nw) {
    const int b = rowptr[i], e = rowptr[i + 1];
    const float4 pi = pos4[i];
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int q = b + lane; q < e; q += 64) {
      const float4 pj = pos4[src[q]];
      const float rx = pj.x - pi.x, ry = pj.y - pi.y, rz = pj.z - pi.z;
      const float d = sqrtf(rx * rx + ry * ry + rz * rz);
      const float s = d > 0.f ? kSqrt3 / d : 0.f;
      const float4 y = make_float4(1.0f, s * rx, s * ry, s * rz);
      if (edge_y) edge_y[q] = y;
      if (edge_d) edge_d[q] = d;
      sx += y.y; sy += y.z; sz += y.w;
    }
    if (node_a) {
      for (int o = 32; o > 0; o >>= 1) {
        sx += __shfl_xor(sx, o);
        sy += __shfl_xor(sy, o);
        sz += __shfl_xor(sz, o);
      }
      if (lane == 0) {
        const float inv = e > b ? 1.0f / (float)(e - b) : 0.f;
        node_a[i] = make_float4(1.0f, sx * inv, sy * inv, sz * inv);
      }
    }
  }
}

// one wave per dst node; h[dst] is read once per row and re-used for all its edges
__global__ __launch_bounds__(256) void gather_concat_kernel(const float* __restrict__ h, int64_t ld_h, int D,
                                                            const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ src, int64_t N,
                                                            const float* __restrict__ extra, int n_extra,
                                                            float* __restrict__ out, int64_t ld_out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave0; i < N; i += nw) {
    const int b = rowptr[i], e = rowptr[i + 1];
    if (b == e) continue;
    const float* hi = h + i * ld_h;
    for (int c0 = 0; c0 < D; c0 += 64) {
      const int c = c0 + lane;
      const float vi = c < D ? hi[c] : 0.f;
      for (int q = b; q < e; ++q) {
        if (c < D) {
          float* o = out + (int64_t)q * ld_out;
          o[c] = vi;
          o[D + c] = h[(int64_t)src[q] * ld_h + c];
        }
      }
    }
    for (int q = b; q < e; ++q)
      if (lane < n_extra) out[(int64_t)q * ld_out + 2 * D + lane] = extra[(int64_t)q * n_extra + lane];
  }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__global__ __launch_bounds__(256) void gate_kernel(const float* __restrict__ in, int64_t ld_in, float* __restrict__ out,
                                                   int64_t ld_out, int64_t B, int ns, int nv) {
  const int W = ns + 3 * nv;  // output width
  const int64_t total = B * (int64_t)W;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / W;
    const int c = (int)(t - row * W);
    const float* x = in + row * ld_in;
    float v;
    if (c < ns) {
      const float s = x[c];
      v = s * sigmoidf_(s);
    } else {
      const int k = (c - ns) / 3;
      v = sigmoidf_(x[ns + k]) * x[ns + nv + (c - ns)];
    }
    out[row * ld_out + c] = v;
  }
}

// one wave per dst node, lanes over columns, sequential over the row's edges (fixed order)
__global__ __launch_bounds__(256) void segment_sum_kernel(const float* __restrict__ msg, int64_t ld_msg,
                                                          const int32_t* __restrict__ rowptr, int64_t N, int D,
                                                          float* __restrict__ agg, int64_t ld_agg) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave0; i < N; i += nw) {
    const int b = rowptr[i], e = rowptr[i + 1];
    for (int c = lane; c < D; c += 64) {
      float acc = 0.f;
      const float* m = msg + (int64_t)b * ld_msg + c;
#pragma unroll 4
      for (int q = b; q < e; ++q) {
        acc += *m;
        m += ld_msg;
      }
      agg[i * ld_agg + c] = acc;
    }
  }
}


// l <= 2 variant: Y [E,9] = [1 | sqrt3 u | sqrt5 b(u)], b = l=2 basis of oracle/cg.py; A [N,9] = [1 | mean Y_1..8]
__global__ __launch_bounds__(256) void edge_geometry_l2_kernel(const float4* __restrict__ pos4,
                                                               const int32_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ src, int64_t N,
                                                               float* __restrict__ edge_y, float* __restrict__ edge_d,
                                                               float* __restrict__ node_a) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const float s5 = 2.2360679774997896f;
  for (int64_t i = wave0; i < N; i += nw) {
    const int b = rowptr[i], e = rowptr[i + 1];
    const float4 pi = pos4[i];
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = b + lane; q < e; q += 64) {
      const float4 pj = pos4[src[q]];
      const float rx = pj.x - pi.x, ry = pj.y - pi.y, rz = pj.z - pi.z;
      const float d = sqrtf(rx * rx + ry * ry + rz * rz);
      const float inv = d > 0.f ? 1.0f / d : 0.f;
      const float x = rx * inv, y = ry * inv, z = rz * inv;
      float Y[9];
      Y[0] = 1.0f;
      Y[1] = kSqrt3 * x; Y[2] = kSqrt3 * y; Y[3] = kSqrt3 * z;
      Y[4] = s5 * kSqrt3 * x * y;
      Y[5] = s5 * kSqrt3 * y * z;
      Y[6] = s5 * 0.5f * (2.f * z * z - x * x - y * y);
      Y[7] = s5 * kSqrt3 * z * x;
      Y[8] = s5 * 0.5f * kSqrt3 * (x * x - y * y);
      if (edge_y) {
        float* o = edge_y + (int64_t)q * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] = Y[k];
      }
      if (edge_d) edge_d[q] = d;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += Y[k + 1];
    }
    if (node_a) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
      if (lane == 0) {
        const float inv = e > b ? 1.0f / (float)(e - b) : 0.f;
        float* a = node_a + i * 9;
        a[0] = 1.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k + 1] = acc[k] * inv;
      }
    }
  }
}

struct GateBlocks {
  int nblocks;
  int l[8], mul[8];
};

__global__ __launch_bounds__(256) void gate_blocks_kernel(const float* __restrict__ in, int64_t ld_in,
                                                          float* __restrict__ out, int64_t ld_out, int64_t B, int ns,
                                                          int ngates, int W, GateBlocks gb) {
  const int64_t total = B * (int64_t)W;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / W;
    int c = (int)(t - row * W);
    const float* x = in + row * ld_in;
    float v;
    if (c < ns) {
      const float s = x[c];
      v = s * sigmoidf_(s);
    } else {
      int rem = c - ns, g0 = 0, bi = 0;
      while (rem >= gb.mul[bi] * (2 * gb.l[bi] + 1)) { rem -= gb.mul[bi] * (2 * gb.l[bi] + 1); g0 += gb.mul[bi]; ++bi; }
      const int k = rem / (2 * gb.l[bi] + 1);
      v = sigmoidf_(x[ns + g0 + k]) * x[ngates + c];
    }
    out[row * ld_out + c] = v;
  }
}

// bf16 storage variant: fp32 accumulation, one rounding at the end
__global__ __launch_bounds__(256) void segment_sum_bf16_kernel(const bf16* __restrict__ msg, int64_t ld_msg,
                                                               const int32_t* __restrict__ rowptr, int64_t N, int D,
                                                               bf16* __restrict__ agg, int64_t ld_agg) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave0; i < N; i += nw) {
    const int b = rowptr[i], e = rowptr[i + 1];
    for (int c = lane; c < D; c += 64) {
      float acc = 0.f;
      const bf16* m = msg + (int64_t)b * ld_msg + c;
#pragma unroll 4
      for (int q = b; q < e; ++q) {
        acc += __bfloat162float(*m);
        m += ld_msg;
      }
      agg[i * ld_agg + c] = __float2bfloat16(acc);
    }
  }
}

static inline int wave_grid(int64_t N) { return (int)std::max<int64_t>(1, std::min<int64_t>((N + 3) / 4, 256 * 16)); }

}  // namespace e3

using namespace e3;

extern "C" {

int e3_edge_geometry(const float* pos4, const int32_t* rowptr, const int32_t* src, int64_t N, float* edge_y,
                     float* edge_d, float* node_a, void* stream) {
  if (N < 0) return E3_ERR_INVALID_ARG;
  if (N == 0) return E3_OK;
  if (!pos4 || !rowptr || !src || (!edge_y && !node_a)) return E3_ERR_INVALID_ARG;
  hipLaunchKernelGGL(edge_geometry_kernel, dim3(wave_grid(N)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)pos4, rowptr, src, N, (float4*)edge_y, edge_d, (float4*)node_a);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

int e3_gather_concat(const float* h, int64_t ld_h, int D, const int32_t* rowptr, const int32_t* src, int64_t N,
                     const float* extra, int n_extra, float* out, int64_t ld_out, void* stream) {
  if (N < 0 || D <= 0 || n_extra < 0 || n_extra > 64 || ld_out < 2 * D + n_extra) return E3_ERR_INVALID_ARG;
  if (N == 0) return E3_OK;
  if (!h || !rowptr || !src || !out || (n_extra > 0 && !extra)) return E3_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gather_concat_kernel, dim3(wave_grid(N)), dim3(256), 0, (hipStream_t)stream, h, ld_h, D, rowptr,
                     src, N, extra, n_extra, out, ld_out);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

int e3_gate(const float* in, int64_t ld_in, float* out, int64_t ld_out, int64_t B, int ns, int nv, void* stream) {
  if (B < 0 || ns < 0 || nv < 0 || ld_in < ns + 4 * nv || ld_out < ns + 3 * nv) return E3_ERR_INVALID_ARG;
  if (B == 0 || ns + nv == 0) return E3_OK;
  if (!in || !out) return E3_ERR_INVALID_ARG;
  int64_t total = B * (int64_t)(ns + 3 * nv);
  int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(gate_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, ld_in, out, ld_out, B, ns, nv);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

int e3_segment_sum(const float* msg, int64_t ld_msg, const int32_t* rowptr, int64_t N, int D, float* agg,
                   int64_t ld_agg, void* stream) {
  if (N < 0 || D <= 0 || ld_msg < D || ld_agg < D) return E3_ERR_INVALID_ARG;
  if (N == 0) return E3_OK;
  if (!msg || !rowptr || !agg) return E3_ERR_INVALID_ARG;
  hipLaunchKernelGGL(segment_sum_kernel, dim3(wave_grid(N)), dim3(256), 0, (hipStream_t)stream, msg, ld_msg, rowptr, N,
                     D, agg, ld_agg);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

int e3_segment_sum_bf16(const void* msg, int64_t ld_msg, const int32_t* rowptr, int64_t N, int D, void* agg,
                        int64_t ld_agg, void* stream) {
  if (N < 0 || D <= 0 || ld_msg < D || ld_agg < D) return E3_ERR_INVALID_ARG;
  if (N == 0) return E3_OK;
  if (!msg || !rowptr || !agg) return E3_ERR_INVALID_ARG;
  hipLaunchKernelGGL(segment_sum_bf16_kernel, dim3(wave_grid(N)), dim3(256), 0, (hipStream_t)stream, (const bf16*)msg,
                     ld_msg, rowptr, N, D, (bf16*)agg, ld_agg);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

int e3_edge_geometry_l2(const float* pos4, const int32_t* rowptr, const int32_t* src, int64_t N, float* edge_y,
                        float* edge_d, float* node_a, void* stream) {
  if (N < 0) return E3_ERR_INVALID_ARG;
  if (N == 0) return E3_OK;
  if (!pos4 || !rowptr || !src || (!edge_y && !node_a)) return E3_ERR_INVALID_ARG;
  hipLaunchKernelGGL(edge_geometry_l2_kernel, dim3(wave_grid(N)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)pos4, rowptr, src, N, edge_y, edge_d, node_a);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

int e3_gate_blocks(const float* in, int64_t ld_in, float* out, int64_t ld_out, int64_t B, int ns, int nblocks,
                   const int32_t* ls, const int32_t* muls, void* stream) {
  if (B < 0 || ns < 0 || nblocks < 0 || nblocks > 8 || (nblocks > 0 && (!ls || !muls))) return E3_ERR_INVALID_ARG;
  GateBlocks gb;
  gb.nblocks = nblocks;
  int ngates = 0, wide = 0;
  for (int i = 0; i < 8; ++i) { gb.l[i] = 0; gb.mul[i] = 1 << 30; }
  for (int i = 0; i < nblocks; ++i) {
    if (ls[i] < 0 || ls[i] > 2 || muls[i] < 0) return E3_ERR_INVALID_ARG;
    gb.l[i] = ls[i]; gb.mul[i] = muls[i];
    ngates += muls[i];
    wide += muls[i] * (2 * ls[i] + 1);
  }
  const int W = ns + wide;
  if (ld_in < W + ngates || ld_out < W) return E3_ERR_INVALID_ARG;
  if (B == 0 || W == 0) return E3_OK;
  if (!in || !out) return E3_ERR_INVALID_ARG;
  int64_t total = B * (int64_t)W;
  int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(gate_blocks_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, ld_in, out, ld_out, B, ns,
                     ngates, W, gb);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

}  // extern "C"
