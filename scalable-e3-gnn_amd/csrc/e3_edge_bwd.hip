// Backward of the edge / node stages around the tensor product (forward: e3_edge_ops.hip) -- what the force head
// (-dE/dpos, BASELINE.json configs[3]) and parameter gradients through a whole SEGNN layer need besides the tensor
// product's own backward (e3_l1tp_backward / e3_tp_backward).  Builder-defined stages; the
// reference's contract for its operator is plain torch autograd (l1_tensor_prod.py:234-299), and these kernels are that
// for the stages it does not contain.  fp32, one wave per CSR row where a row is reduced, HBM-bound streaming kernels.
#include "e3_common.h"

#include <algorithm>

namespace e3 {

constexpr float kS3 = 1.7320508075688772f, kS5 = 2.2360679774997896f;

// Y = [1 | sqrt3 u | sqrt5 b(u)], u = r / |r|, r = x_src - x_dst; d = |r|; A_i = [1 | mean over the row of Y_e[1:]].
//   g_Yeff_e = gY_e + gA_dst[1:] / deg      (components >= 1)
//   g_u      = sqrt3 gY1 + sqrt5 (db/du)^T gY2
//   g_r      = (g_u - u (u . g_u)) / d + gd u          (0 for d = 0: the forward uses u = 0 there)
//   gpos[src] += g_r   (atomics),   gpos[dst] -= sum_e g_r   (wave reduction, one atomic per row and component)
template <int LMAX>
__global__ __launch_bounds__(256) void edge_geometry_bwd_kernel(const float4* __restrict__ pos4,
                                                                const int32_t* __restrict__ rowptr,
                                                                const int32_t* __restrict__ src, int64_t N,
                                                                const float* __restrict__ gY, const float* __restrict__ gd,
                                                                const float* __restrict__ gA, float* __restrict__ gpos) {
  constexpr int NY = (LMAX + 1) * (LMAX + 1);
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave0; i < N; i += nw) {
    const int b = rowptr[i], e = rowptr[i + 1];
    if (b == e) continue;
    const float4 pi = pos4[i];
    const float invdeg = 1.0f / (float)(e - b);
    float ga[NY];
#pragma unroll
    for (int k = 1; k < NY; ++k) ga[k] = gA ? gA[i * NY + k] * invdeg : 0.f;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int q = b + lane; q < e; q += 64) {
      const int j = src[q];
      const float4 pj = pos4[j];
      const float rx = pj.x - pi.x, ry = pj.y - pi.y, rz = pj.z - pi.z;
      const float d = sqrtf(rx * rx + ry * ry + rz * rz);
      if (!(d > 0.f)) continue;
      const float inv = 1.0f / d;
      const float x = rx * inv, y = ry * inv, z = rz * inv;
      float g[NY];
#pragma unroll
      for (int k = 1; k < NY; ++k) g[k] = (gY ? gY[(int64_t)q * NY + k] : 0.f) + ga[k];
      float ux = kS3 * g[1], uy = kS3 * g[2], uz = kS3 * g[3];
      if constexpr (LMAX == 2) {
        // b0 = s3 x y, b1 = s3 y z, b2 = (2 z^2 - x^2 - y^2) / 2, b3 = s3 z x, b4 = (s3 / 2)(x^2 - y^2)
        const float c = kS5;
        ux += c * (kS3 * y * g[4] - x * g[6] + kS3 * z * g[7] + kS3 * x * g[8]);
        uy += c * (kS3 * x * g[4] + kS3 * z * g[5] - y * g[6] - kS3 * y * g[8]);
        uz += c * (kS3 * y * g[5] + 2.f * z * g[6] + kS3 * x * g[7]);
      }
      const float dot = x * ux + y * uy + z * uz;
      const float gdv = gd ? gd[q] : 0.f;
      const float gx = (ux - x * dot) * inv + gdv * x, gy = (uy - y * dot) * inv + gdv * y,
                  gz = (uz - z * dot) * inv + gdv * z;
      atomicAdd(gpos + (int64_t)j * 3 + 0, gx);
      atomicAdd(gpos + (int64_t)j * 3 + 1, gy);
      atomicAdd(gpos + (int64_t)j * 3 + 2, gz);
      sx += gx; sy += gy; sz += gz;
    }
    for (int o = 32; o > 0; o >>= 1) {
      sx += __shfl_xor(sx, o);
      sy += __shfl_xor(sy, o);
      sz += __shfl_xor(sz, o);
    }
    if (lane == 0) {
      atomicAdd(gpos + i * 3 + 0, -sx);
      atomicAdd(gpos + i * 3 + 1, -sy);
      atomicAdd(gpos + i * 3 + 2, -sz);
    }
  }
}

// m[e] = [h[dst] | h[src] | extra[e]]:  g_h[dst] += sum over the row of gm[e, :D] (one atomic per row and column),
// g_h[src[e]] += gm[e, D:2D] (atomics), g_extra[e] = gm[e, 2D:]
__global__ __launch_bounds__(256) void gather_concat_bwd_kernel(const float* __restrict__ gm, int64_t ld_gm, int D,
                                                                const int32_t* __restrict__ rowptr,
                                                                const int32_t* __restrict__ src, int64_t N, int n_extra,
                                                                float* __restrict__ gh, int64_t ld_gh,
                                                                float* __restrict__ gextra) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave0; i < N; i += nw) {
    const int b = rowptr[i], e = rowptr[i + 1];
    if (b == e) continue;
    for (int c = lane; c < D; c += 64) {
      float acc = 0.f;
      for (int q = b; q < e; ++q) {
        const float* g = gm + (int64_t)q * ld_gm;
        acc += g[c];
        atomicAdd(gh + (int64_t)src[q] * ld_gh + c, g[D + c]);
      }
      atomicAdd(gh + i * ld_gh + c, acc);
    }
    if (gextra)
      for (int q = b; q < e; ++q)
        if (lane < n_extra) gextra[(int64_t)q * n_extra + lane] = gm[(int64_t)q * ld_gm + 2 * D + lane];
  }
}

struct GateBlocksB {
  int nblocks;
  int l[8], mul[8];
};
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + __expf(-x)); }

// out = [silu(s) | sigmoid(g_k) v_k]:  d silu = sig (1 + s (1 - sig));  g_v = sig(g) gout;  g_g = sig (1 - sig) sum_m v gout
// One thread per INPUT column (scalar, gate or gated component): the gate column sums over its channel's components.
__global__ __launch_bounds__(256) void gate_blocks_bwd_kernel(const float* __restrict__ in, int64_t ld_in,
                                                              const float* __restrict__ gout, int64_t ld_go,
                                                              float* __restrict__ gin, int64_t ld_gi, int64_t B, int ns,
                                                              int ngates, int W, GateBlocksB gb) {
  const int WI = W + ngates;  // input width: [ns | ngates | wide]
  const int64_t total = B * (int64_t)WI;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / WI;
    const int c = (int)(t - row * WI);
    const float* x = in + row * ld_in;
    const float* go = gout + row * ld_go;
    float v;
    if (c < ns) {
      const float s = x[c], sg = sigm(s);
      v = go[c] * sg * (1.f + s * (1.f - sg));
    } else if (c < ns + ngates) {
      // gate k: find its block and channel
      int k = c - ns, bi = 0, col0 = ns;  // col0: first OUTPUT column of block bi
      while (k >= gb.mul[bi]) { k -= gb.mul[bi]; col0 += gb.mul[bi] * (2 * gb.l[bi] + 1); ++bi; }
      const int w = 2 * gb.l[bi] + 1;
      float acc = 0.f;
      for (int m = 0; m < w; ++m) acc += x[ngates + col0 + k * w + m] * go[col0 + k * w + m];
      const float sg = sigm(x[c]);
      v = acc * sg * (1.f - sg);
    } else {
      const int oc = c - ngates;  // output column
      int rem = oc - ns, g0 = 0, bi = 0;
      while (rem >= gb.mul[bi] * (2 * gb.l[bi] + 1)) { rem -= gb.mul[bi] * (2 * gb.l[bi] + 1); g0 += gb.mul[bi]; ++bi; }
      const int k = rem / (2 * gb.l[bi] + 1);
      v = sigm(x[ns + g0 + k]) * go[oc];
    }
    gin[row * ld_gi + c] = v;
  }
}

// agg[i] = sum over the row of msg[e]  ->  gmsg[e] = gagg[dst(e)]
__global__ __launch_bounds__(256) void segment_sum_bwd_kernel(const float* __restrict__ gagg, int64_t ld_ga,
                                                              const int32_t* __restrict__ rowptr, int64_t N, int D,
                                                              float* __restrict__ gmsg, int64_t ld_gm) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = wave0; i < N; i += nw) {
    const int b = rowptr[i], e = rowptr[i + 1];
    for (int c = lane; c < D; c += 64) {
      const float v = gagg[i * ld_ga + c];
      for (int q = b; q < e; ++q) gmsg[(int64_t)q * ld_gm + c] = v;
    }
  }
}

static inline int wave_grid_b(int64_t N) { return (int)std::max<int64_t>(1, std::min<int64_t>((N + 3) / 4, 256 * 16)); }

}  // namespace e3

using namespace e3;

extern "C" {

int e3_edge_geometry_backward(const float* pos4, const int32_t* rowptr, const int32_t* src, int64_t N, int lmax,
                              const float* g_edge_y, const float* g_edge_d, const float* g_node_a, float* g_pos,
                              void* stream) {
  if (N < 0 || (lmax != 1 && lmax != 2)) return E3_ERR_INVALID_ARG;
  if (N == 0) return E3_OK;
  if (!pos4 || !rowptr || !src || !g_pos) return E3_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  E3_HIP_CHECK(hipMemsetAsync(g_pos, 0, (size_t)N * 3 * sizeof(float), s));
  if (lmax == 1)
    hipLaunchKernelGGL(edge_geometry_bwd_kernel<1>, dim3(wave_grid_b(N)), dim3(256), 0, s, (const float4*)pos4, rowptr, src,
                       N, g_edge_y, g_edge_d, g_node_a, g_pos);
  else
    hipLaunchKernelGGL(edge_geometry_bwd_kernel<2>, dim3(wave_grid_b(N)), dim3(256), 0, s, (const float4*)pos4, rowptr, src,
                       N, g_edge_y, g_edge_d, g_node_a, g_pos);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

int e3_gather_concat_backward(const float* g_out, int64_t ld_gout, int D, const int32_t* rowptr, const int32_t* src,
                              int64_t N, int n_extra, float* g_h, int64_t ld_gh, float* g_extra, void* stream) {
  if (N < 0 || D <= 0 || n_extra < 0 || n_extra > 64 || ld_gout < 2 * D + n_extra || ld_gh < D) return E3_ERR_INVALID_ARG;
  if (N == 0) return E3_OK;
  if (!g_out || !rowptr || !src || !g_h) return E3_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  E3_HIP_CHECK(hipMemset2DAsync(g_h, (size_t)ld_gh * 4, 0, (size_t)D * 4, (size_t)N, s));
  hipLaunchKernelGGL(gather_concat_bwd_kernel, dim3(wave_grid_b(N)), dim3(256), 0, s, g_out, ld_gout, D, rowptr, src, N,
                     n_extra, g_h, ld_gh, n_extra > 0 ? g_extra : nullptr);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

int e3_gate_blocks_backward(const float* in, int64_t ld_in, const float* g_out, int64_t ld_gout, float* g_in,
                            int64_t ld_gin, int64_t B, int ns, int nblocks, const int32_t* ls, const int32_t* muls,
                            void* stream) {
  if (B < 0 || ns < 0 || nblocks < 0 || nblocks > 8 || (nblocks > 0 && (!ls || !muls))) return E3_ERR_INVALID_ARG;
  GateBlocksB gb;
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
  if (ld_in < W + ngates || ld_gout < W || ld_gin < W + ngates) return E3_ERR_INVALID_ARG;
  if (B == 0 || W == 0) return E3_OK;
  if (!in || !g_out || !g_in) return E3_ERR_INVALID_ARG;
  const int64_t total = B * (int64_t)(W + ngates);
  const int grid = (int)std::min<int64_t>((total + 255) / 256, 256 * 16);
  hipLaunchKernelGGL(gate_blocks_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, ld_in, g_out, ld_gout, g_in,
                     ld_gin, B, ns, ngates, W, gb);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

int e3_segment_sum_backward(const float* g_agg, int64_t ld_gagg, const int32_t* rowptr, int64_t N, int D, float* g_msg,
                            int64_t ld_gmsg, void* stream) {
  if (N < 0 || D <= 0 || ld_gagg < D || ld_gmsg < D) return E3_ERR_INVALID_ARG;
  if (N == 0) return E3_OK;
  if (!g_agg || !rowptr || !g_msg) return E3_ERR_INVALID_ARG;
  hipLaunchKernelGGL(segment_sum_bwd_kernel, dim3(wave_grid_b(N)), dim3(256), 0, (hipStream_t)stream, g_agg, ld_gagg,
                     rowptr, N, D, g_msg, ld_gmsg);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

}  // extern "C"
