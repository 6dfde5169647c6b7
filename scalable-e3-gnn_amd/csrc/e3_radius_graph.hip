// Radius graph on the GPU: Morton keys -> stable radix sort -> cell table -> per-cell LDS-staged
// neighbour scan, emitting CSR-by-dst with ascending src.  Spec: include/e3gnn.h (builder-defined,
// SURVEY.md §8a-N1).  All edge decisions use explicitly rounded fp32 operations (no FMA contraction)
// so that the CPU oracle reproduces them bit for bit.
#include "e3_common.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>

namespace e3 {

constexpr int kCandCap = 1024;  // candidates staged per chunk (16 KiB of LDS per wave)

__host__ __device__ inline uint32_t spread3(uint32_t v) {  // 10 bits -> every third bit
  v &= 0x3ff;
  v = (v | (v << 16)) & 0x030000FF;
  v = (v | (v << 8)) & 0x0300F00F;
  v = (v | (v << 4)) & 0x030C30C3;
  v = (v | (v << 2)) & 0x09249249;
  return v;
}
__host__ __device__ inline uint32_t compact3(uint32_t v) {
  v &= 0x09249249;
  v = (v | (v >> 2)) & 0x030C30C3;
  v = (v | (v >> 4)) & 0x0300F00F;
  v = (v | (v >> 8)) & 0x030000FF;
  v = (v | (v >> 16)) & 0x3ff;
  return v;
}
__host__ __device__ inline uint32_t morton3(int cx, int cy, int cz) {
  return spread3((uint32_t)cx) | (spread3((uint32_t)cy) << 1) | (spread3((uint32_t)cz) << 2);
}

struct RgDev {
  float lo[3], inv[3];
  int n[3];
  float r2;
  int bits;
};

__device__ __forceinline__ int cell_of(float p, float lo, float inv, int n) {
  float t = __fmul_rn(__fsub_rn(p, lo), inv);
  int c = (int)floorf(t);
  return c < 0 ? 0 : (c > n - 1 ? n - 1 : c);
}

__global__ void rg_keys_kernel(const float* __restrict__ pos, int64_t N, RgDev g, uint32_t* __restrict__ keys,
                               int32_t* __restrict__ idx) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  int cx = cell_of(pos[3 * i + 0], g.lo[0], g.inv[0], g.n[0]);
  int cy = cell_of(pos[3 * i + 1], g.lo[1], g.inv[1], g.n[1]);
  int cz = cell_of(pos[3 * i + 2], g.lo[2], g.inv[2], g.n[2]);
  keys[i] = morton3(cx, cy, cz);
  idx[i] = (int32_t)i;
}

// sorted positions (x,y,z,0), cell table [begin,end) per Morton code, list of non-empty cells
__global__ void rg_cells_kernel(const float* __restrict__ pos, int64_t N, const uint32_t* __restrict__ skeys,
                                const int32_t* __restrict__ perm, float4* __restrict__ spos,
                                int32_t* __restrict__ cbegin, int32_t* __restrict__ cend,
                                int32_t* __restrict__ heads, int32_t* __restrict__ nheads) {
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int32_t o = perm[i];
  spos[i] = make_float4(pos[3 * (int64_t)o], pos[3 * (int64_t)o + 1], pos[3 * (int64_t)o + 2], 0.0f);
  const uint32_t k = skeys[i];
  if (i == 0 || skeys[i - 1] != k) {
    cbegin[k] = (int32_t)i;
    heads[atomicAdd(nheads, 1)] = (int32_t)i;
  }
  if (i == N - 1 || skeys[i + 1] != k) cend[k] = (int32_t)(i + 1);
}

// One wave per non-empty cell.  FILL=false: deg[i] = #neighbours.  FILL=true: src[rowptr[i]..] = ids.
template <bool FILL>
__global__ __launch_bounds__(64) void rg_scan_kernel(const float4* __restrict__ spos, const uint32_t* __restrict__ skeys,
                                                     const int32_t* __restrict__ cbegin,
                                                     const int32_t* __restrict__ cend,
                                                     const int32_t* __restrict__ heads,
                                                     const int32_t* __restrict__ nheads_p, RgDev g,
                                                     int32_t* __restrict__ deg, const int32_t* __restrict__ rowptr,
                                                     int32_t* __restrict__ src) {
  __shared__ float4 cand[kCandCap];   // x,y,z, id (bit pattern)
  __shared__ int rb[28], re[28], pre[29];
  const int lane = threadIdx.x;
  const int nheads = *nheads_p;
  for (int h = blockIdx.x; h < nheads; h += gridDim.x) {
    const int cb = heads[h];
    const uint32_t code = skeys[cb];
    const int ce = cend[code];
    const int cx = (int)compact3(code), cy = (int)compact3(code >> 1), cz = (int)compact3(code >> 2);
    // the 27 neighbour cells, ordered by Morton code => candidate ids ascend
    uint32_t ncode = 0xFFFFFFFFu;
    int b = 0, e = 0;
    if (lane < 27) {
      int dx = lane % 3 - 1, dy = (lane / 3) % 3 - 1, dz = lane / 9 - 1;
      int x = cx + dx, y = cy + dy, z = cz + dz;
      if (x >= 0 && x < g.n[0] && y >= 0 && y < g.n[1] && z >= 0 && z < g.n[2]) {
        ncode = morton3(x, y, z);
        b = cbegin[ncode];
        e = cend[ncode];
        if (e <= b) ncode = 0xFFFFFFFFu;  // empty
      }
    }
    int rank = 0;
    for (int m = 0; m < 27; ++m) {
      uint32_t other = __shfl(ncode, m);
      rank += (other < ncode) || (other == ncode && m < lane);
    }
    if (lane < 27) {
      rb[rank] = (ncode == 0xFFFFFFFFu) ? 0 : b;
      re[rank] = (ncode == 0xFFFFFFFFu) ? 0 : e;
    }
    __syncthreads();
    if (lane == 0) {
      int acc = 0;
      for (int m = 0; m < 27; ++m) { pre[m] = acc; acc += re[m] - rb[m]; }
      pre[27] = acc;
    }
    __syncthreads();
    const int ncand = pre[27];
    for (int c0 = 0; c0 < ncand; c0 += kCandCap) {
      const int nc = min(kCandCap, ncand - c0);
      for (int f = lane; f < nc; f += 64) {
        const int ff = c0 + f;
        int m = 0;
        while (ff >= pre[m + 1]) ++m;
        const int gid = rb[m] + (ff - pre[m]);
        float4 p = spos[gid];
        p.w = __int_as_float(gid);
        cand[f] = p;
      }
      __syncthreads();
      for (int i = cb; i < ce; ++i) {
        const float4 pi = spos[i];
        int cnt = 0;
        // deg[] carries the running per-row count between candidate chunks (only when ncand > kCandCap);
        // agent-scope atomics keep that hand-over out of the per-CU L1.
        const int prev = (c0 > 0) ? __hip_atomic_load(&deg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        const int cursor = FILL ? (rowptr[i] + prev) : 0;
        for (int base = 0; base < nc; base += 64) {
          const int c = base + lane;
          bool ok = false;
          int id = -1;
          if (c < nc) {
            const float4 pc = cand[c];
            id = __float_as_int(pc.w);
            const float dx = __fsub_rn(pi.x, pc.x), dy = __fsub_rn(pi.y, pc.y), dz = __fsub_rn(pi.z, pc.z);
            const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            ok = (id != i) && (d2 <= g.r2);
          }
          const unsigned long long mask = __ballot(ok);
          if (FILL && ok) src[cursor + cnt + __popcll(mask & ((1ull << lane) - 1ull))] = id;
          cnt += __popcll(mask);
        }
        if (lane == 0 && (!FILL || c0 + kCandCap < ncand))
          __hip_atomic_store(&deg[i], prev + cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
    }
  }
}

struct RgWs {  // workspace carve-up (bytes, 256-aligned)
  size_t keys, skeys, idx, cbegin, cend, heads, nheads, deg, cub, cub_bytes, total;
};
static RgWs rg_ws(int64_t N, int bits) {
  RgWs w;
  size_t pos = 0;
  auto take = [&](size_t bytes) { size_t o = pos; pos += (bytes + 255) / 256 * 256; return o; };
  size_t ncode = (size_t)1 << (3 * bits);
  w.keys = take(N * 4); w.skeys = take((N + 1) * 4); w.idx = take(N * 4);
  w.cbegin = take(ncode * 4); w.cend = take(ncode * 4);
  w.heads = take(N * 4); w.nheads = take(256); w.deg = take((N + 1) * 4);
  size_t s1 = 0, s2 = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, s1, (uint32_t*)nullptr, (uint32_t*)nullptr, (int32_t*)nullptr,
                                     (int32_t*)nullptr, (int)N, 0, 3 * bits);
  hipcub::DeviceScan::ExclusiveSum(nullptr, s2, (int32_t*)nullptr, (int32_t*)nullptr, (int)(N + 1));
  w.cub_bytes = std::max(s1, s2) + 256;
  w.cub = take(w.cub_bytes);
  w.total = pos;
  return w;
}

static RgDev rg_dev(const e3_rg_params* p) {
  RgDev g;
  for (int a = 0; a < 3; ++a) { g.lo[a] = p->lo[a]; g.inv[a] = p->inv[a]; g.n[a] = p->n[a]; }
  g.r2 = p->r * p->r;  // fp32 product
  g.bits = p->bits;
  return g;
}

}  // namespace e3

using namespace e3;

extern "C" {

int e3_rg_grid(e3_rg_params* p) {
  if (!p || !(p->r > 0.0f)) return E3_ERR_INVALID_ARG;
  int nmax = 1;
  for (int a = 0; a < 3; ++a) {
    float ext = p->hi[a] - p->lo[a];
    if (!(ext > 0.0f)) return E3_ERR_INVALID_ARG;
    float q = floorf(ext / (p->r * 1.0001f));
    int n = q < 1.0f ? 1 : (q > 256.0f ? 256 : (int)q);
    p->n[a] = n;
    p->inv[a] = (float)n / ext;
    nmax = std::max(nmax, n);
  }
  int bits = 1;
  while ((1 << bits) < nmax) ++bits;
  p->bits = bits;
  return E3_OK;
}

int64_t e3_rg_workspace_bytes(int64_t N, const e3_rg_params* p) {
  if (!p || N < 0 || N >= (1ll << 31) - 2 || p->bits < 1 || p->bits > 8) return -1;
  return (int64_t)rg_ws(N, p->bits).total;
}

int e3_rg_sort_count(const float* pos, int64_t N, const e3_rg_params* p, int32_t* perm, float* sorted_pos4,
                     int32_t* rowptr, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!p || N < 0 || !rowptr || p->bits < 1 || p->bits > 8) return E3_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) { E3_HIP_CHECK(hipMemsetAsync(rowptr, 0, 4, s)); return E3_OK; }
  if (!pos || !perm || !sorted_pos4 || !workspace) return E3_ERR_INVALID_ARG;
  RgWs w = rg_ws(N, p->bits);
  if ((int64_t)w.total > workspace_bytes) return E3_ERR_INVALID_ARG;
  char* ws = (char*)workspace;
  RgDev g = rg_dev(p);
  uint32_t* keys = (uint32_t*)(ws + w.keys);
  uint32_t* skeys = (uint32_t*)(ws + w.skeys);
  int32_t* idx = (int32_t*)(ws + w.idx);
  int32_t* cbegin = (int32_t*)(ws + w.cbegin);
  int32_t* cend = (int32_t*)(ws + w.cend);
  int32_t* heads = (int32_t*)(ws + w.heads);
  int32_t* nheads = (int32_t*)(ws + w.nheads);
  int32_t* deg = (int32_t*)(ws + w.deg);
  const int nb = (int)((N + 255) / 256);
  hipLaunchKernelGGL(rg_keys_kernel, dim3(nb), dim3(256), 0, s, pos, N, g, keys, idx);
  size_t cb = w.cub_bytes;
  (void)hipGetLastError();
  E3_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(ws + w.cub, cb, keys, skeys, idx, perm, (int)N, 0, 3 * p->bits, s));
  size_t ncode = (size_t)1 << (3 * p->bits);
  E3_HIP_CHECK(hipMemsetAsync(cbegin, 0, ncode * 4, s));
  E3_HIP_CHECK(hipMemsetAsync(cend, 0, ncode * 4, s));
  E3_HIP_CHECK(hipMemsetAsync(nheads, 0, 4, s));
  hipLaunchKernelGGL(rg_cells_kernel, dim3(nb), dim3(256), 0, s, pos, N, skeys, perm, (float4*)sorted_pos4, cbegin,
                     cend, heads, nheads);
  const int grid = (int)std::min<int64_t>(N, 256 * 40);
  hipLaunchKernelGGL(rg_scan_kernel<false>, dim3(grid), dim3(64), 0, s, (const float4*)sorted_pos4, skeys, cbegin,
                     cend, heads, nheads, g, deg, (const int32_t*)nullptr, (int32_t*)nullptr);
  E3_HIP_CHECK(hipMemsetAsync(deg + N, 0, 4, s));
  cb = w.cub_bytes;
  E3_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(ws + w.cub, cb, deg, rowptr, (int)(N + 1), s));
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

int e3_rg_fill(int64_t N, const e3_rg_params* p, const float* sorted_pos4, const int32_t* rowptr, int32_t* src,
               void* workspace, int64_t workspace_bytes, void* stream) {
  if (!p || N < 0 || p->bits < 1 || p->bits > 8) return E3_ERR_INVALID_ARG;
  if (N == 0) return E3_OK;
  if (!sorted_pos4 || !rowptr || !workspace) return E3_ERR_INVALID_ARG;
  RgWs w = rg_ws(N, p->bits);
  if ((int64_t)w.total > workspace_bytes) return E3_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  RgDev g = rg_dev(p);
  int32_t* deg = (int32_t*)(ws + w.deg);  // reused as the per-row fill cursor between chunks
  const int grid = (int)std::min<int64_t>(N, 256 * 40);
  hipLaunchKernelGGL(rg_scan_kernel<true>, dim3(grid), dim3(64), 0, s, (const float4*)sorted_pos4,
                     (const uint32_t*)(ws + w.skeys), (const int32_t*)(ws + w.cbegin), (const int32_t*)(ws + w.cend),
                     (const int32_t*)(ws + w.heads), (const int32_t*)(ws + w.nheads), g, deg, rowptr, src);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

}  // extern "C"
