// General SH tensor product for l <= 2 (builder-defined; contract in include/e3gnn.h).  This file:
// host plan, weight packing and the generic FMA kernel (fp32 / fp64).  It contracts the channel index
// with W first (T[m1] = sum_k W[k,w] x[k,m1]) and applies the coupling tensor and Y afterwards.
#include "e3_common.h"
#include "cg_tables.h"
#include "e3_tp_internal.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

namespace e3 {

// The dynamic-LDS limit of a kernel is a per-device attribute of the function: raise it monotonically under a lock (setting the
// size of each launch would let two host threads with different plans lower it under each other's launches).
static int tp_ensure_dyn_lds(const void* fn, size_t bytes) {
  if (bytes <= 64 * 1024) return E3_OK;
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> have;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return E3_ERR_HIP;
  std::lock_guard<std::mutex> lock(mu);
  size_t& cur = have[std::make_pair(fn, dev)];
  if (bytes > cur) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return E3_ERR_HIP;
    cur = bytes;
  }
  return E3_OK;
}

struct TpDev {
  int D1, Dout, Dy, lmax_sh;
  int n[6], M[6], cbase[6], obase[6], ocol_off[6], npath[6], poff[6], K[6];
  int64_t woff[6];
  int64_t normcol_off, packed_elems;
  const int32_t* cpos;
  const int32_t* ocol;
  const TpPath* paths;
};

}  // namespace e3

struct e3_tp_plan {
  e3::TpDev dev;
  std::vector<int32_t> h_tables;  // [cpos (D1) | ocol (sum M)]
  std::vector<e3::TpPath> h_paths;
  int32_t* d_tables = nullptr;
  e3::TpPath* d_paths = nullptr;
  e3::TpFast fast;
  std::mutex mu;
  int device = -1;  // the device the tables were uploaded to (the one current at first use); calls from another one fail
};

namespace e3 {

template <int L1, int L2, int L3, typename A>
__device__ __forceinline__ void tp_path_apply(const A* __restrict__ x, int n, const A* __restrict__ W, int M,
                                              const A* __restrict__ y, A (&o)[5]) {
  if constexpr (CG<L1, L2, L3>::valid) {
    constexpr int D1 = 2 * L1 + 1, D2 = 2 * L2 + 1, D3 = 2 * L3 + 1;
    A T[D1];
#pragma unroll
    for (int a = 0; a < D1; ++a) T[a] = 0;
    for (int k = 0; k < n; ++k) {
      const A wk = W[(int64_t)k * M];
#pragma unroll
      for (int a = 0; a < D1; ++a) T[a] += x[k * D1 + a] * wk;
    }
#pragma unroll
    for (int a = 0; a < D1; ++a)
#pragma unroll
      for (int b = 0; b < D2; ++b)
#pragma unroll
        for (int c = 0; c < D3; ++c)
          if (CG<L1, L2, L3>::v[a][b][c] != 0.0) o[c] += A(CG<L1, L2, L3>::v[a][b][c]) * T[a] * y[b];
  }
}

template <typename T, int R>
__global__ __launch_bounds__(256) void tp_fwd_generic_kernel(const T* __restrict__ in1, int64_t ld1,
                                                             const T* __restrict__ in2, int64_t ld2,
                                                             const typename AccOf<T>::type* __restrict__ packed,
                                                             T* __restrict__ out, int64_t ldo, int64_t B, TpDev p) {
  using A = typename AccOf<T>::type;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  A* xs = reinterpret_cast<A*>(smem_raw);
  A* ys = xs + (size_t)R * p.D1;
  const int tid = threadIdx.x;
  int Mtot = 0;
  for (int c = 0; c < 6; ++c) Mtot += p.M[c];
  const A* normcol = packed + p.normcol_off;
  const int64_t ntiles = (B + R - 1) / R;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * R;
    for (int i = tid; i < R * p.D1; i += 256) {
      int r = i / p.D1, d = i - r * p.D1;
      int64_t row = row0 + r;
      xs[r * p.D1 + p.cpos[d]] = row < B ? to_acc(in1[row * ld1 + d]) : A(0);
    }
    for (int i = tid; i < R * p.Dy; i += 256) {
      int r = i / p.Dy, d = i - r * p.Dy;
      int64_t row = row0 + r;
      ys[i] = row < B ? to_acc(in2[row * ld2 + d]) : A(0);
    }
    __syncthreads();
    for (int i = tid; i < R * Mtot; i += 256) {
      int r = i / Mtot, w = i - r * Mtot;
      int64_t row = row0 + r;
      if (row >= B) continue;
      int c3 = 0;
      while (w >= p.M[c3]) { w -= p.M[c3]; ++c3; }
      const int l3 = c3 >> 1, M = p.M[c3];
      const A* x = xs + r * p.D1;
      const A* y = ys + r * p.Dy;
      A o[5] = {0, 0, 0, 0, 0};
      for (int pi = 0; pi < p.npath[c3]; ++pi) {
        const TpPath P = p.paths[p.poff[c3] + pi];
        const A* xc = x + p.cbase[P.c1];
        const A* W = packed + p.woff[c3] + (int64_t)P.wrow * M + w;
        const A* yl = y + P.l2 * P.l2;  // offsets 0, 1, 4
        const int n = p.n[P.c1];
        switch (P.l1 * 9 + P.l2 * 3 + l3) {
#define E3_CASE(a, b, c) case a * 9 + b * 3 + c: tp_path_apply<a, b, c, A>(xc, n, W, M, yl, o); break;
          E3_CASE(0, 0, 0) E3_CASE(0, 1, 1) E3_CASE(0, 2, 2) E3_CASE(1, 0, 1) E3_CASE(1, 1, 0) E3_CASE(1, 1, 1)
          E3_CASE(1, 1, 2) E3_CASE(1, 2, 1) E3_CASE(1, 2, 2) E3_CASE(2, 0, 2) E3_CASE(2, 1, 1) E3_CASE(2, 1, 2)
          E3_CASE(2, 2, 0) E3_CASE(2, 2, 1) E3_CASE(2, 2, 2)
#undef E3_CASE
          default: break;
        }
      }
      const int oc = p.ocol[p.ocol_off[c3] + w];
      T* dst = out + row * ldo + oc;
      for (int m = 0; m < 2 * l3 + 1; ++m) dst[m] = from_acc<T, A>(o[m] * normcol[oc + m]);
    }
    __syncthreads();
  }
}

template <typename T>
__global__ void tp_pack_kernel(const T* w0, const T* w1, const T* w2, const T* w3, const T* w4, const T* w5,
                               const T* n0, const T* n1, const T* n2, const T* n3, const T* n4, const T* n5,
                               typename AccOf<T>::type* packed, TpDev p) {
  using A = typename AccOf<T>::type;
  const T* w[6] = {w0, w1, w2, w3, w4, w5};
  const T* nr[6] = {n0, n1, n2, n3, n4, n5};
  int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  for (int c = 0; c < 6; ++c) {
    int64_t nw = (int64_t)p.K[c] * p.M[c];
    for (int64_t i = tid; i < nw; i += stride) packed[p.woff[c] + i] = w[c] ? to_acc(w[c][i]) : A(0);
    int width = 2 * (c >> 1) + 1;
    for (int64_t i = tid; i < (int64_t)p.M[c] * width; i += stride) {
      int m = (int)(i / width), comp = (int)(i - (int64_t)m * width);
      packed[p.normcol_off + p.ocol[p.ocol_off[c] + m] + comp] = nr[c] ? to_acc(nr[c][i]) : A(1);
    }
  }
}

static int tp_ensure_device(const e3_tp_plan* cplan) {
  auto* P = const_cast<e3_tp_plan*>(cplan);
  std::lock_guard<std::mutex> lock(P->mu);
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) return E3_ERR_NO_DEVICE;
  if (P->d_tables) return cur == P->device ? E3_OK : E3_ERR_INVALID_ARG;  // one plan = one device (create one per device)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return E3_ERR_NO_DEVICE;
  int32_t* d = nullptr;
  TpPath* dpaths = nullptr;
  auto fail = [&](int st) {
    if (d) (void)hipFree(d);
    if (dpaths) (void)hipFree(dpaths);
    fast_free(&P->fast);
    return st;
  };
  if (hipMalloc((void**)&d, P->h_tables.size() * sizeof(int32_t)) != hipSuccess) return fail(E3_ERR_HIP);
  if (hipMemcpy(d, P->h_tables.data(), P->h_tables.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
    return fail(E3_ERR_HIP);
  const size_t np = std::max<size_t>(P->h_paths.size(), 1);
  if (hipMalloc((void**)&dpaths, np * sizeof(TpPath)) != hipSuccess) return fail(E3_ERR_HIP);
  if (!P->h_paths.empty() &&
      hipMemcpy(dpaths, P->h_paths.data(), P->h_paths.size() * sizeof(TpPath), hipMemcpyHostToDevice) != hipSuccess)
    return fail(E3_ERR_HIP);
  const int st = fast_upload(&P->fast);
  if (st != E3_OK) return fail(st);
  P->d_paths = dpaths;
  P->dev.cpos = d;
  P->dev.ocol = d + P->dev.D1;
  P->dev.paths = dpaths;
  P->device = cur;
  P->d_tables = d;
  return E3_OK;
}

template <typename T>
static int tp_launch_fwd(const e3_tp_plan* plan, const void* in1, int64_t ld1, const void* in2, int64_t ld2,
                         const void* packed, void* out, int64_t ldo, int64_t B, hipStream_t s) {
  using A = typename AccOf<T>::type;
  constexpr int R = 16;
  size_t smem = (size_t)R * (plan->dev.D1 + plan->dev.Dy) * sizeof(A);
  if (smem > 160 * 1024) return E3_ERR_UNSUPPORTED;
  auto kern = tp_fwd_generic_kernel<T, R>;
  if (smem > 64 * 1024)
    { int st_ = tp_ensure_dyn_lds((const void*)kern, smem); if (st_ != E3_OK) return st_; }
  int grid = (int)std::min<int64_t>((B + R - 1) / R, 256 * 8);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, (const T*)in1, ld1, (const T*)in2, ld2, (const A*)packed,
                     (T*)out, ldo, B, plan->dev);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}


// ---------------------------------------------------------------------------------------------------
// backward of the general tensor product (generic FMA kernels, fp32 / fp64; the reference's operator gets its
// gradients from torch autograd over l1_tensor_prod.py:240-299 -- these kernels are that for l <= 2)
//   out[b, w, c] = norm[w,c] * sum_paths sum_k W[k, w] * sum_{a,bq} C[a][bq][c] x[b,k,a] y[b,bq]
// ---------------------------------------------------------------------------------------------------
template <int L1, int L2, int L3, typename A>
struct TpGrad {
  static constexpr bool ok = CG<L1, L2, L3>::valid;
  static constexpr int D1 = 2 * L1 + 1, D2 = 2 * L2 + 1, D3 = 2 * L3 + 1;
  // gx[a] += wv * sum_{bq,c} C y[bq] g[c]
  static __device__ __forceinline__ void gx(A wv, const A* y, const A* g, A* gxa) {
    if constexpr (ok) {
#pragma unroll
      for (int a = 0; a < D1; ++a) {
        A s = 0;
#pragma unroll
        for (int bq = 0; bq < D2; ++bq)
#pragma unroll
          for (int c = 0; c < D3; ++c)
            if (CG<L1, L2, L3>::v[a][bq][c] != 0.0) s += A(CG<L1, L2, L3>::v[a][bq][c]) * y[bq] * g[c];
        gxa[a] += wv * s;
      }
    }
  }
  // gy[bq] += wv * sum_{a,c} C x[a] g[c]
  static __device__ __forceinline__ void gy(A wv, const A* x, const A* g, A* gyb) {
    if constexpr (ok) {
#pragma unroll
      for (int bq = 0; bq < D2; ++bq) {
        A s = 0;
#pragma unroll
        for (int a = 0; a < D1; ++a)
#pragma unroll
          for (int c = 0; c < D3; ++c)
            if (CG<L1, L2, L3>::v[a][bq][c] != 0.0) s += A(CG<L1, L2, L3>::v[a][bq][c]) * x[a] * g[c];
        gyb[bq] += wv * s;
      }
    }
  }
  // f[c] = sum_{a,bq} C x[a] y[bq]   (the feature the weight row of channel k multiplies)
  static __device__ __forceinline__ void feat(const A* x, const A* y, A* f) {
    if constexpr (ok) {
#pragma unroll
      for (int c = 0; c < D3; ++c) {
        A s = 0;
#pragma unroll
        for (int a = 0; a < D1; ++a)
#pragma unroll
          for (int bq = 0; bq < D2; ++bq)
            if (CG<L1, L2, L3>::v[a][bq][c] != 0.0) s += A(CG<L1, L2, L3>::v[a][bq][c]) * x[a] * y[bq];
        f[c] = s;
      }
    }
  }
  // from t[c] = sum_w W[k,w] g'[w,c]:  gx[a] += sum_{bq,c} C y[bq] t[c],  gy[bq] += sum_{a,c} C x[a] t[c]
  static __device__ __forceinline__ void gxy(const A* x, const A* y, const A* t, A* gxa, A* gyb) {
    if constexpr (ok) {
#pragma unroll
      for (int a = 0; a < D1; ++a)
#pragma unroll
        for (int bq = 0; bq < D2; ++bq) {
          A s = 0;
#pragma unroll
          for (int c = 0; c < D3; ++c)
            if (CG<L1, L2, L3>::v[a][bq][c] != 0.0) s += A(CG<L1, L2, L3>::v[a][bq][c]) * t[c];
          gxa[a] += s * y[bq];
          gyb[bq] += s * x[a];
        }
    }
  }
  // sum_{a,bq,c} C x[a] y[bq] g[c]
  static __device__ __forceinline__ A gw(const A* x, const A* y, const A* g) {
    A s = 0;
    if constexpr (ok) {
#pragma unroll
      for (int a = 0; a < D1; ++a)
#pragma unroll
        for (int bq = 0; bq < D2; ++bq)
#pragma unroll
          for (int c = 0; c < D3; ++c)
            if (CG<L1, L2, L3>::v[a][bq][c] != 0.0) s += A(CG<L1, L2, L3>::v[a][bq][c]) * x[a] * y[bq] * g[c];
    }
    return s;
  }
};
#define E3_GRAD_SWITCH(l1, l2, l3, CALL)                                                                     \
  switch ((l1) * 9 + (l2) * 3 + (l3)) {                                                                      \
    case 0: { using G = TpGrad<0, 0, 0, A>; CALL; } break;   case 4: { using G = TpGrad<0, 1, 1, A>; CALL; } break;  \
    case 8: { using G = TpGrad<0, 2, 2, A>; CALL; } break;   case 10: { using G = TpGrad<1, 0, 1, A>; CALL; } break; \
    case 12: { using G = TpGrad<1, 1, 0, A>; CALL; } break;  case 13: { using G = TpGrad<1, 1, 1, A>; CALL; } break; \
    case 14: { using G = TpGrad<1, 1, 2, A>; CALL; } break;  case 16: { using G = TpGrad<1, 2, 1, A>; CALL; } break; \
    case 17: { using G = TpGrad<1, 2, 2, A>; CALL; } break;  case 20: { using G = TpGrad<2, 0, 2, A>; CALL; } break; \
    case 22: { using G = TpGrad<2, 1, 1, A>; CALL; } break;  case 23: { using G = TpGrad<2, 1, 2, A>; CALL; } break; \
    case 24: { using G = TpGrad<2, 2, 0, A>; CALL; } break;  case 25: { using G = TpGrad<2, 2, 1, A>; CALL; } break; \
    case 26: { using G = TpGrad<2, 2, 2, A>; CALL; } break;  default: break;                                    \
  }

// stage R rows: x in class order, y, g' = grad_out * norm (original out column order)
template <typename T, typename A>
__device__ __forceinline__ void tp_bwd_stage(const T* in1, int64_t ld1, const T* in2, int64_t ld2, const T* go, int64_t ldg,
                                             const A* normcol, int64_t row0, int R, int64_t B, const TpDev& p, A* xs,
                                             A* ys, A* gs) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < R * p.D1; i += nt) {
    int r = i / p.D1, d = i - r * p.D1;
    int64_t row = row0 + r;
    xs[r * p.D1 + p.cpos[d]] = row < B ? to_acc(in1[row * ld1 + d]) : A(0);
  }
  for (int i = tid; i < R * p.Dy; i += nt) {
    int r = i / p.Dy, d = i - r * p.Dy;
    int64_t row = row0 + r;
    ys[i] = row < B ? to_acc(in2[row * ld2 + d]) : A(0);
  }
  for (int i = tid; i < R * p.Dout; i += nt) {
    int r = i / p.Dout, d = i - r * p.Dout;
    int64_t row = row0 + r;
    gs[i] = row < B ? to_acc(go[row * ldg + d]) * normcol[d] : A(0);
  }
}

template <typename T, int R>
__global__ __launch_bounds__(256) void tp_bwd_rows_kernel(const T* __restrict__ in1, int64_t ld1, const T* __restrict__ in2,
                                                          int64_t ld2, const typename AccOf<T>::type* __restrict__ packed,
                                                          const T* __restrict__ go, int64_t ldg, T* __restrict__ gin1,
                                                          int64_t ldg1, typename AccOf<T>::type* __restrict__ gin2,
                                                          int64_t ldg2, int64_t B, TpDev p) {
  using A = typename AccOf<T>::type;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  A* xs = reinterpret_cast<A*>(smem_raw);
  A* ys = xs + (size_t)R * p.D1;
  A* gs = ys + (size_t)R * p.Dy;
  A* gx = gs + (size_t)R * p.Dout;
  A* gy = gx + (size_t)R * p.D1;
  const int tid = threadIdx.x;
  int Ktot = 0, Mtot = 0;
  for (int c = 0; c < 6; ++c) { Ktot += p.n[c]; Mtot += p.M[c]; }
  const A* normcol = packed + p.normcol_off;
  const int64_t ntiles = (B + R - 1) / R;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * R;
    tp_bwd_stage<T, A>(in1, ld1, in2, ld2, go, ldg, normcol, row0, R, B, p, xs, ys, gs);
    for (int i = tid; i < R * 9; i += 256) gy[i] = 0;
    __syncthreads();
    if (gin1) {  // thread = (row, in channel): sum over every path that reads this channel and every out channel
      for (int i = tid; i < R * Ktot; i += 256) {
        int r = i / Ktot, kk = i - r * Ktot, c1 = 0;
        while (kk >= p.n[c1]) { kk -= p.n[c1]; ++c1; }
        const int l1 = c1 >> 1, D1c = 2 * l1 + 1;
        A acc[5] = {0, 0, 0, 0, 0};
        const A* y = ys + r * p.Dy;
        for (int c3 = 0; c3 < 6; ++c3) {
          const int l3 = c3 >> 1, M = p.M[c3];
          for (int pi = 0; pi < p.npath[c3]; ++pi) {
            const TpPath P = p.paths[p.poff[c3] + pi];
            if (P.c1 != c1) continue;
            const A* W = packed + p.woff[c3] + (int64_t)(P.wrow + kk) * M;
            const A* yl = y + P.l2 * P.l2;
            for (int w = 0; w < M; ++w) {
              const A* g = gs + r * p.Dout + p.ocol[p.ocol_off[c3] + w];
              const A wv = W[w];
              E3_GRAD_SWITCH(l1, P.l2, l3, G::gx(wv, yl, g, acc))
            }
          }
        }
        for (int a = 0; a < D1c; ++a) gx[r * p.D1 + p.cbase[c1] + kk * D1c + a] = acc[a];
      }
    }
    if (gin2) {  // thread = (row, out channel): partial sums over the in channels, reduced through LDS atomics
      for (int i = tid; i < R * Mtot; i += 256) {
        int r = i / Mtot, w = i - r * Mtot, c3 = 0;
        while (w >= p.M[c3]) { w -= p.M[c3]; ++c3; }
        const int l3 = c3 >> 1, M = p.M[c3];
        const A* g = gs + r * p.Dout + p.ocol[p.ocol_off[c3] + w];
        A loc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int pi = 0; pi < p.npath[c3]; ++pi) {
          const TpPath P = p.paths[p.poff[c3] + pi];
          const int D1c = 2 * P.l1 + 1;
          const A* xc = xs + r * p.D1 + p.cbase[P.c1];
          const A* W = packed + p.woff[c3] + (int64_t)P.wrow * M + w;
          for (int k = 0; k < p.n[P.c1]; ++k) {
            const A wv = W[(int64_t)k * M];
            E3_GRAD_SWITCH(P.l1, P.l2, l3, G::gy(wv, xc + k * D1c, g, loc + P.l2 * P.l2))
          }
        }
        for (int q = 0; q < p.Dy; ++q) atomicAdd(&gy[r * 9 + q], loc[q]);
      }
    }
    __syncthreads();
    if (gin1)
      for (int i = tid; i < R * p.D1; i += 256) {
        int r = i / p.D1, d = i - r * p.D1;
        if (row0 + r < B) gin1[(row0 + r) * ldg1 + d] = from_acc<T, A>(gx[r * p.D1 + p.cpos[d]]);
      }
    if (gin2)
      for (int i = tid; i < R * p.Dy; i += 256) {
        int r = i / p.Dy, q = i - r * p.Dy;
        if (row0 + r < B) {
          if (ldg2 == 0) atomicAdd(&gin2[q], gy[r * 9 + q]);   // broadcast in2: one row of sums (caller zero-fills)
          else gin2[(row0 + r) * ldg2 + q] = gy[r * 9 + q];
        }
      }
    __syncthreads();
  }
}

// grad_W[c3][krow][w] += sum_b sum C x y g'   (atomics into zero-filled fp32 / fp64 arrays, one per out class)
template <typename T, int R>
__global__ __launch_bounds__(256) void tp_bwd_w_kernel(const T* __restrict__ in1, int64_t ld1, const T* __restrict__ in2,
                                                       int64_t ld2, const typename AccOf<T>::type* __restrict__ packed,
                                                       const T* __restrict__ go, int64_t ldg,
                                                       typename AccOf<T>::type* g0, typename AccOf<T>::type* g1,
                                                       typename AccOf<T>::type* g2, typename AccOf<T>::type* g3,
                                                       typename AccOf<T>::type* g4, typename AccOf<T>::type* g5,
                                                       int64_t B, TpDev p) {
  using A = typename AccOf<T>::type;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  A* xs = reinterpret_cast<A*>(smem_raw);
  A* ys = xs + (size_t)R * p.D1;
  A* gs = ys + (size_t)R * p.Dy;
  A* gw[6] = {g0, g1, g2, g3, g4, g5};
  const int tid = threadIdx.x;
  const A* normcol = packed + p.normcol_off;
  int64_t Wtot = 0;
  for (int c = 0; c < 6; ++c) Wtot += (int64_t)p.K[c] * p.M[c];
  const int64_t ntiles = (B + R - 1) / R;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * R;
    __syncthreads();
    tp_bwd_stage<T, A>(in1, ld1, in2, ld2, go, ldg, normcol, row0, R, B, p, xs, ys, gs);
    __syncthreads();
    for (int64_t i = tid; i < Wtot; i += 256) {
      int64_t e = i;
      int c3 = 0;
      while (e >= (int64_t)p.K[c3] * p.M[c3]) { e -= (int64_t)p.K[c3] * p.M[c3]; ++c3; }
      if (!gw[c3]) continue;
      const int M = p.M[c3], krow = (int)(e / M), w = (int)(e - (int64_t)krow * M), l3 = c3 >> 1;
      TpPath P = p.paths[p.poff[c3]];
      for (int pi = 0; pi < p.npath[c3]; ++pi) {
        const TpPath Q = p.paths[p.poff[c3] + pi];
        if (krow >= Q.wrow) P = Q;
      }
      const int kk = krow - P.wrow, D1c = 2 * P.l1 + 1;
      const int oc = p.ocol[p.ocol_off[c3] + w];
      A acc = 0;
      for (int r = 0; r < R; ++r) {
        const A* x = xs + r * p.D1 + p.cbase[P.c1] + kk * D1c;
        const A* y = ys + r * p.Dy + P.l2 * P.l2;
        const A* g = gs + r * p.Dout + oc;
        E3_GRAD_SWITCH(P.l1, P.l2, l3, acc += G::gw(x, y, g))
      }
      atomicAdd(&gw[c3][e], acc);
    }
  }
}

template <typename T>
static int tp_launch_bwd(const e3_tp_plan* plan, const void* in1, int64_t ld1, const void* in2, int64_t ld2,
                         const void* packed, const void* go, int64_t ldg, void* gin1, int64_t ldg1, void* gin2,
                         int64_t ldg2, void* const gw[6], int64_t B, hipStream_t s) {
  using A = typename AccOf<T>::type;
  const TpDev& p = plan->dev;
  constexpr int R = 4;
  if (gin1 || gin2) {
    const size_t lds = (size_t)R * (2 * p.D1 + p.Dy + p.Dout + 9) * sizeof(A);
    if (lds > 160 * 1024) return E3_ERR_UNSUPPORTED;
    auto k = tp_bwd_rows_kernel<T, R>;
    { int st_ = tp_ensure_dyn_lds((const void*)k, lds); if (st_ != E3_OK) return st_; }
    const int grid = (int)std::min<int64_t>((B + R - 1) / R, 256 * 4);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, (const T*)in1, ld1, (const T*)in2, ld2, (const A*)packed,
                       (const T*)go, ldg, (T*)gin1, ldg1, (A*)gin2, ldg2, B, p);
    E3_HIP_CHECK(hipGetLastError());
  }
  bool anyw = false;
  for (int c = 0; c < 6; ++c) anyw |= gw[c] != nullptr;
  if (anyw) {
    constexpr int RW = 8;
    const size_t lds = (size_t)RW * (p.D1 + p.Dy + p.Dout) * sizeof(A);
    if (lds > 160 * 1024) return E3_ERR_UNSUPPORTED;
    auto k = tp_bwd_w_kernel<T, RW>;
    { int st_ = tp_ensure_dyn_lds((const void*)k, lds); if (st_ != E3_OK) return st_; }
    const int grid = (int)std::min<int64_t>((B + RW - 1) / RW, 256 * 2);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, (const T*)in1, ld1, (const T*)in2, ld2, (const A*)packed,
                       (const T*)go, ldg, (A*)gw[0], (A*)gw[1], (A*)gw[2], (A*)gw[3], (A*)gw[4], (A*)gw[5], B, p);
    E3_HIP_CHECK(hipGetLastError());
  }
  return E3_OK;
}


// ---------------------------------------------------------------------------------------------------
// Backward as two thin passes around library GEMMs (the large-B path; the kernels above stay as the small-B / reference
// path).  Per output class c3 (D3 = 2 l3 + 1, K = weight rows, M = weight cols):
//   operands:  F[b, c, krow] = sum_{a,bq} C[a][bq][c] x[b,k,a] y[b,bq]      G[b, c, w] = grad_out[b, (w,c)] * norm[w,c]
//   caller:    grad_W = F^T G   (one GEMM over the B * D3 rows)             T = G W^T  (same shape as F)
//   contract:  grad_in1[b,k,a] = sum_paths sum_{bq,c} C y[b,bq] T[b,c,wrow+k]   grad_in2[b,bq] = sum_paths sum_{k,a,c} C x T
// One wave owns one row; lanes run over the channels k of a path, so F / T accesses are contiguous runs.
// ---------------------------------------------------------------------------------------------------
// one path, RW rows of one wave: features of channel k = lane, lane + 64, ... (everything indexed at compile time)
template <typename G, typename A, int RW>
__device__ __forceinline__ void tp_bwd_feat_path(const A* xb, int ldx, const A* yb, int ldy, int n, int lane, int nrow,
                                                 A* F, int K) {
  if constexpr (G::ok) {
    for (int k = lane; k < n; k += 64) {
#pragma unroll
      for (int q = 0; q < RW; ++q) {
        if (q < nrow) {
          A f[G::D3];
          G::feat(xb + q * ldx + k * G::D1, yb + q * ldy + G::D2 / 2 * (G::D2 / 2), f);
          A* Fr = F + (int64_t)q * G::D3 * K + k;
#pragma unroll
          for (int c = 0; c < G::D3; ++c) Fr[(int64_t)c * K] = f[c];
        }
      }
    }
  }
}

template <typename G, typename A, int RW>
__device__ __forceinline__ void tp_bwd_contract_path(const A* xb, A* gxb, int ldx, const A* yb, int ldy, int n, int lane,
                                                     const A* ts, int K, A (&gy)[RW][9]) {
  if constexpr (G::ok) {
    constexpr int L2 = G::D2 / 2;
    for (int k = lane; k < n; k += 64) {
#pragma unroll
      for (int q = 0; q < RW; ++q) {
        A t[G::D3], ga[G::D1], gq[G::D2];
#pragma unroll
        for (int c = 0; c < G::D3; ++c) t[c] = ts[(q * G::D3 + c) * K + k];
#pragma unroll
        for (int a = 0; a < G::D1; ++a) ga[a] = 0;
#pragma unroll
        for (int j = 0; j < G::D2; ++j) gq[j] = 0;
        G::gxy(xb + q * ldx + k * G::D1, yb + q * ldy + L2 * L2, t, ga, gq);
        A* gc = gxb + q * ldx + k * G::D1;
#pragma unroll
        for (int a = 0; a < G::D1; ++a) gc[a] += ga[a];
#pragma unroll
        for (int j = 0; j < G::D2; ++j) gy[q][L2 * L2 + j] += gq[j];
      }
    }
  }
}

struct TpPtr6 { void* p[6]; };

// stage the rows of one block: x in class order, y; rows past B read as zero
template <typename T, typename A>
__device__ __forceinline__ void tp_bwd_stage_xy(const T* in1, int64_t ld1, const T* in2, int64_t ld2, int64_t row0, int R,
                                                int64_t B, const TpDev& p, A* xs, A* ys) {
  const int tid = threadIdx.x;
  for (int i = tid; i < R * p.D1; i += 256) {
    int r = i / p.D1, d = i - r * p.D1;
    int64_t row = row0 + r;
    xs[r * p.D1 + p.cpos[d]] = row < B ? to_acc(in1[row * ld1 + d]) : A(0);
  }
  for (int i = tid; i < R * p.Dy; i += 256) {
    int r = i / p.Dy, d = i - r * p.Dy;
    int64_t row = row0 + r;
    ys[i] = row < B ? to_acc(in2[row * ld2 + d]) : A(0);
  }
}

// RW rows per wave (4 waves, R = 4 RW rows per block): the RW rows of a wave go through every path together, so their
// loads / stores are in flight at the same time
template <typename T, int RW>
__global__ __launch_bounds__(256) void tp_bwd_operands_kernel(const T* __restrict__ in1, int64_t ld1,
                                                              const T* __restrict__ in2, int64_t ld2,
                                                              const typename AccOf<T>::type* __restrict__ packed,
                                                              const T* __restrict__ go, int64_t ldg, TpPtr6 Fp, TpPtr6 Gp,
                                                              int64_t B, TpDev p) {
  using A = typename AccOf<T>::type;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  constexpr int R = 4 * RW;
  A* xs = reinterpret_cast<A*>(smem_raw);
  A* ys = xs + (size_t)R * p.D1;
  A* gs = ys + (size_t)R * p.Dy;  // grad_out * norm, original column order (staged with x and y: one round of loads per tile)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const A* normcol = packed + p.normcol_off;
  const int64_t ntiles = (B + R - 1) / R;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {  // uniform trip count per block: barriers are safe
    const int64_t row0 = tile * R;
    __syncthreads();
    if (go) tp_bwd_stage<T, A>(in1, ld1, in2, ld2, go, ldg, normcol, row0, R, B, p, xs, ys, gs);
    else tp_bwd_stage_xy<T, A>(in1, ld1, in2, ld2, row0, R, B, p, xs, ys);
    __syncthreads();
    const int64_t wrow0 = row0 + wave * RW;
    for (int c3 = 0; c3 < 6; ++c3) {
      const int l3 = c3 >> 1, D3 = 2 * l3 + 1, M = p.M[c3], K = p.K[c3];
      if (M == 0 || K == 0) continue;
      A* F = static_cast<A*>(Fp.p[c3]);
      A* Gm = static_cast<A*>(Gp.p[c3]);
      if (F) {
        for (int pi = 0; pi < p.npath[c3]; ++pi) {
          const TpPath P = p.paths[p.poff[c3] + pi];
          const int n = p.n[P.c1];
          const A* xb = xs + wave * RW * p.D1 + p.cbase[P.c1];
          const A* yb = ys + wave * RW * p.Dy;
          const int nrow = B - wrow0 < RW ? (int)(B - wrow0) : RW;  // <= 0 for a wave past the end
          E3_GRAD_SWITCH(P.l1, P.l2, l3, (tp_bwd_feat_path<G, A, RW>(xb, p.D1, yb, p.Dy, n, lane, nrow,
                                                                     F + wrow0 * G::D3 * (int64_t)K + P.wrow, K)))
        }
      }
      if (Gm) {
        for (int i = lane; i < D3 * M; i += 64) {
          const int c = i / M, w = i - c * M;
          const int oc = p.ocol[p.ocol_off[c3] + w] + c;
#pragma unroll
          for (int q = 0; q < RW; ++q)
            if (wrow0 + q < B) Gm[(wrow0 + q) * D3 * (int64_t)M + i] = gs[(wave * RW + q) * p.Dout + oc];
        }
      }
    }
  }
}

template <typename T, int RW>
__global__ __launch_bounds__(256) void tp_bwd_contract_kernel(const T* __restrict__ in1, int64_t ld1,
                                                              const T* __restrict__ in2, int64_t ld2, TpPtr6 Tp,
                                                              T* __restrict__ gin1, int64_t ldg1,
                                                              typename AccOf<T>::type* __restrict__ gin2, int64_t ldg2,
                                                              int64_t B, TpDev p) {
  using A = typename AccOf<T>::type;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  constexpr int R = 4 * RW;
  A* xs = reinterpret_cast<A*>(smem_raw);
  A* ys = xs + (size_t)R * p.D1;
  A* gx = ys + (size_t)R * p.Dy;
  A* ts = gx + (size_t)R * p.D1;  // the R rows of T, class after class: [c3][row][c][K] (a flat copy of each class's rows)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int64_t ntiles = (B + R - 1) / R;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * R;
    const int nr = B - row0 < R ? (int)(B - row0) : R;
    __syncthreads();
    {  // every load of the tile is issued here, back to back; the path loop below only touches LDS
      A* td = ts;
      for (int c3 = 0; c3 < 6; ++c3) {
        const A* Tm = static_cast<const A*>(Tp.p[c3]);
        const int per = (2 * (c3 >> 1) + 1) * p.K[c3];
        if (!Tm || per == 0 || p.M[c3] == 0) continue;
        const A* src = Tm + row0 * per;
        for (int i = tid; i < nr * per; i += 256) td[i] = src[i];
        td += R * per;
      }
    }
    tp_bwd_stage_xy<T, A>(in1, ld1, in2, ld2, row0, R, B, p, xs, ys);
    for (int i = tid; i < R * p.D1; i += 256) gx[i] = 0;
    __syncthreads();
    const int64_t wrow0 = row0 + wave * RW;
    A gy[RW][9];
#pragma unroll
    for (int q = 0; q < RW; ++q)
#pragma unroll
      for (int j = 0; j < 9; ++j) gy[q][j] = 0;
    const A* tc = ts;
    for (int c3 = 0; c3 < 6; ++c3) {
      const int l3 = c3 >> 1, D3 = 2 * l3 + 1, K = p.K[c3];
      if (!Tp.p[c3] || K == 0 || p.M[c3] == 0) continue;
      for (int pi = 0; pi < p.npath[c3]; ++pi) {
        const TpPath P = p.paths[p.poff[c3] + pi];
        // lane <-> channel is the same for every path of a class: no race on gx.  Rows past B hold stale LDS: their
        // results stay in LDS / registers and are dropped
        E3_GRAD_SWITCH(P.l1, P.l2, l3, (tp_bwd_contract_path<G, A, RW>(
                                           xs + wave * RW * p.D1 + p.cbase[P.c1], gx + wave * RW * p.D1 + p.cbase[P.c1],
                                           p.D1, ys + wave * RW * p.Dy, p.Dy, p.n[P.c1], lane,
                                           tc + wave * RW * G::D3 * K + P.wrow, K, gy)))
      }
      tc += R * D3 * K;
    }
    if (gin2) {
#pragma unroll
      for (int q = 0; q < RW; ++q) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
          A v = gy[q][j];
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
          gy[q][j] = v;
        }
        if (wrow0 + q < B && lane < p.Dy) {
          A v = gy[q][0];
#pragma unroll
          for (int j = 1; j < 9; ++j) v = lane == j ? gy[q][j] : v;
          if (ldg2 == 0) atomicAdd(&gin2[lane], v);
          else gin2[(wrow0 + q) * ldg2 + lane] = v;
        }
      }
    }
    __syncthreads();
    if (gin1)
      for (int i = tid; i < nr * p.D1; i += 256) {
        int r = i / p.D1, d = i - r * p.D1;
        gin1[(row0 + r) * ldg1 + d] = from_acc<T, A>(gx[r * p.D1 + p.cpos[d]]);
      }
  }
}

template <typename T, int RW>
static int tp_launch_bwd_operands_rw(const e3_tp_plan* plan, const void* in1, int64_t ld1, const void* in2, int64_t ld2,
                                     const void* packed, const void* go, int64_t ldg, const TpPtr6& f, const TpPtr6& g,
                                     int64_t B, hipStream_t s) {
  using A = typename AccOf<T>::type;
  const TpDev& p = plan->dev;
  constexpr int R = 4 * RW;
  const size_t lds = (size_t)R * (p.D1 + p.Dy + p.Dout) * sizeof(A);
  if (lds > 160 * 1024) return E3_ERR_UNSUPPORTED;
  auto k = tp_bwd_operands_kernel<T, RW>;
  if (lds > 64 * 1024) { int st_ = tp_ensure_dyn_lds((const void*)k, lds); if (st_ != E3_OK) return st_; }
  const int grid = (int)std::min<int64_t>((B + R - 1) / R, 256 * 8);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, (const T*)in1, ld1, (const T*)in2, ld2, (const A*)packed,
                     (const T*)go, ldg, f, g, B, p);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

template <typename T>
static int tp_launch_bwd_operands(const e3_tp_plan* plan, const void* in1, int64_t ld1, const void* in2, int64_t ld2,
                                  const void* packed, const void* go, int64_t ldg, void* const F[6], void* const G[6],
                                  int64_t B, hipStream_t s) {
  TpPtr6 f, g;
  for (int c = 0; c < 6; ++c) { f.p[c] = F ? F[c] : nullptr; g.p[c] = G ? G[c] : nullptr; }
  return tp_launch_bwd_operands_rw<T, 2>(plan, in1, ld1, in2, ld2, packed, go, ldg, f, g, B, s);
}

template <typename A>
static size_t tp_bwd_contract_lds(const TpDev& p, int R) {
  size_t trow = 0;
  for (int c = 0; c < 6; ++c)
    if (p.M[c] > 0) trow += (size_t)(2 * (c >> 1) + 1) * p.K[c];
  return (size_t)R * (2 * p.D1 + p.Dy + trow) * sizeof(A);
}

template <typename T, int RW>
static int tp_launch_bwd_contract_rw(const e3_tp_plan* plan, const void* in1, int64_t ld1, const void* in2, int64_t ld2,
                                     const TpPtr6& t, void* gin1, int64_t ldg1, void* gin2, int64_t ldg2, int64_t B,
                                     hipStream_t s) {
  using A = typename AccOf<T>::type;
  const TpDev& p = plan->dev;
  constexpr int R = 4 * RW;
  const size_t lds = tp_bwd_contract_lds<A>(p, R);
  if (lds > 160 * 1024) return E3_ERR_UNSUPPORTED;
  auto k = tp_bwd_contract_kernel<T, RW>;
  if (lds > 64 * 1024) { int st_ = tp_ensure_dyn_lds((const void*)k, lds); if (st_ != E3_OK) return st_; }
  const int grid = (int)std::min<int64_t>((B + R - 1) / R, 256 * 8);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, s, (const T*)in1, ld1, (const T*)in2, ld2, t, (T*)gin1, ldg1,
                     (A*)gin2, ldg2, B, p);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

template <typename T>
static int tp_launch_bwd_contract(const e3_tp_plan* plan, const void* in1, int64_t ld1, const void* in2, int64_t ld2,
                                  void* const Tm[6], void* gin1, int64_t ldg1, void* gin2, int64_t ldg2, int64_t B,
                                  hipStream_t s) {
  using A = typename AccOf<T>::type;
  TpPtr6 t;
  for (int c = 0; c < 6; ++c) t.p[c] = (plan->dev.M[c] > 0 && plan->dev.K[c] > 0) ? Tm[c] : nullptr;
  for (int c = 0; c < 6; ++c)
    if (plan->dev.M[c] > 0 && plan->dev.K[c] > 0 && !Tm[c]) return E3_ERR_INVALID_ARG;  // every class with weights has a T
  // rows per block by LDS budget: 4 blocks per CU (40 KB each) if possible
  if (tp_bwd_contract_lds<A>(plan->dev, 8) <= 40 * 1024)
    return tp_launch_bwd_contract_rw<T, 2>(plan, in1, ld1, in2, ld2, t, gin1, ldg1, gin2, ldg2, B, s);
  return tp_launch_bwd_contract_rw<T, 1>(plan, in1, ld1, in2, ld2, t, gin1, ldg1, gin2, ldg2, B, s);
}


// ---------------------------------------------------------------------------------------------------
// Fused weight gradient of one output class (fp32): grad_W[c3][krow][w] += sum_{b,c} F[b,c,krow] G[b,c,w] with the features F and
// the normalised output gradient G of a row tile built in LDS and contracted on v_mfma_f32_32x32x2_f32 (exact fp32 products) --
// nothing of size [B, D3, K] reaches HBM.  One workgroup = 4 waves; output tiles of 32 x 32 are dealt to the waves round-robin,
// their accumulators live in registers over the whole row range of the workgroup and leave through atomics at the end.
// LDS: x (class order) and y of R rows, Fs [R D3][ldF], Gs [R D3][ldG]; ldF, ldG = 32 (mod 64) floats, so that the two k rows of
// one MFMA operand read (lanes 0-31 / 32-63) fall on different banks.
// ---------------------------------------------------------------------------------------------------
template <typename G>
__device__ __forceinline__ void tp_wgrad_feat_item(const float* x, const float* y, float* F, int ldF) {
  if constexpr (G::ok) {
    float f[G::D3];
    G::feat(x, y, f);
#pragma unroll
    for (int c = 0; c < G::D3; ++c) F[c * ldF] = f[c];
  }
}

typedef float tp_f32x16 __attribute__((ext_vector_type(16)));

// workgroup barrier that orders LDS accesses only: __syncthreads() also waits for every global load in flight (vmcnt(0)), which
// would end the register prefetch of the next row tile at the first barrier after it was issued
__device__ __forceinline__ void tp_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NTW, int RT>
__global__ __launch_bounds__(256) void tp_wgrad_mfma_kernel(const float* __restrict__ in1, int64_t ld1,
                                                            const float* __restrict__ in2, int64_t ld2,
                                                            const float* __restrict__ packed, const float* __restrict__ go,
                                                            int64_t ldg, float* __restrict__ gw, int c3, int ldF, int ldG,
                                                            int64_t B, TpDev p) {
  using A = float;
  constexpr int R = RT;
  constexpr int XC = 3, GC = 2;   // columns of x / (c, w) pairs of G per thread (host: D1 <= 768, D3 M <= 512, RT Dy <= 256)
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int l3 = c3 >> 1, D3 = 2 * l3 + 1, K = p.K[c3], M = p.M[c3];
  float* xs = reinterpret_cast<float*>(smem_raw);
  float* ys = xs + (size_t)R * p.D1;
  float* Fs = ys + (((size_t)R * p.Dy + 3) & ~(size_t)3);
  float* Gs = Fs + (size_t)R * D3 * ldF;
  // the class's path descriptors in LDS (l1, l2, wrow, n, cbase): a GLOBAL read inside the tile loop is followed by vmcnt(0), which
  // would also wait for the whole register prefetch issued just before it
  int* ptab = reinterpret_cast<int*>(Gs + (size_t)R * D3 * ldG);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* normcol = packed + p.normcol_off;
  const int tm = (K + 31) / 32, tn = (M + 31) / 32, ntile = tm * tn;
  const int npath = p.npath[c3];
  if (tid < npath) {
    const TpPath P = p.paths[p.poff[c3] + tid];
    ptab[8 * tid] = P.l1; ptab[8 * tid + 1] = P.l2; ptab[8 * tid + 2] = P.wrow; ptab[8 * tid + 3] = p.n[P.c1]; ptab[8 * tid + 4] = p.cbase[P.c1];
  }
  tp_f32x16 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // padding columns are zero for the whole launch (the row passes never write them)
  for (int i = tid; i < R * D3 * ldF; i += 256) Fs[i] = 0.f;
  for (int i = tid; i < R * D3 * ldG; i += 256) Gs[i] = 0.f;
  // this thread's share of a tile's global reads (fixed for the launch): columns tid + 256 j of x, pairs tid + 256 j of G, one y
  int xcp[XC], gdst[GC], goc[GC];
  float gnv[GC];
  const int per = D3 * M;
#pragma unroll
  for (int j = 0; j < XC; ++j) xcp[j] = tid + 256 * j < p.D1 ? p.cpos[tid + 256 * j] : -1;
#pragma unroll
  for (int j = 0; j < GC; ++j) {
    const int e = tid + 256 * j;
    goc[j] = -1; gdst[j] = 0; gnv[j] = 0.f;
    if (e < per) {
      const int c = e / M, w = e - c * M;
      goc[j] = p.ocol[p.ocol_off[c3] + w] + c;
      gnv[j] = normcol[goc[j]];
      gdst[j] = c * ldG + w;
    }
  }
  const int yr_r = tid / p.Dy, yr_d = tid - yr_r * p.Dy;
  const bool yok = tid < R * p.Dy;
  // The rows of the NEXT tile travel in registers while this tile is contracted: issued right after the LDS image of the current
  // tile is complete, stored at the top of the next iteration (the loads of a tile used to be waited for in place)
  float xr[XC][RT], gr[GC][RT], yr = 0.f;
  auto issue = [&](const int64_t row0) {
    if (row0 + RT <= B) {   // whole tile inside the batch (every tile but the last): no per-row predicate, i.e. no branch per load
#pragma unroll
      for (int j = 0; j < XC; ++j)
        if (xcp[j] >= 0) {
#pragma unroll
          for (int u = 0; u < RT; ++u) xr[j][u] = in1[(row0 + u) * ld1 + tid + 256 * j];
        }
#pragma unroll
      for (int j = 0; j < GC; ++j)
        if (goc[j] >= 0) {
#pragma unroll
          for (int u = 0; u < RT; ++u) gr[j][u] = go[(row0 + u) * ldg + goc[j]];
        }
      if (yok) yr = in2[(row0 + yr_r) * ld2 + yr_d];
      return;
    }
#pragma unroll
    for (int j = 0; j < XC; ++j)
      if (xcp[j] >= 0) {
#pragma unroll
        for (int u = 0; u < RT; ++u) xr[j][u] = row0 + u < B ? in1[(row0 + u) * ld1 + tid + 256 * j] : 0.f;
      }
#pragma unroll
    for (int j = 0; j < GC; ++j)
      if (goc[j] >= 0) {
#pragma unroll
        for (int u = 0; u < RT; ++u) gr[j][u] = row0 + u < B ? go[(row0 + u) * ldg + goc[j]] : 0.f;
      }
    if (yok) yr = row0 + yr_r < B ? in2[(row0 + yr_r) * ld2 + yr_d] : 0.f;
  };
  const int64_t ntiles = (B + R - 1) / R;
  if ((int64_t)blockIdx.x < ntiles) issue((int64_t)blockIdx.x * R);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {  // uniform trip count per block: barriers are safe
    tp_lds_barrier();   // the previous tile's features / MFMA operands have been read
#pragma unroll
    for (int j = 0; j < XC; ++j)
      if (xcp[j] >= 0) {
#pragma unroll
        for (int u = 0; u < RT; ++u) xs[u * p.D1 + xcp[j]] = xr[j][u];
      }
#pragma unroll
    for (int j = 0; j < GC; ++j)
      if (goc[j] >= 0) {
#pragma unroll
        for (int u = 0; u < RT; ++u) Gs[(size_t)u * D3 * ldG + gdst[j]] = gr[j][u] * gnv[j];
      }
    if (yok) ys[tid] = yr;
    tp_lds_barrier();
    if (tile + gridDim.x < ntiles) issue((tile + gridDim.x) * R);
    // features of the tile: wave w owns rows w R/4 .. (w + 1) R/4; a path with n channels puts 64 / np2(n) rows side by side in
    // the 64 lanes (rows past B: x = y = 0 -> zero features)
    {
      const int rw = R >> 2, rbase = wave * rw;
      for (int pi = 0; pi < npath; ++pi) {
        const int Pl1 = __builtin_amdgcn_readfirstlane(ptab[8 * pi]), Pl2 = __builtin_amdgcn_readfirstlane(ptab[8 * pi + 1]);
        const int Pwrow = __builtin_amdgcn_readfirstlane(ptab[8 * pi + 2]), n = __builtin_amdgcn_readfirstlane(ptab[8 * pi + 3]);
        const int Pcb = __builtin_amdgcn_readfirstlane(ptab[8 * pi + 4]);
        int sh = 6;                                   // np2 = 1 << sh >= n, at most 64
        while (sh > 0 && (1 << (sh - 1)) >= n) --sh;
        const int np2 = 1 << sh, rpi = 64 >> sh, sub = lane >> sh, k0 = lane & (np2 - 1);
        for (int rb = 0; rb < rw; rb += rpi) {
          const int r = rbase + rb + sub;
          for (int k = k0; k < n; k += np2) {   // (one trip unless n > 64)
            if (rb + sub < rw) {
              E3_GRAD_SWITCH(Pl1, Pl2, l3, (tp_wgrad_feat_item<G>(xs + r * p.D1 + Pcb + k * G::D1,
                                                                  ys + r * p.Dy + G::D2 / 2 * (G::D2 / 2),
                                                                  Fs + (size_t)r * G::D3 * ldF + Pwrow + k, ldF)))
            }
          }
        }
      }
    }
    tp_lds_barrier();
    const int kh = lane >> 5, i32 = lane & 31;
    const int nk = R * D3;  // k rows of this tile (even: R is a multiple of 2)
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int tt = wave + 4 * t;
      if (tt < ntile) {
        const int mi = tt / tn, ni = tt - mi * tn;
        const float* fa = Fs + kh * ldF + 32 * mi + i32;
        const float* gb = Gs + kh * ldG + 32 * ni + i32;
        tp_f32x16 a = acc[t];
        for (int kk = 0; kk < nk; kk += 2) a = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk * ldF], gb[kk * ldG], a, 0, 0, 0);
        acc[t] = a;
      }
    }
  }
  // accumulators -> grad_W (C layout of the 32 x 32 MFMA: register r of lane l = C[8 (r >> 2) + 4 (l >> 5) + (r & 3)][l & 31])
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int tt = wave + 4 * t;
    if (tt < ntile) {
      const int mi = tt / tn, ni = tt - mi * tn;
      const int w = 32 * ni + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int krow = 32 * mi + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        if (krow < K && w < M) atomicAdd(&gw[(size_t)krow * M + w], acc[t][r]);
      }
    }
  }
}

static int tp_wgrad_geometry(const TpDev& p, int c3, int* R, int* ldF, int* ldG, size_t* lds, int* ntw) {
  const int D3 = 2 * (c3 >> 1) + 1, K = p.K[c3], M = p.M[c3];
  if (K <= 0 || M <= 0) return E3_ERR_INVALID_ARG;
  auto pad = [](int v) { int q = (v + 31) / 32 * 32; return (q % 64 == 32) ? q : q + 32; };
  *ldF = pad(K);
  *ldG = pad(M);
  const int ntile = ((K + 31) / 32) * ((M + 31) / 32);
  *ntw = (ntile + 3) / 4;
  if (*ntw > 8) return E3_ERR_UNSUPPORTED;
  auto need_of = [&](int r) {
    return ((size_t)r * p.D1 + (((size_t)r * p.Dy + 3) & ~(size_t)3) + (size_t)r * D3 * (*ldF + *ldG)) * sizeof(float) + 64 * 8 * sizeof(int);
  };
  // per-thread register share of a tile's reads (tp_wgrad_mfma_kernel): 3 columns of x, 2 (c, w) pairs of G, one y
  if (p.D1 > 768 || D3 * M > 512 || p.npath[c3] > 64) return E3_ERR_UNSUPPORTED;
  for (int r : {16, 8})
    if (need_of(r) <= (size_t)76 * 1024 && r * p.Dy <= 256) { *R = r; *lds = need_of(r); return E3_OK; }
  if (need_of(8) <= (size_t)150 * 1024) { *R = 8; *lds = need_of(8); return E3_OK; }
  return E3_ERR_UNSUPPORTED;
}

}  // namespace e3

using namespace e3;

extern "C" {

int e3_tp_plan_create(const int32_t* in1_blocks, int n_in1, int lmax_sh, const int32_t* out_blocks, int n_out,
                      e3_tp_plan** plan_out) {
  if (!in1_blocks || !out_blocks || n_in1 <= 0 || n_out <= 0 || !plan_out) return E3_ERR_INVALID_ARG;
  if (lmax_sh < 1 || lmax_sh > 2) return E3_ERR_BAD_IRREPS;
  auto* P = new e3_tp_plan();
  TpDev& d = P->dev;
  d.lmax_sh = lmax_sh;
  d.Dy = (lmax_sh + 1) * (lmax_sh + 1);
  for (int c = 0; c < 6; ++c) d.n[c] = d.M[c] = 0;
  struct Blk { int l, p, mul, col; };
  std::vector<Blk> bi, bo;
  auto parse = [&](const int32_t* b, int n, std::vector<Blk>& v, int* dim) {
    int col = 0;
    for (int i = 0; i < n; ++i) {
      int l = b[3 * i], p = b[3 * i + 1], mul = b[3 * i + 2];
      if (l < 0 || l > 2 || (p != 1 && p != -1) || mul < 0) return false;
      v.push_back({l, p, mul, col});
      col += (2 * l + 1) * mul;
    }
    *dim = col;
    return true;
  };
  if (!parse(in1_blocks, n_in1, bi, &d.D1) || !parse(out_blocks, n_out, bo, &d.Dout)) {
    delete P;
    return E3_ERR_BAD_IRREPS;
  }
  auto cls = [](int l, int p) { return 2 * l + (p == 1 ? 0 : 1); };
  for (auto& b : bi) d.n[cls(b.l, b.p)] += b.mul;
  for (auto& b : bo) d.M[cls(b.l, b.p)] += b.mul;
  int pos = 0;
  for (int c = 0; c < 6; ++c) { d.cbase[c] = pos; pos += d.n[c] * (2 * (c >> 1) + 1); }
  pos = 0;
  for (int c = 0; c < 6; ++c) { d.obase[c] = pos; pos += d.M[c] * (2 * (c >> 1) + 1); }
  std::vector<int32_t> tables(d.D1, 0);
  {
    int fill[6] = {0, 0, 0, 0, 0, 0};
    for (auto& b : bi) {
      int c = cls(b.l, b.p), w = 2 * b.l + 1;
      for (int i = 0; i < b.mul * w; ++i) tables[b.col + i] = d.cbase[c] + fill[c] + i;
      fill[c] += b.mul * w;
    }
  }
  int off = 0;
  for (int c = 0; c < 6; ++c) {
    d.ocol_off[c] = off;
    for (auto& b : bo)
      if (cls(b.l, b.p) == c)
        for (int i = 0; i < b.mul; ++i) tables.push_back(b.col + i * (2 * b.l + 1));
    off += d.M[c];
  }
  tables.push_back(0);
  P->h_tables = std::move(tables);
  // paths per out class, ordered by (l1, l2)
  int64_t wpos = 0;
  for (int c3 = 0; c3 < 6; ++c3) {
    const int l3 = c3 >> 1, p3 = (c3 & 1) ? -1 : 1;
    d.poff[c3] = (int)P->h_paths.size();
    d.npath[c3] = 0;
    int wrow = 0;
    for (int l1 = 0; l1 <= 2; ++l1)
      for (int l2 = 0; l2 <= lmax_sh; ++l2) {
        if (l3 < std::abs(l1 - l2) || l3 > l1 + l2) continue;
        const int p2 = (l2 & 1) ? -1 : 1;
        const int c1 = cls(l1, p3 * p2);
        if (d.n[c1] == 0) continue;
        if (d.M[c3] > 0) {
          P->h_paths.push_back({c1, l1, l2, wrow});
          d.npath[c3]++;
        }
        wrow += d.n[c1];
      }
    d.K[c3] = d.M[c3] > 0 ? wrow : 0;
    d.woff[c3] = wpos;
    wpos += (int64_t)d.K[c3] * d.M[c3];
  }
  d.normcol_off = wpos;
  d.packed_elems = wpos + d.Dout;
  {
    std::vector<std::array<int, 4>> blocks;
    for (auto& b : bi) blocks.push_back({b.l, b.p, b.mul, b.col});
    std::vector<TpPath> by_class[6];
    for (int c3 = 0; c3 < 6; ++c3)
      for (int i = 0; i < d.npath[c3]; ++i) by_class[c3].push_back(P->h_paths[d.poff[c3] + i]);
    fast_plan_init(&P->fast, d.n, d.M, lmax_sh, d.Dout, d.Dy, blocks, by_class, d.ocol_off);
    // fused gate epilogue: every natural-parity out class must be one contiguous run of columns, in class order
    {
      bool ok = true;
      int expect = 0;
      for (int c : {0, 3, 4}) {
        const int w = 2 * (c >> 1) + 1;
        for (int i = 0; i < d.M[c]; ++i) ok = ok && P->h_tables[d.D1 + d.ocol_off[c] + i] == expect + i * w;
        expect += d.M[c] * w;
      }
      P->fast.gate_layout = ok;
    }
  }
  *plan_out = P;
  return E3_OK;
}

int e3_tp_plan_destroy(e3_tp_plan* P) {
  if (!P) return E3_OK;
  if (P->d_tables) (void)hipFree(P->d_tables);
  if (P->d_paths) (void)hipFree(P->d_paths);
  fast_free(&P->fast);
  delete P;
  return E3_OK;
}

int e3_tp_in1_dim(const e3_tp_plan* p) { return p ? p->dev.D1 : -1; }
int e3_tp_in2_dim(const e3_tp_plan* p) { return p ? p->dev.Dy : -1; }
int e3_tp_out_dim(const e3_tp_plan* p) { return p ? p->dev.Dout : -1; }
int e3_tp_weight_shape(const e3_tp_plan* p, int cls, int* rows, int* cols) {
  if (!p || cls < 0 || cls > 5 || !rows || !cols) return E3_ERR_INVALID_ARG;
  bool present = p->dev.K[cls] > 0 && p->dev.M[cls] > 0;
  *rows = present ? p->dev.K[cls] : 0;
  *cols = present ? p->dev.M[cls] : 0;
  return E3_OK;
}
int e3_tp_norm_len(const e3_tp_plan* p, int cls) {
  return (p && cls >= 0 && cls < 6) ? p->dev.M[cls] * (2 * (cls >> 1) + 1) : -1;
}
int64_t e3_tp_packed_bytes(const e3_tp_plan* p, int dtype) {
  if (!p || dtype < 0 || dtype > 2) return -1;
  int64_t b = ((p->dev.packed_elems + 64) * (dtype == E3_F64 ? 8 : 4) + 255) / 256 * 256;
  if (dtype != E3_F64) b += fast_packed_bytes(&p->fast);
  return b;
}
static inline int64_t fast_section_offset(const e3_tp_plan* p) { return ((p->dev.packed_elems + 64) * 4 + 255) / 256 * 256; }

int e3_tp_pack_weights(const e3_tp_plan* plan, const void* const w[6], const void* const n[6], int dtype, void* packed,
                       void* stream) {
  if (!plan || !w || !packed || dtype < 0 || dtype > 2) return E3_ERR_INVALID_ARG;
  for (int c = 0; c < 6; ++c)
    if (plan->dev.M[c] > 0 && plan->dev.K[c] > 0 && !w[c]) return E3_ERR_MISSING_WEIGHT;
  int st = tp_ensure_device(plan);
  if (st != E3_OK) return st;
  const void* nn[6] = {0, 0, 0, 0, 0, 0};
  if (n)
    for (int c = 0; c < 6; ++c) nn[c] = plan->dev.M[c] > 0 ? n[c] : nullptr;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == E3_F32)
    hipLaunchKernelGGL(tp_pack_kernel<float>, dim3(64), dim3(256), 0, s, (const float*)w[0], (const float*)w[1],
                       (const float*)w[2], (const float*)w[3], (const float*)w[4], (const float*)w[5],
                       (const float*)nn[0], (const float*)nn[1], (const float*)nn[2], (const float*)nn[3],
                       (const float*)nn[4], (const float*)nn[5], (float*)packed, plan->dev);
  else if (dtype == E3_BF16)
    hipLaunchKernelGGL(tp_pack_kernel<bf16>, dim3(64), dim3(256), 0, s, (const bf16*)w[0], (const bf16*)w[1],
                       (const bf16*)w[2], (const bf16*)w[3], (const bf16*)w[4], (const bf16*)w[5], (const bf16*)nn[0],
                       (const bf16*)nn[1], (const bf16*)nn[2], (const bf16*)nn[3], (const bf16*)nn[4],
                       (const bf16*)nn[5], (float*)packed, plan->dev);
  else
    hipLaunchKernelGGL(tp_pack_kernel<double>, dim3(64), dim3(256), 0, s, (const double*)w[0], (const double*)w[1],
                       (const double*)w[2], (const double*)w[3], (const double*)w[4], (const double*)w[5],
                       (const double*)nn[0], (const double*)nn[1], (const double*)nn[2], (const double*)nn[3],
                       (const double*)nn[4], (const double*)nn[5], (double*)packed, plan->dev);
  E3_HIP_CHECK(hipGetLastError());
  if (dtype != E3_F64 && plan->fast.usable)
    return fast_pack(&plan->fast, w, nn, dtype, (char*)packed + fast_section_offset(plan), plan->dev.ocol, s);
  return E3_OK;
}

int e3_tp_forward(const e3_tp_plan* plan, const void* in1, int64_t ld1, const void* in2, int64_t ld2,
                  const void* packed, void* out, int64_t ldo, int64_t B, int dtype, const float* in_scale, int exact,
                  void* stream) {
  if (!plan || B < 0 || dtype < 0 || dtype > 2) return E3_ERR_INVALID_ARG;
  if (B == 0) return E3_OK;
  if (!in1 || !in2 || !packed || !out) return E3_ERR_INVALID_ARG;
  if (dtype == E3_BF16 && (!plan->fast.usable || ld2 == 0)) return E3_ERR_UNSUPPORTED;  // bf16 storage: MFMA path only
  if (ld1 < plan->dev.D1 || ldo < plan->dev.Dout || (ld2 != 0 && ld2 < plan->dev.Dy)) return E3_ERR_INVALID_ARG;
  int st = tp_ensure_device(plan);
  if (st != E3_OK) return st;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == E3_BF16 && exact) return E3_ERR_UNSUPPORTED;
  if (dtype != E3_F64 && plan->fast.usable && ld2 != 0 && !exact) {
    e3_tp_segment seg = {in1, ld1, nullptr, plan->dev.D1, 0};
    return fast_forward(&plan->fast, &seg, 1, plan->dev.D1, in2, ld2, (const char*)packed + fast_section_offset(plan),
                        out, ldo, B, 0, dtype, plan->dev.ocol, in_scale, s);
  }
  return dtype == E3_F32 ? tp_launch_fwd<float>(plan, in1, ld1, in2, ld2, packed, out, ldo, B, s)
                         : tp_launch_fwd<double>(plan, in1, ld1, in2, ld2, packed, out, ldo, B, s);
}

int e3_tp_forward_fused(const e3_tp_plan* plan, const e3_tp_segment* segs, int nseg, const void* in2, int64_t ld2,
                        const void* packed, void* out, int64_t ldo, int64_t B, int dtype, int gate,
                        const float* in_scale, void* stream) {
  if (!plan || !segs || B < 0) return E3_ERR_INVALID_ARG;
  if ((dtype != E3_F32 && dtype != E3_BF16) || !plan->fast.usable || ld2 == 0) return E3_ERR_UNSUPPORTED;
  if (B == 0) return E3_OK;
  if (!in2 || !packed || !out || ld2 < plan->dev.Dy) return E3_ERR_INVALID_ARG;
  int st = tp_ensure_device(plan);
  if (st != E3_OK) return st;
  return fast_forward(&plan->fast, segs, nseg, plan->dev.D1, in2, ld2, (const char*)packed + fast_section_offset(plan),
                      out, ldo, B, gate, dtype, plan->dev.ocol, in_scale, (hipStream_t)stream);
}

int e3_tp_forward_fused_epilogue(const e3_tp_plan* plan, const e3_tp_segment* segs, int nseg, const void* in2,
                                 int64_t ld2, const void* packed, void* out, int64_t ldo, int64_t B, int dtype, int gate,
                                 const float* in_scale, const void* residual, int64_t ld_residual, float* out_scale4,
                                 int target_log2, void* stream) {
  if (!plan || !segs || B < 0 || (residual && ld_residual <= 0)) return E3_ERR_INVALID_ARG;
  if (out_scale4 && (target_log2 < -20 || target_log2 > 14)) return E3_ERR_INVALID_ARG;
  if ((dtype != E3_F32 && dtype != E3_BF16) || !plan->fast.usable || ld2 == 0) return E3_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (out_scale4) E3_HIP_CHECK(hipMemsetAsync(out_scale4, 0, 16, s));
  if (B > 0) {
    if (!in2 || !packed || !out || ld2 < plan->dev.Dy) return E3_ERR_INVALID_ARG;
    int st = tp_ensure_device(plan);
    if (st != E3_OK) return st;
    st = fast_forward(&plan->fast, segs, nseg, plan->dev.D1, in2, ld2, (const char*)packed + fast_section_offset(plan), out,
                      ldo, B, gate, dtype, plan->dev.ocol, in_scale, s, nullptr, residual, ld_residual,
                      out_scale4 ? reinterpret_cast<uint32_t*>(out_scale4) + 2 : nullptr);
    if (st != E3_OK) return st;
  }
  return out_scale4 ? scale_finalize(out_scale4, target_log2, s) : E3_OK;
}

int e3_tp_forward_fused_scatter(const e3_tp_plan* plan, const e3_tp_segment* segs, int nseg, const void* in2,
                                int64_t ld2, const void* packed, const int32_t* row_node, void* out_nodes,
                                int64_t ldo, int64_t B, int dtype, int gate, const float* in_scale, void* stream) {
  if (!plan || !segs || !row_node || B < 0) return E3_ERR_INVALID_ARG;
  if ((dtype != E3_F32 && dtype != E3_BF16) || !gate || !plan->fast.usable || ld2 == 0) return E3_ERR_UNSUPPORTED;
  if (B == 0) return E3_OK;
  if (!in2 || !packed || !out_nodes || ld2 < plan->dev.Dy) return E3_ERR_INVALID_ARG;
  int st = tp_ensure_device(plan);
  if (st != E3_OK) return st;
  return fast_forward(&plan->fast, segs, nseg, plan->dev.D1, in2, ld2, (const char*)packed + fast_section_offset(plan),
                      out_nodes, ldo, B, gate, dtype, plan->dev.ocol, in_scale, (hipStream_t)stream, row_node);
}

int e3_tp_backward(const e3_tp_plan* plan, const void* in1, int64_t ld1, const void* in2, int64_t ld2,
                   const void* packed, const void* grad_out, int64_t ldg, void* grad_in1, int64_t ldg1,
                   void* grad_in2, int64_t ldg2, void* const grad_weights[6], int64_t B, int dtype, void* stream) {
  if (!plan || B < 0 || (dtype != E3_F32 && dtype != E3_F64)) return E3_ERR_INVALID_ARG;
  if (B == 0) return E3_OK;
  if (!in1 || !in2 || !packed || !grad_out) return E3_ERR_INVALID_ARG;
  if (ld1 < plan->dev.D1 || ldg < plan->dev.Dout || (ld2 != 0 && ld2 < plan->dev.Dy)) return E3_ERR_INVALID_ARG;
  if (grad_in1 && ldg1 < plan->dev.D1) return E3_ERR_INVALID_ARG;
  if (grad_in2 && ((ld2 == 0) != (ldg2 == 0) || (ldg2 != 0 && ldg2 < plan->dev.Dy))) return E3_ERR_INVALID_ARG;
  int st = tp_ensure_device(plan);
  if (st != E3_OK) return st;
  void* none[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  void* const* gw = grad_weights ? grad_weights : none;
  hipStream_t s = (hipStream_t)stream;
  return dtype == E3_F32 ? tp_launch_bwd<float>(plan, in1, ld1, in2, ld2, packed, grad_out, ldg, grad_in1, ldg1, grad_in2,
                                                ldg2, gw, B, s)
                         : tp_launch_bwd<double>(plan, in1, ld1, in2, ld2, packed, grad_out, ldg, grad_in1, ldg1,
                                                 grad_in2, ldg2, gw, B, s);
}

int e3_tp_backward_operands(const e3_tp_plan* plan, const void* in1, int64_t ld1, const void* in2, int64_t ld2,
                            const void* packed, const void* grad_out, int64_t ldg, void* const features[6],
                            void* const gout[6], int64_t B, int dtype, void* stream) {
  if (!plan || B < 0 || (dtype != E3_F32 && dtype != E3_F64)) return E3_ERR_INVALID_ARG;
  if (B == 0) return E3_OK;
  if (!in1 || !in2 || !packed || (!features && !gout) || (gout && !grad_out)) return E3_ERR_INVALID_ARG;
  if (ld1 < plan->dev.D1 || (gout && ldg < plan->dev.Dout) || (ld2 != 0 && ld2 < plan->dev.Dy)) return E3_ERR_INVALID_ARG;
  int st = tp_ensure_device(plan);
  if (st != E3_OK) return st;
  hipStream_t s = (hipStream_t)stream;
  return dtype == E3_F32
             ? tp_launch_bwd_operands<float>(plan, in1, ld1, in2, ld2, packed, grad_out, ldg, features, gout, B, s)
             : tp_launch_bwd_operands<double>(plan, in1, ld1, in2, ld2, packed, grad_out, ldg, features, gout, B, s);
}

int e3_tp_backward_contract(const e3_tp_plan* plan, const void* in1, int64_t ld1, const void* in2, int64_t ld2,
                            void* const t[6], void* grad_in1, int64_t ldg1, void* grad_in2, int64_t ldg2, int64_t B,
                            int dtype, void* stream) {
  if (!plan || B < 0 || (dtype != E3_F32 && dtype != E3_F64)) return E3_ERR_INVALID_ARG;
  if (B == 0) return E3_OK;
  if (!in1 || !in2 || !t || (!grad_in1 && !grad_in2)) return E3_ERR_INVALID_ARG;
  if (ld1 < plan->dev.D1 || (ld2 != 0 && ld2 < plan->dev.Dy)) return E3_ERR_INVALID_ARG;
  if (grad_in1 && ldg1 < plan->dev.D1) return E3_ERR_INVALID_ARG;
  if (grad_in2 && ((ld2 == 0) != (ldg2 == 0) || (ldg2 != 0 && ldg2 < plan->dev.Dy))) return E3_ERR_INVALID_ARG;
  int st = tp_ensure_device(plan);
  if (st != E3_OK) return st;
  hipStream_t s = (hipStream_t)stream;
  return dtype == E3_F32 ? tp_launch_bwd_contract<float>(plan, in1, ld1, in2, ld2, t, grad_in1, ldg1, grad_in2, ldg2, B, s)
                         : tp_launch_bwd_contract<double>(plan, in1, ld1, in2, ld2, t, grad_in1, ldg1, grad_in2, ldg2, B, s);
}

int e3_tp_backward_weights(const e3_tp_plan* plan, const void* in1, int64_t ld1, const void* in2, int64_t ld2,
                           const void* packed, const void* grad_out, int64_t ldg, void* const grad_weights[6], int64_t B,
                           int dtype, void* stream) {
  if (!plan || B < 0 || !grad_weights) return E3_ERR_INVALID_ARG;
  if (dtype != E3_F32) return E3_ERR_UNSUPPORTED;
  if (B == 0) return E3_OK;
  if (!in1 || !in2 || !packed || !grad_out) return E3_ERR_INVALID_ARG;
  if (ld1 < plan->dev.D1 || ldg < plan->dev.Dout || (ld2 != 0 && ld2 < plan->dev.Dy)) return E3_ERR_INVALID_ARG;
  int st = tp_ensure_device(plan);
  if (st != E3_OK) return st;
  const TpDev& p = plan->dev;
  hipStream_t s = (hipStream_t)stream;
  // every requested class must have a kernel before anything is launched (the caller falls back as a whole)
  int R[6], ldF[6], ldG[6], ntw[6];
  size_t lds[6];
  for (int c = 0; c < 6; ++c) {
    if (!grad_weights[c]) continue;
    st = tp_wgrad_geometry(p, c, &R[c], &ldF[c], &ldG[c], &lds[c], &ntw[c]);
    if (st != E3_OK) return st;
  }
  for (int c = 0; c < 6; ++c) {
    if (!grad_weights[c]) continue;
    const void* k16 = ntw[c] <= 2 ? (const void*)tp_wgrad_mfma_kernel<2, 16> : ntw[c] <= 4 ? (const void*)tp_wgrad_mfma_kernel<4, 16>
                    : ntw[c] <= 6 ? (const void*)tp_wgrad_mfma_kernel<6, 16> : (const void*)tp_wgrad_mfma_kernel<8, 16>;
    const void* k8 = ntw[c] <= 2 ? (const void*)tp_wgrad_mfma_kernel<2, 8> : ntw[c] <= 4 ? (const void*)tp_wgrad_mfma_kernel<4, 8>
                   : ntw[c] <= 6 ? (const void*)tp_wgrad_mfma_kernel<6, 8> : (const void*)tp_wgrad_mfma_kernel<8, 8>;
    const void* k = R[c] == 16 ? k16 : k8;
    { int st_ = tp_ensure_dyn_lds(k, lds[c]); if (st_ != E3_OK) return st_; }
    const int64_t ntiles = (B + R[c] - 1) / R[c];
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(156 * 1024) / lds[c]));
    const int grid = (int)std::min<int64_t>(ntiles, 256 * per_cu);
    const float* a1 = (const float*)in1; const float* a2 = (const float*)in2; const float* pk = (const float*)packed;
    const float* g = (const float*)grad_out; float* gw = (float*)grad_weights[c];
    int cc = c, lf = ldF[c], lg = ldG[c];
    TpDev pd = p;
    void* args[] = {&a1, &ld1, &a2, &ld2, &pk, &g, &ldg, &gw, &cc, &lf, &lg, &B, &pd};
    if (hipLaunchKernel(k, dim3(grid), dim3(256), args, lds[c], s) != hipSuccess) return E3_ERR_HIP;
  }
  return E3_OK;
}

const char* e3_tp_last_fused_kernel(void) { return fast_last_kernel(); }

int e3_tp_fused_supported(const e3_tp_plan* plan, int gate) {
  if (!plan || !plan->fast.usable) return 0;
  if (!gate) return 1;
  if (!plan->fast.gate_layout) return 0;
  return e3::fast_gate_shape_ok(&plan->fast) ? 1 : 0;
}

}  // extern "C"
