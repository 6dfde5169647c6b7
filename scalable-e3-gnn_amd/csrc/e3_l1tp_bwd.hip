// L1 tensor product backward (generic kernels: any irreps, fp32 / fp64 / bf16 storage).
//
// The reference relies on torch autograd through L1TP.py:234-299; here the adjoint is written out.
// With gn = grad_out * norm (per output column) and, per row,
//   out0e[m]   = Y0 S0e.A0e[:,m] + c3 Y1.(V1o.B0e[:,m])
//   out1o[w,c] = c3 Y1[c] S0e.A1o[:,w] + c3 Y0 (V1o.B1o[:,w])[c] + c6 ((V1e.C1o[:,w]) x Y1)[c]
// (0o / 1e: swap parities), A/B/C being the row blocks of the weight matrices (L1TP.py:81-88):
//   d s0e[k]   = Y0 A0e[k,:].gn0e + c3 Y1.(A1o[k,:].gn1o)
//   d v1o[k]   = c3 Y1 (B0e[k,:].gn0e) + c3 Y0 (B1o[k,:].gn1o) + c6 Y1 x (C1e[k,:].gn1e)
//   d Y0       = sum_m gn0e[m] (S0e.A0e[:,m]) + c3 sum_w gn1o[w].(V1o.B1o[:,w])           (+ 0o/1e terms)
//   d Y1       = c3 sum_m gn0e[m] (V1o.B0e[:,m]) + c3 sum_w gn1o[w] (S0e.A1o[:,w])
//                + c6 sum_w gn1o[w] x (V1e.C1o[:,w])                                       (+ 0o/1e terms)
//   d W        = sum over rows of feature (x) gn  (two-stage, deterministic reduction)
#include "e3_common.h"

namespace e3 {

constexpr int kR = 16;          // rows per tile
constexpr int kWChunk = 2048;   // weight elements per block in the dW kernel (8 per thread)
constexpr int kMaxRowChunks = 512;

template <typename T>
struct BwdArgs {
  const T* in1; int64_t ld1;
  const T* in2; int64_t ld2;
  const T* w[4];
  const T* nrm[4];
  const T* gout; int64_t ldg;
  int64_t B;
};

// Stage one tile: xs [R][D1] canonical in1, gs [R][Dout] canonical grad_out*norm, ys [R][4].
template <typename T, typename A>
__device__ __forceinline__ void stage_tile(const BwdArgs<T>& a, const PlanDev& p, int64_t row0, A* xs, A* gs, A* ys) {
  const int tid = threadIdx.x;
  for (int i = tid; i < kR * p.D1; i += 256) {
    int r = i / p.D1, d = i - r * p.D1;
    int64_t row = row0 + r;
    xs[r * p.D1 + p.cpos[d]] = row < a.B ? to_acc(a.in1[row * a.ld1 + d]) : A(0);
  }
  for (int i = tid; i < kR * p.Dout; i += 256) {
    int r = i / p.Dout, d = i - r * p.Dout;
    int64_t row = row0 + r;
    int pos = p.opos[d];
    int c = pos >= p.obase[3] ? 3 : pos >= p.obase[2] ? 2 : pos >= p.obase[1] ? 1 : 0;
    A nv = a.nrm[c] ? to_acc(a.nrm[c][pos - p.obase[c]]) : A(1);
    gs[r * p.Dout + pos] = row < a.B ? to_acc(a.gout[row * a.ldg + d]) * nv : A(0);
  }
  for (int i = tid; i < kR * 4; i += 256) {
    int64_t row = row0 + (i >> 2);
    ys[i] = row < a.B ? to_acc(a.in2[row * a.ld2 + (i & 3)]) : A(0);
  }
}

// ---------------------------------------------------------------------------------------------
// grad_in1 and grad_in2
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void l1tp_bwd_rows_kernel(BwdArgs<T> a, T* __restrict__ gin1, int64_t ldgi,
                                                            typename AccOf<T>::type* __restrict__ gy_rows,
                                                            T* __restrict__ gin2, PlanDev p) {
  using A = typename AccOf<T>::type;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  A* xs = reinterpret_cast<A*>(smem_raw);
  A* gs = xs + (size_t)kR * p.D1;
  A* ys = gs + (size_t)kR * p.Dout;
  const int tid = threadIdx.x;
  const A c3 = A(kC3), c6 = A(kC6);
  const int Ntot = p.n[0] + p.n[1] + p.n[2] + p.n[3];
  const int Mtot = p.M[0] + p.M[1] + p.M[2] + p.M[3];
  const int64_t ntiles = (a.B + kR - 1) / kR;

  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * kR;
    stage_tile<T, A>(a, p, row0, xs, gs, ys);
    __syncthreads();

    if (gin1) {
      for (int i = tid; i < kR * Ntot; i += 256) {
        int r = i / Ntot, it = i - r * Ntot;
        int64_t row = row0 + r;
        if (row >= a.B) continue;
        int cls = 0;
        while (it >= p.n[cls]) { it -= p.n[cls]; ++cls; }
        const A* g = gs + r * p.Dout;
        const A y0 = ys[r * 4], y1x = ys[r * 4 + 1], y1y = ys[r * 4 + 2], y1z = ys[r * 4 + 3];
        T* o = gin1 + row * ldgi + p.icol[p.icol_off[cls] + it];
        if (cls < 2) {
          // scalar channel k of parity class `cls`: rows k of W0(cls) and of W1(opposite parity out)
          const int so = cls;                  // scalar out class fed through Y0
          const int vo = (cls == 0) ? 3 : 2;   // vector out class fed through Y1
          A acc = 0, vx = 0, vy = 0, vz = 0;
          if (a.w[so]) {
            const T* W = a.w[so] + (int64_t)it * p.M[so];
            const A* gg = g + p.obase[so];
            for (int m = 0; m < p.M[so]; ++m) acc += to_acc(W[m]) * gg[m];
          }
          if (a.w[vo]) {
            const T* W = a.w[vo] + (int64_t)it * p.M[vo];
            const A* gg = g + p.obase[vo];
            for (int w = 0; w < p.M[vo]; ++w) {
              A wk = to_acc(W[w]);
              vx += wk * gg[3 * w];
              vy += wk * gg[3 * w + 1];
              vz += wk * gg[3 * w + 2];
            }
          }
          o[0] = from_acc<T, A>(y0 * acc + c3 * (y1x * vx + y1y * vy + y1z * vz));
        } else {
          // vector channel k of class `cls` (3 = 1o, 2 = 1e)
          const int so = (cls == 3) ? 0 : 1;   // scalar out fed by <v,Y1>:   0e <- v1o, 0o <- v1e
          const int vs = cls;                  // same-parity vector out (x Y0)
          const int vx_ = (cls == 3) ? 2 : 3;  // opposite-parity vector out (cross)
          const int ns_so = p.n[so];                                   // scalar rows before us in W0(so)
          const int ns_vs = p.n[(cls == 3) ? 0 : 1];                   // scalar rows before us in W1(vs)
          const int rows_before_cross = p.n[(vx_ == 3) ? 0 : 1] + p.n[vx_];  // [scalars | same vectors] of W1(vx_)
          A t = 0, sx = 0, sy = 0, sz = 0, cx = 0, cy = 0, cz = 0;
          if (a.w[so]) {
            const T* W = a.w[so] + (int64_t)(ns_so + it) * p.M[so];
            const A* gg = g + p.obase[so];
            for (int m = 0; m < p.M[so]; ++m) t += to_acc(W[m]) * gg[m];
          }
          if (a.w[vs]) {
            const T* W = a.w[vs] + (int64_t)(ns_vs + it) * p.M[vs];
            const A* gg = g + p.obase[vs];
            for (int w = 0; w < p.M[vs]; ++w) {
              A wk = to_acc(W[w]);
              sx += wk * gg[3 * w];
              sy += wk * gg[3 * w + 1];
              sz += wk * gg[3 * w + 2];
            }
          }
          if (a.w[vx_]) {
            const T* W = a.w[vx_] + (int64_t)(rows_before_cross + it) * p.M[vx_];
            const A* gg = g + p.obase[vx_];
            for (int w = 0; w < p.M[vx_]; ++w) {
              A wk = to_acc(W[w]);
              cx += wk * gg[3 * w];
              cy += wk * gg[3 * w + 1];
              cz += wk * gg[3 * w + 2];
            }
          }
          // Y1 x c
          A kx = y1y * cz - y1z * cy, ky = y1z * cx - y1x * cz, kz = y1x * cy - y1y * cx;
          o[0] = from_acc<T, A>(c3 * (y1x * t + y0 * sx) + c6 * kx);
          o[1] = from_acc<T, A>(c3 * (y1y * t + y0 * sy) + c6 * ky);
          o[2] = from_acc<T, A>(c3 * (y1z * t + y0 * sz) + c6 * kz);
        }
      }
    }

    if (gin2 || gy_rows) {
      // one wave per row (4 rows per wave), lanes over output channels, butterfly reduce
      const int wave = tid >> 6, lane = tid & 63;
      for (int r = wave; r < kR; r += 4) {
        int64_t row = row0 + r;
        if (row >= a.B) continue;  // wave-uniform
        const A* x = xs + r * p.D1;
        const A* g = gs + r * p.Dout;
        A d0 = 0, dx = 0, dy = 0, dz = 0;
        for (int i = lane; i < Mtot; i += 64) {
          int it = i, cls = 0;
          while (it >= p.M[cls]) { it -= p.M[cls]; ++cls; }
          const int M = p.M[cls];
          const T* W = a.w[cls] + it;
          if (cls < 2) {
            const int sc = cls, vc = (cls == 0) ? 3 : 2;
            const A* s = x + p.cbase[sc];
            const A* v = x + p.cbase[vc];
            A as = 0, ax = 0, ay = 0, az = 0;
            for (int k = 0; k < p.n[sc]; ++k) as += s[k] * to_acc(W[(int64_t)k * M]);
            const T* Wv = W + (int64_t)p.n[sc] * M;
            for (int k = 0; k < p.n[vc]; ++k) {
              A wk = to_acc(Wv[(int64_t)k * M]);
              ax += v[3 * k] * wk;
              ay += v[3 * k + 1] * wk;
              az += v[3 * k + 2] * wk;
            }
            A gg = g[p.obase[cls] + it];
            d0 += gg * as;
            dx += c3 * gg * ax;
            dy += c3 * gg * ay;
            dz += c3 * gg * az;
          } else {
            const int sc = (cls == 3) ? 0 : 1, v1 = cls, v2 = (cls == 3) ? 2 : 3;
            const A* s = x + p.cbase[sc];
            const A* va = x + p.cbase[v1];
            const A* vb = x + p.cbase[v2];
            A t0 = 0, ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0;
            for (int k = 0; k < p.n[sc]; ++k) t0 += s[k] * to_acc(W[(int64_t)k * M]);
            const T* Wa = W + (int64_t)p.n[sc] * M;
            for (int k = 0; k < p.n[v1]; ++k) {
              A wk = to_acc(Wa[(int64_t)k * M]);
              ax += va[3 * k] * wk;
              ay += va[3 * k + 1] * wk;
              az += va[3 * k + 2] * wk;
            }
            const T* Wb = Wa + (int64_t)p.n[v1] * M;
            for (int k = 0; k < p.n[v2]; ++k) {
              A wk = to_acc(Wb[(int64_t)k * M]);
              bx += vb[3 * k] * wk;
              by += vb[3 * k + 1] * wk;
              bz += vb[3 * k + 2] * wk;
            }
            const A* gg = g + p.obase[cls] + 3 * it;
            A gx = gg[0], gy = gg[1], gz = gg[2];
            d0 += c3 * (gx * ax + gy * ay + gz * az);
            // d/dY1 of c3 Y1[c] t0 g[c] + c6 (b x Y1).g  =  c3 t0 g + c6 (g x b)
            dx += c3 * t0 * gx + c6 * (gy * bz - gz * by);
            dy += c3 * t0 * gy + c6 * (gz * bx - gx * bz);
            dz += c3 * t0 * gz + c6 * (gx * by - gy * bx);
          }
        }
        for (int o = 32; o > 0; o >>= 1) {
          d0 += __shfl_xor(d0, o);
          dx += __shfl_xor(dx, o);
          dy += __shfl_xor(dy, o);
          dz += __shfl_xor(dz, o);
        }
        if (lane == 0) {
          if (gy_rows) {
            A* o = gy_rows + row * 4;
            o[0] = d0; o[1] = dx; o[2] = dy; o[3] = dz;
          } else {
            T* o = gin2 + row * 4;
            o[0] = from_acc<T, A>(d0); o[1] = from_acc<T, A>(dx);
            o[2] = from_acc<T, A>(dy); o[3] = from_acc<T, A>(dz);
          }
        }
      }
    }
    __syncthreads();
  }
}

// column-sum of a [n,4] array in two deterministic stages (used when in2 was broadcast)
template <typename A>
__global__ __launch_bounds__(256) void colsum4_stage1(const A* __restrict__ rows, int64_t n, A* __restrict__ part) {
  __shared__ A sm[256 * 4];
  A acc[4] = {0, 0, 0, 0};
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    for (int c = 0; c < 4; ++c) acc[c] += rows[i * 4 + c];
  for (int c = 0; c < 4; ++c) sm[threadIdx.x * 4 + c] = acc[c];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s)
      for (int c = 0; c < 4; ++c) sm[threadIdx.x * 4 + c] += sm[(threadIdx.x + s) * 4 + c];
    __syncthreads();
  }
  if (threadIdx.x < 4) part[blockIdx.x * 4 + threadIdx.x] = sm[threadIdx.x];
}
template <typename T, typename A>
__global__ void colsum4_stage2(const A* __restrict__ part, int nparts, T* __restrict__ out) {
  if (threadIdx.x < 4) {
    A acc = 0;
    for (int i = 0; i < nparts; ++i) acc += part[i * 4 + threadIdx.x];
    out[threadIdx.x] = from_acc<T, A>(acc);
  }
}

// ---------------------------------------------------------------------------------------------
// grad_weights: stage 1 = per row-chunk partial sums into the workspace, stage 2 = ordered reduce
// ---------------------------------------------------------------------------------------------
struct WOffsets { int64_t w[4]; int64_t sz[4]; int64_t total; };
__host__ __device__ inline int w_class_of(const WOffsets& wo, int64_t e) {
  for (int c = 0; c < 4; ++c)
    if (e >= wo.w[c] && e < wo.w[c] + wo.sz[c]) return c;
  return -1;
}

template <typename T>
__global__ __launch_bounds__(256) void l1tp_bwd_w_kernel(BwdArgs<T> a, typename AccOf<T>::type* __restrict__ partial,
                                                         WOffsets wo, int rows_per_chunk_tiles, PlanDev p) {
  using A = typename AccOf<T>::type;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  A* xs = reinterpret_cast<A*>(smem_raw);
  A* gs = xs + (size_t)kR * p.D1;
  A* ys = gs + (size_t)kR * p.Dout;
  const int tid = threadIdx.x;
  const A c3 = A(kC3), c6 = A(kC6);
  constexpr int EPT = kWChunk / 256;
  // decode this thread's weight elements once
  int e_cls[EPT], e_k[EPT], e_m[EPT];
  A acc[EPT];
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    int64_t e = (int64_t)blockIdx.y * kWChunk + j * 256 + tid;
    acc[j] = 0;
    e_cls[j] = -1; e_k[j] = 0; e_m[j] = 0;
    if (e < wo.total) {
      int c = w_class_of(wo, e);
      int64_t loc = e - wo.w[c];
      e_cls[j] = c;
      e_k[j] = (int)(loc / p.M[c]);
      e_m[j] = (int)(loc - (int64_t)e_k[j] * p.M[c]);
    }
  }
  const int64_t ntiles = (a.B + kR - 1) / kR;
  const int64_t t0 = (int64_t)blockIdx.x * rows_per_chunk_tiles;
  const int64_t t1 = t0 + rows_per_chunk_tiles < ntiles ? t0 + rows_per_chunk_tiles : ntiles;
  for (int64_t tile = t0; tile < t1; ++tile) {
    const int64_t row0 = tile * kR;
    stage_tile<T, A>(a, p, row0, xs, gs, ys);
    __syncthreads();
    const int nrows = (int)((a.B - row0) < kR ? (a.B - row0) : kR);
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int cls = e_cls[j];
      if (cls < 0) continue;
      const int k = e_k[j], m = e_m[j];
      A sum = 0;
      if (cls < 2) {
        const int sc = cls, vc = (cls == 0) ? 3 : 2;
        const int ns = p.n[sc];
        for (int r = 0; r < nrows; ++r) {
          const A* x = xs + r * p.D1;
          const A* y = ys + r * 4;
          A g = gs[r * p.Dout + p.obase[cls] + m];
          A f;
          if (k < ns) f = x[p.cbase[sc] + k] * y[0];
          else {
            const A* v = x + p.cbase[vc] + 3 * (k - ns);
            f = c3 * (v[0] * y[1] + v[1] * y[2] + v[2] * y[3]);
          }
          sum += f * g;
        }
      } else {
        const int sc = (cls == 3) ? 0 : 1, v1 = cls, v2 = (cls == 3) ? 2 : 3;
        const int ns = p.n[sc], na = p.n[v1];
        for (int r = 0; r < nrows; ++r) {
          const A* x = xs + r * p.D1;
          const A* y = ys + r * 4;
          const A* g = gs + r * p.Dout + p.obase[cls] + 3 * m;
          A f;
          if (k < ns) f = c3 * x[p.cbase[sc] + k] * (y[1] * g[0] + y[2] * g[1] + y[3] * g[2]);
          else if (k < ns + na) {
            const A* v = x + p.cbase[v1] + 3 * (k - ns);
            f = c3 * y[0] * (v[0] * g[0] + v[1] * g[1] + v[2] * g[2]);
          } else {
            const A* v = x + p.cbase[v2] + 3 * (k - ns - na);
            // (v x Y1) . g
            f = c6 * ((v[1] * y[3] - v[2] * y[2]) * g[0] + (v[2] * y[1] - v[0] * y[3]) * g[1] +
                      (v[0] * y[2] - v[1] * y[1]) * g[2]);
          }
          sum += f;
        }
      }
      acc[j] += sum;
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < EPT; ++j) {
    int64_t e = (int64_t)blockIdx.y * kWChunk + j * 256 + tid;
    if (e < wo.total) partial[(int64_t)blockIdx.x * wo.total + e] = acc[j];
  }
}

template <typename T>
__global__ void l1tp_bwd_w_reduce_kernel(const typename AccOf<T>::type* __restrict__ partial, int nchunks, WOffsets wo,
                                         T* g0, T* g1, T* g2, T* g3) {
  using A = typename AccOf<T>::type;
  T* gw[4] = {g0, g1, g2, g3};
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < wo.total; e += (int64_t)gridDim.x * blockDim.x) {
    A acc = 0;
    for (int c = 0; c < nchunks; ++c) acc += partial[(int64_t)c * wo.total + e];
    int cls = w_class_of(wo, e);
    if (gw[cls]) gw[cls][e - wo.w[cls]] = from_acc<T, A>(acc);
  }
}

static WOffsets w_offsets(const e3_l1tp_plan* p) {
  WOffsets o;
  int64_t pos = 0;
  for (int c = 0; c < 4; ++c) {
    o.w[c] = pos;
    o.sz[c] = (int64_t)p->wrows[c] * p->wcols[c];
    pos += o.sz[c];
  }
  o.total = pos;
  return o;
}

struct BwdGeom {
  int64_t ntiles;
  int nchunks, tiles_per_chunk;
  int64_t gy_off, part_off, part2_off, total;  // bytes
};
static BwdGeom bwd_geom(const e3_l1tp_plan* plan, int64_t B, size_t asz) {
  BwdGeom g;
  g.ntiles = (B + kR - 1) / kR;
  g.nchunks = (int)std::min<int64_t>(std::max<int64_t>(g.ntiles, 1), kMaxRowChunks);
  g.tiles_per_chunk = (int)((g.ntiles + g.nchunks - 1) / std::max(g.nchunks, 1));
  if (g.tiles_per_chunk < 1) g.tiles_per_chunk = 1;
  g.nchunks = (int)((g.ntiles + g.tiles_per_chunk - 1) / g.tiles_per_chunk);
  if (g.nchunks < 1) g.nchunks = 1;
  int64_t pos = 0;
  g.gy_off = pos;   pos += (B * 4 * (int64_t)asz + 255) / 256 * 256;
  g.part2_off = pos; pos += 256 * 4 * (int64_t)asz;
  g.part_off = pos; pos += ((int64_t)g.nchunks * w_offsets(plan).total * (int64_t)asz + 255) / 256 * 256;
  g.total = pos;
  return g;
}

template <typename T>
static int launch_backward(const e3_l1tp_plan* plan, const void* in1, int64_t ld1, const void* in2, int64_t ld2,
                           const void* const w[4], const void* const n[4], const void* gout, int64_t ldg,
                           void* gin1, int64_t ldgi, void* gin2, void* const gw[4], void* workspace, int64_t B,
                           hipStream_t stream) {
  using A = typename AccOf<T>::type;
  BwdArgs<T> a;
  a.in1 = (const T*)in1; a.ld1 = ld1; a.in2 = (const T*)in2; a.ld2 = ld2;
  for (int c = 0; c < 4; ++c) {
    a.w[c] = plan->wrows[c] > 0 ? (const T*)w[c] : nullptr;
    a.nrm[c] = (n && plan->normlen[c] > 0) ? (const T*)n[c] : nullptr;
  }
  a.gout = (const T*)gout; a.ldg = ldg; a.B = B;
  const PlanDev& p = plan->dev;
  size_t smem = (size_t)kR * (p.D1 + p.Dout + 4) * sizeof(A);
  if (smem > 160 * 1024) return E3_ERR_UNSUPPORTED;
  BwdGeom g = bwd_geom(plan, B, sizeof(A));
  char* ws = (char*)workspace;
  const bool bcast = (ld2 == 0);

  if (gin1 || gin2) {
    auto kern = l1tp_bwd_rows_kernel<T>;
    if (smem > 64 * 1024)
      E3_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    int grid = (int)std::min<int64_t>(g.ntiles, 256 * 8);
    A* gy_rows = (gin2 && bcast) ? (A*)(ws + g.gy_off) : nullptr;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, a, (T*)gin1, ldgi, gy_rows,
                       (T*)((gin2 && !bcast) ? gin2 : nullptr), p);
    E3_HIP_CHECK(hipGetLastError());
    if (gin2 && bcast) {
      A* part = (A*)(ws + g.part2_off);
      int nb = (int)std::min<int64_t>((B + 255) / 256, 256);
      hipLaunchKernelGGL(colsum4_stage1<A>, dim3(nb), dim3(256), 0, stream, gy_rows, B, part);
      hipLaunchKernelGGL((colsum4_stage2<T, A>), dim3(1), dim3(64), 0, stream, part, nb, (T*)gin2);
      E3_HIP_CHECK(hipGetLastError());
    }
  }
  bool any_w = false;
  if (gw)
    for (int c = 0; c < 4; ++c) any_w |= (gw[c] != nullptr && plan->wrows[c] > 0);
  if (any_w) {
    WOffsets wo = w_offsets(plan);
    auto kern = l1tp_bwd_w_kernel<T>;
    if (smem > 64 * 1024)
      E3_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    A* partial = (A*)(ws + g.part_off);
    dim3 grid(g.nchunks, (unsigned)((wo.total + kWChunk - 1) / kWChunk));
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, a, partial, wo, g.tiles_per_chunk, p);
    E3_HIP_CHECK(hipGetLastError());
    int rb = (int)std::min<int64_t>((wo.total + 255) / 256, 1024);
    hipLaunchKernelGGL(l1tp_bwd_w_reduce_kernel<T>, dim3(rb), dim3(256), 0, stream, partial, g.nchunks, wo,
                       (T*)gw[0], (T*)gw[1], (T*)gw[2], (T*)gw[3]);
    E3_HIP_CHECK(hipGetLastError());
  }
  return E3_OK;
}

}  // namespace e3

using namespace e3;

extern "C" {

int64_t e3_l1tp_backward_workspace_bytes(const e3_l1tp_plan* plan, int64_t B, int dtype) {
  if (!plan || B < 0 || dtype < 0 || dtype > 2) return -1;
  return bwd_geom(plan, B, dtype == E3_F64 ? 8 : 4).total;
}

int e3_l1tp_backward(const e3_l1tp_plan* plan, const void* in1, int64_t ld_in1, const void* in2, int64_t ld_in2,
                     const void* const weights[4], const void* const norms[4], const void* grad_out, int64_t ld_gout,
                     void* grad_in1, int64_t ld_gin1, void* grad_in2, void* const grad_weights[4], void* workspace,
                     int64_t B, int dtype, void* stream) {
  if (!plan || B < 0 || dtype < 0 || dtype > 2 || !weights) return E3_ERR_INVALID_ARG;
  if (B == 0) return E3_OK;  // caller zero-fills
  if (!in1 || !in2 || !grad_out || !workspace) return E3_ERR_INVALID_ARG;
  if (ld_in1 < plan->dev.D1 || ld_gout < plan->dev.Dout || (grad_in1 && ld_gin1 < plan->dev.D1)) return E3_ERR_INVALID_ARG;
  for (int c = 0; c < 4; ++c)
    if (plan->dev.M[c] > 0 && !(plan->wrows[c] > 0 && weights[c])) return E3_ERR_MISSING_WEIGHT;
  int st = ensure_device(plan);
  if (st != E3_OK) return st;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case E3_F32:
      return launch_backward<float>(plan, in1, ld_in1, in2, ld_in2, weights, norms, grad_out, ld_gout, grad_in1,
                                    ld_gin1, grad_in2, grad_weights, workspace, B, s);
    case E3_F64:
      return launch_backward<double>(plan, in1, ld_in1, in2, ld_in2, weights, norms, grad_out, ld_gout, grad_in1,
                                     ld_gin1, grad_in2, grad_weights, workspace, B, s);
    default:
      return launch_backward<bf16>(plan, in1, ld_in1, in2, ld_in2, weights, norms, grad_out, ld_gout, grad_in1,
                                   ld_gin1, grad_in2, grad_weights, workspace, B, s);
  }
}

}  // extern "C"
