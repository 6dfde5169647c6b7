// L1 tensor product: plan, weight packing, generic forward kernel and the C ABI.
//
// Reference semantics: /root/reference/models/segnn/l1_tensor_prod.py (cited as L1TP.py:<line>).
//   out0e = ([s0e*Y0 | c3 <v1o,Y1>] @ W0e) * norm0e                                  L1TP.py:242-256
//   out0o = ([s0o*Y0 | c3 <v1e,Y1>] @ W0o) * norm0o                                  L1TP.py:258-269
//   out1e = ([c3 s0o (x) Y1 | c3 v1e Y0 | c6 v1o x Y1] ._k W1e) * norm1e             L1TP.py:271-284
//   out1o = ([c3 s0e (x) Y1 | c3 v1o Y0 | c6 v1e x Y1] ._k W1o) * norm1o             L1TP.py:286-297
// The kernels use bilinearity to contract the channel index with W first and apply Y afterwards:
//   out1o[w,c] = norm * ( c3 (Y1[c] T0[w] + Y0 T1[w,c]) + c6 (T2[w] x Y1)[c] ),  T* = W-mixes of in1.
//
// This file holds the *generic* kernel: any irreps, fp32 / fp64 / bf16 storage (fp32 or fp64
// accumulate), plain FMA.  The fp32 MFMA kernel lives in e3_l1tp_mfma.hip.
#include "e3_common.h"

#include <cmath>
#include <cstring>
#include <mutex>

namespace e3 {

static thread_local std::string g_hip_err;
void set_hip_error(hipError_t e, const char* what) {
  g_hip_err = std::string(hipGetErrorName(e)) + ": " + hipGetErrorString(e) + " in " + what;
}


// Packed buffer, section 1 (all dtypes): [W0e | W0o | W1e | W1o | normcol[Dout]] in the accumulate
// type; section 2 (MFMA tiles) follows at a 256-byte aligned offset.
struct PackOffsets {
  int64_t w[4];
  int64_t normcol;
  int64_t end;  // elements
};
static PackOffsets pack_offsets(const e3_l1tp_plan* p) {
  PackOffsets o;
  int64_t pos = 0;
  for (int c = 0; c < 4; ++c) {
    o.w[c] = pos;
    pos += (int64_t)p->wrows[c] * p->wcols[c];
  }
  o.normcol = pos;
  pos += p->dev.Dout;
  o.end = (pos + 63) / 64 * 64;
  return o;
}

// -------------------------------------------------------------------------------------------------
// pack: convert weights to the accumulate type and expand the 4 norm buffers to one per-column vector
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_generic_kernel(const T* w0, const T* w1, const T* w2, const T* w3,
                                    const T* n0, const T* n1, const T* n2, const T* n3,
                                    typename AccOf<T>::type* packed, PackOffsets off, PlanDev p,
                                    int64_t nw0, int64_t nw1, int64_t nw2, int64_t nw3) {
  using A = typename AccOf<T>::type;
  const T* w[4] = {w0, w1, w2, w3};
  const T* nr[4] = {n0, n1, n2, n3};
  const int64_t nw[4] = {nw0, nw1, nw2, nw3};
  int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int c = 0; c < 4; ++c)
    for (int64_t i = tid; i < nw[c]; i += stride) packed[off.w[c] + i] = w[c] ? to_acc(w[c][i]) : A(0);
  // normcol: class c, channel m -> out columns ocol[m] (+comp for vectors); norm index runs along the
  // class's columns in order of appearance (L1TP.py:174,178,183,187)
  for (int c = 0; c < 4; ++c) {
    int width = (c >= 2) ? 3 : 1;
    for (int64_t i = tid; i < (int64_t)p.M[c] * width; i += stride) {
      int m = (int)(i / width), comp = (int)(i - (int64_t)m * width);
      int col = p.ocol[p.ocol_off[c] + m] + comp;
      packed[off.normcol + col] = nr[c] ? to_acc(nr[c][i]) : A(1);
    }
  }
}

// -------------------------------------------------------------------------------------------------
// generic forward
// -------------------------------------------------------------------------------------------------
// One workgroup (256 threads) per tile of R rows.  The tile of in1 is staged into LDS in canonical
// order [s0e | s0o | v1e xyz.. | v1o xyz..] (accumulate type); one thread then produces one
// (row, class, channel) output: 1 value for scalar classes, 3 for vector classes.
template <typename A>
struct VecAcc {
  A t0, t1x, t1y, t1z, t2x, t2y, t2z;
};

template <typename T, int R>
__global__ __launch_bounds__(256) void l1tp_fwd_generic_kernel(
    const T* __restrict__ in1, int64_t ld1, const T* __restrict__ in2, int64_t ld2,
    const typename AccOf<T>::type* __restrict__ packed, PackOffsets off, T* __restrict__ out,
    int64_t ldo, int64_t B, PlanDev p) {
  using A = typename AccOf<T>::type;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  A* xs = reinterpret_cast<A*>(smem_raw);  // [R][D1]
  A* ys = xs + (size_t)R * p.D1;           // [R][4]
  const int tid = threadIdx.x;
  const int D1 = p.D1;
  const int Mtot = p.M[0] + p.M[1] + p.M[2] + p.M[3];
  const A c3 = A(kC3), c6 = A(kC6);
  const A* normcol = packed + off.normcol;
  const int64_t ntiles = (B + R - 1) / R;

  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = tile * R;
    for (int i = tid; i < R * D1; i += 256) {
      int r = i / D1, d = i - r * D1;
      int64_t row = row0 + r;
      xs[r * D1 + p.cpos[d]] = row < B ? to_acc(in1[row * ld1 + d]) : A(0);
    }
    for (int i = tid; i < R * 4; i += 256) {
      int r = i >> 2, c = i & 3;
      int64_t row = row0 + r;
      ys[i] = row < B ? to_acc(in2[row * ld2 + c]) : A(0);
    }
    __syncthreads();

    for (int i = tid; i < R * Mtot; i += 256) {
      int r = i / Mtot, it = i - r * Mtot;
      int64_t row = row0 + r;
      if (row >= B) continue;
      int cls = 0;
      while (it >= p.M[cls]) { it -= p.M[cls]; ++cls; }
      const A* x = xs + r * D1;
      const A y0 = ys[r * 4 + 0], y1x = ys[r * 4 + 1], y1y = ys[r * 4 + 2], y1z = ys[r * 4 + 3];
      const int M = p.M[cls];
      const A* W = packed + off.w[cls] + it;  // column `it`, row stride M
      const int ocol = p.ocol[p.ocol_off[cls] + it];
      T* o = out + row * ldo + ocol;
      if (cls < 2) {
        // scalar output: sources = same-parity scalars (x Y0) and opposite-parity vectors (. Y1)
        const int sc = cls;                 // 0e <- s0e ; 0o <- s0o
        const int vc = (cls == 0) ? 3 : 2;  // 0e <- v1o ; 0o <- v1e
        const A* s = x + p.cbase[sc];
        const A* v = x + p.cbase[vc];
        const int ns = p.n[sc], nv = p.n[vc];
        A as = 0, ax = 0, ay = 0, az = 0;
        for (int k = 0; k < ns; ++k) as += s[k] * W[(int64_t)k * M];
        const A* Wv = W + (int64_t)ns * M;
        for (int k = 0; k < nv; ++k) {
          A wk = Wv[(int64_t)k * M];
          ax += v[3 * k + 0] * wk;
          ay += v[3 * k + 1] * wk;
          az += v[3 * k + 2] * wk;
        }
        A res = (y0 * as + c3 * (y1x * ax + y1y * ay + y1z * az)) * normcol[ocol];
        o[0] = from_acc<T, A>(res);
      } else {
        // vector output: rows = [scalars of parity -p | vectors of parity p | vectors of parity -p]
        const int sc = (cls == 3) ? 0 : 1;  // 1o <- s0e ; 1e <- s0o
        const int v1 = cls;                 // same-parity vectors (x Y0)
        const int v2 = (cls == 3) ? 2 : 3;  // opposite-parity vectors (cross Y1)
        const A* s = x + p.cbase[sc];
        const A* va = x + p.cbase[v1];
        const A* vb = x + p.cbase[v2];
        const int ns = p.n[sc], na = p.n[v1], nb = p.n[v2];
        A t0 = 0, ax = 0, ay = 0, az = 0, bx = 0, by = 0, bz = 0;
        for (int k = 0; k < ns; ++k) t0 += s[k] * W[(int64_t)k * M];
        const A* Wa = W + (int64_t)ns * M;
        for (int k = 0; k < na; ++k) {
          A wk = Wa[(int64_t)k * M];
          ax += va[3 * k + 0] * wk;
          ay += va[3 * k + 1] * wk;
          az += va[3 * k + 2] * wk;
        }
        const A* Wb = Wa + (int64_t)na * M;
        for (int k = 0; k < nb; ++k) {
          A wk = Wb[(int64_t)k * M];
          bx += vb[3 * k + 0] * wk;
          by += vb[3 * k + 1] * wk;
          bz += vb[3 * k + 2] * wk;
        }
        // (b x Y1)
        A cx = by * y1z - bz * y1y;
        A cy = bz * y1x - bx * y1z;
        A cz = bx * y1y - by * y1x;
        o[0] = from_acc<T, A>((c3 * (y1x * t0 + y0 * ax) + c6 * cx) * normcol[ocol + 0]);
        o[1] = from_acc<T, A>((c3 * (y1y * t0 + y0 * ay) + c6 * cy) * normcol[ocol + 1]);
        o[2] = from_acc<T, A>((c3 * (y1z * t0 + y0 * az) + c6 * cz) * normcol[ocol + 2]);
      }
    }
    __syncthreads();
  }
}

template <typename T>
static int launch_generic(const e3_l1tp_plan* plan, const void* in1, int64_t ld1, const void* in2, int64_t ld2,
                          const void* packed, void* out, int64_t ldo, int64_t B, hipStream_t stream) {
  using A = typename AccOf<T>::type;
  constexpr int R = 16;
  size_t smem = (size_t)R * (plan->dev.D1 + 4) * sizeof(A);
  if (smem > 160 * 1024) return E3_ERR_UNSUPPORTED;
  auto kern = l1tp_fwd_generic_kernel<T, R>;
  if (smem > 64 * 1024)
    E3_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  int64_t ntiles = (B + R - 1) / R;
  int grid = (int)std::min<int64_t>(ntiles, 256 * 8);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, (const T*)in1, ld1, (const T*)in2, ld2,
                     (const A*)packed, pack_offsets(plan), (T*)out, ldo, B, plan->dev);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

template <typename T>
static int launch_pack(const e3_l1tp_plan* plan, const void* const w[4], const void* const n[4], void* packed,
                       hipStream_t stream) {
  using A = typename AccOf<T>::type;
  const void* nn[4] = {nullptr, nullptr, nullptr, nullptr};
  if (n)
    for (int c = 0; c < 4; ++c) nn[c] = n[c];
  int64_t nw[4];
  for (int c = 0; c < 4; ++c) nw[c] = (int64_t)plan->wrows[c] * plan->wcols[c];
  hipLaunchKernelGGL(pack_generic_kernel<T>, dim3(64), dim3(256), 0, stream, (const T*)w[0], (const T*)w[1],
                     (const T*)w[2], (const T*)w[3], (const T*)nn[0], (const T*)nn[1], (const T*)nn[2],
                     (const T*)nn[3], (A*)packed, pack_offsets(plan), plan->dev, nw[0], nw[1], nw[2], nw[3]);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

int64_t generic_packed_elems(const e3_l1tp_plan* plan) { return pack_offsets(plan).end; }

}  // namespace e3

// =================================================================================================
// C ABI
// =================================================================================================
using namespace e3;

extern "C" {

int e3_abi_version(void) { return E3GNN_ABI_VERSION; }

const char* e3_status_string(int s) {
  switch (s) {
    case E3_OK: return "ok";
    case E3_ERR_INVALID_ARG: return "invalid argument";
    case E3_ERR_BAD_IRREPS: return "bad irreps (need l in {0,1}, lmax == 1, p in {+1,-1}, mul >= 0)";
    case E3_ERR_MISSING_WEIGHT: return "an output class has columns but no weight matrix";
    case E3_ERR_UNSUPPORTED: return "unsupported shape/dtype for the requested kernel";
    case E3_ERR_HIP: return "HIP runtime error";
    case E3_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown status";
  }
}

const char* e3_last_hip_error(void) { return g_hip_err.c_str(); }

static int parse_blocks(const int32_t* b, int n, std::vector<Block>& out, int* dim) {
  int col = 0, lmax = -1;
  for (int i = 0; i < n; ++i) {
    int l = b[3 * i], p = b[3 * i + 1], mul = b[3 * i + 2];
    if (l < 0 || l > 1 || (p != 1 && p != -1) || mul < 0) return E3_ERR_BAD_IRREPS;
    out.push_back({l, p, mul, col});
    col += (2 * l + 1) * mul;
    if (l > lmax) lmax = l;
  }
  if (lmax != 1) return E3_ERR_BAD_IRREPS;  // L1TP.py:13-14
  *dim = col;
  return E3_OK;
}

static inline int cls_of(int l, int p) { return l * 2 + (p == 1 ? 0 : 1); }

int e3_l1tp_plan_create(const int32_t* in1_blocks, int n_in1, const int32_t* out_blocks, int n_out,
                        e3_l1tp_plan** plan_out) {
  if (!in1_blocks || !out_blocks || n_in1 <= 0 || n_out <= 0 || !plan_out) return E3_ERR_INVALID_ARG;
  auto* P = new e3_l1tp_plan();
  int st;
  if ((st = parse_blocks(in1_blocks, n_in1, P->in1, &P->dev.D1)) != E3_OK ||
      (st = parse_blocks(out_blocks, n_out, P->out, &P->dev.Dout)) != E3_OK) {
    delete P;
    return st;
  }
  for (int c = 0; c < 4; ++c) P->dev.n[c] = P->dev.M[c] = 0;
  for (auto& b : P->in1) {
    int c = cls_of(b.l, b.p);
    if (b.mul > 0) P->irun[c].push_back({b.col, b.mul, 2 * b.l + 1});
    P->dev.n[c] += b.mul;
  }
  for (auto& b : P->out) {
    int c = cls_of(b.l, b.p);
    if (b.mul > 0) P->orun[c].push_back({b.col, b.mul, 2 * b.l + 1});
    P->dev.M[c] += b.mul;
  }
  const int* n = P->dev.n;
  // weight rows in forward-concat order (L1TP.py:81-88)
  int rows[4] = {n[0] + n[3], n[1] + n[2], n[1] + n[2] + n[3], n[0] + n[3] + n[2]};
  for (int c = 0; c < 4; ++c) {
    bool present = rows[c] > 0 && P->dev.M[c] > 0;
    P->wrows[c] = present ? rows[c] : 0;
    P->wcols[c] = present ? P->dev.M[c] : 0;
    P->normlen[c] = P->dev.M[c] * (c >= 2 ? 3 : 1);
  }
  // canonical staging order: [s0e | s0o | v1e | v1o] for in1 rows and for out / grad_out rows
  const int* M = P->dev.M;
  P->dev.cbase[0] = 0;
  P->dev.cbase[1] = n[0];
  P->dev.cbase[2] = n[0] + n[1];
  P->dev.cbase[3] = n[0] + n[1] + 3 * n[2];
  P->dev.obase[0] = 0;
  P->dev.obase[1] = M[0];
  P->dev.obase[2] = M[0] + M[1];
  P->dev.obase[3] = M[0] + M[1] + 3 * M[2];
  std::vector<int32_t> tables(P->dev.D1 + P->dev.Dout, 0);
  {
    int fill[4] = {0, 0, 0, 0};
    for (auto& b : P->in1) {
      int c = cls_of(b.l, b.p), w = 2 * b.l + 1;
      for (int i = 0; i < b.mul * w; ++i) tables[b.col + i] = P->dev.cbase[c] + fill[c] + i;
      fill[c] += b.mul * w;
    }
    int ofill[4] = {0, 0, 0, 0};
    for (auto& b : P->out) {
      int c = cls_of(b.l, b.p), w = 2 * b.l + 1;
      for (int i = 0; i < b.mul * w; ++i) tables[P->dev.D1 + b.col + i] = P->dev.obase[c] + ofill[c] + i;
      ofill[c] += b.mul * w;
    }
  }
  int off = 0;
  for (int c = 0; c < 4; ++c) {
    P->dev.icol_off[c] = off;
    for (auto& r : P->irun[c])
      for (int i = 0; i < r.count; ++i) tables.push_back(r.col + i * r.cstride);
    off += n[c];
  }
  off = 0;
  for (int c = 0; c < 4; ++c) {
    P->dev.ocol_off[c] = off;
    for (auto& r : P->orun[c])
      for (int i = 0; i < r.count; ++i) tables.push_back(r.col + i * r.cstride);
    off += M[c];
  }
  tables.push_back(0);  // never zero-sized
  P->h_tables = std::move(tables);
  st = mfma_plan_init(P);
  if (st != E3_OK) {
    delete P;
    return st;
  }
  *plan_out = P;
  return E3_OK;
}

// Device tables are uploaded on first use (so a plan can be created, and its shapes queried, on a
// host without a GPU).  One plan belongs to the device that is current at that first use.
}  // extern "C"
namespace e3 {
int ensure_device(const e3_l1tp_plan* cplan) {
  auto* P = const_cast<e3_l1tp_plan*>(cplan);
  std::lock_guard<std::mutex> lock(P->mu);
  int ndev = 0, cur = -1;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0 || hipGetDevice(&cur) != hipSuccess) return E3_ERR_NO_DEVICE;
  if (P->d_tables) return cur == P->device ? E3_OK : E3_ERR_INVALID_ARG;  // one plan = one device (create one per device)
  int32_t* d = nullptr;
  E3_HIP_CHECK(hipMalloc((void**)&d, P->h_tables.size() * sizeof(int32_t)));
  hipError_t e = hipMemcpy(d, P->h_tables.data(), P->h_tables.size() * sizeof(int32_t), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    set_hip_error(e, "plan table upload");
    (void)hipFree(d);
    return E3_ERR_HIP;
  }
  P->dev.cpos = d;
  P->dev.opos = d + P->dev.D1;
  P->dev.icol = P->dev.opos + P->dev.Dout;
  P->dev.ocol = P->dev.icol + (P->dev.n[0] + P->dev.n[1] + P->dev.n[2] + P->dev.n[3]);
  int st = mfma_plan_upload(P);
  if (st != E3_OK) {
    (void)hipFree(d);
    return st;
  }
  P->device = cur;
  P->d_tables = d;
  return E3_OK;
}
}  // namespace e3
extern "C" {

int e3_l1tp_plan_destroy(e3_l1tp_plan* plan) {
  if (!plan) return E3_OK;
  mfma_plan_free(plan);
  if (plan->d_tables) (void)hipFree(plan->d_tables);
  delete plan;
  return E3_OK;
}

int e3_l1tp_in1_dim(const e3_l1tp_plan* p) { return p ? p->dev.D1 : -1; }
int e3_l1tp_out_dim(const e3_l1tp_plan* p) { return p ? p->dev.Dout : -1; }
int e3_l1tp_weight_shape(const e3_l1tp_plan* p, int cls, int* rows, int* cols) {
  if (!p || cls < 0 || cls > 3 || !rows || !cols) return E3_ERR_INVALID_ARG;
  *rows = p->wrows[cls];
  *cols = p->wcols[cls];
  return E3_OK;
}
int e3_l1tp_norm_len(const e3_l1tp_plan* p, int cls) { return (p && cls >= 0 && cls < 4) ? p->normlen[cls] : -1; }

static inline size_t acc_size(int dtype) { return dtype == E3_F64 ? 8 : 4; }

int64_t e3_l1tp_packed_bytes(const e3_l1tp_plan* plan, int dtype) {
  if (!plan || dtype < 0 || dtype > 2) return -1;
  int64_t bytes = generic_packed_elems(plan) * (int64_t)acc_size(dtype);
  bytes = (bytes + 255) / 256 * 256;
  if (dtype != E3_F64) bytes += mfma_packed_bytes(plan);
  return bytes;
}

int e3_l1tp_pack_weights(const e3_l1tp_plan* plan, const void* const weights[4], const void* const norms[4],
                         int dtype, void* packed, void* stream) {
  if (!plan || !weights || !packed || dtype < 0 || dtype > 2) return E3_ERR_INVALID_ARG;
  for (int c = 0; c < 4; ++c) {
    if (plan->dev.M[c] > 0 && !(plan->wrows[c] > 0 && weights[c])) return E3_ERR_MISSING_WEIGHT;
  }
  hipStream_t s = (hipStream_t)stream;
  int st = ensure_device(plan);
  if (st != E3_OK) return st;
  switch (dtype) {
    case E3_F32: st = launch_pack<float>(plan, weights, norms, packed, s); break;
    case E3_F64: st = launch_pack<double>(plan, weights, norms, packed, s); break;
    default: st = launch_pack<bf16>(plan, weights, norms, packed, s); break;
  }
  if (st != E3_OK) return st;
  if (dtype != E3_F64 && mfma_packed_bytes(plan) > 0) {
    int64_t off = (generic_packed_elems(plan) * 4 + 255) / 256 * 256;
    st = mfma_pack(plan, weights, norms, dtype, (char*)packed + off, s);
  }
  return st;
}

int e3_l1tp_forward(const e3_l1tp_plan* plan, const void* in1, int64_t ld_in1, const void* in2, int64_t ld_in2,
                    const void* packed, void* out, int64_t ld_out, int64_t B, int dtype, int kernel, void* stream) {
  if (!plan || B < 0 || dtype < 0 || dtype > 2 || kernel < 0 || kernel > 2) return E3_ERR_INVALID_ARG;
  if (B == 0) return E3_OK;
  if (!in1 || !in2 || !packed || !out) return E3_ERR_INVALID_ARG;
  if (ld_in1 < plan->dev.D1 || ld_out < plan->dev.Dout || (ld_in2 != 0 && ld_in2 < 4)) return E3_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  int st = ensure_device(plan);
  if (st != E3_OK) return st;
  bool can_mfma = mfma_supported(plan, dtype);
  if (kernel == 2 && !can_mfma) return E3_ERR_UNSUPPORTED;
  if (kernel != 1 && can_mfma) {
    int64_t off = (generic_packed_elems(plan) * 4 + 255) / 256 * 256;
    return mfma_forward(plan, in1, ld_in1, in2, ld_in2, (const char*)packed + off, out, ld_out, B, dtype, s);
  }
  switch (dtype) {
    case E3_F32: return launch_generic<float>(plan, in1, ld_in1, in2, ld_in2, packed, out, ld_out, B, s);
    case E3_F64: return launch_generic<double>(plan, in1, ld_in1, in2, ld_in2, packed, out, ld_out, B, s);
    default: return launch_generic<bf16>(plan, in1, ld_in1, in2, ld_in2, packed, out, ld_out, B, s);
  }
}

}  // extern "C"
