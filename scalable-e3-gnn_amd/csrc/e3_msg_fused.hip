// The SEGNN message function as ONE launch per layer (BASELINE.json north_star: "fused per-edge CDNA4 HIP kernel"):
//
//     a_i = sum_{e: dst(e) = i}  gate( TP2( gate( TP1( [h_dst | h_src | d_e] ; Y_e ) ) ; Y_e ) )
//
// per 16-edge tile and wave: spherical harmonics from the two positions, message TP #1, gate, message TP #2, gate and the
// segment-sum over dst -- the [E, 288] messages, Y [E, 9] and d [E] never exist in HBM.  Hidden irreps Hx0e+Hx1o(+Hx2e),
// H in {16, 32, 64}; TP semantics, weight row order and norms are those of e3_tp_* (include/e3gnn.h), i.e. of the
// reference operator for l <= 1 (l1_tensor_prod.py:242-297).
//
// Formulation ("mix first"): for a path (l1, l2, l3) with weights W and coupling z[a][c] = sum_b C[a][b][c] Y_l2[b],
//     out[c][w] = sum_a z[a][c] * u[a][w],      u[a][w] = sum_k W[k][w] x[k][a]
// u is a plain GEMM of the RAW input channels -- the MFMA B operand is x itself, split into fp16 (hi, lo) ONCE per input
// degree and shared by every path and tile -- and the Y-dependent part is a small per-lane fold of accumulator tiles.
// Two consequences used here:
//   * the dst half of TP #1 does not depend on the edge at all: u_dst = W_dst h_dst is computed once per NODE by
//     msg_premix_kernel (1/24 of the rows) and enters the edge kernel as the initial value of the MFMA accumulator;
//   * the gated output of TP #1 sits in accumulator layout (lane = (edge, 4 channels)), which is exactly the B-operand
//     layout of TP #2 when the k order of the weights is permuted to match (kperm below): no transpose between them.
// Paths into scalar outputs (l3 = 0, l1 > 0) are cheaper "feature first": f[k] = sum_a z[a] x[k][a], one MFMA group.
//
// MFMA: v_mfma_f32_16x16x32_f16, A = weights [16 out channels][32 k], B = features [32 k][16 edges], D = [channel 4g + r]
// [edge j] in lane (j = lane & 15, g = lane >> 4).  fp32 products as hi*hi + hi*lo + lo*hi (split2_f16), operands scaled
// by powers of two: weights at pack time (header), h by `in_scale`, the gated messages per edge row in-kernel.
#include "e3_common.h"
#include "cg_tables.h"
#include "e3_msg_ws.h"

#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

// development knobs (tools/build_variant.sh builds experiment libraries with other values; the product uses the defaults)
#ifndef E3_MSG_WD
#define E3_MSG_WD 1    // weight prefetch distance in blocks (2: 27.1 vs 26.0 ms at 2 waves per SIMD -- registers)
#endif
#ifndef E3_MSG_UDP
#define E3_MSG_UDP 1   // pre-mix prefetch distance in blocks
#endif
#ifndef E3_MSG_WPS
#define E3_MSG_WPS 2   // waves per SIMD the H = 32, l_max = 2 kernel is compiled for (1: 48-55 ms at any prefetch distance)
#endif

#ifndef E3_MSG_STAMP
#define E3_MSG_STAMP 0  // 1: per-phase cycle counters (s_memtime) summed into g_msg_stamps -- development builds only
#endif                  //    (tools/build_variant.sh stamp -DE3_MSG_STAMP=1; STAMPS=1 tools/msg_micro.py prints them)

namespace e3 {
#if E3_MSG_STAMP
__device__ unsigned long long g_msg_stamps[8];
#endif

#include "e3_tp_mfma_core.h"

#include "e3_msg_common.h"

// ------------------------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------------------------
struct MsgPackDesc {  // one weight block
  int role;           // 0: TP #1 src rows, 1: TP #2, 2: TP #1 dst rows (pre-mix)
  int mat;            // output degree l3 = class matrix index
  int rowbase;        // row of channel 0 of this K step in the class matrix
  int nvalid;         // channels of this K step that exist (<= 32)
  int M, colbase;     // matrix width, first output channel of the tile
  int blk;            // block index inside the role
};
struct MsgPackArgs {
  const void* w1[3];
  const void* w2[3];
  const void* n1[3];
  const void* n2[3];
  int64_t nw1[3], nw2[3];  // elements per class matrix
  int rowd[3];             // row of the distance channel in class l's matrix (path (0, l, l)), TP #1
  int M0, H, LMAX, TT, NS, WD, o_norm1, o_norm2, o_wd, o_w, nblk;
};

__global__ void msg_absmax_kernel(MsgPackArgs a, uint32_t* hdr) {  // fp32 storage only
  float m1 = 0.f, m2 = 0.f;
  for (int c = 0; c < 3; ++c) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < a.nw1[c]; i += (int64_t)gridDim.x * blockDim.x)
      m1 = fmaxf(m1, fabsf(static_cast<const float*>(a.w1[c])[i]));
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < a.nw2[c]; i += (int64_t)gridDim.x * blockDim.x)
      m2 = fmaxf(m2, fabsf(static_cast<const float*>(a.w2[c])[i]));
  }
  for (int o = 32; o > 0; o >>= 1) { m1 = fmaxf(m1, __shfl_xor(m1, o)); m2 = fmaxf(m2, __shfl_xor(m2, o)); }
  if ((threadIdx.x & 63) == 0) {
    if (m1 > 0.f && m1 < INFINITY) atomicMax(hdr + 0, __builtin_bit_cast(uint32_t, m1));
    if (m2 > 0.f && m2 < INFINITY) atomicMax(hdr + 3, __builtin_bit_cast(uint32_t, m2));
  }
}

// T = float: fp16 (hi, lo) halves of w * sw.  T = bf16 (bf16 storage): the bf16 weights themselves in the hi half, sw = 1.
template <typename T>
__global__ void msg_pack_kernel(MsgPackArgs a, const MsgPackDesc* desc, int ndesc, float* packed) {
  constexpr bool IO16 = std::is_same<T, bf16>::value;
  const uint32_t* hb = reinterpret_cast<const uint32_t*>(packed);
  const float sw1 = IO16 ? 1.0f : pow2_scale_from_bits(hb[0], 13), sw2 = IO16 ? 1.0f : pow2_scale_from_bits(hb[3], 13);
  auto rd = [](const void* p, int64_t i) { return to_acc(static_cast<const T*>(p)[i]); };
  for (int r = blockIdx.x; r < ndesc; r += gridDim.x) {
    const MsgPackDesc q = desc[r];
    const void* W = q.role == 1 ? a.w2[q.mat] : a.w1[q.mat];
    const float sw = q.role == 1 ? sw2 : sw1;
    uint16_t* dst = reinterpret_cast<uint16_t*>(packed + a.o_w + ((size_t)q.role * a.nblk + q.blk) * 512);
    for (int i = threadIdx.x; i < 512; i += blockDim.x) {
      const int lane = i >> 3, jj = i & 7, ch = lane & 15, g = lane >> 4;
      const int k = kperm(g, jj);
      float v = 0.f;
      if (k < q.nvalid && q.colbase + ch < q.M) v = rd(W, (int64_t)(q.rowbase + k) * q.M + q.colbase + ch) * sw;
      if constexpr (IO16) {
        dst[lane * 8 + jj] = __builtin_bit_cast(uint16_t, (__bf16)v);
        dst[512 + lane * 8 + jj] = 0;
      } else {
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        dst[lane * 8 + jj] = __builtin_bit_cast(uint16_t, hi);
        dst[512 + lane * 8 + jj] = __builtin_bit_cast(uint16_t, lo);
      }
    }
  }
  if (blockIdx.x == 0) {
    // norm tables in accumulator order, already divided by the weight scale; d-term weights scaled like the others
    for (int i = threadIdx.x; i < a.NS * 16; i += blockDim.x) {
      const int s = i >> 4, ch = i & 15;
      const int T0 = a.TT * (1 + a.LMAX);
      int l3, t, c;
      if (s < T0) { l3 = 0; t = s; c = 0; }
      else if (s < T0 + 3 * a.TT) { l3 = 1; t = (s - T0) / 3; c = (s - T0) % 3; }
      else { l3 = 2; t = (s - T0 - 3 * a.TT) / 5; c = (s - T0 - 3 * a.TT) % 5; }
      const int idx = (16 * t + ch) * (2 * l3 + 1) + c;
      packed[a.o_norm1 + i] = (a.n1[l3] ? rd(a.n1[l3], idx) : 1.0f) / sw1;
      packed[a.o_norm2 + i] = (a.n2[l3] ? rd(a.n2[l3], idx) : 1.0f) / sw2;
    }
    for (int i = threadIdx.x; i < a.WD; i += blockDim.x) {
      const int T0 = a.TT * (1 + a.LMAX);
      int l, rem;
      if (i < T0 * 16) { l = 0; rem = i; }
      else if (i < (T0 + a.TT) * 16) { l = 1; rem = i - T0 * 16; }
      else { l = 2; rem = i - (T0 + a.TT) * 16; }
      const int M = l == 0 ? a.M0 : a.H;
      packed[a.o_wd + i] = rd(a.w1[l], (int64_t)a.rowd[l] * M + rem) * sw1;
    }
    if (threadIdx.x == 0) {
      packed[1] = sw1; packed[2] = 1.0f / sw1;
      packed[4] = sw2; packed[5] = 1.0f / sw2;
    }
    // header[6]: Bw with |gated message of product #1| <= Bw * max|input of the row| (true units):
    //   |sum_paths sum_a z[a][c] (W^T x)[a][w]| <= kZMax * (column sum of |W|) * max|x|,   kZMax = max over paths of
    //   max_c sum_a ||C[a][.][c]||_2 * sqrt(2 l2 + 1) = 5 (component-normalised harmonics; oracle/cg.py tables), gates <= 1.
    // The weights-stationary kernel derives the per-row power-of-two scale of product #2's fp16 (hi, lo) operands from it
    // (e3_msg_ws.hip); a bound that is loose by up to 2^14 costs no accuracy there.
    __shared__ uint32_t red[2];
    float bw = 0.f;
    for (int l = 0; l <= a.LMAX; ++l) {
      const int M = l == 0 ? a.M0 : a.H;
      const int rows = (int)(a.nw1[l] / M);
      if (threadIdx.x < 2) red[threadIdx.x] = 0u;
      __syncthreads();
      float cs = 0.f, nm = 0.f;
      for (int col = threadIdx.x; col < M; col += blockDim.x) {
        float sum = 0.f;
        for (int r = 0; r < rows; ++r) sum += fabsf(rd(a.w1[l], (int64_t)r * M + col));
        cs = fmaxf(cs, sum);
      }
      for (int i = threadIdx.x; i < M * (2 * l + 1); i += blockDim.x) nm = fmaxf(nm, a.n1[l] ? fabsf(rd(a.n1[l], i)) : 1.0f);
      if (cs > 0.f && cs < INFINITY) atomicMax(&red[0], __builtin_bit_cast(uint32_t, cs));
      if (nm > 0.f && nm < INFINITY) atomicMax(&red[1], __builtin_bit_cast(uint32_t, nm));
      __syncthreads();
      bw = fmaxf(bw, 5.0f * __builtin_bit_cast(float, red[0]) * __builtin_bit_cast(float, red[1]));
      __syncthreads();
    }
    if (threadIdx.x == 0) packed[6] = bw;
  }
}


// Everything one tensor product needs besides its inputs.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct TpCtx {
  __amdgpu_buffer_rsrc_t w;  // buffer descriptor over ALL weight blocks (wave-uniform, 4 SGPRs)
  uint32_t wrole;       // byte offset of this role's first block; block b at wrole + b * 2048 (1 KiB hi then 1 KiB lo)
  uint32_t woff;        // this lane's byte offset inside a block half: lane * 16
  const float* ud;      // FIRST: this lane's row of the dst pre-mix table (+ 4 g), else nullptr
  const float* wd;      // FIRST: d-term weights in LDS (+ 4 g)
  float dsc;            // FIRST: distance * xs
};

// A operand of one weight block: buffer load = descriptor (SGPRs) + scalar block offset + this lane's 32-bit offset, so
// the ~70 block addresses of a product cost no vector registers and no 64-bit address arithmetic
template <bool IO16>
__device__ __forceinline__ void load_w(const TpCtx& cx, const int blk, uint4& hi, uint4& lo) {
  const uint32_t so = cx.wrole + (uint32_t)blk * 2048u;
  const u32x4 h = __builtin_amdgcn_raw_buffer_load_b128(cx.w, cx.woff, so, 0);
  hi = uint4{h[0], h[1], h[2], h[3]};
  if constexpr (!IO16) {
    const u32x4 l = __builtin_amdgcn_raw_buffer_load_b128(cx.w, cx.woff + 1024u, so, 0);
    lo = uint4{l[0], l[1], l[2], l[3]};
  }
}

// ---- one tensor product as a flat, software-pipelined list of blocks ----------------------------------------------
// A block = one 16-channel output tile of one path: its weight operand(s) and (product #1) its 4 D1 pre-mix values are
// fetched while the PREVIOUS block computes, so no block starts with an exposed L2 round trip.  Blocks are ordered by input
// degree; the first block of a degree also loads that degree's inputs, builds the feature-first operand and the (hi, lo)
// halves.  Everything is indexed at compile time (BlkList::at(IDX)); state that survives a block lives in TpState.
struct BlkDesc {
  int l1, l2, l3, t;
  bool ff;             // feature-first path (scalar outputs from l1 > 0)
  bool first_of_l1;    // load / split the inputs of degree l1 here
  bool first_of_path;  // build z here
};
template <int LMAX, int TT>
struct BlkList {
  using G = MsgGeom<LMAX, TT>;
  // order inside a degree: the feature-first path (needs the fp32 inputs), then the mix-first paths by (l3, l2)
  static constexpr BlkDesc at(int idx) {
    int n = 0;
    for (int l1 = 0; l1 <= LMAX; ++l1) {
      bool first = true;
      for (int pass = 0; pass < 2; ++pass)
        for (int l3 = 0; l3 <= LMAX; ++l3)
          for (int l2 = 0; l2 <= LMAX; ++l2) {
            if (!G::ok(l1, l2, l3)) continue;
            const bool ff = l3 == 0 && l1 > 0;
            if (ff != (pass == 0)) continue;
            for (int t = 0; t < G::T(l3); ++t) {
              if (n == idx) return BlkDesc{l1, l2, l3, t, ff, first, t == 0};
              first = false;
              ++n;
            }
          }
    }
    return BlkDesc{-1, 0, 0, 0, false, false, false};
  }
  static constexpr int count() {
    int n = 0;
    while (at(n).l1 >= 0) ++n;
    return n;
  }
};

template <int KS>
struct TpState {
  uint4 xh[KS][5], xl[KS][5];  // (hi, lo) halves of the current degree's inputs, per component
  uint4 fh[KS], fl[KS];        // feature-first operand of the current degree
  float z[5][5];               // coupling of the current path
  // operand rings (all indices are compile-time): block IDX reads slot IDX % (D + 1) and requests block IDX + D
  uint4 wh[E3_MSG_WD + 1][KS], wl[E3_MSG_WD + 1][KS];
  f32x4 uin[E3_MSG_UDP + 1][5];
};

// weights of block IDX and, for product #1, its pre-mix values, both requested one block ahead.  (Measured on 1 M
// particles: a weight prefetch distance of two blocks costs 8 registers -> spills -> 27.1 vs 26.0 ms; touching the
// pre-mix rows of the tile's dst nodes ahead of product #1 (L2 warm-up) 25.9 vs 26.2 ms: neither is kept.)
template <int LMAX, int TT, bool IO16, int IDX>
__device__ __forceinline__ void tp_prefetch_w(const TpCtx& cx, uint4 (&wh)[MsgGeom<LMAX, TT>::KS],
                                              uint4 (&wl)[MsgGeom<LMAX, TT>::KS]) {
  using G = MsgGeom<LMAX, TT>;
  if constexpr (IDX < BlkList<LMAX, TT>::count()) {
    constexpr BlkDesc B = BlkList<LMAX, TT>::at(IDX);
    constexpr int T = G::T(B.l3), B0 = G::blk(B.l1, B.l2, B.l3);
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) load_w<IO16>(cx, B0 + ks * T + B.t, wh[ks], wl[ks]);
  }
}
template <int LMAX, int TT, bool FIRST, int IDX>
__device__ __forceinline__ void tp_prefetch_u(const TpCtx& cx, f32x4 (&uin)[5]) {
  using G = MsgGeom<LMAX, TT>;
  if constexpr (FIRST && IDX < BlkList<LMAX, TT>::count()) {
    constexpr BlkDesc B = BlkList<LMAX, TT>::at(IDX);
    constexpr int T = G::T(B.l3), U0 = G::uoff(B.l1, B.l2, B.l3);
#pragma unroll
    for (int a = 0; a < 2 * B.l1 + 1; ++a) uin[a] = *reinterpret_cast<const f32x4*>(cx.ud + U0 + (a * T + B.t) * 16);
  }
}

template <int LMAX, int TT, bool FIRST, bool IO16, int IDX, class XLOAD>
__device__ __forceinline__ void tp_block(const TpCtx& cx, const float (&y)[9], XLOAD& xload,
                                         TpState<MsgGeom<LMAX, TT>::KS>& st, f32x4 (&acc0)[MsgGeom<LMAX, TT>::T(0)],
                                         f32x4 (&acc1)[TT][3], f32x4 (&acc2)[LMAX == 2 ? TT : 1][5]) {
  using G = MsgGeom<LMAX, TT>;
  using BL = BlkList<LMAX, TT>;
  constexpr int KS = G::KS;
  constexpr BlkDesc B = BL::at(IDX);
  constexpr int L1 = B.l1, L2 = B.l2, L3 = B.l3, t = B.t;
  constexpr int D1 = 2 * L1 + 1, D3 = 2 * L3 + 1;
  // ---- later blocks' operands are requested first ----
  tp_prefetch_w<LMAX, TT, IO16, IDX + E3_MSG_WD>(cx, st.wh[(IDX + E3_MSG_WD) % (E3_MSG_WD + 1)], st.wl[(IDX + E3_MSG_WD) % (E3_MSG_WD + 1)]);
  tp_prefetch_u<LMAX, TT, FIRST, IDX + E3_MSG_UDP>(cx, st.uin[(IDX + E3_MSG_UDP) % (E3_MSG_UDP + 1)]);
  const uint4 (&wh)[KS] = st.wh[IDX % (E3_MSG_WD + 1)];
  const uint4 (&wl)[KS] = st.wl[IDX % (E3_MSG_WD + 1)];
  const f32x4 (&uin)[5] = st.uin[IDX % (E3_MSG_UDP + 1)];
  // ---- first block of a degree: inputs, feature-first operand, (hi, lo) halves ----
  if constexpr (B.first_of_l1) {
    float x[KS][8][D1];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xload(std::integral_constant<int, L1>{}, ks, x[ks]);
    if constexpr (L1 > 0 && G::ok(L1, L1, 0)) {
      float zz[D1][1];
      make_z<L1, L1, 0>(y, zz);
#pragma unroll
      for (int a = 0; a < D1; ++a) st.z[a][0] = zz[a][0];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float sum = zz[0][0] * x[ks][i][0];
#pragma unroll
          for (int a = 1; a < D1; ++a) sum = __builtin_fmaf(zz[a][0], x[ks][i][a], sum);
          f[i] = sum;
        }
        split8<IO16>(f, st.fh[ks], st.fl[ks]);
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int a = 0; a < D1; ++a) {
        float f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = x[ks][i][a];
        split8<IO16>(f, st.xh[ks][a], st.xl[ks][a]);
      }
  }
  if constexpr (B.first_of_path && !B.ff) {
    float zz[D1][D3];
    make_z<L1, L2, L3>(y, zz);
#pragma unroll
    for (int a = 0; a < D1; ++a)
#pragma unroll
      for (int c = 0; c < D3; ++c) st.z[a][c] = zz[a][c];
  }
  // ---- compute ----
  if constexpr (B.ff) {
    f32x4 o = acc0[t];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) o = mma3<IO16>(wh[ks], wl[ks], st.fh[ks], st.fl[ks], o);
    if constexpr (FIRST) {  // dst half: fold of the per-node pre-mix
#pragma unroll
      for (int a = 0; a < D1; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = __builtin_fmaf(uin[a][r], st.z[a][0], o[r]);
    }
    acc0[t] = o;
  } else {
    f32x4 u[D1];
#pragma unroll
    for (int a = 0; a < D1; ++a) {
      if constexpr (FIRST) u[a] = uin[a];
      else u[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (FIRST && L1 == 0) {
      const f32x4 wdv = *reinterpret_cast<const f32x4*>(cx.wd + G::wdoff(L3) + t * 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) u[0][r] = __builtin_fmaf(wdv[r], cx.dsc, u[0][r]);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int a = 0; a < D1; ++a) u[a] = mma3<IO16>(wh[ks], wl[ks], st.xh[ks][a], st.xl[ks][a], u[a]);
#pragma unroll
    for (int c = 0; c < D3; ++c)
#pragma unroll
      for (int a = 0; a < D1; ++a)
        if (z_nonzero<L1, L2, L3>(a, c)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (L3 == 0) acc0[t][r] = __builtin_fmaf(u[a][r], st.z[a][c], acc0[t][r]);
            else if constexpr (L3 == 1) acc1[t][c][r] = __builtin_fmaf(u[a][r], st.z[a][c], acc1[t][c][r]);
            else acc2[t][c][r] = __builtin_fmaf(u[a][r], st.z[a][c], acc2[t][c][r]);
          }
        }
  }
  if constexpr (IDX + 1 < BL::count())
    tp_block<LMAX, TT, FIRST, IO16, IDX + 1>(cx, y, xload, st, acc0, acc1, acc2);
}

// One tensor product on the lane's 16-edge tile.  XLOAD(l1tag, ks, x[8][D1]) delivers the (scaled) fp32 inputs of this
// lane: x[jj][a] = channel 32 ks + kperm(g, jj), component a.  acc0 / acc1 / acc2: output tiles per degree.
template <int LMAX, int TT, bool FIRST, bool IO16, class XLOAD>
__device__ __forceinline__ void tp_core(const TpCtx& cx, const float (&y)[9], XLOAD&& xload,
                                        f32x4 (&acc0)[MsgGeom<LMAX, TT>::T(0)], f32x4 (&acc1)[TT][3],
                                        f32x4 (&acc2)[LMAX == 2 ? TT : 1][5]) {
  constexpr int KS = MsgGeom<LMAX, TT>::KS;
  TpState<KS> st;
  // prologue of the rings: blocks 0 .. D - 1
  auto prime = [&](auto itag) {
    constexpr int I = decltype(itag)::value;
    if constexpr (I < E3_MSG_WD) tp_prefetch_w<LMAX, TT, IO16, I>(cx, st.wh[I % (E3_MSG_WD + 1)], st.wl[I % (E3_MSG_WD + 1)]);
    if constexpr (I < E3_MSG_UDP) tp_prefetch_u<LMAX, TT, FIRST, I>(cx, st.uin[I % (E3_MSG_UDP + 1)]);
  };
  prime(std::integral_constant<int, 0>{}); prime(std::integral_constant<int, 1>{});
  prime(std::integral_constant<int, 2>{}); prime(std::integral_constant<int, 3>{});
  static_assert(E3_MSG_WD <= 4 && E3_MSG_UDP <= 4, "ring prologue");
  tp_block<LMAX, TT, FIRST, IO16, 0>(cx, y, xload, st, acc0, acc1, acc2);
}


// ------------------------------------------------------------------------------------------------------------------
// pre-mix: U[n] = (W_dst * sw) (h[n] * xs) for every path of TP #1, in the layout the edge kernel reads as accumulator
// initial values.  One wave per 16 nodes; inputs straight from global memory (each lane reads its own k slots).
// ------------------------------------------------------------------------------------------------------------------
template <int LMAX, int TT, bool IO16>
__global__ __launch_bounds__(256) void msg_premix_kernel(const void* __restrict__ hv, int64_t ldh, int64_t N,
                                                         const float* __restrict__ packed,
                                                         const float* __restrict__ in_scale, float* __restrict__ U,
                                                         float* __restrict__ hmax) {
  using G = MsgGeom<LMAX, TT>;
  constexpr int ES = IO16 ? 2 : 4;
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int64_t wave0 = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const float xs = (!IO16 && in_scale) ? in_scale[0] : 1.0f;
  constexpr bool STAGE = G::KS == 1;  // (two K steps: the 6 weight blocks of a scalar path would take 96 registers)
  __shared__ __align__(16) float stage_all[STAGE ? 4 * 16 * (6 * 16 + 4) : 4];
  float* stage = stage_all + (STAGE ? ((threadIdx.x >> 6) * 16 * (6 * 16 + 4)) : 0);
  TpCtx cx;
  cx.w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(packed + G::o_w), 0, 3 * G::nblk() * 2048, 0x00020000);
  cx.wrole = 2 * G::nblk() * 2048;  // role 2
  cx.woff = lane * 16; cx.ud = nullptr; cx.wd = nullptr; cx.dsc = 0.f;
  const int64_t ntiles = (N + 15) / 16;
  for (int64_t tile = wave0; tile < ntiles; tile += nw) {
    const int64_t n = tile * 16 + j;
    const bool ok = n < N;
    const char* row = reinterpret_cast<const char*>(hv) + (ok ? n : N - 1) * ldh * ES;
    float* urow = U + (ok ? n : N - 1) * (int64_t)G::UD + 4 * g;
    float rowmax = 0.f;  // max |h[n] * xs| over this lane's channels (joined over the 4 lanes of the node below)
    auto per_l1 = [&](auto l1tag) {
      constexpr int L1 = decltype(l1tag)::value, D1 = 2 * L1 + 1;
      uint4 xh[G::KS][D1], xl[G::KS][D1];
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        float x[8][D1];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          if (16 * (2 * ks + p) < G::H)
            read_piece<D1, IO16>(row + (G::col0(L1) + (32 * ks + 16 * p + 4 * g) * D1) * ES, p, xs, x);
          else
            zero_piece<D1>(p, x);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int aa = 0; aa < D1; ++aa) rowmax = fmaxf(rowmax, fabsf(x[i][aa]));
#pragma unroll
        for (int a = 0; a < D1; ++a) {
          float f[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = x[i][a];
          split8<IO16>(f, xh[ks][a], xl[ks][a]);
        }
      }
      auto per_path = [&](auto l2tag, auto l3tag) {
        constexpr int L2 = decltype(l2tag)::value, L3 = decltype(l3tag)::value;
        if constexpr (G::ok(L1, L2, L3)) {
          constexpr int T = G::T(L3), B0 = G::blk(L1, L2, L3), U0 = G::uoff(L1, L2, L3);
          if constexpr (STAGE) {
            // Component-outer: the path's T weight blocks stay in registers, each component's [16 nodes][T x 16 channels]
            // block is transposed through LDS and leaves as 16-byte units that are CONTIGUOUS per node row (128 or 384
            // bytes) -- the direct form below stores 64-byte pieces, and the launch runs at the per-CU rate of such stores.
            uint4 wh[T][G::KS], wl[T][G::KS];
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
              for (int ks = 0; ks < G::KS; ++ks) load_w<IO16>(cx, B0 + ks * T + t, wh[t][ks], wl[t][ks]);
            constexpr int RSP = T * 16 + 4, UPR = T * 4;  // stage row stride (floats), 16-byte units per row
#pragma unroll
            for (int a = 0; a < D1; ++a) {
#pragma unroll
              for (int t = 0; t < T; ++t) {
                f32x4 u = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) u = mma3<IO16>(wh[t][ks], wl[t][ks], xh[ks][a], xl[ks][a], u);
                *reinterpret_cast<f32x4*>(stage + j * RSP + t * 16 + 4 * g) = u;
              }
              wave_sync_lds();
#pragma unroll
              for (int it = 0; it < (16 * UPR + 63) / 64; ++it) {
                const int unit = it * 64 + lane, row = unit / UPR, col = unit - row * UPR;
                const int64_t nn = tile * 16 + row;
                if (unit < 16 * UPR && nn < N)
                  *reinterpret_cast<f32x4*>(U + nn * (int64_t)G::UD + U0 + a * T * 16 + col * 4) =
                      *reinterpret_cast<const f32x4*>(stage + row * RSP + col * 4);
              }
              wave_sync_lds();
            }
          } else {
#pragma unroll
            for (int t = 0; t < T; ++t) {
              f32x4 u[D1];
#pragma unroll
              for (int a = 0; a < D1; ++a) u[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int ks = 0; ks < G::KS; ++ks) {
                uint4 ah, al;
                load_w<IO16>(cx, B0 + ks * T + t, ah, al);
#pragma unroll
                for (int a = 0; a < D1; ++a) u[a] = mma3<IO16>(ah, al, xh[ks][a], xl[ks][a], u[a]);
              }
              if (ok) {
#pragma unroll
                for (int a = 0; a < D1; ++a) *reinterpret_cast<f32x4*>(urow + U0 + (a * T + t) * 16) = u[a];
              }
            }
          }
        }
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      per_path(I0{}, I0{}); per_path(I1{}, I0{}); per_path(I2{}, I0{});
      per_path(I0{}, I1{}); per_path(I1{}, I1{}); per_path(I2{}, I1{});
      per_path(I0{}, I2{}); per_path(I1{}, I2{}); per_path(I2{}, I2{});
    };
    per_l1(std::integral_constant<int, 0>{});
    per_l1(std::integral_constant<int, 1>{});
    if constexpr (LMAX == 2) per_l1(std::integral_constant<int, 2>{});
    // per-node row maximum (scaled units): the weights-stationary edge kernel bounds a row's messages with it
    rowmax = fmaxf(rowmax, __shfl_xor(rowmax, 16));
    rowmax = fmaxf(rowmax, __shfl_xor(rowmax, 32));
    if (hmax && ok && lane < 16) hmax[n] = rowmax;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// the edge kernel
// ------------------------------------------------------------------------------------------------------------------
constexpr int msg_waves_per_simd(int lmax, int tt) {
  return (lmax == 2 ? 44 : 20) * tt <= 96 ? ((lmax == 2 && tt == 2) ? E3_MSG_WPS : 2) : 1;
}

template <int LMAX, int TT, bool IO16>
// (waves per SIMD fixed from both sides: with the minimum alone the scheduler of a small instantiation -- l_max = 1 needs ~120
// registers -- trades its load / compute overlap for an occupancy the launch does not use: 6.7 -> 8.7 ms)
__global__ __launch_bounds__(256, msg_waves_per_simd(LMAX, TT))
__attribute__((amdgpu_waves_per_eu(msg_waves_per_simd(LMAX, TT), msg_waves_per_simd(LMAX, TT)))) void msg_fused_kernel(
    const void* __restrict__ hv, int64_t ldh, const float4* __restrict__ pos4, const int32_t* __restrict__ src,
    const int32_t* __restrict__ dst, int64_t E, const float* __restrict__ packed, const float* __restrict__ U,
    const float* __restrict__ in_scale, float* __restrict__ out, int64_t ldo, int blk) {
  using G = MsgGeom<LMAX, TT>;
  constexpr int H = G::H, D = G::D, T0 = G::T(0), NS = G::NS;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (j = lane & 15 = the lane's edge, g = lane >> 4 = its channel group: every phase of the tile loop derives them from a
  // regenerated lane id, see item "nothing waits behind the atomics" in DESIGN.md §4.1)

  // workgroup tables: norm1 (x 1/xs), norm2, d-term weights
  float* n1tab = lds;
  float* n2tab = n1tab + NS * 16;
  float* wdtab = n2tab + NS * 16;
  float* wbuf = wdtab + G::WD + (size_t)wave * G::lds_wave;
  int* idbuf = reinterpret_cast<int*>(wdtab + G::WD + (size_t)4 * G::lds_wave) + wave * 64;  // edge ids of the wave's next tile
  // bf16 storage needs no operand scales (bf16 has the fp32 exponent range): xs = 1, messages unscaled
  constexpr int ES = IO16 ? 2 : 4;  // bytes per stored feature element
  const char* h = reinterpret_cast<const char*>(hv);
  const float xs = (!IO16 && in_scale) ? in_scale[0] : 1.0f, ixs = (!IO16 && in_scale) ? in_scale[1] : 1.0f;
  for (int i = tid; i < NS * 16; i += blockDim.x) {
    n1tab[i] = packed[G::o_norm1 + i] * ixs;
    n2tab[i] = packed[G::o_norm2 + i];
  }
  for (int i = tid; i < G::WD; i += blockDim.x) wdtab[i] = packed[G::o_wd + i];
  __syncthreads();

  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(packed + G::o_w), 0, 3 * G::nblk() * 2048, 0x00020000);

  // tile -> wave map, XCD-aware.  Workgroups b, b + 8, b + 16, ... share an XCD (round-robin dispatch; `b & 7` is a label
  // of the group, not the XCD's id): the group gets one contiguous eighth of the (Morton-ordered) tiles, and INSIDE the
  // eighth its workgroups are dealt chunks of 4 blk tiles round-robin -- so the ~256 waves that share one 4 MiB L2 sweep
  // through the same spatial neighbourhood together and the gathered h[src] / pre-mix rows are fetched once per XCD
  // instead of once per edge.  A wave owns `blk` consecutive tiles of a chunk (its segment sum carries across them).
  // (all of it in 32-bit scalars -- E < 2^31 because the edge ids are int32: 64-bit products are VALU instructions, which
  // turn the tile loop's control flow into vector code with its counters spilled to scratch)
  const int ntiles = (int)((E + 15) / 16);
  const int per_xcd = (int)(gridDim.x >> 3);  // grid is a multiple of 8
  const int tiles_per_xcd = (ntiles + 7) / 8;
  const int xcd_lo = (int)(blockIdx.x & 7) * tiles_per_xcd;
  const int xcd_hi = xcd_lo + tiles_per_xcd < ntiles ? xcd_lo + tiles_per_xcd : ntiles;
  const int wg_idx = (int)(blockIdx.x >> 3);
  const int Ei = (int)E;

  // running segment sum across consecutive tiles of this wave: node id (wave uniform) + NQ output columns per lane
  // (the gated message row [H | 3 H | 5 H] is exactly an `out` row: column 64 q + lane)
  constexpr int NQ = (D + 63) / 64;
  // LDS row stride of the transposed tile (floats): rows stay 16-byte aligned, so a lane stores its 4 channels x (2l+1)
  // components of a degree -- contiguous in the output row -- as 2l+1 ds_write_b128 (18 per tile instead of 288 b32 stores
  // with 2-4 way bank conflicts); D + 4 = 73 16-byte units per row for H = 32: the 16 rows of a lane group fall on 16 different bank
  // quads.  The column reads of the run sums are consecutive lanes -> consecutive words for any stride.
  constexpr int RS = D + 4;
#if E3_MSG_STAMP
  uint32_t st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = (uint32_t)__builtin_readcyclecounter();
#define E3_STAMP(i) { const uint32_t t_ = (uint32_t)__builtin_readcyclecounter(); st_acc[i] += t_ - st_t; st_t = t_; }
#else
#define E3_STAMP(i)
#endif
  int cur = -1;
  float carry[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) carry[q] = 0.f;
  // (vmcnt is in-order: a wait for ANY vector-memory operation issued after a flush also waits for the acknowledgement of
  // its atomics, a round trip to the memory side.  The flush must therefore not reload anything from scratch -- the lane
  // id is regenerated (v_mbcnt) instead of kept, and the row address is scalar base + lane offset, so that the compiler
  // has no `out + lane` pointer pair to hoist out of the tile loop and spill.)
  auto flush = [&]() {
    if (cur >= 0) {
      int ln;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
      float* o = out + (int64_t)cur * ldo;  // wave uniform
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (64 * q + ln < D) __builtin_amdgcn_global_atomic_fadd_f32(o + (64 * q + ln), carry[q]);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) carry[q] = 0.f;
  };

  // The edge ids of a tile (16 src, 16 dst) are copied into LDS one tile ahead by a 4-byte LDS-DMA issued together with the
  // previous tile's gather copies (lanes 0-15: src, 16-31: dst): they land under the wait the
  // gather needs anyway and occupy no register across the products.  (As prefetched VGPRs they were spilled right after
  // the load -- `global_load; s_waitcnt vmcnt(0); scratch_store`, twice per tile -- and reloaded behind the atomics.)
  auto copy_ids = [&](int tile, int ln) {
    const int row0 = __builtin_amdgcn_readfirstlane(tile * 16);  // (pinned to SGPRs: as vector induction variables they
    const int nrows = __builtin_amdgcn_readfirstlane(Ei - row0 < 16 ? Ei - row0 : 16);  // get spilled and reloaded here)
    const int jl = ln & 15;
    const int e = row0 + (jl < nrows ? jl : nrows - 1);
    // two exec-masked copies with scalar bases (a per-lane choice between the two pointers becomes a 64-bit VGPR select
    // whose operands get hoisted and spilled)
    if (ln < 16) __builtin_amdgcn_global_load_lds((glb_void_t*)(src + e), (lds_void_t*)idbuf, 4, 0, 0);
    else if (ln < 32) __builtin_amdgcn_global_load_lds((glb_void_t*)(dst + e), (lds_void_t*)idbuf, 4, 0, 0);
  };
  // outer = this workgroup's chunks, inner = the tiles of this wave's block of the chunk.
  // (A workgroup barrier per tile, so that the waves share one weight stream through L1, measured 26.9 vs 26.5 ms.)
  const int chunk = 4 * blk;
  const int n_outer = (tiles_per_xcd + chunk * per_xcd - 1) / (chunk * per_xcd);
  for (int ob = 0; ob < n_outer; ++ob) {
    const int b0 = __builtin_amdgcn_readfirstlane(xcd_lo + (ob * per_xcd + wg_idx) * chunk + wave * blk);
    const int b1 = __builtin_amdgcn_readfirstlane(b0 + blk < xcd_hi ? b0 + blk : xcd_hi);
    if (b0 < b1) {  // the block's first tile: its ids are copied now
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wave_sync_lds();
      copy_ids(b0, lane);
      wait_vm0();
      wave_sync_lds();
    }
    for (int tile = b0; tile < b1; ++tile) {
      const int row0 = __builtin_amdgcn_readfirstlane(tile * 16);
      const int nrows = __builtin_amdgcn_readfirstlane(Ei - row0 < 16 ? Ei - row0 : 16);
      int lt;  // lane id, regenerated per tile (the kernel-level copy is spilled across the products)
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lt));
      const int sid = idbuf[lt & 15], did = idbuf[16 + (lt & 15)];
      // per-tile opaque copies of loop-invariant addresses: without them LICM hoists ~50 table reads (200 registers) and
      // the block addresses out of the tile loop and spills them
      uint32_t woff = lt * 16;
      int tab0 = 0;
      asm volatile("" : "+v"(woff), "+v"(tab0));
      const float *n1p = n1tab + tab0, *n2p = n2tab + tab0, *wdp = wdtab + tab0;
      const int sd = (lt & 15) < nrows ? did : -1;

      // ---- stage the 16 h[src] rows by 16-byte LDS-DMA (the per-lane source address is the gather).  LDS image per wave:
      //      l_max 2: region A [row][1o | 2e] (2 H units of 16 bytes per row) then region B [row][0e] (H / 4 units);
      //      l_max 1: one region [row][0e | 1o] (H units).  Unit counts are powers of two, so a DMA instruction covers whole
      //      rows (row id by v_readlane: scalar address math) or 2^k rows (one shuffle), and no lane divides anything.
      // (the previous tile's LDS reads must have returned before the copies may land)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wave_sync_lds();
      E3_STAMP(5)  // loop top: ids, reloads
      {
        int ls;  // lane id, regenerated: the kernel-level copy is spilled across the products
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ls));
        auto stage_region = [&](auto utag, auto otag, char* dstb) {
          constexpr int UNITS = decltype(utag)::value, SRCOFF = decltype(otag)::value;  // units per row, first source element
          if constexpr (UNITS >= 64) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rid = __builtin_amdgcn_readlane(sid, r);
              const char* rowp = h + ((int64_t)rid * ldh + SRCOFF) * ES;
#pragma unroll
              for (int k = 0; k < UNITS / 64; ++k)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(rowp + (k * 64 + ls) * 16),
                                                 (lds_void_t*)(dstb + (r * UNITS + k * 64) * 16), 16, 0, 0);
            }
          } else {
            constexpr int RPI = 64 / UNITS;  // rows per instruction
            static_assert(RPI <= 16, "region rows shorter than 4 units are not supported");
            const int u = ls & (UNITS - 1), rl = ls / UNITS;
            int rid[16 / RPI];  // all shuffles first: they are LDS-pipe instructions, and hipcc guards every LDS access
#pragma unroll                  // behind an LDS-DMA in flight with vmcnt(0), which would serialise the copies
            for (int it = 0; it < 16 / RPI; ++it) rid[it] = __shfl(sid, it * RPI + rl);
#pragma unroll
            for (int it = 0; it < 16 / RPI; ++it)
              __builtin_amdgcn_global_load_lds((glb_void_t*)(h + ((int64_t)rid[it] * ldh + SRCOFF) * ES + u * 16),
                                               (lds_void_t*)(dstb + it * 1024), 16, 0, 0);
          }
        };
        // element counts per row: l_max 2: A = [1o | 2e] = 8 H, B = [0e] = H; l_max 1: 4 H
        char* wb = reinterpret_cast<char*>(wbuf);
        if (tile + 1 < b1) copy_ids(tile + 1, ls);
        if constexpr (LMAX == 2) {
          stage_region(std::integral_constant<int, 8 * H * ES / 16>{}, std::integral_constant<int, H>{}, wb);
          stage_region(std::integral_constant<int, H * ES / 16>{}, std::integral_constant<int, 0>{}, wb + 16 * 8 * H * ES);
        } else {
          stage_region(std::integral_constant<int, 4 * H * ES / 16>{}, std::integral_constant<int, 0>{}, wb);
        }
      }
      // ---- geometry while the copies fly ----
      float y[9], dist;
      {
        const float4 ps = pos4[sid], pd = pos4[did];
        if constexpr (LMAX == 2) edge_sh(ps, pd, y, dist); else edge_sh1(ps, pd, y, dist);
      }
      f32x4 a0[T0], a1[TT][3], a2[LMAX == 2 ? TT : 1][5];
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < T0; ++t) a0[t] = zero4;
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) a1[t][c] = zero4;
#pragma unroll
      for (int t = 0; t < (LMAX == 2 ? TT : 1); ++t)
#pragma unroll
        for (int c = 0; c < 5; ++c) a2[t][c] = zero4;

      E3_STAMP(0)  // gather issue + geometry
      wait_vm0();
      wave_sync_lds();
      E3_STAMP(1)  // gather wait

      // ---- tensor product #1: x from the staged rows, dst half as accumulator initial values ----
      {
        // (every phase regenerates the lane id it needs: v_mbcnt costs two instructions, a lane-derived value kept across
        // the products costs a register there or a scratch reload here)
        int LL;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(LL));
        const int j = LL & 15, g = LL >> 4;
        TpCtx cx;
        cx.w = wrsrc;
        cx.wrole = 0;
        cx.woff = woff;
        cx.ud = U + (int64_t)did * G::UD + 4 * g;
        cx.wd = wdp + 4 * g;
        cx.dsc = dist * xs;
        auto xload = [&](auto l1tag, int ks, auto& x) {
          constexpr int L1 = decltype(l1tag)::value, D1 = 2 * L1 + 1;
          // first element of degree L1 in this lane's staged row (see the LDS image above), in elements
          const int e0 = LMAX == 2 ? (L1 == 0 ? 16 * 8 * H + j * H : j * 8 * H + (L1 == 1 ? 0 : 3 * H))
                                   : j * 4 * H + (L1 == 0 ? 0 : H);
          const char* xrow = reinterpret_cast<const char*>(wbuf) + e0 * ES;
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            if (16 * (2 * ks + p) < H) read_piece<D1, IO16>(xrow + (32 * ks + 16 * p + 4 * g) * D1 * ES, p, xs, x);
            else zero_piece<D1>(p, x);
          }
        };
        tp_core<LMAX, TT, true, IO16>(cx, y, xload, a0, a1, a2);
      }

      E3_STAMP(2)  // product #1
      // ---- gate #1; the messages stay in accumulator layout = the B-operand layout of product #2 ----
      // (norm1 already carries 1 / (sw1 xs)); row scale for the fp16 split: max |m| of the lane's edge -> 2^10
      float amax = 0.f;
      {
        int LL;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(LL));
        const int g = LL >> 4;
        const f32x4* nt = reinterpret_cast<const f32x4*>(n1p) + g;  // slot s at nt[4 s]
        auto gate_block = [&](auto dtag, auto& acc, const int slot, const int gslot) {
          constexpr int Dc = decltype(dtag)::value;
#pragma unroll
          for (int t = 0; t < TT; ++t) {
            const f32x4 gn = nt[4 * (gslot + t)];
            float gt[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) gt[r] = sigmoid_(a0[gslot + t][r] * gn[r]);
#pragma unroll
            for (int c = 0; c < Dc; ++c) {
              const f32x4 nv = nt[4 * (slot + Dc * t + c)];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float v = acc[t][c][r] * nv[r] * gt[r];
                acc[t][c][r] = v;
                amax = fmaxf(amax, fabsf(v));
              }
            }
          }
        };
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          const f32x4 nv = nt[4 * t];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sv = a0[t][r] * nv[r];
            const float v = sv * sigmoid_(sv);
            a0[t][r] = v;
            amax = fmaxf(amax, fabsf(v));
          }
        }
        gate_block(std::integral_constant<int, 3>{}, a1, G::slot0(1), TT);
        if constexpr (LMAX == 2) gate_block(std::integral_constant<int, 5>{}, a2, G::slot0(2), 2 * TT);
      }
      float srow = 1.0f, isrow = 1.0f;
      if constexpr (!IO16) {
        amax = fmaxf(amax, __shfl_xor(amax, 16));
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        srow = pow2_scale_from_bits(__builtin_bit_cast(uint32_t, amax), 10);
        isrow = 1.0f / srow;
      }

      // ---- park the (scaled) messages: slot q of lane l at wbuf[(q * 64 + l) * 4] ----
      wave_sync_lds();
      {
        int LL;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(LL));
        f32x4* pk = reinterpret_cast<f32x4*>(wbuf) + LL;
        auto sc4 = [&](const f32x4 v) {
          if constexpr (IO16) {  // bf16 storage: the messages between the two products are bf16 values
            f32x4 o;
            for (int r = 0; r < 4; ++r) o[r] = (float)(__bf16)v[r];
            return o;
          } else {
            return f32x4{v[0] * srow, v[1] * srow, v[2] * srow, v[3] * srow};
          }
        };
#pragma unroll
        for (int t = 0; t < TT; ++t) pk[64 * t] = sc4(a0[t]);
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
          for (int c = 0; c < 3; ++c) pk[64 * (TT + 3 * t + c)] = sc4(a1[t][c]);
        if constexpr (LMAX == 2) {
#pragma unroll
          for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int c = 0; c < 5; ++c) pk[64 * (4 * TT + 5 * t + c)] = sc4(a2[t][c]);
        }
      }
      wave_sync_lds();

      E3_STAMP(3)  // gate #1 + park
      // ---- tensor product #2 ----
#pragma unroll
      for (int t = 0; t < T0; ++t) a0[t] = zero4;
#pragma unroll
      for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) a1[t][c] = zero4;
#pragma unroll
      for (int t = 0; t < (LMAX == 2 ? TT : 1); ++t)
#pragma unroll
        for (int c = 0; c < 5; ++c) a2[t][c] = zero4;
      {
        TpCtx cx;
        cx.w = wrsrc; cx.wrole = G::nblk() * 2048; cx.woff = woff; cx.ud = nullptr; cx.wd = nullptr; cx.dsc = 0.f;
        int LL;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(LL));
        const f32x4* pk = reinterpret_cast<const f32x4*>(wbuf) + LL;
        auto xload = [&](auto l1tag, int ks, auto& x) {
          constexpr int L1 = decltype(l1tag)::value, D1 = 2 * L1 + 1;
          constexpr int S0 = L1 == 0 ? 0 : L1 == 1 ? TT : 4 * TT;  // first parked slot of degree L1
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const int t = 2 * ks + p;
#pragma unroll
            for (int a = 0; a < D1; ++a) {
              f32x4 v = zero4;
              if (t < TT) v = pk[64 * (S0 + D1 * t + a)];
#pragma unroll
              for (int r = 0; r < 4; ++r) x[4 * p + r][a] = v[r];
            }
          }
        };
        tp_core<LMAX, TT, false, IO16>(cx, y, xload, a0, a1, a2);
      }

      E3_STAMP(4)  // product #2
      // ---- gate #2, then the segment sum: the whole gated tile goes to LDS as [row][output column] (the accumulators are
      //      dead from here on), lane = output column walks the 16 rows and adds runs of equal dst; the last run is carried
      //      into the wave's next tile ----
      {
        int LL;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(LL));
        const int j = LL & 15, g = LL >> 4;
        const f32x4* nt2 = reinterpret_cast<const f32x4*>(n2p) + g;
        wave_sync_lds();
        float* wp = wbuf + j * RS + 4 * g;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
          const f32x4 nv = nt2[4 * t];
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sv = a0[t][r] * nv[r] * isrow;
            o[r] = sv * sigmoid_(sv);
          }
          *reinterpret_cast<f32x4*>(wp + 16 * t) = o;
        }
        auto put_block = [&](auto dtag, auto& acc, const int slot, const int gslot, float* wq) {
          constexpr int Dc = decltype(dtag)::value;
#pragma unroll
          for (int t = 0; t < TT; ++t) {
            const f32x4 gn = nt2[4 * (gslot + t)];
            float gt[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) gt[r] = sigmoid_(a0[gslot + t][r] * gn[r] * isrow) * isrow;
            float o[4 * Dc];  // the lane's 4 channels x Dc components: one contiguous piece of the output row
#pragma unroll
            for (int c = 0; c < Dc; ++c) {
              const f32x4 nv = nt2[4 * (slot + Dc * t + c)];
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r * Dc + c] = acc[t][c][r] * nv[r] * gt[r];
            }
            f32x4* dq = reinterpret_cast<f32x4*>(wq + 16 * t * Dc);
#pragma unroll
            for (int k = 0; k < Dc; ++k) dq[k] = f32x4{o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]};
          }
        };
        put_block(std::integral_constant<int, 3>{}, a1, G::slot0(1), TT, wbuf + j * RS + H + 12 * g);
        if constexpr (LMAX == 2) put_block(std::integral_constant<int, 5>{}, a2, G::slot0(2), 2 * TT, wbuf + j * RS + 4 * H + 20 * g);
        wave_sync_lds();
        E3_STAMP(6)  // gate #2 + transposed writes
        // from here to the next tile's gather nothing may wait on vmcnt behind a flush (see `flush`): the lane id is
        // regenerated, the next tile's ids are made to have arrived
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        const float* sp = wbuf + ln;
#pragma unroll
        for (int rb = 0; rb < 16; rb += 4) {
          float v[4][NQ];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int q = 0; q < NQ; ++q) v[rr][q] = (64 * q + ln < D) ? sp[(rb + rr) * RS + 64 * q] : 0.f;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int dn = __builtin_amdgcn_readlane(sd, rb + rr);
            if (dn >= 0) {
              if (dn != cur) {
                flush();
                cur = dn;
              }
#pragma unroll
              for (int q = 0; q < NQ; ++q) carry[q] += v[rr][q];
            }
          }
        }
      }
      E3_STAMP(7)  // run sums + flushes
    }
    flush();  // the wave's next tile is not the successor of this one
    cur = -1;
  }
#if E3_MSG_STAMP
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_msg_stamps[i], (unsigned long long)st_acc[i]);
#endif
#undef E3_STAMP
}

// ------------------------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------------------------
struct MsgKernels {
  int lmax, tt;
  const void* fused[2];   // [0] fp32 storage, [1] bf16 storage (nullptr: not instantiated)
  const void* premix[2];
  int64_t total_floats;
  int UD, D, lds_tab, lds_wave, nblk, NS, WD, o_norm1, o_norm2, o_wd, o_w, waves_per_simd;
};
template <int LMAX, int TT>
static MsgKernels make_entry() {
  using G = MsgGeom<LMAX, TT>;
  MsgKernels k = {LMAX, TT, {(const void*)msg_fused_kernel<LMAX, TT, false>, nullptr},
                  {(const void*)msg_premix_kernel<LMAX, TT, false>, nullptr}, G::total_floats,
                  G::UD, G::D, G::lds_tab, G::lds_wave, G::nblk(), G::NS, G::WD, G::o_norm1, G::o_norm2, G::o_wd, G::o_w,
                  msg_waves_per_simd(LMAX, TT)};
  if constexpr (TT >= 2) {  // bf16 storage: the [0e] region of a staged row must be at least 4 units of 16 bytes (H >= 32)
    k.fused[1] = (const void*)msg_fused_kernel<LMAX, TT, true>;
    k.premix[1] = (const void*)msg_premix_kernel<LMAX, TT, true>;
  }
  return k;
}
static const std::vector<MsgKernels>& msg_kernels() {
  static const std::vector<MsgKernels> k = {make_entry<2, 2>(), make_entry<1, 2>(), make_entry<2, 1>(), make_entry<1, 1>(),
                                            make_entry<2, 4>(), make_entry<1, 4>()};
  return k;
}

}  // namespace e3

using namespace e3;

struct e3_msg_plan {
  int lmax, H;
  const MsgKernels* k;
  std::vector<MsgPackDesc> desc;
  int rowd[3];
  int K1[3], K2[3], M[3];
  MsgPackDesc* d_desc = nullptr;
  int device = -1;
  int grid[2] = {0, 0};  // workgroups of a full launch per storage type: CUs x resident workgroups per CU (occupancy query)
  std::mutex mu;
};

static int msg_ensure_device(e3_msg_plan* P) {
  std::lock_guard<std::mutex> lock(P->mu);
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess) return E3_ERR_NO_DEVICE;
  if (P->d_desc) return cur == P->device ? E3_OK : E3_ERR_INVALID_ARG;
  MsgPackDesc* d = nullptr;
  E3_HIP_CHECK(hipMalloc((void**)&d, P->desc.size() * sizeof(MsgPackDesc)));
  if (hipMemcpy(d, P->desc.data(), P->desc.size() * sizeof(MsgPackDesc), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(d);
    return E3_ERR_HIP;
  }
  const size_t lds = (size_t)(P->k->lds_tab + 4 * P->k->lds_wave) * 4 + 4 * 256;  // + edge-id buffers
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cur) != hipSuccess || cus <= 0) cus = 256;
  for (int io = 0; io < 2; ++io) {
    if (!P->k->fused[io]) continue;
    // The grid fills the chip exactly once: what the registers the compiler ended up with and the LDS image allow per CU
    // (2 workgroups for H = 32, l_max = 2; the small instantiations fit 3-4), not a number assumed at compile time.
    int per_cu = 0;
    if (hipFuncSetAttribute(P->k->fused[io], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, P->k->fused[io], 256, lds) != hipSuccess || per_cu < 1) {
      (void)hipFree(d);
      return E3_ERR_HIP;
    }
    P->grid[io] = cus * (per_cu < 4 ? per_cu : 4);
  }
  P->device = cur;
  P->d_desc = d;
  return E3_OK;
}

extern "C" {

int e3_msg_plan_create(int lmax, int hidden, e3_msg_plan** plan_out) {
  if (!plan_out || lmax < 1 || lmax > 2) return E3_ERR_INVALID_ARG;
  const MsgKernels* k = nullptr;
  for (auto& e : msg_kernels())
    if (e.lmax == lmax && 16 * e.tt == hidden) k = &e;
  if (!k) return E3_ERR_UNSUPPORTED;  // hidden in {16, 32, 64}
  auto* P = new e3_msg_plan();
  P->lmax = lmax; P->H = hidden; P->k = k;
  const int H = hidden, TT = k->tt, KS = (H + 31) / 32;
  auto ok = [&](int l1, int l2, int l3) {
    return l1 <= lmax && l2 <= lmax && l3 <= lmax && ((l1 + l2 + l3) % 2 == 0) && l3 >= std::abs(l1 - l2) && l3 <= l1 + l2;
  };
  auto T = [&](int l3) { return l3 == 0 ? TT * (1 + lmax) : TT; };
  // class-matrix row offsets: paths ordered by (l1, l2) ascending (e3_tp_plan_create), n rows each
  const int n1[3] = {2 * H + 1, 2 * H, 2 * H}, n2[3] = {H, H, H};
  int wrow1[3][3][3], wrow2[3][3][3];
  for (int l3 = 0; l3 <= 2; ++l3) {
    int r1 = 0, r2 = 0;
    for (int l1 = 0; l1 <= 2; ++l1)
      for (int l2 = 0; l2 <= 2; ++l2) {
        wrow1[l1][l2][l3] = wrow2[l1][l2][l3] = -1;
        if (l3 > lmax || !ok(l1, l2, l3)) continue;
        wrow1[l1][l2][l3] = r1; r1 += n1[l1];
        wrow2[l1][l2][l3] = r2; r2 += n2[l1];
      }
    P->K1[l3] = l3 <= lmax ? r1 : 0;
    P->K2[l3] = l3 <= lmax ? r2 : 0;
    P->M[l3] = l3 <= lmax ? 16 * T(l3) : 0;
    P->rowd[l3] = (l3 <= lmax) ? wrow1[0][l3][l3] + 2 * H : 0;
  }
  // blocks in the kernels' order: l1 outer, then l3, then l2; (ks, t) inside a path
  int b = 0;
  for (int l1 = 0; l1 <= lmax; ++l1)
    for (int l3 = 0; l3 <= lmax; ++l3)
      for (int l2 = 0; l2 <= lmax; ++l2) {
        if (!ok(l1, l2, l3)) continue;
        for (int ks = 0; ks < KS; ++ks)
          for (int t = 0; t < T(l3); ++t) {
            const int nvalid = std::min(32, H - 32 * ks);
            const int M = 16 * T(l3);
            P->desc.push_back({0, l3, wrow1[l1][l2][l3] + H + 32 * ks, nvalid, M, 16 * t, b + ks * T(l3) + t});  // src rows
            P->desc.push_back({2, l3, wrow1[l1][l2][l3] + 32 * ks, nvalid, M, 16 * t, b + ks * T(l3) + t});      // dst rows
            P->desc.push_back({1, l3, wrow2[l1][l2][l3] + 32 * ks, nvalid, M, 16 * t, b + ks * T(l3) + t});
          }
        b += KS * T(l3);
      }
  if (b != k->nblk) { delete P; return E3_ERR_INVALID_ARG; }
  *plan_out = P;
  return E3_OK;
}

int e3_msg_plan_destroy(e3_msg_plan* P) {
  if (!P) return E3_OK;
  if (P->d_desc) (void)hipFree(P->d_desc);
  delete P;
  return E3_OK;
}

int64_t e3_msg_packed_bytes(const e3_msg_plan* P) { return P ? (P->k->total_floats * 4 + 255) / 256 * 256 : -1; }
// the table row (UD floats) + one float of the per-node row maxima, stored behind the N table rows
int64_t e3_msg_premix_floats_per_node(const e3_msg_plan* P) { return P ? P->k->UD + 1 : -1; }
int e3_msg_weight_shape(const e3_msg_plan* P, int tp, int l3, int* rows, int* cols) {
  if (!P || !rows || !cols || l3 < 0 || l3 > 2 || (tp != 1 && tp != 2)) return E3_ERR_INVALID_ARG;
  *rows = tp == 1 ? P->K1[l3] : P->K2[l3];
  *cols = P->M[l3];
  return E3_OK;
}

int e3_msg_supports(const e3_msg_plan* P, int dtype) {
  if (!P) return 0;
  return dtype == E3_F32 ? 1 : (dtype == E3_BF16 && P->k->fused[1] != nullptr) ? 1 : 0;
}

int e3_msg_pack_weights(e3_msg_plan* P, const void* const w1[3], const void* const n1[3], const void* const w2[3],
                        const void* const n2[3], int dtype, void* packed, void* stream) {
  if (!P || !w1 || !w2 || !packed) return E3_ERR_INVALID_ARG;
  if (!e3_msg_supports(P, dtype)) return E3_ERR_UNSUPPORTED;
  for (int l = 0; l <= P->lmax; ++l)
    if (!w1[l] || !w2[l]) return E3_ERR_MISSING_WEIGHT;
  int st = msg_ensure_device(P);
  if (st != E3_OK) return st;
  const MsgKernels& k = *P->k;
  MsgPackArgs a;
  for (int l = 0; l < 3; ++l) {
    const bool on = l <= P->lmax;
    a.w1[l] = on ? w1[l] : nullptr; a.w2[l] = on ? w2[l] : nullptr;
    a.n1[l] = (on && n1) ? n1[l] : nullptr; a.n2[l] = (on && n2) ? n2[l] : nullptr;
    a.nw1[l] = on ? (int64_t)P->K1[l] * P->M[l] : 0;
    a.nw2[l] = on ? (int64_t)P->K2[l] * P->M[l] : 0;
    a.rowd[l] = P->rowd[l];
  }
  a.M0 = P->M[0]; a.H = P->H; a.LMAX = P->lmax; a.TT = k.tt; a.NS = k.NS; a.WD = k.WD;
  a.o_norm1 = k.o_norm1; a.o_norm2 = k.o_norm2; a.o_wd = k.o_wd; a.o_w = k.o_w; a.nblk = k.nblk;
  hipStream_t s = (hipStream_t)stream;
  E3_HIP_CHECK(hipMemsetAsync(packed, 0, 256, s));
  const dim3 grid(std::min<int>((int)P->desc.size(), 512));
  if (dtype == E3_F32) {
    hipLaunchKernelGGL(msg_absmax_kernel, dim3(64), dim3(256), 0, s, a, (uint32_t*)packed);
    hipLaunchKernelGGL(msg_pack_kernel<float>, grid, dim3(256), 0, s, a, P->d_desc, (int)P->desc.size(), (float*)packed);
  } else {
    hipLaunchKernelGGL(msg_pack_kernel<bf16>, grid, dim3(256), 0, s, a, P->d_desc, (int)P->desc.size(), (float*)packed);
  }
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

int e3_msg_premix(e3_msg_plan* P, const void* h, int64_t ld_h, int64_t N, const void* packed, const float* in_scale,
                  float* premix, int dtype, void* stream) {
  if (!P || N < 0) return E3_ERR_INVALID_ARG;
  if (!e3_msg_supports(P, dtype)) return E3_ERR_UNSUPPORTED;
  const MsgKernels& k = *P->k;
  const int io = dtype == E3_BF16 ? 1 : 0, es = io ? 2 : 4;
  if (N == 0) return E3_OK;
  if (!h || !packed || !premix || ld_h < k.D) return E3_ERR_INVALID_ARG;
  if ((ld_h * es & 15) || ((uintptr_t)h & 15) || ((uintptr_t)premix & 15)) return E3_ERR_INVALID_ARG;  // 16-byte row accesses
  int st = msg_ensure_device(P);
  if (st != E3_OK) return st;
  const int64_t ntiles = (N + 15) / 16;
  const int grid = (int)std::min<int64_t>((ntiles + 3) / 4, 2048);
  float* hmax = premix + (size_t)N * k.UD;
  void* args[] = {&h, &ld_h, &N, &packed, &in_scale, &premix, &hmax};
  if (hipLaunchKernel(k.premix[io], dim3(grid), dim3(256), args, 0, (hipStream_t)stream) != hipSuccess) return E3_ERR_HIP;
  return E3_OK;
}

int e3_msg_forward(e3_msg_plan* P, const void* h, int64_t ld_h, int64_t N, const float* pos4, const int32_t* src,
                   const int32_t* dst, int64_t E, const void* packed, const float* in_scale, const float* premix,
                   float* out, int64_t ld_out, int dtype, int accumulate, int tiles_per_block, void* stream) {
  if (!P || N < 0 || E < 0 || E > 0x7fffffffLL - 16) return E3_ERR_INVALID_ARG;  // edge ids are int32
  if (!e3_msg_supports(P, dtype)) return E3_ERR_UNSUPPORTED;
  const MsgKernels& k = *P->k;
  const int io = dtype == E3_BF16 ? 1 : 0, es = io ? 2 : 4;
  if (N == 0) return E3_OK;
  if (!h || !pos4 || !packed || !premix || !out || ld_h < k.D || ld_out < k.D || (E > 0 && (!src || !dst)))
    return E3_ERR_INVALID_ARG;
  if ((ld_h * es & 15) || ((uintptr_t)h & 15) || ((uintptr_t)premix & 15)) return E3_ERR_INVALID_ARG;  // 16-byte row gathers
  int st = msg_ensure_device(P);
  if (st != E3_OK) return st;
  hipStream_t s = (hipStream_t)stream;
  // out is an accumulation target of atomics: the rows start from zero unless this launch continues an earlier one
  if (!accumulate) E3_HIP_CHECK(hipMemset2DAsync(out, (size_t)ld_out * 4, 0, (size_t)k.D * 4, (size_t)N, s));
  if (E == 0) return E3_OK;
  // H = 32, fp32: the weights-stationary kernel (e3_msg_ws.hip); tiles_per_block > 0 = its chunk size in 16-edge units.
  // tiles_per_block < 0 asks for this file's one-wave-per-tile kernel with |tiles_per_block| tiles per wave block.
  if (tiles_per_block >= 0 && msg_ws_supported(P->lmax, P->H, dtype)) {
    st = msg_ws_launch(P->lmax, P->H, dtype, h, ld_h, N, pos4, src, dst, E, packed, in_scale, premix, out, ld_out,
                       tiles_per_block > (1 << 20) ? (1 << 24) : tiles_per_block * 16, s);
    if (st != E3_ERR_UNSUPPORTED) return st;
  }
  if (tiles_per_block < 0) tiles_per_block = -tiles_per_block;
  const int64_t ntiles = (E + 15) / 16;
  int nwg = P->grid[io];  // 4 waves per workgroup, every CU filled once
  nwg = (int)std::min<int64_t>(nwg, (ntiles + 3) / 4);
  nwg = std::max(8, (nwg + 7) / 8 * 8);
  int blk = tiles_per_block > 0 ? tiles_per_block : 4;  // default: 64 edges (2-3 dst nodes) per wave block
  if (blk > (1 << 16)) blk = 1 << 16;
  const size_t lds = (size_t)(k.lds_tab + 4 * k.lds_wave) * 4 + 4 * 256;
  void* args[] = {&h, &ld_h, &pos4, &src, &dst, &E, &packed, &premix, &in_scale, &out, &ld_out, &blk};
  if (hipLaunchKernel(k.fused[io], dim3(nwg), dim3(256), args, lds, s) != hipSuccess) return E3_ERR_HIP;
  return E3_OK;
}

#if E3_MSG_STAMP
int e3_msg_debug_stamps(unsigned long long* out8, int reset) {
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(e3::g_msg_stamps), 64) != hipSuccess) return E3_ERR_HIP;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(e3::g_msg_stamps), z, 64) != hipSuccess) return E3_ERR_HIP;
  }
  return E3_OK;
}
#endif
}  // extern "C"
