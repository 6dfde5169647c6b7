// The SEGNN message function with the weights STATIONARY in registers (same operator, C ABI and packed-weight format as
// e3_msg_fused.hip; this file is the kernel e3_msg_forward launches for H = 32).
//
//     a_i = sum_{e: dst(e) = i}  gate( TP2( gate( TP1( [h_dst | h_src | d_e] ; Y_e ) ) ; Y_e ) )
//
// Why a second structure: in e3_msg_fused.hip every wave owns a 16-edge tile from gather to scatter and therefore streams
// ALL weight blocks of both products (140 KB) plus 106 pre-mix loads per tile through the CU's vector-memory path -- 283
// KB per 16 edges, and that path (one 1-KiB instruction per ~28 cycles per CU) is what the kernel ran at.  Here ONE
// workgroup of 8 waves per CU walks a stream of tiles and the waves split the OUTPUT tiles of the two products:
//
//   team 0 (waves 0-3): product #1 of tile s           team 1 (waves 4-7): product #2 of tile s - 1
//   role r = wave & 3 owns  r < 2: the l3 = l_max tile t = r and its gate tile;  r >= 2: the l3 = l_max - 1 tile t = r - 2,
//   its gate tile and the scalar tile t  (17 / 17 / 18 / 18 MFMA groups at l_max = 2: gates are wave-local)
//
// so a wave's 7-10 weight blocks stay in its registers for the whole launch, and everything a tile needs travels through
// LDS: the gathered h[src] rows, the two pre-mix rows of its (at most two) dst nodes and the positions arrive by LDS-DMA one
// step ahead; the MFMA B operands are converted ONCE per tile into fp16 (hi, lo) fragments that every wave of a team reads.
// A step is two phases with one workgroup barrier each, and the two teams are STAGGERED: while one team multiplies, the other
// converts, so every SIMD always holds one matrix-heavy and one vector-only wave (with both teams in the same kind of phase
// the two waves of a SIMD competed for the same pipe and a step took the SUM of their issue times: 7.6 k cycles, now ~4 k):
//
//   phase X(s)   waves 0-2: gathered rows of tile s -> B fragments of product #1, spherical harmonics
//                wave 3   : look at the ids of tile s + 1, cut the tile (<= 16 edges, <= 2 dst runs), copy its positions and
//                           pre-mix rows (a row that is already staged for the previous tile is not copied again), ids of s + 2
//                team 1   : product #2 + gate of tile s - 2 -> out tile
//   phase Y(s)   team 0   : product #1 + gate of tile s -> gated messages (double buffered)
//                team 1   : copies of the h[src] rows of tile s + 1 (4-5 per wave), gated messages of tile s - 1 -> B fragments
//                           of product #2, then the run sums of tile s - 2 (fp32 atomics, one per node, column and chunk)
//
// Vector-memory instructions per tile: ~26 copies (was 283); nothing in the tile loop waits for a global load.
//
// Tiles are cut adaptively: a tile never spans more than two dst runs (so the pre-mix rows of a tile fit two of the four
// staged row slots and the segment sum has at most two runs); with dst-sorted edges and ~24 edges per node most tiles are full.
#include "e3_common.h"
#include "cg_tables.h"
#include "e3_msg_ws.h"

#include <algorithm>

#ifndef E3_WS_STAMP
#define E3_WS_STAMP 0   // 1: per-wave cycle counters (phase A work / barrier / phase B work / barrier) -- development builds only
#endif                  //    (tools/build_variant.sh wsstamp "-DE3_WS_STAMP=1" e3_msg_ws; WS_STAMPS=1 tools/msg_micro.py prints them)

namespace e3 {
#include "e3_tp_mfma_core.h"
#include "e3_msg_common.h"

#if E3_WS_STAMP
__device__ unsigned long long g_ws_stamps[8][4];
#endif

// ------------------------------------------------------------------------------------------------------------------
// LDS image of the workgroup
// ------------------------------------------------------------------------------------------------------------------
template <int LMAX, int TT, bool IO16>
struct Ws {
  using G = MsgGeom<LMAX, TT>;
  static_assert(TT == 2 && LMAX == 2, "instantiated for H = 32, l_max = 2");
  static constexpr int H = G::H, D = G::D, NS = G::NS, UD = G::UD;
  static constexpr int ES = IO16 ? 2 : 4;                  // bytes per stored feature element
  static constexpr int NC = (LMAX + 1) * (LMAX + 1);       // components of a feature row
  static constexpr int NFR = NC + LMAX;                    // B fragments per product: one per component + one feature-first operand per degree > 0
  static constexpr int FRB = IO16 ? 1024 : 2048;           // bytes per fragment: 64 lanes x 16 B hi (+ lo)
  static constexpr int frag(int l, int a) { return l * l + a; }
  static constexpr int frag_ff(int l) { return NC + l - 1; }
  static constexpr int NM = NC * TT;                       // message slots (f32x4 per lane): degree l, tile t, component a
  static constexpr int mslot(int l, int t, int a) { return l * l * TT + (2 * l + 1) * t + a; }
  // gathered rows: region A = [1o | 2e] (8 H elements per row, one row per copy, rows padded by 16 bytes so that the 16-byte
  // column reads of the 16 rows fall on different banks), region B = [0e] rows, linear (8 or 16 rows per copy)
  static constexpr int GA_ROW = 8 * H * ES;
  static constexpr int GA_STRIDE = GA_ROW + 16;
  static constexpr int GA_BYTES = 16 * GA_STRIDE;
  static constexpr int GB_ROW = H * ES;
  static constexpr int GB_BYTES = 16 * GB_ROW;
  static constexpr int G_BYTES = GA_BYTES + GB_BYTES;
  static constexpr int RS = D + 4;                         // row stride of the out tile (floats): 16-byte aligned rows
  static constexpr int U_ROW = UD * 4;                     // bytes of one pre-mix row
  static constexpr int U_PIECES = (U_ROW + 1023) / 1024, U_LAST = (U_ROW - (U_PIECES - 1) * 1024) / 16;
  static_assert(U_ROW % 16 == 0, "pre-mix rows are copied in 16-byte units");
  // byte offsets
  static constexpr int o_tab = 0;                                       // norm1 / xs | norm2 | d-term weights (floats)
  static constexpr int o_tinfo = (o_tab + (2 * NS * 16 + G::WD) * 4 + 15) / 16 * 16;   // ring of 4 x 8 ints
  static constexpr int o_ids = o_tinfo + 4 * 32;                        // ring of 4 x (16 src + 16 dst)
  static constexpr int o_idok = o_ids + 4 * 128;                        // ring of 4 flags: the ids of tile t were requested
  static constexpr int o_pos = o_idok + 16;                             // ring of 2 x (16 src + 16 dst) float4
  static constexpr int o_y = o_pos + 2 * 512;                           // ring of 4 x 16 edges x 12 floats (y[9], d, -, -)
  static constexpr int o_pmax = o_y + 4 * 768;                          // ring of 2 x 4 roles x 16 edges
  static constexpr int o_g = o_pmax + 2 * 256;                          // gather image (copied in phase Y, read in phase X)
  static constexpr int o_b1 = o_g + G_BYTES;
  static constexpr int o_m = o_b1 + NFR * FRB;                          // ring of 2 message buffers
  static constexpr int o_b2 = o_m + 2 * NM * 1024;
  static constexpr int o_o = o_b2 + NFR * FRB;
  static constexpr int o_u = o_o + 16 * RS * 4;                         // 4 pre-mix row slots
  static constexpr int total = o_u + 4 * U_ROW;
  static_assert(total <= 160 * 1024, "LDS image exceeds the CU");
};

// tile descriptor (LDS, ring of 4): n == 0: no such tile; sl0 / sl1: pre-mix row slots of the two runs
struct TileInfo { int e0, n, n0, node0, node1, sl0, sl1; };

// ------------------------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------------------------
// LDS-DMA as inline asm: hipcc neither counts these copies nor guards later LDS accesses behind them (with the builtin it
// puts vmcnt(0) in front of every LDS access that follows a copy in flight).  The issuing wave waits for them itself
// (ws_wait_vm0) before the barrier that ends the step.  M0 = wave-uniform LDS byte address; lane l lands at +16 l (+4 l).
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void ws_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// workgroup barrier: this wave's LDS writes have completed; no vmcnt wait (copies and atomics stay in flight across it)
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class F>
__device__ __forceinline__ void sfor3(F&& f) {  // f(integral_constant<int, 0 .. 2>)
  f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{}); f(std::integral_constant<int, 2>{});
}

// ------------------------------------------------------------------------------------------------------------------
// what a role owns
// ------------------------------------------------------------------------------------------------------------------
template <int LMAX, int TT, int ROLE>
struct Own {
  using G = MsgGeom<LMAX, TT>;
  static constexpr int t = ROLE & 1;
  static constexpr int LV = ROLE < 2 ? LMAX : LMAX - 1;        // degree of the vector tile (0: none)
  static constexpr int tG = LV > 0 ? TT * LV + t : -1;         // its gate tile among the l3 = 0 tiles
  static constexpr int tS = ROLE >= 2 ? t : -1;                // scalar tile
  // tile slots: 0 = (LV, t), 1 = (0, tG), 2 = (0, tS)
  static constexpr int sl3(int s) { return s == 0 ? LV : 0; }
  static constexpr int stile(int s) { return s == 0 ? t : s == 1 ? tG : tS; }
  static constexpr bool shas(int s) { return s == 0 ? LV > 0 : stile(s) >= 0; }
  static constexpr bool has(int s, int l1, int l2) { return shas(s) && G::ok(l1, l2, sl3(s)); }
  static constexpr int widx(int s, int l1, int l2) {  // index of weight block (slot s, path (l1, l2, sl3(s)))
    int n = 0;
    for (int ss = 0; ss < 3; ++ss)
      for (int a = 0; a <= LMAX; ++a)
        for (int b = 0; b <= LMAX; ++b) {
          if (ss == s && a == l1 && b == l2) return has(ss, a, b) ? n : -1;
          if (has(ss, a, b)) ++n;
        }
    return n;
  }
  static constexpr int NW = widx(3, 0, 0);
};

template <int NW>
struct RoleW { uint4 h[NW], l[NW]; };

// the role's weight blocks -> registers (once per launch)
template <int LMAX, int TT, int ROLE, bool IO16>
__device__ __forceinline__ void ws_load_w(const float* packed, const int prod /*0: #1 src rows, 1: #2*/, const int lane,
                                          RoleW<Own<LMAX, TT, ROLE>::NW>& w) {
  using G = MsgGeom<LMAX, TT>;
  using O = Own<LMAX, TT, ROLE>;
  const unsigned char* base = reinterpret_cast<const unsigned char*>(packed + G::o_w) + (size_t)prod * G::nblk() * 2048 + lane * 16;
  auto one = [&](auto stag, auto atag, auto btag) {
    constexpr int s = decltype(stag)::value, l1 = decltype(atag)::value, l2 = decltype(btag)::value;
    if constexpr (l1 <= LMAX && l2 <= LMAX && O::has(s, l1, l2)) {
      constexpr int i = O::widx(s, l1, l2), blk = G::blk(l1, l2, O::sl3(s)) + O::stile(s);  // KS == 1
      w.h[i] = *reinterpret_cast<const uint4*>(base + (size_t)blk * 2048);
      if constexpr (!IO16) w.l[i] = *reinterpret_cast<const uint4*>(base + (size_t)blk * 2048 + 1024);
      else w.l[i] = uint4{0, 0, 0, 0};
    }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  auto slot = [&](auto stag) {
    one(stag, I0{}, I0{}); one(stag, I0{}, I1{}); one(stag, I0{}, I2{});
    one(stag, I1{}, I0{}); one(stag, I1{}, I1{}); one(stag, I1{}, I2{});
    one(stag, I2{}, I0{}); one(stag, I2{}, I1{}); one(stag, I2{}, I2{});
  };
  slot(I0{}); slot(I1{}); slot(I2{});
}

// ------------------------------------------------------------------------------------------------------------------
// one tensor product for one role.  bfr: this product's B fragments + lane * 16; urow (FIRST): this lane's pre-mix row in LDS
// + 4 g floats; wd (FIRST): d-term weights + 4 g; dsc (FIRST): distance * xs.  Results: accV (vector tile), accG (its gate
// tile), accS (scalar tile) -- raw contractions, norms applied by the gates.
//
// The B fragments are consumed as a flat list of items -- (degree 0), (degree 1: 3 components, feature-first operand),
// (degree 2: 5 components, feature-first operand) -- and item i + WS_PF is REQUESTED (its two 16-byte LDS reads, and the pre-mix
// values that initialise its accumulators) before item i computes.  hipcc's scheduler left to itself issues every LDS read
// right in front of its use (it minimises registers): ~25 exposed LDS round trips per product; the requests are therefore
// pinned with sched_barrier.  All indices are compile-time constants (operand ring = registers).
// ------------------------------------------------------------------------------------------------------------------
#ifndef WS_PF
#define WS_PF 2
#endif
template <int LMAX>
struct TpItems {
  static constexpr int N = (LMAX + 1) * (LMAX + 1) + LMAX;
  static constexpr int start(int l) { return l == 0 ? 0 : l * l + l - 1; }          // first item of degree l
  static constexpr int l1(int i) { return i < start(1) ? 0 : (LMAX < 2 || i < start(2)) ? 1 : 2; }
  static constexpr int a(int i) { const int l = l1(i), k = i - start(l); return k < 2 * l + 1 ? k : -1; }  // -1: feature-first operand
  static constexpr bool last(int i) { return i + 1 == N || l1(i + 1) != l1(i); }
};

template <int LMAX, int TT, int ROLE, bool FIRST, bool IO16>
struct TpRun {
  using G = MsgGeom<LMAX, TT>;
  using O = Own<LMAX, TT, ROLE>;
  using L = Ws<LMAX, TT, IO16>;
  using IT = TpItems<LMAX>;
  static constexpr int LV = O::LV, t = O::t, tG = O::tG, tS = O::tS;
  static constexpr bool SC = tG >= 0 || tS >= 0;  // owns scalar-type (l3 = 0) tiles

  const RoleW<O::NW>& w;
  const unsigned char* bfr;
  const float (&y)[9];
  const float* urow;
  const float* wd;
  const float dsc;
  f32x4 (&accV)[5];
  f32x4& accG;
  f32x4& accS;
  // operand ring and per-degree state (everything indexed at compile time)
  uint4 bh[WS_PF + 1], bl[WS_PF + 1];
  f32x4 uV[3][3][5];       // [l1][l2][a]: mix-first accumulators of the paths into the vector tile
  f32x4 u0G, u0S;          // path (0, 0, 0) into the scalar-type tiles
  f32x4 ffG[3][5], ffS[3][5];  // FIRST: pre-mix values of the feature-first paths (l1, l1, 0), folded after the product
  f32x4 wdV, wdG, wdS;     // FIRST: d-term weights

  template <int I>
  __device__ __forceinline__ void request() {
    if constexpr (I < IT::N) {
      constexpr int L1 = IT::l1(I), a = IT::a(I), slot = I % (WS_PF + 1);
      constexpr int fr = a >= 0 ? L::frag(L1, a) : L::frag_ff(L1);
      if constexpr (a >= 0 || SC) {
        bh[slot] = *reinterpret_cast<const uint4*>(bfr + fr * L::FRB);
        if constexpr (!IO16) bl[slot] = *reinterpret_cast<const uint4*>(bfr + fr * L::FRB + 1024);
        else bl[slot] = uint4{0, 0, 0, 0};
      }
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (a >= 0) {
        sfor3([&](auto l2tag) {
          constexpr int L2 = decltype(l2tag)::value;
          if constexpr (LV > 0 && G::ok(L1, L2, LV)) {
            if constexpr (FIRST) uV[L1][L2][a] = *reinterpret_cast<const f32x4*>(urow + G::uoff(L1, L2, LV) + (a * G::T(LV) + t) * 16);
            else uV[L1][L2][a] = zero4;
          }
        });
        if constexpr (L1 == 0) {
          u0G = zero4; u0S = zero4;
          if constexpr (FIRST) {
            if constexpr (tG >= 0) u0G = *reinterpret_cast<const f32x4*>(urow + G::uoff(0, 0, 0) + tG * 16);
            if constexpr (tS >= 0) u0S = *reinterpret_cast<const f32x4*>(urow + G::uoff(0, 0, 0) + tS * 16);
            if constexpr (LV > 0) wdV = *reinterpret_cast<const f32x4*>(wd + G::wdoff(LV) + t * 16);
            if constexpr (tG >= 0) wdG = *reinterpret_cast<const f32x4*>(wd + G::wdoff(0) + tG * 16);
            if constexpr (tS >= 0) wdS = *reinterpret_cast<const f32x4*>(wd + G::wdoff(0) + tS * 16);
          }
        }
        // the fold values of this degree's feature-first path travel with its first component: consumed a degree later
        if constexpr (FIRST && L1 > 0 && a == 0 && SC) {
#pragma unroll
          for (int c = 0; c < 2 * L1 + 1; ++c) {
            if constexpr (tG >= 0) ffG[L1][c] = *reinterpret_cast<const f32x4*>(urow + G::uoff(L1, L1, 0) + (c * G::T(0) + tG) * 16);
            if constexpr (tS >= 0) ffS[L1][c] = *reinterpret_cast<const f32x4*>(urow + G::uoff(L1, L1, 0) + (c * G::T(0) + tS) * 16);
          }
        }
      }
    }
  }

  template <int I>
  __device__ __forceinline__ void compute() {
    constexpr int L1 = IT::l1(I), a = IT::a(I), slot = I % (WS_PF + 1), D1 = 2 * L1 + 1;
    const uint4 xh = bh[slot], xl = bl[slot];
    if constexpr (a >= 0) {
      if constexpr (L1 == 0 && FIRST) {  // distance channel of product #1: couples through (0, l, l)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (LV > 0) uV[0][LV][0][r] = __builtin_fmaf(wdV[r], dsc, uV[0][LV][0][r]);
          if constexpr (tG >= 0) u0G[r] = __builtin_fmaf(wdG[r], dsc, u0G[r]);
          if constexpr (tS >= 0) u0S[r] = __builtin_fmaf(wdS[r], dsc, u0S[r]);
        }
      }
      sfor3([&](auto l2tag) {
        constexpr int L2 = decltype(l2tag)::value;
        if constexpr (LV > 0 && G::ok(L1, L2, LV)) {
          constexpr int i = O::widx(0, L1, L2);
          uV[L1][L2][a] = mma3<IO16>(w.h[i], w.l[i], xh, xl, uV[L1][L2][a]);
        }
      });
      if constexpr (L1 == 0) {
        if constexpr (tG >= 0) u0G = mma3<IO16>(w.h[O::widx(1, 0, 0)], w.l[O::widx(1, 0, 0)], xh, xl, u0G);
        if constexpr (tS >= 0) u0S = mma3<IO16>(w.h[O::widx(2, 0, 0)], w.l[O::widx(2, 0, 0)], xh, xl, u0S);
      }
    } else if constexpr (SC) {  // feature-first path (L1, L1, 0) into the scalar-type tiles
      if constexpr (tG >= 0) accG = mma3<IO16>(w.h[O::widx(1, L1, L1)], w.l[O::widx(1, L1, L1)], xh, xl, accG);
      if constexpr (tS >= 0) accS = mma3<IO16>(w.h[O::widx(2, L1, L1)], w.l[O::widx(2, L1, L1)], xh, xl, accS);
    }
    if constexpr (IT::last(I)) {  // ---- end of degree L1: folds ----
      if constexpr (L1 == 0) {
        const float z000 = (float)CG<0, 0, 0>::v[0][0][0] * y[0];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (tG >= 0) accG[r] = __builtin_fmaf(u0G[r], z000, accG[r]);
          if constexpr (tS >= 0) accS[r] = __builtin_fmaf(u0S[r], z000, accS[r]);
        }
      } else if constexpr (FIRST && SC) {
        float zz[D1][1];
        make_z<L1, L1, 0>(y, zz);
#pragma unroll
        for (int c = 0; c < D1; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (tG >= 0) accG[r] = __builtin_fmaf(ffG[L1][c][r], zz[c][0], accG[r]);
            if constexpr (tS >= 0) accS[r] = __builtin_fmaf(ffS[L1][c][r], zz[c][0], accS[r]);
          }
      }
      sfor3([&](auto l2tag) {
        constexpr int L2 = decltype(l2tag)::value;
        if constexpr (LV > 0 && G::ok(L1, L2, LV)) {
          constexpr int D3 = 2 * LV + 1;
          float zz[D1][D3];
          make_z<L1, L2, LV>(y, zz);
#pragma unroll
          for (int c = 0; c < D3; ++c)
#pragma unroll
            for (int aa = 0; aa < D1; ++aa)
              if (z_nonzero<L1, L2, LV>(aa, c)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) accV[c][r] = __builtin_fmaf(uV[L1][L2][aa][r], zz[aa][c], accV[c][r]);
              }
        }
      });
    }
  }

  template <int I>
  __device__ __forceinline__ void step() {
    request<I + WS_PF>();
    __builtin_amdgcn_sched_barrier(0);
    compute<I>();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I + 1 < IT::N) step<I + 1>();
  }

  __device__ __forceinline__ void run() {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 5; ++c) accV[c] = zero4;
    accG = zero4; accS = zero4;
    request<0>();
    if constexpr (WS_PF >= 2) request<1>();
    if constexpr (WS_PF >= 3) request<2>();
    static_assert(WS_PF >= 1 && WS_PF <= 3, "prefetch distance");
    __builtin_amdgcn_sched_barrier(0);
    step<0>();
  }
};

template <int LMAX, int TT, int ROLE, bool FIRST, bool IO16>
__device__ __forceinline__ void ws_tp(const RoleW<Own<LMAX, TT, ROLE>::NW>& w, const unsigned char* bfr, const float (&y)[9],
                                      const float* urow, const float* wd, const float dsc,
                                      f32x4 (&accV)[5], f32x4& accG, f32x4& accS) {
  TpRun<LMAX, TT, ROLE, FIRST, IO16> r{w, bfr, y, urow, wd, dsc, accV, accG, accS};
  r.run();
}

// ------------------------------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------------------------------
struct WsArgs {
  const void* h; int64_t ldh;
  const float4* pos4; const int32_t* src; const int32_t* dst; int64_t E;
  const float* packed; const float* U; const float* in_scale; float* out; int64_t ldo;
  int chunk;  // edges per chunk (multiple of 16)
};

template <int LMAX, int TT, bool IO16, int W>
__device__ __forceinline__ void ws_run(const WsArgs& A, unsigned char* smem) {
  using G = MsgGeom<LMAX, TT>;
  using L = Ws<LMAX, TT, IO16>;
  constexpr int H = G::H, D = G::D, ES = L::ES;
  constexpr bool TEAM1 = W >= 4;
  constexpr int ROLE = W & 3;
  using O = Own<LMAX, TT, ROLE>;
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const float* n1tab = reinterpret_cast<const float*>(smem + L::o_tab);
  const float* n2tab = n1tab + G::NS * 16;
  const float* wdtab = n2tab + G::NS * 16;
  const float xs = (!IO16 && A.in_scale) ? A.in_scale[0] : 1.0f;
  const char* hb = reinterpret_cast<const char*>(A.h);

  auto tinfo = [&](const int tile) -> TileInfo {  // uniform; tile < 0: none
    TileInfo ti = {0, 0, 0, -1, -1, 0, 0};
    if (tile >= 0) {
      const int4* p = reinterpret_cast<const int4*>(smem + L::o_tinfo + (tile & 3) * 32);
      const int4 a = p[0], b = p[1];
      ti.e0 = sgpr(a.x); ti.n = sgpr(a.y); ti.n0 = sgpr(a.z); ti.node0 = sgpr(a.w);
      ti.node1 = sgpr(b.x); ti.sl0 = sgpr(b.y); ti.sl1 = sgpr(b.z);
    }
    return ti;
  };

  RoleW<O::NW> w;
  ws_load_w<LMAX, TT, ROLE, IO16>(A.packed, TEAM1 ? 1 : 0, lane, w);
  // the loads have returned before the tile loop (wait_vm0 = the instruction + its form the backend's wait-count pass sees):
  // otherwise hipcc places counted vmcnt waits for them at their first uses INSIDE the loop, where the counter also holds
  // this wave's copies / atomics
  wait_vm0();

  // ---- the tile stream of this workgroup: the XCD group (blockIdx & 7) owns one contiguous eighth of the edges, cut into
  //      chunks of `chunk` edges that the group's workgroups take round-robin (they sweep one neighbourhood of the Morton
  //      order together: the gathered rows are fetched once per XCD); inside a chunk tiles are cut at run boundaries ----
  const int Ei = (int)A.E;
  const int per_xcd = (int)(gridDim.x >> 3), wg_idx = (int)(blockIdx.x >> 3);
  const int e_per_xcd = ((Ei + 7) / 8 + 15) / 16 * 16;
  const int xlo = (int)(blockIdx.x & 7) * e_per_xcd;
  const int xhi = xlo + e_per_xcd < Ei ? xlo + e_per_xcd : Ei;

  // ---- copies of the h[src] rows of a tile into the (single) gather image.  The image is cut by READER: wave 0 alone reads
  //      the 2e part of region A, wave 1 alone the 1o part and region B, so each of them re-fills its own part for tile s + 1
  //      right after its last read of tile s -- a whole phase before the barrier that publishes it.  Rows beyond the end of
  //      the tile are the src rows of the following edges (valid rows; never summed), so only the ids are needed.
  //      part 0: [2e] (5 H elements per row), part 1: [1o] (3 H elements per row) + region B ([0e], 8 or 16 rows per copy) ----
  auto gather_part = [&](const int tile, const int part) {
    const int* ids = reinterpret_cast<const int*>(smem + L::o_ids) + (tile & 3) * 32;
    const uint32_t gb = lds0 + L::o_g;
    const int off = part == 0 ? 3 * H * ES : 0, lanes = (part == 0 ? 5 : 3) * H * ES / 16;
    // every LDS read first (the copies are asm statements with a memory clobber: a read between two of them stays there and
    // exposes its latency once per row -- 200 cycles per copy measured); row ids reach the scalar unit by v_readlane
    constexpr int UPR = L::GB_ROW / 16, RPC = 64 / UPR;  // region B: units per row, rows per copy
    const int myid = ids[lane & 15];
    int ridb[16 / RPC];
#pragma unroll
    for (int it = 0; it < 16 / RPC; ++it) ridb[it] = part == 1 ? ids[it * RPC + lane / UPR] : 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rid = __builtin_amdgcn_readlane(myid, r);
      const char* rowp = hb + ((int64_t)rid * A.ldh + H) * ES + off;
      if (lane < lanes) dma16(rowp + lane * 16, sgpr((int)(gb + r * L::GA_STRIDE + off)));
    }
    if (part == 1) {
#pragma unroll
      for (int it = 0; it < 16 / RPC; ++it)
        dma16(hb + (int64_t)ridb[it] * A.ldh * ES + (lane % UPR) * 16, sgpr((int)(gb + L::GA_BYTES + it * 1024)));
    }
  };
  auto ids_requested = [&](const int tile) -> bool {
    return sgpr(reinterpret_cast<const int*>(smem + L::o_idok)[tile & 3]) != 0;
  };

  // ---- wave 3: the cutter.  State (uniform): the next tile starts at pe0 inside chunk [.., pcend); pok: it exists;
  //      pre-mix row slots: those of the previous tile (psl0, psl1) stay untouched, the last run's row may be reused ----
  int pci = 0, pe0 = 0, pcend = 0, psl0 = -1, psl1 = -1, plast_node = -1, plast_slot = 0, pnext = 0;
  bool pok = false;
  auto chunk_start = [&](const int ci) {
    pe0 = sgpr(xlo + (ci * per_xcd + wg_idx) * A.chunk);
    pok = pe0 < xhi;
    pcend = sgpr(pe0 + A.chunk < xhi ? pe0 + A.chunk : xhi);
  };
  auto issue_ids = [&](const int tile) {  // ids of the 16 edges from pe0 (clamped to the last edge) -> ids ring
    if (lane == 0) reinterpret_cast<int*>(smem + L::o_idok)[tile & 3] = pok ? 1 : 0;
    if (pok && lane < 32) {
      int e = pe0 + (lane & 15);
      e = e < Ei ? e : Ei - 1;
      const int32_t* p = (lane < 16 ? A.src : A.dst) + e;
      dma4(p, sgpr((int)(lds0 + L::o_ids + (tile & 3) * 128)));
    }
  };
  // cut tile `tile` (its ids have landed), publish its descriptor, copy its positions and pre-mix rows, request the ids of
  // the tile after it
  auto cut = [&](const int tile) {
    int* tip = reinterpret_cast<int*>(smem + L::o_tinfo) + (tile & 3) * 8;
    if (!pok) {
      if (lane < 8) tip[lane] = lane < 3 ? 0 : (lane < 5 ? -1 : 0);
      issue_ids(tile + 1);  // (publishes "not requested")
      return;
    }
    const int* ids = reinterpret_cast<const int*>(smem + L::o_ids) + (tile & 3) * 32;
    const int nmax = sgpr(pcend - pe0 < 16 ? pcend - pe0 : 16);
    const int did = ids[16 + j], dprev = ids[16 + (j > 0 ? j - 1 : 0)];
    const unsigned long long m = __ballot(lane >= 1 && lane < nmax && did != dprev);
    const unsigned long long m2 = m & (m - 1);
    const int n0 = sgpr(m ? (int)__builtin_ctzll(m) : nmax);
    const int n = sgpr(m2 ? (int)__builtin_ctzll(m2) : nmax);
    const int node0 = __builtin_amdgcn_readlane(did, 0);
    const int node1 = n0 < n ? __builtin_amdgcn_readlane(did, n0) : -1;
    // pre-mix rows: the first run often continues the previous tile's last node -- its row is staged already
    auto alloc = [&](const int avoid) {
      int sl = pnext;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (sl == psl0 || sl == psl1 || sl == avoid) sl = (sl + 1) & 3;
      pnext = (sl + 1) & 3;
      return sl;
    };
    auto urow_copy = [&](const int node, const int slot) {
      const char* up = reinterpret_cast<const char*>(A.U) + (int64_t)node * L::U_ROW + lane * 16;
      const uint32_t dstb = lds0 + L::o_u + slot * L::U_ROW;
#pragma unroll
      for (int i = 0; i < L::U_PIECES; ++i)
        if (i + 1 < L::U_PIECES || lane < L::U_LAST) dma16(up + i * 1024, sgpr((int)(dstb + i * 1024)));
    };
    int sl0, sl1 = -1;
    if (node0 == plast_node) sl0 = plast_slot;
    else { sl0 = sgpr(alloc(-1)); urow_copy(node0, sl0); }
    if (node1 >= 0) { sl1 = sgpr(alloc(sl0)); urow_copy(node1, sl1); }
    psl0 = sl0; psl1 = sl1;
    plast_node = node1 >= 0 ? node1 : node0;
    plast_slot = node1 >= 0 ? sl1 : sl0;
    if (lane == 0) {
      tip[0] = pe0; tip[1] = n; tip[2] = n0; tip[3] = node0; tip[4] = node1; tip[5] = sl0; tip[6] = sl1 >= 0 ? sl1 : sl0;
    }
    // positions: lanes 0-15 src, 16-31 dst
    if (lane < 32) {
      const int id = ids[(lane & 16) + ((lane & 15) < n ? (lane & 15) : n - 1)];
      dma16(A.pos4 + id, sgpr((int)(lds0 + L::o_pos + (tile & 1) * 512)));
    }
    // next tile
    pe0 = sgpr(pe0 + n);
    if (pe0 >= pcend) chunk_start(++pci);
    issue_ids(tile + 1);
  };

  if constexpr (W == 3) {
    chunk_start(0);
    issue_ids(0);
    ws_wait_vm0();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const bool ok0 = pok;
    cut(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ok0) { gather_part(0, 0); gather_part(0, 1); }
    ws_wait_vm0();
  }
  ws_barrier();

  // run-sum state of team 1: column chunk(s) of this wave
  constexpr int NQ = (D + 63) / 64;
  // waves 4..7 take column chunks {NQ-1}, {NQ-2}, {NQ-3}, {0 .. NQ-4}: the conversion load of waves 4-6 is larger
  constexpr int Q0 = !TEAM1 ? 0 : (ROLE == 3 ? 0 : NQ - 1 - ROLE), Q1 = !TEAM1 ? 0 : (ROLE == 3 ? NQ - 3 : NQ - ROLE);
  constexpr int NQW = Q1 - Q0 > 0 ? Q1 - Q0 : 1;
  int cur = -1;
  float carry[NQW];
#pragma unroll
  for (int q = 0; q < NQW; ++q) carry[q] = 0.f;
  auto flush = [&]() {
    if (cur >= 0) {
      float* o = A.out + (int64_t)cur * A.ldo;
#pragma unroll
      for (int q = 0; q < Q1 - Q0; ++q)
        if (64 * (Q0 + q) + lane < D) __builtin_amdgcn_global_atomic_fadd_f32(o + 64 * (Q0 + q) + lane, carry[q]);
    }
#pragma unroll
    for (int q = 0; q < NQW; ++q) carry[q] = 0.f;
  };
  auto load_y = [&](const int tile, float (&y)[9], float& dist) {
    const f32x4* yp = reinterpret_cast<const f32x4*>(smem + L::o_y + (tile & 3) * 768 + j * 48);
    const f32x4 ya = yp[0], yb = yp[1], yc = yp[2];
    y[0] = ya[0]; y[1] = ya[1]; y[2] = ya[2]; y[3] = ya[3]; y[4] = yb[0]; y[5] = yb[1]; y[6] = yb[2]; y[7] = yb[3]; y[8] = yc[0];
    dist = yc[1];
  };
  auto row_scale = [&](const int tile) -> float {  // power-of-two scale of this lane's message row (fp16 split of product #2)
    if constexpr (IO16) return 1.0f;
    const float* pm = reinterpret_cast<const float*>(smem + L::o_pmax) + (tile & 1) * 64 + j;
    const float amax = fmaxf(fmaxf(pm[0], pm[16]), fmaxf(pm[32], pm[48]));
    return pow2_scale_from_bits(__builtin_bit_cast(uint32_t, amax), 10);
  };

#if E3_WS_STAMP
  uint32_t st_acc[4] = {0, 0, 0, 0}, st_t = (uint32_t)__builtin_readcyclecounter();
#define WS_STAMP(i) { const uint32_t t_ = (uint32_t)__builtin_readcyclecounter(); st_acc[i] += t_ - st_t; st_t = t_; }
#else
#define WS_STAMP(i)
#endif
  TileInfo t0 = tinfo(0), t1 = tinfo(-1), t2 = tinfo(-1);
  for (int s = 0;; ++s) {
    if (s >= 2 && t2.n == 0) break;
    WS_STAMP(3)
    // =========================================== phase X ===========================================
    if constexpr (W == 3) {
      cut(s + 1);
    } else if constexpr (!TEAM1) {
      // ---- gathered rows of tile s -> B fragments of product #1 (wave 0: degree 2, wave 1: degrees 1 and 0) ----
      if (W <= 1 && t0.n > 0) {
        const unsigned char* gimg = smem + L::o_g;
        unsigned char* b1 = smem + L::o_b1 + lane * 16;
        auto put = [&](const int fr, const float (&f)[8]) {
          uint4 bh, bl;
          split8<IO16>(f, bh, bl);
          *reinterpret_cast<uint4*>(b1 + fr * L::FRB) = bh;
          if constexpr (!IO16) *reinterpret_cast<uint4*>(b1 + fr * L::FRB + 1024) = bl;
        };
        float y[9], dist = 0.f;
        if constexpr (W <= 1) {  // harmonics of this lane's edge (feature-first operands; wave 1 publishes them)
          const float4* pp = reinterpret_cast<const float4*>(smem + L::o_pos + (s & 1) * 512);
          const float4 ps = pp[j], pd = pp[16 + j];
          if constexpr (LMAX == 2) edge_sh(ps, pd, y, dist); else edge_sh1(ps, pd, y, dist);
        }
        auto load_deg = [&](auto ltag, auto& x) {  // x[8][D1]: channel 4 p + r of this lane's k slots, scaled
          constexpr int L1 = decltype(ltag)::value, D1 = 2 * L1 + 1;
          // first element of degree L1 in this lane's staged row: region B holds 0e, region A [1o | 2e]
          const unsigned char* xrow = L1 == 0 ? gimg + L::GA_BYTES + j * L::GB_ROW
                                              : gimg + j * L::GA_STRIDE + (L1 == 1 ? 0 : 3 * H) * ES;
#pragma unroll
          for (int p = 0; p < 2; ++p) read_piece<D1, IO16>(xrow + (16 * p + 4 * g) * D1 * ES, p, xs, x);
        };
        auto comps = [&](auto ltag, auto& x, const int a0, const int a1) {
          constexpr int L1 = decltype(ltag)::value;
#pragma unroll
          for (int a = 0; a < 2 * L1 + 1; ++a)
            if (a >= a0 && a < a1) {
              float f[8];
#pragma unroll
              for (int i = 0; i < 8; ++i) f[i] = x[i][a];
              put(L::frag(L1, a), f);
            }
        };
        auto ffop = [&](auto ltag, auto& x) {  // f[k] = sum_a z[a] x[k][a]
          constexpr int L1 = decltype(ltag)::value, D1 = 2 * L1 + 1;
          float zz[D1][1];
          make_z<L1, L1, 0>(y, zz);
          float f[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float sum = zz[0][0] * x[i][0];
#pragma unroll
            for (int a = 1; a < D1; ++a) sum = __builtin_fmaf(zz[a][0], x[i][a], sum);
            f[i] = sum;
          }
          put(L::frag_ff(L1), f);
        };
        if constexpr (W == 0) {
          float x[8][5];
          load_deg(I2{}, x);
          comps(I2{}, x, 0, 5);
          ffop(I2{}, x);
        } else if constexpr (W == 1) {
          {
            float x[8][3];
            load_deg(I1{}, x);
            comps(I1{}, x, 0, 3);
            ffop(I1{}, x);
          }
          {
            float x[8][1];
            load_deg(I0{}, x);
            comps(I0{}, x, 0, 1);
          }
          if (lane < 16) {
            f32x4* yp = reinterpret_cast<f32x4*>(smem + L::o_y + (s & 3) * 768 + lane * 48);
            yp[0] = f32x4{y[0], y[1], y[2], y[3]};
            yp[1] = f32x4{y[4], y[5], y[6], y[7]};
            yp[2] = f32x4{y[8], dist, 0.f, 0.f};
          }
        }
      }
      // this wave's part of the image has been read (the reads have returned: their values were converted above, and the
      // wait below covers a tile that was skipped): re-fill it for tile s + 1
      if constexpr (W <= 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (ids_requested(s + 1)) gather_part(s + 1, W);
      }
    } else {
      // ---- team 1: product #2 + gate #2 of tile s - 2 -> out tile [row j][output column] ----
      if (t2.n > 0) {
        float y[9], dist;
        load_y(s - 2, y, dist);
        const float isrow = 1.0f / row_scale(s - 2);
        f32x4 accV[5], accG, accS;
        ws_tp<LMAX, TT, ROLE, false, IO16>(w, smem + L::o_b2 + lane * 16, y, nullptr, nullptr, 0.f, accV, accG, accS);
        const f32x4* nt = reinterpret_cast<const f32x4*>(n2tab) + g;
        float* orow = reinterpret_cast<float*>(smem + L::o_o) + j * L::RS;
        if constexpr (O::tS >= 0) {
          const f32x4 nv = nt[4 * O::tS];
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sv = accS[r] * nv[r] * isrow;
            o[r] = sv * sigmoid_(sv);
          }
          *reinterpret_cast<f32x4*>(orow + 16 * O::t + 4 * g) = o;
        }
        if constexpr (O::LV > 0) {  // a lane's 4 channels x (2l+1) components are contiguous in the output row
          constexpr int Dc = 2 * O::LV + 1;
          const f32x4 gn = nt[4 * O::tG];
          float gt[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) gt[r] = sigmoid_(accG[r] * gn[r] * isrow) * isrow;
          float o[4 * Dc];
#pragma unroll
          for (int c = 0; c < Dc; ++c) {
            const f32x4 nv = nt[4 * (G::slot0(O::LV) + Dc * O::t + c)];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r * Dc + c] = accV[c][r] * nv[r] * gt[r];
          }
          f32x4* dq = reinterpret_cast<f32x4*>(orow + G::col0(O::LV) + (16 * O::t + 4 * g) * Dc);
#pragma unroll
          for (int k = 0; k < Dc; ++k) dq[k] = f32x4{o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]};
        }
      }
    }
    WS_STAMP(0)
    ws_barrier();
    WS_STAMP(1)
    // =========================================== phase Y ===========================================
    const TileInfo tn = tinfo(s + 1);  // published in phase X
    if constexpr (!TEAM1) {
      // ---- team 0: product #1 + gate #1 of tile s -> gated messages in accumulator layout = B layout of product #2 ----
      if (t0.n > 0) {
        float y[9], dist;
        load_y(s, y, dist);
        const int slot = (j >= t0.n0 && j < t0.n) ? t0.sl1 : t0.sl0;
        const float* urow = reinterpret_cast<const float*>(smem + L::o_u + slot * L::U_ROW) + 4 * g;
        f32x4 accV[5], accG, accS;
        ws_tp<LMAX, TT, ROLE, true, IO16>(w, smem + L::o_b1 + lane * 16, y, urow, wdtab + 4 * g, dist * xs, accV, accG, accS);
        const f32x4* nt = reinterpret_cast<const f32x4*>(n1tab) + g;  // norm slot s at nt[4 s]; carries 1 / (sw1 xs)
        f32x4* mp = reinterpret_cast<f32x4*>(smem + L::o_m + (s & 1) * L::NM * 1024) + lane;
        float amax = 0.f;
        if constexpr (O::tS >= 0) {
          const f32x4 nv = nt[4 * O::tS];
          f32x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sv = accS[r] * nv[r];
            o[r] = sv * sigmoid_(sv);
            amax = fmaxf(amax, fabsf(o[r]));
          }
          mp[64 * L::mslot(0, O::t, 0)] = o;
        }
        if constexpr (O::LV > 0) {
          constexpr int Dc = 2 * O::LV + 1;
          const f32x4 gn = nt[4 * O::tG];
          float gt[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) gt[r] = sigmoid_(accG[r] * gn[r]);
#pragma unroll
          for (int c = 0; c < Dc; ++c) {
            const f32x4 nv = nt[4 * (G::slot0(O::LV) + Dc * O::t + c)];
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              o[r] = accV[c][r] * nv[r] * gt[r];
              amax = fmaxf(amax, fabsf(o[r]));
            }
            mp[64 * L::mslot(O::LV, O::t, c)] = o;
          }
        }
        if constexpr (!IO16) {  // row maximum of this role's part (joined by the readers)
          amax = fmaxf(amax, __shfl_xor(amax, 16));
          amax = fmaxf(amax, __shfl_xor(amax, 32));
          if (lane < 16) reinterpret_cast<float*>(smem + L::o_pmax)[(s & 1) * 64 + ROLE * 16 + lane] = amax;
        }
      }
      if constexpr (W != 2) ws_wait_vm0();  // the copies of phase X have landed before the barrier that publishes them
    } else {
      // ---- team 1: gated messages of tile s - 1 -> B fragments of product #2 ----
      if (t1.n > 0) {
        const f32x4* mp = reinterpret_cast<const f32x4*>(smem + L::o_m + ((s - 1) & 1) * L::NM * 1024) + lane;
        unsigned char* b2 = smem + L::o_b2 + lane * 16;
        const float srow = row_scale(s - 1);
        auto put = [&](const int fr, const float (&f)[8]) {
          uint4 bh, bl;
          split8<IO16>(f, bh, bl);
          *reinterpret_cast<uint4*>(b2 + fr * L::FRB) = bh;
          if constexpr (!IO16) *reinterpret_cast<uint4*>(b2 + fr * L::FRB + 1024) = bl;
        };
        auto getc = [&](const int l, const int a, float (&f)[8]) {  // k slot jj = 4 t + r  <-  slot (l, t, a), element r
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) {
            const f32x4 v = mp[64 * L::mslot(l, tt, a)];
#pragma unroll
            for (int r = 0; r < 4; ++r) f[4 * tt + r] = v[r] * srow;
          }
        };
        float y[9], dist;
        if constexpr (ROLE == 1 || ROLE == 2) load_y(s - 1, y, dist);
        auto conv = [&](auto ltag, const int a0, const int a1, const bool ff) {
          constexpr int L1 = decltype(ltag)::value, D1 = 2 * L1 + 1;
          float fsum[8];
          float zz[D1][1];
          if (ff) make_z<L1, L1, 0>(y, zz);
#pragma unroll
          for (int a = 0; a < D1; ++a) {
            if (!(ff || (a >= a0 && a < a1))) continue;
            float f[8];
            getc(L1, a, f);
            if (a >= a0 && a < a1) put(L::frag(L1, a), f);
            if (ff) {
#pragma unroll
              for (int i = 0; i < 8; ++i) fsum[i] = a == 0 ? zz[0][0] * f[i] : __builtin_fmaf(zz[a][0], f[i], fsum[i]);
            }
          }
          if (ff) put(L::frag_ff(L1), fsum);
        };
        if constexpr (ROLE == 0) conv(I2{}, 0, 3, false);
        else if constexpr (ROLE == 1) conv(I2{}, 3, 5, true);
        else if constexpr (ROLE == 2) conv(I1{}, 0, 3, true);
        else conv(I0{}, 0, 1, false);
      }
      // ---- run sums of tile s - 2 (out tile rows -> at most two runs per column) ----
      if (t2.n > 0) {
        const float* op = reinterpret_cast<const float*>(smem + L::o_o) + lane;
        float s0[NQW], s1[NQW];
#pragma unroll
        for (int q = 0; q < NQW; ++q) { s0[q] = 0.f; s1[q] = 0.f; }
        if (t2.n0 == 16) {  // one run, full tile: the common case
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int q = 0; q < Q1 - Q0; ++q)
              s0[q] += (64 * (Q0 + q) + lane < D) ? op[r * L::RS + 64 * (Q0 + q)] : 0.f;
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int q = 0; q < Q1 - Q0; ++q) {
              const float v = (64 * (Q0 + q) + lane < D) ? op[r * L::RS + 64 * (Q0 + q)] : 0.f;
              s0[q] += r < t2.n0 ? v : 0.f;
              s1[q] += (r >= t2.n0 && r < t2.n) ? v : 0.f;
            }
          }
        }
        if (t2.node0 != cur) { flush(); cur = t2.node0; }
#pragma unroll
        for (int q = 0; q < NQW; ++q) carry[q] += s0[q];
        if (t2.node1 >= 0) {
          flush();
          cur = t2.node1;
#pragma unroll
          for (int q = 0; q < NQW; ++q) carry[q] = s1[q];
        }
      }
    }
    WS_STAMP(2)
    ws_barrier();
    t2 = t1; t1 = t0; t0 = tn;
  }
  if constexpr (TEAM1) flush();
#if E3_WS_STAMP
  if (lane == 0)
    for (int i = 0; i < 4; ++i) atomicAdd(&g_ws_stamps[W][i], (unsigned long long)st_acc[i]);
#endif
#undef WS_STAMP
}

template <int LMAX, int TT, bool IO16>
// (waves per SIMD fixed from both sides: the LDS image admits one workgroup per CU = 2 waves per SIMD; with the minimum alone
// hipcc schedules for a third wave -- it held the kernel at 167 registers by issuing every LDS read right in front of its use)
__global__ __launch_bounds__(512, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void msg_ws_kernel(const WsArgs A) {
  using G = MsgGeom<LMAX, TT>;
  using L = Ws<LMAX, TT, IO16>;
  extern __shared__ __align__(16) unsigned char ws_smem[];
  {
    float* n1tab = reinterpret_cast<float*>(ws_smem + L::o_tab);
    float* n2tab = n1tab + G::NS * 16;
    float* wdtab = n2tab + G::NS * 16;
    const float ixs = (!IO16 && A.in_scale) ? A.in_scale[1] : 1.0f;
    for (int i = threadIdx.x; i < G::NS * 16; i += blockDim.x) {
      n1tab[i] = A.packed[G::o_norm1 + i] * ixs;
      n2tab[i] = A.packed[G::o_norm2 + i];
    }
    for (int i = threadIdx.x; i < G::WD; i += blockDim.x) wdtab[i] = A.packed[G::o_wd + i];
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  switch (wave) {
    case 0: ws_run<LMAX, TT, IO16, 0>(A, ws_smem); break;
    case 1: ws_run<LMAX, TT, IO16, 1>(A, ws_smem); break;
    case 2: ws_run<LMAX, TT, IO16, 2>(A, ws_smem); break;
    case 3: ws_run<LMAX, TT, IO16, 3>(A, ws_smem); break;
    case 4: ws_run<LMAX, TT, IO16, 4>(A, ws_smem); break;
    case 5: ws_run<LMAX, TT, IO16, 5>(A, ws_smem); break;
    case 6: ws_run<LMAX, TT, IO16, 6>(A, ws_smem); break;
    default: ws_run<LMAX, TT, IO16, 7>(A, ws_smem); break;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------------------------
bool msg_ws_supported(int lmax, int hidden, int dtype) { return lmax == 2 && hidden == 32 && (dtype == E3_F32); }

int msg_ws_launch(int lmax, int hidden, int dtype, const void* h, int64_t ldh, const float* pos4, const int32_t* src,
                  const int32_t* dst, int64_t E, const void* packed, const float* in_scale, const float* premix, float* out,
                  int64_t ldo, int chunk_edges, hipStream_t stream) {
  if (!msg_ws_supported(lmax, hidden, dtype)) return E3_ERR_UNSUPPORTED;
  using L = Ws<2, 2, false>;
  if (E > 0x7fffffffLL - 65536) return E3_ERR_UNSUPPORTED;  // 32-bit edge arithmetic with chunk head room
  int dev = 0;
  E3_HIP_CHECK(hipGetDevice(&dev));
  static std::mutex mu;
  static int cus_of[64];
  int cus = 0;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64) return E3_ERR_INVALID_ARG;
    if (cus_of[dev] == 0) {  // once per device: the kernel needs its LDS image admitted
      E3_HIP_CHECK(hipFuncSetAttribute((const void*)msg_ws_kernel<2, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, L::total));
      int n = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
      cus_of[dev] = n;
    }
    cus = cus_of[dev];
  }
  int chunk = chunk_edges > 0 ? (chunk_edges + 15) / 16 * 16 : 256;
  const int64_t nchunks = (E + chunk - 1) / chunk;
  int nwg = (int)std::min<int64_t>(cus, nchunks);  // one workgroup of 8 waves per CU
  nwg = std::max(8, (nwg + 7) / 8 * 8);
  WsArgs a = {h, ldh, reinterpret_cast<const float4*>(pos4), src, dst, E, static_cast<const float*>(packed), premix, in_scale,
              out, ldo, chunk};
  hipLaunchKernelGGL((msg_ws_kernel<2, 2, false>), dim3(nwg), dim3(512), L::total, stream, a);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

}  // namespace e3

#if E3_WS_STAMP
extern "C" int e3_msg_ws_debug_stamps(unsigned long long* out32, int reset) {
  if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(e3::g_ws_stamps), 256) != hipSuccess) return E3_ERR_HIP;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(e3::g_ws_stamps), z, 256) != hipSuccess) return E3_ERR_HIP;
  }
  return E3_OK;
}
#endif
