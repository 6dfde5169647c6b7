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
// does vector / copy work, so every SIMD holds one matrix-heavy and one light wave (with both teams in the same kind of phase
// the two waves of a SIMD competed for the same pipe and a step took the SUM of their issue times):
//
//   phase X(s)   waves 0-1: gathered rows of tile s -> B fragments of product #1 (wave p: rows 8 p .. 8 p + 7), harmonics
//                wave 3   : look at the ids of tile s + 1, cut the tile (<= 16 edges, <= 2 dst runs), copy its positions, row
//                           maxima and pre-mix rows (a row already staged for the previous tile is not copied again), ids of s + 2
//                team 1   : product #2 + gate of tile s - 1 -> out tile
//   phase Y(s)   team 0   : product #1 + gate of tile s -> B fragments of product #2, written directly (below)
//                waves 4-5: run sums of tile s - 1 (fp32 atomics, one per node, column and chunk)
//                waves 6-7: copies of the h[src] rows of tile s + 1 (9 x 1 KiB each)
//
// Product #1 -> product #2 without a message buffer: the gated output of product #1 sits in accumulator layout, which IS the
// B-operand layout of product #2 (k order permuted at pack time), so each wave stores its own 8-byte half of every fragment
// lane.  The fp16 (hi, lo) split needs one power-of-two scale per edge row that ALL four waves agree on without talking:
// it is derived from a rigorous bound, |message| <= Bw * max(|h[src]|, |h[dst]|, d) with Bw from the weights (pack time,
// header[6]) and the per-node row maxima the pre-mix launch leaves behind its table.  The pair (hi, lo) keeps an absolute
// error of 2^-25 of the scaled range, so a bound that is loose by many binades costs nothing (e3_tp_mfma_core.h: split2_f16).
//
// Vector-memory instructions per tile: ~27 copies (was 283); nothing in the tile loop waits for a global load.
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
__device__ unsigned long long g_ws_tp[8][8];   // per wave: cycles to the marks inside the tensor product
#define WS_TPMARK(i) { if (tpm) { const uint32_t t_ = (uint32_t)__builtin_readcyclecounter(); tpm[i] += t_ - tpm0; tpm0 = t_; } }
#else
#define WS_TPMARK(i)
#endif

// ------------------------------------------------------------------------------------------------------------------
// LDS image of the workgroup
// ------------------------------------------------------------------------------------------------------------------
template <int LMAX, int TT, bool IO16>
struct Ws {
  using G = MsgGeom<LMAX, TT>;
  static_assert(TT == 2 && (LMAX == 1 || LMAX == 2), "instantiated for H = 32, l_max = 1 / 2");
  static constexpr int H = G::H, D = G::D, NS = G::NS, UD = G::UD;
  static constexpr int ES = IO16 ? 2 : 4;                  // bytes per stored feature element
  static constexpr int NC = (LMAX + 1) * (LMAX + 1);       // components of a feature row
  static constexpr int NFR = NC + LMAX;                    // B fragments per product: one per component + one feature-first operand per degree > 0
  static constexpr int FRB = IO16 ? 1024 : 2048;           // bytes per fragment: 64 lanes x 16 B hi (+ lo)
  static constexpr int frag(int l, int a) { return l * l + a; }
  static constexpr int frag_ff(int l) { return NC + l - 1; }
  // gathered rows: region A = [1o | 2e] (8 H elements per row, one row per copy, rows padded by 16 bytes so that the 16-byte
  // column reads of the 16 rows fall on different banks), region B = [0e] rows, linear (8 or 16 rows per copy)
  static constexpr int GA_ROW = (NC - 1) * H * ES;         // [1o | 2e] (l_max = 1: [1o])
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
  static constexpr int o_hmx = o_pos + 2 * 512;                         // ring of 2 x (16 src + 16 dst) row maxima
  static constexpr int o_srow = o_hmx + 2 * 128;                        // ring of 2 x 16 row scales of product #2's operands
  // edge table (ring of 2 tiles x 16 edges x ZT floats): harmonics, distance and the dense couplings z[a][c] = sum_b C[a][b][c] Y[b]
  // of the four paths whose coupling is not a multiple of one harmonic -- computed ONCE per tile by wave 2 and read by the four
  // waves of both products (each of them used to recompute its paths' couplings: ~80 vector instructions per wave and product)
  static constexpr int z_y = 0, z_112 = 12, z_121 = 28, z_211 = 40, z_222 = 56, ZT = LMAX == 2 ? 84 : 12;
  static constexpr int o_zt = o_srow + 2 * 64;
  static constexpr int o_init = o_zt + 2 * 16 * ZT * 4;                 // initial values of the T(0) scalar-type tiles of product #1
  static constexpr int o_g = o_init + G::T(0) * 1024;                   // gather image (copied in phase Y, read in phase X)
  static constexpr int o_b1 = o_g + G_BYTES;
  static constexpr int o_b2 = o_b1 + NFR * FRB;
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
// the same copy marked non-temporal: rows that are read once (the pre-mix table) should not displace the gathered h rows,
// which every neighbouring tile of the XCD reads again, from the L2.  Measured (FETCH_SIZE, 1 M particles): 2 x 4.84 -> 2 x 4.62
// GB per launch, time unchanged.  (Plain stores instead of atomics for rows whose run lies inside one chunk were tried too:
// FETCH_SIZE and the time did not move -- the zero-filled rows are still on chip when the atomics arrive.)
#ifndef WS_UNT
#define WS_UNT 1
#endif
__device__ __forceinline__ void dma16_stream(const void* gsrc, uint32_t lds_dst) {
#if WS_UNT
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#else
  dma16(gsrc, lds_dst);
#endif
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
// 4 values = half of a fragment lane (k slots 4 t .. 4 t + 3): fp32 storage: fp16 (hi, lo) halves, 8 bytes each (lo 1 KiB
// behind hi); bf16 storage: 4 bf16, rounded once
// (s v0, s v1) -> fp16 pairs hi = rne16(s v), lo = rne16(s v - hi) in FOUR instructions: v_fma_mix{lo,hi}_f16 multiply in fp32,
// round once to fp16 and write one half of the destination (split2_f16 + a separate scale multiply compile to eight).
// s is a power of two, so s v is exact and both roundings are the ones of split2_f16.
__device__ __forceinline__ void split2_f16_scaled(const float v0, const float v1, const float s, uint32_t& hi, uint32_t& lo) {
  uint32_t h, l;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(h) : "v"(s), "v"(v0));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(h) : "v"(s), "v"(v1));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(s), "v"(v0), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(s), "v"(v1), "v"(h));
  hi = h; lo = l;
}
// f * s (s a power of two; bf16 storage: s = 1, ignored)
template <bool IO16>
__device__ __forceinline__ void ws_put4(unsigned char* dst, const float (&f)[4], const float s) {
  if constexpr (IO16) {
    const uint32_t a = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{f[0], f[1]}, bf16x2_t));
    const uint32_t b = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{f[2], f[3]}, bf16x2_t));
    *reinterpret_cast<uint2*>(dst) = uint2{a, b};
  } else {
    uint32_t h0, l0, h1, l1;
    split2_f16_scaled(f[0], f[1], s, h0, l0);
    split2_f16_scaled(f[2], f[3], s, h1, l1);
    *reinterpret_cast<uint2*>(dst) = uint2{h0, h1};
    *reinterpret_cast<uint2*>(dst + 1024) = uint2{l0, l1};
  }
}
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
#ifndef WS_XROT
#define WS_XROT 2   // measured 19.43 -> 19.00 ms per launch pair (1: 19.24, 3: 19.38)
#endif
#ifndef WS_NREG
#define WS_NREG 1
#endif
#ifndef WS_PRIO1
#define WS_PRIO1 WS_PRIO   // product #1 (phase Y: its partners have slack)
#endif
#ifndef WS_LMAX1
#define WS_LMAX1 0   // 1: also instantiate the kernel for l_max = 1 (correct, but 10.0 vs 6.4 ms per launch pair against the one-wave-per-tile kernel: the phase structure does not pay for so little work per edge)
#endif
#ifndef WS_INITMAP
#define WS_INITMAP 0
#endif
#ifndef WS_YROT
#define WS_YROT 0
#endif
#ifndef WS_PF
#define WS_PF 2
#endif
// Whole-vector forms (f32x4 fma / products -> v_pk_fma_f32 / v_pk_mul_f32 with the per-edge factor broadcast through op_sel) of
// the folds and the gate epilogues, fp32 storage only (bf16 storage measured 4 % slower with them).  bit 0: folds with table
// couplings, 1: folds with one-harmonic couplings, 2: epilogue of product #2, 3: epilogue of product #1.  Measured 19.76 ->
// 19.50 ms per launch pair: the product waves are bound by LDS reads and dependent-MFMA latency more than by vector issue.
#ifndef WS_PKT
#define WS_PKT 15
#endif
// per table coupling: bit 0 (1,1,2), 1 (1,2,1), 2 (2,1,1), 3 (2,2,2).  Path (2, 1, 1) stays in scalar form: its vector form gives
// wrong sums with this compiler (hipcc 7.2; the other three and every other use check out against the oracle) --
// tools/build_variant.sh x "-DWS_PKZ=15" e3_msg_ws + tools/exp_check.py x tests/test_msg_fused_gpu.py reproduces it.
#ifndef WS_PKZ
#define WS_PKZ 11
#endif
#ifndef WS_PRIO
#define WS_PRIO 0   // s_setprio of a product wave while it multiplies (0: none; 2 measured 20.4 vs 19.6 ms: the partner waves are near-critical too)
#endif
template <int LMAX>
struct TpItems {  // degrees in DESCENDING order: the largest fold (degree l_max) overlaps the products of the degrees after it
  static constexpr int N = (LMAX + 1) * (LMAX + 1) + LMAX;
  static constexpr int len(int l) { return 2 * l + 1 + (l > 0 ? 1 : 0); }
  static constexpr int start(int l) { int n = 0; for (int k = LMAX; k > l; --k) n += len(k); return n; }  // first item of degree l
  static constexpr int l1(int i) { int l = LMAX; while (l > 0 && i >= start(l) + len(l)) --l; return l; }
  static constexpr int a(int i) { const int l = l1(i), k = i - start(l); return k < 2 * l + 1 ? k : -1; }  // -1: feature-first operand
  static constexpr bool first(int i) { return i == start(l1(i)); }
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
  const float* zt;         // this lane's row of the edge table
  const float* urow;
  const float* wd;
  const f32x4* init;       // FIRST: initial values of the scalar-type tiles (+ lane): pre-mix of (l, l, 0), d-term
  f32x4 (&accV)[5];
  f32x4& accG;
  f32x4& accS;
  float dsc;
  // operand ring and per-degree state (everything indexed at compile time)
  uint4 bh[WS_PF + 1], bl[WS_PF + 1];
  f32x4 uV[3][3][5];       // [l1][l2][a]: mix-first accumulators of the paths into the vector tile
  f32x4 wdV;               // FIRST: d-term weights of the vector tile
  f32x4 yv[3];             // harmonics of this lane's edge (y[0..8], d xs)
  f32x4 zq[3][3][7];       // [l1][l2][.]: dense coupling of path (l1, l2, LV) as loaded from the table
#if E3_WS_STAMP
  uint32_t* tpm = nullptr;
  uint32_t tpm0 = 0;
#endif
  static constexpr int zoff(int l1, int l2) {  // table offset of path (l1, l2, LV); -1: coupling is a multiple of one harmonic
    return (l1 == 1 && l2 == 1 && LV == 2) ? L::z_112 : (l1 == 1 && l2 == 2 && LV == 1) ? L::z_121
         : (l1 == 2 && l2 == 1 && LV == 1) ? L::z_211 : (l1 == 2 && l2 == 2 && LV == 2) ? L::z_222 : -1;
  }
  __device__ __forceinline__ float yy(const int i) const { return yv[i >> 2][i & 3]; }

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
            // the path's coupling travels with its first component: consumed after its last one
            if constexpr (a == 0 && zoff(L1, L2) >= 0) {
              constexpr int NZ = ((2 * L1 + 1) * (2 * LV + 1) + 3) / 4;
#pragma unroll
              for (int q = 0; q < NZ; ++q) zq[L1][L2][q] = *reinterpret_cast<const f32x4*>(zt + zoff(L1, L2) + 4 * q);
            }
          }
        });
        if constexpr (L1 == 0 && FIRST && LV > 0) wdV = *reinterpret_cast<const f32x4*>(wd + G::wdoff(LV) + t * 16);
      }
    }
  }

  template <int I>
  __device__ __forceinline__ void compute() {
    constexpr int L1 = IT::l1(I), a = IT::a(I), slot = I % (WS_PF + 1);
    const uint4 xh = bh[slot], xl = bl[slot];
    if constexpr (a >= 0) {
      if constexpr (L1 == 0 && FIRST && LV > 0) {  // distance channel of product #1: couples through (0, l, l)
        uV[0][LV][0] = __builtin_elementwise_fma(wdV, f32x4{dsc, dsc, dsc, dsc}, uV[0][LV][0]);
      }
      sfor3([&](auto l2tag) {
        constexpr int L2 = decltype(l2tag)::value;
        if constexpr (LV > 0 && G::ok(L1, L2, LV)) {
          constexpr int i = O::widx(0, L1, L2);
          uV[L1][L2][a] = mma3<IO16>(w.h[i], w.l[i], xh, xl, uV[L1][L2][a]);
        }
      });
      if constexpr (L1 == 0) {  // path (0, 0, 0): coupling 1 -- straight into the scalar-type tiles
        if constexpr (tG >= 0) accG = mma3<IO16>(w.h[O::widx(1, 0, 0)], w.l[O::widx(1, 0, 0)], xh, xl, accG);
        if constexpr (tS >= 0) accS = mma3<IO16>(w.h[O::widx(2, 0, 0)], w.l[O::widx(2, 0, 0)], xh, xl, accS);
      }
    } else if constexpr (SC) {  // feature-first path (L1, L1, 0) into the scalar-type tiles
      if constexpr (tG >= 0) accG = mma3<IO16>(w.h[O::widx(1, L1, L1)], w.l[O::widx(1, L1, L1)], xh, xl, accG);
      if constexpr (tS >= 0) accS = mma3<IO16>(w.h[O::widx(2, L1, L1)], w.l[O::widx(2, L1, L1)], xh, xl, accS);
    }
  }

  template <int L1>
  __device__ __forceinline__ void fold() {  // ---- after the last product of degree L1: accV += sum_a z[a][c] u[a] ----
    constexpr int D1 = 2 * L1 + 1;
    sfor3([&](auto l2tag) {
      constexpr int L2 = decltype(l2tag)::value;
      if constexpr (LV > 0 && G::ok(L1, L2, LV)) {
        constexpr int D3 = 2 * LV + 1;
        if constexpr (zoff(L1, L2) >= 0) {
#pragma unroll
          for (int c = 0; c < D3; ++c)
#pragma unroll
            for (int aa = 0; aa < D1; ++aa)
              if (z_nonzero<L1, L2, LV>(aa, c)) {
                // whole-vector form: two v_pk_fma_f32 with the coupling broadcast through op_sel instead of four v_fmac
                const float z = zq[L1][L2][(aa * D3 + c) >> 2][(aa * D3 + c) & 3];
                constexpr int zi = (L1 == 1 && L2 == 1) ? 0 : (L1 == 1 && L2 == 2) ? 1 : (L1 == 2 && L2 == 1) ? 2 : 3;
                if constexpr (!IO16 && (WS_PKT & 1) && ((WS_PKZ >> zi) & 1)) {
                  accV[c] = __builtin_elementwise_fma(uV[L1][L2][aa], f32x4{z, z, z, z}, accV[c]);
                } else {
#pragma unroll
                  for (int r = 0; r < 4; ++r) accV[c][r] = __builtin_fmaf(uV[L1][L2][aa][r], z, accV[c][r]);
                }
              }
        } else {  // (0, l, l) and (l, 0, l): one harmonic times a constant
          float y9[9];
#pragma unroll
          for (int i = 0; i < 9; ++i) y9[i] = yy(i);
          float zz[D1][D3];
          make_z<L1, L2, LV>(y9, zz);
#pragma unroll
          for (int c = 0; c < D3; ++c)
#pragma unroll
            for (int aa = 0; aa < D1; ++aa)
              if (z_nonzero<L1, L2, LV>(aa, c)) {
                if constexpr (!IO16 && (WS_PKT & 2)) {
                  accV[c] = __builtin_elementwise_fma(uV[L1][L2][aa], f32x4{zz[aa][c], zz[aa][c], zz[aa][c], zz[aa][c]}, accV[c]);
                } else {
#pragma unroll
                  for (int r = 0; r < 4; ++r) accV[c][r] = __builtin_fmaf(uV[L1][L2][aa][r], zz[aa][c], accV[c][r]);
                }
              }
        }
      }
    });
  }

  // One scheduling region per item: the requests of item I + WS_PF, the products of item I and -- at the first item of a
  // degree -- the fold of the PREVIOUS degree (vector work that depends only on finished products: it fills the issue slots
  // between this degree's MFMAs).
  template <int I>
  __device__ __forceinline__ void step() {
    request<I + WS_PF>();
    __builtin_amdgcn_sched_barrier(0);
    compute<I>();
    if constexpr (IT::first(I) && IT::l1(I) < LMAX) fold<IT::l1(I) + 1>();
    if constexpr (I + 1 < IT::N && IT::first(I + 1)) { WS_TPMARK(1 + LMAX - IT::l1(I)) }
    if constexpr (I + 1 < IT::N) step<I + 1>();
    else { WS_TPMARK(1 + LMAX) fold<0>(); WS_TPMARK(2 + LMAX) }
  }

  __device__ __forceinline__ void run() {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 5; ++c) accV[c] = zero4;
    // harmonics of the edge and the initial values of the scalar-type tiles: requested with the first operands
#pragma unroll
    for (int q = 0; q < 3; ++q) yv[q] = *reinterpret_cast<const f32x4*>(zt + L::z_y + 4 * q);
    accG = zero4; accS = zero4;
    if constexpr (FIRST) {
      if constexpr (tG >= 0) accG = init[64 * tG];
      if constexpr (tS >= 0) accS = init[64 * tS];
    }
    request<0>();
    if constexpr (WS_PF >= 2) request<1>();
    if constexpr (WS_PF >= 3) request<2>();
    static_assert(WS_PF >= 1 && WS_PF <= 3, "prefetch distance");
    __builtin_amdgcn_sched_barrier(0);
    dsc = yv[2][1];
    WS_TPMARK(0)
    step<0>();
  }
};

template <int LMAX, int TT, int ROLE, bool FIRST, bool IO16>
__device__ __forceinline__ void ws_tp(const RoleW<Own<LMAX, TT, ROLE>::NW>& w, const unsigned char* bfr, const float* zt,
                                      const float* urow, const float* wd, const f32x4* init,
                                      f32x4 (&accV)[5], f32x4& accG, f32x4& accS, float (&y)[10], uint32_t* tpmarks = nullptr) {
  TpRun<LMAX, TT, ROLE, FIRST, IO16> r{w, bfr, zt, urow, wd, init, accV, accG, accS};
#if E3_WS_STAMP
  r.tpm = tpmarks;
  r.tpm0 = (uint32_t)__builtin_readcyclecounter();
#endif
  r.run();
#pragma unroll
  for (int i = 0; i < 10; ++i) y[i] = r.yy(i);  // y[9] = distance * xs
}

// ------------------------------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------------------------------
struct WsArgs {
  const void* h; int64_t ldh;
  const float4* pos4; const int32_t* src; const int32_t* dst; int64_t E;
  const float* packed; const float* U; const float* hmax; const float* in_scale; float* out; int64_t ldo;
  int chunk;  // edges per chunk (multiple of 16)
};

template <int LMAX, int TT, bool IO16, int W>
__device__ __forceinline__ void ws_run(const WsArgs& A, unsigned char* smem) {
  using G = MsgGeom<LMAX, TT>;
  using L = Ws<LMAX, TT, IO16>;
  constexpr int H = G::H, D = G::D, ES = L::ES;
  constexpr bool TEAM1 = W >= 4;
  constexpr int ROLE = W & 3;
  // service job of a team-0 wave in phase X: 0 / 1 conversion of rows 0-7 / 8-15 (+ initial values), 2 initial values + edge
  // table, 3 the tile cutter.  Rotated against the product roles (WS_XROT) so that the two conversion waves -- the longest,
  // vector-heavy jobs -- share their SIMDs with the lighter roles of product #2 (l3 = 1 tiles: waves 6, 7), not with the
  // l3 = 2 roles (waves 4, 5), which are the longest of phase X
  constexpr int SX = TEAM1 ? -1 : ((W + WS_XROT) & 3);
  // scalar-type tiles whose initial values this wave computes (two at a time): WS_INITMAP 0: two tiles on each of the jobs
  // 0-2; 1: four on the cutter's wave (the shortest job), two on the table's wave, none on the conversion waves (the longest)
  constexpr int ITB = SX < 0 ? 0 : (WS_INITMAP ? (SX == 3 ? 0 : 4) : 2 * SX);
  constexpr int ITE0 = SX < 0 ? 0 : (WS_INITMAP ? (SX == 3 ? 4 : (SX == 2 ? 6 : 4)) : (SX <= 2 ? 2 * SX + 2 : 2 * SX));
  constexpr int ITE = ITE0 < G::T(0) ? ITE0 : G::T(0);
  constexpr int SY = TEAM1 ? ((W + WS_YROT) & 3) : -1;  // phase Y jobs of team 1: 0 / 1 run sums, 2 / 3 row copies
  using O = Own<LMAX, TT, ROLE>;
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const float* n1tab = reinterpret_cast<const float*>(smem + L::o_tab);
  const float* n2tab = n1tab + G::NS * 16;
  const float* wdtab = n2tab + G::NS * 16;
  const float xs = (!IO16 && A.in_scale) ? A.in_scale[0] : 1.0f;
  // |message of product #1| <= bwx * (largest scaled input of the row): Bw (true units, header[6]) / xs
  const float bwx = A.packed[6] * ((!IO16 && A.in_scale) ? A.in_scale[1] : 1.0f);
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
  // per-column norms of this wave's output tiles: they do not change from tile to tile, so they live in registers (WS_NREG;
  // read from the LDS table in the epilogue, each read sat with its latency exposed right in front of its use: ~6 round trips
  // per product)
  f32x4 nrS = {0.f, 0.f, 0.f, 0.f}, nrG = {0.f, 0.f, 0.f, 0.f}, nrV[5];
  {
    const f32x4* ntq = reinterpret_cast<const f32x4*>(TEAM1 ? n2tab : n1tab) + g;
    if constexpr (O::tS >= 0) nrS = ntq[4 * O::tS];
#pragma unroll
    for (int c = 0; c < 5; ++c) nrV[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (O::LV > 0) {
      nrG = ntq[4 * O::tG];
#pragma unroll
      for (int c = 0; c < 2 * O::LV + 1; ++c) nrV[c] = ntq[4 * (G::slot0(O::LV) + (2 * O::LV + 1) * O::t + c)];
    }
  }

  // ---- the tile stream of this workgroup: the XCD group (blockIdx & 7) owns one contiguous eighth of the edges, cut into
  //      chunks of `chunk` edges that the group's workgroups take round-robin (they sweep one neighbourhood of the Morton
  //      order together: the gathered rows are fetched once per XCD); inside a chunk tiles are cut at run boundaries ----
  const int Ei = (int)A.E;
  const int per_xcd = (int)(gridDim.x >> 3), wg_idx = (int)(blockIdx.x >> 3);
  const int e_per_xcd = ((Ei + 7) / 8 + 15) / 16 * 16;
  const int xlo = (int)(blockIdx.x & 7) * e_per_xcd;
  const int xhi = xlo + e_per_xcd < Ei ? xlo + e_per_xcd : Ei;

  // ---- copies of the h[src] rows of a tile into the gather image (read in phase X, re-filled in phase Y): part p = rows
  //      8 p .. 8 p + 7 of region A ([1o | 2e], one full 1-KiB row per copy) and piece p of region B (8 [0e] rows).  Rows beyond
  //      the end of the tile are the src rows of the following edges (valid rows; never summed): only the ids are needed ----
  auto gather_part = [&](const int tile, const int part) {
    const int* ids = reinterpret_cast<const int*>(smem + L::o_ids) + (tile & 3) * 32;
    const uint32_t gb = lds0 + L::o_g;
    static_assert(L::GA_ROW % 16 == 0 && L::GA_ROW <= 1024 && (L::GB_ROW == 128 || L::GB_ROW == 64), "copy shapes of the image");
    constexpr int LA = L::GA_ROW / 16;          // lanes of one region-A row copy (64 fp32, 32 bf16)
    constexpr int UPR = L::GB_ROW / 16;         // 16-byte units per [0e] row (8 / 4)
    constexpr int RPC = 64 / UPR;               // [0e] rows per copy (8: one piece per part / 16: part 0 copies them all)
    // every LDS read first (the copies are asm statements with a memory clobber: a read between two of them stays there and
    // exposes its latency once per row); row ids reach the scalar unit by v_readlane
    const int myid = ids[lane & 15];
    const int ridb = ids[(RPC == 8 ? 8 * part : 0) + lane / UPR];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = 8 * part + k;
      const int rid = part == 0 ? __builtin_amdgcn_readlane(myid, k) : __builtin_amdgcn_readlane(myid, 8 + k);
      const char* rowp = hb + ((int64_t)rid * A.ldh + H) * ES;
      if (LA >= 64 || lane < LA) dma16(rowp + lane * 16, sgpr((int)(gb + r * L::GA_STRIDE)));
    }
    if (RPC == 8 || part == 0)
      dma16(hb + (int64_t)ridb * A.ldh * ES + (lane % UPR) * 16, sgpr((int)(gb + L::GA_BYTES + (RPC == 8 ? part * 1024 : 0))));
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
  // cut tile `tile` (its ids have landed), publish its descriptor, copy its positions, row maxima and pre-mix rows, request
  // the ids of the tile after it
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
    // src / dst id of lane l < 32 (rows beyond the tile repeat its last edge)
    const int myid = ids[(lane & 16) + ((lane & 15) < n ? (lane & 15) : n - 1)];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
        if (i + 1 < L::U_PIECES || lane < L::U_LAST) dma16_stream(up + i * 1024, sgpr((int)(dstb + i * 1024)));
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
    // positions and row maxima: lanes 0-15 src, 16-31 dst
    if (lane < 32) {
      dma16(A.pos4 + myid, sgpr((int)(lds0 + L::o_pos + (tile & 1) * 512)));
      dma4(A.hmax + myid, sgpr((int)(lds0 + L::o_hmx + (tile & 1) * 128)));
    }
    // next tile
    pe0 = sgpr(pe0 + n);
    if (pe0 >= pcend) chunk_start(++pci);
    issue_ids(tile + 1);
  };

  if constexpr (SX == 3) {
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

  // run-sum state of waves 4 / 5: column chunks {0, 1, 2} / {3, 4}
  constexpr int NQ = (D + 63) / 64;
  constexpr bool RSW = SY == 0 || SY == 1;
  constexpr int Q0 = !RSW ? 0 : (SY == 0 ? 0 : (NQ + 1) / 2), Q1 = !RSW ? 0 : (SY == 0 ? (NQ + 1) / 2 : NQ);
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
#if E3_WS_STAMP
  uint32_t tpmk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t st_acc[4] = {0, 0, 0, 0}, st_t = (uint32_t)__builtin_readcyclecounter();
#define WS_STAMP(i) { const uint32_t t_ = (uint32_t)__builtin_readcyclecounter(); st_acc[i] += t_ - st_t; st_t = t_; }
  // marks 5-7 of the tensor-product table: [5] phase start -> product start, [6] product end (or phase start) -> end of
  // phase X, [7] the same for phase Y
  uint32_t pm_t = 0;
#define WS_PM0 { pm_t = (uint32_t)__builtin_readcyclecounter(); }
#define WS_PM(i) { const uint32_t t_ = (uint32_t)__builtin_readcyclecounter(); tpmk[i] += t_ - pm_t; pm_t = t_; }
#else
  uint32_t* const tpmk = nullptr;
#define WS_STAMP(i)
#define WS_PM0
#define WS_PM(i)
#endif
  TileInfo t0 = tinfo(0), t1 = tinfo(-1);
  for (int s = 0;; ++s) {
    if (s >= 1 && t1.n == 0) break;
    WS_STAMP(3)
    WS_PM0
    // =========================================== phase X ===========================================
    if constexpr (SX == 3) cut(s + 1);
    if constexpr (!TEAM1) {
      // ---- gathered rows of tile s -> B fragments of product #1.  Wave p (0, 1) converts rows 8 p .. 8 p + 7 of every
      //      degree: lane = (row j8, channel group g, half pc) handles ONE piece (4 channels x all components) per degree,
      //      i.e. half of a fragment lane's 8 k slots -> 8-byte stores of the hi and the lo halves ----
      if (SX <= 1 && t0.n > 0) {
        const int j8 = lane & 7, gg = (lane >> 3) & 3, pc = lane >> 5, row = 8 * SX + j8;
        const unsigned char* gimg = smem + L::o_g;
        unsigned char* b1 = smem + L::o_b1 + (16 * gg + row) * 16 + 8 * pc;
        // all reads first: positions, then the three pieces
        const float4* pp = reinterpret_cast<const float4*>(smem + L::o_pos + (s & 1) * 512);
        const float4 ps = pp[row], pd = pp[16 + row];
        float q2[20], q1[12], q0[4];
        {
          const unsigned char* r2 = gimg + row * L::GA_STRIDE + (LMAX == 2 ? (3 * H + (16 * pc + 4 * gg) * 5) * ES : 0);
          const unsigned char* r1 = gimg + row * L::GA_STRIDE + (16 * pc + 4 * gg) * 3 * ES;
          const unsigned char* r0 = gimg + L::GA_BYTES + row * L::GB_ROW + (16 * pc + 4 * gg) * ES;
          if constexpr (!IO16) {
#pragma unroll
            for (int u = 0; u < (LMAX == 2 ? 5 : 0); ++u) { const f32x4 v = reinterpret_cast<const f32x4*>(r2)[u]; q2[4 * u] = v[0]; q2[4 * u + 1] = v[1]; q2[4 * u + 2] = v[2]; q2[4 * u + 3] = v[3]; }
#pragma unroll
            for (int u = 0; u < 3; ++u) { const f32x4 v = reinterpret_cast<const f32x4*>(r1)[u]; q1[4 * u] = v[0]; q1[4 * u + 1] = v[1]; q1[4 * u + 2] = v[2]; q1[4 * u + 3] = v[3]; }
            { const f32x4 v = reinterpret_cast<const f32x4*>(r0)[0]; q0[0] = v[0]; q0[1] = v[1]; q0[2] = v[2]; q0[3] = v[3]; }
          } else {  // 4 bf16 per 8-byte read, widened (exact)
            auto widen = [](const uint2 v, float* o) {
              o[0] = __builtin_bit_cast(float, v.x << 16); o[1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
              o[2] = __builtin_bit_cast(float, v.y << 16); o[3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
            };
#pragma unroll
            for (int u = 0; u < (LMAX == 2 ? 5 : 0); ++u) widen(reinterpret_cast<const uint2*>(r2)[u], q2 + 4 * u);
#pragma unroll
            for (int u = 0; u < 3; ++u) widen(reinterpret_cast<const uint2*>(r1)[u], q1 + 4 * u);
            widen(reinterpret_cast<const uint2*>(r0)[0], q0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        float y[9], dist;
        if constexpr (LMAX == 2) edge_sh(ps, pd, y, dist); else edge_sh1(ps, pd, y, dist);
        auto put4 = [&](const int fr, const float (&f)[4]) { ws_put4<IO16>(b1 + fr * L::FRB, f, xs); };  // (operand scale in the split)
        auto degree = [&](auto ltag, float* q) {  // q[r * D1 + a]: channel r of the piece, component a
          constexpr int L1 = decltype(ltag)::value, D1 = 2 * L1 + 1;
#pragma unroll
          for (int a = 0; a < D1; ++a) {
            const float f[4] = {q[a], q[D1 + a], q[2 * D1 + a], q[3 * D1 + a]};
            put4(L::frag(L1, a), f);
          }
          if constexpr (L1 > 0) {  // feature-first operand f[k] = sum_a z[a] x[k][a]
            float zz[D1][1];
            make_z<L1, L1, 0>(y, zz);
            float f[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float sum = zz[0][0] * q[r * D1];
#pragma unroll
              for (int a = 1; a < D1; ++a) sum = __builtin_fmaf(zz[a][0], q[r * D1 + a], sum);
              f[r] = sum;
            }
            put4(L::frag_ff(L1), f);
          }
        };
        if constexpr (LMAX == 2) degree(I2{}, q2);
        degree(I1{}, q1);
        degree(I0{}, q0);
      }
      // ---- wave 2: the edge table of tile s (harmonics, distance, dense couplings); waves 0-2: the initial values of the
      //      scalar-type tiles of product #1 (two tiles each): pre-mix of (0, 0, 0) + d-term + the pre-mix of the feature-first paths folded with their
      //      couplings -- all of it used to be recomputed / folded by the four product waves ----
      if ((ITB < ITE || SX == 2) && t0.n > 0) {
        const float4* pp = reinterpret_cast<const float4*>(smem + L::o_pos + (s & 1) * 512);
        const float4 ps = pp[j], pd = pp[16 + j];
        const int slot = (j >= t0.n0 && j < t0.n) ? t0.sl1 : t0.sl0;
        const float* urow = reinterpret_cast<const float*>(smem + L::o_u + slot * L::U_ROW) + 4 * g;
        constexpr int T0 = G::T(0);
        float y[9], dist;
        if constexpr (LMAX == 2) edge_sh(ps, pd, y, dist); else edge_sh1(ps, pd, y, dist);
        const float dsc = dist * xs;
        float z110[3][1], z220[5][1];
        make_z<1, 1, 0>(y, z110);
        if constexpr (LMAX == 2) make_z<2, 2, 0>(y, z220);
        f32x4* ip = reinterpret_cast<f32x4*>(smem + L::o_init) + lane;
        // two tiles at a time: their 20 pre-mix reads first (this wave also holds its product weights: all T(0) tiles at
        // once would need 240 registers), then the folds
        static_assert(T0 == 6 || T0 == 4, "two scalar-type tiles per wave (0, 1, 2 / 0, 1)");
#pragma unroll
        for (int t2 = ITB; t2 < ITE; t2 += 2) {
          f32x4 u0[2], u1[3][2], u2[5][2], wv[2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int tt = t2 + k;
            u0[k] = *reinterpret_cast<const f32x4*>(urow + G::uoff(0, 0, 0) + tt * 16);
            wv[k] = *reinterpret_cast<const f32x4*>(wdtab + 4 * g + G::wdoff(0) + tt * 16);
#pragma unroll
            for (int a = 0; a < 3; ++a) u1[a][k] = *reinterpret_cast<const f32x4*>(urow + G::uoff(1, 1, 0) + (a * T0 + tt) * 16);
            if constexpr (LMAX == 2) {
#pragma unroll
              for (int a = 0; a < 5; ++a) u2[a][k] = *reinterpret_cast<const f32x4*>(urow + G::uoff(2, 2, 0) + (a * T0 + tt) * 16);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            f32x4 v = u0[k];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = __builtin_fmaf(wv[k][r], dsc, v[r]);
#pragma unroll
              for (int a = 0; a < 3; ++a) v[r] = __builtin_fmaf(u1[a][k][r], z110[a][0], v[r]);
              if constexpr (LMAX == 2) {
#pragma unroll
                for (int a = 0; a < 5; ++a) v[r] = __builtin_fmaf(u2[a][k][r], z220[a][0], v[r]);
              }
            }
            ip[64 * (t2 + k)] = v;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // (wave 2) the table row of edge j: lane group 0 writes the harmonics, group k the k-th dense coupling
        if constexpr (SX == 2) {
        float* zr = reinterpret_cast<float*>(smem + L::o_zt + (s & 1) * 16 * L::ZT * 4) + j * L::ZT;
        auto store_dense = [&](auto atag, auto btag, auto ctag, float* dstp) {
          constexpr int A1 = decltype(atag)::value, B1 = decltype(btag)::value, C1 = decltype(ctag)::value;
          constexpr int Da = 2 * A1 + 1, Dc = 2 * C1 + 1, NE = Da * Dc;
          float zz[Da][Dc];
          make_z<A1, B1, C1>(y, zz);
          float flat[(NE + 3) / 4 * 4];
#pragma unroll
          for (int i = 0; i < (NE + 3) / 4 * 4; ++i) flat[i] = i < NE ? zz[i / Dc][i % Dc] : 0.f;
#pragma unroll
          for (int q = 0; q < (NE + 3) / 4; ++q)
            *reinterpret_cast<f32x4*>(dstp + 4 * q) = f32x4{flat[4 * q], flat[4 * q + 1], flat[4 * q + 2], flat[4 * q + 3]};
        };
        if (g == 0) {
          *reinterpret_cast<f32x4*>(zr + L::z_y) = f32x4{y[0], y[1], y[2], y[3]};
          *reinterpret_cast<f32x4*>(zr + L::z_y + 4) = f32x4{y[4], y[5], y[6], y[7]};
          *reinterpret_cast<f32x4*>(zr + L::z_y + 8) = f32x4{y[8], dsc, 0.f, 0.f};
          if constexpr (LMAX == 2) store_dense(I1{}, I1{}, I2{}, zr + L::z_112);
        } else if constexpr (LMAX == 2) {
          if (g == 1) {
            store_dense(I1{}, I2{}, I1{}, zr + L::z_121);
          } else if (g == 2) {
            store_dense(I2{}, I1{}, I1{}, zr + L::z_211);
          } else {
            store_dense(I2{}, I2{}, I2{}, zr + L::z_222);
          }
        }
        }
      }
    } else {
      // ---- team 1: product #2 + gate #2 of tile s - 1 -> out tile [row j][output column] ----
      if (t1.n > 0) {
        float y[10];
        const float* ztr = reinterpret_cast<const float*>(smem + L::o_zt + ((s - 1) & 1) * 16 * L::ZT * 4) + j * L::ZT;
        // 1 / (row scale): the scale is a power of two, its inverse is an exponent flip
        const uint32_t sb = reinterpret_cast<const uint32_t*>(smem + L::o_srow)[((s - 1) & 1) * 16 + j];
        const float isrow = IO16 ? 1.0f : __builtin_bit_cast(float, 0x7F000000u - sb);
        f32x4 accV[5], accG, accS;
        // (the product wave is the longest of its SIMD pair in either phase: it gets the vector issue slots first)
        __builtin_amdgcn_s_setprio(WS_PRIO);
        WS_PM(5)
        ws_tp<LMAX, TT, ROLE, false, IO16>(w, smem + L::o_b2 + lane * 16, ztr, nullptr, nullptr, nullptr, accV, accG, accS, y, tpmk);
        WS_PM0
        const f32x4* nt = reinterpret_cast<const f32x4*>(n2tab) + g;
        float* orow = reinterpret_cast<float*>(smem + L::o_o) + j * L::RS;
        if constexpr (O::tS >= 0) {
          const f32x4 nv = WS_NREG ? nrS : nt[4 * O::tS];
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
          const f32x4 gn = WS_NREG ? nrG : nt[4 * O::tG];
          f32x4 gt;
#pragma unroll
          for (int r = 0; r < 4; ++r) gt[r] = sigmoid_(accG[r] * gn[r] * isrow) * isrow;
          float o[4 * Dc];
#pragma unroll
          for (int c = 0; c < Dc; ++c) {
            const f32x4 nv = WS_NREG ? nrV[c] : nt[4 * (G::slot0(O::LV) + Dc * O::t + c)];
            f32x4 oc;
            if constexpr (!IO16 && (WS_PKT & 4)) {
              oc = accV[c] * nv * gt;  // whole-vector products: v_pk_mul_f32
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) oc[r] = accV[c][r] * nv[r] * gt[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r * Dc + c] = oc[r];
          }
          f32x4* dq = reinterpret_cast<f32x4*>(orow + G::col0(O::LV) + (16 * O::t + 4 * g) * Dc);
#pragma unroll
          for (int k = 0; k < Dc; ++k) dq[k] = f32x4{o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]};
        }
      }
    }
    if constexpr (TEAM1) __builtin_amdgcn_s_setprio(0);
    WS_PM(6)
    WS_STAMP(0)
    ws_barrier();
    WS_STAMP(1)
    WS_PM0
    // =========================================== phase Y ===========================================
    if constexpr (!TEAM1) {
      // ---- team 0: product #1 + gate #1 of tile s; the gated messages leave as the B fragments of product #2 ----
      if (t0.n > 0) {
        float y[10];
        const float* ztr = reinterpret_cast<const float*>(smem + L::o_zt + (s & 1) * 16 * L::ZT * 4) + j * L::ZT;
        const float* hm = reinterpret_cast<const float*>(smem + L::o_hmx + (s & 1) * 128);
        const float hs = hm[j], hd = hm[16 + j];
        const int slot = (j >= t0.n0 && j < t0.n) ? t0.sl1 : t0.sl0;
        const float* urow = reinterpret_cast<const float*>(smem + L::o_u + slot * L::U_ROW) + 4 * g;
        f32x4 accV[5], accG, accS;
        __builtin_amdgcn_s_setprio(WS_PRIO1);
        WS_PM(5)
        ws_tp<LMAX, TT, ROLE, true, IO16>(w, smem + L::o_b1 + lane * 16, ztr, urow, wdtab + 4 * g,
                                          reinterpret_cast<const f32x4*>(smem + L::o_init) + lane, accV, accG, accS, y, tpmk);
        WS_PM0
        const float dsc = y[9];
        // row scale of the fp16 split: bound of the row's messages -> [2^12, 2^13)  (identical in the four waves: it depends
        // on the row's inputs only)
        float srow = 1.0f;
        if constexpr (!IO16) srow = pow2_scale_from_bits(__builtin_bit_cast(uint32_t, bwx * fmaxf(fmaxf(hs, hd), dsc)), 12);
        const f32x4* nt = reinterpret_cast<const f32x4*>(n1tab) + g;  // norm slot s at nt[4 s]; carries 1 / (sw1 xs)
        // this wave's half (tile t: k slots 4 t .. 4 t + 3) of every fragment lane: 8 bytes hi, 8 bytes lo
        unsigned char* b2 = smem + L::o_b2 + lane * 16 + 8 * O::t;
        auto put4 = [&](const int fr, const float (&f)[4]) { ws_put4<IO16>(b2 + fr * L::FRB, f, srow); };
        if constexpr (O::tS >= 0) {
          const f32x4 nv = WS_NREG ? nrS : nt[4 * O::tS];
          float f[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sv = accS[r] * nv[r];
            f[r] = sv * sigmoid_(sv);
          }
          put4(L::frag(0, 0), f);
        }
        if constexpr (O::LV > 0) {
          constexpr int Dc = 2 * O::LV + 1;
          const f32x4 gn = WS_NREG ? nrG : nt[4 * O::tG];
          f32x4 gt;
#pragma unroll
          for (int r = 0; r < 4; ++r) gt[r] = sigmoid_(accG[r] * gn[r]);
          float y9[9];
#pragma unroll
          for (int i = 0; i < 9; ++i) y9[i] = y[i];
          float zz[Dc][1];
          make_z<O::LV, O::LV, 0>(y9, zz);
          f32x4 fsv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int c = 0; c < Dc; ++c) {
            const f32x4 nv = WS_NREG ? nrV[c] : nt[4 * (G::slot0(O::LV) + Dc * O::t + c)];
            f32x4 fv;
            if constexpr (!IO16 && (WS_PKT & 8)) {
              fv = accV[c] * nv * gt;  // whole-vector products / fma: v_pk_mul_f32, v_pk_fma_f32
              const f32x4 zc = {zz[c][0], zz[c][0], zz[c][0], zz[c][0]};
              fsv = c == 0 ? zc * fv : __builtin_elementwise_fma(zc, fv, fsv);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                fv[r] = accV[c][r] * nv[r] * gt[r];
                fsv[r] = c == 0 ? zz[0][0] * fv[r] : __builtin_fmaf(zz[c][0], fv[r], fsv[r]);
              }
            }
            const float f[4] = {fv[0], fv[1], fv[2], fv[3]};
            put4(L::frag(O::LV, c), f);
          }
          const float fs[4] = {fsv[0], fsv[1], fsv[2], fsv[3]};
          put4(L::frag_ff(O::LV), fs);  // feature-first operand of product #2: f[k] = sum_c z[c] m[k][c]
        }
        if constexpr (ROLE == 3 && !IO16) {
          if (lane < 16) reinterpret_cast<float*>(smem + L::o_srow)[(s & 1) * 16 + lane] = srow;
        }
      }
      if constexpr (SX == 3) ws_wait_vm0();  // the cutter's copies of phase X have landed before the barrier that publishes them
    } else if constexpr (RSW) {
      // ---- run sums of tile s - 1 (out tile rows -> at most two runs per column) ----
      if (t1.n > 0) {
        const float* op = reinterpret_cast<const float*>(smem + L::o_o) + lane;
        float s0[NQW], s1[NQW];
#pragma unroll
        for (int q = 0; q < NQW; ++q) { s0[q] = 0.f; s1[q] = 0.f; }
        float v[16][NQW];
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int q = 0; q < Q1 - Q0; ++q) v[r][q] = (64 * (Q0 + q) + lane < D) ? op[r * L::RS + 64 * (Q0 + q)] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
        if (t1.n0 == 16) {  // one run, full tile: the common case
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int q = 0; q < Q1 - Q0; ++q) s0[q] += v[r][q];
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int q = 0; q < Q1 - Q0; ++q) {
              s0[q] += r < t1.n0 ? v[r][q] : 0.f;
              s1[q] += (r >= t1.n0 && r < t1.n) ? v[r][q] : 0.f;
            }
          }
        }
        if (t1.node0 != cur) { flush(); cur = t1.node0; }
#pragma unroll
        for (int q = 0; q < NQW; ++q) carry[q] += s0[q];
        if (t1.node1 >= 0) {
          flush();
          cur = t1.node1;
#pragma unroll
          for (int q = 0; q < NQW; ++q) carry[q] = s1[q];
        }
      }
    } else {
      // ---- waves 6 / 7: copies of the h[src] rows of tile s + 1 (the image is free: phase X has read tile s); they have
      //      landed before the barrier that ends this phase ----
      if (ids_requested(s + 1)) gather_part(s + 1, SY - 2);
      ws_wait_vm0();
    }
    if constexpr (!TEAM1) __builtin_amdgcn_s_setprio(0);
    const TileInfo tn = tinfo(s + 1);  // published in phase X; read here so that its latency hides behind the barrier wait
    WS_PM(7)
    WS_STAMP(2)
    ws_barrier();
    t1 = t0; t0 = tn;
  }
  if constexpr (RSW) flush();
#if E3_WS_STAMP
  if (lane == 0) {
    for (int i = 0; i < 4; ++i) atomicAdd(&g_ws_stamps[W][i], (unsigned long long)st_acc[i]);
    for (int i = 0; i < 8; ++i) atomicAdd(&g_ws_tp[W][i], (unsigned long long)tpmk[i]);
  }
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
bool msg_ws_supported(int lmax, int hidden, int dtype) {
  return (lmax == 2 || (lmax == 1 && WS_LMAX1)) && hidden == 32 && (dtype == E3_F32 || dtype == E3_BF16);
}

int msg_ws_launch(int lmax, int hidden, int dtype, const void* h, int64_t ldh, int64_t N, const float* pos4, const int32_t* src,
                  const int32_t* dst, int64_t E, const void* packed, const float* in_scale, const float* premix, float* out,
                  int64_t ldo, int chunk_edges, hipStream_t stream) {
  if (!msg_ws_supported(lmax, hidden, dtype)) return E3_ERR_UNSUPPORTED;
  if (E > 0x7fffffffLL - 65536) return E3_ERR_UNSUPPORTED;  // 32-bit edge arithmetic with chunk head room
  const int io = dtype == E3_BF16 ? 1 : 0;
  const int li = lmax == 2 ? 1 : 0;
#if WS_LMAX1
  const void* kern = li ? (io ? (const void*)msg_ws_kernel<2, 2, true> : (const void*)msg_ws_kernel<2, 2, false>)
                        : (io ? (const void*)msg_ws_kernel<1, 2, true> : (const void*)msg_ws_kernel<1, 2, false>);
  const int lds = li ? (io ? Ws<2, 2, true>::total : Ws<2, 2, false>::total) : (io ? Ws<1, 2, true>::total : Ws<1, 2, false>::total);
  const int ud = li ? MsgGeom<2, 2>::UD : MsgGeom<1, 2>::UD;
#else
  const void* kern = io ? (const void*)msg_ws_kernel<2, 2, true> : (const void*)msg_ws_kernel<2, 2, false>;
  const int lds = io ? Ws<2, 2, true>::total : Ws<2, 2, false>::total;
  const int ud = MsgGeom<2, 2>::UD;
#endif
  int dev = 0;
  E3_HIP_CHECK(hipGetDevice(&dev));
  static std::mutex mu;
  static int cus_of[64][2][2];
  int cus = 0;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 0 || dev >= 64) return E3_ERR_INVALID_ARG;
    if (cus_of[dev][io][li] == 0) {  // once per device and storage type: the kernel needs its LDS image admitted
      E3_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      int n = 0;
      if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
      cus_of[dev][io][li] = n;
    }
    cus = cus_of[dev][io][li];
  }
  int chunk = chunk_edges > 0 ? (chunk_edges + 15) / 16 * 16 : 256;
  const int64_t nchunks = (E + chunk - 1) / chunk;
  int nwg = (int)std::min<int64_t>(cus, nchunks);  // one workgroup of 8 waves per CU
  nwg = std::max(8, (nwg + 7) / 8 * 8);
  const float* hmax = premix + (size_t)N * ud;  // per-node row maxima behind the table (e3_msg_premix)
  WsArgs a = {h, ldh, reinterpret_cast<const float4*>(pos4), src, dst, E, static_cast<const float*>(packed), premix, hmax,
              in_scale, out, ldo, chunk};
  void* args[] = {&a};
  if (hipLaunchKernel(kern, dim3(nwg), dim3(512), args, lds, stream) != hipSuccess) return E3_ERR_HIP;
  return E3_OK;
}

}  // namespace e3

#if E3_WS_STAMP
extern "C" int e3_msg_ws_debug_stamps(unsigned long long* out96, int reset) {  // [8][4] phases, then [8][8] TP marks
  if (out96 && hipMemcpyFromSymbol(out96, HIP_SYMBOL(e3::g_ws_stamps), 256) != hipSuccess) return E3_ERR_HIP;
  if (out96 && hipMemcpyFromSymbol(out96 + 32, HIP_SYMBOL(e3::g_ws_tp), 512) != hipSuccess) return E3_ERR_HIP;
  if (reset) {
    unsigned long long z[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(e3::g_ws_stamps), z, 256) != hipSuccess) return E3_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(e3::g_ws_tp), z, 512) != hipSuccess) return E3_ERR_HIP;
  }
  return E3_OK;
}
#endif
