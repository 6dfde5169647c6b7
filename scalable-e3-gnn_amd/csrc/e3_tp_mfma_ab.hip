// Two-waves-per-tile variant of the MFMA tensor-product kernel (see e3_tp_mfma.hip for the one-wave kernel and the
// operand / staging / epilogue design; this file only changes who does what).
//
// Why: the one-wave kernel keeps all output accumulators of a 32-row tile in one wave (176 registers for the gated
// l_max = 2 message products), which limits the CU to one wave per SIMD.  Measured there: ~8 cycles per issued
// instruction, 41 % of the wave cycles inside s_waitcnt, MFMA pipe busy 14 % -- the kernel is bound by single-wave
// issue, not by MFMA, LDS or HBM.  Here a workgroup of TWO waves owns a tile and shares its staged chunk:
//   wave A: output degrees below the top one (for the gated products: silu scalars, gates, the 1o block)
//   wave B: the top degree (2e); fp32 storage: its gates arrive from wave A through LDS in the epilogue, bf16 storage:
//           B accumulates the scalar tile with its gates itself (see RoleSplit)
// Each wave issues half of the copies and stores its own columns, so the LDS per tile is unchanged, the registers per
// wave halve, and 4 workgroups = 8 waves = 2 per SIMD are resident on a CU.
//
// Instantiated for the l_max = 2 tensor products of the SEGNN forward, bf16-pipe operand modes only (MODE 1 = fp32
// storage with bf16x3-split operands, MODE 2 = bf16 storage); weights come from L2 (four resident workgroups cannot
// each hold a weight image in LDS).  E3_TP_AB=0 falls back to the one-wave kernel.
#include "e3_common.h"
#include "cg_tables.h"
#include "e3_tp_internal.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace e3 {

#include "e3_tp_mfma_core.h"

// tiles [first, first + count) of output degree l3 that `role` (0 = A, 1 = B) accumulates: B owns the top degree, A
// everything below it -- including, for gated products, the scalar tile with the top block's gates, which A hands to
// B through LDS in the epilogue (B then builds no scalar features at all; measured VALU load was 1.6x heavier on B
// when it also carried that tile)
// AG ("A holds the gates"): used for fp32 storage.  For bf16 storage (cheaper feature builds, other balance) it
// measured 9 % slower than letting B accumulate its own gate tile, so there B owns scalar tile NT0-1 as well.
constexpr int kChunkAb16 = 32 * (16 * 3 + 4) + 32 * (16 * 5 + 4);  // 4352 dwords >= kChunk16

template <int NT0, int NT1, int NT2, bool GATE, bool AG>
struct RoleSplit {
  static constexpr int top = NT2 > 0 ? 2 : (NT1 > 0 ? 1 : 0);
  static constexpr int nt(int l3) { return l3 == 0 ? NT0 : l3 == 1 ? NT1 : NT2; }
  static constexpr int first(int role, int l3) { return (!AG && role == 1 && l3 == 0 && GATE) ? NT0 - 1 : 0; }
  static constexpr int count(int role, int l3) {
    if (l3 == top) return role == 1 ? nt(l3) : 0;
    if (l3 == 0 && GATE && !AG) return role == 1 ? 1 : NT0 - 1;
    return role == 0 ? nt(l3) : 0;
  }
};

// SCAT: rows are not stored but summed per node id (segs.scatter, ascending) into the FP32 array out[node] (also for
// bf16 storage: the caller rounds the sums once) with fp32 atomics --
// the segment-sum of the message pass fused into the epilogue (the [E, width] messages never reach HBM)
template <int LSH, int NT0, int NT1, int NT2, bool GATE, int MODE, bool SCAT, int... L1S>
__global__ __launch_bounds__(128, 2) void tp_fwd_mfma_ab_kernel(SegArgs segs, const float* __restrict__ in2, int64_t ld2,
                                                                 const float* __restrict__ packed, void* __restrict__ outv,
                                                                 int64_t ldo, int64_t B, const FDev* __restrict__ dp,
                                                                 const FChunk* __restrict__ chunks,
                                                                 const int32_t* __restrict__ ocol_tab) {
  static_assert(MODE == 1 || MODE == 2, "bf16-pipe modes only");
  constexpr bool IO16 = MODE == 2;
  // the chunk buffer doubles as the epilogue's transposition space: two passes of 16 channels for both waves need 4352
  // dwords, more than a bf16 chunk (2688) -- the buffer is sized for the larger of the two
  constexpr int CHUNK = IO16 ? kChunkAb16 : kChunkFloats;
  constexpr bool AG = GATE && MODE == 1;
  using RS = RoleSplit<NT0, NT1, NT2, GATE, AG>;
  static_assert(RS::top >= 1, "needs two output degrees");
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, half = lane >> 5;
  const int Dout = dp->Dout, Dy = dp->Dy, wtotal = dp->wtotal, nchunks = dp->nchunks;
  int cM[3], cMpad[3], cOoff[3], cBfoff[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { cM[c] = dp->M[c]; cMpad[c] = dp->Mpad[c]; cOoff[c] = dp->ooff[c]; cBfoff[c] = dp->bfoff[c]; }
  const int ntab = dp->ntab;
  const int bftotal = dp->bftotal;
  const int dbg = dp->dbg;

  // LDS: [normcol (+4 ones) | ocol | chunk buffer | Y tile]
  float* nrm = lds;
  int* ocl = reinterpret_cast<int*>(nrm + ((Dout + 4 + 15) & ~15));
  float* cbuf = reinterpret_cast<float*>(ocl + ((ntab + 15) & ~15));
  float* ybuf = cbuf + CHUNK;
  for (int i = tid; i < Dout; i += blockDim.x) nrm[i] = packed[wtotal + i];
  if (tid < 4) nrm[Dout + tid] = 1.f;
  for (int i = tid; i < ntab; i += blockDim.x) ocl[i] = ocol_tab[i];
  __syncthreads();
  const float* wglob = packed + wtotal + ((Dout + 3) & ~3);
  const uint4* whi_base = reinterpret_cast<const uint4*>(wglob);
  const uint4* wlo_base = reinterpret_cast<const uint4*>(wglob + (bftotal >> 1));

  const int64_t ntiles = (B + 31) / 32;
  const int64_t tstride = gridDim.x;
  unsigned long long* const prof = dp->prof;
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
  auto tick = [&](int phase) {
    if (prof) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      tacc[phase] += now - tlast;
      tlast = now;
    }
  };
  if (prof) tlast = __builtin_amdgcn_s_memtime();

  // row ids of the gathered segments (both waves need all 32: each issues copies for rows of both halves)
  int mc0 = 0, mc1 = 0, mc2 = 0, mc3 = 0, mn0 = 0, mn1 = 0, mn2 = 0, mn3 = 0;
  auto fetch_ids = [&](int64_t t) {
    const int64_t r = t * 32 + j;
    if (r < B) {
      if (segs.nseg > 0 && segs.index[0]) mn0 = segs.index[0][r];
      if (segs.nseg > 1 && segs.index[1]) mn1 = segs.index[1][r];
      if (segs.nseg > 2 && segs.index[2]) mn2 = segs.index[2][r];
      if (segs.nseg > 3 && segs.index[3]) mn3 = segs.index[3][r];
    }
  };
  if ((int64_t)blockIdx.x < ntiles) fetch_ids(blockIdx.x);
  const int inv_dy = (65536 + Dy - 1) / Dy;

  // everything a wave does for its role; instantiated twice, each wave runs one copy
  auto body = [&](auto roletag) {
    constexpr int ROLE = decltype(roletag)::value;
    constexpr int N0 = RS::count(ROLE, 0), N1 = RS::count(ROLE, 1), N2 = RS::count(ROLE, 2);
    constexpr int F0 = RS::first(ROLE, 0), F1 = RS::first(ROLE, 1), F2 = RS::first(ROLE, 2);
    using Slots = PathSlots<LSH, N0, N1, N2>;
    using Seq = IntSeq<L1S...>;

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += tstride) {
      const int64_t row0 = tile * 32;
      const int nrows = (int)((B - row0) < 32 ? (B - row0) : 32);
      mc0 = mn0; mc1 = mn1; mc2 = mn2; mc3 = mn3;
      if (tile + tstride < ntiles) fetch_ids(tile + tstride);
      int sd = -1;  // SCAT: node id of this lane's row (lane & 31), -1 beyond the batch
      if constexpr (SCAT) {
        if (j < nrows) sd = segs.scatter[row0 + j];
      }

      // stage this wave's half of chunk ci (batches of rows alternate between the two waves)
      auto stage = [&](int ci, float* dst) {
        const FChunk ch = chunks[ci];
        const int s = (segs.nseg > 1 && ch.col >= segs.col0[1]) + (segs.nseg > 2 && ch.col >= segs.col0[2]) +
                      (segs.nseg > 3 && ch.col >= segs.col0[3]);
        auto pick = [&](auto v0, auto v1, auto v2, auto v3) {
          auto v = v0;
          v = s == 1 ? v1 : v;
          v = s == 2 ? v2 : v;
          v = s == 3 ? v3 : v;
          return v;
        };
        const int64_t ld = pick(segs.ld[0], segs.ld[1], segs.ld[2], segs.ld[3]);
        const int32_t* idx = pick(segs.index[0], segs.index[1], segs.index[2], segs.index[3]);
        const void* segbase = pick(segs.base[0], segs.base[1], segs.base[2], segs.base[3]);
        const int segcol = ch.col - pick(segs.col0[0], segs.col0[1], segs.col0[2], segs.col0[3]);
        const int cw = ch.count * (2 * ch.l1 + 1);
        const int mg = pick(mc0, mc1, mc2, mc3);
        const int mr = idx ? mg : (int)row0 + j;
        constexpr int ESZ = IO16 ? 2 : 4, EPU = 16 / ESZ, MI = IO16 ? 1 : 0;
        const int cwp = ((ch.count + 15) & ~15) * (2 * ch.l1 + 1);
        const int upr = cw / EPU, S = ch.S[MI];
        const char* base = reinterpret_cast<const char*>(segbase);
        const bool wide = (cw % EPU == 0) && (segcol % EPU == 0) && (ld % EPU == 0) &&
                          ((reinterpret_cast<uintptr_t>(base) & 15) == 0);
        if (wide) {
          const int rows_per = ch.rows_per[MI];
          const int rl = (lane * ch.inv[MI]) >> 16, u = lane - rl * S;
          const bool lane_ok = rl < rows_per && u < upr;
          const char* lsrc = base + (int64_t)segcol * ESZ + u * 16;
          const uint32_t ldb = (uint32_t)(ld * ESZ);
          for (int b0 = wave; b0 * rows_per < 32; b0 += 8) {  // this wave's batches: b0, b0+2, b0+4, b0+6
            int ridx[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) ridx[k] = __shfl(mr, ((b0 + 2 * k) * rows_per + rl) & 31);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int rbase = (b0 + 2 * k) * rows_per;
              if (rbase < 32 && lane_ok && rbase + rl < nrows)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(lsrc + (uint64_t)(uint32_t)ridx[k] * ldb),
                                                 (lds_void_t*)(dst + rbase * S * 4), 16, 0, 0);
            }
          }
          if (cwp > cw || nrows < 32) {
            uint16_t* d16 = reinterpret_cast<uint16_t*>(dst);
            for (int r = wave; r < 32; r += 2) {
              const int e0 = (r < nrows) ? cw : 0;
              for (int e = e0 + lane; e < cwp; e += 64) {
                if (IO16) d16[r * S * 8 + e] = 0;
                else dst[r * S * 4 + e] = 0.f;
              }
            }
          }
        } else if (wave == 0) {
          const bool rok = j < nrows;
          if (IO16) {
            uint16_t* drow = reinterpret_cast<uint16_t*>(dst) + j * S * 8;
            const uint16_t* srow = reinterpret_cast<const uint16_t*>(base) + (int64_t)mr * ld + segcol;
            for (int e = half; e < cwp; e += 2) drow[e] = (rok && e < cw) ? srow[e] : (uint16_t)0;
          } else {
            float* drow = dst + j * S * 4;
            const float* srow = reinterpret_cast<const float*>(base) + (int64_t)mr * ld + segcol;
            for (int e = half; e < cwp; e += 2) drow[e] = (rok && e < cw) ? srow[e] : 0.f;
          }
        }
      };

      __syncthreads();  // the previous tile's epilogue has released the chunk buffer
      for (int h = wave; h * 64 < 32 * Dy; h += 2) {
        const int e = h * 64 + lane;
        const int yr = (e * inv_dy) >> 16, yc = e - yr * Dy;
        if (e < 32 * Dy) {
          if (yr < nrows)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(in2 + (row0 + yr) * ld2 + yc), (lds_void_t*)(ybuf + h * 64), 4,
                                             0, 0);
          else
            ybuf[e] = 0.f;
        }
      }
      stage(0, cbuf);
      tick(0);

      f32x16 a0[N0 > 0 ? N0 : 1][1], a1[N1 > 0 ? N1 : 1][3], a2[N2 > 0 ? N2 : 1][5];
#pragma unroll
      for (int t = 0; t < (N0 > 0 ? N0 : 1); ++t) a0[t][0] = f32x16{0};
#pragma unroll
      for (int t = 0; t < (N1 > 0 ? N1 : 1); ++t)
#pragma unroll
        for (int c = 0; c < 3; ++c) a1[t][c] = f32x16{0};
#pragma unroll
      for (int t = 0; t < (N2 > 0 ? N2 : 1); ++t)
#pragma unroll
        for (int c = 0; c < 5; ++c) a2[t][c] = f32x16{0};

      float y[9];
      int ci = 0;
      constexpr bool PRELOAD = IO16;  // fp32 storage: the 24 preload registers cost more (spills) than the exposed round trips
      constexpr int PWN = Slots::max_total();
      uint4 pwh[PWN], pwl[IO16 ? 1 : PWN];
      auto preload = [&](auto l1tag, int cidx) {
        constexpr int L1n = decltype(l1tag)::value;
        if constexpr (L1n >= 0 && PRELOAD) {
          const FChunk chn = chunks[cidx];
#define E3_PRE(L2v, L3v, NTv, T0v)                                                                              \
  if constexpr (Slots::valid(L1n, L2v, L3v)) {                                                                  \
    const size_t o = (size_t)(cBfoff[L3v] >> 3) + (size_t)(2 * chn.wblk[L2v][L3v] + half) * cMpad[L3v] + j +    \
                     32 * T0v;                                                                                  \
    constexpr int s0 = Slots::slot(L1n, L2v, L3v);                                                              \
    _Pragma("unroll") for (int t = 0; t < NTv; ++t) {                                                           \
      pwh[s0 + t] = whi_base[o + 32 * t];                                                                       \
      if constexpr (!IO16) pwl[s0 + t] = wlo_base[o + 32 * t];                                                  \
    }                                                                                                           \
  }
          E3_PRE(0, 0, N0, F0) E3_PRE(1, 0, N0, F0) E3_PRE(2, 0, N0, F0)
          E3_PRE(0, 1, N1, F1) E3_PRE(1, 1, N1, F1) E3_PRE(2, 1, N1, F1)
          E3_PRE(0, 2, N2, F2) E3_PRE(1, 2, N2, F2) E3_PRE(2, 2, N2, F2)
#undef E3_PRE
        }
      };
      preload(std::integral_constant<int, Seq::at(0)>{}, 0);
      auto process = [&](auto itag) {
        constexpr int L1 = Seq::at(decltype(itag)::value);
        constexpr int L1N = Seq::at(decltype(itag)::value + 1);
        wait_vm0();
        __syncthreads();  // both halves of the chunk (and of the Y tile) have landed
        tick(1);
        if (ci == 0) {
#pragma unroll
          for (int q = 0; q < 9; ++q) y[q] = (q < Dy) ? ybuf[j * Dy + q] : 0.f;
        }
        const FChunk ch = chunks[ci];
        const int cwp = ((ch.count + 15) & ~15) * (2 * L1 + 1);
        const float* xr = cbuf + j * (((cwp / (IO16 ? 8 : 4)) | 1) * 4);
#define E3_RUN(L2v, L3v, ACC, NTv, T0v)                                                                          \
  if constexpr (Slots::valid(L1, L2v, L3v)) {                                                                    \
    static_assert(CG<L1, L2v, L3v>::valid, "path bookkeeping");                                                  \
    const size_t o = (size_t)(cBfoff[L3v] >> 3) + (size_t)(2 * ch.wblk[L2v][L3v] + half) * cMpad[L3v] + j +      \
                     32 * T0v;                                                                                   \
    if constexpr (!PRELOAD) {                                                                                    \
      _Pragma("unroll") for (int t = 0; t < NTv; ++t) {                                                          \
        pwh[Slots::slot(L1, L2v, L3v) + t] = whi_base[o + 32 * t];                                               \
        if constexpr (!IO16) pwl[Slots::slot(L1, L2v, L3v) + t] = wlo_base[o + 32 * t];                          \
      }                                                                                                          \
    }                                                                                                            \
    if (dbg & 4) {                                                                                               \
    } else if constexpr (IO16) {                                                                                 \
      run_steps_io16_lean<L1, L2v, L3v, NTv>(reinterpret_cast<const uint32_t*>(xr), ch.count, whi_base + o,           \
                                        pwh + Slots::slot(L1, L2v, L3v), cMpad[L3v], half, y, ACC);              \
    } else {                                                                                                     \
      run_steps_bf_lean<L1, L2v, L3v, NTv>(xr, ch.count, whi_base + o, wlo_base + o, pwh + Slots::slot(L1, L2v, L3v), \
                                      pwl + Slots::slot(L1, L2v, L3v), cMpad[L3v], half, y, ACC);                \
    }                                                                                                            \
  }
        E3_RUN(0, 0, a0, N0, F0) E3_RUN(1, 0, a0, N0, F0) E3_RUN(2, 0, a0, N0, F0)
        E3_RUN(0, 1, a1, N1, F1) E3_RUN(1, 1, a1, N1, F1) E3_RUN(2, 1, a1, N1, F1)
        E3_RUN(0, 2, a2, N2, F2) E3_RUN(1, 2, a2, N2, F2) E3_RUN(2, 2, a2, N2, F2)
#undef E3_RUN
        tick(3);
        if (ci + 1 < nchunks) preload(std::integral_constant<int, L1N>{}, ci + 1);
        tick(7);
        __syncthreads();  // both waves are done reading the chunk
        tick(6);          // (phase 6 = waiting for the partner wave at this barrier)
        if (ci + 1 < nchunks) stage(ci + 1, cbuf);
        tick(2);
        ++ci;
      };
      for_each_index(process, std::make_index_sequence<sizeof...(L1S)>{});

      // ---- epilogue: each wave transposes its own tiles through its own region of the (now dead) chunk buffer ----
      wait_vm0();
      constexpr int NPASS = 2, NCH = 32 / NPASS, RPP = 16 / NPASS;
      constexpr int DA = NT1 > 0 && RS::top > 1 ? 3 : 1;  // widest tile of role A
      static_assert(32 * (NCH * DA + 4) + 32 * (NCH * (2 * RS::top + 1) + 4) <= CHUNK, "epilogue regions");
      float* ot = cbuf + (ROLE == 1 ? 32 * (NCH * DA + 4) : 0);
      auto chan_of = [&](int r) { return 8 * (r >> 2) + 4 * half + (r & 3); };
      const bool out_vec = !(ldo & 3) && ((reinterpret_cast<uintptr_t>(outv) & 15) == 0);
      auto emit = [&](auto dtag, auto val, auto col, auto ncol, const int width, const bool affine) {
        constexpr int D = decltype(dtag)::value;
        constexpr int TS = NCH * D + 4;
        constexpr int UPR = NCH * D / 4;
        const int colb = col(0), ncolb = ncol(0);
        const bool vec = affine && out_vec && !(colb & 3) && !(width & 3);
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
          for (int r = 0; r < RPP; ++r)
#pragma unroll
            for (int c = 0; c < D; ++c) ot[j * TS + D * (chan_of(ps * RPP + r) - ps * NCH) + c] = val(ps * RPP + r, c);
          wave_sync_lds();
          tick(4);
          if constexpr (SCAT) {
            // lane = column of this pass: running sum over the 32 rows, flushed with one fp32 atomic per lane whenever
            // the node id changes (ids ascend: a run is contiguous) and at the end.  The node ids are wave-uniform
            // (v_readlane), so the row loop is scalar control flow around one LDS read and one FMA; an atomic
            // instruction covers up to 64 consecutive floats of one node row.
            for (int cb = 0; cb < NCH * D; cb += 64) {
              const int lc = cb + lane, glc = ps * NCH * D + lc;
              const bool cok = lc < NCH * D && glc < width;
              const float nv = cok ? (ncolb >= 0 ? nrm[ncolb + glc] : 1.f) : 0.f;
              float* const obase = reinterpret_cast<float*>(outv) + colb + glc;
              const float* src = ot + (cok ? lc : 0);
              float acc = 0.f;
              int cur = -1;
#pragma unroll 8
              for (int r = 0; r < 32; ++r) {
                const int dn = __builtin_amdgcn_readlane(sd, r);
                if (dn != cur) {
                  if (cur >= 0 && cok && !(dbg & 1)) __builtin_amdgcn_global_atomic_fadd_f32(obase + (int64_t)cur * ldo, acc);
                  acc = 0.f;
                  cur = dn;
                }
                if (dn >= 0) acc = __builtin_fmaf(src[r * TS], nv, acc);
              }
              if (cur >= 0 && cok && !(dbg & 1)) __builtin_amdgcn_global_atomic_fadd_f32(obase + (int64_t)cur * ldo, acc);
            }
          } else if (vec) {
            constexpr uint32_t INV = (65536 + UPR - 1) / UPR;
            static_assert(((32u * UPR - 1) * INV >> 16) == 31 && ((31u * UPR) * INV >> 16) == 31 &&
                          ((30u * UPR + UPR - 1) * INV >> 16) == 30, "reciprocal");
            const uint32_t ldo32 = (uint32_t)ldo;
            const float* nbase = ncolb >= 0 ? nrm + ncolb + ps * NCH * D : nrm + Dout;
            const uint32_t nstep = ncolb >= 0 ? 4u : 0u;
#pragma unroll 2
            for (int it = 0; it < (UPR + 1) / 2; ++it) {
              const uint32_t u = it * 64 + lane;
              const uint32_t row = __umul24(u, INV) >> 16, un = u - __umul24(row, UPR);
              const uint32_t lc0 = ps * NCH * D + un * 4;
              if (row < 32u) {
                float4 v = *reinterpret_cast<const float4*>(ot + __umul24(row, TS) + un * 4);
                const float* np = nbase + un * nstep;
                const float n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                if ((int)row < nrows && (int)lc0 < width && !(dbg & 1)) {
                  v.x *= n0; v.y *= n1; v.z *= n2; v.w *= n3;
                  const uint32_t o = __umul24(row, ldo32) + (uint32_t)colb + lc0;
                  if (IO16) {
                    uint2 pk;
                    pk.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{v.x, v.y}, bf16x2_t));
                    pk.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{v.z, v.w}, bf16x2_t));
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(outv) + row0 * ldo + o) = pk;
                  } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(outv) + row0 * ldo + o) = v;
                  }
                }
              }
            }
          } else {
            for (int lc = lane; lc < NCH * D; lc += 64) {
              const int glc = ps * NCH * D + lc;
              if (glc >= width) continue;
              const int64_t c0 = row0 * ldo + col(glc);
              const int nc = ncol(glc);
              const float nv = nc >= 0 ? nrm[nc] : 1.f;
              const float* src = ot + lc;
#pragma unroll 1
              for (int r = 0; r < nrows; ++r) {
                const float v = src[r * TS] * nv;
                if (dbg & 1) continue;
                if (IO16)
                  reinterpret_cast<uint16_t*>(outv)[c0 + (int64_t)r * ldo] = __builtin_bit_cast(uint16_t, (__bf16)v);
                else
                  reinterpret_cast<float*>(outv)[c0 + (int64_t)r * ldo] = v;
              }
            }
          }
          wave_sync_lds();
          tick(5);
        }
      };
      using I1 = std::integral_constant<int, 1>;
      using I3 = std::integral_constant<int, 3>;
      using I5 = std::integral_constant<int, 5>;
      if constexpr (GATE) {
        // TP out irreps = [32 scalars | 32 gates per gated block | 32x1o | 32x2e]; written layout =
        // [silu(s) (32) | sigmoid(g1) v1 (96) | sigmoid(g2) v2 (160)].  Wave A holds every scalar tile: it first parks
        // sigmoid(gates of the top block) in wave B's (still unused) LDS region, lane-for-lane in the accumulator
        // layout, then both meet at a barrier and B picks its 16 values up.
        const float* nrm0 = nrm + ocl[cOoff[0]];
        constexpr int GT = NT0 - 1;  // scalar tile holding the gates of the top block
        float* gx = cbuf + 32 * (NCH * DA + 4);  // = wave B's region
        if constexpr (ROLE == 0) {
          if constexpr (AG) {
#pragma unroll
            for (int r = 0; r < 16; ++r) gx[r * 64 + lane] = sigmoid_(a0[GT][0][r] * nrm0[32 * GT + chan_of(r)]);
            __syncthreads();
          }
          emit(I1{}, [&](int r, int) { const float s = a0[0][0][r] * nrm0[chan_of(r)]; return s * sigmoid_(s); },
               [&](int lc) { return lc; }, [&](int) { return -1; }, 32, true);
          if constexpr (RS::top == 2 && NT1 > 0) {
            float g[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) g[r] = sigmoid_(a0[1][0][r] * nrm0[32 + chan_of(r)]);
            const int nb = ocl[cOoff[1]];
            emit(I3{}, [&](int r, int c) { return g[r] * a1[0][c][r]; }, [&](int lc) { return 32 + lc; },
                 [&](int lc) { return nb + lc; }, 96, true);
          }
        } else {
          float g[16];
          if constexpr (AG) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) g[r] = gx[r * 64 + lane];
            wave_sync_lds();  // all 16 reads done before this wave overwrites the region
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) g[r] = sigmoid_(a0[0][0][r] * nrm0[32 * GT + chan_of(r)]);
          }
          const int nb = ocl[cOoff[RS::top]];
          const int ocol = 32 + (RS::top == 2 && NT1 > 0 ? 96 : 0);
          if constexpr (RS::top == 2)
            emit(I5{}, [&](int r, int c) { return g[r] * a2[0][c][r]; }, [&](int lc) { return ocol + lc; },
                 [&](int lc) { return nb + lc; }, 160, true);
          else
            emit(I3{}, [&](int r, int c) { return g[r] * a1[0][c][r]; }, [&](int lc) { return ocol + lc; },
                 [&](int lc) { return nb + lc; }, 96, true);
        }
      } else {
        auto tile = [&](auto dtag, int l3, int t, auto val) {
          constexpr int D = decltype(dtag)::value;
          const int base = cOoff[l3] + t * 32;
          const int cnt = cM[l3] - t * 32 < 32 ? cM[l3] - t * 32 : 32;
          const bool affine = ocl[base + cnt - 1] == ocl[base] + (cnt - 1) * D;
          auto colf = [&](int lc) { return ocl[base + lc / D] + lc % D; };
          emit(dtag, val, colf, colf, cnt * D, affine);
        };
#pragma unroll
        for (int t = 0; t < N0; ++t) tile(I1{}, 0, F0 + t, [&](int r, int) { return a0[t][0][r]; });
#pragma unroll
        for (int t = 0; t < N1; ++t) tile(I3{}, 1, F1 + t, [&](int r, int c) { return a1[t][c][r]; });
#pragma unroll
        for (int t = 0; t < N2; ++t) tile(I5{}, 2, F2 + t, [&](int r, int c) { return a2[t][c][r]; });
      }
      tick(4);
    }
  };
  if (wave == 0) body(std::integral_constant<int, 0>{});
  else body(std::integral_constant<int, 1>{});
  // E3_TP_DBG & 128 / & 256: only wave B / only wave A reports (per-role phase shares)
  if (prof && lane == 0 && !((dbg & 128) && wave == 0) && !((dbg & 256) && wave == 1))
    for (int q = 0; q < 8; ++q) atomicAdd(&prof[q], tacc[q]);
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
struct AbKernelEntry {
  int lsh, nt0, nt1, nt2;
  std::vector<int> l1s;
  const void* fn[2][2];  // [mode - 1][gate]
  const void* fn_scat[2];  // [mode - 1]: gated, fused segment-sum into an fp32 node array (nullptr = not instantiated)
};
#define E3_AB(LSH, a, b, c, SC, ...)                                                                          \
  {LSH, a, b, c, {__VA_ARGS__},                                                                               \
   {{(const void*)tp_fwd_mfma_ab_kernel<LSH, a, b, c, false, 1, false, __VA_ARGS__>,                           \
     (const void*)tp_fwd_mfma_ab_kernel<LSH, a, b, c, true, 1, false, __VA_ARGS__>},                           \
    {(const void*)tp_fwd_mfma_ab_kernel<LSH, a, b, c, false, 2, false, __VA_ARGS__>,                           \
     (const void*)tp_fwd_mfma_ab_kernel<LSH, a, b, c, true, 2, false, __VA_ARGS__>}},                          \
   {SC ? (const void*)tp_fwd_mfma_ab_kernel<LSH, a, b, c, true, 1, SC, __VA_ARGS__> : nullptr,                 \
    nullptr /* bf16 storage: measured 2 % slower than the two kernels (the bf16 segment-sum only reads 13.5 GB) */}}
static const std::vector<AbKernelEntry>& ab_kernels() {
  static const std::vector<AbKernelEntry> k = {
      E3_AB(2, 3, 1, 1, false, 0, 1, 2, 0, 1, 2, 0),  // message TP #1
      E3_AB(2, 3, 1, 1, true, 0, 1, 2),               // message TP #2 (+ fused segment-sum)
#ifndef E3_TP_SUBSET
      E3_AB(2, 3, 1, 1, false, 0, 1, 2, 0, 1, 2),     // update TP #1
#endif
  };
  return k;
}

static bool ab_enabled() {
  static const bool on = [] { const char* e = getenv("E3_TP_AB"); return !(e && atoi(e) == 0); }();
  return on;
}

// 1 = launched, 0 = not applicable (caller uses the one-wave kernel), < 0 = -status
int fast_forward_ab(const TpFast* F, const void* sa_, const void* in2, int64_t ld2, const void* packed, void* out,
                    int64_t ldo, int64_t B, int gate, int mode, const int32_t* ocol_tab, hipStream_t s) {
  const bool scat = static_cast<const SegArgs*>(sa_)->scatter != nullptr;
  if (!ab_enabled() || mode < 1) return 0;
  const FDev& d = mode == 2 ? F->dev16 : F->dev;
  std::vector<int> l1s;
  for (auto& c : F->h_chunks) l1s.push_back(c.l1);
  const AbKernelEntry* e = nullptr;
  for (auto& k : ab_kernels())
    if (k.lsh == d.lsh && k.nt0 == d.NT[0] && k.nt1 == d.NT[1] && k.nt2 == d.NT[2] && k.l1s == l1s) e = &k;
  if (!e) return 0;
  if (scat && (!gate || !e->fn_scat[mode - 1])) return 0;
  const void* fn = scat ? e->fn_scat[mode - 1] : e->fn[mode - 1][gate ? 1 : 0];
  const size_t lds_bytes = (size_t)(((d.Dout + 4 + 15) & ~15) + ((d.ntab + 15) & ~15) +
                                    (mode == 2 ? kChunkAb16 : kChunkFloats) + 320) * 4;
  static std::vector<const void*> configured;
  if (std::find(configured.begin(), configured.end(), fn) == configured.end()) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return -E3_ERR_HIP;
    configured.push_back(fn);
  }
  const int per_cu = (int)std::min<size_t>(4, (size_t)kFastLds / lds_bytes);
  if (per_cu < 1) return 0;
  const int64_t ntiles = (B + 31) / 32;
  const int grid = (int)std::min<int64_t>(ntiles, (int64_t)256 * per_cu);
  const float* in2f = (const float*)in2;
  const float* pk = (const float*)packed;
  void* outf = out;
  const FDev* dd = mode == 2 ? F->d_dev16 : F->d_dev;
  const FChunk* dc = F->d_chunks;
  void* args[] = {const_cast<void*>(sa_), &in2f, &ld2, &pk, &outf, &ldo, &B, &dd, &dc, &ocol_tab};
  if (hipLaunchKernel(fn, dim3(grid), dim3(128), args, lds_bytes, s) != hipSuccess) return -E3_ERR_HIP;
  fast_note_kernel("e3::tp_fwd_mfma_ab_kernel");
  return 1;
}

}  // namespace e3
