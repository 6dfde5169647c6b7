// 16-row variant of the MFMA tensor-product kernel: one wave owns 16 rows and ALL their outputs.
//
// e3_tp_mfma.hip keeps the outputs of a 32-row tile in one wave (176 accumulator registers for the gated l_max = 2
// message products: 1 wave per SIMD); e3_tp_mfma_ab.hip splits the output degrees over two waves (2 per SIMD, but uneven
// work, two barriers per chunk, both waves at the 256-register limit).  Here the MFMA shape is
// v_mfma_f32_16x16x32_bf16: a 16-channel x 16-row tile per instruction, K = 32 = one whole input chunk.  The outputs of
// 16 rows are 88 registers (4 per 16-channel tile and component), so every wave is independent and identical, there are
// no barriers and no gate hand-off, and 8 waves fit a CU (2 per SIMD) with half the LDS per wave.
//
// Lane = (row = lane & 15, k group g = lane >> 4).  B operand: features of the lane's own row for the 8 channels of its
// k group (the same 2*D1 ds_read_b128 as the 32-row kernel's half).  A operand: W[k = 8g + i][channel = lane & 15] -- the
// packed [16-row block][k half][channel][8] layout already serves it: uint4 index (2 * wblk + g) * Mpad + channel.
// Accumulator: channels 4g + r (r < 4) of the lane's row.  Operand mode: fp32 storage with fp16 (hi, lo)-split operands on
// v_mfma_f32_16x16x32_f16 (MODE 1: three products per fp32 product, operands scaled by powers of two -- split2_f16 in
// e3_tp_mfma_core.h) or bf16 storage on v_mfma_f32_16x16x32_bf16 (MODE 2).
#include "e3_common.h"
#include "cg_tables.h"
#include "e3_tp_internal.h"

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

namespace e3 {

#include "e3_tp_mfma_core.h"

// one product group: acc[t] += A[t] * B for NT tiles, bf16 storage (one MFMA) or fp16-split fp32 (three)
template <int NT16, bool IO16, int STRIDE>
__device__ __forceinline__ void mma_group(const uint4 (&ah)[NT16], const uint4 (&al)[IO16 ? 1 : NT16], const uint4 bh,
                                          const uint4 bl, f32x4* acc) {
  if constexpr (IO16) {
#pragma unroll
    for (int t = 0; t < NT16; ++t)
      acc[t * STRIDE] = mfma16(__builtin_bit_cast(bf16x8, ah[t]), __builtin_bit_cast(bf16x8, bh), acc[t * STRIDE]);
  } else {
    // product-major order: consecutive MFMAs write different accumulators
#pragma unroll
    for (int t = 0; t < NT16; ++t)
      acc[t * STRIDE] = mfma16h(__builtin_bit_cast(f16x8, ah[t]), __builtin_bit_cast(f16x8, bh), acc[t * STRIDE]);
#pragma unroll
    for (int t = 0; t < NT16; ++t)
      acc[t * STRIDE] = mfma16h(__builtin_bit_cast(f16x8, ah[t]), __builtin_bit_cast(f16x8, bl), acc[t * STRIDE]);
#pragma unroll
    for (int t = 0; t < NT16; ++t)
      acc[t * STRIDE] = mfma16h(__builtin_bit_cast(f16x8, al[t]), __builtin_bit_cast(f16x8, bh), acc[t * STRIDE]);
  }
}

// 8 fp32 features -> B operand(s): bf16 (rounded once) or fp16 (hi, lo)
template <bool IO16>
__device__ __forceinline__ void pack_b(const float (&f)[8], uint4& bh, uint4& bl) {
  uint32_t ph[4], pl[4] = {0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if constexpr (IO16)
      ph[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{f[2 * q], f[2 * q + 1]}, bf16x2_t));
    else
      split2_f16(f[2 * q], f[2 * q + 1], ph[q], pl[q]);
  }
  bh = uint4{ph[0], ph[1], ph[2], ph[3]};
  bl = uint4{pl[0], pl[1], pl[2], pl[3]};
}


// this lane's operand slice of a staged chunk: 8 channels (k group g) x D1 components, read ONCE per chunk and shared
// by every path of the chunk
template <int L1, bool IO16>
__device__ __forceinline__ void load_x16(const float* __restrict__ xr, const int g, const float xs,
                                         float (&x)[8][2 * L1 + 1]) {
  constexpr int D1 = 2 * L1 + 1;
  if constexpr (IO16) {
    const uint4* xv = reinterpret_cast<const uint4*>(reinterpret_cast<const uint32_t*>(xr) + 4 * g * D1);
#pragma unroll
    for (int u = 0; u < D1; ++u) {
      const uint4 v = xv[u];
      const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t w = w4[e >> 1];
        (&x[0][0])[8 * u + e] = __builtin_bit_cast(float, (e & 1) ? (w & 0xffff0000u) : (w << 16));
      }
    }
  } else {
    const float4* xv = reinterpret_cast<const float4*>(xr + 8 * g * D1);
#pragma unroll
    for (int u = 0; u < 2 * D1; ++u) {
      const float4 v = xv[u];
      (&x[0][0])[4 * u + 0] = v.x * xs; (&x[0][0])[4 * u + 1] = v.y * xs;
      (&x[0][0])[4 * u + 2] = v.z * xs; (&x[0][0])[4 * u + 3] = v.w * xs;
    }
  }
}

// One input chunk (degree L1, <= 32 channels = one K = 32 step) into NT16 16-channel tiles of output degree L3.
// `x`: the lane's operand slice (load_x16); `whi`/`wlo`: packed weights at (block 2*wblk + g', channel lane & 15) of
// tile 0; `live`: this lane's k group lies inside the (16-padded) chunk -- otherwise its features are zero.
template <int L1, int L2, int L3, int NT16, bool IO16>
__device__ __forceinline__ void run16(const float (&x)[8][2 * L1 + 1], const bool live,
                                      const uint4* __restrict__ whi, const uint4* __restrict__ wlo,
                                      const float (&y)[9], f32x4 (&acc)[NT16][2 * L3 + 1]) {
  constexpr int D1 = 2 * L1 + 1, D2 = 2 * L2 + 1, D3 = 2 * L3 + 1;
  using C = CG<L1, L2, L3>;
  __builtin_amdgcn_sched_barrier(0);
  uint4 ah[NT16], al[IO16 ? 1 : NT16];
#pragma unroll
  for (int t = 0; t < NT16; ++t) {
    ah[t] = whi[16 * t];
    if constexpr (!IO16) al[t] = wlo[16 * t];
  }
  // z[a][c] = sum_b C[a][b][c] Y_l2[b] of THIS lane's row (features and accumulator columns both belong to row lane & 15)
  float z[D1][D3];
#pragma unroll
  for (int a = 0; a < D1; ++a)
#pragma unroll
    for (int c = 0; c < D3; ++c) {
      float s = 0.f;
      bool have = false;
#pragma unroll
      for (int b = 0; b < D2; ++b)
        if (C::v[a][b][c] != 0.0) {
          s = have ? __builtin_fmaf((float)C::v[a][b][c], y[L2 * L2 + b], s) : (float)C::v[a][b][c] * y[L2 * L2 + b];
          have = true;
        }
      z[a][c] = s;
    }
  if constexpr (L1 == 0 && D3 > 1) {
    // scalar input channels into a vector output: out[c] = z[0][c] * (W . x) -- contract the raw channels ONCE into a
    // temporary tile and fold with z afterwards (one feature build and one MFMA group instead of D3 of each)
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = live ? x[i][0] : 0.f;  // a dead k group contributes nothing (its x reads are clamped)
    uint4 bh, bl;
    pack_b<IO16>(f, bh, bl);
    f32x4 T[NT16];
#pragma unroll
    for (int t = 0; t < NT16; ++t) T[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    mma_group<NT16, IO16, 1>(ah, al, bh, bl, &T[0]);
#pragma unroll
    for (int t = 0; t < NT16; ++t)
#pragma unroll
      for (int c = 0; c < D3; ++c) acc[t][c] += T[t] * z[0][c];
    __builtin_amdgcn_sched_barrier(0);
    return;
  }
#pragma unroll
  for (int c = 0; c < D3; ++c) {
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float b = 0.f;
      bool have = false;
#pragma unroll
      for (int m = 0; m < D1; ++m) {
        bool nz = false;
#pragma unroll
        for (int q = 0; q < D2; ++q) nz |= (C::v[m][q][c] != 0.0);
        if (nz) {
          b = have ? __builtin_fmaf(z[m][c], x[i][m], b) : z[m][c] * x[i][m];
          have = true;
        }
      }
      f[i] = live ? b : 0.f;
    }
    uint4 bh, bl;
    pack_b<IO16>(f, bh, bl);
    mma_group<NT16, IO16, D3>(ah, al, bh, bl, &acc[0][c]);
  }
  __builtin_amdgcn_sched_barrier(0);
}

// NT0/NT1/NT2 = 32-channel tile counts per output degree as in the other kernels (each = two 16-channel MFMA tiles)
// waves per SIMD the register allocation is capped for: 2 with l_max = 2 outputs (88 accumulator registers), 3 with the
// l_max = 1 products (40)
// accumulators (f32x4 per lane): 2 NT0 + 6 NT1 + 10 NT2.  Up to 22 (hidden 32: 88 registers) two waves per SIMD for l_max = 2,
// three for l_max = 1 (4 spills 24-49 VGPRs: 61 -> 66 ms); above that (hidden 64: 44 = 176 registers) one wave with the whole file
constexpr int r16_acc(int nt0, int nt1, int nt2) { return 2 * nt0 + 6 * nt1 + 10 * nt2; }
constexpr int r16_waves_per_simd(int nt0, int nt1, int nt2) {
  return r16_acc(nt0, nt1, nt2) > 22 ? 1 : ((nt2 > 0 || r16_acc(nt0, nt1, nt2) > 12) ? 2 : 3);
}
// per-wave chunk buffer (dwords): the larger of a staged chunk (input degree lin) and the out tile (output degree lout):
// 16 rows x (32 channels x (2l + 1) + 4)
constexpr int r16_chunk(int lin, int lout) { return 16 * (32 * (2 * (lin > lout ? lin : lout) + 1) + 4); }
constexpr int r16_max(std::initializer_list<int> v) {
  int m = 0;
  for (int x : v) m = x > m ? x : m;
  return m;
}

template <int LSH, int NT0, int NT1, int NT2, bool GATE, int MODE, bool SCAT, int... L1S>
__global__ __launch_bounds__(256, r16_waves_per_simd(NT0, NT1, NT2)) void tp_fwd_mfma_r16_kernel(SegArgs segs, const float* __restrict__ in2, int64_t ld2,
                                                                  const float* __restrict__ packed, void* __restrict__ outv,
                                                                  int64_t ldo, int64_t B, const FDev* __restrict__ dp,
                                                                  const FChunk* __restrict__ chunks,
                                                                  const int32_t* __restrict__ ocol_tab,
                                                                  const float* __restrict__ in_scale) {
  static_assert(MODE == 1 || MODE == 2, "bf16-pipe modes only");
  constexpr bool IO16 = MODE == 2;
  constexpr int CHUNK = r16_chunk(r16_max({L1S...}), NT2 > 0 ? 2 : (NT1 > 0 ? 1 : 0));
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = blockDim.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int Dout = dp->Dout, Dy = dp->Dy, nchunks = dp->nchunks;
  int cM[3], cMpad[3], cOoff[3], cBfoff[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { cM[c] = dp->M[c]; cMpad[c] = dp->Mpad[c]; cOoff[c] = dp->ooff[c]; cBfoff[c] = dp->bfoff[c]; }
  const int ntab = dp->ntab;
  const int bftotal = dp->bftotal;

  // LDS: [normcol (+4 ones) | ocol | per wave: chunk buffer | Y tile (16 x 9, padded to 160)]
  float* nrm = lds;
  int* ocl = reinterpret_cast<int*>(nrm + ((Dout + 4 + 15) & ~15));
  float* wbase = reinterpret_cast<float*>(ocl + ((ntab + 15) & ~15));
  float* cbuf = wbase + (size_t)wave * (CHUNK + 160);
  float* ybuf = cbuf + CHUNK;
  // operand scales (powers of two): the weights were packed as w * sw, the input features are multiplied by xs on the way
  // into the B operands; both leave through the per-column norm table
  const float* hdr = packed + ((Dout + 3) & ~3);
  const float xs = (!IO16 && in_scale) ? in_scale[0] : 1.0f;
  const float unscale = IO16 ? 1.0f : hdr[2] * (in_scale ? in_scale[1] : 1.0f);
  for (int i = tid; i < Dout; i += blockDim.x) nrm[i] = packed[i] * unscale;
  if (tid < 4) nrm[Dout + tid] = 1.f;
  for (int i = tid; i < ntab; i += blockDim.x) ocl[i] = ocol_tab[i];
  __syncthreads();
  const float* wglob = hdr + 4;
  const uint4* whi_base = reinterpret_cast<const uint4*>(wglob);
  const uint4* wlo_base = reinterpret_cast<const uint4*>(wglob + (bftotal >> 1));

  const int64_t ntiles = (B + 15) / 16;
  const int64_t tstride = (int64_t)gridDim.x * nwaves;
  int mc0 = 0, mc1 = 0, mc2 = 0, mc3 = 0, mn0 = 0, mn1 = 0, mn2 = 0, mn3 = 0;
  auto fetch_ids = [&](int64_t t) {
    const int64_t r = t * 16 + j;
    if (r < B) {
      if (segs.nseg > 0 && segs.index[0]) mn0 = segs.index[0][r];
      if (segs.nseg > 1 && segs.index[1]) mn1 = segs.index[1][r];
      if (segs.nseg > 2 && segs.index[2]) mn2 = segs.index[2][r];
      if (segs.nseg > 3 && segs.index[3]) mn3 = segs.index[3][r];
    }
  };
  const int64_t tile0 = (int64_t)blockIdx.x * nwaves + wave;
  if (tile0 < ntiles) fetch_ids(tile0);
  const int inv_dy = (65536 + Dy - 1) / Dy;
  using Slots = PathSlots<LSH, NT0, NT1, NT2>;
  using Seq = IntSeq<L1S...>;
  float amax = 0.f;  // running max |out| of this lane's stores (epilogue extras)

  for (int64_t tile = tile0; tile < ntiles; tile += tstride) {
    const int64_t row0 = tile * 16;
    const int nrows = (int)((B - row0) < 16 ? (B - row0) : 16);
    mc0 = mn0; mc1 = mn1; mc2 = mn2; mc3 = mn3;
    if (tile + tstride < ntiles) fetch_ids(tile + tstride);
    int sd = -1;
    if constexpr (SCAT) {
      if (j < nrows) sd = segs.scatter[row0 + j];
    }

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
      const int mr = idx ? mg : (int)row0 + j;   // lane & 15 = row
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
        for (int r0 = 0; r0 < 16; r0 += 4 * rows_per) {
          int ridx[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) ridx[k] = __shfl(mr, (r0 + k * rows_per + rl) & 15);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int rbase = r0 + k * rows_per;
            if (rbase < 16 && lane_ok && rbase + rl < nrows)
              __builtin_amdgcn_global_load_lds((glb_void_t*)(lsrc + (uint64_t)(uint32_t)ridx[k] * ldb),
                                               (lds_void_t*)(dst + rbase * S * 4), 16, 0, 0);
          }
        }
        if (cwp > cw || nrows < 16) {
          uint16_t* d16 = reinterpret_cast<uint16_t*>(dst);
          for (int r = 0; r < 16; ++r) {
            const int e0 = (r < nrows) ? cw : 0;
            for (int e = e0 + lane; e < cwp; e += 64) {
              if (IO16) d16[r * S * 8 + e] = 0;
              else dst[r * S * 4 + e] = 0.f;
            }
          }
        }
      } else {
        // narrow / unaligned chunks: lane = (row, element index mod 4), all rows in flight, zeros for padding and tail
        const bool rok = j < nrows;
        if (IO16) {
          uint16_t* drow = reinterpret_cast<uint16_t*>(dst) + j * S * 8;
          const uint16_t* srow = reinterpret_cast<const uint16_t*>(base) + (int64_t)mr * ld + segcol;
          for (int e = g; e < cwp; e += 4) drow[e] = (rok && e < cw) ? srow[e] : (uint16_t)0;
        } else {
          float* drow = dst + j * S * 4;
          const float* srow = reinterpret_cast<const float*>(base) + (int64_t)mr * ld + segcol;
          for (int e = g; e < cwp; e += 4) drow[e] = (rok && e < cw) ? srow[e] : 0.f;
        }
      }
    };

    // Y tile [16][Dy]
    for (int h = 0; h * 64 < 16 * Dy; ++h) {
      const int e = h * 64 + lane;
      const int yr = (e * inv_dy) >> 16, yc = e - yr * Dy;
      if (e < 16 * Dy) {
        if (yr < nrows)
          __builtin_amdgcn_global_load_lds((glb_void_t*)(in2 + (row0 + yr) * ld2 + yc), (lds_void_t*)(ybuf + h * 64), 4,
                                           0, 0);
        else
          ybuf[e] = 0.f;
      }
    }
    stage(0, cbuf);

    constexpr int T0 = 2 * NT0, T1 = 2 * NT1, T2 = 2 * NT2;  // 16-channel tiles
    f32x4 a0[T0 > 0 ? T0 : 1][1], a1[T1 > 0 ? T1 : 1][3], a2[T2 > 0 ? T2 : 1][5];
#pragma unroll
    for (int t = 0; t < (T0 > 0 ? T0 : 1); ++t) a0[t][0] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < (T1 > 0 ? T1 : 1); ++t)
#pragma unroll
      for (int c = 0; c < 3; ++c) a1[t][c] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < (T2 > 0 ? T2 : 1); ++t)
#pragma unroll
      for (int c = 0; c < 5; ++c) a2[t][c] = f32x4{0, 0, 0, 0};

    float y[9];
    int ci = 0;
    auto process = [&](auto itag) {
      constexpr int L1 = Seq::at(decltype(itag)::value);
      wait_vm0();
      wave_sync_lds();
      if (ci == 0) {
#pragma unroll
        for (int q = 0; q < 9; ++q) y[q] = (q < Dy) ? ybuf[j * Dy + q] : 0.f;
      }
      const FChunk ch = chunks[ci];
      const int cpad = (ch.count + 15) & ~15;
      const int cwp = cpad * (2 * L1 + 1);
      const float* xr = cbuf + j * (((cwp / (IO16 ? 8 : 4)) | 1) * 4);
      const bool live = 8 * g < cpad;          // k group inside the padded chunk
      const int gw = live ? g : (g & 1);       // keep the (unused) weight reads of a dead group inside the matrix
      const float* xrl = live ? xr : cbuf;     // and its x reads inside the buffer
      float x[8][2 * L1 + 1];
      load_x16<L1, IO16>(xrl, live ? g : 0, xs, x);
#define E3_RUN(L2v, L3v, ACC, NTv)                                                                             \
  if constexpr (Slots::valid(L1, L2v, L3v)) {                                                                  \
    static_assert(CG<L1, L2v, L3v>::valid, "path bookkeeping");                                                \
    const size_t o = (size_t)(cBfoff[L3v] >> 3) + (size_t)(2 * ch.wblk[L2v][L3v] + gw) * cMpad[L3v] + j;       \
    run16<L1, L2v, L3v, 2 * NTv, IO16>(x, live, whi_base + o, wlo_base + o, y, ACC);                            \
  }
      E3_RUN(0, 0, a0, NT0) E3_RUN(1, 0, a0, NT0) E3_RUN(2, 0, a0, NT0)
      E3_RUN(0, 1, a1, NT1) E3_RUN(1, 1, a1, NT1) E3_RUN(2, 1, a1, NT1)
      E3_RUN(0, 2, a2, NT2) E3_RUN(1, 2, a2, NT2) E3_RUN(2, 2, a2, NT2)
#undef E3_RUN
      wave_sync_lds();
      if (ci + 1 < nchunks) stage(ci + 1, cbuf);
      ++ci;
    };
    for_each_index(process, std::make_index_sequence<sizeof...(L1S)>{});

    // ---- epilogue: transpose through the (dead) chunk buffer, 16 rows x one 32-channel tile at a time ----
    wait_vm0();
    wave_sync_lds();
    float* ot = cbuf;
    // this lane's 8 channels of a 32-channel tile: q = 4 * (16-tile) + r  ->  channel 16 * (q >> 2) + 4 * g + (q & 3)
    auto chan_of = [&](int q) { return 16 * (q >> 2) + 4 * g + (q & 3); };
    const bool out_vec = !(ldo & 3) && ((reinterpret_cast<uintptr_t>(outv) & 15) == 0) &&
                         (!segs.residual || (!(segs.ldr & 3) && (reinterpret_cast<uintptr_t>(segs.residual) & 15) == 0));
    auto emit = [&](auto dtag, auto val, auto col, auto ncol, const int width, const bool affine) {
      constexpr int D = decltype(dtag)::value;
      constexpr int TS = 32 * D + 4;
      constexpr int UPR = 32 * D / 4;
      const int colb = col(0), ncolb = ncol(0);
      const bool vec = affine && out_vec && !(colb & 3) && !(width & 3);
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int c = 0; c < D; ++c) ot[j * TS + D * chan_of(q) + c] = val(q, c);
      wave_sync_lds();
      if constexpr (SCAT) {
        for (int cb = 0; cb < 32 * D; cb += 64) {
          const int lc = cb + lane;
          const bool cok = lc < 32 * D && lc < width;
          const float nv = cok ? (ncolb >= 0 ? nrm[ncolb + lc] : 1.f) : 0.f;
          float* const obase = reinterpret_cast<float*>(outv) + colb + lc;
          const float* src = ot + (cok ? lc : 0);
          float acc = 0.f;
          int cur = -1;
#pragma unroll 8
          for (int r = 0; r < 16; ++r) {
            const int dn = __builtin_amdgcn_readlane(sd, r);
            if (dn != cur) {
              if (cur >= 0 && cok) __builtin_amdgcn_global_atomic_fadd_f32(obase + (int64_t)cur * ldo, acc);
              acc = 0.f;
              cur = dn;
            }
            if (dn >= 0) acc = __builtin_fmaf(src[r * TS], nv, acc);
          }
          if (cur >= 0 && cok) __builtin_amdgcn_global_atomic_fadd_f32(obase + (int64_t)cur * ldo, acc);
        }
      } else if (vec) {
        constexpr uint32_t INV = (65536 + UPR - 1) / UPR;
        static_assert(((16u * UPR - 1) * INV >> 16) == 15 && ((15u * UPR) * INV >> 16) == 15 &&
                      ((14u * UPR + UPR - 1) * INV >> 16) == 14, "reciprocal");
        const uint32_t ldo32 = (uint32_t)ldo;
        const float* nbase = ncolb >= 0 ? nrm + ncolb : nrm + Dout;
        const uint32_t nstep = ncolb >= 0 ? 4u : 0u;
#pragma unroll 2
        for (int it = 0; it < UPR / 4; ++it) {   // 16 * UPR units / 64 lanes
          const uint32_t u = it * 64 + lane;
          const uint32_t row = __umul24(u, INV) >> 16, un = u - __umul24(row, UPR);
          const uint32_t lc0 = un * 4;
          float4 v = *reinterpret_cast<const float4*>(ot + __umul24(row, TS) + un * 4);
          const float* np = nbase + un * nstep;
          const float n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
          if ((int)row < nrows && (int)lc0 < width) {
            v.x *= n0; v.y *= n1; v.z *= n2; v.w *= n3;
            const uint32_t o = __umul24(row, ldo32) + (uint32_t)colb + lc0;
            if constexpr (!SCAT) {
              if (segs.residual) {
                const int64_t ro = (row0 + row) * segs.ldr + colb + lc0;
                if (IO16) {
                  const uint2 rk = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(segs.residual) + ro);
                  v.x += __builtin_bit_cast(float, rk.x << 16); v.y += __builtin_bit_cast(float, rk.x & 0xffff0000u);
                  v.z += __builtin_bit_cast(float, rk.y << 16); v.w += __builtin_bit_cast(float, rk.y & 0xffff0000u);
                } else {
                  const float4 rv = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(segs.residual) + ro);
                  v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                }
              }
              amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            }
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
      } else {
        for (int lc = lane; lc < 32 * D; lc += 64) {
          if (lc >= width) continue;
          const int64_t c0 = row0 * ldo + col(lc);
          const int nc = ncol(lc);
          const float nv = nc >= 0 ? nrm[nc] : 1.f;
          const float* src = ot + lc;
#pragma unroll 1
          for (int r = 0; r < nrows; ++r) {
            float v = src[r * TS] * nv;
            if (segs.residual) {
              const int64_t ro = (row0 + r) * segs.ldr + col(lc);
              v += IO16 ? __builtin_bit_cast(float, (uint32_t)reinterpret_cast<const uint16_t*>(segs.residual)[ro] << 16)
                        : reinterpret_cast<const float*>(segs.residual)[ro];
            }
            amax = fmaxf(amax, fabsf(v));
            if (IO16)
              reinterpret_cast<uint16_t*>(outv)[c0 + (int64_t)r * ldo] = __builtin_bit_cast(uint16_t, (__bf16)v);
            else
              reinterpret_cast<float*>(outv)[c0 + (int64_t)r * ldo] = v;
          }
        }
      }
      wave_sync_lds();
    };
    using I1 = std::integral_constant<int, 1>;
    using I3 = std::integral_constant<int, 3>;
    using I5 = std::integral_constant<int, 5>;
    // accumulator element of 32-channel tile t32, lane value q (see chan_of): 16-tile 2*t32 + (q >> 2), register q & 3
    if constexpr (GATE) {
      // out irreps [H x 0e | H x 0e per gated block | H x 1o | H x 2e], H = 16 HT channels (hidden 16 / 32 / 64 ...): scalar
      // channel H l + c gates channel c of degree l.  A lane's 8 channels of a 32-channel tile sit in two 16-channel
      // accumulator tiles, and H l is a multiple of 16, so the gate of a value is in the SAME lane and register of a0.
      constexpr int NL = (NT1 > 0 ? 1 : 0) + (NT2 > 0 ? 1 : 0);
      constexpr int HT = (2 * NT0) / (1 + NL);       // 16-channel tiles per block
      constexpr int Hh = 16 * HT;
      constexpr int NTS = (Hh + 31) / 32;             // 32-channel tiles of the plain scalars (= of every gated block)
      // (the macro below instantiates the gated form for every tile shape; shapes that are no gated layout -- e.g. the
      // update product #2 -- are never launched with GATE and compile to an empty epilogue)
      constexpr bool GOK = HT > 0 && (NT1 == 0 || NT1 == NTS) && (NT2 == 0 || NT2 == NTS) && (1 + NL) * HT <= 2 * NT0;
      if constexpr (GOK) {
      const float* nrm0 = nrm + ocl[cOoff[0]];
      // a0 tile 2 t + (q >> 2) of block `blk` (0: scalars, 1 / 2: gates); tiles beyond the block hold other channels: masked
      auto a0_of = [&](auto btag, auto ttag, int q) {
        constexpr int BLK = decltype(btag)::value, T32 = decltype(ttag)::value;
        return (q >> 2) == 0 ? a0[(BLK * HT + 2 * T32 < 2 * NT0) ? BLK * HT + 2 * T32 : 0][0][q & 3]
                             : a0[(BLK * HT + 2 * T32 + 1 < 2 * NT0) ? BLK * HT + 2 * T32 + 1 : 0][0][q & 3];
      };
      auto per_tile = [&](auto ttag) {
        constexpr int T32 = decltype(ttag)::value;
        constexpr int W = Hh - 32 * T32 < 32 ? Hh - 32 * T32 : 32;   // channels of this tile that exist
        using B0 = std::integral_constant<int, 0>;
        emit(I1{}, [&](int q, int) {
               const float s = a0_of(B0{}, ttag, q) * nrm0[32 * T32 + chan_of(q)];
               return s * sigmoid_(s);
             },
             [&](int lc) { return 32 * T32 + lc; }, [&](int) { return -1; }, W, true);
        if constexpr (NT1 > 0) {
          using B1 = std::integral_constant<int, 1>;
          float gq[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) gq[q] = sigmoid_(a0_of(B1{}, ttag, q) * nrm0[Hh + 32 * T32 + chan_of(q)]);
          const int nb = ocl[cOoff[1]] + 96 * T32;
          emit(I3{}, [&](int q, int c) { return gq[q] * a1[2 * T32 + (q >> 2)][c][q & 3]; },
               [&](int lc) { return Hh + 96 * T32 + lc; }, [&](int lc) { return nb + lc; }, 3 * W, true);
        }
        if constexpr (NT2 > 0) {
          using B2 = std::integral_constant<int, (NT1 > 0) ? 2 : 1>;
          float gq[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) gq[q] = sigmoid_(a0_of(B2{}, ttag, q) * nrm0[B2::value * Hh + 32 * T32 + chan_of(q)]);
          const int nb = ocl[cOoff[2]] + 160 * T32;
          emit(I5{}, [&](int q, int c) { return gq[q] * a2[2 * T32 + (q >> 2)][c][q & 3]; },
               [&](int lc) { return Hh * (NT1 > 0 ? 4 : 1) + 160 * T32 + lc; }, [&](int lc) { return nb + lc; }, 5 * W, true);
        }
      };
      for_each_index(per_tile, std::make_index_sequence<NTS>{});
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
      for (int t = 0; t < NT0; ++t) tile(I1{}, 0, t, [&](int q, int) { return a0[2 * t + (q >> 2)][0][q & 3]; });
#pragma unroll
      for (int t = 0; t < NT1; ++t) tile(I3{}, 1, t, [&](int q, int c) { return a1[2 * t + (q >> 2)][c][q & 3]; });
#pragma unroll
      for (int t = 0; t < NT2; ++t) tile(I5{}, 2, t, [&](int q, int c) { return a2[2 * t + (q >> 2)][c][q & 3]; });
    }
  }
  if constexpr (!SCAT) {
    if (segs.amax) {  // NaN / inf leave the maximum at the largest finite value seen (as e3_pow2_scale does)
      amax = amax < INFINITY ? amax : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
      if (lane == 0 && amax > 0.f) atomicMax(segs.amax, __builtin_bit_cast(uint32_t, amax));
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
struct R16KernelEntry {
  int lsh, nt0, nt1, nt2;
  std::vector<int> l1s;
  const void* fn[2][2];  // [mode - 1][gate]
  const void* fn_scat;   // fp32 storage, gated, fused segment-sum
};
#define E3_R16(LSH, a, b, c, SC, ...)                                                                          \
  {LSH, a, b, c, {__VA_ARGS__},                                                                                \
   {{(const void*)tp_fwd_mfma_r16_kernel<LSH, a, b, c, false, 1, false, __VA_ARGS__>,                           \
     (const void*)tp_fwd_mfma_r16_kernel<LSH, a, b, c, true, 1, false, __VA_ARGS__>},                           \
    {(const void*)tp_fwd_mfma_r16_kernel<LSH, a, b, c, false, 2, false, __VA_ARGS__>,                           \
     (const void*)tp_fwd_mfma_r16_kernel<LSH, a, b, c, true, 2, false, __VA_ARGS__>}},                          \
   SC ? (const void*)tp_fwd_mfma_r16_kernel<LSH, a, b, c, true, 1, SC, __VA_ARGS__> : nullptr}
static const std::vector<R16KernelEntry>& r16_kernels() {
  static const std::vector<R16KernelEntry> k = {
      E3_R16(2, 3, 1, 1, false, 0, 1, 2, 0, 1, 2, 0),  // l_max 2: message TP #1
      E3_R16(2, 3, 1, 1, true, 0, 1, 2),               //          message TP #2 (+ fused segment-sum)
      E3_R16(2, 3, 1, 1, false, 0, 1, 2, 0, 1, 2),     //          update TP #1
      E3_R16(1, 2, 1, 0, false, 0, 1, 0, 1, 0),        // l_max 1: message TP #1
      E3_R16(1, 2, 1, 0, true, 0, 1),                  //          message TP #2 (+ fused segment-sum)
      E3_R16(1, 2, 1, 0, false, 0, 1, 0, 1),           //          update TP #1
      E3_R16(2, 1, 1, 1, false, 0, 1, 2),              // l_max 2: update TP #2
      E3_R16(2, 1, 1, 1, false, 0, 1),                 //          embedding
      E3_R16(2, 0, 1, 0, false, 0, 1, 2),              //          readout
      E3_R16(1, 1, 1, 0, false, 0, 1),                 // l_max 1: update TP #2 / embedding
      E3_R16(1, 0, 1, 0, false, 0, 1),                 //          readout
      // hidden 16 (chunks of 16 channels: one dead K half) -- the other products share the hidden-32 entries above
      E3_R16(2, 2, 1, 1, false, 0, 1, 2, 0, 1, 2),     // l_max 2: update TP #1
      E3_R16(1, 1, 1, 0, false, 0, 1, 0, 1),           // l_max 1: update TP #1
      // hidden 64 (two 32-channel chunks per degree and segment)
      E3_R16(2, 6, 2, 2, false, 0, 0, 1, 1, 2, 2, 0, 0, 1, 1, 2, 2),   // l_max 2: update TP #1
      E3_R16(2, 2, 2, 2, false, 0, 0, 1, 1, 2, 2),     //          update TP #2
      E3_R16(2, 2, 2, 2, false, 0, 1),                 //          embedding
      E3_R16(2, 0, 1, 0, false, 0, 0, 1, 1, 2, 2),     //          readout
      E3_R16(1, 4, 2, 0, false, 0, 0, 1, 1, 0, 0, 1, 1),   // l_max 1: update TP #1
      E3_R16(1, 2, 2, 0, false, 0, 0, 1, 1),           //          update TP #2
      E3_R16(1, 2, 2, 0, false, 0, 1),                 //          embedding
      E3_R16(1, 0, 1, 0, false, 0, 0, 1, 1),           //          readout
  };
  return k;
}

static const R16KernelEntry* r16_find(const TpFast* F) {
  const FDev& d = F->dev;
  std::vector<int> l1s;
  for (auto& c : F->h_chunks) {
    l1s.push_back(c.l1);
    if (c.count != 32 && c.count > 16) return nullptr;  // K = 32 steps: whole 32-channel chunks (or <= 16 channels: one dead half)
  }
  for (auto& k : r16_kernels())
    if (k.lsh == d.lsh && k.nt0 == d.NT[0] && k.nt1 == d.NT[1] && k.nt2 == d.NT[2] && k.l1s == l1s) return &k;
  return nullptr;
}
static size_t r16_lds_bytes(const TpFast* F, int nwaves) {
  const FDev& d = F->dev;
  const size_t tables = (size_t)(((d.Dout + 4 + 15) & ~15) + ((d.ntab + 15) & ~15)) * 4;
  int lin = 0;
  for (auto& c : F->h_chunks) lin = std::max(lin, c.l1);
  const size_t per_wave = (size_t)(r16_chunk(lin, d.NT[2] > 0 ? 2 : (d.NT[1] > 0 ? 1 : 0)) + 160) * 4;
  return tables + nwaves * per_wave;
}
constexpr int kR16Waves = 4;  // per workgroup; two or three workgroups per CU

bool r16_supported(const TpFast* F) {
  return r16_find(F) != nullptr &&
         (size_t)r16_waves_per_simd(F->dev.NT[0], F->dev.NT[1], F->dev.NT[2]) * r16_lds_bytes(F, kR16Waves) <= (size_t)kFastLds;
}

// 1 = launched, 0 = not applicable, < 0 = -status
int fast_forward_r16(const TpFast* F, const void* sa_, const void* in2, int64_t ld2, const void* packed, void* out,
                     int64_t ldo, int64_t B, int gate, int io16, const int32_t* ocol_tab, const float* in_scale,
                     hipStream_t s) {
  const bool scat = static_cast<const SegArgs*>(sa_)->scatter != nullptr;
  const FDev& d = F->dev;
  const R16KernelEntry* e = r16_find(F);
  if (!e) return 0;
  if (scat && (io16 || !gate || !e->fn_scat)) return 0;
  if (scat && (static_cast<const SegArgs*>(sa_)->residual || static_cast<const SegArgs*>(sa_)->amax)) return 0;
  const void* fn = scat ? e->fn_scat : e->fn[io16 ? 1 : 0][gate ? 1 : 0];
  const size_t lds_bytes = r16_lds_bytes(F, kR16Waves);
  if ((size_t)r16_waves_per_simd(d.NT[0], d.NT[1], d.NT[2]) * lds_bytes > (size_t)kFastLds) return 0;
  {  // the dynamic-LDS limit is a per-device attribute of the function
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -E3_ERR_HIP;
    static std::mutex mu;
    static std::vector<std::pair<const void*, int>> configured;
    std::lock_guard<std::mutex> lock(mu);
    if (std::find(configured.begin(), configured.end(), std::make_pair(fn, dev)) == configured.end()) {
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return -E3_ERR_HIP;
      configured.emplace_back(fn, dev);
    }
  }
  const int64_t ntiles = (B + 15) / 16;
  const int grid = (int)std::min<int64_t>((ntiles + kR16Waves - 1) / kR16Waves, 256 * r16_waves_per_simd(d.NT[0], d.NT[1], d.NT[2]));
  const float* in2f = (const float*)in2;
  const float* pk = (const float*)packed;
  void* outf = out;
  const FDev* dd = F->d_dev;
  const FChunk* dc = F->d_chunks;
  void* args[] = {const_cast<void*>(sa_), &in2f, &ld2, &pk, &outf, &ldo, &B, &dd, &dc, &ocol_tab, &in_scale};
  if (hipLaunchKernel(fn, dim3(grid), dim3(64 * kR16Waves), args, lds_bytes, s) != hipSuccess) return -E3_ERR_HIP;
  fast_note_kernel("e3::tp_fwd_mfma_r16_kernel");
  return 1;
}

}  // namespace e3
