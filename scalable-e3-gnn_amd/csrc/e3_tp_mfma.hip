// fp32 MFMA kernel for the general SH tensor product (l <= 2, natural-parity classes 0e / 1o / 2e),
// with optional fused row gather (in1 = concatenation of row-indexed segments) and fused gate.
//
// Structure ("K-outer, all outputs resident"): one wave owns a tile of 32 rows and keeps EVERY output
// accumulator of those rows in registers (NT0 scalar tiles + 3*NT1 + 5*NT2 accumulators of 16 VGPRs).  It
// then walks the input irreps blocks ("chunks" of <= 32 channels): each chunk is staged once into a small
// LDS buffer by LDS-DMA (gathered per row when a segment carries a row index) and contracted on the
// matrix core (v_mfma_f32_32x32x2_f32) into every output class it couples to:
//     A operand = packed weights W'[k][32 t + (lane&31)]           (LDS when they fit, else L2)
//     B operand = per-row feature  sum_m1 z[m1][m3] x[k][m1]        z = sum_m2 C[m1][m2][m3] Y[m2]  (per lane)
// or, when 2 l1 + 1 < 2 l3 + 1, the raw x[k][m1] into temporaries that are folded with z afterwards.
// So every input element is read from HBM/L2 once per tile and every output is written once; the gather,
// the concat and the gate of the SEGNN message function never materialise in HBM.
#include "e3_common.h"
#include "cg_tables.h"
#include "e3_tp_internal.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace e3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

constexpr int kFastLds = 160 * 1024;
constexpr int kChunkFloats = 32 * 161;  // 32 rows x (32 ch x 5 comps | 1)
constexpr int kChunk16 = 32 * 81;       // bf16 storage: 32 rows x (80 dwords | 1) (>= 16 rows x 161 floats for the out tile)

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Contract one staged chunk (degree l1) into ALL NT output tiles of degree l3 through SH degree l2.
// The per-row B features are built once per k-step and shared by the NT tiles (one A load + MFMA set per
// tile); loads run U steps ahead of the MFMAs so that >= 12 MFMAs (>= 768 cycles) cover an LDS / L2 round trip.
template <int L1, int L2, int L3, int NT>
__device__ __forceinline__ void run_steps(const float* __restrict__ xr, const int count, const float* __restrict__ wp,
                                          const int Mpad, const int half, const float (&y)[9],
                                          f32x16 (&acc)[NT][2 * L3 + 1]) {
  constexpr int D1 = 2 * L1 + 1, D2 = 2 * L2 + 1, D3 = 2 * L3 + 1;
  constexpr bool MIX = D1 < D3;
  constexpr int PER_STEP = NT * (MIX ? D1 : D3);
  constexpr int U = (PER_STEP >= 3) ? 4 : 8;
  using C = CG<L1, L2, L3>;
  __builtin_amdgcn_sched_barrier(0);
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
  f32x16 T[MIX ? NT : 1][MIX ? D1 : 1];
  if (MIX) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int a = 0; a < D1; ++a) T[t][a] = f32x16{0};
  }
  const float* xp = xr + half * D1;
  auto load = [&](int p, float (&a)[NT], float (&x)[D1]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) a[t] = wp[(2 * p) * Mpad + 32 * t];
#pragma unroll
    for (int m = 0; m < D1; ++m) x[m] = xp[2 * p * D1 + m];
  };
  auto step = [&](const float (&a)[NT], const float (&x)[D1], auto validtag) {
    constexpr bool ALWAYS = decltype(validtag)::value;
    const bool valid = ALWAYS || (half == 0);
    if (MIX) {
#pragma unroll
      for (int m = 0; m < D1; ++m) {
        const float b = valid ? x[m] : 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) T[t][m] = mfma32(a[t], b, T[t][m]);
      }
    } else {
#pragma unroll
      for (int c = 0; c < D3; ++c) {
        float b = 0.f;
        bool have = false;  // folds at compile time: one v_mul then a pure v_fma chain (no "0 + x", no SLP packing)
#pragma unroll
        for (int m = 0; m < D1; ++m) {
          bool nz = false;
#pragma unroll
          for (int q = 0; q < D2; ++q) nz |= (C::v[m][q][c] != 0.0);
          if (nz) {
            b = have ? __builtin_fmaf(z[m][c], x[m], b) : z[m][c] * x[m];
            have = true;
          }
        }
        if (!valid) b = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t][c] = mfma32(a[t], b, acc[t][c]);
      }
    }
  };
  const int npair = count >> 1;
  const int ngrp = npair / U;
  if (ngrp > 0) {
    float a[U][NT], x[U][D1];
#pragma unroll
    for (int u = 0; u < U; ++u) load(u, a[u], x[u]);
    for (int g = 1; g < ngrp; ++g) {
      float an[U][NT], xn[U][D1];
#pragma unroll
      for (int u = 0; u < U; ++u) load(g * U + u, an[u], xn[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) step(a[u], x[u], std::true_type{});
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int t = 0; t < NT; ++t) a[u][t] = an[u][t];
#pragma unroll
        for (int m = 0; m < D1; ++m) x[u][m] = xn[u][m];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) step(a[u], x[u], std::true_type{});
  }
  for (int p = ngrp * U; p < npair; ++p) {
    float a[NT], x[D1];
    load(p, a, x);
    step(a, x, std::true_type{});
  }
  if (count & 1) {  // odd tail: the partner k is a zero weight row; its B lane must be a clean 0
    float a[NT], x[D1];
    load(npair, a, x);
    step(a, x, std::false_type{});
  }
  if (MIX) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int c = 0; c < D3; ++c)
#pragma unroll
        for (int a = 0; a < D1; ++a) {
          bool nz = false;
#pragma unroll
          for (int q = 0; q < D2; ++q) nz |= (C::v[a][q][c] != 0.0);
          if (nz) acc[t][c] += T[t][a] * z[a][c];
        }
  }
  __builtin_amdgcn_sched_barrier(0);
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// bf16-split variant of run_steps: every fp32 operand is written as hi + lo (two bf16 values, 16 significant bits
// together) and a product is accumulated as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulate):
// 16 k per instruction instead of 2, 3 instructions of 32 cycles instead of 8 of 64.  Lane (row j, half h) supplies
// the features of channels 16 kb + 8 h + 0..7 of its own row; the A operand is one 16-byte read of the packed
// [block][half][channel][8] weight layout.
template <int L1, int L2, int L3, int NT>
__device__ __forceinline__ void run_steps_bf(const float* __restrict__ xr, const int count,
                                             const uint4* __restrict__ whi, const uint4* __restrict__ wlo,
                                             const int Mpad, const int half, const float (&y)[9],
                                             f32x16 (&acc)[NT][2 * L3 + 1]) {
  constexpr int D1 = 2 * L1 + 1, D2 = 2 * L2 + 1, D3 = 2 * L3 + 1;
  constexpr bool MIX = false;  // bf16 matrix pipe has slack: extra MFMAs are cheaper than folds through the AGPR file
  constexpr int NB = MIX ? D1 : D3;  // B operands per k block
  using C = CG<L1, L2, L3>;
  __builtin_amdgcn_sched_barrier(0);
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
  f32x16 T[MIX ? NT : 1][MIX ? D1 : 1];
  if (MIX) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int a = 0; a < D1; ++a) T[t][a] = f32x16{0};
  }
  const float* xp = xr + 8 * half * D1;
  const int nkb = (count + 15) >> 4;
  auto load = [&](int kb, uint4 (&ah)[NT], uint4 (&al)[NT], float (&x)[8][D1]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      ah[t] = whi[(2 * kb) * Mpad + 32 * t];
      al[t] = wlo[(2 * kb) * Mpad + 32 * t];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int m = 0; m < D1; ++m) x[i][m] = xp[(16 * kb + i) * D1 + m];
  };
  auto compute = [&](int kb, const uint4 (&ah)[NT], const uint4 (&al)[NT], const float (&x)[8][D1]) {
    (void)kb;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      float f[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float b = 0.f;
        if (MIX) {
          b = x[i][c];
        } else {
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
        }
        f[i] = b;  // channels beyond `count` were staged as zeros (and their weight rows are zero)
      }
      bf16x8 bh, bl;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)f[i];
        bh[i] = h;
        bl[i] = (__bf16)(f[i] - (float)h);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bf16x8 wh = __builtin_bit_cast(bf16x8, ah[t]);
        const bf16x8 wl = __builtin_bit_cast(bf16x8, al[t]);
        f32x16& dst = MIX ? T[t][c] : acc[t][c];
        dst = mfma_bf16(wh, bh, dst);
        dst = mfma_bf16(wh, bl, dst);
        dst = mfma_bf16(wl, bh, dst);
      }
    }
  };
  {
    uint4 ah[NT], al[NT];
    float x[8][D1];
    load(0, ah, al, x);
    for (int kb = 0; kb + 1 < nkb; ++kb) {
      uint4 ahn[NT], aln[NT];
      float xn[8][D1];
      load(kb + 1, ahn, aln, xn);
      compute(kb, ah, al, x);
#pragma unroll
      for (int t = 0; t < NT; ++t) { ah[t] = ahn[t]; al[t] = aln[t]; }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int m = 0; m < D1; ++m) x[i][m] = xn[i][m];
    }
    compute(nkb - 1, ah, al, x);
  }
  if (MIX) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int c = 0; c < D3; ++c)
#pragma unroll
        for (int a = 0; a < D1; ++a) {
          bool nz = false;
#pragma unroll
          for (int q = 0; q < D2; ++q) nz |= (C::v[a][q][c] != 0.0);
          if (nz) acc[t][c] += T[t][a] * z[a][c];
        }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// bf16-storage variant (BASELINE config 3): x staged as bf16 in LDS (two channels per dword), features built in
// fp32, rounded once to bf16, ONE v_mfma_f32_32x32x16_bf16 per 16 k with fp32 accumulation.  `xr32` points at this
// lane's row (dwords); channels beyond `count` were staged as zeros.
template <int L1, int L2, int L3, int NT>
__device__ __forceinline__ void run_steps_io16(const uint32_t* __restrict__ xr32, const int count,
                                               const uint4* __restrict__ whi, const int Mpad, const int half,
                                               const float (&y)[9], f32x16 (&acc)[NT][2 * L3 + 1]) {
  constexpr int D1 = 2 * L1 + 1, D2 = 2 * L2 + 1, D3 = 2 * L3 + 1;
  constexpr int NQ = 4 * D1;  // dwords holding this lane's 8 channels x D1 components
  using C = CG<L1, L2, L3>;
  __builtin_amdgcn_sched_barrier(0);
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
  const uint32_t* xp = xr32 + 4 * half * D1;
  const int nkb = (count + 15) >> 4;
  auto load = [&](int kb, uint4 (&ah)[NT], uint32_t (&q)[NQ]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) ah[t] = whi[(2 * kb) * Mpad + 32 * t];
#pragma unroll
    for (int i = 0; i < NQ; ++i) q[i] = xp[8 * kb * D1 + i];
  };
  auto compute = [&](const uint4 (&ah)[NT], const uint32_t (&q)[NQ]) {
    float x[8][D1];
#pragma unroll
    for (int e = 0; e < 8 * D1; ++e) {
      const uint32_t w = q[e >> 1];
      x[e / D1][e % D1] = __builtin_bit_cast(float, (e & 1) ? (w & 0xffff0000u) : (w << 16));
    }
#pragma unroll
    for (int c = 0; c < D3; ++c) {
      bf16x8 bh;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float b = 0.f;
        bool have = false;
#pragma unroll
        for (int m = 0; m < D1; ++m) {
          bool nz = false;
#pragma unroll
          for (int qq = 0; qq < D2; ++qq) nz |= (C::v[m][qq][c] != 0.0);
          if (nz) {
            b = have ? __builtin_fmaf(z[m][c], x[i][m], b) : z[m][c] * x[i][m];
            have = true;
          }
        }
        bh[i] = (__bf16)b;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][c] = mfma_bf16(__builtin_bit_cast(bf16x8, ah[t]), bh, acc[t][c]);
    }
  };
  {
    uint4 ah[NT];
    uint32_t q[NQ];
    load(0, ah, q);
    for (int kb = 0; kb + 1 < nkb; ++kb) {
      uint4 ahn[NT];
      uint32_t qn[NQ];
      load(kb + 1, ahn, qn);
      compute(ah, q);
#pragma unroll
      for (int t = 0; t < NT; ++t) ah[t] = ahn[t];
#pragma unroll
      for (int i = 0; i < NQ; ++i) q[i] = qn[i];
    }
    compute(ah, q);
  }
  __builtin_amdgcn_sched_barrier(0);
}

struct SegArgs {
  const void* base[4];
  int64_t ld[4];
  const int32_t* index[4];
  int col0[5];  // first in1 column of each segment; col0[nseg] = D1
  int nseg;
};

__device__ __forceinline__ float sigmoid_(float v) { return 1.0f / (1.0f + __expf(-v)); }

// LSH = SH degree of in2; NT* = number of 32-channel output tiles per degree; L1S... = degrees of the input
// chunks in order (compile-time so that the chunk walk is straight-line code: no control-flow merges of the
// 16-register accumulator tuples, which otherwise explode the register allocation).
// MODE: 0 = exact fp32 MFMA, 1 = fp32 in/out with bf16x3-split operands, 2 = bf16 in/out, single bf16 MFMA.
template <int LSH, int NT0, int NT1, int NT2, bool WLDS, bool GATE, int MODE, int... L1S>
__global__ __launch_bounds__(256) void tp_fwd_mfma_kernel(SegArgs segs, const float* __restrict__ in2, int64_t ld2,
                                                          const float* __restrict__ packed, void* __restrict__ outv,
                                                          int64_t ldo, int64_t B, const FDev* __restrict__ dp,
                                                          const FChunk* __restrict__ chunks,
                                                          const int32_t* __restrict__ ocol_tab) {
  constexpr bool BF = MODE >= 1;     // operands go through the bf16 matrix pipe
  constexpr bool IO16 = MODE == 2;   // bf16 storage
  constexpr int CHUNK = IO16 ? kChunk16 : kChunkFloats;  // dwords per chunk buffer
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, half = lane >> 5;
  const int Dout = dp->Dout, Dy = dp->Dy, wtotal = dp->wtotal, nwaves = dp->nwaves, nchunks = dp->nchunks,
            nbuf = dp->nbuf;
  int cM[3], cMpad[3], cWoff[3], cOoff[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { cM[c] = dp->M[c]; cMpad[c] = dp->Mpad[c]; cWoff[c] = dp->woff[c]; cOoff[c] = dp->ooff[c]; }
  const int ntab = dp->ntab;
  const int bftotal = dp->bftotal;
  const int dbg = dp->dbg;
  int cBfoff[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) cBfoff[c] = dp->bfoff[c];
  // weights section of `packed`: [fp32 W' (wtotal) | normcol (Dout) | Whi (bftotal u16) | Wlo (bftotal u16)]
  const int wwords = IO16 ? (bftotal >> 1) : (BF ? bftotal : wtotal);  // 32-bit words of the weight image kept in LDS

  float* wl = lds;
  float* nrm = lds + (WLDS ? wwords : 0);
  int* ocl = reinterpret_cast<int*>(nrm + ((Dout + 15) & ~15));
  float* wbase_lds = reinterpret_cast<float*>(ocl + ((ntab + 15) & ~15));
  const int per_wave = nbuf * CHUNK + 320;
  float* cbuf = wbase_lds + (size_t)wave * per_wave;
  float* ybuf = cbuf + nbuf * CHUNK;
  const float* wglob = BF ? packed + wtotal + ((Dout + 3) & ~3) : packed;
  if (WLDS)
    for (int i = tid; i < wwords; i += blockDim.x) wl[i] = wglob[i];
  for (int i = tid; i < Dout; i += blockDim.x) nrm[i] = packed[wtotal + i];
  for (int i = tid; i < ntab; i += blockDim.x) ocl[i] = ocol_tab[i];
  __syncthreads();
  const float* wsrc = WLDS ? wl : wglob;
  const uint4* whi_base = reinterpret_cast<const uint4*>(wsrc);                      // Whi, 8 bf16 per uint4
  const uint4* wlo_base = reinterpret_cast<const uint4*>(wsrc + (bftotal >> 1));     // Wlo follows Whi

  const int64_t ntiles = (B + 31) / 32;
  const int64_t tstride = (int64_t)gridDim.x * nwaves;

  for (int64_t tile = (int64_t)blockIdx.x * nwaves + wave; tile < ntiles; tile += tstride) {
    const int64_t row0 = tile * 32;
    const int nrows = (int)((B - row0) < 32 ? (B - row0) : 32);

    // stage one chunk of 32 rows (LDS-DMA; per-row gather through the segment's row index).  Row stride (dwords) is
    // odd => the lane=row reads are bank-conflict free.  BF modes zero-pad the chunk to a multiple of 16 channels.
    auto stage = [&](int ci, float* dst) {
      if ((dbg & 2) && tile != (int64_t)blockIdx.x * nwaves + wave) return;
      const FChunk ch = chunks[ci];
      int s = 0;
      while (s + 1 < segs.nseg && ch.col >= segs.col0[s + 1]) ++s;
      const int64_t ld = segs.ld[s];
      const int32_t* idx = segs.index[s];
      const int segcol = ch.col - segs.col0[s];
      const int cw = ch.count * (2 * ch.l1 + 1);                                   // elements per row
      const int cwp = BF ? ((ch.count + 15) & ~15) * (2 * ch.l1 + 1) : cw;         // padded elements per row
      const int dw = IO16 ? (cw >> 1) : cw;                                        // whole dwords per row to DMA
      const int dwp = IO16 ? (cwp >> 1) : cwp;                                     // dwords per padded row
      const int stride = dwp | 1;
      int64_t myrow = row0 + j;
      if (idx && j < nrows) myrow = idx[row0 + j];
      const int mr = (int)myrow;  // row ids fit int32 (N, E < 2^31)
      if (cwp > cw) {  // zero the padding (small chunks only, e.g. the distance scalar)
        for (int r = 0; r < 32; ++r)
          for (int dc = dw + lane; dc < dwp; dc += 64) dst[r * stride + dc] = 0.f;
      }
      if (IO16 && ((cw & 1) || (segcol & 1) || (ld & 1))) {
        // odd widths / 2-byte aligned sources (e.g. the distance scalar, a single 1o channel): the dword DMA cannot be
        // used; copy element-wise through registers (tiny chunks only)
        const uint16_t* b16 = reinterpret_cast<const uint16_t*>(segs.base[s]);
        uint16_t* d16 = reinterpret_cast<uint16_t*>(dst);
        for (int r = 0; r < 32; ++r) {
          const int rr = __builtin_amdgcn_readlane(mr, r < nrows ? r : 0);
          for (int e = lane; e < cw; e += 64)
            d16[r * stride * 2 + e] = (r < nrows) ? b16[(int64_t)rr * ld + segcol + e] : (uint16_t)0;
        }
        return;
      }
      if (!IO16 && cw == 1) {  // one column (the distance scalar): lane r fetches row r
        const float* base = reinterpret_cast<const float*>(segs.base[s]);
        if (stride == 1) {
          if (lane < nrows)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(base + (int64_t)mr * ld + segcol), (lds_void_t*)dst, 4, 0, 0);
          else if (lane < 32)
            dst[lane] = 0.f;
        } else if (lane < 32) {  // padded rows are not contiguous: the DMA cannot scatter, use a register load
          dst[lane * stride] = (lane < nrows) ? base[(int64_t)mr * ld + segcol] : 0.f;
        }
        return;
      }
      const int full = dw & ~63;
      float* drow = dst;
      for (int r = 0; r < 32; ++r) {
        if (r < nrows) {
          const int rr = __builtin_amdgcn_readlane(mr, r);
          const float* srow;  // dword view of the source row segment (+ lane)
          if (IO16)
            srow = reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(segs.base[s]) + (int64_t)rr * ld + segcol) + lane;
          else
            srow = reinterpret_cast<const float*>(segs.base[s]) + (int64_t)rr * ld + segcol + lane;
          for (int dc = 0; dc < full; dc += 64)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(srow + dc), (lds_void_t*)(drow + dc), 4, 0, 0);
          if (full + lane < dw)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(srow + full), (lds_void_t*)(drow + full), 4, 0, 0);
        } else {
          for (int dc = lane; dc < dw; dc += 64) drow[dc] = 0.f;
        }
        drow += stride;
      }
    };

    // Y tile [32][Dy] fp32 (lane e of piece h fetches element h*64+e of the flattened tile)
    for (int h = 0; h * 64 < 32 * Dy; ++h) {
      const int e = h * 64 + lane;
      const int yr = e / Dy, yc = e - yr * Dy;
      if (e < 32 * Dy) {
        if (yr < nrows)
          __builtin_amdgcn_global_load_lds((glb_void_t*)(in2 + (row0 + yr) * ld2 + yc), (lds_void_t*)(ybuf + h * 64), 4,
                                           0, 0);
        else
          ybuf[e] = 0.f;
      }
    }
    stage(0, cbuf);

    f32x16 a0[NT0 > 0 ? NT0 : 1][1], a1[NT1 > 0 ? NT1 : 1][3], a2[NT2 > 0 ? NT2 : 1][5];
#pragma unroll
    for (int t = 0; t < (NT0 > 0 ? NT0 : 1); ++t) a0[t][0] = f32x16{0};
#pragma unroll
    for (int t = 0; t < (NT1 > 0 ? NT1 : 1); ++t)
#pragma unroll
      for (int c = 0; c < 3; ++c) a1[t][c] = f32x16{0};
#pragma unroll
    for (int t = 0; t < (NT2 > 0 ? NT2 : 1); ++t)
#pragma unroll
      for (int c = 0; c < 5; ++c) a2[t][c] = f32x16{0};

    float y[9];
    int cur = 0;
    int ci = 0;
    auto process = [&](auto l1tag) {
      constexpr int L1 = decltype(l1tag)::value;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      wave_sync_lds();
      if (ci == 0) {
#pragma unroll
        for (int q = 0; q < 9; ++q) y[q] = (q < Dy) ? ybuf[j * Dy + q] : 0.f;
      }
      const float* xt = cbuf + cur * CHUNK;
      if (nbuf == 2 && ci + 1 < nchunks) stage(ci + 1, cbuf + (cur ^ 1) * CHUNK);
      const FChunk ch = chunks[ci];
      const int cwp = (BF ? ((ch.count + 15) & ~15) : ch.count) * (2 * L1 + 1);
      const int dwp = IO16 ? (cwp >> 1) : cwp;
      const float* xr = xt + j * (dwp | 1);
#define E3_RUN(L2v, L3v, ACC, NTv)                                                                             \
  if constexpr (NTv > 0 && L2v <= LSH && CG<L1, L2v, L3v>::valid && ((L1 + L2v + L3v) % 2 == 0)) {             \
    if (dbg & 4) {                                                                                             \
    } else if constexpr (IO16) {                                                                               \
      const size_t o = (size_t)(cBfoff[L3v] >> 3) + (size_t)(2 * ch.wblk[L2v][L3v] + half) * cMpad[L3v] + j;   \
      run_steps_io16<L1, L2v, L3v, NTv>(reinterpret_cast<const uint32_t*>(xr), ch.count, whi_base + o,         \
                                        cMpad[L3v], half, y, ACC);                                             \
    } else if constexpr (BF) {                                                                                 \
      const size_t o = (size_t)(cBfoff[L3v] >> 3) + (size_t)(2 * ch.wblk[L2v][L3v] + half) * cMpad[L3v] + j;   \
      run_steps_bf<L1, L2v, L3v, NTv>(xr, ch.count, whi_base + o, wlo_base + o, cMpad[L3v], half, y, ACC);     \
    } else {                                                                                                   \
      const float* wp = wsrc + cWoff[L3v] + (size_t)(ch.wrow[L2v][L3v] + half) * cMpad[L3v] + j;               \
      run_steps<L1, L2v, L3v, NTv>(xr, ch.count, wp, cMpad[L3v], half, y, ACC);                                \
    }                                                                                                          \
  }
      E3_RUN(0, 0, a0, NT0) E3_RUN(1, 0, a0, NT0) E3_RUN(2, 0, a0, NT0)
      E3_RUN(0, 1, a1, NT1) E3_RUN(1, 1, a1, NT1) E3_RUN(2, 1, a1, NT1)
      E3_RUN(0, 2, a2, NT2) E3_RUN(1, 2, a2, NT2) E3_RUN(2, 2, a2, NT2)
#undef E3_RUN
      if (nbuf == 1) {
        wave_sync_lds();
        if (ci + 1 < nchunks) stage(ci + 1, cbuf);
      } else {
        cur ^= 1;
      }
      ++ci;
    };
    (process(std::integral_constant<int, L1S>{}), ...);

    // ---- epilogue: norm (+ gate) in registers, transpose through LDS, coalesced stores ----
    // Every input chunk is consumed, so the chunk buffer becomes the out tile.  fp32 modes: one pass of 32 rows;
    // bf16 storage: two passes of 16 rows (the buffer is half as large), values rounded to bf16 on the way out.
    wave_sync_lds();
    float* ot = cbuf;
    constexpr int NPASS = IO16 ? 2 : 1, RP = 32 / NPASS;
    auto chan_of = [&](int r) { return 8 * (r >> 2) + 4 * half + (r & 3); };
    // emit one job: this lane's values val(r,c) (reg r, component c) of a tile with D components per channel; the
    // tile-local index lc = D*channel + c maps to global column col(lc); `width` = valid lc count; `affine`: col(lc)
    // = col(0) + lc (then bf16 pairs are stored as dwords).
    auto emit = [&](auto dtag, auto val, auto col, const int width, const bool affine) {
      constexpr int D = decltype(dtag)::value;
      constexpr int TS = (32 * D) | 1;
      for (int ps = 0; ps < NPASS; ++ps) {
        if ((j / RP) == ps) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int c = 0; c < D; ++c) ot[(j % RP) * TS + D * chan_of(r) + c] = val(r, c);
        }
        wave_sync_lds();
        const int r0 = ps * RP, r1 = (nrows < r0 + RP) ? nrows : r0 + RP;
        if (!IO16) {
          float* const obase = reinterpret_cast<float*>(outv) + row0 * ldo;
          for (int lc = lane; lc < width; lc += 64) {
            uint32_t off = (uint32_t)col(lc) + (uint32_t)r0 * (uint32_t)ldo;
            const float* src = ot + lc;
            for (int r = r0; r < r1; ++r) { if (!(dbg & 1)) obase[off] = *src; off += (uint32_t)ldo; src += TS; }
          }
        } else {
          uint16_t* const obase = reinterpret_cast<uint16_t*>(outv) + row0 * ldo;
          if (affine && !(width & 1) && !(col(0) & 1) && !(ldo & 1)) {
            uint32_t* const ob32 = reinterpret_cast<uint32_t*>(obase);
            const uint32_t c0 = (uint32_t)col(0) >> 1, ld32 = (uint32_t)ldo >> 1;
            for (int q = lane; q < (width >> 1); q += 64) {
              uint32_t off = c0 + q + (uint32_t)r0 * ld32;
              const float* src = ot + 2 * q;
              for (int r = r0; r < r1; ++r) {
                const uint32_t lo = __builtin_bit_cast(uint16_t, (__bf16)src[0]);
                const uint32_t hi = __builtin_bit_cast(uint16_t, (__bf16)src[1]);
                if (!(dbg & 1)) ob32[off] = lo | (hi << 16);
                off += ld32;
                src += TS;
              }
            }
          } else {
            for (int lc = lane; lc < width; lc += 64) {
              uint32_t off = (uint32_t)col(lc) + (uint32_t)r0 * (uint32_t)ldo;
              const float* src = ot + lc;
              for (int r = r0; r < r1; ++r) {
                if (!(dbg & 1)) obase[off] = __builtin_bit_cast(uint16_t, (__bf16)*src);
                off += (uint32_t)ldo;
                src += TS;
              }
            }
          }
        }
        wave_sync_lds();
      }
    };
    using I1 = std::integral_constant<int, 1>;
    using I3 = std::integral_constant<int, 3>;
    using I5 = std::integral_constant<int, 5>;
    if (GATE) {
      // TP out irreps = [32 scalars | 32 gates per gated block | 32x1o | 32x2e]: a0[0] scalars, a0[1..] gates;
      // written layout = [silu(s) (32) | sigmoid(g1) v1 (96) | sigmoid(g2) v2 (160)]
      emit(I1{}, [&](int r, int) { const float s = a0[0][0][r] * nrm[ocl[cOoff[0] + chan_of(r)]]; return s * sigmoid_(s); },
           [&](int lc) { return lc; }, 32, true);
      int ocol = 32;
      if (NT1 > 0) {
        float g[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = sigmoid_(a0[NT0 > 1 ? 1 : 0][0][r] * nrm[ocl[cOoff[0] + 32 + chan_of(r)]]);
        emit(I3{}, [&](int r, int c) { return g[r] * a1[0][c][r] * nrm[ocl[cOoff[1] + chan_of(r)] + c]; },
             [&](int lc) { return ocol + lc; }, 96, true);
        ocol += 96;
      }
      if (NT2 > 0) {
        constexpr int G2 = (NT1 > 0) ? 2 : 1;  // which scalar tile holds the gates of the 2e block
        float g[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = sigmoid_(a0[NT0 > G2 ? G2 : 0][0][r] * nrm[ocl[cOoff[0] + 32 * G2 + chan_of(r)]]);
        emit(I5{}, [&](int r, int c) { return g[r] * a2[0][c][r] * nrm[ocl[cOoff[2] + chan_of(r)] + c]; },
             [&](int lc) { return ocol + lc; }, 160, true);
      }
    } else {
      // channels of a tile may belong to several irreps blocks: per-channel column lookup; padded channels
      // (>= M) map to column 0 of a clamped entry and are never copied (lc >= width)
      auto nrm_of = [&](int l3, int t, int r, int c) {
        const int chn = t * 32 + chan_of(r);
        return chn < cM[l3] ? nrm[ocl[cOoff[l3] + chn] + c] : 0.f;
      };
#pragma unroll
      for (int t = 0; t < NT0; ++t)
        emit(I1{}, [&](int r, int) { return a0[t][0][r] * nrm_of(0, t, r, 0); },
             [&](int lc) { return ocl[cOoff[0] + t * 32 + lc]; }, (cM[0] - t * 32 < 32 ? cM[0] - t * 32 : 32), false);
#pragma unroll
      for (int t = 0; t < NT1; ++t)
        emit(I3{}, [&](int r, int c) { return a1[t][c][r] * nrm_of(1, t, r, c); },
             [&](int lc) { return ocl[cOoff[1] + t * 32 + lc / 3] + lc % 3; },
             (cM[1] - t * 32 < 32 ? cM[1] - t * 32 : 32) * 3, false);
#pragma unroll
      for (int t = 0; t < NT2; ++t)
        emit(I5{}, [&](int r, int c) { return a2[t][c][r] * nrm_of(2, t, r, c); },
             [&](int lc) { return ocl[cOoff[2] + t * 32 + lc / 5] + lc % 5; },
             (cM[2] - t * 32 < 32 ? cM[2] - t * 32 : 32) * 5, false);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void fast_pack_kernel(const T* w0, const T* w1, const T* w2, const T* n0, const T* n1, const T* n2,
                                 float* packed, FDev d, const FPack* pk, int npk, const int32_t* ocol_tab) {
  const T* w[3] = {w0, w1, w2};
  const T* nr[3] = {n0, n1, n2};
  for (int r = blockIdx.x; r < npk; r += gridDim.x) {
    const FPack q = pk[r];
    const int M = d.M[q.l3], Mpad = d.Mpad[q.l3];
    uint16_t* whi = reinterpret_cast<uint16_t*>(packed + d.wtotal + ((d.Dout + 3) & ~3));
    uint16_t* wlo = whi + d.bftotal;
    for (int i = threadIdx.x; i < q.count * M; i += blockDim.x) {
      int k = i / M, mm = i - k * M;
      const float v = to_acc(w[q.l3][(int64_t)(q.orig_row + k) * M + mm]);
      packed[d.woff[q.l3] + (size_t)(q.wrow + k) * Mpad + mm] = v;
      // bf16 split, layout [16-row block][k half][channel][8]
      const __bf16 h = (__bf16)v;
      const __bf16 l = (__bf16)(v - (float)h);
      const size_t e = (size_t)d.bfoff[q.l3] + ((size_t)(2 * (q.wblk + (k >> 4)) + ((k >> 3) & 1)) * Mpad + mm) * 8 + (k & 7);
      whi[e] = __builtin_bit_cast(uint16_t, h);
      wlo[e] = __builtin_bit_cast(uint16_t, l);
    }
  }
  if (blockIdx.x == 0)
    for (int l3 = 0; l3 < 3; ++l3) {
      const int width = 2 * l3 + 1;
      for (int i = threadIdx.x; i < d.M[l3] * width; i += blockDim.x) {
        int mm = i / width, comp = i - mm * width;
        packed[d.wtotal + ocol_tab[d.ooff[l3] + mm] + comp] = nr[l3] ? to_acc(nr[l3][i]) : 1.0f;
      }
    }
}

struct FastKernelEntry {
  int lsh, nt0, nt1, nt2;
  std::vector<int> l1s;
  const void* fn[3][2][2];  // [mode: 0 exact fp32, 1 fp32 + bf16x3 split, 2 bf16 storage][wlds][gate]
};
#define E3_FAST_M(LSH, a, b, c, M, ...)                                                                \
    {{(const void*)tp_fwd_mfma_kernel<LSH, a, b, c, false, false, M, __VA_ARGS__>,                      \
      (const void*)tp_fwd_mfma_kernel<LSH, a, b, c, false, true, M, __VA_ARGS__>},                      \
     {(const void*)tp_fwd_mfma_kernel<LSH, a, b, c, true, false, M, __VA_ARGS__>,                       \
      (const void*)tp_fwd_mfma_kernel<LSH, a, b, c, true, true, M, __VA_ARGS__>}}
#define E3_FAST(LSH, a, b, c, ...)                                                                     \
  {LSH, a, b, c, {__VA_ARGS__},                                                                         \
   {E3_FAST_M(LSH, a, b, c, 0, __VA_ARGS__), E3_FAST_M(LSH, a, b, c, 1, __VA_ARGS__),                   \
    E3_FAST_M(LSH, a, b, c, 2, __VA_ARGS__)}}
// Instantiated signatures = the tensor products of the SEGNN forward (H <= 32 per block):
//   l_max 1: embed (0,1 -> hid), msg1 (0,1,0,1,0 -> gated), msg2 (0,1 -> gated), upd1 (0,1,0,1 -> gated),
//            upd2 (0,1 -> hid), readout (0,1 -> 1o);   l_max 2: the same with (0,1,2) blocks.
static const std::vector<FastKernelEntry>& fast_kernels() {
  static const std::vector<FastKernelEntry> k = {
      E3_FAST(1, 1, 1, 0, 0, 1),          E3_FAST(1, 2, 1, 0, 0, 1, 0, 1, 0), E3_FAST(1, 2, 1, 0, 0, 1),
      E3_FAST(1, 2, 1, 0, 0, 1, 0, 1),    E3_FAST(1, 0, 1, 0, 0, 1),
      E3_FAST(2, 1, 1, 1, 0, 1),          E3_FAST(2, 3, 1, 1, 0, 1, 2, 0, 1, 2, 0), E3_FAST(2, 3, 1, 1, 0, 1, 2),
      E3_FAST(2, 3, 1, 1, 0, 1, 2, 0, 1, 2), E3_FAST(2, 1, 1, 1, 0, 1, 2),  E3_FAST(2, 0, 1, 0, 0, 1, 2),
  };
  return k;
}

static const FastKernelEntry* find_fast(int lsh, int a, int b, int c, const std::vector<int>& l1s) {
  for (auto& e : fast_kernels())
    if (e.lsh == lsh && e.nt0 == a && e.nt1 == b && e.nt2 == c && e.l1s == l1s) return &e;
  return nullptr;
}

int fast_plan_init(TpFast* F, const int n[6], const int M[6], int lmax_sh, int Dout, int Dy,
                   const std::vector<std::array<int, 4>>& in_blocks /* l,p,mul,col */,
                   const std::vector<TpPath>* paths_by_class /*[6]*/, const int ocol_off[6]) {
  F->usable = false;
  if (n[1] || n[2] || n[5] || M[1] || M[2] || M[5]) return E3_OK;  // natural parity only
  const int cls3[3] = {0, 3, 4};
  FDev& d = F->dev;
  d.Dout = Dout; d.Dy = Dy;
  int ntab = 0;
  for (int c = 0; c < 6; ++c) ntab += M[c];
  d.ntab = ntab;
  for (int l3 = 0; l3 < 3; ++l3) {
    d.M[l3] = M[cls3[l3]];
    d.NT[l3] = (d.M[l3] + 31) / 32;
    d.Mpad[l3] = d.NT[l3] * 32;
    d.ooff[l3] = ocol_off[cls3[l3]];
  }
  d.lsh = lmax_sh;
  d.bf = getenv("E3_TP_EXACT") ? 0 : 1;
  d.dbg = getenv("E3_TP_DBG") ? atoi(getenv("E3_TP_DBG")) : 0;  // timing-only diagnostics, results are wrong when set  // default: bf16-split operands (fp32-grade accuracy)
  int next_row[3] = {0, 0, 0};
  int next_blk[3] = {0, 0, 0};
  int chan_seen[3] = {0, 0, 0};  // channels of in class l1 seen so far
  for (auto& b : in_blocks) {
    const int l1 = b[0], mul = b[2];
    for (int c0 = 0; c0 < mul; c0 += 32) {
      FChunk ch;
      ch.col = b[3] + c0 * (2 * l1 + 1);
      ch.count = std::min(32, mul - c0);
      ch.l1 = l1;
      for (int l2 = 0; l2 < 3; ++l2)
        for (int l3 = 0; l3 < 3; ++l3) {
          ch.wrow[l2][l3] = -1;
          ch.wblk[l2][l3] = -1;
          if (l2 > lmax_sh || d.M[l3] == 0 || ((l1 + l2 + l3) & 1) || l3 < std::abs(l1 - l2) || l3 > l1 + l2) continue;
          // original row offset of path (l1,l2) in class cls3[l3]
          int orig = -1;
          for (auto& p : paths_by_class[cls3[l3]])
            if (p.l1 == l1 && p.l2 == l2) orig = p.wrow;
          if (orig < 0) continue;
          ch.wrow[l2][l3] = next_row[l3];
          ch.wblk[l2][l3] = next_blk[l3];
          F->h_pack.push_back({l3, orig + chan_seen[l1] + 0, ch.count, next_row[l3], next_blk[l3]});
          next_row[l3] += (ch.count + 1) & ~1;
          next_blk[l3] += (ch.count + 15) >> 4;
        }
      chan_seen[l1] += ch.count;
      F->h_chunks.push_back(ch);
    }
  }
  int woff = 0;
  for (int l3 = 0; l3 < 3; ++l3) {
    d.woff[l3] = woff;
    woff += next_row[l3] * d.Mpad[l3];
  }
  d.wtotal = woff;
  int bfo = 0;
  for (int l3 = 0; l3 < 3; ++l3) {
    d.bfoff[l3] = bfo;
    bfo += next_blk[l3] * 16 * d.Mpad[l3];
  }
  d.bftotal = bfo;
  d.nchunks = (int)F->h_chunks.size();
  if (d.nchunks == 0 || d.wtotal == 0) return E3_OK;
  std::vector<int> l1s;
  for (auto& c : F->h_chunks) l1s.push_back(c.l1);
  if (!find_fast(lmax_sh, d.NT[0], d.NT[1], d.NT[2], l1s)) return E3_OK;
  // LDS plan: [weights?][normcol][ocol][nwaves x (nbuf chunk buffers + Y tile)], once per storage class
  size_t tables = (size_t)((Dout + 15) & ~15) * 4 + (size_t)((ntab + 15) & ~15) * 4;
  auto lds_plan = [&](FDev& dd, size_t wbytes, int chunk_dwords, size_t* lds_bytes) -> bool {
    auto per_wave = [&](int nbuf) { return (size_t)(nbuf * chunk_dwords + 320) * 4; };
    auto fit = [&](size_t fixed, int nbuf) -> int {
      return fixed + per_wave(nbuf) <= (size_t)kFastLds ? (int)std::min<size_t>(((size_t)kFastLds - fixed) / per_wave(nbuf), 4) : 0;
    };
    const int w2 = fit(tables + wbytes, 2), w1 = fit(tables + wbytes, 1);
    if (w2 >= 3) { dd.w_in_lds = 1; dd.nbuf = 2; dd.nwaves = w2; }
    else if (w1 >= 4) { dd.w_in_lds = 1; dd.nbuf = 1; dd.nwaves = 4; }
    else { dd.w_in_lds = 0; dd.nbuf = 1; dd.nwaves = fit(tables, 1); }
    if (const char* e = getenv("E3_TP_NBUF")) { int v = atoi(e); if (v == 1 || (v == 2 && dd.w_in_lds && fit(tables + wbytes, 2) >= 1)) { dd.nbuf = v; dd.nwaves = fit(tables + (dd.w_in_lds ? wbytes : 0), v); } }
    *lds_bytes = tables + (dd.w_in_lds ? wbytes : 0) + (size_t)dd.nwaves * per_wave(dd.nbuf);
    return dd.nwaves >= 1 && *lds_bytes <= (size_t)kFastLds;
  };
  F->dev16 = d;
  const bool ok32 = lds_plan(d, d.bf ? (size_t)d.bftotal * 4 : (size_t)d.wtotal * 4, kChunkFloats, &F->lds_bytes);
  const bool ok16 = lds_plan(F->dev16, (size_t)d.bftotal * 2, kChunk16, &F->lds_bytes16);
  if (!ok32 || !ok16) return E3_OK;
  F->usable = true;
  return E3_OK;
}

int fast_upload(TpFast* F) {
  if (!F->usable || F->d_dev) return E3_OK;
  E3_HIP_CHECK(hipMalloc((void**)&F->d_chunks, F->h_chunks.size() * sizeof(FChunk)));
  E3_HIP_CHECK(hipMemcpy(F->d_chunks, F->h_chunks.data(), F->h_chunks.size() * sizeof(FChunk), hipMemcpyHostToDevice));
  E3_HIP_CHECK(hipMalloc((void**)&F->d_pack, std::max<size_t>(F->h_pack.size(), 1) * sizeof(FPack)));
  if (!F->h_pack.empty())
    E3_HIP_CHECK(hipMemcpy(F->d_pack, F->h_pack.data(), F->h_pack.size() * sizeof(FPack), hipMemcpyHostToDevice));
  E3_HIP_CHECK(hipMalloc((void**)&F->d_dev, sizeof(FDev)));
  E3_HIP_CHECK(hipMemcpy(F->d_dev, &F->dev, sizeof(FDev), hipMemcpyHostToDevice));
  E3_HIP_CHECK(hipMalloc((void**)&F->d_dev16, sizeof(FDev)));
  E3_HIP_CHECK(hipMemcpy(F->d_dev16, &F->dev16, sizeof(FDev), hipMemcpyHostToDevice));
  for (auto& e : fast_kernels())
    for (int f = 0; f < 3; ++f)
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
          E3_HIP_CHECK(hipFuncSetAttribute(e.fn[f][a][b], hipFuncAttributeMaxDynamicSharedMemorySize, kFastLds));
  return E3_OK;
}

void fast_free(TpFast* F) {
  if (F->d_chunks) (void)hipFree(F->d_chunks);
  if (F->d_pack) (void)hipFree(F->d_pack);
  if (F->d_dev) (void)hipFree(F->d_dev);
  if (F->d_dev16) (void)hipFree(F->d_dev16);
}

int64_t fast_packed_bytes(const TpFast* F) {
  if (!F->usable) return 0;
  int64_t words = (int64_t)F->dev.wtotal + ((F->dev.Dout + 3) & ~3) + F->dev.bftotal;  // fp32 W' | normcol | Whi+Wlo
  return (words * 4 + 255) / 256 * 256;
}

int fast_pack(const TpFast* F, const void* const w[6], const void* const n[6], int dtype, void* packed,
              const int32_t* ocol_tab, hipStream_t s) {
  if (!F->usable) return E3_OK;
  E3_HIP_CHECK(hipMemsetAsync(packed, 0, (size_t)fast_packed_bytes(F), s));
  int npk = (int)F->h_pack.size();
  dim3 grid(std::max(1, std::min(npk, 256)));
  if (dtype == E3_BF16)
    hipLaunchKernelGGL(fast_pack_kernel<bf16>, grid, dim3(256), 0, s, (const bf16*)w[0], (const bf16*)w[3],
                       (const bf16*)w[4], (const bf16*)(n ? n[0] : nullptr), (const bf16*)(n ? n[3] : nullptr),
                       (const bf16*)(n ? n[4] : nullptr), (float*)packed, F->dev, F->d_pack, npk, ocol_tab);
  else
    hipLaunchKernelGGL(fast_pack_kernel<float>, grid, dim3(256), 0, s, (const float*)w[0], (const float*)w[3],
                       (const float*)w[4], (const float*)(n ? n[0] : nullptr), (const float*)(n ? n[3] : nullptr),
                       (const float*)(n ? n[4] : nullptr), (float*)packed, F->dev, F->d_pack, npk, ocol_tab);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

int fast_forward(const TpFast* F, const e3_tp_segment* segs, int nseg, int D1, const void* in2, int64_t ld2,
                 const void* packed, void* out, int64_t ldo, int64_t B, int gate, int dtype, const int32_t* ocol_tab,
                 hipStream_t s) {
  if (!F->usable) return E3_ERR_UNSUPPORTED;
  const bool io16 = dtype == E3_BF16;
  const FDev& d = io16 ? F->dev16 : F->dev;
  const int mode = io16 ? 2 : (d.bf ? 1 : 0);
  if (nseg < 1 || nseg > 4) return E3_ERR_INVALID_ARG;
  SegArgs sa;
  int col = 0;
  for (int i = 0; i < 4; ++i) { sa.base[i] = nullptr; sa.ld[i] = 0; sa.index[i] = nullptr; }
  for (int i = 0; i < nseg; ++i) {
    if (!segs[i].base || segs[i].ncols <= 0 || segs[i].ld < segs[i].ncols) return E3_ERR_INVALID_ARG;
    sa.base[i] = segs[i].base;
    sa.ld[i] = segs[i].ld;
    sa.index[i] = segs[i].row_index;
    sa.col0[i] = col;
    col += segs[i].ncols;
  }
  for (int i = nseg; i < 5; ++i) sa.col0[i] = col;
  sa.nseg = nseg;
  if (col != D1) return E3_ERR_INVALID_ARG;
  for (auto& ch : F->h_chunks) {  // a chunk must not straddle two segments
    int cw = ch.count * (2 * ch.l1 + 1), sidx = 0;
    while (sidx + 1 < nseg && ch.col >= sa.col0[sidx + 1]) ++sidx;
    if (ch.col + cw > sa.col0[sidx + 1]) return E3_ERR_INVALID_ARG;
  }
  if (gate) {
    const int nb = (d.NT[1] > 0) + (d.NT[2] > 0);
    if (d.NT[0] != 1 + nb || d.M[0] != 32 * (1 + nb) || (d.NT[1] && d.M[1] != 32) || (d.NT[2] && d.M[2] != 32) ||
        d.NT[1] > 1 || d.NT[2] > 1)
      return E3_ERR_UNSUPPORTED;
  }
  std::vector<int> l1s;
  for (auto& c : F->h_chunks) l1s.push_back(c.l1);
  const FastKernelEntry* e = find_fast(d.lsh, d.NT[0], d.NT[1], d.NT[2], l1s);
  if (!e) return E3_ERR_UNSUPPORTED;
  int64_t ntiles = (B + 31) / 32;
  int grid = (int)std::min<int64_t>((ntiles + d.nwaves - 1) / d.nwaves, 256);
  const float* in2f = (const float*)in2;
  const float* pk = (const float*)packed;
  void* outf = out;
  const FDev* dd = io16 ? F->d_dev16 : F->d_dev;
  const size_t lds_bytes = io16 ? F->lds_bytes16 : F->lds_bytes;
  const FChunk* dc = F->d_chunks;
  void* args[] = {&sa, &in2f, &ld2, &pk, &outf, &ldo, &B, &dd, &dc, &ocol_tab};
  E3_HIP_CHECK(hipLaunchKernel(e->fn[mode][d.w_in_lds ? 1 : 0][gate ? 1 : 0], dim3(grid), dim3(64 * d.nwaves), args,
                               lds_bytes, s));
  return E3_OK;
}

}  // namespace e3
