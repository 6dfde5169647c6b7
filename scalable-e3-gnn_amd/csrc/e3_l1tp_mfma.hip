// fp32 MFMA kernel for the L1 tensor product (gfx950, v_mfma_f32_32x32x2_f32: exact fp32 FMA chain).
//
// Mapping.  One wave owns a tile of 32 rows (edges / nodes).  For every output class and every tile of
// 32 output channels it runs   D[ch][row] += W'[k][ch] * f[k][row]   on the matrix core with
//   A operand = packed weights  W'[k0 + (lane>>5)][32 t + (lane&31)]   (from LDS, zero padded, CG folded)
//   B operand = per-row feature f[k0 + (lane>>5)] of row (lane&31), built on the fly from the staged
//               in1 tile:   scalars: x*Y0 (0e/0o outputs) or x (1e/1o: Y1 applied after the mix);
//               vectors: <v,Y1> (0e/0o), v_c*Y0 or (v x Y1)_c (1e/1o, one accumulator per component c)
// so a lane ends up with 16 output channels of *its own row*: the Y-dependent epilogue is per-lane
// scalar math and Y lives in 4 VGPRs.  The in1 tile is staged global -> LDS with coalesced loads in its
// original column order (row stride odd => the lane=row reads are bank-conflict free); results go
// through an LDS out-tile so that the global stores are coalesced 256-B row segments, with the
// per-column norm applied on the way out (L1TP.py:256,269,284,297).
#include "e3_common.h"

#include <algorithm>
#include <cstdlib>

namespace e3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MRun {  // one run of in1 channels feeding one output class
  int col;      // first in1 column of the run
  int count;    // channels in the run
  int cstride;  // 1 (scalars) or 3 (vectors, xyz adjacent)
  int wrow;     // first packed weight row (runs are padded to an even number of rows)
};

struct MfmaDev {
  int D1, Dout, SI, SJ;  // SJ: row stride of the per-job out tile (odd)
  int M[4], NT[4], Mpad[4];
  int woff[4];       // float offset of the class's packed weights [Kpad][Mpad]
  int wtotal;        // floats of packed weights
  int nrun[4][3];    // per out class: #runs of {scalars, same-parity vectors, cross vectors}
  int roff[4][3];    // offsets into `runs`
  int ocol_off[4];
  int w_in_lds;
  int nwaves;
  int nbuf;          // in-tile buffers per wave (2 = LDS-DMA prefetch of the next tile)
  const MRun* runs;
  const int32_t* ocol;
};

struct PackRun {  // host description used by the pack kernel
  int cls, orig_row, count, wrow, kind;  // kind 0 scalar, 1 same-parity vector, 2 cross vector
};

struct Mfma {
  MfmaDev dev;
  std::vector<MRun> h_runs;
  std::vector<PackRun> h_pack;
  MRun* d_runs = nullptr;
  PackRun* d_pack = nullptr;
  MfmaDev* d_dev = nullptr;
  bool usable = false;
  size_t lds_bytes = 0;
};

constexpr int kLdsBudget = 160 * 1024;
int mfma_set_lds_attr();

int mfma_plan_init(e3_l1tp_plan* P) {
  auto* m = new Mfma();
  P->mfma = m;
  MfmaDev& d = m->dev;
  const PlanDev& p = P->dev;
  d.D1 = p.D1;
  d.Dout = p.Dout;
  d.SI = p.D1 | 1;
  d.SJ = (p.M[2] + p.M[3] > 0) ? 97 : 33;
  int woff = 0;
  for (int c = 0; c < 4; ++c) {
    d.M[c] = p.M[c];
    d.NT[c] = (p.M[c] + 31) / 32;
    d.Mpad[c] = d.NT[c] * 32;
    d.ocol_off[c] = p.ocol_off[c];
    // source classes: scalars / same-parity (or dot) vectors / cross vectors
    int src[3];
    if (c == 0) { src[0] = 0; src[1] = 3; src[2] = -1; }       // 0e <- s0e, <v1o,Y1>
    else if (c == 1) { src[0] = 1; src[1] = 2; src[2] = -1; }  // 0o <- s0o, <v1e,Y1>
    else if (c == 2) { src[0] = 1; src[1] = 2; src[2] = 3; }   // 1e <- s0o (x) Y1, v1e Y0, v1o x Y1
    else { src[0] = 0; src[1] = 3; src[2] = 2; }               // 1o <- s0e (x) Y1, v1o Y0, v1e x Y1
    int wrow = 0, orig = 0;
    for (int s = 0; s < 3; ++s) {
      d.roff[c][s] = (int)m->h_runs.size();
      d.nrun[c][s] = 0;
      if (src[s] < 0) continue;
      for (auto& r : P->irun[src[s]]) {
        if (p.M[c] > 0) {
          m->h_runs.push_back({r.col, r.count, r.cstride, wrow});
          m->h_pack.push_back({c, orig, r.count, wrow, s});
          d.nrun[c][s]++;
        }
        wrow += (r.count + 1) & ~1;
        orig += r.count;
      }
    }
    d.woff[c] = woff;
    if (p.M[c] > 0) woff += wrow * d.Mpad[c];
  }
  d.wtotal = woff;
  // LDS plan: [weights?][ocol table][nwaves x (nbuf in-tiles + job out-tile)]
  const size_t in_tile = ((size_t)32 * d.SI + 128) * 4, out_tile = (size_t)32 * d.SJ * 4;
  size_t tables = (size_t)((p.M[0] + p.M[1] + p.M[2] + p.M[3] + 15) & ~15) * 4 + (size_t)((p.Dout + 15) & ~15) * 4;
  size_t wbytes = (size_t)d.wtotal * 4;
  auto waves_for = [&](size_t fixed, int nbuf) -> int {
    size_t per_wave = in_tile * nbuf + out_tile;
    if (fixed + per_wave > (size_t)kLdsBudget) return 0;
    return (int)std::min<size_t>(((size_t)kLdsBudget - fixed) / per_wave, 8);
  };
  // preference: weights in LDS, then double buffering with >= 3 waves, else single buffer with more waves
  d.w_in_lds = waves_for(tables + wbytes, 1) >= 2 ? 1 : 0;
  size_t fixed = tables + (d.w_in_lds ? wbytes : 0);
  const int nw1 = waves_for(fixed, 1);
  // single buffering with as many waves as the LDS budget admits: measured faster than double buffering with fewer waves
  // (profiles/r01_l1tp_micro_*); the library reads no environment variables
  d.nbuf = 1;
  d.nwaves = nw1;
  m->usable = d.nwaves >= 1 && d.wtotal > 0;
  m->lds_bytes = fixed + (size_t)d.nwaves * (in_tile * d.nbuf + out_tile);
  return E3_OK;
}

int mfma_plan_upload(e3_l1tp_plan* P) {
  Mfma* m = P->mfma;
  if (!m || !m->usable) return E3_OK;
  size_t nr = std::max<size_t>(m->h_runs.size(), 1), np = std::max<size_t>(m->h_pack.size(), 1);
  E3_HIP_CHECK(hipMalloc((void**)&m->d_runs, nr * sizeof(MRun)));
  E3_HIP_CHECK(hipMalloc((void**)&m->d_pack, np * sizeof(PackRun)));
  if (!m->h_runs.empty())
    E3_HIP_CHECK(hipMemcpy(m->d_runs, m->h_runs.data(), m->h_runs.size() * sizeof(MRun), hipMemcpyHostToDevice));
  if (!m->h_pack.empty())
    E3_HIP_CHECK(hipMemcpy(m->d_pack, m->h_pack.data(), m->h_pack.size() * sizeof(PackRun), hipMemcpyHostToDevice));
  m->dev.runs = m->d_runs;
  m->dev.ocol = P->dev.ocol;
  E3_HIP_CHECK(hipMalloc((void**)&m->d_dev, sizeof(MfmaDev)));
  E3_HIP_CHECK(hipMemcpy(m->d_dev, &m->dev, sizeof(MfmaDev), hipMemcpyHostToDevice));
  return mfma_set_lds_attr();
}

void mfma_plan_free(e3_l1tp_plan* P) {
  if (!P->mfma) return;
  if (P->mfma->d_runs) (void)hipFree(P->mfma->d_runs);
  if (P->mfma->d_pack) (void)hipFree(P->mfma->d_pack);
  if (P->mfma->d_dev) (void)hipFree(P->mfma->d_dev);
  delete P->mfma;
  P->mfma = nullptr;
}

bool mfma_supported(const e3_l1tp_plan* P, int dtype) { return dtype == E3_F32 && P->mfma && P->mfma->usable; }

int64_t mfma_packed_bytes(const e3_l1tp_plan* P) {
  if (!P->mfma || !P->mfma->usable) return 0;
  return ((int64_t)(P->mfma->dev.wtotal + P->dev.Dout) * 4 + 255) / 256 * 256;
}

// packed = [W' (wtotal floats) | normcol (Dout floats)]
template <typename T>
__global__ void mfma_pack_kernel(const T* w0, const T* w1, const T* w2, const T* w3, const T* n0, const T* n1,
                                 const T* n2, const T* n3, float* packed, MfmaDev d, PlanDev p, const PackRun* pr,
                                 int npr) {
  const T* w[4] = {w0, w1, w2, w3};
  const T* nr[4] = {n0, n1, n2, n3};
  for (int r = blockIdx.x; r < npr; r += gridDim.x) {
    PackRun q = pr[r];
    const int M = d.M[q.cls], Mpad = d.Mpad[q.cls];
    const float cg = (q.cls < 2) ? (q.kind == 0 ? 1.0f : (float)kC3) : (q.kind == 2 ? (float)kC6 : (float)kC3);
    for (int i = threadIdx.x; i < q.count * M; i += blockDim.x) {
      int k = i / M, mm = i - k * M;
      packed[d.woff[q.cls] + (q.wrow + k) * Mpad + mm] = to_acc(w[q.cls][(int64_t)(q.orig_row + k) * M + mm]) * cg;
    }
  }
  if (blockIdx.x == 0) {
    for (int c = 0; c < 4; ++c) {
      int width = (c >= 2) ? 3 : 1;
      for (int i = threadIdx.x; i < p.M[c] * width; i += blockDim.x) {
        int mm = i / width, comp = i - mm * width;
        packed[d.wtotal + p.ocol[p.ocol_off[c] + mm] + comp] = nr[c] ? to_acc(nr[c][i]) : 1.0f;
      }
    }
  }
}

int mfma_pack(const e3_l1tp_plan* P, const void* const w[4], const void* const n[4], int dtype, void* packed,
              hipStream_t stream) {
  Mfma* m = P->mfma;
  if (!m || !m->usable) return E3_OK;
  E3_HIP_CHECK(hipMemsetAsync(packed, 0, (size_t)m->dev.wtotal * 4, stream));
  const void* nn[4] = {nullptr, nullptr, nullptr, nullptr};
  if (n)
    for (int c = 0; c < 4; ++c) nn[c] = P->normlen[c] > 0 ? n[c] : nullptr;
  int npr = (int)m->h_pack.size();
  int grid = std::max(1, std::min(npr, 256));
  if (dtype == E3_F32)
    hipLaunchKernelGGL(mfma_pack_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)w[0],
                       (const float*)w[1], (const float*)w[2], (const float*)w[3], (const float*)nn[0],
                       (const float*)nn[1], (const float*)nn[2], (const float*)nn[3], (float*)packed, m->dev, P->dev,
                       m->d_pack, npr);
  else
    hipLaunchKernelGGL(mfma_pack_kernel<bf16>, dim3(grid), dim3(256), 0, stream, (const bf16*)w[0],
                       (const bf16*)w[1], (const bf16*)w[2], (const bf16*)w[3], (const bf16*)nn[0], (const bf16*)nn[1],
                       (const bf16*)nn[2], (const bf16*)nn[3], (float*)packed, m->dev, P->dev, m->d_pack, npr);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

// -------------------------------------------------------------------------------------------------
// forward kernel
// -------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct YRow { float y0, yx, yy, yz; };

// One run of input channels contracted on the matrix core.  KIND selects how the B operand is built:
//   0: x*y0 (or x when y0 == 1 is passed)   -> acc A0
//   1: <v, Y1>                              -> acc A0
//   2: v_c * y0   (c = x,y,z)               -> acc A0,A1,A2
//   3: (v x Y1)_c                           -> acc A0,A1,A2
// `xp` points at this lane's row + run column (+ cstride*half), `wp` at W'[wrow+half][32t+j].
// Steps are issued in groups of U with the next group's LDS reads in flight behind the MFMAs.
template <int KIND, int U>
__device__ __forceinline__ void run_gemm(const float* __restrict__ xp, const float* __restrict__ wp, const int Mpad,
                                         const int count, const int half, const YRow y, f32x16& A0, f32x16& A1,
                                         f32x16& A2) {
  constexpr int NB = (KIND == 0) ? 1 : 3;
  constexpr int CS = (KIND == 0) ? 1 : 3;
  const int npair = count >> 1;
  const int ngrp = npair / U;
  auto load = [&](int p, float& a, float (&x)[NB]) {
    a = wp[(2 * p) * Mpad];
#pragma unroll
    for (int c = 0; c < NB; ++c) x[c] = xp[CS * 2 * p + c];
  };
  auto step = [&](float a, const float (&x)[NB], bool valid) {
    if (KIND == 0) {
      float b = x[0] * y.y0;
      A0 = mfma32(a, valid ? b : 0.0f, A0);
    } else if (KIND == 1) {
      float b = x[0] * y.yx + x[1] * y.yy + x[2] * y.yz;
      A0 = mfma32(a, valid ? b : 0.0f, A0);
    } else if (KIND == 2) {
      A0 = mfma32(a, valid ? x[0] * y.y0 : 0.0f, A0);
      A1 = mfma32(a, valid ? x[1] * y.y0 : 0.0f, A1);
      A2 = mfma32(a, valid ? x[2] * y.y0 : 0.0f, A2);
    } else {
      float bx = x[1] * y.yz - x[2] * y.yy;
      float by = x[2] * y.yx - x[0] * y.yz;
      float bz = x[0] * y.yy - x[1] * y.yx;
      A0 = mfma32(a, valid ? bx : 0.0f, A0);
      A1 = mfma32(a, valid ? by : 0.0f, A1);
      A2 = mfma32(a, valid ? bz : 0.0f, A2);
    }
  };
  if (ngrp > 0) {
    float a[U], x[U][NB];
#pragma unroll
    for (int u = 0; u < U; ++u) load(u, a[u], x[u]);
    for (int g = 1; g < ngrp; ++g) {
      float an[U], xn[U][NB];
#pragma unroll
      for (int u = 0; u < U; ++u) load(g * U + u, an[u], xn[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) step(a[u], x[u], true);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        a[u] = an[u];
#pragma unroll
        for (int c = 0; c < NB; ++c) x[u][c] = xn[u][c];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) step(a[u], x[u], true);
  }
  for (int p = ngrp * U; p < npair; ++p) {
    float a, x[NB];
    load(p, a, x);
    step(a, x, true);
  }
  if (count & 1) {  // odd tail: the second k of the pair is a zero weight row; its B lane must be a clean 0
    float a, x[NB];
    load(npair, a, x);
    step(a, x, half == 0);
  }
}

template <bool W_IN_LDS>
__global__ __launch_bounds__(512) void l1tp_fwd_mfma_kernel(const float* __restrict__ in1, int64_t ld1,
                                                            const float* __restrict__ in2, int64_t ld2,
                                                            const float* __restrict__ packed, float* __restrict__ out,
                                                            int64_t ldo, int64_t B, const MfmaDev* __restrict__ dp,
                                                            const MRun* __restrict__ runs,
                                                            const int32_t* __restrict__ ocol_tab) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31;     // row of the tile owned by this lane
  const int half = lane >> 5;  // k parity supplied by this lane
  const int D1 = dp->D1, Dout = dp->Dout, SI = dp->SI, SJ = dp->SJ, wtotal = dp->wtotal, nwaves = dp->nwaves,
            nbuf = dp->nbuf;
  // per-class constants, hoisted once (kept in SGPRs / lanes of a spill VGPR, not re-fetched per job)
  int cM[4], cMpad[4], cNT[4], cWoff[4], cOoff[4], cNrun[4][3], cRoff[4][3];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    cM[c] = dp->M[c]; cMpad[c] = dp->Mpad[c]; cNT[c] = dp->NT[c]; cWoff[c] = dp->woff[c]; cOoff[c] = dp->ocol_off[c];
#pragma unroll
    for (int s = 0; s < 3; ++s) { cNrun[c][s] = dp->nrun[c][s]; cRoff[c][s] = dp->roff[c][s]; }
  }
  const int Mtot = cM[0] + cM[1] + cM[2] + cM[3];

  // ---- LDS carve-up: [W'][normcol][ocol][per wave: nbuf x (in-tile + Y tile), one job out-tile] ----
  float* wl = lds;
  float* nrm = lds + (W_IN_LDS ? wtotal : 0);
  int* ocl = reinterpret_cast<int*>(nrm + ((Dout + 15) & ~15));
  float* tiles = reinterpret_cast<float*>(ocl + ((Mtot + 15) & ~15));
  const int in_sz = 32 * SI + 128;  // in-tile + [32][4] Y tile
  float* xbuf = tiles + (size_t)wave * (in_sz * nbuf + 32 * SJ);
  float* ot = xbuf + in_sz * nbuf;
  if (W_IN_LDS)
    for (int i = tid; i < wtotal; i += blockDim.x) wl[i] = packed[i];
  for (int i = tid; i < Dout; i += blockDim.x) nrm[i] = packed[wtotal + i];
  for (int i = tid; i < Mtot; i += blockDim.x) ocl[i] = ocol_tab[i];
  __syncthreads();
  const float* wsrc = W_IN_LDS ? wl : packed;

  const int64_t ntiles = (B + 31) / 32;
  const int64_t tstride = (int64_t)gridDim.x * nwaves;

  // Async stage of one 32-row tile by LDS-DMA (no VGPR round trip, nothing for the compiler to wait on):
  // in1 rows as 256-B segments, then the [32][4] Y tile (two pieces of 16 rows, per-lane source).
  auto stage = [&](int64_t tile, float* dst) {
    const int64_t row0 = tile * 32;
    const int nrows = (int)((B - row0) < 32 ? (B - row0) : 32);
    const int full = D1 & ~63;
    const float* srow = in1 + row0 * ld1 + lane;
    float* drow = dst;
    for (int r = 0; r < 32; ++r) {
      if (r < nrows) {
        for (int dc = 0; dc < full; dc += 64)
          __builtin_amdgcn_global_load_lds((glb_void_t*)(srow + dc), (lds_void_t*)(drow + dc), 4, 0, 0);
        if (full + lane < D1)
          __builtin_amdgcn_global_load_lds((glb_void_t*)(srow + full), (lds_void_t*)(drow + full), 4, 0, 0);
      } else {
        for (int dc = lane; dc < D1; dc += 64) drow[dc] = 0.0f;
      }
      srow += ld1;
      drow += SI;
    }
    float* ydst = dst + 32 * SI;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int yr = h * 16 + (lane >> 2);
      if (yr < nrows)
        __builtin_amdgcn_global_load_lds((glb_void_t*)(in2 + (row0 + yr) * ld2 + (lane & 3)),
                                         (lds_void_t*)(ydst + h * 64), 4, 0, 0);
      else
        ydst[h * 64 + lane] = 0.0f;
    }
  };

  int64_t tile = (int64_t)blockIdx.x * nwaves + wave;
  int cur = 0;
  if (tile < ntiles) stage(tile, xbuf);
  for (; tile < ntiles; tile += tstride) {
    const int64_t row0 = tile * 32;
    const int nrows = (int)((B - row0) < 32 ? (B - row0) : 32);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);  // the same wait, visible to the backend's wait-count pass (see e3_tp_mfma.hip)
    wave_sync_lds();
    const float* xt = xbuf + cur * in_sz;
    if (nbuf == 2 && tile + tstride < ntiles) stage(tile + tstride, xbuf + (cur ^ 1) * in_sz);
    const float* yp = xt + 32 * SI + 4 * j;
    const YRow y = {yp[0], yp[1], yp[2], yp[3]};
    const YRow yone = {1.0f, y.yx, y.yy, y.yz};
    const float* xr = xt + j * SI;  // this lane's row
    float* const obase = out + row0 * ldo;
    const uint32_t ldo32 = (uint32_t)ldo;

    // ---- scalar output classes (0e, 0o) ----
#pragma unroll
    for (int cls = 0; cls < 2; ++cls) {
      const int M = cM[cls];
      if (M == 0) continue;
      const int Mpad = cMpad[cls];
      const int* oc = ocl + cOoff[cls];
      for (int t = 0; t < cNT[cls]; ++t) {
        f32x16 acc = {0}, d1 = {0}, d2 = {0};
        const float* wbase = wsrc + cWoff[cls] + t * 32 + j + half * Mpad;
        for (int ri = 0; ri < cNrun[cls][0]; ++ri) {
          const MRun run = runs[cRoff[cls][0] + ri];
          run_gemm<0, 4>(xr + run.col + half, wbase + run.wrow * Mpad, Mpad, run.count, half, y, acc, d1, d2);
        }
        for (int ri = 0; ri < cNrun[cls][1]; ++ri) {
          const MRun run = runs[cRoff[cls][1] + ri];
          run_gemm<1, 4>(xr + run.col + 3 * half, wbase + run.wrow * Mpad, Mpad, run.count, half, y, acc, d1, d2);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[j * SJ + 8 * (r >> 2) + 4 * half + (r & 3)] = acc[r];
        wave_sync_lds();
        {  // copy out: two rows per store instruction (128-B segments)
          const int lc = lane & 31, rs = lane >> 5;
          const int ch = t * 32 + lc;
          if (ch < M) {
            const int col = oc[ch];
            const float nm = nrm[col];
            uint32_t off = (uint32_t)col + rs * ldo32;
            const float* src = ot + rs * SJ + lc;
#pragma unroll 4
            for (int r = rs; r < nrows; r += 2) {
              obase[off] = *src * nm;
              off += 2 * ldo32;
              src += 2 * SJ;
            }
          }
        }
        wave_sync_lds();
      }
    }
    // ---- vector output classes (1e, 1o) ----
#pragma unroll
    for (int cls = 2; cls < 4; ++cls) {
      const int M = cM[cls];
      if (M == 0) continue;
      const int Mpad = cMpad[cls];
      const int* oc = ocl + cOoff[cls];
      for (int t = 0; t < cNT[cls]; ++t) {
        const float* wbase = wsrc + cWoff[cls] + t * 32 + j + half * Mpad;
        f32x16 t0 = {0}, d1 = {0}, d2 = {0};
        for (int ri = 0; ri < cNrun[cls][0]; ++ri) {
          const MRun run = runs[cRoff[cls][0] + ri];
          run_gemm<0, 4>(xr + run.col + half, wbase + run.wrow * Mpad, Mpad, run.count, half, yone, t0, d1, d2);
        }
        f32x16 fx = t0 * y.yx, fy = t0 * y.yy, fz = t0 * y.yz;
        for (int ri = 0; ri < cNrun[cls][1]; ++ri) {
          const MRun run = runs[cRoff[cls][1] + ri];
          run_gemm<2, 2>(xr + run.col + 3 * half, wbase + run.wrow * Mpad, Mpad, run.count, half, y, fx, fy, fz);
        }
        for (int ri = 0; ri < cNrun[cls][2]; ++ri) {
          const MRun run = runs[cRoff[cls][2] + ri];
          run_gemm<3, 2>(xr + run.col + 3 * half, wbase + run.wrow * Mpad, Mpad, run.count, half, y, fx, fy, fz);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* o = ot + j * SJ + 3 * (8 * (r >> 2) + 4 * half + (r & 3));
          o[0] = fx[r]; o[1] = fy[r]; o[2] = fz[r];
        }
        wave_sync_lds();
        const int width = (M - t * 32 < 32 ? M - t * 32 : 32) * 3;
        for (int lc = lane; lc < width; lc += 64) {
          const int chl = lc / 3, comp = lc - 3 * chl;
          const int col = oc[t * 32 + chl] + comp;
          const float nm = nrm[col];
          uint32_t off = (uint32_t)col;
          const float* src = ot + lc;
#pragma unroll 8
          for (int r = 0; r < nrows; ++r) {
            obase[off] = *src * nm;
            off += ldo32;
            src += SJ;
          }
        }
        wave_sync_lds();
      }
    }
    if (nbuf == 1) {
      if (tile + tstride < ntiles) stage(tile + tstride, xbuf);
    } else {
      cur ^= 1;
    }
  }
}

int mfma_set_lds_attr() {
  E3_HIP_CHECK(hipFuncSetAttribute((const void*)l1tp_fwd_mfma_kernel<true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
  E3_HIP_CHECK(hipFuncSetAttribute((const void*)l1tp_fwd_mfma_kernel<false>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget));
  return E3_OK;
}

int mfma_forward(const e3_l1tp_plan* P, const void* in1, int64_t ld1, const void* in2, int64_t ld2,
                 const void* packed, void* out, int64_t ldo, int64_t B, int dtype, hipStream_t stream) {
  Mfma* m = P->mfma;
  if (!m || !m->usable || dtype != E3_F32) return E3_ERR_UNSUPPORTED;
  const MfmaDev& d = m->dev;
  int64_t ntiles = (B + 31) / 32;
  int grid = (int)std::min<int64_t>((ntiles + d.nwaves - 1) / d.nwaves, 256);
  if (d.w_in_lds)
    hipLaunchKernelGGL(l1tp_fwd_mfma_kernel<true>, dim3(grid), dim3(64 * d.nwaves), m->lds_bytes, stream,
                       (const float*)in1, ld1, (const float*)in2, ld2, (const float*)packed, (float*)out, ldo, B,
                       m->d_dev, m->d_runs, P->dev.ocol);
  else
    hipLaunchKernelGGL(l1tp_fwd_mfma_kernel<false>, dim3(grid), dim3(64 * d.nwaves), m->lds_bytes, stream,
                       (const float*)in1, ld1, (const float*)in2, ld2, (const float*)packed, (float*)out, ldo, B,
                       m->d_dev, m->d_runs, P->dev.ocol);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

}  // namespace e3
