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

namespace e3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MRun {  // one run of in1 channels feeding one output class
  int col;      // first in1 column of the run
  int count;    // channels in the run
  int cstride;  // 1 (scalars) or 3 (vectors, xyz adjacent)
  int wrow;     // first packed weight row (runs are padded to an even number of rows)
};

struct MfmaDev {
  int D1, Dout, SI, SO;
  int M[4], NT[4], Mpad[4];
  int woff[4];       // float offset of the class's packed weights [Kpad][Mpad]
  int wtotal;        // floats of packed weights
  int nrun[4][3];    // per out class: #runs of {scalars, same-parity vectors, cross vectors}
  int roff[4][3];    // offsets into `runs`
  int ocol_off[4];
  int w_in_lds;
  int nwaves;
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
  d.SO = p.Dout | 1;
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
  // LDS plan: [weights?][ocol table][nwaves x (in tile + out tile)]
  size_t per_wave = (size_t)32 * (d.SI + d.SO) * 4;
  size_t tables = (size_t)(p.M[0] + p.M[1] + p.M[2] + p.M[3]) * 4 + 64;
  size_t wbytes = (size_t)d.wtotal * 4;
  d.w_in_lds = (wbytes + tables + 4 * per_wave <= (size_t)kLdsBudget) ? 1 : 0;
  size_t fixed = tables + (d.w_in_lds ? wbytes : 0);
  int nw = fixed + per_wave <= (size_t)kLdsBudget ? (int)(((size_t)kLdsBudget - fixed) / per_wave) : 0;
  nw = std::min(nw, 8);
  d.nwaves = nw;
  m->usable = nw >= 1 && d.wtotal > 0;
  m->lds_bytes = fixed + (size_t)nw * per_wave;
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
  return mfma_set_lds_attr();
}

void mfma_plan_free(e3_l1tp_plan* P) {
  if (!P->mfma) return;
  if (P->mfma->d_runs) (void)hipFree(P->mfma->d_runs);
  if (P->mfma->d_pack) (void)hipFree(P->mfma->d_pack);
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
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

template <bool W_IN_LDS>
__global__ __launch_bounds__(512) void l1tp_fwd_mfma_kernel(const float* __restrict__ in1, int64_t ld1,
                                                            const float* __restrict__ in2, int64_t ld2,
                                                            const float* __restrict__ packed, float* __restrict__ out,
                                                            int64_t ldo, int64_t B, MfmaDev d) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float* lds = reinterpret_cast<float*>(smem_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31;     // row of the tile owned by this lane
  const int half = lane >> 5;  // k parity supplied by this lane
  const int Mtot = d.M[0] + d.M[1] + d.M[2] + d.M[3];

  // ---- LDS carve-up: [W'][ocol][per wave: in tile, out tile] ----
  float* wl = lds;
  int* ocl = reinterpret_cast<int*>(lds + (W_IN_LDS ? d.wtotal : 0));
  float* tiles = reinterpret_cast<float*>(ocl + ((Mtot + 15) & ~15));
  float* xt = tiles + (size_t)wave * 32 * (d.SI + d.SO);
  float* ot = xt + 32 * d.SI;
  if (W_IN_LDS)
    for (int i = tid; i < d.wtotal; i += blockDim.x) wl[i] = packed[i];
  for (int i = tid; i < Mtot; i += blockDim.x) ocl[i] = d.ocol[i];
  __syncthreads();
  const float* wsrc = W_IN_LDS ? wl : packed;
  const float* normcol = packed + d.wtotal;

  const int64_t ntiles = (B + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * d.nwaves + wave; tile < ntiles; tile += (int64_t)gridDim.x * d.nwaves) {
    const int64_t row0 = tile * 32;
    const int nrows = (int)((B - row0) < 32 ? (B - row0) : 32);
    // ---- stage in1 tile (coalesced 256-B row segments) ----
    for (int dc = lane; dc < d.D1; dc += 64) {
      float v[32];
#pragma unroll
      for (int r = 0; r < 32; ++r) v[r] = (r < nrows) ? in1[(row0 + r) * ld1 + dc] : 0.0f;
#pragma unroll
      for (int r = 0; r < 32; ++r) xt[r * d.SI + dc] = v[r];
    }
    float y0 = 0.f, y1x = 0.f, y1y = 0.f, y1z = 0.f;
    if (j < nrows) {
      const float* yp = in2 + (row0 + j) * ld2;
      y0 = yp[0]; y1x = yp[1]; y1y = yp[2]; y1z = yp[3];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const float* xr = xt + j * d.SI;  // this lane's row

    // ---- scalar output classes (0e, 0o) ----
    for (int cls = 0; cls < 2; ++cls) {
      if (d.M[cls] == 0) continue;
      const int Mpad = d.Mpad[cls];
      for (int t = 0; t < d.NT[cls]; ++t) {
        f32x16 acc = {0};
        const float* wbase = wsrc + d.woff[cls] + t * 32 + j;
        for (int ri = 0; ri < d.nrun[cls][0]; ++ri) {
          const MRun run = d.runs[d.roff[cls][0] + ri];
          const float* wp = wbase + (size_t)(run.wrow + half) * Mpad;
#pragma unroll 4
          for (int k0 = 0; k0 < run.count; k0 += 2) {
            const int kk = k0 + half;
            float b = (kk < run.count) ? xr[run.col + kk] * y0 : 0.0f;
            float a = wp[(size_t)k0 * Mpad];
            acc = mfma32(a, b, acc);
          }
        }
        for (int ri = 0; ri < d.nrun[cls][1]; ++ri) {
          const MRun run = d.runs[d.roff[cls][1] + ri];
          const float* wp = wbase + (size_t)(run.wrow + half) * Mpad;
#pragma unroll 4
          for (int k0 = 0; k0 < run.count; k0 += 2) {
            const int kk = k0 + half;
            float b = 0.0f;
            if (kk < run.count) {
              const float* v = xr + run.col + 3 * kk;
              b = v[0] * y1x + v[1] * y1y + v[2] * y1z;
            }
            float a = wp[(size_t)k0 * Mpad];
            acc = mfma32(a, b, acc);
          }
        }
        const int* oc = ocl + d.ocol_off[cls];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int ch = t * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
          if (ch < d.M[cls]) ot[j * d.SO + oc[ch]] = acc[r];
        }
      }
    }
    // ---- vector output classes (1e, 1o) ----
    for (int cls = 2; cls < 4; ++cls) {
      if (d.M[cls] == 0) continue;
      const int Mpad = d.Mpad[cls];
      for (int t = 0; t < d.NT[cls]; ++t) {
        const float* wbase = wsrc + d.woff[cls] + t * 32 + j;
        f32x16 t0 = {0};
        for (int ri = 0; ri < d.nrun[cls][0]; ++ri) {
          const MRun run = d.runs[d.roff[cls][0] + ri];
          const float* wp = wbase + (size_t)(run.wrow + half) * Mpad;
#pragma unroll 4
          for (int k0 = 0; k0 < run.count; k0 += 2) {
            const int kk = k0 + half;
            float b = (kk < run.count) ? xr[run.col + kk] : 0.0f;
            t0 = mfma32(wp[(size_t)k0 * Mpad], b, t0);
          }
        }
        f32x16 fx = t0 * y1x, fy = t0 * y1y, fz = t0 * y1z;
        for (int ri = 0; ri < d.nrun[cls][1]; ++ri) {
          const MRun run = d.runs[d.roff[cls][1] + ri];
          const float* wp = wbase + (size_t)(run.wrow + half) * Mpad;
#pragma unroll 2
          for (int k0 = 0; k0 < run.count; k0 += 2) {
            const int kk = k0 + half;
            float bx = 0.f, by = 0.f, bz = 0.f;
            if (kk < run.count) {
              const float* v = xr + run.col + 3 * kk;
              bx = v[0] * y0; by = v[1] * y0; bz = v[2] * y0;
            }
            float a = wp[(size_t)k0 * Mpad];
            fx = mfma32(a, bx, fx);
            fy = mfma32(a, by, fy);
            fz = mfma32(a, bz, fz);
          }
        }
        for (int ri = 0; ri < d.nrun[cls][2]; ++ri) {
          const MRun run = d.runs[d.roff[cls][2] + ri];
          const float* wp = wbase + (size_t)(run.wrow + half) * Mpad;
#pragma unroll 2
          for (int k0 = 0; k0 < run.count; k0 += 2) {
            const int kk = k0 + half;
            float bx = 0.f, by = 0.f, bz = 0.f;
            if (kk < run.count) {
              const float* v = xr + run.col + 3 * kk;
              bx = v[1] * y1z - v[2] * y1y;  // (v x Y1)
              by = v[2] * y1x - v[0] * y1z;
              bz = v[0] * y1y - v[1] * y1x;
            }
            float a = wp[(size_t)k0 * Mpad];
            fx = mfma32(a, bx, fx);
            fy = mfma32(a, by, fy);
            fz = mfma32(a, bz, fz);
          }
        }
        const int* oc = ocl + d.ocol_off[cls];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int ch = t * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
          if (ch < d.M[cls]) {
            float* o = ot + j * d.SO + oc[ch];
            o[0] = fx[r]; o[1] = fy[r]; o[2] = fz[r];
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- copy out (coalesced), norm applied per output column ----
    for (int oc = lane; oc < d.Dout; oc += 64) {
      const float nm = normcol[oc];
#pragma unroll 8
      for (int r = 0; r < nrows; ++r) out[(row0 + r) * ldo + oc] = ot[r * d.SO + oc] * nm;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
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
                       (const float*)in1, ld1, (const float*)in2, ld2, (const float*)packed, (float*)out, ldo, B, d);
  else
    hipLaunchKernelGGL(l1tp_fwd_mfma_kernel<false>, dim3(grid), dim3(64 * d.nwaves), m->lds_bytes, stream,
                       (const float*)in1, ld1, (const float*)in2, ld2, (const float*)packed, (float*)out, ldo, B, d);
  E3_HIP_CHECK(hipGetLastError());
  return E3_OK;
}

}  // namespace e3
