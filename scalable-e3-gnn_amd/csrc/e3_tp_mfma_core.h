// Device-side building blocks shared by the MFMA tensor-product kernels (e3_tp_mfma.hip: one wave per 32-row tile;
// e3_tp_mfma_ab.hip: two waves per tile).  Internal header, included inside namespace e3.
#pragma once
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;

constexpr int kFastLds = 160 * 1024;
constexpr int kChunkFloats = 32 * 41 * 4;  // fp32 storage: 32 rows x 41 16-byte units (40 data units for 32 ch x 5 comps + 1 pad)
constexpr int kChunk16 = 32 * 21 * 4;      // bf16 storage: 32 rows x 21 units (>= 16 rows x 161 floats for the out tile)

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// Wait for every outstanding vector-memory operation (the LDS-DMA copies).  The asm is the compiler barrier; the builtin
// is the same instruction again in a form the backend's wait-count pass can see -- without it the pass believes the
// copies are still in flight and guards later LDS accesses with its own vmcnt(0), e.g. in every iteration of the
// epilogue's store loop (which then waits for the previous iteration's global store: ~650 cycles each).
__device__ __forceinline__ void wait_vm0() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0x0F70);  // gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15 (= no wait on those)
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
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

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
                                             const uint4* pre_h, const uint4* pre_l,  // A operands of k block 0 (preloaded)
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
  auto load = [&](int kb, uint4 (&ah)[NT], uint4 (&al)[NT], float (&x)[8][D1], bool first = false) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      ah[t] = first ? pre_h[t] : whi[(2 * kb) * Mpad + 32 * t];
      al[t] = first ? pre_l[t] : wlo[(2 * kb) * Mpad + 32 * t];
    }
    // this lane's 8 channels x D1 components are 2*D1 consecutive 16-byte units of its (16-byte aligned) row
    const float4* xv = reinterpret_cast<const float4*>(xp + 16 * kb * D1);
#pragma unroll
    for (int u = 0; u < 2 * D1; ++u) {
      const float4 v = xv[u];
      (&x[0][0])[4 * u + 0] = v.x; (&x[0][0])[4 * u + 1] = v.y; (&x[0][0])[4 * u + 2] = v.z; (&x[0][0])[4 * u + 3] = v.w;
    }
  };
  // Features are built for two output components at a time on the packed-fp32 pipe (v_pk_fma_f32: z pair x broadcast
  // x), which halves the VALU count of the contraction with z; the hi/lo split works on the same pairs.
  auto compute = [&](int kb, const uint4 (&ah)[NT], const uint4 (&al)[NT], const float (&x)[8][D1]) {
    (void)kb;
#pragma unroll
    for (int c = 0; c < NB; c += 2) {
      const bool pr = c + 1 < NB;
      f32x2_t b2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x2_t b = {0.f, 0.f};
        if (MIX) {
          b.x = x[i][c];
          if (pr) b.y = x[i][c + 1];
        } else {
          bool hx = false, hy = false;
#pragma unroll
          for (int m = 0; m < D1; ++m) {
            bool nza = false, nzb = false;
#pragma unroll
            for (int q = 0; q < D2; ++q) {
              nza |= (C::v[m][q][c] != 0.0);
              if (pr) nzb |= (C::v[m][q][c + 1 < D3 ? c + 1 : c] != 0.0);
            }
            if (nza && nzb) {
              const f32x2_t z2 = {z[m][c], z[m][c + 1 < D3 ? c + 1 : c]}, xx = {x[i][m], x[i][m]};
              b = (hx || hy) ? __builtin_elementwise_fma(z2, xx, b) : z2 * xx;
              hx = hy = true;
            } else if (nza) {
              b.x = hx ? __builtin_fmaf(z[m][c], x[i][m], b.x) : z[m][c] * x[i][m];
              hx = true;
            } else if (nzb) {
              b.y = hy ? __builtin_fmaf(z[m][c + 1 < D3 ? c + 1 : c], x[i][m], b.y) : z[m][c + 1 < D3 ? c + 1 : c] * x[i][m];
              hy = true;
            }
          }
        }
        b2[i] = b;  // channels beyond `count` were staged as zeros (and their weight rows are zero)
      }
      // hi/lo split: hi parts packed two channels at a time (v_cvt_pk), unpacked by shift / mask, residuals on
      // the packed pipe (one v_pk_add per channel for both components), then packed again
      uint32_t ph[2][4], pl[2][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ph[0][q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{b2[2 * q].x, b2[2 * q + 1].x}, bf16x2_t));
        ph[1][q] = pr ? __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{b2[2 * q].y, b2[2 * q + 1].y}, bf16x2_t)) : 0u;
      }
      f32x2_t l2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const uint32_t wa = ph[0][i >> 1], wb = ph[1][i >> 1];
        const f32x2_t h2 = {__builtin_bit_cast(float, (i & 1) ? (wa & 0xffff0000u) : (wa << 16)),
                            __builtin_bit_cast(float, (i & 1) ? (wb & 0xffff0000u) : (wb << 16))};
        l2[i] = b2[i] - h2;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pl[0][q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{l2[2 * q].x, l2[2 * q + 1].x}, bf16x2_t));
        pl[1][q] = pr ? __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{l2[2 * q].y, l2[2 * q + 1].y}, bf16x2_t)) : 0u;
      }
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        if (cc == 1 && !pr) break;
        const bf16x8 bh = __builtin_bit_cast(bf16x8, uint4{ph[cc][0], ph[cc][1], ph[cc][2], ph[cc][3]});
        const bf16x8 bl = __builtin_bit_cast(bf16x8, uint4{pl[cc][0], pl[cc][1], pl[cc][2], pl[cc][3]});
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const bf16x8 wh = __builtin_bit_cast(bf16x8, ah[t]);
          const bf16x8 wl = __builtin_bit_cast(bf16x8, al[t]);
          f32x16& dst = MIX ? T[t][c + cc] : acc[t][c + cc < D3 ? c + cc : c];
          dst = mfma_bf16(wh, bh, dst);
          dst = mfma_bf16(wh, bl, dst);
          dst = mfma_bf16(wl, bh, dst);
        }
      }
    }
  };
  {
    uint4 ah[NT], al[NT];
    float x[8][D1];
    load(0, ah, al, x, true);
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
                                               const uint4* __restrict__ whi, const uint4* pre_h, const int Mpad, const int half,
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
  auto load = [&](int kb, uint4 (&ah)[NT], uint32_t (&q)[NQ], bool first = false) {
#pragma unroll
    for (int t = 0; t < NT; ++t) ah[t] = first ? pre_h[t] : whi[(2 * kb) * Mpad + 32 * t];
    const uint4* xv = reinterpret_cast<const uint4*>(xp + 8 * kb * D1);  // D1 consecutive 16-byte units
#pragma unroll
    for (int u = 0; u < D1; ++u) {
      const uint4 v = xv[u];
      q[4 * u + 0] = v.x; q[4 * u + 1] = v.y; q[4 * u + 2] = v.z; q[4 * u + 3] = v.w;
    }
  };
  auto compute = [&](const uint4 (&ah)[NT], const uint32_t (&q)[NQ]) {
    float x[8][D1];
#pragma unroll
    for (int e = 0; e < 8 * D1; ++e) {
      const uint32_t w = q[e >> 1];
      x[e / D1][e % D1] = __builtin_bit_cast(float, (e & 1) ? (w & 0xffff0000u) : (w << 16));
    }
#pragma unroll
    for (int c = 0; c < D3; c += 2) {  // two output components at a time on the packed-fp32 pipe (see run_steps_bf)
      const bool pr = c + 1 < D3;
      f32x2_t b2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f32x2_t b = {0.f, 0.f};
        bool hx = false, hy = false;
#pragma unroll
        for (int m = 0; m < D1; ++m) {
          bool nza = false, nzb = false;
#pragma unroll
          for (int qq = 0; qq < D2; ++qq) {
            nza |= (C::v[m][qq][c] != 0.0);
            if (pr) nzb |= (C::v[m][qq][c + 1 < D3 ? c + 1 : c] != 0.0);
          }
          if (nza && nzb) {
            const f32x2_t z2 = {z[m][c], z[m][c + 1 < D3 ? c + 1 : c]}, xx = {x[i][m], x[i][m]};
            b = (hx || hy) ? __builtin_elementwise_fma(z2, xx, b) : z2 * xx;
            hx = hy = true;
          } else if (nza) {
            b.x = hx ? __builtin_fmaf(z[m][c], x[i][m], b.x) : z[m][c] * x[i][m];
            hx = true;
          } else if (nzb) {
            b.y = hy ? __builtin_fmaf(z[m][c + 1 < D3 ? c + 1 : c], x[i][m], b.y) : z[m][c + 1 < D3 ? c + 1 : c] * x[i][m];
            hy = true;
          }
        }
        b2[i] = b;
      }
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        if (cc == 1 && !pr) break;
        uint32_t pk[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          pk[qq] = __builtin_bit_cast(uint32_t, __builtin_convertvector(
                       cc ? f32x2_t{b2[2 * qq].y, b2[2 * qq + 1].y} : f32x2_t{b2[2 * qq].x, b2[2 * qq + 1].x}, bf16x2_t));
        const bf16x8 bh = __builtin_bit_cast(bf16x8, uint4{pk[0], pk[1], pk[2], pk[3]});
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          f32x16& dst = acc[t][c + cc < D3 ? c + cc : c];
          dst = mfma_bf16(__builtin_bit_cast(bf16x8, ah[t]), bh, dst);
        }
      }
    }
  };
  {
    uint4 ah[NT];
    uint32_t q[NQ];
    load(0, ah, q, true);
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

// "Lean" run loops for the two-waves-per-SIMD kernel (e3_tp_mfma_ab.hip): same arithmetic as run_steps_bf /
// run_steps_io16, but sized for 256 registers per lane -- the x operands are read right before use (the second wave on
// the SIMD covers the LDS round trip), features are built one output component at a time; only the A operands
// (weights, an L2 round trip away) are still fetched one k block ahead.
template <int L1, int L2, int L3, int NT>
__device__ __forceinline__ void run_steps_bf_lean(const float* __restrict__ xr, const int count,
                                                  const uint4* __restrict__ whi, const uint4* __restrict__ wlo,
                                                  const uint4* pre_h, const uint4* pre_l, const int Mpad, const int half,
                                                  const float (&y)[9], f32x16 (&acc)[NT][2 * L3 + 1]) {
  constexpr int D1 = 2 * L1 + 1, D2 = 2 * L2 + 1, D3 = 2 * L3 + 1;
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
  const float* xp = xr + 8 * half * D1;
  const int nkb = (count + 15) >> 4;
  uint4 ah[NT], al[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { ah[t] = pre_h[t]; al[t] = pre_l[t]; }
  for (int kb = 0; kb < nkb; ++kb) {
    uint4 ahn[NT], aln[NT];
    if (kb + 1 < nkb) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        ahn[t] = whi[(2 * (kb + 1)) * Mpad + 32 * t];
        aln[t] = wlo[(2 * (kb + 1)) * Mpad + 32 * t];
      }
    }
    float x[8][D1];
    const float4* xv = reinterpret_cast<const float4*>(xp + 16 * kb * D1);
#pragma unroll
    for (int u = 0; u < 2 * D1; ++u) {
      const float4 v = xv[u];
      (&x[0][0])[4 * u + 0] = v.x; (&x[0][0])[4 * u + 1] = v.y; (&x[0][0])[4 * u + 2] = v.z; (&x[0][0])[4 * u + 3] = v.w;
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
        f[i] = b;
      }
      uint32_t ph[4], pl[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        ph[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{f[2 * q], f[2 * q + 1]}, bf16x2_t));
        const float h0 = __builtin_bit_cast(float, ph[q] << 16), h1 = __builtin_bit_cast(float, ph[q] & 0xffff0000u);
        pl[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{f[2 * q] - h0, f[2 * q + 1] - h1}, bf16x2_t));
      }
      const bf16x8 bh = __builtin_bit_cast(bf16x8, uint4{ph[0], ph[1], ph[2], ph[3]});
      const bf16x8 bl = __builtin_bit_cast(bf16x8, uint4{pl[0], pl[1], pl[2], pl[3]});
      // product-major order: consecutive MFMAs write different accumulators (the three products of one tile are a
      // dependent chain)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][c] = mfma_bf16(__builtin_bit_cast(bf16x8, ah[t]), bh, acc[t][c]);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][c] = mfma_bf16(__builtin_bit_cast(bf16x8, ah[t]), bl, acc[t][c]);
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][c] = mfma_bf16(__builtin_bit_cast(bf16x8, al[t]), bh, acc[t][c]);
    }
    if (kb + 1 < nkb) {
#pragma unroll
      for (int t = 0; t < NT; ++t) { ah[t] = ahn[t]; al[t] = aln[t]; }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <int L1, int L2, int L3, int NT>
__device__ __forceinline__ void run_steps_io16_lean(const uint32_t* __restrict__ xr32, const int count,
                                                    const uint4* __restrict__ whi, const uint4* pre_h, const int Mpad,
                                                    const int half, const float (&y)[9], f32x16 (&acc)[NT][2 * L3 + 1]) {
  constexpr int D1 = 2 * L1 + 1, D2 = 2 * L2 + 1, D3 = 2 * L3 + 1;
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
  uint4 ah[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) ah[t] = pre_h[t];
  if constexpr (L1 == 0 && D3 > 1) {
    // scalar input channels into a vector output: contract the raw bf16 channels once (they ARE the B operand: no
    // VALU at all in the k loop) into a temporary tile, fold with z afterwards
    f32x16 T[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) T[t] = f32x16{0};
    for (int kb = 0; kb < nkb; ++kb) {
      const uint4 xv = *reinterpret_cast<const uint4*>(xp + 8 * kb);
      const bf16x8 bh = __builtin_bit_cast(bf16x8, xv);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        T[t] = mfma_bf16(__builtin_bit_cast(bf16x8, ah[t]), bh, T[t]);
        if (kb + 1 < nkb) ah[t] = whi[(2 * (kb + 1)) * Mpad + 32 * t];
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int c = 0; c < D3; ++c) acc[t][c] += T[t] * z[0][c];
    __builtin_amdgcn_sched_barrier(0);
    return;
  }
  for (int kb = 0; kb < nkb; ++kb) {
    uint4 ahn[NT];
    if (kb + 1 < nkb) {
#pragma unroll
      for (int t = 0; t < NT; ++t) ahn[t] = whi[(2 * (kb + 1)) * Mpad + 32 * t];
    }
    float x[8][D1];
    const uint4* xv = reinterpret_cast<const uint4*>(xp + 8 * kb * D1);
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
        f[i] = b;
      }
      uint32_t pk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        pk[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{f[2 * q], f[2 * q + 1]}, bf16x2_t));
      const bf16x8 bh = __builtin_bit_cast(bf16x8, uint4{pk[0], pk[1], pk[2], pk[3]});
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t][c] = mfma_bf16(__builtin_bit_cast(bf16x8, ah[t]), bh, acc[t][c]);
    }
    if (kb + 1 < nkb) {
#pragma unroll
      for (int t = 0; t < NT; ++t) ah[t] = ahn[t];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// Compile-time bookkeeping for the weight preload: the (l2, l3) paths of an input chunk of degree l1 in the order the
// kernel runs them (l3 outer, l2 inner), each owning nt(l3) consecutive slots of the preload registers.
template <int LSH, int NT0, int NT1, int NT2>
struct PathSlots {
  static constexpr int nt(int l3) { return l3 == 0 ? NT0 : l3 == 1 ? NT1 : NT2; }
  static constexpr bool valid(int l1, int l2, int l3) {
    return l1 >= 0 && nt(l3) > 0 && l2 <= LSH && ((l1 + l2 + l3) % 2 == 0) && l3 >= (l1 > l2 ? l1 - l2 : l2 - l1) &&
           l3 <= l1 + l2;
  }
  static constexpr int slot(int l1, int l2, int l3) {
    int o = 0;
    for (int c = 0; c < 3; ++c)
      for (int b = 0; b < 3; ++b) {
        if (c == l3 && b == l2) return o;
        if (valid(l1, b, c)) o += nt(c);
      }
    return o;
  }
  static constexpr int total(int l1) { return slot(l1, 3, 3); }
  static constexpr int max_total() {
    int m = 1;
    for (int l1 = 0; l1 < 3; ++l1) m = total(l1) > m ? total(l1) : m;
    return m;
  }
};
template <int... V>
struct IntSeq {
  static constexpr int n = sizeof...(V);
  static constexpr int at(int i) {
    constexpr int a[] = {V...};
    return (i >= 0 && i < n) ? a[i] : -1;
  }
};
template <class F, size_t... I>
__device__ __forceinline__ void for_each_index(F&& f, std::index_sequence<I...>) {
  (f(std::integral_constant<int, (int)I>{}), ...);
}

struct SegArgs {
  const void* base[4];
  int64_t ld[4];
  const int32_t* index[4];
  int col0[5];  // first in1 column of each segment; col0[nseg] = D1
  int nseg;
  const int32_t* scatter;  // fused segment-sum (e3_tp_forward_fused_scatter): node id of every row, ascending; else null
};

__device__ __forceinline__ float sigmoid_(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

