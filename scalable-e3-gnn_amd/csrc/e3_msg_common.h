// Shared pieces of the fused message kernels (e3_msg_fused.hip: one wave per 16-edge tile; e3_msg_ws.hip: weights
// stationary in registers, one workgroup per tile stream).  Internal header, included inside namespace e3 after
// e3_tp_mfma_core.h and cg_tables.h.
#pragma once

// ------------------------------------------------------------------------------------------------------------------
// geometry of the operator, shared by host and device
// ------------------------------------------------------------------------------------------------------------------
template <int LMAX, int TT>
struct MsgGeom {
  static constexpr int H = 16 * TT;             // channels per degree
  static constexpr int KS = (H + 31) / 32;      // K = 32 steps per degree and segment
  static constexpr int D = H * (LMAX + 1) * (LMAX + 1);  // floats per feature row: [H x0e | H x1o | H x2e]
  static constexpr int T(int l3) { return l3 == 0 ? TT * (1 + LMAX) : TT; }  // 16-channel output tiles (l3 = 0: scalars + gates)
  static constexpr int col0(int l) { return H * l * l; }
  static constexpr bool ok(int l1, int l2, int l3) {
    return l1 >= 0 && l2 >= 0 && l3 >= 0 && l1 <= LMAX && l2 <= LMAX && l3 <= LMAX && ((l1 + l2 + l3) % 2 == 0) &&
           l3 >= (l1 > l2 ? l1 - l2 : l2 - l1) && l3 <= l1 + l2;
  }
  // paths in kernel order: l1 outer, then l3, then l2.  Weight block (path, ks, t) = 64 lanes x (16 B hi + 16 B lo).
  static constexpr int blk(int l1, int l2, int l3) {
    int n = 0;
    for (int a = 0; a <= LMAX; ++a)
      for (int c = 0; c <= LMAX; ++c)
        for (int b = 0; b <= LMAX; ++b) {
          if (a == l1 && b == l2 && c == l3) return n;
          if (ok(a, b, c)) n += KS * T(c);
        }
    return n;
  }
  static constexpr int nblk() { return blk(LMAX + 1, 0, 0); }
  // dst pre-mix table U [N][UD]: per path [a][t][16 channels]
  static constexpr int uoff(int l1, int l2, int l3) {
    int n = 0;
    for (int a = 0; a <= LMAX; ++a)
      for (int c = 0; c <= LMAX; ++c)
        for (int b = 0; b <= LMAX; ++b) {
          if (a == l1 && b == l2 && c == l3) return n;
          if (ok(a, b, c)) n += (2 * a + 1) * T(c) * 16;
        }
    return n;
  }
  static constexpr int UD = uoff(LMAX + 1, 0, 0);
  // accumulator slots (one f32x4 per lane each): l3 = 0: t; l3 = 1: T0 + 3 t + c; l3 = 2: T0 + 3 TT + 5 t + c
  static constexpr int slot0(int l3) { return l3 == 0 ? 0 : l3 == 1 ? T(0) : T(0) + 3 * TT; }
  static constexpr int NS = T(0) + 3 * TT + (LMAX == 2 ? 5 * TT : 0);
  // d-term weights: the distance channel of TP #1 couples through (0, l, l): [l][t][16]
  static constexpr int wdoff(int l) { return l == 0 ? 0 : l == 1 ? T(0) * 16 : (T(0) + TT) * 16; }
  static constexpr int WD = (T(0) + LMAX * TT) * 16;
  // packed buffer (floats): [header 64 | norm1 NS*16 | norm2 NS*16 | Wd | pad to 64 | W src1 | W tp2 | W dst1]
  static constexpr int o_norm1 = 64, o_norm2 = o_norm1 + NS * 16, o_wd = o_norm2 + NS * 16;
  static constexpr int o_w = (o_wd + WD + 63) / 64 * 64;
  static constexpr int blk_floats = 64 * 8;  // 64 lanes x (hi uint4 + lo uint4)
  static constexpr int64_t total_floats = (int64_t)o_w + 3LL * nblk() * blk_floats;
  static constexpr int lds_tab = 2 * NS * 16 + WD;  // floats of tables kept in LDS per workgroup
  static constexpr int lds_wave = 16 * (D + 4);     // floats per wave: staged h[src] rows, later the parked messages, then the
                                                    // transposed tile with row stride D + 4
};

// k slot jj (0..7) of k group g inside a 32-channel K step  <->  channel: the accumulator layout of the previous product
__host__ __device__ constexpr int kperm(int g, int jj) { return 16 * (jj >> 2) + 4 * g + (jj & 3); }


// ------------------------------------------------------------------------------------------------------------------
// device building blocks
// ------------------------------------------------------------------------------------------------------------------
template <int L1, int L2, int L3>
__device__ __forceinline__ void make_z(const float (&y)[9], float (&z)[2 * L1 + 1][2 * L3 + 1]) {
  using C = CG<L1, L2, L3>;
#pragma unroll
  for (int a = 0; a < 2 * L1 + 1; ++a)
#pragma unroll
    for (int c = 0; c < 2 * L3 + 1; ++c) {
      float s = 0.f;
      bool have = false;
#pragma unroll
      for (int b = 0; b < 2 * L2 + 1; ++b)
        if (C::v[a][b][c] != 0.0) {
          s = have ? __builtin_fmaf((float)C::v[a][b][c], y[L2 * L2 + b], s) : (float)C::v[a][b][c] * y[L2 * L2 + b];
          have = true;
        }
      z[a][c] = s;
    }
}
template <int L1, int L2, int L3>
__host__ __device__ constexpr bool z_nonzero(int a, int c) {
  for (int b = 0; b < 2 * L2 + 1; ++b)
    if (CG<L1, L2, L3>::v[a][b][c] != 0.0) return true;
  return false;
}

// one product group: fp32 storage = three f16 MFMAs on (hi, lo) halves (fp32-grade), bf16 storage = one bf16 MFMA
template <bool IO16>
__device__ __forceinline__ f32x4 mma3(const uint4 ah, const uint4 al, const uint4 bh, const uint4 bl, f32x4 c) {
  if constexpr (IO16) {
    return mfma16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), c);
  } else {
    c = mfma16h(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), c);
    c = mfma16h(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bl), c);
    c = mfma16h(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, bh), c);
    return c;
  }
}

// 8 fp32 values -> B operand(s): fp16 (hi, lo), or bf16 rounded once (exact when the values came from bf16 storage)
template <bool IO16>
__device__ __forceinline__ void split8(const float (&f)[8], uint4& bh, uint4& bl) {
  uint32_t ph[4], pl[4] = {0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if constexpr (IO16) ph[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{f[2 * q], f[2 * q + 1]}, bf16x2_t));
    else split2_f16(f[2 * q], f[2 * q + 1], ph[q], pl[q]);
  }
  bh = uint4{ph[0], ph[1], ph[2], ph[3]};
  bl = uint4{pl[0], pl[1], pl[2], pl[3]};
}

// real "component" spherical harmonics of the edge vector (same expressions as edge_geometry_l2_kernel, e3_edge_ops.hip)
__device__ __forceinline__ void edge_sh(const float4 ps, const float4 pd, float (&y)[9], float& dist) {
  const float rx = ps.x - pd.x, ry = ps.y - pd.y, rz = ps.z - pd.z;
  const float d = sqrtf(rx * rx + ry * ry + rz * rz);
  const float inv = d > 0.f ? 1.0f / d : 0.f;
  const float x = rx * inv, yy = ry * inv, z = rz * inv;
  constexpr float s3 = 1.7320508075688772f, s5 = 2.2360679774997896f;
  y[0] = 1.0f;
  y[1] = s3 * x; y[2] = s3 * yy; y[3] = s3 * z;
  y[4] = s5 * s3 * x * yy;
  y[5] = s5 * s3 * yy * z;
  y[6] = s5 * 0.5f * (2.f * z * z - x * x - yy * yy);
  y[7] = s5 * s3 * z * x;
  y[8] = s5 * 0.5f * s3 * (x * x - yy * yy);
  dist = d;
}
// l <= 1 variant: identical to edge_geometry_kernel (s = sqrt3 / d folded first)
__device__ __forceinline__ void edge_sh1(const float4 ps, const float4 pd, float (&y)[9], float& dist) {
  const float rx = ps.x - pd.x, ry = ps.y - pd.y, rz = ps.z - pd.z;
  const float d = sqrtf(rx * rx + ry * ry + rz * rz);
  const float s = d > 0.f ? 1.7320508075688772f / d : 0.f;
  y[0] = 1.0f; y[1] = s * rx; y[2] = s * ry; y[3] = s * rz;
#pragma unroll
  for (int q = 4; q < 9; ++q) y[q] = 0.f;
  dist = d;
}

// 4 consecutive channels x D1 components of a feature row (channel-major, component-minor) -> x[4 p + r][a], scaled.
// fp32 storage: D1 16-byte reads; bf16 storage: D1 8-byte reads, widened (exact).  `piece` points at channel 0 of the 4.
template <int D1, bool IO16>
__device__ __forceinline__ void read_piece(const void* piece, const int p, const float xs, float (&x)[8][D1]) {
  float q[4 * D1];
  if constexpr (IO16) {
    const uint2* sp = reinterpret_cast<const uint2*>(piece);
#pragma unroll
    for (int u = 0; u < D1; ++u) {
      const uint2 v = sp[u];
      q[4 * u + 0] = __builtin_bit_cast(float, v.x << 16); q[4 * u + 1] = __builtin_bit_cast(float, v.x & 0xffff0000u);
      q[4 * u + 2] = __builtin_bit_cast(float, v.y << 16); q[4 * u + 3] = __builtin_bit_cast(float, v.y & 0xffff0000u);
    }
  } else {
    const float4* sp = reinterpret_cast<const float4*>(piece);
#pragma unroll
    for (int u = 0; u < D1; ++u) {
      const float4 v = sp[u];
      q[4 * u] = v.x * xs; q[4 * u + 1] = v.y * xs; q[4 * u + 2] = v.z * xs; q[4 * u + 3] = v.w * xs;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int a = 0; a < D1; ++a) x[4 * p + r][a] = q[r * D1 + a];
}
template <int D1>
__device__ __forceinline__ void zero_piece(const int p, float (&x)[8][D1]) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int a = 0; a < D1; ++a) x[4 * p + r][a] = 0.f;
}

