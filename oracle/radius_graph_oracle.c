/* CPU oracle for the radius graph.  TEST INFRASTRUCTURE ONLY (see oracle/l1tp_oracle.py header).
 *
 * The reference mount holds no graph code: this is the CPU statement of the
 * repo's own contract written in include/e3gnn.h ("Radius graph"), "parity unpinned" w.r.t. the
 * upstream project.  Two independent searches are provided and cross-checked in tests:
 *   rg_bruteforce : all pairs, O(N^2)            (ground truth for small N)
 *   rg_celllist   : x-fastest linear cell list   (large N; shares only the edge predicate)
 * Both emit CSR-by-dst in NEW ids (rank under the stable Morton sort), src ascending per row.
 * Build: gcc -O2 -shared -fPIC -ffp-contract=off -o _build/libgraph_oracle.so radius_graph_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  float lo[3], hi[3];
  float r;
  int32_t n[3];
  float inv[3];
  int32_t bits;
} rg_params;

static uint32_t spread3(uint32_t v) {
  uint32_t out = 0;
  for (int b = 0; b < 10; ++b) out |= ((v >> b) & 1u) << (3 * b);
  return out;
}

int rg_grid(rg_params* p) {
  int nmax = 1;
  for (int a = 0; a < 3; ++a) {
    volatile float ext = p->hi[a] - p->lo[a];
    volatile float rr = p->r * 1.0001f;
    volatile float q = floorf(ext / rr);
    int n = q < 1.0f ? 1 : (q > 256.0f ? 256 : (int)q);
    p->n[a] = n;
    p->inv[a] = (float)n / ext;
    if (n > nmax) nmax = n;
  }
  int bits = 1;
  while ((1 << bits) < nmax) ++bits;
  p->bits = bits;
  return 0;
}

static int cell_of(float x, float lo, float inv, int n) {
  volatile float d = x - lo;
  volatile float t = d * inv;
  int c = (int)floorf(t);
  return c < 0 ? 0 : (c > n - 1 ? n - 1 : c);
}

static int cmp_u64(const void* a, const void* b) {
  uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return x < y ? -1 : (x > y);
}

/* perm[new] = old under the stable sort by Morton key; cells3 (optional) [N,3] cell coords in new order */
int rg_order(const float* pos, int64_t N, const rg_params* p, int32_t* perm, uint32_t* keys_sorted) {
  uint64_t* kv = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(N > 0 ? N : 1));
  for (int64_t i = 0; i < N; ++i) {
    int cx = cell_of(pos[3 * i], p->lo[0], p->inv[0], p->n[0]);
    int cy = cell_of(pos[3 * i + 1], p->lo[1], p->inv[1], p->n[1]);
    int cz = cell_of(pos[3 * i + 2], p->lo[2], p->inv[2], p->n[2]);
    uint32_t key = spread3(cx) | (spread3(cy) << 1) | (spread3(cz) << 2);
    kv[i] = ((uint64_t)key << 32) | (uint32_t)i;
  }
  qsort(kv, (size_t)N, sizeof(uint64_t), cmp_u64);
  for (int64_t i = 0; i < N; ++i) {
    perm[i] = (int32_t)(kv[i] & 0xffffffffu);
    if (keys_sorted) keys_sorted[i] = (uint32_t)(kv[i] >> 32);
  }
  free(kv);
  return 0;
}

static int within(const float* a, const float* b, float r2) {
  volatile float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  volatile float xx = dx * dx, yy = dy * dy, zz = dz * dz;
  volatile float s = xx + yy;
  volatile float d2 = s + zz;
  return d2 <= r2;
}

/* sp: positions in NEW order [N,3].  Pass src=NULL to only count (rowptr is always written). */
int64_t rg_bruteforce(const float* sp, int64_t N, float r, int32_t* rowptr, int32_t* src) {
  volatile float r2 = r * r;
  int64_t e = 0;
  for (int64_t i = 0; i < N; ++i) {
    rowptr[i] = (int32_t)e;
    for (int64_t j = 0; j < N; ++j)
      if (j != i && within(sp + 3 * i, sp + 3 * j, r2)) {
        if (src) src[e] = (int32_t)j;
        ++e;
      }
  }
  rowptr[N] = (int32_t)e;
  return e;
}

static int cmp_i32(const void* a, const void* b) {
  int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
  return x < y ? -1 : (x > y);
}

int64_t rg_celllist(const float* sp, int64_t N, const rg_params* p, int32_t* rowptr, int32_t* src) {
  volatile float r2 = p->r * p->r;
  const int nx = p->n[0], ny = p->n[1], nz = p->n[2];
  const int64_t nc = (int64_t)nx * ny * nz;
  int32_t* head = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nc + 1));
  int32_t* cell = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  memset(head, 0, sizeof(int32_t) * (size_t)(nc + 1));
  for (int64_t i = 0; i < N; ++i) {
    int cx = cell_of(sp[3 * i], p->lo[0], p->inv[0], nx);
    int cy = cell_of(sp[3 * i + 1], p->lo[1], p->inv[1], ny);
    int cz = cell_of(sp[3 * i + 2], p->lo[2], p->inv[2], nz);
    cell[i] = cx + nx * (cy + ny * cz);
    head[cell[i] + 1]++;
  }
  for (int64_t c = 0; c < nc; ++c) head[c + 1] += head[c];
  int32_t* fill = (int32_t*)malloc(sizeof(int32_t) * (size_t)(nc + 1));
  memcpy(fill, head, sizeof(int32_t) * (size_t)(nc + 1));
  for (int64_t i = 0; i < N; ++i) order[fill[cell[i]]++] = (int32_t)i;
  int32_t tmp[8192];
  int64_t e = 0;
  for (int64_t i = 0; i < N; ++i) {
    rowptr[i] = (int32_t)e;
    int c = cell[i];
    int cx = c % nx, cy = (c / nx) % ny, cz = c / (nx * ny);
    int cnt = 0;
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          int x = cx + dx, y = cy + dy, z = cz + dz;
          if (x < 0 || x >= nx || y < 0 || y >= ny || z < 0 || z >= nz) continue;
          int cc = x + nx * (y + ny * z);
          for (int32_t q = head[cc]; q < head[cc + 1]; ++q) {
            int32_t j = order[q];
            if (j != i && within(sp + 3 * i, sp + 3 * j, r2)) {
              if (cnt >= 8192) { free(head); free(cell); free(order); free(fill); return -1; }
              tmp[cnt++] = j;
            }
          }
        }
    qsort(tmp, (size_t)cnt, sizeof(int32_t), cmp_i32);
    if (src) memcpy(src + e, tmp, sizeof(int32_t) * (size_t)cnt);
    e += cnt;
  }
  rowptr[N] = (int32_t)e;
  free(head); free(cell); free(order); free(fill);
  return e;
}
