/*
 * e3gnn.h — C ABI of libe3gnn_hip.so, the MI355X (gfx950) implementation of the Scalable-E3-GNN
 * hot path.  Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * Every entry point returns an int status (E3_OK == 0); nothing throws across the ABI and nothing
 * allocates device memory behind the caller's back except the small per-plan tables created in
 * e3_l1tp_plan_create.  All kernels are enqueued on the caller's hipStream_t (passed as void*),
 * never synchronise, and are graph-capturable.
 *
 * Reference interface replaced (file:line in /root/reference/models/segnn/l1_tensor_prod.py):
 *   e3_l1tp_plan_create   <- L1TensorProduct.__init__ irreps partition, masks, counts   (:13-77)
 *   e3_l1tp_pack_weights  <- parameters + CG constants + norm buffers                   (:81-94,159-189)
 *   e3_l1tp_forward       <- L1TensorProduct.forward                                    (:234-299)
 *   e3_l1tp_backward      <- autograd of forward (reference relies on torch autograd)    (:234-299)
 * Builder-defined stages of the pipeline (no reference code in the mount) are
 * declared further below and say so.
 */
#ifndef E3GNN_H
#define E3GNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define E3GNN_ABI_VERSION 3

/* status codes */
enum {
  E3_OK = 0,
  E3_ERR_INVALID_ARG = 1,   /* null pointer, negative size, bad dtype ...                     */
  E3_ERR_BAD_IRREPS = 2,    /* l > 1, lmax != 1, parity not +-1, negative multiplicity          */
  E3_ERR_MISSING_WEIGHT = 3,/* an output class has columns but its weight pointer is NULL        */
  E3_ERR_UNSUPPORTED = 4,   /* shape exceeds what the kernels were built for                    */
  E3_ERR_HIP = 5,           /* a HIP runtime call failed; see e3_last_hip_error()              */
  E3_ERR_NO_DEVICE = 6
};

/* element types of in1 / in2 / weights / norms / out (all the same in one call) */
enum { E3_F32 = 0, E3_F64 = 1, E3_BF16 = 2 };

/* class index used for the 4-element pointer arrays below: l0e, l0o, l1e, l1o */
enum { E3_CLS_0E = 0, E3_CLS_0O = 1, E3_CLS_1E = 2, E3_CLS_1O = 3 };

typedef struct e3_l1tp_plan e3_l1tp_plan;

int         e3_abi_version(void);
const char* e3_status_string(int status);
const char* e3_last_hip_error(void);

/*
 * Plan = host+device description of one (in1_irreps -> out_irreps) tensor product with the fixed
 * second operand 1x0e+1x1o (spherical harmonics of an edge vector).
 * blocks: n x 3 int32, each (l, p, mul) with l in {0,1}, p in {+1,-1}, mul >= 0, in declaration
 * order (irreps are NOT sorted or merged: the column layout follows the declaration, :28-36,57-65).
 * Requires max l == 1 for both (the reference's asserts, :13-14).
 */
int e3_l1tp_plan_create(const int32_t* in1_blocks, int n_in1,
                        const int32_t* out_blocks, int n_out,
                        e3_l1tp_plan** plan);
int e3_l1tp_plan_destroy(e3_l1tp_plan* plan);

/* Shape queries (so a binding needs no second implementation of the bookkeeping). */
int e3_l1tp_in1_dim(const e3_l1tp_plan* plan);
int e3_l1tp_out_dim(const e3_l1tp_plan* plan);
/* rows/cols of weight matrix `cls` exactly as the reference allocates it (:81-88); 0,0 if absent */
int e3_l1tp_weight_shape(const e3_l1tp_plan* plan, int cls, int* rows, int* cols);
/* length of norm buffer `cls` (:159-162) */
int e3_l1tp_norm_len(const e3_l1tp_plan* plan, int cls);

/*
 * Packed weights: one device buffer holding the four weight matrices re-tiled for the MFMA
 * kernel (K padded per irreps block, 32-column tiles, CG constants 1/sqrt3, 1/sqrt6 folded in)
 * plus the per-output-column norm vector.  Size in bytes for a given dtype:
 */
int64_t e3_l1tp_packed_bytes(const e3_l1tp_plan* plan, int dtype);
/*
 * weights[cls], norms[cls]: device pointers of element type `dtype`, row-major contiguous, shapes
 * per e3_l1tp_weight_shape / e3_l1tp_norm_len.  weights[cls] may be NULL only when the class has
 * no output columns or no input rows.  norms may be NULL (module built without normalisation)
 * or hold NULL entries for empty classes.  `packed` must hold e3_l1tp_packed_bytes().
 */
int e3_l1tp_pack_weights(const e3_l1tp_plan* plan,
                         const void* const weights[4], const void* const norms[4],
                         int dtype, void* packed, void* stream);

/*
 * out[b, :] = L1TP(in1[b, :], in2[b, :])   for b in [0, B)
 *   in1 : [B, in1_dim]   row stride ld_in1 elements
 *   in2 : [B, 4] = [Y0, Y1x, Y1y, Y1z], row stride ld_in2 elements; ld_in2 == 0 broadcasts row 0
 *   out : [B, out_dim]   row stride ld_out; every column is written
 * `kernel`: 0 = auto, 1 = force the generic kernel, 2 = force the MFMA kernel (E3_ERR_UNSUPPORTED
 * if the plan / dtype cannot use it).
 */
int e3_l1tp_forward(const e3_l1tp_plan* plan,
                    const void* in1, int64_t ld_in1,
                    const void* in2, int64_t ld_in2,
                    const void* packed,
                    void* out, int64_t ld_out,
                    int64_t B, int dtype, int kernel, void* stream);

/*
 * Gradients of sum(out * grad_out).  grad_in1 [B,in1_dim] and grad_in2 [B,4] are overwritten;
 * grad_weights[cls] (same shapes as weights) are overwritten.  When in2 was broadcast
 * (ld_in2 == 0) grad_in2 is a single row [1,4].  `workspace` must hold
 * e3_l1tp_backward_workspace_bytes() bytes.  Any of grad_in1 / grad_in2 / grad_weights may be
 * NULL to skip that gradient.
 */
int64_t e3_l1tp_backward_workspace_bytes(const e3_l1tp_plan* plan, int64_t B, int dtype);
int e3_l1tp_backward(const e3_l1tp_plan* plan,
                     const void* in1, int64_t ld_in1,
                     const void* in2, int64_t ld_in2,
                     const void* const weights[4], const void* const norms[4],
                     const void* grad_out, int64_t ld_gout,
                     void* grad_in1, int64_t ld_gin1,
                     void* grad_in2,
                     void* const grad_weights[4],
                     void* workspace,
                     int64_t B, int dtype, void* stream);

/* =================================================================================================
 * Radius graph (builder-defined: the reference mount has no graph code; the spec
 * below is this repo's contract and oracle/radius_graph_oracle.c is its CPU statement).
 *
 *   grid   : per axis a,  n_a = clamp(floor((hi_a-lo_a) / (r*1.0001f)), 1, 256)   (fp32 arithmetic)
 *            cell c_a(p) = clamp((int)floorf((p_a-lo_a) * (n_a/(hi_a-lo_a))), 0, n_a-1)
 *   key    : 30-bit Morton interleave of (cx,cy,cz) (x lowest bit)
 *   order  : stable sort by key; new id = rank; perm[new] = old
 *   edge   : (src=j -> dst=i), i != j, iff d2 <= fl32(r*r) with
 *            d2 = fl32(fl32(fl32(dx*dx)+fl32(dy*dy))+fl32(dz*dz)), dx = fl32(x_i-x_j) (no FMA contraction)
 *   output : CSR by dst in NEW ids: rowptr[N+1] (int32), src[E] (int32) ascending inside each row.
 * All integer outputs are bit-exact functions of the inputs.
 *
 * Call sequence (sizes are only known after the count pass, so the caller allocates `src`):
 *   e3_rg_workspace_bytes -> e3_rg_sort_count (fills perm, sorted_pos4, rowptr) -> read E = rowptr[N]
 *   -> e3_rg_fill (fills src).
 * ================================================================================================= */
typedef struct e3_rg_params {
  float lo[3], hi[3];
  float r;
  int32_t n[3];   /* filled by e3_rg_grid */
  float inv[3];   /* filled by e3_rg_grid */
  int32_t bits;   /* filled by e3_rg_grid: Morton bits per axis */
} e3_rg_params;

int     e3_rg_grid(e3_rg_params* prm);                     /* host only: derives n, inv, bits from lo/hi/r */
int64_t e3_rg_workspace_bytes(int64_t N, const e3_rg_params* prm);
/* pos [N,3] fp32 contiguous (device). Outputs (device): perm [N] int32, sorted_pos4 [N,4] fp32
 * (x,y,z,0 in new order), rowptr [N+1] int32. */
int e3_rg_sort_count(const float* pos, int64_t N, const e3_rg_params* prm,
                     int32_t* perm, float* sorted_pos4, int32_t* rowptr,
                     void* workspace, int64_t workspace_bytes, void* stream);
/* src [E] int32 (device), E = rowptr[N]; must follow e3_rg_sort_count with the same workspace. */
int e3_rg_fill(int64_t N, const e3_rg_params* prm, const float* sorted_pos4, const int32_t* rowptr,
               int32_t* src, void* workspace, int64_t workspace_bytes, void* stream);

/* =================================================================================================
 * Edge / node stages of the SEGNN forward around the tensor product (builder-defined, SURVEY.md
 * §8a-N2, N3; fp32).  Graph = CSR by dst from e3_rg_* (rowptr [N+1], src [E], positions pos4 [N,4]).
 *   rel_e  = x[src_e] - x[dst_e];  d_e = |rel_e|
 *   Y_e    = [1, sqrt(3) rel_e/d_e]   ("component" normalised real SH, l<=1, xyz order; Y1 = 0 if d_e = 0)
 *   A_i    = [1, mean_{e -> i} Y1_e]  (node attribute; [1,0,0,0] for isolated nodes)
 * ================================================================================================= */
/* edge_y [E,4], edge_d [E] (may be NULL), node_a [N,4] (may be NULL) */
int e3_edge_geometry(const float* pos4, const int32_t* rowptr, const int32_t* src, int64_t N,
                     float* edge_y, float* edge_d, float* node_a, void* stream);
/* out[e] = [ h[dst_e] (D) | h[src_e] (D) | extra[e] (n_extra, may be 0) ]   row strides in elements */
int e3_gather_concat(const float* h, int64_t ld_h, int D, const int32_t* rowptr, const int32_t* src, int64_t N,
                     const float* extra, int n_extra, float* out, int64_t ld_out, void* stream);
/* gate: in = [ns scalars | nv gate scalars | nv vectors (xyz adjacent)] -> out = [silu(s) | sigmoid(g)*v] */
int e3_gate(const float* in, int64_t ld_in, float* out, int64_t ld_out, int64_t B, int ns, int nv, void* stream);
/* agg[i] = sum over row i of msg[e] (fixed edge order => bitwise reproducible); D columns */
int e3_segment_sum(const float* msg, int64_t ld_msg, const int32_t* rowptr, int64_t N, int D,
                   float* agg, int64_t ld_agg, void* stream);

/* Sharded graph (sharding.GridHalo.split_graph): from the CSR graph of the local cloud (owned + ghost rows) and the per-row
 * ghost flags (uint8), drop the edges INTO ghost rows and split the rest by the ownership of their src -- classification and
 * compaction in one call, every output sorted by dst:
 *   rowptr_kept [N+1], (src_kept, dst_kept) [E_kept]: the graph without the dropped rows' edges
 *   (src_interior, dst_interior): kept edges with an owned src;  (src_boundary, dst_boundary): kept edges with a ghost src
 *   counts[0..2] = E_kept, E_interior, E_boundary (device memory; the caller reads them once to size its views)
 * The six edge outputs need room for E elements each; workspace = e3_split_edges_workspace_bytes(N) bytes. */
int64_t e3_split_edges_workspace_bytes(int64_t N);
int e3_split_edges(const int32_t* rowptr, const int32_t* src, const uint8_t* is_ghost, int64_t N, int64_t E,
                   int32_t* rowptr_kept, int32_t* src_kept, int32_t* dst_kept, int32_t* src_interior, int32_t* dst_interior,
                   int32_t* src_boundary, int32_t* dst_boundary, int32_t* counts, void* workspace, void* stream);

/* =================================================================================================
 * General SH tensor product, l <= 2 (builder-defined generalisation of the reference operator, which
 * hard-asserts lmax == 1, l1_tensor_prod.py:13-14).
 *
 *   in1  : irreps blocks (l, p, mul), l in {0,1,2}, declaration order = column order (m fastest)
 *   in2  : real "component" spherical harmonics up to lmax_sh in {1,2}: [Y0 | Y1 xyz | Y2 (5)]
 *   out[c3, w, m3] = norm * sum_{paths (l1,l2) -> c3} sum_k W_c3[row(path,k), w]
 *                           sum_{m1,m2} C^{l1 l2 l3}[m1,m2,m3] in1[c1,k,m1] in2[l2,m2]
 *   classes c = 2*l + (p == -1 ? 1 : 0)  (0e,0o,1e,1o,2e,2o); p1 = p3 * (-1)^l2; triangle rule on (l1,l2,l3)
 *   weight rows of class c3: paths ordered by (l1, l2) ascending, channels in order of appearance —
 *   for l <= 1 this is exactly the reference's row order (l1_tensor_prod.py:81-88)
 *   C: real-basis 3j tensors, unit Frobenius norm, basis/sign convention of oracle/cg.py; for l <= 1 they
 *   are the reference's constants (l1_tensor_prod.py:91-94) and xyz dot / cross products.
 * With l <= 1 irreps, lmax_sh = 1 and the same weights/norms, e3_tp_forward == e3_l1tp_forward.
 * ================================================================================================= */
typedef struct e3_tp_plan e3_tp_plan;
/* one column segment of in1 for the fused entry point: rows are base[row_index[b] * ld + c] (row_index NULL = b) */
typedef struct e3_tp_segment {
  const void* base;
  int64_t ld;
  const int32_t* row_index;
  int32_t ncols;
  int32_t reserved;
} e3_tp_segment;
int e3_tp_plan_create(const int32_t* in1_blocks, int n_in1, int lmax_sh,
                      const int32_t* out_blocks, int n_out, e3_tp_plan** plan);
int e3_tp_plan_destroy(e3_tp_plan* plan);
int e3_tp_in1_dim(const e3_tp_plan* plan);
int e3_tp_in2_dim(const e3_tp_plan* plan);
int e3_tp_out_dim(const e3_tp_plan* plan);
/* cls in 0..5; rows = sum of in1 multiplicities over the class's paths ("fan-in"), cols = out multiplicity */
int e3_tp_weight_shape(const e3_tp_plan* plan, int cls, int* rows, int* cols);
int e3_tp_norm_len(const e3_tp_plan* plan, int cls);
int64_t e3_tp_packed_bytes(const e3_tp_plan* plan, int dtype);
int e3_tp_pack_weights(const e3_tp_plan* plan, const void* const weights[6], const void* const norms[6],
                       int dtype, void* packed, void* stream);
/* dtype E3_F32 / E3_F64 (in2 of the same type; ld_in2 == 0 broadcasts row 0) or E3_BF16 (in2 fp32, see below).
 * E3_F32 runs on the matrix cores where the plan has an MFMA instantiation (fp16 (hi, lo)-split operands, fp32
 * accumulation: ~2^-21 per product, see "Operand scales" below); `exact` != 0 forces the generic fp32 FMA kernel.
 * in_scale: device pointer to {s, 1/s} from e3_pow2_scale (NULL = 1: inputs must then lie in about [2^-3, 2^12]). */
int e3_tp_forward(const e3_tp_plan* plan, const void* in1, int64_t ld_in1, const void* in2, int64_t ld_in2,
                  const void* packed, void* out, int64_t ld_out, int64_t B, int dtype, const float* in_scale,
                  int exact, void* stream);
/*
 * Operand scales.  The fp32-storage MFMA kernels split every fp32 operand into two fp16 halves (hi = rne16(v),
 * lo = rne16(v - hi)) and accumulate hi*hi + hi*lo + lo*hi in fp32 on v_mfma_f32_16x16x32_f16.  hi + lo carries 22
 * significant bits while lo is a normal fp16 number, so tensors are multiplied by a power of two that puts their largest
 * magnitude at 2^target (weights: at pack time, 2^13; input features: `in_scale`, 2^10 -- the per-row CG / SH factors
 * add < 2^4).  e3_pow2_scale reduces max |x| over up to 4 fp32 tensors (segs[i].base/ld/ncols x nrows[i]; row_index
 * ignored) and writes out4 = {s, 1/s, scratch, scratch} on the stream; no host synchronisation.  Exact powers of two:
 * scaling itself adds no rounding.  e3_add_pow2_scale: out = h + u (n elements, n % 4 == 0, 16-byte aligned) and the
 * scale of `out` in the same pass (the residual update of a layer yields the next layer's scale for free).
 */
int e3_pow2_scale(const e3_tp_segment* segs, const int64_t* nrows, int nseg, int target_log2, float* out4,
                  void* stream);
int e3_add_pow2_scale(const float* h, const float* u, float* out, int64_t n, int target_log2, float* out4,
                      void* stream);
/*
 * Fused gather + concat (+ gate) form (MFMA kernel, natural-parity irreps 0e/1o/2e only; one plan belongs to the device
 * that was current at its first use -- a call with another current device returns E3_ERR_INVALID_ARG):
 *   in1[b] = [ seg0.base[idx0[b]] | seg1.base[idx1[b]] | ... ]   (the gather and the concat never reach HBM;
 *            segment boundaries must fall on irreps-block boundaries, at most 4 segments)
 *   gate != 0: out irreps must be [32x0e | 32x0e per gated block | 32x1o | 32x2e]; the kernel writes
 *              [silu(s) | sigmoid(g1) v1 | sigmoid(g2) v2] (width 32 + 96 (+160)) instead of the raw TP output.
 * Returns E3_ERR_UNSUPPORTED when the plan / shape has no MFMA instantiation (callers then use
 * e3_gather_concat + e3_tp_forward + e3_gate_blocks).  e3_tp_fused_supported() answers that up front.
 */
int e3_tp_forward_fused(const e3_tp_plan* plan, const e3_tp_segment* segs, int nseg,
                        const void* in2, int64_t ld_in2, const void* packed, void* out, int64_t ld_out,
                        int64_t B, int dtype, int gate, const float* in_scale, void* stream);
/* e3_tp_forward_fused with two epilogue extras, so that a layer's residual update and the next operand scale cost no pass
 * of their own:  out = (gated) product + residual   (residual: [B, width of out], storage dtype, null = none), and
 * out_scale4 (null = none) receives what e3_pow2_scale(out, target_log2) would: {s, 1/s, bits of max |out|, -}. */
int e3_tp_forward_fused_epilogue(const e3_tp_plan* plan, const e3_tp_segment* segs, int nseg, const void* in2,
                                 int64_t ld_in2, const void* packed, void* out, int64_t ld_out, int64_t B, int dtype,
                                 int gate, const float* in_scale, const void* residual, int64_t ld_residual,
                                 float* out_scale4, int target_log2, void* stream);
/* Gradients of e3_tp_forward (fp32 / fp64; the reference operator relies on torch autograd, l1_tensor_prod.py:240-299).
 * `packed` = the buffer e3_tp_pack_weights wrote for this dtype.  Any of grad_in1 [B, in1_dim] (storage dtype),
 * grad_in2 [B, in2_dim] (ACCUMULATION dtype: fp32 for E3_F32, fp64 for E3_F64; with broadcast in2, ld_in2 == 0, pass
 * ld_gin2 == 0 and a zero-filled [in2_dim] row) and grad_weights[6] (one [rows, cols] array per output class as
 * e3_tp_weight_shape reports, accumulation dtype, ZERO-FILLED by the caller, nullptr to skip a class) may be null.
 * grad_weights (and the broadcast grad_in2) are accumulated with atomics: sums are not bitwise reproducible. */
int e3_tp_backward(const e3_tp_plan* plan, const void* in1, int64_t ld_in1, const void* in2, int64_t ld_in2,
                   const void* packed, const void* grad_out, int64_t ld_gout, void* grad_in1, int64_t ld_gin1,
                   void* grad_in2, int64_t ld_gin2, void* const grad_weights[6], int64_t B, int dtype, void* stream);
/*
 * The same gradients as two thin passes around library GEMMs (the path for large B; fp32 / fp64).  Per output class c3
 * (D3 = 2 l3 + 1, [rows, cols] = e3_tp_weight_shape):
 *   e3_tp_backward_operands   features[c3][b, c, r] = sum_{a,q} C[a][q][c] in1[b,k,a] in2[b,q]   ([B, D3, rows], r = path row + k)
 *                             gout[c3][b, c, w]     = grad_out[b, col(w) + c] * norm[w, c]       ([B, D3, cols])
 *   caller (any GEMM)         grad_W[c3] = features^T gout  over the B * D3 rows;   t[c3] = gout W[c3]^T   ([B, D3, rows])
 *   e3_tp_backward_contract   grad_in1[b,k,a] = sum_paths sum_{q,c} C in2[b,q] t[b,c,r]
 *                             grad_in2[b,q]   = sum_paths sum_{k,a,c} C in1[b,k,a] t[b,c,r]
 * Arrays are dense, accumulation dtype (fp32 / fp64); a null entry skips that class (features / gout) or means the class
 * has no weights (t).  grad_in2 follows e3_tp_backward's convention (accumulation dtype; broadcast in2: ld_in2 == 0,
 * ld_gin2 == 0 and a zero-filled [in2_dim] row, summed with atomics).  Either pointer array of _operands may be null.
 */
/* grad_weights alone, fused (fp32): the features and the normalised output gradient of a row tile are built in LDS and
 * contracted on v_mfma_f32_32x32x2_f32 -- nothing of size [B, D3, rows] reaches HBM.  grad_weights[c]: [rows, cols] fp32,
 * ZERO-FILLED by the caller (sums arrive through atomics), nullptr to skip a class.  E3_ERR_UNSUPPORTED (nothing launched) when a
 * requested class has more than 32 output tiles of 32 x 32 or no row tile fits the LDS: use the operand pass + GEMM then. */
int e3_tp_backward_weights(const e3_tp_plan* plan, const void* in1, int64_t ld_in1, const void* in2, int64_t ld_in2,
                           const void* packed, const void* grad_out, int64_t ld_gout, void* const grad_weights[6],
                           int64_t B, int dtype, void* stream);
int e3_tp_backward_operands(const e3_tp_plan* plan, const void* in1, int64_t ld_in1, const void* in2, int64_t ld_in2,
                            const void* packed, const void* grad_out, int64_t ld_gout, void* const features[6],
                            void* const gout[6], int64_t B, int dtype, void* stream);
int e3_tp_backward_contract(const e3_tp_plan* plan, const void* in1, int64_t ld_in1, const void* in2, int64_t ld_in2,
                            void* const t[6], void* grad_in1, int64_t ld_gin1, void* grad_in2, int64_t ld_gin2,
                            int64_t B, int dtype, void* stream);
int e3_tp_fused_supported(const e3_tp_plan* plan, int gate);
/* kernel family ("e3::tp_fwd_mfma_r16_kernel") that this thread's most recent
 * e3_tp_forward_fused / _scatter / MFMA e3_tp_forward call launched; "" before the first one (diagnostics, bench labels) */
const char* e3_tp_last_fused_kernel(void);
/* e3_tp_forward_fused with the message pass's segment-sum fused into the epilogue: row b of the (gated) product is not
 * stored but ADDED to out_nodes[row_node[b]] (fp32 atomics; row_node ascending, e.g. the dst column of a CSR-by-dst
 * edge list; out_nodes zero-initialised by the caller, ld_out a multiple of 4 elements, 16-byte aligned).  The
 * [B, width] messages never reach HBM.  Summation order is not fixed: results agree with e3_tp_forward_fused +
 * e3_segment_sum to fp32 rounding of the sum, not bit for bit.  out_nodes is FP32 for both storage types (`dtype` is
 * the storage of the inputs; with E3_BF16 the caller rounds the sums once).  gate = 1 only; E3_ERR_UNSUPPORTED when the
 * plan has no two-wave instantiation with this epilogue (callers then run the two kernels). */
int e3_tp_forward_fused_scatter(const e3_tp_plan* plan, const e3_tp_segment* segs, int nseg,
                                const void* in2, int64_t ld_in2, const void* packed, const int32_t* row_node,
                                void* out_nodes, int64_t ld_out, int64_t B, int dtype, int gate,
                                const float* in_scale, void* stream);
/* =================================================================================================
 * Fused SEGNN message function (builder-defined; the north_star's "fused per-edge CDNA4 HIP kernel"):
 *
 *   out[i] = sum_{e: dst[e] = i} gate( TP2( gate( TP1( [h[dst[e]] | h[src[e]] | d_e] ; Y_e ) ) ; Y_e ) )
 *
 * with Y_e, d_e = component spherical harmonics (l <= lmax) and length of pos[src[e]] - pos[dst[e]] (the expressions of
 * e3_edge_geometry / e3_edge_geometry_l2), TP1 / TP2 = e3_tp_* products with in irreps Hx0e+Hx1o(+Hx2e) (TP1: twice that
 * plus 1x0e) and out irreps Hx0e + lmax*H x0e + Hx1o (+Hx2e), gate = [silu(s) | sigmoid(g_l) v_l].  One launch (plus a
 * per-node pre-mix launch) per layer: neither Y [E, 9], d [E] nor any [E, width] message tensor exists in HBM.
 * hidden in {16, 32, 64}.  dtype E3_F32: fp32 storage, fp16 (hi, lo)-split MFMA products (see "Operand scales"; `in_scale` =
 * scale of h, NULL = 1; the gated messages between the two products are scaled per edge row inside the kernel).
 * dtype E3_BF16 (hidden 32 / 64; e3_msg_supports): h, weights and norms bf16, positions / harmonics / accumulators / the
 * pre-mix table / `out` fp32, one bf16 MFMA per product, the messages between the two products rounded to bf16.
 * Edges must be sorted by dst (CSR order, as e3_rg_fill emits them): runs of equal dst are summed on chip and leave as
 * one fp32 atomic add per node and wave -- sums agree with e3_segment_sum to fp32 rounding, not bit for bit.
 *   weights: w1[l3] / w2[l3] = the class matrices of TP1 / TP2 for output degree l3 (0e, 1o, 2e), row order and shapes
 *            as e3_tp_weight_shape reports for those irreps (e3_msg_weight_shape returns the same numbers);
 *            n1 / n2 = their norm buffers (length M, 3 M, 5 M) or NULL for 1.
 *   premix : N * e3_msg_premix_floats_per_node() floats written by e3_msg_premix (W_dst h per node: the dst half of TP1
 *            does not depend on the edge, so it is contracted once per NODE and enters the edge kernel as the MFMA
 *            accumulator's initial value; behind the N table rows: max |h[n] * in_scale| per node, from which the edge kernel
 *            bounds a row's messages); call e3_msg_premix(h) before e3_msg_forward on the same h and in_scale
 *   out    : [N, ld_out] fp32, columns [H | 3 H | 5 H]; zero-filled by the call unless accumulate != 0
 *   accumulate != 0: a second edge list for the SAME h rows of the dst nodes (e.g. the halo's boundary edges after the
 *            interior ones): out keeps its contents (premix is reused: it depends on dst rows only)
 *   tiles_per_block: work granularity (0 = default).  hidden = 32, l_max = 2, fp32: the weights-stationary kernel
 *            (e3_msg_ws.hip: one workgroup of 8 waves per CU walks chunks of 16 * tiles_per_block edges, default 256 edges,
 *            dealt round-robin to the workgroups of an XCD; tiles inside a chunk are cut at dst-run boundaries: <= 16 edges,
 *            <= 2 runs).  Other shapes, or tiles_per_block < 0: the one-wave-per-tile kernel (e3_msg_fused.hip), where
 *            |tiles_per_block| = consecutive 16-edge tiles a wave processes before it jumps to its workgroup's next chunk (0 = 4)
 *   E      : edges of this call, dst-sorted (src / dst int32); E <= 2^31 - 17 (E3_ERR_INVALID_ARG beyond: the edge ids are
 *            int32 and the kernel's tile arithmetic is 32-bit)
 *   The launch fills the device once: one 512-thread workgroup per CU (weights-stationary kernel), or CUs x the workgroups
 *   per CU that hipOccupancyMaxActiveBlocksPerMultiprocessor reports for the kernel's registers and LDS image (queried at
 *   the plan's first use).
 * One plan belongs to the device current at its first use.
 * ================================================================================================= */
typedef struct e3_msg_plan e3_msg_plan;
int e3_msg_plan_create(int lmax, int hidden, e3_msg_plan** plan);
int e3_msg_plan_destroy(e3_msg_plan* plan);
int64_t e3_msg_packed_bytes(const e3_msg_plan* plan);
int64_t e3_msg_premix_floats_per_node(const e3_msg_plan* plan);
int e3_msg_weight_shape(const e3_msg_plan* plan, int tp /* 1 | 2 */, int l3, int* rows, int* cols);
int e3_msg_supports(const e3_msg_plan* plan, int dtype);
int e3_msg_pack_weights(e3_msg_plan* plan, const void* const w1[3], const void* const n1[3],
                        const void* const w2[3], const void* const n2[3], int dtype, void* packed, void* stream);
int e3_msg_premix(e3_msg_plan* plan, const void* h, int64_t ld_h, int64_t N, const void* packed,
                  const float* in_scale, float* premix, int dtype, void* stream);
int e3_msg_forward(e3_msg_plan* plan, const void* h, int64_t ld_h, int64_t N, const float* pos4,
                   const int32_t* src, const int32_t* dst, int64_t E, const void* packed, const float* in_scale,
                   const float* premix, float* out, int64_t ld_out, int dtype, int accumulate, int tiles_per_block,
                   void* stream);
/*
 * bf16 storage (dtype E3_BF16, BASELINE config 3): segments / in1 / out / weights / norms are bf16, in2 (the
 * spherical harmonics) stays fp32, products run once on v_mfma_f32_16x16x32_bf16 with fp32 accumulation and one
 * rounding of the result.  Only the MFMA path exists for bf16 (E3_ERR_UNSUPPORTED otherwise).
 */
int e3_segment_sum_bf16(const void* msg, int64_t ld_msg, const int32_t* rowptr, int64_t N, int D,
                        void* agg, int64_t ld_agg, void* stream);
/* SH / geometry for lmax 2: edge_y [E,9], node_a [N,9] (same definitions as e3_edge_geometry, Y2 = sqrt5 b(r^)) */
int e3_edge_geometry_l2(const float* pos4, const int32_t* rowptr, const int32_t* src, int64_t N,
                        float* edge_y, float* edge_d, float* node_a, void* stream);
/* general gate: in = [ns scalars | g gate scalars | gated blocks], block i = mul_i x (2 l_i + 1) with one gate per
 * channel, gates consumed in block order; out = [silu(s) | sigmoid(gate) * block].  ls/muls: host int arrays. */
int e3_gate_blocks(const float* in, int64_t ld_in, float* out, int64_t ld_out, int64_t B, int ns,
                   int nblocks, const int32_t* ls, const int32_t* muls, void* stream);

/* =================================================================================================
 * Backward of the edge / node stages (fp32): with e3_l1tp_backward / e3_tp_backward they make a whole SEGNN layer
 * differentiable, i.e. parameter gradients and the force head -dE/dpos (BASELINE.json configs[3]).  The reference's
 * contract for its own operator is torch autograd (l1_tensor_prod.py:234-299); these are the same for the stages it lacks.
 *   e3_edge_geometry_backward : g_pos [N,3] (zero-filled by the call) from g_edge_y [E,(lmax+1)^2], g_edge_d [E],
 *                               g_node_a [N,(lmax+1)^2] (any may be NULL); atomics on g_pos.
 *   e3_gather_concat_backward : g_out [E, 2D+n_extra] -> g_h [N,D] (zero-filled by the call; dst rows summed per CSR row,
 *                               src rows by atomics), g_extra [E,n_extra] (may be NULL).
 *   e3_gate_blocks_backward   : in = the forward's input [B, ns+ngates+wide], g_out [B, ns+wide] -> g_in (same layout
 *                               as in); e3_gate is the one-block case {l = 1, mul = nv}.
 *   e3_segment_sum_backward   : g_msg[e] = g_agg[dst(e)].
 * ================================================================================================= */
int e3_edge_geometry_backward(const float* pos4, const int32_t* rowptr, const int32_t* src, int64_t N, int lmax,
                              const float* g_edge_y, const float* g_edge_d, const float* g_node_a, float* g_pos,
                              void* stream);
int e3_gather_concat_backward(const float* g_out, int64_t ld_gout, int D, const int32_t* rowptr, const int32_t* src,
                              int64_t N, int n_extra, float* g_h, int64_t ld_gh, float* g_extra, void* stream);
int e3_gate_blocks_backward(const float* in, int64_t ld_in, const float* g_out, int64_t ld_gout, float* g_in,
                            int64_t ld_gin, int64_t B, int ns, int nblocks, const int32_t* ls, const int32_t* muls,
                            void* stream);
int e3_segment_sum_backward(const float* g_agg, int64_t ld_gagg, const int32_t* rowptr, int64_t N, int D, float* g_msg,
                            int64_t ld_gmsg, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* E3GNN_H */
