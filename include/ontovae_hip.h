/*
 * ontovae_hip.h  --  C ABI of libontovae_hip.so, the MI355X (gfx950) OntoVAE hot path.
 *
 * The reference (hdsu-bioquant/onto-vae) is pure Python on top of torch; it has no FFI.  Its
 * "plugin API" for this path is the Python class surface of onto_vae/vae_model.py:OntoVAE.  Each
 * entry point below therefore names the reference function (file:line, relative to the reference
 * root) whose arithmetic it replaces; the Python host layer (onto_vae_amd/) mirrors the reference
 * classes one to one and is the only caller that ships.
 *
 * Conventions
 *  - plain pointers and sizes only; every device pointer is fp32/int32 HBM owned by the caller
 *    (torch allocates it; this library allocates no device memory);
 *  - every device entry point is asynchronous on `stream` (a hipStream_t passed as void*) and
 *    never synchronises; scalars (losses) are written to device memory;
 *  - return value 0 = ok, negative = error; ontovae_last_error() gives the message;
 *  - activations inside the library are FEATURE-MAJOR: a [F, ldb] fp32 matrix, one row per
 *    feature, batch along the contiguous axis (ldb = B rounded up to 64).  Caller-facing
 *    tensors (x, eps, mu, log_var, recon, activities) stay sample-major like the reference's.
 */
#ifndef ONTOVAE_HIP_H
#define ONTOVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ONTOVAE_OK 0
#define ONTOVAE_EINVAL (-1)
#define ONTOVAE_EHIP (-2)
#define ONTOVAE_ENOMEM (-3)

const char* ontovae_last_error(void);
/* ABI version of this header; the host layer refuses a library that reports another one. */
int ontovae_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Mask packing (host side, no GPU needed).
 *
 * Replaces the dense mask expansion of OntoDecoder.__init__ (onto_vae/modules.py:268-275:
 * repeat_interleave(neuronnum) on both axes of the inner masks, on the columns of the last) and
 * the dense gradient-mask multiply of train_round (onto_vae/vae_model.py:168-169).  Input is
 * the reference's own mask list format, Ontobj.masks[key]['decoder'] (onto_vae/ontobj.py:386-392):
 * mask i has rows = level i+1, columns = [level i | ... | level 0]; the last mask is [G, T].
 *
 * Node space.  The decoder is ONE sparse lower-triangular operator over C + G nodes:
 *   nodes [0, C)      the activation panel, C = nu * T, laid out as the reference's final
 *                     concatenation out_{K-1} = [level K-1 | ... | level 1 | level 0 (z)]
 *                     (onto_vae/modules.py:308-310), nu adjacent features per term;
 *   nodes [C, C+G)    the genes (reconstruction).
 * Packed forms, both over expanded (per-neuron) entries, sorted, no duplicates:
 *   CSR  rowptr[C+G+1], colidx[nnz]            entry k of row n:  W[n, colidx[k]] = vals[k]
 *   CSC  cscptr[C+1], csc_row[nnz], csc_k[nnz] entry of column p: W[csc_row, p] = vals[csc_k]
 * level_off[K+1] gives the node range of each forward level: level d (1..K-1) is nodes
 * [level_off[d], level_off[d-1]) ... see ontovae_plan_levels().
 * -------------------------------------------------------------------------------------------*/
typedef struct ontovae_plan ontovae_plan_t;

typedef struct ontovae_dims {
    int32_t G;        /* genes = in_features                        (vae_model.py:50) */
    int32_t T;        /* terms                                                         */
    int32_t K;        /* decoder layers = number of masks                              */
    int32_t nu;       /* neuronnum                                   (vae_model.py:56) */
    int32_t L;        /* latent width = nu * n_0                     (vae_model.py:54) */
    int32_t C;        /* panel width  = nu * T                                         */
    int64_t nnz;      /* structurally non-zero decoder weights (expanded)              */
    int64_t nnz_links;/* ones in the input masks                                       */
    int64_t nnz_term; /* of nnz, the entries of the K-1 term layers (rows [0, C) of the CSR)  */
} ontovae_dims_t;

/* masks[i]: row-major uint8 [rows[i], cols[i]], values 0/1.  Shapes are validated against the
 * reference's layout rule (cols[i] = sum of the level sizes so far). */
int ontovae_plan_create(int32_t K, const uint8_t* const* masks, const int32_t* rows,
                        const int32_t* cols, int32_t nu, ontovae_plan_t** out);
/* Same plan from a link list (child, parent) in mask coordinates: layer[i], row[i], col[i] name a
 * one in mask layer[i].  This is the dense-free path for large ontologies. */
int ontovae_plan_create_from_links(int32_t K, const int32_t* rows, const int32_t* cols, int32_t nu,
                                   int64_t n_links, const int32_t* layer, const int32_t* row,
                                   const int32_t* col, ontovae_plan_t** out);
void ontovae_plan_destroy(ontovae_plan_t* plan);
int ontovae_plan_dims(const ontovae_plan_t* plan, ontovae_dims_t* out);
/* level_node_begin/end[d], d = 0..K: node range of level d (0 = z, K = genes). */
int ontovae_plan_levels(const ontovae_plan_t* plan, int32_t* node_begin, int32_t* node_end);
/* Copy the packed arrays out (host buffers sized from ontovae_plan_dims). */
int ontovae_plan_export(const ontovae_plan_t* plan, int32_t* rowptr, int32_t* colidx,
                        int32_t* cscptr, int32_t* csc_row, int32_t* csc_k);
/* csc_split[p], p in [0, C): index of the first gene entry of CSC column p; the entries before it
 * are the term-to-term links of that column (rows ascend and term nodes precede gene nodes). */
int ontovae_plan_csc_split(const ontovae_plan_t* plan, int32_t* csc_split);
/* Packed position k (index into dec_vals and into every buffer laid out like it) of the link at CSR position j, [nnz].
 * Term layers: k = j.  Gene layer: k orders the links by (panel node, gene), the order of the gene part of the CSC lists
 * (csc_k of those entries ascends by 1), so that the backward gene kernel reads weights and writes gradient partials in
 * contiguous runs; rowptr / colidx stay indexed by j. */
int ontovae_plan_row_k(const ontovae_plan_t* plan, int32_t* row_k);
/* Dense coordinates of every packed entry, in vals order: entry k is
 * decoder.decoder[layer[k]][0].weight[row[k], col[k]] of the reference's state_dict. */
int ontovae_plan_entry_coords(const ontovae_plan_t* plan, int32_t* layer, int32_t* row, int32_t* col);
/* Node id of bias element j of layer i: used to map decoder.decoder[i][0].bias <-> dec_bias. */
int ontovae_plan_layer_node0(const ontovae_plan_t* plan, int32_t layer, int32_t* node0, int32_t* n_out);
/* Unpack layer i back into the reference's EXPANDED dense mask (uint8 [out_i, in_i]); the
 * bit-exactness check is unpack(pack(M)) == OntoDecoder.masks[i]. */
int ontovae_plan_unpack_mask(const ontovae_plan_t* plan, int32_t layer, uint8_t* out, int64_t out_len);

/* ------------------------------------------------------------------------------------------
 * Device context: dimensions + device pointers, filled by the caller.
 * Parameters live in ONE flat fp32 buffer `P` (so that a data-parallel gradient exchange is a
 * few large buffers); `M`, `V` (AdamW moments) and `Gr` (gradients, only used when
 * grads_to_buffer != 0) have the same layout.  Offsets in floats:
 * -------------------------------------------------------------------------------------------*/
/* Level-scheduled SELL-64 form of the TERM layers (the K-1 small, strictly sequential layers), one
 * per direction: dir 0 = forward (CSR rows of levels 1..K-1), dir 1 = backward (term part of the CSC
 * columns of levels K-1..0).  Inside a level the nodes are sorted by list length and cut into slices
 * of 64 lanes; a node of up to 8 entries owns one lane, longer nodes own 2, 4, .. 64 adjacent lanes
 * (their partial sums are combined with DPP), and a lane holds its entries in 64x8 chunks
 * ([8][64] index / weight-position arrays, so every load of a slice is one contiguous 256-B row per
 * entry slot).  info[lane] = entries of the lane | log2(lanes per node) << 16 | entries per lane of the
 * slice (its width, <= 8) << 20 | writer lane << 24.  Padding slots have k = -1 (weight 0) and the index
 * of a row the slice really reads.
 * The library keeps a weight copy in this order (`w`, refreshed from P at the start of every pass). */
typedef struct ontovae_sell {
    int32_t n_levels;
    int32_t n_slices;
    int32_t n_chunks;
    int32_t all_single_chunk;     /* every slice has exactly one chunk: chunk id == slice id */
    const int32_t* level_first;   /* host [n_levels]: first slice of each level, in processing order */
    const int32_t* level_count;   /* host [n_levels] */
    const int32_t* node;          /* device [n_slices * 64], -1 = unused lane */
    const int32_t* info;          /* device [n_slices * 64] */
    const int32_t* chunk0;        /* device [n_slices + 1] */
    const int32_t* idx;           /* device [n_chunks * 512] gather node of each entry */
    const int32_t* k;             /* device [n_chunks * 512] position in dec_vals, -1 = padding */
    float* w;                     /* device scratch [n_chunks * 512] */
    float* bias;                  /* device scratch [n_slices * 64] (forward only) */
} ontovae_sell_t;

int ontovae_plan_sell_sizes(const ontovae_plan_t* plan, int32_t dir, int32_t* n_levels, int32_t* n_slices,
                            int32_t* n_chunks, int32_t* all_single_chunk);
int ontovae_plan_sell_export(const ontovae_plan_t* plan, int32_t dir, int32_t* level_first, int32_t* level_count,
                             int32_t* node, int32_t* info, int32_t* chunk0, int32_t* idx, int32_t* k);

/* SELL-64 form of the GENE layer for the LDS-panel forward kernel of the training pass (a workgroup holds the 4 batch columns
 * of all C panel rows in LDS and walks the genes' lists against it): one gene per lane, genes sorted by list length inside
 * windows of 512 consecutive genes and cut into slices of 64 lanes.  The lists are stored in quads of 4 entries: quad q of
 * slice s, lane l is element (quad0[s] + q) * 64 + l of `col` (4 x uint16 panel rows), `k` (4 packed positions, -1 =
 * padding: weight 0, row 0) and `w`.  n_slices == 0 (not provided, or a panel index that does not fit 16 bits): the
 * 32-column-block kernel runs instead.
 * The library keeps a weight copy in this order (`w`, refreshed from P at the start of every training pass).
 * Replaces the per-row walk of the masked matmul of onto_vae/modules.py:308-312 for the last decoder layer. */
typedef struct ontovae_gsell {
    int32_t n_slices;
    int32_t n_quads;
    const int32_t* gene;          /* device [n_slices * 64] gene of each lane, -1 = unused lane */
    const int32_t* quad0;         /* device [n_slices + 1] */
    const uint16_t* col;          /* device [n_quads * 256] */
    const int32_t* k;             /* device [n_quads * 256] */
    float* w;                     /* device scratch [n_quads * 256] */
} ontovae_gsell_t;

int ontovae_plan_gsell_sizes(const ontovae_plan_t* plan, int32_t* n_slices, int32_t* n_quads);
int ontovae_plan_gsell_export(const ontovae_plan_t* plan, int32_t* gene, int32_t* quad0, uint16_t* col, int32_t* k);

/* Tiles of 32 rows as the column-block gene kernels take them: for every tile the 32 rows in order of decreasing list length,
 * one record of 4 ints per slot {first list position, list length, row, packed position of the list's first link}.
 * which = 0: gene tiles ([ceil(G / 32) * 32 * 4]; rows are genes, lists the CSR rows of the gene layer, unused slots have row G);
 * which = 1: panel tiles ([ceil(C / 32) * 32 * 4]; rows are panel nodes, lists the gene part of their CSC columns, unused: row C);
 * which = 2: term tiles ([ceil((C - L) / 32) * 32 * 4]; rows are the term layers' output nodes [0, C - L), lists their CSR rows,
 *            unused: row C - L): what the weight-gradient workgroups of the term layers read first.
 * which = 3: term-panel tiles ([4 + ceil((C / nu) / 32) * 32 * 4]: out[0] = 1 if present, else 0 and nothing follows; out[1] = the
 *            largest sum of list lengths over the tiles; the records start at out + 4): rows are the TERMS of the panel (record.z = the term's first node nu * t, C if unused), the list is
 *            the gene part of that node's CSC column, which the term's nu nodes share (the gene mask is expanded along its
 *            columns, modules.py:275); the packed positions of node nu * t + e start at record.w + e * length.  The backward
 *            gene kernel then gathers a gene's row once for the nu nodes.
 * Optional in the context (NULL: every workgroup ranks its tile's rows itself; tpanel_tiles NULL: one lane group per node). */
int ontovae_plan_tile_info(const ontovae_plan_t* plan, int32_t which, int32_t* out);

typedef struct ontovae_layout {
    int64_t W0;      /* [L, G]   encoder.encoder.0.0.weight                                   */
    int64_t b0;      /* [L]      encoder.encoder.0.0.bias                                     */
    int64_t gamma;   /* [L]      encoder.encoder.0.1.weight                                   */
    int64_t beta;    /* [L]      encoder.encoder.0.1.bias                                     */
    int64_t Wh;      /* [2L, L]  rows 0..L-1 encoder.mu.0.weight, rows L..2L-1 encoder.logvar  */
    int64_t bh;      /* [2L]                                                                  */
    int64_t dec_vals;/* [nnz]    packed decoder weights, CSR order                            */
    int64_t dec_bias;/* [C+G]    decoder biases by node id (entries of the z level unused)    */
    int64_t total;   /* floats in P                                                           */
} ontovae_layout_t;

int ontovae_param_layout(const ontovae_dims_t* dims, ontovae_layout_t* out);

typedef struct ontovae_ctx {
    ontovae_dims_t dims;
    ontovae_layout_t lay;
    /* parameters and optimizer state (device) */
    float* P;
    float* M;
    float* V;
    float* Gr;                 /* may be NULL unless grads_to_buffer */
    float* bn_rmean;           /* [L] running_mean  */
    float* bn_rvar;            /* [L] running_var   */
    int64_t* bn_nbt;           /* [1] num_batches_tracked */
    /* packed decoder (device) */
    const int32_t* rowptr;
    const int32_t* colidx;
    const int32_t* cscptr;
    const int32_t* csc_row;
    const int32_t* csc_k;
    const int32_t* csc_split;  /* [C] first GENE entry of each CSC column (term rows sort first) */
    const int32_t* row_k;      /* [nnz] packed position of CSR position j (ontovae_plan_row_k) */
    /* level schedule (host arrays, K+1 entries each) */
    const int32_t* level_begin;
    const int32_t* level_end;
    /* workspace (device), ontovae_workspace_bytes(dims, max_batch) bytes, 256-B aligned */
    void* workspace;
    int64_t workspace_bytes;
    int32_t max_batch;
    int32_t max_train_batch;   /* rows a TRAINING pass may take (ontovae_workspace_bytes_split); 0: max_batch */
    /* dropout (train mode only) and eps generation */
    float drop;                /* Encoder dropout p            (modules.py:48) */
    float z_drop;              /* latent dropout p             (modules.py:58, 63) */
    uint64_t seed;             /* Philox key for dropout masks and internally drawn eps */
    uint32_t flags;            /* ONTOVAE_FLAG_*: force the general-purpose kernel variants (tests) */
    ontovae_sell_t sell_fwd;   /* n_slices == 0: not provided, the per-level kernels run instead */
    ontovae_sell_t sell_bwd;
    /* operand copies of the bf16-MFMA variant (ONTOVAE_FLAG_BF16_GEMM): ontovae_bf16_workspace_bytes(dims, max_batch) bytes of
     * device memory, 256-B aligned; NULL / 0 otherwise */
    void* bf16_ws;
    int64_t bf16_ws_bytes;
    ontovae_gsell_t gsell;     /* n_slices == 0: not provided */
    const int32_t* gene_tiles; /* device, ontovae_plan_tile_info(plan, 0), or NULL */
    const int32_t* panel_tiles;/* device, ontovae_plan_tile_info(plan, 1), or NULL */
    const int32_t* term_tiles; /* device, ontovae_plan_tile_info(plan, 2), or NULL */
    const int32_t* tpanel_tiles;/* device, the RECORDS of ontovae_plan_tile_info(plan, 3) (out + 4), or NULL */
    int32_t tpanel_max_entries; /* out[1] of ontovae_plan_tile_info(plan, 3): the largest sum of list lengths of a term-panel tile */
} ontovae_ctx_t;

#define ONTOVAE_FLAG_LEVEL_LAUNCHES 1u  /* one launch per decoder level instead of the LDS-fused chain */
#define ONTOVAE_FLAG_ROW_GENE_LAYER 2u  /* row-per-wave gene layer instead of the LDS-tiled one */
#define ONTOVAE_FLAG_CB_GENE_FWD 128u   /* training forward of the gene layer on 32-column blocks (gathers from L2) instead of the LDS-panel kernel */
#define ONTOVAE_FLAG_WIDE_GENE_TILES 8u /* gene layer: waves that span 64-256 batch columns instead of XCD-pinned 32-column blocks */
/* Any other bit is refused with ONTOVAE_EINVAL (measurement switches exist only in -DOV_MEASUREMENT_SWITCHES builds). */
/* BASELINE.json configs[4], "bf16 MFMA variant": the two large dense products of a pass -- x W0^T (onto_vae/modules.py:46) and its
 * weight gradient dW0 = dh x -- take their inputs rounded to bf16 and accumulate in fp32 on v_mfma_f32_32x32x16_bf16; weights,
 * optimizer state and every other operation stay fp32.  Stated tolerance against the fp32 reference path (tests): loss 2e-2
 * relative per step, pathway activities cosine >= 0.999 per sample.  Needs ctx->bf16_ws, ctx->Gr (dW0 is staged there) and
 * G % 8 == 0.  The gathered-dh data-parallel form (ontovae_dw0_global) stays fp32. */
#define ONTOVAE_FLAG_BF16_GEMM     32u
#define ONTOVAE_FLAG_PER_BLOCK_WGRAD 64u /* one partial copy of the decoder's weight gradient per 32-column block at every batch size: the backward
                                           gene kernel without its LDS accumulators, the term layers' weight gradient from riders of the latent backward
                                           instead of the per-level backward kernels (the path of an ontology whose tiles do not fit the accumulators;
                                           tests compare the two) */
#define ONTOVAE_FLAG_NO_RIDERS     16u  /* weight gradient of the gene rows as its own launch instead of inside the backward chain's */
#define ONTOVAE_FLAG_SIDE_STREAM    4u  /* run the decoder weight gradient and dWh on a library-owned second stream, forked from
                                           and joined into the caller's stream with events (off by default: on this pool the event
                                           hand-offs cost about what the overlap gains) */

int64_t ontovae_workspace_bytes(const ontovae_dims_t* dims, int32_t max_batch);
/* The buffers only a training pass touches (weight-gradient partials: batch / 32 copies of the packed decoder, the backward's
 * matrices) sized by max_train_batch <= max_batch, everything else by max_batch: a model that trains at 128 rows and
 * serves 16384-row inference batches then needs 2.7 GB of workspace at GO-BP scale instead of 4.5.  ctx->max_train_batch
 * must be set to the same value; training entry points (ontovae_train_step*, ontovae_dw0_*, ontovae_dp_*,
 * ontovae_forward with train != 0) refuse more rows. */
int64_t ontovae_workspace_bytes_split(const ontovae_dims_t* dims, int32_t max_batch, int32_t max_train_batch);
int64_t ontovae_bf16_workspace_bytes(const ontovae_dims_t* dims, int32_t max_batch);

/* AdamW hyper-parameters of one step, pre-folded on the host in double like torch does
 * (torch.optim.AdamW defaults at onto_vae/vae_model.py:251). */
typedef struct ontovae_adamw {
    float decay_mul;     /* 1 - lr * weight_decay          */
    float one_m_beta1;   /* 1 - beta1                      */
    float beta2;
    float one_m_beta2;   /* 1 - beta2                      */
    float step_size;     /* lr / (1 - beta1^t)             */
    float bc2_sqrt;      /* sqrt(1 - beta2^t)              */
    float eps;
} ontovae_adamw_t;

/* Batch addressing.  The batcher keeps the whole dataset X [N, G] fp32 resident in HBM and hands
 * out batches as (X, row index list): row b of the batch is X[xidx[b], :].  xidx == NULL means
 * rows 0..B-1 of `x` (a plain [B, G] batch like the reference's data[0].to(device),
 * onto_vae/fast_data_loader.py:41, onto_vae/vae_model.py:156). */

/* OntoVAE.forward (onto_vae/vae_model.py:108-118) = Encoder.forward (modules.py:74-84) +
 * reparameterize (vae_model.py:91-93) + OntoDecoder.forward (modules.py:303-313).
 * eps: [B, L] sample-major or NULL (drawn internally from seed/rng_offset).
 * Outputs (any may be NULL): recon [B, G], mu [B, L], log_var [B, L], sample-major.
 * train != 0: BN uses batch statistics and updates the running buffers, dropout is on. */
int ontovae_forward(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B,
                    const float* eps, uint64_t rng_offset, int32_t train, float* recon, float* mu,
                    float* log_var, void* stream);

/* One body of OntoVAE.train_round (onto_vae/vae_model.py:156-176): forward, vae_loss
 * (vae_model.py:124-129), backward, gradient mask (:168-169, implicit: gradients exist only at
 * mask positions), AdamW (:172) and the clamp of decoder weights at 0 (:175-176).
 * loss_out: device float[3] = {rec_loss, kl_loss, rec + kl_coeff * kl}.
 * apply_update != 0: fused AdamW in the weight-gradient kernels (single GPU).
 * apply_update == 0: gradients are written to ctx->Gr (same layout as P) and nothing is updated;
 *   BN batch statistics are local to this rank (see ontovae_train_step_dp for SyncBN).
 *   The caller all-reduces Gr and calls ontovae_adamw_apply. */
int ontovae_train_step(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B,
                       const float* eps, uint64_t rng_offset, float kl_coeff,
                       const ontovae_adamw_t* opt, int32_t apply_update, float* loss_out,
                       void* stream);

/* Data-parallel step in three phases around the caller's collectives (RCCL via torch.distributed);
 * the reference has no counterpart (single device), the contract is "equal to the single-device
 * step on the global batch" (SURVEY.md 8(e)):
 *   phase 0: encoder GEMM + local BN partial statistics -> bn_stats [2, L] {sum, sum of squares}
 *            caller: all-reduce SUM of bn_stats, passes global batch size in B_global
 *   phase 1: BN normalise .. decoder .. backward down to BN -> bn_bwd_stats [2, L] {sum dy, sum dy*xhat}
 *            caller: all-reduce SUM of bn_bwd_stats
 *   phase 2: BN backward, dW0; all gradients land in ctx->Gr; loss_out holds LOCAL rec/kl sums.
 *            caller: all-reduce SUM of Gr (+ loss_out), then ontovae_adamw_apply. */
int ontovae_train_step_dp(const ontovae_ctx_t* ctx, int32_t phase, const float* x, const int32_t* xidx,
                          int32_t B, int32_t B_global, const float* eps, uint64_t rng_offset,
                          float kl_coeff, float* bn_stats, float* bn_bwd_stats, float* loss_out,
                          void* stream);

/* Phase 3 of ontovae_train_step_dp is phase 2 without the dW0 product (BatchNorm backward only); ontovae_dw0_rows then
 * forms rows [l0, l1) of dW0 = dh x (the local rows' contribution) into ctx->Gr, so that the caller can all-reduce W0's
 * gradient in contiguous row panels, each panel's exchange under the next panel's product (SURVEY.md 8(e) bucket 3; row
 * panels instead of gene-column panels because rows are contiguous in the flat buffer).  No counterpart in the
 * reference (single device). */
int ontovae_dw0_rows(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B, int32_t l0, int32_t l1,
                     void* stream);

/* torch.optim.AdamW.step over the flat buffers (vae_model.py:172) + clamp(0) of the packed
 * decoder weights (vae_model.py:175-176), reading gradients from ctx->Gr. */
int ontovae_adamw_apply(const ontovae_ctx_t* ctx, const ontovae_adamw_t* opt, void* stream);
/* The same over the float range [begin, end) of the layout only. */
int ontovae_adamw_apply_range(const ontovae_ctx_t* ctx, const ontovae_adamw_t* opt, int64_t begin, int64_t end,
                              void* stream);

/* Data-parallel step WITHOUT an all-reduce of the first layer's weight gradient (28.8 MB at GO-BP scale, the
 * bulk of the gradient): dW0 = dh x is bilinear in two small things every rank can have, the BN-backward output
 * dh [L, B_global] (1.5 MB at 8 x 128 rows) and the rows of the replicated, HBM-resident dataset.  So the ranks
 * all-gather dh and each forms dW0 of the GLOBAL batch (+ AdamW, identical on every rank) itself.  No counterpart
 * in the reference (single device); same contract as ontovae_train_step_dp.  After phases 0 and 1 of
 * ontovae_train_step_dp:
 *   ontovae_dp_bn_bwd: BN backward with the all-reduced bn_bwd_stats.  Gradients of gamma and beta (GLOBAL sums)
 *     and of b0 (LOCAL sum) go to ctx->Gr; send = this rank's block for the all-gather, floats:
 *     [L][blm] dh (columns >= B zero) then [L] the local b0 gradient; blm % 4 == 0, blm >= B.
 *   caller: all-gather of the blocks (rank stride `rstride` floats, % 4 == 0).
 *   ontovae_dw0_global: dW0 over the gathered blocks: column k = r * blm + j of rank block r is column
 *     c = colmap ? colmap[k] : k of the global batch, i.e. dataset row xidx ? xidx[c] : c (padding slots: any valid
 *     column, their dh is 0); AdamW on W0 if apply_update != 0, else the gradient to ctx->Gr.  Also sums the
 *     blocks' b0 parts into ctx->Gr (when Gr is there).
 * ontovae_train_step_dp phase 16 + R: phase 1 with bn_stats = the R ranks' all-gathered blocks [mean | M2 | rows]
 *   (2 L + 1 floats each) instead of the combined [mean | var]; Chan's combination then happens in the kernel. */
int ontovae_dp_bn_bwd(const ontovae_ctx_t* ctx, int32_t B, int32_t B_global, const float* bn_bwd_stats,
                      uint64_t rng_offset, float* send, int32_t blm, void* stream);
int ontovae_dw0_global(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, const int32_t* colmap,
                       int32_t n_ranks, int32_t blm, const float* gathered, int64_t rstride, const ontovae_adamw_t* opt,
                       int32_t apply_update, void* stream);
/* The same with ONE exchange between the backward pass and dW0 instead of two (the all-reduce of the BatchNorm
 * backward sums and the all-gather of dh): after phase 1 every rank packs what it has,
 *   ontovae_dp_pack_dy: send = [L][blm] dy | [L] (b0 slot, zero) | [L][blm] xhat | [2 L] local (sum dy, sum dy xhat),
 *     2 L blm + 3 L floats; columns >= B zero;
 * the ranks all-gather these blocks (rank stride `rstride` floats, % 4 == 0), and
 *   ontovae_dp_dh_global: adds the ranks' sums in rank order (the same bits on every rank), turns dy of ALL blocks into
 *     dh in place, fills the b0 slots with the blocks' sums of dh, and writes the gradients of gamma and beta (the global
 *     sums) to ctx->Gr.  sizes[r] (HOST array) = rows rank r really holds; their total must be B_global.
 * ontovae_dw0_global then runs on the same buffer with the same `rstride`. */
#define ONTOVAE_MAX_DP_RANKS 64
int ontovae_dp_pack_dy(const ontovae_ctx_t* ctx, int32_t B, float* send, int32_t blm, void* stream);
int ontovae_dp_dh_global(const ontovae_ctx_t* ctx, int32_t B_global, int32_t n_ranks, const int32_t* sizes,
                         int32_t blm, float* gathered, int64_t rstride, void* stream);

/* One body of OntoVAE.val_round (onto_vae/vae_model.py:204-209): eval-mode forward + vae_loss. */
int ontovae_eval_step(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B,
                      const float* eps, uint64_t rng_offset, float kl_coeff, float* loss_out,
                      void* stream);

/* OntoVAE._pass_data (onto_vae/vae_model.py:276-327), eval mode.  eps1 feeds get_embedding
 * (:294), eps2 the second forward (:314); NULL = drawn internally.
 * act_out: [B, T] sample-major nu-averaged activities hstack((z, act)) (:297, :317, :325), or NULL
 * rec_out: [B, G] sample-major reconstruction (:327), or NULL (then the gene layer is skipped). */
int ontovae_pass_data(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B,
                      const float* eps1, const float* eps2, uint64_t rng_offset, float* act_out,
                      int64_t act_ld, float* rec_out, int64_t rec_ld, void* stream);

/* OntoVAE.perturbation (onto_vae/vae_model.py:398-456) for screens of many perturbations of one dataset.
 * The reference overwrites the listed gene columns of the data and runs the whole pass again (:437-440); the
 * perturbed input differs from the original in those columns only, so its first encoder layer is
 *   x' W0^T = x W0^T + sum_p (value_p - x[:, gene_p]) W0[:, gene_p]^T.
 * ontovae_encode_pre stores x W0^T of a batch (no bias) feature-major in hpre[l * ldh + b]; the caller keeps it for
 * the screen.  ontovae_pass_data_perturbed is ontovae_pass_data on the UNPERTURBED x plus that cache and the lists
 * pert_gene / pert_val (HOST arrays, n_pert <= ONTOVAE_MAX_PERTURBED_GENES distinct column indices): the encoder GEMM,
 * three quarters of the time of a pass, is replaced by the rank-1 terms.  Outputs as ontovae_pass_data; they agree with
 * a pass over the overwritten data to float32 rounding of the first layer (tests: 1e-4 relative).
 * act_feature_major != 0: act_out is written as act_out[t * act_ld + b] (act_ld >= B), the layout
 * ontovae_wilcoxon_zsplit reads, so a screen needs no transposition between the pass and the test. */
#define ONTOVAE_MAX_PERTURBED_GENES 64
int ontovae_encode_pre(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B, float* hpre,
                       int64_t ldh, void* stream);
int ontovae_pass_data_perturbed(const ontovae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B,
                                const float* hpre, int64_t ldh, int32_t n_pert, const int32_t* pert_gene,
                                const float* pert_val, const float* eps1, const float* eps2,
                                uint64_t rng_offset, float* act_out, int64_t act_ld, int32_t act_feature_major,
                                float* rec_out, int64_t rec_ld, void* stream);

/* OntoDecoder.forward alone (onto_vae/modules.py:303-313) on a caller-supplied z [B, L]:
 * the "decoder-only sparse throughput" leg of the inference benchmark. */
int ontovae_decoder_forward(const ontovae_ctx_t* ctx, const float* z, int32_t B, float* act_out,
                            int64_t act_ld, float* rec_out, int64_t rec_ld, void* stream);

/* ontovae_decoder_forward for a TRAINING pass of a model whose ontology sits in the encoder (OntoEncVAE): Dropout(ctx->drop)
 * on the output of every term layer -- OntoEncoder's [Linear, Dropout] blocks, onto_vae/modules.py:219-223, 231-233 -- keyed by
 * rng_offset (Philox, like every other dropout of the library); the last layer (the heads) carries none here, their
 * Dropout(z_drop) belongs to ontovae_vae_decode_step.  ctx->drop == 0: the same as ontovae_decoder_forward. */
int ontovae_decoder_forward_train(const ontovae_ctx_t* ctx, const float* z, int32_t B, uint64_t rng_offset, float* rec_out,
                                  int64_t rec_ld, void* stream);
/* Backward of ontovae_decoder_forward[_train] plus the update of the masked operator: autograd of OntoDecoder.forward
 * (onto_vae/modules.py:303-313) and of OntoEncoder.forward (modules.py:226-238: the same operator with the genes as its input
 * and [mu | log_var] as its last layer), the gradient mask (onto_vae/vae_model.py:168-169, 611-616), AdamW (:172, :619) and the
 * clamp of the weights at zero (:175-176, :622-625; biases are not clamped).  Must follow the forward call on the same ctx and
 * batch (the panel of that pass is read from the workspace) with the same rng_offset (ctx->drop > 0: the same dropout masks).
 * drec [B, G] contiguous: dL/d(rec_out).  apply_update != 0: AdamW + clamp in place; else the gradients of
 * [dec_vals | dec_bias] go to ctx->Gr.  dz_out [B, dz_ld >= L] or NULL: dL/dz. */
int ontovae_decoder_backward(const ontovae_ctx_t* ctx, const float* drec, int32_t B, uint64_t rng_offset, const ontovae_adamw_t* opt,
                             int32_t apply_update, float* dz_out, int64_t dz_ld, void* stream);

/* HBM-resident batcher helpers (replace onto_vae/fast_data_loader.py:33-34 on the device):
 * order_out[i] = order_in[perm[i]]  (the reference's cumulative in-place shuffle t[r]);
 * gather: out[b, :] = X[idx[b], :]. */
int ontovae_compose_order(const int32_t* order_in, const int64_t* perm, int32_t* order_out, int64_t n,
                          void* stream);
int ontovae_gather_rows(const float* X, int64_t ld, const int32_t* idx, int32_t B, int32_t G, float* out,
                        void* stream);

/* Dense <-> packed decoder weights on the device (state_dict import/export):
 * scatter: dense[layer row/col of entry k] = vals[k] for the entries of `layer` (dense pre-zeroed);
 * gather:  vals[k] = dense[...]. entry_row/entry_col are the device copies of
 * ontovae_plan_entry_coords restricted to [k0, k1). */
int ontovae_vals_to_dense(const float* vals, const int32_t* entry_row, const int32_t* entry_col,
                          int64_t k0, int64_t k1, float* dense, int64_t ld, void* stream);
int ontovae_dense_to_vals(const float* dense, int64_t ld, const int32_t* entry_row,
                          const int32_t* entry_col, int64_t k0, int64_t k1, float* vals, void* stream);

/* ------------------------------------------------------------------------------------------
 * Statistics of a perturbation screen, all on device memory.
 *
 * Ontobj.wilcox_test (onto_vae/ontobj.py:651-701) runs, per term or gene i,
 *   scipy.stats.wilcoxon(perturbed[:, i], control[:, i], zero_method='zsplit', alternative=...)  (:673)
 * and statsmodels' fdrcorrection on the p-values (:676).  ontovae_wilcoxon_zsplit restates the former
 * (scipy 1.15 _wilcoxon_nd with method='auto', correction=False: exact null distribution for n <= 50
 * without ties and zeros, all 2^n sign assignments for n <= 13 with them, normal approximation with
 * tie correction otherwise), ontovae_fdr_bh the latter (Benjamini-Hochberg, 'indep').
 *
 * Both matrices are FEATURE-major: column i of the test is perturbed[i * ld .. i * ld + n), float32;
 * the difference is taken in float32 as numpy does for float32 inputs.  stat / pval: [ncols] float64;
 * stat = r_plus (ONTOVAE_ALT_TWO_SIDED: min(r_plus, r_minus)); a NaN in a column gives NaN, NaN.
 * Columns of up to 32768 samples are ranked inside one workgroup's LDS and need no workspace; longer
 * ones go through `workspace` (ontovae_wilcoxon_workspace_bytes, 256-byte aligned), in groups of
 * as many columns as it holds.
 * ontovae_fdr_bh: nsets independent sets of n p-values, pvals / qvals [nsets, n] float64; workspace of
 * ontovae_fdr_workspace_bytes, 256-byte aligned.  A NaN p-value makes every q of its set NaN.
 * ontovae_to_feature_major: dst[f * dst_ld + b] = src[b * src_ld + f], b < B, f < F -- moves one
 * batch of ontovae_pass_data's sample-major output into the columns of a feature-major matrix.
 * -------------------------------------------------------------------------------------------*/
enum { ONTOVAE_ALT_GREATER = 0, ONTOVAE_ALT_LESS = 1, ONTOVAE_ALT_TWO_SIDED = 2 };
int64_t ontovae_wilcoxon_workspace_bytes(int32_t n, int32_t ncols);
int ontovae_wilcoxon_zsplit(const float* perturbed, const float* control, int64_t ld, int32_t ncols,
                            int32_t n, int32_t alternative, double* stat, double* pval, void* workspace,
                            int64_t workspace_bytes, void* stream);
int64_t ontovae_fdr_workspace_bytes(int32_t n, int32_t nsets);
int ontovae_fdr_bh(const double* pvals, int32_t n, int32_t nsets, double* qvals, void* workspace,
                   int64_t workspace_bytes, void* stream);
/* ontovae_fdr_bh without a workspace and without any allocation.  No counterpart in the reference: it is for callers
 * that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  The `raw` values and NaN flags that ontovae_fdr_bh keeps in the caller's workspace live in a file-scope
 * device array of the library instead (2 MiB + 16 KiB per GPU, held until the process ends).  Limits: n * nsets <= 262144
 * p-values and nsets <= 4096; larger inputs return ONTOVAE_EINVAL (use ontovae_fdr_bh).  Results are bit-identical to
 * ontovae_fdr_bh.  All calls on one GPU share the array: calls that overlap on one GPU (two streams) race.  The first call
 * on a device resolves the array's address (hipGetSymbolAddress, under a mutex): make one call per device outside any graph
 * capture.  qvals must not alias pvals, here and in ontovae_fdr_bh (ONTOVAE_EINVAL). */
int ontovae_fdr_bh_static(const double* pvals, int32_t n, int32_t nsets, double* qvals, void* stream);
int ontovae_to_feature_major(const float* src, int64_t src_ld, int32_t B, int32_t F, float* dst,
                             int64_t dst_ld, void* stream);

/* ------------------------------------------------------------------------------------------
 * The plain VAE sibling: reference class VAE (onto_vae/vae_model.py:913-1140) = Encoder
 * (onto_vae/modules.py:15-84, any number of [Linear, BatchNorm1d, Dropout, ReLU] blocks, then the mu / log_var heads with
 * Dropout(z_drop)) + dense Decoder (modules.py:94-154: the same blocks, then a Linear to the genes).  Same conventions as
 * above: one flat fp32 parameter buffer (ontovae_vae_layout gives the offsets, in floats), moments M / V and gradients Gr in
 * the same layout, BatchNorm running statistics in bn_state ([running_mean | running_var] per block, offsets *_bn), one
 * num_batches_tracked counter per block in bn_nbt (encoder blocks first), caller-owned workspace.
 * -------------------------------------------------------------------------------------------*/
#define ONTOVAE_VAE_MAX_LAYERS 8
typedef struct ontovae_vae_dims {
    int32_t G;                                  /* in_features                      (vae_model.py:934) */
    int32_t Z;                                  /* latent_dim                       (vae_model.py:937) */
    int32_t n_enc, n_dec;                       /* blocks of the encoder / decoder (len(layer_dims_enc / _dec)) */
    int32_t enc[ONTOVAE_VAE_MAX_LAYERS];        /* layer_dims_enc                   (vae_model.py:935) */
    int32_t dec[ONTOVAE_VAE_MAX_LAYERS];        /* layer_dims_dec                   (vae_model.py:936) */
} ontovae_vae_dims_t;

typedef struct ontovae_vae_layout {
    int64_t enc_W[ONTOVAE_VAE_MAX_LAYERS], enc_b[ONTOVAE_VAE_MAX_LAYERS], enc_gamma[ONTOVAE_VAE_MAX_LAYERS], enc_beta[ONTOVAE_VAE_MAX_LAYERS];
    int64_t dec_W[ONTOVAE_VAE_MAX_LAYERS], dec_b[ONTOVAE_VAE_MAX_LAYERS], dec_gamma[ONTOVAE_VAE_MAX_LAYERS], dec_beta[ONTOVAE_VAE_MAX_LAYERS];
    int64_t enc_bn[ONTOVAE_VAE_MAX_LAYERS], dec_bn[ONTOVAE_VAE_MAX_LAYERS];   /* into bn_state: running_mean, then running_var */
    int64_t Wh;      /* [2 Z, enc[-1]]: rows 0..Z-1 encoder.mu.0.weight, rows Z..2Z-1 encoder.logvar.0.weight */
    int64_t bh;      /* [2 Z] */
    int64_t Wout;    /* [G, dec[-1]]  decoder.decoder[n_dec].0.weight */
    int64_t bout;    /* [G] */
    int64_t total;   /* floats in P */
    int64_t bn_total;/* floats in bn_state */
} ontovae_vae_layout_t;

typedef struct ontovae_vae_ctx {
    ontovae_vae_dims_t dims;
    float* P; float* M; float* V; float* Gr;    /* M, V: training with apply_update; Gr: apply_update == 0 */
    float* bn_state;                            /* [bn_total] */
    int64_t* bn_nbt;                            /* [n_enc + n_dec] */
    void* workspace; int64_t workspace_bytes;   /* ontovae_vae_workspace_bytes(dims, max_batch), 256-B aligned */
    int32_t max_batch;
    float drop, z_drop;                         /* modules.py:48, 58, 63, 123 */
    uint64_t seed;
} ontovae_vae_ctx_t;

int ontovae_vae_layout(const ontovae_vae_dims_t* dims, ontovae_vae_layout_t* out);
int64_t ontovae_vae_workspace_bytes(const ontovae_vae_dims_t* dims, int32_t max_batch);
/* One body of VAE.train_round (vae_model.py:1026-1037): forward, vae_loss (:995-1004), backward, AdamW (:1101; every
 * parameter, no masks, no clamp).  Arguments as ontovae_train_step. */
int ontovae_vae_train_step(const ontovae_vae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B, const float* eps,
                           uint64_t rng_offset, float kl_coeff, const ontovae_adamw_t* opt, int32_t apply_update, float* loss_out,
                           void* stream);
/* VAE.forward (vae_model.py:979-990) / get_embedding (:965-977) and, with loss_out != NULL, one body of val_round
 * (:1061-1065).  Outputs (any may be NULL): recon [B, G], mu, log_var, z [B, Z], sample-major; loss_out: device float[3].
 * With recon == NULL and loss_out == NULL only the encoder and the reparameterisation run (Encoder.forward, modules.py:74-84:
 * the decoder's BatchNorm buffers do not move). */
int ontovae_vae_forward(const ontovae_vae_ctx_t* ctx, const float* x, const int32_t* xidx, int32_t B, const float* eps,
                        uint64_t rng_offset, int32_t train, float kl_coeff, float* recon, float* mu, float* log_var, float* z_out,
                        float* loss_out, void* stream);
/* OntoEncVAE (onto_vae/vae_model.py:468-905): ontology in the encoder, dense Decoder.  The masked encoder is
 * ontovae_decoder_forward / ontovae_decoder_backward on a plan built from the encoder masks (input = the genes, last layer =
 * the 2 Z head rows [mu; log_var]); these two entry points are everything between those calls.  The ctx is a dense VAE ctx
 * whose decoder is the model's (its encoder side is a one-unit dummy that is never touched here).
 * ontovae_vae_decode_forward: OntoEncVAE.forward after the encoder (vae_model.py:557-563) and, with loss_out, the rest of one
 *   val_round body (:654-655).  heads [B, 2 Z] = [mu | log_var] before the heads' Dropout.  Outputs (any may be NULL):
 *   recon [B, G], mu / log_var [B, Z] after that Dropout, z [B, Z]; loss_out: device float[3] {rec, kl, rec + kl_coeff kl}, needs x.
 * ontovae_vae_decode_step: the middle of one train_round body (vae_model.py:603-619): reparameterisation, Decoder, ELBO,
 *   backward down to the heads, AdamW of every decoder parameter (or gradients into ctx->Gr); dheads_out [B, 2 Z] is
 *   dL/d(heads), what ontovae_decoder_backward takes as drec. */
int ontovae_vae_decode_forward(const ontovae_vae_ctx_t* ctx, const float* heads, const float* x, const int32_t* xidx, int32_t B,
                               const float* eps, uint64_t rng_offset, int32_t train, float kl_coeff, float* recon, float* mu,
                               float* log_var, float* z_out, float* loss_out, void* stream);
int ontovae_vae_decode_step(const ontovae_vae_ctx_t* ctx, const float* heads, const float* x, const int32_t* xidx, int32_t B,
                            const float* eps, uint64_t rng_offset, float kl_coeff, const ontovae_adamw_t* opt, int32_t apply_update,
                            float* dheads_out, float* loss_out, void* stream);
/* torch.optim.AdamW.step over the flat buffers with gradients from ctx->Gr (after apply_update == 0). */
int ontovae_vae_adamw_apply(const ontovae_vae_ctx_t* ctx, const ontovae_adamw_t* opt, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-phase timing with HIP events on the launch stream (measurement only; the reference has no
 * counterpart).  ontovae_profile_begin enables the phases in `phase_mask` (bit = phase id) and
 * reserves `capacity` event pairs per phase; every later launch of an enabled phase is bracketed
 * by two events until the capacity is used up.  ontovae_profile_read synchronises on them,
 * returns the elapsed milliseconds of each recorded launch and resets that phase's count.
 * Not thread safe; begin/end allocate and free events, so call them outside timed regions.
 * -------------------------------------------------------------------------------------------*/
enum {
    ONTOVAE_PHASE_ENC_GEMM = 0,      /* x W0^T (split-K MFMA)                       */
    ONTOVAE_PHASE_BN_FWD = 1,        /* split-K combine + BatchNorm + ReLU          */
    ONTOVAE_PHASE_HEADS = 2,         /* mu / log_var heads                          */
    ONTOVAE_PHASE_REPARAM = 3,       /* head slabs + bias, z_drop, reparameterisation, KL (+ packed-weight refresh) */
    ONTOVAE_PHASE_DEC_FWD_TERMS = 4, /* term levels of the decoder (K-1 launches)   */
    ONTOVAE_PHASE_DEC_FWD_GENE = 5,  /* gene layer + ELBO epilogue                  */
    ONTOVAE_PHASE_DEC_BWD = 6,       /* decoder backward data, term levels (+ riders: dW of the gene rows) */
    ONTOVAE_PHASE_LOSS = 7,          /* loss partial reduction (own launch in eval only) */
    ONTOVAE_PHASE_DEC_WGRAD = 8,     /* decoder weight gradient + AdamW + clamp (own launches only without riders) */
    ONTOVAE_PHASE_LATENT_BWD = 9,    /* d mu, d log_var, dbh, loss (+ riders: dW partials of the term rows) */
    ONTOVAE_PHASE_DA_GEMM = 10,      /* heads backward data (+ riders: decoder AdamW finish pass) */
    ONTOVAE_PHASE_DWH = 11,          /* heads weight gradient + AdamW (+ BatchNorm backward in the same launch) */
    ONTOVAE_PHASE_BN_BWD = 12,       /* BatchNorm backward + AdamW of gamma/beta/b0 (own launch only without riders / DP) */
    ONTOVAE_PHASE_DW0 = 13,          /* dW0 = dh x (MFMA) + fused AdamW             */
    ONTOVAE_PHASE_DEC_BWD_GENE = 14, /* decoder backward data, gene layer           */
    ONTOVAE_PHASE_SELL_REFRESH = 15, /* packed term-level weight copies <- params   */
    /* Differential timing of dW0 (ontovae_train_step, single stream): with PAIR_A enabled, even-numbered steps bracket the launch
     * in front of dW0 alone (slot PAIR_A), odd-numbered ones that launch and dW0 in ONE bracket with no event between them (slot
     * PAIR_AB, filled whenever PAIR_A is enabled); mean(PAIR_AB) - mean(PAIR_A) = dW0 as it runs in an un-instrumented step. */
    ONTOVAE_PHASE_PAIR_A = 16,
    ONTOVAE_PHASE_PAIR_AB = 17,
    ONTOVAE_PHASE_COUNT = 18
};
int ontovae_profile_begin(uint32_t phase_mask, int32_t capacity);
int ontovae_profile_read(int32_t phase, float* ms, int32_t cap, int32_t* n_out);
int ontovae_profile_end(void);
/* What a bracket costs by itself, on `stream`, with events of the kind the brackets use: the mean elapsed milliseconds
 * between the two events of `n` brackets that hold NOTHING (null_ms) and of `n` that hold one launch of a kernel of one
 * wave that does nothing (empty_ms).  bench.py reports kernel times net of null_ms next to the raw ones.  Synchronises. */
int ontovae_profile_overhead(void* stream, int32_t n, float* null_ms, float* empty_ms);

#ifdef __cplusplus
}
#endif
#endif /* ONTOVAE_HIP_H */
