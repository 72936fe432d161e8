/*
 * daisyrec_amd.h — C ABI of the MI355X-native MF + BPR training hot path.
 *
 * This is the drop-in boundary: every entry point takes plain device pointers,
 * sizes and a HIP stream (void* == hipStream_t).  No torch / C++ types cross
 * it.  The reference (AmazingDD/daisyRec v2.3.0) is pure Python; each function
 * below names the reference interface it replaces (file:line relative to the
 * reference checkout).  INTEGRATION.md shows the ctypes binding a daisyRec
 * maintainer would add.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in `_host`
 *   - tables are row-major fp32  P[user_num][d], Q[item_num][d]
 *     (nn.Embedding.weight, MFRecommender.py:53-54); they are updated IN PLACE
 *   - indices are int32 (`.astype(np.int32)`, sampler.py:101) unless stated
 *   - every call only ENQUEUES work on `stream`; nothing synchronises the host
 *   - return value: 0 = ok, otherwise an error code; daisy_last_error() gives
 *     the message (thread local)
 */
#ifndef DAISYREC_AMD_H
#define DAISYREC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DAISY_ABI_VERSION 7

typedef void *daisy_stream_t; /* hipStream_t */

enum daisy_status {
    DAISY_OK = 0,
    DAISY_ERR_ARG = 1,  /* bad argument (null pointer, size out of range, unsupported d) */
    DAISY_ERR_HIP = 2,  /* a HIP runtime call failed */
    DAISY_ERR_STATE = 3 /* call order violated (e.g. step before set_batch) */
};

/* daisy/utils/loss.py:5-33.  (AbstractRecommender.py:79-93 builds them.) */
enum daisy_loss {
    DAISY_LOSS_BPR = 0, /* -(gamma + sigmoid(pos-neg)).log().sum()          loss.py:10-13 */
    DAISY_LOSS_HL = 1,  /* clamp(1-(pos-neg), min=0).sum()                  loss.py:20-23 */
    DAISY_LOSS_TL = 2,  /* sigmoid(neg-pos).sum()+sigmoid(neg**2).sum()     loss.py:30-33 */
    /* point-wise (AbstractRecommender.py:80-83, MFRecommender.py:75-81): rows are (user, item, label) */
    DAISY_LOSS_CL = 3,  /* nn.BCEWithLogitsLoss(reduction='sum')(pred, label)               */
    DAISY_LOSS_SL = 4   /* nn.MSELoss(reduction='sum')(pred, label)                         */
};

/* how the item-side gradient is accumulated (entries of a batch are always sorted by item) */
enum daisy_item_mode {
    DAISY_ITEM_ATOMIC = 0,  /* one fp32 atomic row per entry (kept for A/B measurements only) */
    DAISY_ITEM_SORTED = 1,  /* one owner per row, entries summed in plan order: bitwise reproducible */
    DAISY_ITEM_CHUNKED = 2, /* segmented reduction through LDS accumulators (throughput mode) */
    DAISY_ITEM_FUSED = 3    /* daisy_bpr_sgd_step / daisy_bpr_fit_epoch_sgd only: the STAGED step (forward fused
                               into the user pass, coefficient-scaled user rows staged for the item pass,
                               item rows committed by the owner of their segment); every loss of loss.py:5-33
                               (point-wise ones on point-wise batches), with or without FM biases (ABI 4).
                               Bitwise reproducible.  After a step inside an epoch loop stats[DAISY_ST_SQ_U_PRE]
                               already holds the NEXT batch's value (it rides on this step's item pass). */
};

/* order of an epoch (what DataLoader(shuffle=...) decides, dataset.py:5-7) */
enum daisy_order_mode {
    DAISY_ORDER_IDENTITY = 0, /* shuffle=False: triples in array order                          */
    DAISY_ORDER_PERM = 1,     /* explicit permutation: perm[p] = triple served at position p     */
    DAISY_ORDER_FEISTEL = 2   /* shuffle=True on the device: keyed bijection of (seed, epoch)    */
};

/* layout of the caller-owned `stats` vector (device, double[DAISY_STATS_LEN]) */
enum daisy_stats_slot {
    DAISY_ST_LOSS_DATA = 0, /* sum of per-sample loss terms                    */
    DAISY_ST_L1_U = 1,      /* |P[u]|_1  over the gathered batch               */
    DAISY_ST_L1_I = 2,      /* |Q[i]|_1                                        */
    DAISY_ST_L1_J = 3,      /* |Q[j]|_1                                        */
    DAISY_ST_SQ_U = 4,      /* sum P[u]^2  (slots 0..6 are SUMS: all-reducible)*/
    DAISY_ST_SQ_I = 5,
    DAISY_ST_SQ_J = 6,
    DAISY_ST_LOSS = 7,      /* total loss of the step (written by finalize)    */
    DAISY_ST_NORM_U = 8,    /* |P[u]|_F (written by finalize)                  */
    DAISY_ST_NORM_I = 9,
    DAISY_ST_NORM_J = 10,
    DAISY_ST_NORM_U_PRE = 11, /* (unused since ABI 3; see DAISY_ST_SQ_U_PRE)                 */
    DAISY_ST_SUM_COEF = 12,   /* sum_b (dL/dpos + dL/dneg) = dL/d bias_ (FM); a batch sum like 0..6:
                                 multi-GPU callers all-reduce it with them before finalize       */
    DAISY_ST_SQ_U_PRE = 13,   /* staged step: sum P[u]^2 over the batch from the row-norm cache, known
                                 BEFORE the user pass (a sum: multi-GPU callers all-reduce it)    */
    DAISY_STATS_LEN = 16
};

const char *daisy_last_error(void);
int daisy_abi_version(void);

/* ------------------------------------------------------------------------
 * Training context: owns the per-step scratch (grouped batch, coefficients,
 * sort buffers, partial sums).  One per (process, GPU).
 * ---------------------------------------------------------------------- */
typedef struct daisy_bpr_ctx daisy_bpr_ctx;

int daisy_bpr_ctx_create(daisy_bpr_ctx **out, int64_t max_batch, int32_t d, int64_t user_num,
                         int64_t item_num);
int daisy_bpr_ctx_destroy(daisy_bpr_ctx *ctx);
/* bytes of device scratch the context holds (for reporting) */
size_t daisy_bpr_ctx_scratch_bytes(const daisy_bpr_ctx *ctx);

/* ------------------------------------------------------------------------
 * Epoch plan: replaces one pass of `for batch in DataLoader(BasicDataset(triples),
 * batch_size, shuffle)` (dataset.py:5-27) + `.to(device)` (MFRecommender.py:71-72,83).
 * One build per epoch lays the epoch out batch by batch in HBM — batch k holds exactly
 * the triples DataLoader's k-th batch would (last batch partial, drop_last=False) —
 * each batch grouped by user, plus its item entries sorted by item, so the step
 * kernels update every table row from a single owner instead of through atomics.
 * ---------------------------------------------------------------------- */
typedef struct daisy_epoch_plan daisy_epoch_plan;

int daisy_epoch_plan_create(daisy_epoch_plan **out, int64_t max_triples, int64_t user_num,
                            int64_t item_num);
int daisy_epoch_plan_destroy(daisy_epoch_plan *plan);
size_t daisy_epoch_plan_bytes(const daisy_epoch_plan *plan);
/* triples: int32 [n_triples][3]; perm: int64 [n_triples], only for DAISY_ORDER_PERM;
 * (seed, epoch) only for DAISY_ORDER_FEISTEL; user_base is subtracted from the user
 * ids (user-sharded tables).  flags: DAISY_PLAN_TRIPLES_USER_SORTED promises that the
 * triple array is sorted by user (CSR order), so grouping a batch by user only needs a
 * stable partition by batch (one radix pass); a wrong promise gives wrong updates. */
#define DAISY_PLAN_TRIPLES_USER_SORTED 1
#define DAISY_PLAN_POINTWISE 2 /* rows are (user, item, label): CL / SL losses */
int daisy_epoch_plan_build(daisy_epoch_plan *plan, const int32_t *triples, int64_t n_triples,
                           const int64_t *perm, int32_t order_mode, uint64_t seed, uint64_t epoch,
                           int64_t batch_size, int32_t user_base, int32_t flags,
                           daisy_stream_t stream);
int64_t daisy_epoch_plan_num_batches(const daisy_epoch_plan *plan);
/* daisy_epoch_plan_build only enqueues; ids outside [0,user_num) x [0,item_num) (after user_base) are
 * replaced by 0 there - never an out-of-bounds access - and remembered.  This call synchronises the stream
 * and returns DAISY_ERR_ARG if the last build saw one (the reference raises IndexError in nn.Embedding,
 * MFRecommender.py:64-65).  daisy_bpr_ctx_validate_batch: the same for the batch last set with
 * daisy_bpr_set_batch / daisy_bpr_set_batch_from_triples. */
int daisy_epoch_plan_validate(const daisy_epoch_plan *plan, daisy_stream_t stream);
int daisy_bpr_ctx_validate_batch(const daisy_bpr_ctx *ctx, daisy_stream_t stream);
/* copy batch k of a built plan into caller buffers (inspection / tests): u,i,j int32 [B]
 * grouped by user; optional item entries [2B] sorted by item: ent_item, ent_s (sample
 * position | 0x80000000 for the negative slot), ent_u (user of that sample).
 * *B_out_host (host pointer, may be NULL) receives the batch size. */
int daisy_epoch_plan_read_batch(const daisy_epoch_plan *plan, int64_t k, int32_t *u, int32_t *i,
                                int32_t *j, int32_t *ent_item, uint32_t *ent_s, int32_t *ent_u,
                                int64_t *B_out_host, daisy_stream_t stream);
/* out[t] = position of triple t in the DAISY_ORDER_FEISTEL order of (seed, epoch) */
int daisy_feistel_positions(int64_t n, uint64_t seed, uint64_t epoch, int64_t *out,
                            daisy_stream_t stream);
/* the same for a subset of the triples: out[k] = position of triple ids[k] among all n (-1 for an id outside 0..n-1):
 * what a rank of a multi-GPU fit needs for its rows (daisy_epoch_plan_build_positions) */
int daisy_feistel_positions_at(const int64_t *ids, int64_t n_ids, int64_t n, uint64_t seed, uint64_t epoch,
                               int64_t *out, daisy_stream_t stream);

/* ------------------------------------------------------------------------
 * Static index of a training set + the PARTITIONED epoch plan (ABI 3).
 * BasicDataset holds one immutable triple array for the whole fit (dataset.py:21); only the batch
 * membership changes per epoch (DataLoader(shuffle=True), dataset.py:5-7).  So the set is indexed once -
 * triples in CSR (user-sorted) order, their 2n item entries sorted by item - and an epoch plan is two
 * stable one-digit partitions of those arrays by batch id (a stable partition of a sorted list leaves
 * every batch sorted): 32 B of plan per interaction instead of 112, one counting + one scatter pass
 * instead of two payload-carrying radix sorts.  The resulting plan feeds the staged step
 * (DAISY_ITEM_FUSED) only.
 * daisy_train_index_create validates 0 <= user - user_base < user_num, 0 <= item < item_num for every
 * triple (one host sync) and returns DAISY_ERR_ARG where the reference raises IndexError in
 * nn.Embedding (MFRecommender.py:64-65).  flags: DAISY_PLAN_TRIPLES_USER_SORTED - the array is already
 * in CSR order and is used in place (the caller keeps it alive); otherwise the index owns a sorted copy;
 * DAISY_PLAN_POINTWISE (ABI 4) - rows are (user, item, label) (CL / SL, sampler.py:93-98): ONE item entry per
 * row, the third column is carried as the label and not validated as an id.
 * ---------------------------------------------------------------------- */
typedef struct daisy_train_index daisy_train_index;
int daisy_train_index_create(daisy_train_index **out, const int32_t *triples, int64_t n_triples,
                             int64_t user_num, int64_t item_num, int32_t user_base, int32_t flags,
                             daisy_stream_t stream);
int daisy_train_index_destroy(daisy_train_index *index);
size_t daisy_train_index_bytes(const daisy_train_index *index);
/* same batches as daisy_epoch_plan_build on the same (order_mode, perm | seed, epoch, batch_size): batch k
 * holds the triples at epoch positions [k*B, (k+1)*B), grouped by user (CSR order inside a batch); its
 * entries are sorted by item, ties in CSR order */
int daisy_epoch_plan_build_indexed(daisy_epoch_plan *plan, const daisy_train_index *index,
                                   const int64_t *perm, int32_t order_mode, uint64_t seed, uint64_t epoch,
                                   int64_t batch_size, daisy_stream_t stream);
/* ONE RANK'S SHARE of an epoch (multi-GPU fit; the reference iterates one DataLoader on one device,
 * dataset.py:5-7 + AbstractRecommender.py:117-129).  `index` holds the rows this rank owns (its user range),
 * positions[r] in [0, n_total) is the place of the caller's row r in the epoch order of ALL n_total rows (distinct
 * values; int64).  Batch k = the held rows with position in [k*B, (k+1)*B): the union of the ranks' batches k is
 * exactly batch k of the single-device epoch.  Batches differ in size (possibly 0 rows:
 * daisy_epoch_plan_batch_rows); num_batches = ceil(n_total / B); one host sync per build.  Contexts that read the
 * plan need max_batch >= batch_size (stage slots are epoch positions minus k*B). */
int daisy_epoch_plan_build_positions(daisy_epoch_plan *plan, const daisy_train_index *index,
                                     const int64_t *positions, int64_t n_total, int64_t batch_size,
                                     daisy_stream_t stream);
/* The item pass of a multi-GPU step in item slices, so that the exchange of a finished slice (reduce-scatter, owner
 * update, all-gather) runs while the next slice is reduced.  item_bounds[0..n_slices] (host; item_bounds[0] = 0,
 * non-decreasing, item_bounds[n_slices] >= item_num; at most 16 slices) cut the items; _slices locates the cuts in
 * the current batch's item-sorted entries (one tiny kernel), _slice(s) writes gQ / cnt of the items of slice s like
 * daisy_bpr_staged_item(gQ, cnt).  Same counts; same gQ up to the order in which the partial sums of a long segment
 * meet (the cuts move workgroup boundaries); the same cuts give the same bits on every rank and in every run. */
int daisy_bpr_staged_item_slices(daisy_bpr_ctx *ctx, const int32_t *item_bounds, int32_t n_slices,
                                 daisy_stream_t stream);
int daisy_bpr_staged_item_slice(daisy_bpr_ctx *ctx, int32_t loss_type, float *gQ, float *cnt, int32_t slice,
                                float lr, float reg_1, float reg_2, const double *stats, daisy_stream_t stream);
/* The staged step with torch.optim.Adam on both tables (AbstractRecommender.py:54; ABI 4), lazy form: the rows the
 * current batch references are first brought to step `step`-1 (zero-gradient replays, see daisy_adam_lazy_catchup),
 * then the user pass and the item pass apply step `step` to the rows they own (moments m*, v*, stamps last*: the
 * state of daisy_adam_lazy_* / ops.LazyAdam; `table` = daisy_adam_lazy_table(lr, ...) copied to the device).  Rows no batch has
 * referenced stay behind until daisy_adam_lazy_flush.  Same arithmetic as daisy_adam_dense on every element.
 * FM biases (daisy_bpr_ctx_set_bias): their gradients go to g_u_bias / g_i_bias / stats[DAISY_ST_SUM_COEF] for the
 * caller's dense optimiser.  Every loss of loss.py:5-33; point-wise ones on point-wise batches. */
int daisy_bpr_staged_adam_step(daisy_bpr_ctx *ctx, float *P, float *Q, int32_t loss_type, float gamma, float lr,
                               float reg_1, float reg_2, float *mP, float *vP, int32_t *lastP, float *mQ, float *vQ, int32_t *lastQ,
                               const float *table, float beta1, float beta2, float eps, int64_t step, double *stats,
                               double *epoch_acc, double *step_loss, daisy_stream_t stream);
/* The same in phases, for a multi-GPU step (user-sharded): the users' rows of P are rank-local
 * and keep the lazy form - catchup_users brings the rows the local batch references to step-1, staged_user_adam is
 * daisy_bpr_staged_user with the Adam owner update; the item side goes through the gradient form of
 * daisy_bpr_staged_item and, after the reduce-scatter, daisy_item_apply_counts_adam: torch's DENSE Adam over the owner's
 * block of Q (every row steps in every step; moments m, v [rows][d] of the block). */
int daisy_bpr_staged_adam_catchup_users(daisy_bpr_ctx *ctx, float *P, float *mP, float *vP, int32_t *lastP,
                                        const float *table, float beta1, float beta2, float eps, int64_t step,
                                        daisy_stream_t stream);
int daisy_bpr_staged_user_adam(daisy_bpr_ctx *ctx, float *P, const float *Q, int32_t loss_type, float gamma, float lr,
                               float reg_1, float reg_2, float *mP, float *vP, int32_t *lastP, float beta1, float beta2,
                               float eps, int64_t step, double *stats, daisy_stream_t stream);
int daisy_item_apply_counts_adam(float *Q, float *g, float *cnt, float *m, float *v, int64_t rows, int32_t d, float lr,
                                 float reg_1, float reg_2, float beta1, float beta2, float eps, int64_t step,
                                 const double *stats, daisy_stream_t stream);
/* rows of batch k held by this plan (host value; -1: no such batch) */
int64_t daisy_epoch_plan_batch_rows(const daisy_epoch_plan *plan, int64_t k);

/* The staged step keeps |P[u]|^2 of every row in the context; every entry point that writes P through the
 * context keeps it current or drops it.  A caller that changes P by other means calls this first. */
int daisy_bpr_ctx_invalidate_cache(daisy_bpr_ctx *ctx);

/* Staged step, user pass: rows of P read and written PAST the caches (nontemporal loads + owner stores) - for user
 * tables so far beyond the 256 MB Infinity Cache that a row returns only once every few steps (then the cache is left to
 * Q and the stage).  mode -1 (default): automatic, on for tables > 512 MB (measured: -3.4 ... -5.1 % per step at
 * 10 M x 64, +0.7 ... +3 % at 1 M x 64); 0: off; 1: on.  Same arithmetic either way (MFRecommender.py:63-97): only the
 * cache policy of the accesses changes. */
int daisy_bpr_ctx_set_p_stream(daisy_bpr_ctx *ctx, int32_t mode);

/* The staged step in phases (what daisy_bpr_sgd_step(DAISY_ITEM_FUSED) runs back to back), so that a
 * multi-GPU step can put its collectives between them:
 *   prenorm : stats[DAISY_ST_SQ_U_PRE] = sum_b |P[u_b]|^2                      -> all-reduce (8 B)
 *   user    : forward + criterion + user-side backward + SGD on the touched rows of P
 *             (MFRecommender.py:63-97, AbstractRecommender.py:125-126); stats[0..6] = the batch sums
 *                                                                              -> all-reduce, daisy_bpr_finalize
 *   item    : item-side backward.  Q != NULL: + SGD on the touched rows of Q in place (needs the finalized
 *             norms in stats); gQ/cnt != NULL instead: gQ[r] = data term of dL/dQ[r] and cnt[r] = (number
 *             of positive, negative entries) f32[I][2] for the touched rows (others untouched: keep both
 *             zero between steps)                                             -> reduce-scatter both
 *   daisy_item_apply_counts : the row owner's SGD step from reduced (g, cnt): adds the regulariser
 *             share from the GLOBAL counts, clears g and cnt. */
int daisy_bpr_staged_prenorm(daisy_bpr_ctx *ctx, const float *P, double *stats, daisy_stream_t stream);
int daisy_bpr_staged_user(daisy_bpr_ctx *ctx, float *P, const float *Q, int32_t loss_type, float gamma,
                          float lr, float reg_1, float reg_2, double *stats, daisy_stream_t stream);
int daisy_bpr_staged_item(daisy_bpr_ctx *ctx, int32_t loss_type, float *Q, float *gQ, float *cnt, float lr,
                          float reg_1, float reg_2, const double *stats, daisy_stream_t stream);
int daisy_item_apply_counts(float *Q, float *g, float *cnt, int64_t rows, int32_t d, float lr, float reg_1,
                            float reg_2, const double *stats, daisy_stream_t stream);

/* batches set with daisy_bpr_set_batch / _from_triples are (user, item, label) rows (CL / SL) */
int daisy_bpr_ctx_set_pointwise(daisy_bpr_ctx *ctx, int32_t pointwise);
/* FM (FMRecommender.py:46-68): score(u,item) = <P[u],Q[item]> + u_bias[u] + i_bias[item] + bias_.
 * Attaches the bias parameters (device, caller-owned: u_bias f32[U], i_bias f32[I], bias f32[1]) to
 * the context; u_bias == NULL detaches them (plain MF).  From then on
 *   daisy_bpr_forward        adds them to both scores and reduces stats[DAISY_ST_SUM_COEF];
 *   daisy_bpr_item_grad*     accumulates g_i_bias[item] = sum of the item's coefficients (f32[I],
 *                            zero between steps like gQ; mandatory);
 *   daisy_bpr_item_sgd_apply does i_bias -= lr*g_i_bias and clears g_i_bias;
 *   daisy_bpr_user_sgd       does u_bias[u] -= lr*sum(cp+cn) and bias -= lr*stats[SUM_COEF];
 *   daisy_bpr_user_grad      writes g_u_bias[u] (f32[U]) and g_bias[0] instead (needed only for this
 *                            call; may be NULL otherwise) - dense Adam then runs daisy_adam_dense on them.
 * The regularisers of FM.calc_loss (FMRecommender.py:77-93) are those of MF: embeddings only.
 * DAISY_ITEM_FUSED (ABI 4): the staged step carries the biases itself - the user pass adds them to the scores and
 * updates u_bias[u] along each user run, the item pass i_bias[item] along each segment, bias from
 * stats[DAISY_ST_SUM_COEF] (SGD in place; daisy_bpr_staged_adam_step writes g_u_bias / g_i_bias instead). */
int daisy_bpr_ctx_set_bias(daisy_bpr_ctx *ctx, float *u_bias, float *i_bias, float *bias,
                           float *g_u_bias, float *g_i_bias, float *g_bias);
/* make batch k of a built plan current (no copy) */
int daisy_bpr_set_batch_from_plan(daisy_bpr_ctx *ctx, const daisy_epoch_plan *plan, int64_t k,
                                  daisy_stream_t stream);
/* One collated batch outside a plan: rows idx[0..B) of the int32 [N,3] triple array
 * (idx == NULL: rows start..start+B); builds a one-batch plan inside the context. */
int daisy_bpr_set_batch_from_triples(daisy_bpr_ctx *ctx, const int32_t *triples, int64_t n_triples,
                                     const int64_t *idx, int64_t start, int64_t B,
                                     int32_t user_base, daisy_stream_t stream);
/* Same from three separate int32 arrays (what default_collate yields). */
int daisy_bpr_set_batch(daisy_bpr_ctx *ctx, const int32_t *u, const int32_t *i, const int32_t *j,
                        int64_t B, daisy_stream_t stream);

/* MF.forward x2 + criterion (MFRecommender.py:63-68,73,83-85; loss.py): per-sample
 * d(loss)/d(pos), d(loss)/d(neg) kept in the context; the seven batch SUMS go
 * to stats[0..6]. */
int daisy_bpr_forward(daisy_bpr_ctx *ctx, const float *P, const float *Q, int32_t loss_type,
                      float gamma, double *stats, daisy_stream_t stream);

/* Regulariser + total loss of MF.calc_loss (MFRecommender.py:88-89,94-95) from the
 * (possibly all-reduced) sums: stats[7..10].  epoch_acc (device double[2], may
 * be NULL): [0] += loss  (`current_loss += loss.item()`, AbstractRecommender.py:128),
 * [1] += isnan(loss)     (AbstractRecommender.py:122-123).
 * step_loss (may be NULL) receives the step's loss. */
int daisy_bpr_finalize(daisy_bpr_ctx *ctx, double *stats, float reg_1, float reg_2,
                       double *epoch_acc, double *step_loss, daisy_stream_t stream);

/* autograd backward w.r.t. embed_item.weight, restricted to the touched rows
 * (AbstractRecommender.py:125): gQ[r] = dL/dQ[r] for every item r of the batch.
 * gQ must be all zero on entry (daisy_bpr_item_sgd_apply / daisy_adam_dense leave
 * it so). */
int daisy_bpr_item_grad(daisy_bpr_ctx *ctx, const float *P, const float *Q, const double *stats,
                        float reg_1, float reg_2, float *gQ, int32_t item_mode,
                        daisy_stream_t stream);

/* The same gradient in two parts (chunked mode only), so that a multi-GPU step can form the
 * data term  sum_e c_e * p_u(e)  while the all-reduce of the batch sums is still in flight, and
 * add the regulariser share (which needs the global norms in stats[8..10]) afterwards. */
int daisy_bpr_item_grad_data(daisy_bpr_ctx *ctx, const float *P, const float *Q, const double *stats,
                             float *gQ, int32_t item_mode, daisy_stream_t stream);
int daisy_bpr_item_grad_reg(daisy_bpr_ctx *ctx, const float *Q, const double *stats, float reg_1,
                            float reg_2, float *gQ, daisy_stream_t stream);

/* backward w.r.t. embed_user.weight + optim.SGD.step on the touched user rows
 * (AbstractRecommender.py:125-126).  Reads Q, so it must run BEFORE the item
 * rows are committed. */
int daisy_bpr_user_sgd(daisy_bpr_ctx *ctx, float *P, const float *Q, const double *stats, float lr,
                       float reg_1, float reg_2, daisy_stream_t stream);
/* same gradient written to gP[U][d] instead (dense optimisers) */
int daisy_bpr_user_grad(daisy_bpr_ctx *ctx, const float *P, const float *Q, const double *stats,
                        float reg_1, float reg_2, float *gP, daisy_stream_t stream);

/* optim.SGD.step on the item table: Q[r] -= lr*gQ[r]; gQ[r] = 0 for the distinct
 * items of the current batch (dense != 0: every row, used after an all-reduce
 * of gQ). */
int daisy_bpr_item_sgd_apply(daisy_bpr_ctx *ctx, float *Q, float *gQ, float lr, int32_t dense,
                             daisy_stream_t stream);

/* torch.optim.Adam.step (defaults, AbstractRecommender.py:54), DENSE like the
 * reference: every element moves every step.  g is zeroed.  step is 1-based. */
int daisy_adam_dense(float *W, float *g, float *m, float *v, int64_t n, float lr, float beta1,
                     float beta2, float eps, int64_t step, daisy_stream_t stream);

/* The same optimiser without moving every row in every step (MF tables; same result bit for bit).  A row's Adam
 * sequence depends only on its own gradients, so rows without a gradient are left behind and replayed in registers
 * when they are next needed: last[r] (int32, zero-initialised) = the step row r has been updated to.
 *   daisy_adam_lazy_table    host: table[2*s], table[2*s+1] = lr/(1-beta1^s), sqrt(1-beta2^s) for s = 1..n_steps
 *                            (float[2*(n_steps+1)], the host arithmetic of daisy_adam_dense); upload it once.
 *   daisy_adam_lazy_catchup  before the forward pass of step `step`: every row the current batch references is
 *                            brought to step-1 (zero-gradient steps replayed).
 *   daisy_adam_lazy_step     after the gradients: the referenced rows take step `step` with gP / gQ (cleared).
 *   daisy_adam_lazy_flush    all rows of one table to `step` (end of an epoch / of fit, before anything else reads
 *                            the table).
 * The batch is the context's current one (sorted plan layout or daisy_bpr_set_batch*). */
int daisy_adam_lazy_table(float lr, float beta1, float beta2, int64_t n_steps, float *table_host);
int daisy_adam_lazy_catchup(daisy_bpr_ctx *ctx, float *P, float *mP, float *vP, int32_t *lastP, float *Q, float *mQ,
                            float *vQ, int32_t *lastQ, const float *table, float beta1, float beta2, float eps,
                            int64_t step, daisy_stream_t stream);
int daisy_adam_lazy_step(daisy_bpr_ctx *ctx, float *P, float *gP, float *mP, float *vP, int32_t *lastP, float *Q,
                         float *gQ, float *mQ, float *vQ, int32_t *lastQ, const float *table, float beta1, float beta2,
                         float eps, int64_t step, daisy_stream_t stream);
int daisy_adam_lazy_flush(float *W, float *m, float *v, int32_t *last, int64_t rows, int32_t d, const float *table,
                          float beta1, float beta2, float eps, int64_t step, daisy_stream_t stream);

/* torch.optim.Adagrad.step / torch.optim.RMSprop.step with torch's defaults (AbstractRecommender.py:58,60;
 * Adagrad: lr_decay 0, eps 1e-10; RMSprop: alpha 0.99, eps 1e-8, no momentum), dense like the reference;
 * g is zeroed.  (optim.SparseAdam, :62, refuses the reference's dense embedding gradients at its first step:
 * the host mirror raises the same RuntimeError.) */
int daisy_adagrad_dense(float *W, float *g, float *state_sum, int64_t n, float lr, float eps,
                        daisy_stream_t stream);
int daisy_rmsprop_dense(float *W, float *g, float *square_avg, int64_t n, float lr, float alpha, float eps,
                        daisy_stream_t stream);

/* One whole `zero_grad / calc_loss / backward / SGD.step` on one GPU
 * (AbstractRecommender.py:119-128) for the batch set by daisy_bpr_set_batch*. */
int daisy_bpr_sgd_step(daisy_bpr_ctx *ctx, float *P, float *Q, int32_t loss_type, float gamma,
                       float lr, float reg_1, float reg_2, float *gQ, double *stats,
                       double *epoch_acc, double *step_loss, int32_t item_mode,
                       daisy_stream_t stream);

/* The inner `for batch in pbar` loop of GeneralRecommender.fit
 * (AbstractRecommender.py:118-128) for one epoch: every batch of a built plan,
 * enqueued natively with no host synchronisation.  step_losses (may be NULL)
 * gets one loss per step. */
int daisy_bpr_fit_epoch_sgd(daisy_bpr_ctx *ctx, const daisy_epoch_plan *plan, float *P, float *Q,
                            int32_t loss_type, float gamma, float lr, float reg_1, float reg_2,
                            float *gQ, double *stats, double *epoch_acc, double *step_losses,
                            int32_t item_mode, daisy_stream_t stream);

/* The same loop with torch.optim.Adam (AbstractRecommender.py:54,118-128; ABI 6): every batch of a built plan through
 * daisy_bpr_staged_adam_step - steps first_step, first_step + 1, ... (the constants table must hold table_steps >=
 * first_step + num_batches - 1 steps) - enqueued natively: a reference run at its default batch (256 ... a few thousand
 * rows) is not bound by one host round trip per batch.  flush != 0: the rows no batch referenced are brought up to the
 * epoch's last step (daisy_adam_lazy_flush on both tables), i.e. both tables equal the dense optimiser's after the
 * epoch.  MF only: contexts with FM biases are driven step by step (their biases step through the caller's dense
 * optimiser between two steps) and are refused. */
int daisy_bpr_fit_epoch_adam(daisy_bpr_ctx *ctx, const daisy_epoch_plan *plan, float *P, float *Q, int32_t loss_type,
                             float gamma, float lr, float reg_1, float reg_2, float *mP, float *vP, int32_t *lastP,
                             float *mQ, float *vQ, int32_t *lastQ, const float *table, int64_t table_steps, float beta1,
                             float beta2, float eps, int64_t first_step, int32_t flush, double *stats,
                             double *epoch_acc, double *step_losses, daisy_stream_t stream);

/* ABI 7.  1 when the epochs of `plan` (built, sorted layout: daisy_epoch_plan_build) run inside ONE persistent workgroup
 * (csrc/bpr_small.hip: batches of at most 256 samples - the reference's default, basic.yaml:23 -, pairwise losses, no FM
 * biases, rows that fit the LDS); daisy_bpr_fit_epoch_sgd takes that path by itself, and daisy_bpr_fit_epoch_adam accepts
 * a sorted-layout plan exactly when this returns 1 (AbstractRecommender.py:54,103-137: the loop body at B = 256 is bound
 * by kernel boundaries, not by bytes: SGD 8.7 us, Adam 48 -> 35 us per step at ml-100k shapes).  Bit 0: supported; bit 1
 * (value 3): the Adam form is also expected to beat the chain of launches - tables of at most 32 rows per sample of a batch,
 * so that the rows a step references sat out a few steps, not thousands (its zero-gradient replay is the cost). */
int daisy_bpr_small_epoch_supported(const daisy_bpr_ctx *ctx, const daisy_epoch_plan *plan, int32_t loss_type);

/* ------------------------------------------------------------------------
 * Scoring / ranking  (MFRecommender.py:99-133)
 * ---------------------------------------------------------------------- */
/* MF.forward (MFRecommender.py:63-68): out[b] = <P[u_b], Q[i_b]> */
int daisy_mf_predict(const float *P, const float *Q, int32_t d, const int64_t *u, const int64_t *i,
                     int64_t B, float *out, daisy_stream_t stream);
size_t daisy_mf_rank_workspace_bytes(int64_t B, int64_t C);
/* MF.rank inner loop (MFRecommender.py:109-121): scores = bmm, argsort descending
 * (stable), gather candidate ids, first topk -> out_ids int64 [B][topk].
 * scores_out (may be NULL) receives the fp32 [B][C] score matrix. */
int daisy_mf_rank_topk(const float *P, const float *Q, int32_t d, const int64_t *us,
                       const int64_t *cands, int64_t B, int64_t C, int32_t topk, int64_t *out_ids,
                       float *scores_out, void *workspace, size_t workspace_bytes,
                       daisy_stream_t stream);
size_t daisy_mf_full_rank_workspace_bytes(int64_t item_num);
/* FM.predict / rank / full_rank (FMRecommender.py:95-133): the MF entry points plus
 * `+ (u_bias[u] + i_bias[item]) + bias[0]` on every score, in the reference's order of additions;
 * same workspaces; u_bias == NULL gives the MF result. */
int daisy_fm_predict(const float *P, const float *Q, const float *u_bias, const float *i_bias,
                     const float *bias, int32_t d, const int64_t *u, const int64_t *i, int64_t B,
                     float *out, daisy_stream_t stream);
int daisy_fm_rank_topk(const float *P, const float *Q, const float *u_bias, const float *i_bias,
                       const float *bias, int32_t d, const int64_t *us, const int64_t *cands,
                       int64_t B, int64_t C, int32_t topk, int64_t *out_ids, float *scores_out,
                       void *workspace, size_t workspace_bytes, daisy_stream_t stream);
int daisy_fm_full_rank(const float *P, const float *Q, const float *u_bias, const float *i_bias,
                       const float *bias, int32_t d, int64_t item_num, int64_t u, int32_t topk,
                       int64_t *out_ids, void *workspace, size_t workspace_bytes,
                       daisy_stream_t stream);
/* MF.full_rank (MFRecommender.py:126-133): argsort(P[u] @ Q^T, descending)[:topk] */
int daisy_mf_full_rank(const float *P, const float *Q, int32_t d, int64_t item_num, int64_t u,
                       int32_t topk, int64_t *out_ids, void *workspace, size_t workspace_bytes,
                       daisy_stream_t stream);

/* ------------------------------------------------------------------------
 * Uniform negative sampler (sampler.py:55-103, uniform branch :82-89)
 * ---------------------------------------------------------------------- */
size_t daisy_csr_workspace_bytes(int64_t n);
/* get_ur (utils.py:19-34) as a CSR: for n (user,item) pairs build indptr int64[U+1]
 * and the per-user SORTED item lists int32[n] (pairs must be duplicate free,
 * as loader.py:201 guarantees). */
int daisy_build_user_csr(const int32_t *users, const int32_t *items, int64_t n, int64_t user_num,
                         int64_t *indptr, int32_t *csr_items, void *workspace,
                         size_t workspace_bytes, daisy_stream_t stream);
/* js[u][k], k<num_ng, for EVERY user id (sampler.py:63,84-89): uniform with
 * replacement over {0..item_num-1} minus the user's CSR row; -1 if that set is
 * empty.  Counter-based Philox4x32-10: stream = epoch, index = u*num_ng+k. */
int daisy_sample_neg_per_user(const int64_t *indptr, const int32_t *csr_items, int64_t user_num,
                              int64_t item_num, int32_t num_ng, uint64_t seed, uint64_t epoch,
                              int32_t *js, daisy_stream_t stream);
/* SkipGramNegativeSampler.sampling (sampler.py:133-155, Item2Vec's sampler): the user sequences (items in train-set
 * order, seq_items[n] with seq_ptr[U+1] and the owner seq_user[n] of every element), the window half-width, the
 * users' train rows as a CSR with sorted distinct items.  Element e writes rows [row_offsets[e], row_offsets[e+1]) of
 * out (int32 [rows][3]): its (target, context, 1) rows in window order, then as many (target, negative, 0) rows with
 * negatives uniform from the complement of its user's row (Philox(seed, stream_id, row/2 + k)); row_offsets = the
 * exclusive scan of 2 * (window size of e).  bad_flag |= 1 when a user has no negative to draw. */
int daisy_skipgram_samples(const int32_t *seq_items, const int32_t *seq_user, const int64_t *seq_ptr,
                           const int64_t *row_offsets, int64_t n, int32_t context_window, const int64_t *ur_indptr,
                           const int32_t *ur_items, int64_t item_num, uint64_t seed, uint64_t stream_id, int32_t *out,
                           int32_t *bad_flag, daisy_stream_t stream);
/* sampler.py:76-80, the 'high-pop' / 'low-pop' share of a user's negatives: k draws per row from the categorical
 * distribution whose inclusive cumulative sums are cdf[0..item_num) (float64, any positive total), by inverse CDF on
 * Philox(seed, stream_id, row*k + c); written to out[row*ld + col0 + c].  Like np.random.choice(p=...) in the
 * reference, the draws do not exclude the user's positives. */
int daisy_sample_categorical(const double *cdf, int64_t item_num, int64_t rows, int32_t k, uint64_t seed,
                             uint64_t stream_id, int32_t *out, int32_t ld, int32_t col0, daisy_stream_t stream);
/* df.explode('neg_set') (sampler.py:91,100-101): triples int32 [n*num_ng][3] in
 * train-set row order, each interaction repeated num_ng times. */
int daisy_expand_triples(const int32_t *users, const int32_t *items, int64_t n, const int32_t *js,
                         int32_t num_ng, int32_t *triples, daisy_stream_t stream);
/* per-interaction variant (fresh negatives for every triple, re-drawable per
 * epoch): rewrites column 2 of triples [n][3] in place. */
int daisy_resample_neg_per_interaction(const int64_t *indptr, const int32_t *csr_items,
                                       int64_t item_num, int32_t *triples, int64_t n,
                                       uint64_t seed, uint64_t epoch, daisy_stream_t stream);

/* build_candidates_set (utils.py:53-85): for each of n_users test users (ids in `users`)
 * cand_num candidates int64 [n_users][cand_num]: uniform negatives (with replacement) from
 * the items in neither the user's test CSR row nor train CSR row, then the test items in
 * ascending order; a user with more than cand_num test items gets cand_num draws from them.
 * The two rows of a user must be disjoint (they are, by construction of the split). */
int daisy_build_candidates(const int64_t *indptr_test, const int32_t *items_test,
                           const int64_t *indptr_train, const int32_t *items_train,
                           const int64_t *users, int64_t n_users, int64_t item_num,
                           int32_t cand_num, uint64_t seed, int64_t *out, daisy_stream_t stream);

/* ------------------------------------------------------------------------
 * Device-side epoch order (replaces RandomSampler's torch.randperm for the
 * throughput loader; dataset.py:5-7 shuffle=True): perm = a uniformly random
 * permutation of 0..n-1 derived from (seed, epoch) by sorting Philox keys.
 * ---------------------------------------------------------------------- */
size_t daisy_randperm_workspace_bytes(int64_t n);
int daisy_randperm(int64_t n, uint64_t seed, uint64_t epoch, int64_t *perm, void *workspace,
                   size_t workspace_bytes, daisy_stream_t stream);

/* -------------------------------------------------------------------------
 * NeuMF (daisy/model/NeuMFRecommender.py:15-233)
 *   GMF branch  g = uG[u] * iG[item]                                  (:119-122)
 *   MLP branch  x0 = [uM[u] | iM[item]];  x_l = ReLU(Linear_l(Dropout(x_{l-1}))), l = 1..L   (:60-66,123-127)
 *   pred = predict_layer(concat(g, x_L))            (model GMF / MLP: one branch only)  (:129-137)
 * The MLP tower runs on the matrix cores (fp32 MFMA tiles, bias+ReLU+dropout fused into the
 * epilogue; backward data/weight GEMMs with the ReLU gate fused); gathers, the loss epilogue
 * (shared with MF: daisy_loss ids) and the embedding-gradient scatter are HBM-bound kernels.
 * Parameters stay caller-owned (the nn.Module's tensors), passed as a table of device pointers.
 * ---------------------------------------------------------------------- */
#define DAISY_NEUMF_MAX_LAYERS 8
typedef enum { DAISY_NEUMF_FULL = 0, DAISY_NEUMF_GMF = 1, DAISY_NEUMF_MLP = 2 } daisy_neumf_model;
typedef struct {
    float *uG, *iG;                       /* embed_user_GMF [U,d], embed_item_GMF [I,d]            */
    float *uM, *iM;                       /* embed_user_MLP [U,dm], embed_item_MLP [I,dm], dm = d*2^(L-1) */
    float *W[DAISY_NEUMF_MAX_LAYERS];     /* Linear l weight [n_l/2, n_l] row-major, n_1 = 2*dm (:61-65) */
    float *b[DAISY_NEUMF_MAX_LAYERS];     /* Linear l bias   [n_l/2]                                */
    float *Wp, *bp;                       /* predict_layer weight [1, d or 2d], bias [1]   (:68-73) */
} daisy_neumf_params;
/* stats vector of a training step (device, double[DAISY_NEUMF_STATS_LEN]) */
enum {
    DAISY_NST_LOSS_DATA = 0,                         /* sum of the criterion terms                  */
    DAISY_NST_L1 = 1,  /* +0 uG[u], +1 uM[u], +2 iG[i], +3 iM[i], +4 iG[j]: sum |x| over the batch */
    DAISY_NST_SQ = 6,  /* same five, sum x^2                                                        */
    DAISY_NST_LOSS = 11,                             /* NeuMF.calc_loss value (:139-169)             */
    DAISY_NST_NORM = 12,                             /* same five, Frobenius norms                   */
    DAISY_NST_LOSS_SUM = 17,  /* running sum of DAISY_NST_LOSS over the steps since the caller zeroed it: a step   */
                              /* clears slots 0..16 only (an epoch's loss without a launch per step to add it up) */
    DAISY_NEUMF_STATS_LEN = 24
};
typedef struct daisy_neumf_ctx daisy_neumf_ctx;
/* activation workspace for up to max_rows (user,item) pairs per call (a training batch of B
 * pairwise samples forwards 2B rows).  factors % 4 == 0, 1 <= num_layers <= DAISY_NEUMF_MAX_LAYERS. */
int daisy_neumf_ctx_create(daisy_neumf_ctx **out, int64_t max_rows, int32_t factors, int32_t num_layers,
                           int32_t model, int64_t user_num, int64_t item_num);
int daisy_neumf_ctx_destroy(daisy_neumf_ctx *ctx);
size_t daisy_neumf_ctx_bytes(const daisy_neumf_ctx *ctx);
/* bf16_gemm = 1: the MLP tower's GEMMs round their fp32 operands to bf16 on the way into LDS and run
 * v_mfma_f32_32x32x16_bf16 (fp32 accumulate) wherever the tile shape allows (128-row / 64|128-column
 * multiples, K multiple of 32, 16-byte aligned operands; other shapes stay fp32).
 * bf16_gemm = 2 (BASELINE configs[3] "MLP via MFMA bf16"): the activations X_l, the back-propagated dZ_l and a
 * per-call copy of the MLP weights are STORED as bf16 in HBM as well (half the operand traffic, which is what
 * bounds level 1); embeddings, GMF branch, weight gradients and the optimiser stay fp32.  Applies to calls
 * whose row count is a multiple of 128 and whose layer widths are multiples of 64, level 1 otherwise.
 * Throughput modes: results differ from the fp32 parity mode by bf16 rounding (~3 significant digits).
 * Default: 0. */
int daisy_neumf_ctx_set_precision(daisy_neumf_ctx *ctx, int32_t bf16_gemm);
/* NeuMF.forward in eval mode (:118-137) on n pairs -> out f32[n]; processed in chunks of max_rows.
 * Pairs are given in one of three layouts (the three callers of the reference):
 *   users/items i64[n]                                   predict (:171-176), forward
 *   users i64[B], items = cands i64[B*C], C > 0          rank  (:178-209): user of pair e = users[e / C]
 *   users i64[1], items == NULL, C == 0                  full_rank (:211-233): item of pair e = e   */
int daisy_neumf_scores(daisy_neumf_ctx *ctx, const daisy_neumf_params *params, const int64_t *users,
                       const int64_t *items, int64_t n, int64_t C, float *out, daisy_stream_t stream);
/* One training batch of NeuMF.calc_loss + backward (:139-169): rows (u, i, j) int32[B] (point-wise
 * losses: j = label).  ACCUMULATES the dense gradients into `grads` (same table layout; zero before
 * the first step, the optimiser kernels below clear what they consume) and writes stats.
 * dropout_p in [0,1): keep masks come from a counter hash of (seed, layer, row, column) - a fresh
 * `seed` per step gives fresh masks; p == 0 is the deterministic path the golden vectors pin. */
int daisy_neumf_step_grads(daisy_neumf_ctx *ctx, const daisy_neumf_params *params,
                           const daisy_neumf_params *grads, const int32_t *u, const int32_t *i,
                           const int32_t *j, int64_t B, int32_t loss_type, float gamma, float reg_1,
                           float reg_2, float dropout_p, uint64_t seed, double *stats,
                           daisy_stream_t stream);
/* One epoch of the reference's training loop (AbstractRecommender.py:112-128: zero_grad / calc_loss / backward /
 * optimizer.step per batch) over the batches [s, s+batch) of the n samples (u, i, j), issued from the library: step k
 * (1-based, counted on from step0) = daisy_neumf_step_grads with seed = seed_hi | (step0 + k), then the dense optimiser
 * (0 SGD, 1 Adam, 2 Adagrad, 3 RMSprop: torch defaults, the daisy_*_dense kernels) on the flat parameter vector W and
 * its gradient g (n_flat floats: every tensor of `params` / `grads` is a view into them; state0 / state1: exp_avg /
 * exp_avg_sq, state_sum, square_avg; Adam's step count is step0 + k).  stats[DAISY_NST_LOSS_SUM] grows by the
 * steps' losses.  Same kernels as the per-step calls - what it removes is the host's per-step work. */
int daisy_neumf_fit_epoch(daisy_neumf_ctx *ctx, const daisy_neumf_params *params, const daisy_neumf_params *grads,
                          const int32_t *u, const int32_t *i, const int32_t *j, int64_t n, int64_t batch, int32_t loss_type,
                          float gamma, float reg_1, float reg_2, float dropout_p, uint64_t seed_hi, int64_t step0,
                          int32_t optimizer, float lr, float *W, float *g, float *state0, float *state1, int64_t n_flat,
                          double *stats, daisy_stream_t stream);
/* optim.SGD step on one dense tensor: W -= lr*g; g = 0   (AbstractRecommender.py:56) */
int daisy_sgd_dense(float *W, float *g, int64_t n, float lr, daisy_stream_t stream);
/* the argsort / top-k tail of every rank(): scores f32[B,C] (+ candidate ids i64[B,C]) -> ids of the
 * topk best per row, stable descending like torch.argsort(descending=True); workspace as
 * daisy_mf_rank_workspace_bytes(B, C).  full-rank variant: scores f32[I] -> item ids i64[topk],
 * workspace as daisy_mf_full_rank_workspace_bytes(I). */
int daisy_topk_from_scores(const float *scores, const int64_t *cands, int64_t B, int64_t C, int32_t topk,
                           int64_t *out_ids, void *workspace, size_t workspace_bytes,
                           daisy_stream_t stream);
int daisy_full_topk_from_scores(const float *scores, int64_t item_num, int32_t topk, int64_t *out_ids,
                                void *workspace, size_t workspace_bytes, daisy_stream_t stream);
/* C[M,N] = A[M,K] * B[N,K]^T on the fp32 MFMA tile kernel the MLP tower uses (test / bench hook) */
int daisy_gemm_nt_f32(const float *A, const float *B, float *C, int64_t M, int32_t N, int32_t K,
                      daisy_stream_t stream);
/* C[M,N] = A[M,K] * B[N,K]^T with all three matrices stored as bf16 (fp32 accumulate, round to nearest even): the
 * GEMM of precision level 2 (test / bench hook); M % 128 == 0, N % 64 == 0 (% 128 when N > 64), K % 32 == 0 */
int daisy_gemm_nt_bf16(const uint16_t *A, const uint16_t *B, uint16_t *C, int64_t M, int32_t N, int32_t K,
                       daisy_stream_t stream);
/* C[M,N] (fp32, accumulated into: zero it first) += At[K,M]^T * Bt[K,N] with both operands stored as bf16 and
 * contiguous along their rows, the reduction over K cut into k_chunk slices that add with fp32 atomics: the
 * weight-gradient GEMM of precision level 2, gW = dZ^T X (NeuMFRecommender.py:139-169 through autograd; test / bench
 * hook); M % 128 == 0, N % 64 == 0 (% 128 when N > 64), K % 32 == 0, k_chunk % 32 == 0 */
int daisy_gemm_tn_bf16(const uint16_t *At, const uint16_t *Bt, float *C, int64_t M, int32_t N, int64_t K,
                       int64_t k_chunk, daisy_stream_t stream);
/* same with the precision switch of daisy_neumf_ctx_set_precision */
int daisy_gemm_nt(const float *A, const float *B, float *C, int64_t M, int32_t N, int32_t K, int32_t bf16,
                  daisy_stream_t stream);

/* -------------------------------------------------------------------------
 * LightGCN (daisy/model/LightGCNRecommender.py:17-210)
 *   A_hat = D^-1/2 A D^-1/2 over the N = U + I nodes of the bipartite interaction graph  (:74-107)
 *   out   = mean(E_0, A_hat E_0, ..., A_hat^L E_0),  E_0 = [embed_user; embed_item]        (:117-129)
 * and the BPR (or HL/TL/CL/SL) loss of MF on the rows of `out`, regularisers on the rows of E_0
 * (:131-169).  The propagation is a sparse x dense product: HBM-bound row gathers + a segmented
 * reduction, run on the same kernel as MF's item pass.  The loss / gradient-wrt-out part IS the MF
 * path (daisy_bpr_forward / _item_grad_data / _user_grad on the two halves of `out`); these entry
 * points add the graph, the propagation and its transpose (A_hat is symmetric).
 * ---------------------------------------------------------------------- */
typedef struct daisy_lgcn_graph daisy_lgcn_graph;
/* build A_hat from the training interactions (device int32[n] each; duplicate pairs collapse like the
 * reference's dict, :88-90).  Values are formed in float64 and stored as float32 like scipy -> torch
 * (:93-105).  Synchronises the stream once (number of distinct pairs). */
int daisy_lgcn_graph_create(daisy_lgcn_graph **out, const int32_t *users, const int32_t *items, int64_t n,
                            int64_t user_num, int64_t item_num, daisy_stream_t stream);
int daisy_lgcn_graph_destroy(daisy_lgcn_graph *g);
int64_t daisy_lgcn_graph_nnz(const daisy_lgcn_graph *g);          /* stored entries = 2 x distinct pairs */
/* flag != 0: the products use a row-owner kernel that sums every row's entries in stored order (bitwise
 * reproducible, slower on long rows) instead of the chunked segmented reduction (whose rows that span
 * three or more chunks are combined with fp32 atomics in arrival order).  Default 0. */
int daisy_lgcn_graph_set_reproducible(daisy_lgcn_graph *g, int32_t flag);
size_t daisy_lgcn_graph_bytes(const daisy_lgcn_graph *g);
/* COO copy of A_hat, rows then columns ascending (inspection / tests): int32[nnz], int32[nnz], f32[nnz] */
int daisy_lgcn_graph_read(const daisy_lgcn_graph *g, int32_t *row, int32_t *col, float *val,
                          daisy_stream_t stream);
/* Y = A_hat X, X and Y f32[N, d] (Y != X) */
int daisy_lgcn_spmm(const daisy_lgcn_graph *g, const float *X, float *Y, int32_t d, daisy_stream_t stream);
/* Rows [row_lo, row_hi) of the same product: Yrows[r - row_lo] = (A_hat X)[r] - the per-rank share of a
 * row-sharded multi-GPU propagation (BASELINE configs[4]; the ranks all-gather their row blocks per layer).
 * Yrows must be preceded and followed by one spare row of d floats (they may be overwritten).  The first call
 * on a graph synchronises the stream once (row offsets). */
int daisy_lgcn_spmm_rows(const daisy_lgcn_graph *g, const float *X, float *Yrows, int32_t d, int64_t row_lo,
                         int64_t row_hi, daisy_stream_t stream);
/* LightGCN.forward (:117-129): out = mean_k A_hat^k E0;  work: f32[2*N*d] scratch */
int daisy_lgcn_propagate(const daisy_lgcn_graph *g, const float *E0, int32_t d, int32_t num_layers,
                         float *work, float *out, daisy_stream_t stream);
/* its transpose applied to the gradient G = dL/d out:  dE0 += 1/(L+1) sum_k A_hat^k G */
int daisy_lgcn_backprop(const daisy_lgcn_graph *g, const float *G, int32_t d, int32_t num_layers, float *work,
                        float *dE0, daisy_stream_t stream);
/* regulariser gradient on the ego rows of one batch (:150-163): for every sample
 * dE0[row] += reg_1*sign(e) + reg_2*e/|rows|_F for row in (u, U+i, U+j [pairwise only]); the three
 * Frobenius norms are stats[DAISY_ST_NORM_U/I/J] of a finalized MF context whose sums were taken on E0.
 * Deterministic: occurrences are counted with integer atomics into count_ws (int32[2*(U+I)], all zero on
 * entry, left all zero), then every touched row is updated once. */
int daisy_lgcn_reg_grad(const float *E0, const int32_t *u, const int32_t *i, const int32_t *j, int64_t B,
                        int64_t user_num, int64_t item_num, int32_t d, int32_t pointwise, float reg_1,
                        float reg_2, const double *stats, int32_t *count_ws, float *dE0,
                        daisy_stream_t stream);

/* -------------------------------------------------------------------------
 * Item2Vec (rest of SURVEY.md §8f rank 4; daisy/model/Item2VecRecommender.py:15-112).
 * Training IS the point-wise MF path on ONE shared table S (rows (target, context, label), loss CL, no
 * regulariser): daisy_bpr_forward(P = Q = S) -> daisy_bpr_user_grad (targets) + daisy_bpr_item_grad_data
 * (contexts) into two gradient buffers -> daisy_axpby_f32 -> daisy_adam_dense.  After training the user
 * table is user_embedding[u] = sum of S over the user's training items (:56-59): daisy_csr_row_sum on the
 * CSR of daisy_build_user_csr.  predict / rank / full_rank are the MF entry points.
 * ---------------------------------------------------------------------- */
/* y = a*x + b*y over n floats; zero_x != 0 also clears x (a consumed gradient buffer) */
int daisy_axpby_f32(float *x, float a, float b, float *y, int64_t n, int32_t zero_x, daisy_stream_t stream);
/* out[r] = sum_{e in [indptr[r], indptr[r+1])} X[cols[e]]  (rows with no entries are left untouched) */
int daisy_csr_row_sum(const int64_t *indptr, const int32_t *cols, const float *X, int64_t rows, int32_t d,
                      float *out, daisy_stream_t stream);

/* micro-benchmarks of the memory system used to place the kernels on the
 * roofline (tools/membench.py); not part of the reference surface. */
int daisy_membench(int32_t what, float *table, int64_t rows, int32_t d, const int32_t *idx,
                   int64_t n, float *out, daisy_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DAISYREC_AMD_H */
