/*
 * easyrec_hip.h - C ABI of libeasyrec_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for
 * EasyRec's sparse-embedding + feature-interaction + MLP training hot path.
 *
 * The reference (alibaba/EasyRec v0.8.7) owns no device code: every kernel on this path is a
 * TensorFlow op issued from Python.  Each entry point below therefore names the reference call
 * site (file:line under /root/reference) whose TF-op chain it replaces - that is the "FFI" a
 * maintainer would rebind (INTEGRATION.md shows the ctypes / tf.load_op_library stubs).
 *
 * Conventions:
 *   - every function returns 0 on success, non-zero on error; er_last_error() gives the message
 *     (thread-local).  Mirrors the OP_REQUIRES_OK status style of the reference's own C++ ops
 *     (easy_rec/python/ops/src/load_dense_embed.cc:54,125-130).
 *   - the caller owns every buffer.  Pointers are raw device pointers unless the name ends in
 *     `_host`.  The library allocates only inside opaque handles (er_*_create / er_*_destroy).
 *   - device work is asynchronous on the given stream (a hipStream_t passed as void*); no
 *     function synchronises; all are safe to capture in a hipGraph after handle creation.
 *   - fp32 arithmetic is compiled with -ffp-contract=off and IEEE div/sqrt so that results can
 *     be compared bit-for-bit with the step-by-step CPU oracle where the order is defined.
 *   - per-step scalars (learning rate, beta powers) are read from DEVICE memory (er_opt_hyper)
 *     so that a captured graph can be replayed with new values.
 *   - two pieces of library-owned device scratch are shared by all calls of a process: the column-
 *     reduction scratch (er_reserve_scratch) and the split-K workspace (er_gemm_reserve).  Calls that
 *     use them (BatchNorm / colsum / loss reductions; split-K and grouped GEMMs) must therefore be
 *     ordered on ONE stream (or by events); everything else may run on any stream.
 */
#ifndef EASYREC_HIP_H_
#define EASYREC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ER_ABI_VERSION 1

typedef void* er_stream_t; /* hipStream_t */

int er_abi_version(void);
const char* er_last_error(void);
/* Pre-size the library's internal column-reduction scratch (floats).  Call once before capturing
 * a hipGraph: growing it calls hipMalloc, which is not capturable. */
int er_reserve_scratch(int64_t floats);
/* Tuning knobs (process-wide).  Keys: "sweep_blocks_per_cu" (1..8, default 8): workgroups per CU of the
 * dense-decay sweep; lower it when the sweep overlaps other kernels on a second stream. */
int er_config_set(const char* key, int64_t value);
/* number of compute units / XCDs of the current device (launch sizing, reported by bench) */
int er_device_info(int* cu_count, int* wave_size, char* arch_name, int arch_name_len);

/* --------------------------------------------------------------------------------------------
 * K1  id hashing.  Replaces StringToHashBucketFast issued at
 *     easy_rec/python/compat/feature_column/feature_column_v2.py:3915-3921
 *     (HashedCategoricalColumn._transform_input_tensor) and layers/input_layer.py:235,240.
 *     bucket = FarmHash Fingerprint64(bytes) mod num_buckets.
 * Strings are packed: string i = bytes[offsets[i], offsets[i+1]).  Strings are column-major:
 * string i belongs to column i / n_per_col and uses num_buckets[column].
 * drop_empty != 0: an empty string yields -1 (the reference drops '' cells of dense string inputs
 * before hashing: compat/feature_column/feature_column.py:2599-2643; -1 ids are pruned by the
 * lookup, giving the zero vector).  drop_empty == 0: '' is hashed like any other string (Tag /
 * Sequence tokens, which arrive already sparse).
 * -------------------------------------------------------------------------------------------- */
int er_hash_bucket_fast_host(const uint8_t* bytes_host, const int64_t* offsets_host, int64_t n,
                             int64_t n_per_col, const uint64_t* num_buckets_host, int drop_empty,
                             int64_t* out_host);
int er_hash_bucket_fast(const uint8_t* bytes, const int64_t* offsets, int64_t n, int64_t n_per_col,
                        const uint64_t* num_buckets, int drop_empty, int64_t* out,
                        er_stream_t stream);
/* CSVInput's decode step (reference input/csv_input.py:33-76: tf.decode_csv, a TensorFlow C++ kernel), host-side:
 * up to max_rows newline-terminated lines of `text` (blank lines skipped, a trailing "\r" dropped) are split on the ONE
 * separator byte into exactly n_fields cells.  kinds[f]: 0 string, 1 integer, 2 floating point.  Column-major outputs
 * (index f * max_rows + row): int_out / flt_out = the parsed number of a numeric field (0 for an empty cell),
 * empty_out = 1 for an empty cell (the caller substitutes the field's default), str_begin / str_len = the cell's
 * bytes inside `text` (no copy).  n_rows_out = lines decoded, consumed_out = bytes of text they took (an unterminated
 * last line is not consumed).  A wrong field count or a malformed number fails with the line number. */
int er_decode_csv_host(const uint8_t* text_host, int64_t n_bytes, uint8_t separator, int32_t n_fields,
                       const int32_t* kinds_host, int64_t max_rows, int64_t* int_out, double* flt_out,
                       uint8_t* empty_out, int64_t* str_begin, int32_t* str_len, int64_t* n_rows_out,
                       int64_t* consumed_out);
/* ... on n_threads host threads (<= 0: one per hardware thread, at most 8; at least 256 rows per thread): one pass finds
 * the lines, the threads parse disjoint row ranges into the same outputs; plain decimal cells take an inline fast path
 * (integers; decimals of <= 15 significant digits as ONE exact division - the correctly rounded value, i.e. strtod's),
 * everything else strtoll / strtod.  Same results, same errors (the failing line with the smallest index is reported).
 * out_stride (>= max_rows): elements between consecutive fields of the outputs - give it a pitch that is not a power of
 * two (a row's fields at a 4096-element pitch all fall into the same cache sets).
 * tf.decode_csv under tf.data's parallel map (input/csv_input.py:33-76, input/input.py:1057-1110) is the reference's
 * form of the same thing. */
int er_decode_csv_host_mt(const uint8_t* text, int64_t n_bytes, uint8_t sep, int32_t n_fields, const int32_t* kinds,
                          int64_t max_rows, int64_t out_stride, int64_t* int_out, double* flt_out, uint8_t* empty_out,
                          int64_t* str_begin, int32_t* str_len, int64_t* n_rows_out, int64_t* consumed_out,
                          int32_t n_threads);
/* The string cells of a decoded batch (any order, e.g. feature-major) -> packed bytes + offsets[n + 1], the input of
 * er_hash_bucket_fast(_host).  out_bytes holds sum(length) bytes. */
int er_pack_cells_host(const uint8_t* text_host, const int64_t* begin, const int32_t* length, int64_t n,
                       uint8_t* out_bytes, int64_t* out_offsets);
/* n int64 values as decimal strings (Python's str(int): the reference's `_as_string` of an integer column,
 * input/input.py:356-376), packed like er_pack_cells_host's output - hashed IdFeatures fed from integer columns (the
 * Criteo binary format's uint32 categories, input/criteo_input.py:75-85) without a Python object per value.  out_bytes:
 * 20 bytes per value. */
int er_pack_int_decimal_host(const int64_t* values, int64_t n, uint8_t* out_bytes, int64_t* out_offsets);
/* The cells (begin, length) of a text buffer split into tokens as (begin, length) views of the same buffer + row offsets
 * [n + 1] - what tf.string_split / tf.strings.split do to a TagFeature's or SequenceFeature's column (reference
 * input/input.py:488-530, 680-690), in the ragged layout the lookup kernels take.  keep_empty 0 (tags): every byte of
 * `seps` is a delimiter, empty tokens are skipped; keep_empty 1 (sequences, one-byte separator): empty tokens stay, an
 * empty cell is one empty token, at most max_tokens per cell (<= 0: all).  tok_begin / tok_len: `capacity` entries
 * (sum(length) + n always suffices); the token count is left in row_offsets[n]. */
int er_split_cells_host(const uint8_t* text, const int64_t* begin, const int32_t* length, int64_t n, const uint8_t* seps,
                        int32_t n_seps, int32_t keep_empty, int32_t max_tokens, int64_t* tok_begin, int32_t* tok_len,
                        int64_t capacity, int64_t* row_offsets);
/* ComboFeature through `crossed_column` (reference feature_column/feature_column.py:434-445 ->
 * CrossedColumn._transform_feature, compat/feature_column/feature_column_v2.py:4527-4560 -> TF's
 * sparse_cross_hashed): one string per (column, row), column-major (string i = c * n_rows + r);
 *   out[r] = Fold_c FingerprintCat64(h, Fingerprint64(string[c][r])) mod num_buckets, h0 = hash_key
 * (TF's default 0xDECAFCAFFE when the config gives none).  '' is crossed like any other value: the column passes
 * the dense string tensors to the op unfiltered (:4556-4558), unlike the hashed id columns.  Host-side: it belongs to the input pipeline, the ids then go
 * through the lookup like any identity column.  Pinned by the example in the Keras `HashedCrossing` docs. */
int er_sparse_cross_hashed_host(const uint8_t* bytes_host, const int64_t* offsets_host, int64_t n_rows,
                                int32_t n_cols, uint64_t num_buckets, uint64_t hash_key, int64_t* out_host);
/* AsString for integer id columns (feature_column_v2.py:3918, input/input.py:356-376): decimal
 * text of int64 -> hashed directly on device without materialising strings. */
int er_hash_bucket_fast_int64(const int64_t* values, int64_t n, int64_t n_per_col,
                              const uint64_t* num_buckets, int64_t* out, er_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * K2  embedding lookup + combine, all lookups of a model in ONE launch.  Replaces, per embedding
 *     column, the chain  Where/GatherNd -> SparseFillEmptyRows -> Unique -> GatherV2 ->
 *     SparseSegmentSum/Mean/SqrtN | Mul+SegmentSum -> Select -> Reshape -> ConcatV2  issued by
 *       EmbeddingColumn._get_dense_tensor_internal_helper  feature_column_v2.py:3434-3462
 *       (safe_embedding_lookup_sparse, readable copy: compat/embedding_ops.py:37-162)
 *       _internal_input_layer / _get_logits concat         compat/feature_column/feature_column.py:384-414
 *     and the embedding-output L2 term of layers/input_layer.py:369-375 (sum of squares emitted
 *     as per-block partials).
 * -------------------------------------------------------------------------------------------- */
enum { ER_COMBINER_SUM = 0, ER_COMBINER_MEAN = 1, ER_COMBINER_SQRTN = 2 };

typedef struct er_lookup_desc {
  const float* table;     /* row 0 of this lookup's table; [rows, dim] row-major fp32           */
  const int64_t* ids;     /* dense mode: [n_rows]; ragged mode: [nnz]                            */
  const int32_t* offsets; /* NULL = dense mode (one id per output row, id<0 = missing);
                             else CSR row offsets [n_rows+1]                                     */
  const float* weights;   /* NULL or one weight per id                                           */
  float* out;             /* forward: output matrix base; backward: upstream-gradient base      */
  int64_t rows;           /* table rows (ids outside [0, rows) are pruned)                       */
  int64_t key_base;       /* index of table row 0 inside its table group (backward keys)        */
  int32_t dim;            /* embedding dim                                                       */
  int32_t out_stride;     /* floats between consecutive output rows                              */
  int32_t out_col;        /* first output column of this lookup                                  */
  int32_t combiner;       /* ER_COMBINER_*                                                       */
  int32_t n_rows;         /* output rows: batch size, or batch*seq_len for sequence lookups      */
  int32_t max_nnz;        /* capacity of ids[] (dense mode: == n_rows)                           */
  int32_t table_ld;       /* er_emb_fwd only: floats between consecutive table rows, 0 = dim (the rows
                             received from the owners lie side by side with another dim group's)    */
} er_lookup_desc;

typedef struct er_emb_plan er_emb_plan;

/* descs_host: n descriptors (copied).  Lookups may have different dims. */
int er_emb_plan_create(const er_lookup_desc* descs_host, int n, er_emb_plan** plan);
int er_emb_plan_update(er_emb_plan* plan, const er_lookup_desc* descs_host, int n);
int er_emb_plan_destroy(er_emb_plan* plan);
/* number of thread blocks (= number of sum-of-squares partials written by er_emb_fwd) */
int er_emb_plan_num_blocks(const er_emb_plan* plan);
/* sumsq_partials: NULL or [num_blocks] floats; partial b = sum of out^2 written by block b. */
int er_emb_fwd(const er_emb_plan* plan, float* sumsq_partials, er_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * K3+K4  embedding backward + row-wise optimizer for one table group (all tables of one dim
 *     stored back to back: var/m/v are [total_rows, dim]).  Replaces the gradient of the chain
 *     above (UnsortedSegmentSum -> IndexedSlices(unique ids), AddN for shared tables) and the
 *     sparse apply of the optimizer:
 *       tf.train.AdamOptimizer._apply_sparse   (builders/optimizer_builder.py:61-66): m and v of
 *         EVERY row decay each step, every row's var moves        -> ER_OPT_ADAM
 *       AdamOptimizerS._apply_sparse_shared    (compat/adam_s.py:185-213): touched rows only
 *                                                                  -> ER_OPT_LAZY_ADAM
 *       tf.train.AdagradOptimizer sparse apply (optimizer_builder.py:110-116) -> ER_OPT_ADAGRAD
 *       tf.train.GradientDescent / Momentum(0)                     -> ER_OPT_SGD
 *     Pipeline: build 32-bit keys (global row) -> stable radix sort -> two-level in-order
 *     segmented reduction (deterministic) fused with the row update; ER_OPT_ADAM additionally
 *     runs one streaming sweep over the untouched rows (bitmap-skipped).
 * -------------------------------------------------------------------------------------------- */
enum { ER_OPT_SGD = 0, ER_OPT_ADAM = 1, ER_OPT_LAZY_ADAM = 2, ER_OPT_ADAGRAD = 3 };

/* Lives in DEVICE memory; the host refreshes it every step (fp32 scalars computed the way TF
 * computes them, see easyrec_amd/builders/optimizer_builder.py). */
typedef struct er_opt_hyper {
  float lr;            /* scheduled learning rate for this step                                  */
  float lr_t;          /* Adam: lr * sqrt(1 - beta2^t) / (1 - beta1^t)                           */
  float beta1;
  float beta2;
  float one_minus_beta1;
  float one_minus_beta2;
  float eps;
  float grad_scale;    /* multiplies summed row gradients (embedding lr multiplier, 1/world)     */
  float clip_scale;    /* gradient clipping by global norm: multiplies the whole gradient (after grad_scale and
                          the kernel-L2 term); 0 = no clipping.  Written per step by er_clip_scale.              */
  float reserved[7];
} er_opt_hyper;

typedef struct er_emb_group er_emb_group;

/* descs_host: the group's lookups with `out` = gradient base of the matching forward output.
 * var/m/v: [total_rows, dim]; m and/or v may be NULL when the optimizer does not use them
 * (SGD: both; Adagrad: m).  touched_bitmap: [ceil(total_rows/32)] zero-initialised uint32 words,
 * required for ER_OPT_ADAM only. */
/* Per-step scalars without a host round trip: `table` holds n_slots consecutive er_opt_hyper-sized
 * records (floats_per_slot floats each, several optimizers may be packed in one slot) precomputed by
 * the host for the coming steps; this copies slot (*counter % n_slots) to `out` and increments
 * *counter.  Keeps a captured hipGraph free of host-written memory (no race with the host running
 * ahead of the device). */
int er_hyper_select(const float* table, int64_t* counter, int32_t n_slots, int32_t floats_per_slot,
                    float* out, float* history, int64_t history_capacity, int32_t history_index,
                    er_stream_t stream);
/* history != NULL (2 * history_capacity floats): also history[*counter] = slot[history_index] (before the increment):
 * the per-step record of Adam's lr_t that er_emb_catch_up / er_emb_flush_decay replay; and
 * history[history_capacity + *counter] = the largest such value so far (er_emb_group_set_lr_max). */

int er_emb_group_create(const er_lookup_desc* descs_host, int n, int32_t dim, int64_t total_rows,
                        float* var, float* m, float* v, uint32_t* touched_bitmap,
                        er_emb_group** group);
int er_emb_group_update(er_emb_group* group, const er_lookup_desc* descs_host, int n);
int er_emb_group_destroy(er_emb_group* group);
/* total entries (sort length) of the group */
int64_t er_emb_group_num_entries(const er_emb_group* group);
/* backward + update.  opt_kind: ER_OPT_*; hyper: device pointer. */
int er_emb_bwd_update(er_emb_group* group, int opt_kind, const er_opt_hyper* hyper,
                      er_stream_t stream);
/* backward only: writes the de-duplicated gradient: unique_keys [<= num_entries] (global rows,
 * ascending), unique_grads [<= num_entries, dim], *n_unique (device int32).  Used by parity tests
 * and by embedding-parallel training (grads are sent to the row owner instead of applied). */
int er_emb_bwd_reduce(er_emb_group* group, uint32_t* unique_keys, float* unique_grads,
                      int32_t* n_unique, er_stream_t stream);
/* TF-exact Adam WITHOUT the dense sweep ("lazy dense decay").  _apply_sparse decays m, v and moves var of every
 * row at every step, but a row nobody looks up between two of its touches cannot influence anything meanwhile:
 * its decay-only steps are replayed (same fp32 operations, same order, lr_t(s) from the history written by
 * er_hyper_select) right before the next lookup that reads it - bit-identical to the sweep; only once m has
 * settled on its denormal fixed point (~900 idle steps; var then absorbs the update) is the remaining decay of
 * v applied in closed form (<= 1e-6 relative).  Assumes lr_t <= 1.
 *   er_emb_group_enable_lazy_decay: last_step [total_rows] int32 initialised to -1 (device), lr_t_history
 *     [capacity >= number of steps] (device), step_counter (device; the counter er_hyper_select increments).
 *     Afterwards er_emb_bwd_update(ER_OPT_ADAM) updates the touched rows only (no bitmap, no sweep).
 *   er_emb_catch_up: call after er_emb_route (whose unique keys / count it takes) and BEFORE the lookup of
 *     the step; brings the rows the step touches to "after the previous step".
 *   er_emb_flush_decay: brings EVERY row current (checkpoint, state_dict, evaluation of untouched rows).
 *   er_emb_group_set_lr_max: lr_max_history[s] = max of lr_t_history[0..s] (the second half of the buffer
 *     er_hyper_select fills).  Enables the ABSORBED regime of the replay: once a row's update has fallen below a
 *     quarter ulp of var - and, bounded with the largest lr_t so far, can only fall further (beta1 < sqrt(beta2)) -
 *     var stops changing and only m *= beta1, v *= beta2 are replayed: still bit-identical to the sweep, ~10x fewer
 *     instructions per idle step (reached ~150 steps after a row's last touch).
 *   er_emb_flush_window: the ROLLING flush of up to 4 table groups in one launch: step t brings the rows
 *     [w * chunk, (w + 1) * chunk), w = t mod n_windows, chunk = ceil(total_rows / n_windows), current.  Called once
 *     per step (after the row update) it bounds how far behind any row can be to n_windows steps, so the catch-up of a
 *     cold row replays <= n_windows steps instead of its whole idle time, for one pass over 1 / n_windows of the
 *     tables per step.  Same arithmetic as er_emb_flush_decay.  lag 0: call after the step's row update (rows brought
 *     to the step just executed).  lag 1: call DURING step t, after er_emb_catch_up, on ANOTHER stream: rows are
 *     brought to step t-1, which is where the catch-up left the rows the step touches (it records that in last_step),
 *     so the launch has nothing to do on them and overlaps the lookup, the dense part and the row update; join the
 *     stream before the next step's er_emb_route.  max_blocks > 0 caps the grid (workgroups walk the tiles): the
 *     form for the concurrent launch - ~2 workgroups per CU leave the CUs' wave slots and LDS to the step. */
int er_emb_group_enable_lazy_decay(er_emb_group* group, int32_t* last_step, const float* lr_t_history,
                                   const int64_t* step_counter);
/* Row records.  By default var, m, v are three plain [total_rows, dim] arrays and last_step an int32 array of its own:
 * every touched row costs three or four scattered HBM accesses per pass.  er_emb_group_set_row_pitch(group, ld,
 * last_step_ld) tells the group that consecutive rows of var / m / v lie ld >= dim floats apart and consecutive rows'
 * last_step last_step_ld int32 words apart - e.g. ONE record [var(dim) | m(dim) | v(dim) | last_step | pad] per row with
 * var = rec, m = rec + dim, v = rec + 2 dim, last_step = (int32*)(rec + 3 dim), ld = last_step_ld = the record size: a
 * touched row is then one contiguous, aligned access (TF keeps each slot in a variable of its own -
 * tf.train.AdamOptimizer._create_slots; the layout is converted at the state_dict / checkpoint boundary only).  Call
 * after er_emb_group_create (and again if the arrays move); lookups read such a table through er_lookup_desc.table_ld.
 * 16-byte lanes (dim % 4 == 0) need ld % 4 == 0 and 16-byte aligned bases. */
int er_emb_group_set_row_pitch(er_emb_group* group, int64_t ld, int64_t last_step_ld);
int er_emb_group_set_lr_max(er_emb_group* group, const float* lr_max_history);
int er_emb_flush_window(er_emb_group* const* groups_host, int n, int32_t n_windows, int32_t lag, int32_t max_blocks,
                        const er_opt_hyper* hyper,
                        er_stream_t stream);
int er_emb_catch_up(er_emb_group* group, const uint32_t* unique_keys, const int32_t* n_unique,
                    const er_opt_hyper* hyper, er_stream_t stream);
int er_emb_flush_decay(er_emb_group* group, const er_opt_hyper* hyper, er_stream_t stream);
/* CLOSED-FORM replay of the decay-only steps (the default of the training step; the step-by-step replay above stays
 * as the exact mode).  north_star's bar is 1e-4 on logits / loss, not bit-equality of Adam's slots with a dense sweep: k
 * idle steps of a row after step t0 are, with a = sqrt(v0), d = a + eps, z = a / d, w_s = 1 - sqrt(beta2)^s,
 *     var -= (m0 / d) * sum_n z^n T_n(t0, k),   T_n(t0, k) = sum_{s=1..k} lr_t(t0+s) beta1^s w_s^n,
 *     m = m0 beta1^k,  v = v0 beta2^k
 * (csrc/er_decay.h): one sqrt, two divisions and a degree-5 polynomial per element whatever k is - closer to an fp64
 * evaluation of tf.train.AdamOptimizer's recurrence (builders/optimizer_builder.py:61-66, compat/adam_s.py:185-213)
 * than the fp32 step-by-step replay itself (2e-7 against 1e-6 of the update).  The T_n are table lookups:
 *   er_decay_tables_supported: K > 0 (terms kept: beta1^K <= 2e-9, K <= 512) if the betas are in range, else 0.
 *   er_decay_tables_bytes / _create: `buffer` (device, 16-byte aligned, er_decay_tables_bytes(history_capacity) bytes)
 *     holds the coefficient block, the per-launch table A[k] = T(s_end-1-k, k) (rebuilt by a 64-workgroup launch in
 *     front of er_emb_catch_up(_multi) / er_emb_flush_decay / er_emb_flush_window / er_emb_owner_serve) and the
 *     per-step table C[t0] = T(t0, K) that er_step_prologue_decay appends to (rows idle for more than K steps).
 *   er_emb_group_set_decay_tables: the group's replays use the closed form (NULL: back to the exact replay); the group
 *     must have been given the tables' history and counter by er_emb_group_enable_lazy_decay.  No rolling flush
 *     (er_emb_flush_window) is needed: the catch-up costs the same for a row idle 1 step or 100,000. */
typedef struct er_decay_tables er_decay_tables;
int er_decay_tables_supported(float beta1, float beta2);
int64_t er_decay_tables_bytes(int64_t history_capacity);
int er_decay_tables_create(void* buffer, int64_t history_capacity, const float* lr_t_history,
                           const int64_t* step_counter, float beta1, float beta2, er_decay_tables** tables);
int er_decay_tables_destroy(er_decay_tables* tables);
int er_emb_group_set_decay_tables(er_emb_group* group, er_decay_tables* tables);
/* The lag-1 table (what a training step's catch-up / lazy lookup / row update read) built ONE LAUNCH EARLY, by surplus
 * workgroups of the step prologue (er_step_prologue_decay / _hash) instead of by the consumers' own front launch - which
 * lets the step's sort and its lookup share one launch (er_emb_front_fwd).  The prologue increments the step counter in
 * the same launch, so its table workgroups read the step from a word the LOOKUP launch of the previous step left
 * (er_emb_fwd_lazy / er_emb_front_fwd write it; every consumer checks that the table it reads was built for its step and
 * raises a sticky error otherwise).  The owner of the tables must keep that word current whenever a training step ran
 * without such a lookup, or the counter was set: er_decay_tables_sync (stream-ordered: word = *step_counter).
 * er_decay_tables_error: the sticky flag (host read: a sync). */
int er_decay_tables_set_prologue_build(er_decay_tables* tables, int on);
int er_decay_tables_sync(er_decay_tables* tables, er_stream_t stream);
int er_decay_tables_error(er_decay_tables* tables, int32_t* error_host);
/* TF-exact Adam with the sweep OVERLAPPED (two streams).  The rows a step touches are known as soon as
 * its ids are (before the forward): er_emb_mark_touched sets their bitmap bits; er_emb_sweep_untouched
 * then decays every other row (m*=beta1, v*=beta2, var-=lr_t*m/(sqrt(v)+eps): what
 * tf.train.AdamOptimizer._apply_sparse does to rows absent from the IndexedSlices) and clears the
 * bitmap.  Untouched rows are neither read by this step's lookups nor written by its row updates, so
 * the sweep may run on a second stream concurrently with forward/backward; the touched rows are then
 * updated by er_emb_bwd_update(..., ER_OPT_LAZY_ADAM, ...) (identical per-row arithmetic).  Both must
 * complete before the next step's lookups. */
int er_emb_mark_touched(er_emb_group* group, er_stream_t stream);
int er_emb_sweep_untouched(er_emb_group* group, const er_opt_hyper* hyper, er_stream_t stream);
/* ER_OPT_ADAM's dense-decay sweep alone (exposed for benchmarking / roofline measurement). */
int er_adam_decay_sweep(float* var, float* m, float* v, uint32_t* touched_bitmap,
                        int64_t total_rows, int32_t dim, const er_opt_hyper* hyper,
                        er_stream_t stream);
/* the same sweep over rows at a pitch of ld >= dim floats (row records: er_emb_group_set_row_pitch) */
int er_adam_decay_sweep_ld(float* var, float* m, float* v, uint32_t* touched_bitmap, int64_t total_rows, int32_t dim,
                           int64_t ld, const er_opt_hyper* hyper, er_stream_t stream);
/* Bandwidth probe: dst[0:bytes] = src[0:bytes] with the sweep's access pattern (16 B/lane, nontemporal,
 * grid-stride).  Moves exactly 2*bytes of HBM traffic: calibrates rocprofv3 FETCH_SIZE/WRITE_SIZE and
 * gives the achievable copy bandwidth quoted next to the 8 TB/s spec peak. */
int er_stream_copy(const void* src, void* dst, int64_t bytes, er_stream_t stream);

/* n device-to-device copies (src[i] -> dst[i], bytes[i] each; non-overlapping) in one launch: the parts of a
 * device-resident batch that a step copies into its static input buffers (input/features.py load()). */
int er_copy_multi(const void* const* src_host, void* const* dst_host, const int64_t* bytes_host, int n, er_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * K5  FM second-order interaction + wide sum.  Replaces Pack/Sum/Square/Sub/Mul of
 *     FM.__call__ easy_rec/python/layers/fm.py:20-26 (keras variant layers/keras/interaction.py:24-44)
 *     and reduce_sum of model/deepfm.py:62-63.
 * x: [B, F*D] (row stride x_stride); fm_out [B, D] = 0.5*((sum_f e)^2 - sum_f e^2);
 * sum_out [B, D] = sum_f e (saved for backward).
 * -------------------------------------------------------------------------------------------- */
int er_fm_fwd(const float* x, int32_t B, int32_t F, int32_t D, int32_t x_stride, float* fm_out,
              float* sum_out, er_stream_t stream);
/* dx[b,f,:] (+)= g[b,:] * (S[b,:] - x[b,f,:]).  accumulate != 0 adds into dx. */
/* DeepFM's final-DNN input out = [sum over the n_w wide columns | FM over the F fields of width D | the n_d deep columns]
 * ([B, 1 + D + n_d] at leading dimension ld_out; reference model/deepfm.py:60-83) in ONE launch - the arithmetic of
 * er_rowsum_fwd, er_fm_fwd (sum_out [B, D] = the field sums its backward needs) and a copy, unchanged. */
int er_wide_fm_concat(const float* wide, int32_t n_w, int32_t ld_w, const float* fm_x, int32_t F, int32_t D, int32_t ld_x,
                      const float* deep, int32_t n_d, int32_t ld_d, int32_t B, float* out, int32_t ld_out, float* sum_out,
                      er_stream_t stream);
/* er_wide_fm_concat with the BatchNorm finalize + apply of the layer that PRODUCES the deep block (the last layer of
 * DeepFM's deep tower: er_gemm_f32's column statistics -> er_bn_apply_from_stats, reference layers/dnn.py:63-79) in the same
 * launch: y [B, N] = act(BN(x)) is stored as the layer's output AND as columns [1 + D, 1 + D + N) of out; save_mean /
 * save_invstd / the moving statistics as er_bn_apply_from_stats leaves them.  The FM and row-sum workgroups need the
 * embeddings only and run beside the BatchNorm's.  Returns 3 (nothing launched) when the statistics need the merge launch
 * (more than 256 row tiles).  Same arithmetic, same order as the two launches. */
int er_bn_apply_wide_fm(const float* x, const float* col_stats, int32_t chunks, const float* gamma, const float* beta,
                        int32_t B, int32_t N, float eps, float momentum, float* moving_mean, float* moving_var, int act,
                        float* y, float* save_mean, float* save_invstd, const float* wide, int32_t n_w, int32_t ld_w,
                        const float* fm_x, int32_t F, int32_t D, int32_t ld_x, float* out, int32_t ld_out, float* sum_out,
                        er_stream_t stream);
int er_fm_bwd(const float* x, const float* sum_saved, const float* g, int32_t B, int32_t F,
              int32_t D, int32_t x_stride, float* dx, int32_t dx_stride, int accumulate,
              er_stream_t stream);
/* out[b] = sum_j x[b, j], j < n */
int er_rowsum_fwd(const float* x, int32_t B, int32_t n, int32_t x_stride, float* out,
                  er_stream_t stream);
/* The gradient buffers of a model's embedding group outputs, finished in ONE launch (instead of er_rowsum_bwd +
 * er_fm_bwd + er_axpy2d per regularised group + zero fills).  Per group: dout[b, c] = (has_base ? dout[b, c] : 0)
 * [what dgrad GEMMs accumulated there] + the deferred terms in order + lambda * out[b, c] [gradient of the
 * embedding-output L2, layers/input_layer.py:369-375].  Terms: ER_GRAD_TERM_ROWSUM: + g[b * g_ld] for columns
 * [col0, col0 + width) (gradient of reduce_sum over the wide columns, model/deepfm.py:62-63); ER_GRAD_TERM_FM:
 * + g[b * g_ld + d] * (saved[b * dim + d] - out[b, c]), d = (c - col0) % dim (gradient of layers/fm.py:20-26; saved =
 * the field sums er_fm_fwd kept).  groups: HOST array; all pointers DEVICE. */
#define ER_GRAD_TERM_ROWSUM 0
#define ER_GRAD_TERM_FM 1
typedef struct er_grad_term {
  int32_t kind, col0, width, dim;
  const float* g;      /* upstream gradient: [B] (row sum) or [B, dim] (FM), row stride g_ld */
  int32_t g_ld, pad_;
  const float* saved;  /* FM: field sums [B, dim] */
} er_grad_term;
typedef struct er_grad_group {
  float* dout;         /* [batch, width] gradient buffer, row stride ld */
  const float* out;    /* the group's forward output, same layout */
  int32_t ld, batch, width, has_base, n_terms;
  float lambda;
  er_grad_term terms[4];
} er_grad_group;
int er_group_grad_finish(const er_grad_group* groups_host, int n, er_stream_t stream);
/* The fused single-GPU embedding step (round 4).  Both return 0, an error code, or 3 = "these groups need the general path"
 * (er_last_error says why; nothing was launched).
 *   er_emb_front: what er_emb_route + er_emb_catch_up_multi do for groups of dense-mode lookups (one id per output row)
 *     that own their tables - entries built, sorted and run-head-flagged by ONE launch per sort leader (followers of a
 *     shared sort adopt it), whose surplus workgroups also rebuild the closed-form replay's per-launch table; then ONE
 *     launch brings the step's rows current straight from the run heads (no de-duplicated key list).  hyper may be NULL
 *     for groups without lazy dense decay.  skip_one_row != 0: entries into one-row tables (RawFeature projections,
 *     input/input.py:648-673: every valid id is row 0) stay out of the sort - then the step's gradients MUST be reduced
 *     by er_emb_bwd_fused (the other reduce entry points refuse such a sort).
 *   er_emb_bwd_fused: er_group_grad_finish + er_emb_bwd_update_multi in ONE launch: the finished output gradient
 *     (finish[k]: the descriptors er_group_grad_finish takes, one per feature-group buffer the lookups write; each
 *     lookup's `out` must be the dout of one of them) is evaluated while the sorted entries are gathered, a run of equal
 *     keys is reduced by the workgroup that holds its first entry, and the one-row tables are reduced by columns.  The
 *     gradient buffers themselves are left unfinished.  First use uploads a plan: call once outside stream capture. */
#define ER_FRONT_SKIP_ONE_ROW 1    /* `flags` of er_emb_front: the skip_one_row mode above */
#define ER_FRONT_DEFER_CATCH_UP 2 /* no catch-up launch: see er_emb_fwd_lazy */
int er_emb_front(er_emb_group* const* groups, int n, int flags, const er_opt_hyper* hyper, er_stream_t stream);
/* Round 5: the catch-up of TF-exact Adam's lazy dense decay (tf.train.AdamOptimizer._apply_sparse decays every row at
 * every step, builders/optimizer_builder.py:61-66) WITHOUT a launch and without a write.  After
 * er_emb_front(ER_FRONT_DEFER_CATCH_UP) the rows of the step still carry their pending decay-only steps;
 * er_emb_fwd_lazy = er_emb_fwd in which a lookup into one of `groups` (lazy decay, closed-form replay) evaluates the
 * pending steps of every row it reads in registers - from the row's record: var, m, v, last_step, one contiguous access
 * under er_emb_group_set_row_pitch - and sums the caught-up values; nothing is stored.  er_emb_bwd_fused of the same
 * step repeats the evaluation on the same bits right before it applies the row's gradient, so var / m / v / last_step
 * after the step are exactly those of catch-up launch + lookup + update (compat/embedding_ops.py:37-162 for the lookup,
 * compat/adam_s.py:185-213 for the row arithmetic), while a unique row costs one record read in the lookup and one
 * read + one write in the update instead of the 13 scattered accesses of the three-launch form.  Lookups into tables of
 * no listed group are plain er_emb_fwd lookups.  First use uploads a lookup -> group map: call once outside capture. */
int er_emb_fwd_lazy(er_emb_plan* plan, er_emb_group* const* groups, int n, const er_opt_hyper* hyper,
                    float* sumsq_partials, er_stream_t stream);
/* er_emb_front(flags | ER_FRONT_DEFER_CATCH_UP) + er_emb_fwd_lazy as ONE call, and - when the lag-1 replay table is built by
 * the prologue (er_decay_tables_set_prologue_build) - ONE launch: the lookup needs the ids, not the sort, and the sort keeps
 * one workgroup per lookup (~20 us on 39 CUs for DeepFM-Criteo) busy, so the lookup's blocks run behind the sort's
 * workgroups of the same grid.  Same results as the two calls. */
int er_emb_front_fwd(er_emb_group* const* groups, int n, int flags, er_emb_plan* plan, const er_opt_hyper* hyper,
                     float* sumsq_partials, er_stream_t stream);
int er_emb_bwd_fused(er_emb_group* const* groups, int n, const er_grad_group* finish_host, int n_finish, int opt_kind,
                     const er_opt_hyper* hyper, er_stream_t stream);
/* Probe hook (tools/own_probe.py): the next er_emb_bwd_fused* launches leave 16 wall-clock stamps (100 MHz) per workgroup
 * of the embedding backward in `stamps` (device memory, 16 * grid uint64); NULL: off. */
int er_debug_stamps(unsigned long long* stamps);
/* out[b, sum(widths[<p]) + j] = parts[p][b * lds[p] + j]: tf.concat(values, axis=1) of n <= 8 row-major blocks
 * (model/deepfm.py:75-83, model/multi_tower_din.py:96,117) in one launch.  parts / widths / lds: HOST arrays. */
/* ... er_concat_cols_b16: also a bf16 copy of out (row stride ld_bf16 >= the joined width; the columns past it zeroed): the
 * operand of the bf16 contraction that reads the concatenation next (dense_dtype 'bf16'). */
int er_concat_cols_b16(const float* const* parts_host, const int32_t* widths_host, const int32_t* lds_host, int n,
                       int32_t batch, float* out, int32_t out_ld, uint16_t* out_bf16, int32_t ld_bf16, er_stream_t stream);
int er_concat_cols(const float* const* parts_host, const int32_t* widths_host, const int32_t* lds_host, int n,
                   int32_t batch, float* out, int32_t out_ld, er_stream_t stream);
/* y[i] (+)= alpha * x[i]  over a strided 2-D view (used for the embedding L2 gradient
 * lambda*out and for broadcasting d(wide_sum) back to the wide columns) */
int er_axpy2d(const float* x, int32_t x_stride, float alpha, float* y, int32_t y_stride,
              int32_t rows, int32_t cols, int accumulate, er_stream_t stream);
/* y[b, j] (+)= g[b]  (gradient of the row sum) */
int er_rowsum_bwd(const float* g, int32_t B, int32_t n, float* dx, int32_t dx_stride,
                  int accumulate, er_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * K6  DCN-v1 cross network.  Replaces Mul/Sum/Add x L of DCN._cross_net model/dcn.py:32-45:
 *     x_{l+1} = x0 * (x_l . w_l) + b_l + x_l.   All L layers in one launch, one wave per row.
 * x0 [B, d]; w, b [L, d]; out [B, d]; xl_dots [B, L] saves (x_l . w_l) for backward.
 * -------------------------------------------------------------------------------------------- */
int er_cross_v1_fwd(const float* x0, const float* w, const float* b, int32_t B, int32_t d,
                    int32_t L, float* out, float* xl_dots, er_stream_t stream);
/* dx0 [B, d]; dw_partials, db_partials [n_partial, L, d] (reduced by er_colsum);
 * returns n_partial through er_cross_v1_bwd_partials(). Recomputes x_l from x0 and xl_dots. */
int er_cross_v1_bwd_partials(int32_t B);
int er_cross_v1_bwd(const float* x0, const float* w, const float* b, const float* xl_dots,
                    const float* dout, int32_t B, int32_t d, int32_t L, float* dx0,
                    float* dw_partials, float* db_partials, er_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * K7  DCN-v2 cross epilogue.  Replaces BiasAdd/Mul/Add of Cross.call
 *     layers/keras/interaction.py:276-286 after the  W x_l  GEMM:
 *     out = x0 * (u + bias + diag_scale * x) + x          (u = x W, or (x U) V for low rank)
 * -------------------------------------------------------------------------------------------- */
int er_cross_v2_epilogue_fwd(const float* x0, const float* x, const float* u, const float* bias,
                             float diag_scale, int32_t B, int32_t d, float* out,
                             er_stream_t stream);
/* given dout: dx0 (+)= dout * (u + bias + diag*x); du = dout * x0; dx (+)= dout*(1 + diag*x0);
 * dbias via er_colsum(du). */
int er_cross_v2_epilogue_bwd(const float* x0, const float* x, const float* u, const float* bias,
                             float diag_scale, const float* dout, int32_t B, int32_t d,
                             float* dx0, int accumulate_dx0, float* dx, float* du,
                             er_stream_t stream);
/* the same with caller-provided destinations: dx0 (row stride ld_dx0) and dx (row stride ld_dx) are stored or, with their
 * accumulate flag, ADDED to what the buffers hold - an embedding group's gradient buffer, the tensor another gradient of the
 * same x_l went to - so that the sums autograd would run as separate add kernels happen here; dx == NULL: x is x0 (the first
 * cross layer of layers/keras/interaction.py:276-286) and its gradient joins dx0's.  dout: row stride ld_dout (a column
 * block of a wider gradient - tf.concat's backward - is read in place). */
/* The elementwise backward of the TOP layer of a cross stack whose input gradients run through er_gemm_f32_cross /
 * er_gemm_bf16_nt_epi (ER_EPI_CROSS_BWD): du = dout * x0 (fp32, and bf16 with row stride ld_du_bf16 when du_bf16 != NULL),
 * dx0 (+)= dout * (u + bias + diag * x), partial [er_gemm_row_tiles(B)][d] = per 64-row tile column sums of du (NULL: not
 * wanted).  It leaves dout's own term of d/dx to the contraction's epilogue. */
int er_cross_v2_bwd_top(const float* x0, const float* x, const float* u, const float* bias, float diag_scale, const float* dout,
                        int32_t ld_dout, int32_t B, int32_t d, float* dx0, int32_t ld_dx0, int accumulate_dx0, float* du,
                        uint16_t* du_bf16, int32_t ld_du_bf16, float* partial, er_stream_t stream);
int er_cross_v2_epilogue_bwd_acc(const float* x0, const float* x, const float* u, const float* bias, float diag_scale,
                                 const float* dout, int32_t ld_dout, int32_t B, int32_t d, float* dx0, int32_t ld_dx0,
                                 int accumulate_dx0, float* dx, int32_t ld_dx, int accumulate_dx, float* du, er_stream_t stream);

/* ---- K1b: hash-table ("KV") embedding tables: `ev_params` on a feature / model config ---------------------------
 * In the reference an embedding column with ev_params is backed by PAI-TF's get_embedding_variable
 * (compat/feature_column/feature_column_v2.py:3478-3513) or, under embedding parallelism, SOK's DynamicVariable
 * (compat/feature_column/feature_column.py:470-503): rows exist only for ids that have been looked up in training, a new
 * row is drawn from the column's initializer, unseen ids read zeros at evaluation (:3487-3493).  Both are closed
 * dependencies absent from /root/reference; their call-site behaviour is what is reproduced.
 * Here the rows live in an arena of `capacity` rows inside the ordinary table-group storage; er_kv_translate turns an id
 * array into arena rows (open-addressing map in HBM: map_keys int64[map_slots] initialised to -1, map_rows
 * int32[map_slots], map_slots a power of two >= 2 * capacity, *next_row = rows in use) and everything downstream is the
 * dense-table path.  insert != 0 (training): ids not in the map get the next arena row, var[row] = init_mean +
 * init_stddev * z(seed, id, column) from a counter-based generator (a row is a pure function of its key: which arena row
 * a key gets is run-dependent, nothing computed from it is).  insert == 0 (evaluation / prediction): unseen ids map to -1
 * = a zero embedding and no gradient.  ids < 0 ('' / padding) map to -1.  *overflow is set when the arena is full (the
 * ids beyond it read zeros).  er_kv_export compacts the (key, row) pairs for checkpoints / state_dict. */
int er_kv_translate(const int64_t* ids, int64_t n, int64_t* map_keys, int32_t* map_rows, int64_t map_slots,
                    int32_t* next_row, int32_t capacity, float* var, int32_t dim, uint64_t seed, float init_mean,
                    float init_stddev, int insert, int64_t* rows_out, int32_t* overflow, er_stream_t stream);
/* every hash-table lookup of a model in two launches: jobs_dev / blk_start_dev are DEVICE arrays (n_jobs descriptors and
 * n_jobs + 1 workgroup offsets, job j owning ceil(n_j / 256) workgroups), written once when the model is built */
typedef struct {
  const int64_t* ids;
  int64_t n;
  int64_t* map_keys;
  int32_t* map_rows;
  int64_t map_slots;
  int32_t* next_row;
  float* var;
  uint64_t seed;
  int64_t* rows_out;
  int32_t* overflow;
  int32_t capacity, dim;
  float init_mean, init_stddev;
  const int32_t* n_limit; /* NULL, or a device count: only the first min(n, *n_limit) ids are valid (a ragged id list
                             in a fixed-capacity buffer: the number of ids of this step) */
  /* ev_params.filter_freq / steps_to_live (feature_config.proto:27-29; CounterFilter / steps_to_live of the embedding
   * variable at feature_column_v2.py:3497-3512).  All NULL / 0: the plain table.
   *   filter_freq > 1: freq int32[map_slots] (zeroed) counts a key's occurrences in training lookups; the key gets its
   *     row in the launch in which the count reaches filter_freq and reads zeros (row -1, no update) until then; n_keys
   *     int32[1] counts the occupied slots (keys without a row hold one: the overflow flag is raised past map_slots / 2).
   *   version int32[map_slots] + step (device step counter): every training lookup stamps version[slot] = *step, which
   *     the eviction at checkpoint time reads (er_kv_export_all -> er_kv_rebuild). */
  int32_t* freq;
  int32_t* version;
  int32_t* n_keys;
  const int64_t* step;
  int32_t filter_freq;
  int32_t table_ld; /* floats between consecutive arena rows of var (0 = dim): the arena is a column block of the table
                       group's row records (er_emb_group_set_row_pitch) */
} er_kv_job;
/* one job handed over by value from the host (n_limit must be NULL) */
int er_kv_translate_job(const er_kv_job* job, int insert, er_stream_t stream);
/* every occupied slot with its state, rows -1 for keys that have no row yet; *count (zeroed by the caller) = records */
int er_kv_export_all(const int64_t* map_keys, const int32_t* map_rows, const int32_t* map_freq, const int32_t* map_version,
                     int64_t map_slots, int64_t* out_keys, int32_t* out_rows, int32_t* out_freq, int32_t* out_version,
                     int32_t* count, er_stream_t stream);
/* n (key, row, freq, version) records with distinct keys into a CLEARED map (map_keys = -1, map_rows = -1, freq /
 * version = 0): what a checkpoint restore and the steps_to_live eviction (which compacts the arena) end with */
int er_kv_rebuild(const int64_t* keys, const int32_t* rows, const int32_t* freq, const int32_t* version, int64_t n,
                  int64_t* map_keys, int32_t* map_rows, int32_t* map_freq, int32_t* map_version, int64_t map_slots,
                  int32_t* overflow, er_stream_t stream);
int er_kv_translate_multi(const er_kv_job* jobs_dev, const int32_t* blk_start_dev, int n_jobs, int total_blocks, int insert,
                          er_stream_t stream);
/* Embedding-parallel hash tables (the reference: SOK DynamicVariable sharded by id % world,
 * compat/feature_column/feature_column.py:470-503).  Rank r owns the map and arena of the ids with id % world == r.
 * Before the route of a step: er_kv_bucket sorts every job's ids by owner into send[world][block_stride] (job j's ids for
 * an owner at [send_off_j, send_off_j + n_j) of that owner's block, -1 behind them; slot[i] = where id i went), an
 * equal-split all-to-all moves the blocks, the owners run er_kv_translate_multi over what they received, a second
 * all-to-all returns the arena rows, and er_kv_unbucket writes rows_out[i] = arena_row * world + owner (-1: no row) -
 * a dense id that the ordinary embedding-parallel route (owner = id % world, local row = id / world) resolves to that
 * arena row.  counts: int32[n_jobs * world] scratch.  jobs / blk_start: device arrays as in er_kv_translate_multi. */
typedef struct {
  const int64_t* ids;
  int64_t n;
  const int32_t* n_limit; /* NULL or the device count of valid ids (ragged lists) */
  int64_t* rows_out;      /* [n] the virtual dense ids */
  int32_t* slot;          /* [n] position in the send buffer, -1 for padding */
  int64_t send_off;       /* this job's region inside an owner's block */
} er_kv_route_job;
int er_kv_bucket(const er_kv_route_job* jobs_dev, const int32_t* blk_start_dev, int n_jobs, int total_blocks, int world,
                 int64_t block_stride, int64_t* send, int32_t* counts, er_stream_t stream);
int er_kv_unbucket(const er_kv_route_job* jobs_dev, const int32_t* blk_start_dev, int n_jobs, int total_blocks, int world,
                   int64_t block_stride, const int64_t* back, er_stream_t stream);
int er_kv_export(const int64_t* map_keys, const int32_t* map_rows, int64_t map_slots, int64_t* out_keys,
                 int32_t* out_rows, int32_t* count, er_stream_t stream);
/* ---- K9b: CIN, xDeepFM's compressed interaction network (reference layers/keras/interaction.py:370-409) ----
 *   x_{k+1}[b, n, d] = relu(sum_{h, m} W_k[n, h, m] * x_k[b, h, d] * x_0[b, m, d] + bias_k[n]);
 *   output = concat over the layers of sum_d x_{k+1}[b, :, d].
 * The reference materialises [B, H_k+1, H_k, H_0, D].  Here: er_cin_outer_fwd writes the outer product
 * z[(b, d), h * H0 + m] = x_k[b, h, d] * x_0[b, m, d] ([B * D, H * H0], k-contiguous); er_gemm_f32(ER_GEMM_NT, z, W_k
 * viewed as [H_k+1, H * H0]) contracts it into c [B * D, H_k+1]; er_cin_act_pool_fwd turns c into the feature map
 * relu(c + bias) in place and writes its sum over d into columns [col0, col0 + N) of the layer's output block.  c IS
 * x_{k+1} in [B, D, H_k+1] layout: x_k is addressed through (stride_b, stride_h, stride_d) - x_0 is [B, H_0, D].
 * Backward: er_cin_act_pool_bwd -> dc; dW_k = dc^T . z (ER_GEMM_TN), dz = dc . W_k (ER_GEMM_NN); er_cin_outer_bwd:
 * dx_k[b, h, d] (=|+=) sum_m dz * x_0, dx_0[b, m, d] += sum_h dz * x_k (the first layer has x_k == x_0: dxi == dx0, add). */
int er_cin_outer_fwd(const float* xi, int64_t xi_stride_b, int32_t xi_stride_h, int32_t xi_stride_d, int32_t H,
                     const float* x0, int32_t H0, int32_t D, int64_t B, float* z, er_stream_t stream);
int er_cin_act_pool_fwd(float* c, const float* bias, int64_t B, int32_t D, int32_t N, float* pooled, int32_t pooled_ld,
                        int32_t col0, er_stream_t stream);
int er_cin_act_pool_bwd(const float* fm, const float* dpooled, int32_t dpooled_ld, int32_t col0, const float* dnext,
                        int64_t B, int32_t D, int32_t N, float* dc, er_stream_t stream);
int er_cin_outer_bwd(const float* dz, const float* xi, int64_t xi_stride_b, int32_t xi_stride_h, int32_t xi_stride_d,
                     int32_t H, const float* x0, int32_t H0, int32_t D, int64_t B, float* dxi, int add_xi, float* dx0,
                     er_stream_t stream);
/* --------------------------------------------------------------------------------------------
 * K8  DIN target attention.  Replaces Tile/ConcatV2 (input of the attention MLP) and
 *     SequenceMask/Select/Softmax/BatchMatMul (pooling) of MultiTowerDIN.din
 *     model/multi_tower_din.py:62-97 (same math: layers/sequence_feature_layer.py:123-189,
 *     layers/keras/din.py:27-67).
 * din_concat: a[b,t,:] = [q_b, h_bt, q_b - h_bt, q_b * h_bt]           ([B, L, 4E])
 * din_pool  : scores [B, L] -> mask t >= len[b] with -2^32+1 -> softmax over L -> p @ hist
 * -------------------------------------------------------------------------------------------- */
int er_din_concat_fwd(const float* query, const float* hist, int32_t B, int32_t L, int32_t E,
                      float* out, er_stream_t stream);
/* dquery [B,E] (+)=, dhist [B,L,E] (+)= from dout [B,L,4E] */
int er_din_concat_bwd(const float* query, const float* hist, const float* dout, int32_t B,
                      int32_t L, int32_t E, float* dquery, int acc_q, float* dhist, int acc_h,
                      er_stream_t stream);
/* probs_out [B, L] saved for backward; out [B, E] */
int er_din_pool_fwd(const float* scores, const float* hist, const int32_t* seq_len, int32_t B,
                    int32_t L, int32_t E, float scale, float* probs_out, float* out,
                    er_stream_t stream);
int er_din_pool_bwd(const float* probs, const float* hist, const int32_t* seq_len,
                    const float* dout, int32_t B, int32_t L, int32_t E, float scale,
                    float* dscores, float* dhist, int acc_h, er_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * K9  MLP layer pieces around the GEMM.  Replaces BiasAdd / FusedBatchNorm(train) / Relu of
 *     DNN.__call__ layers/dnn.py:57-79 (keras MLP layers/keras/blocks.py:84-110) and Dice
 *     layers/keras/activation.py:47-70.
 * er_bn_act_fwd: y = act(gamma * (x + bias - mean) * rsqrt(var + eps) + beta), batch statistics
 *   over the B rows (biased variance); updates moving stats with `momentum` when not NULL.
 *   act: 0 = identity, 1 = relu.  bias/gamma/beta may be NULL (treated as 0/1/0).
 *   use_bn == 0: y = act(x + bias).  save_mean/save_invstd: [N] (needed by backward).
 *   use_bn == ER_BN_FROZEN: tf.layers.batch_normalization(training=False) inside a training graph - what the
 *   experts of the reference's MMoE / DBMTL models run (model/mmoe.py:37-47 and model/dbmtl.py:66-70 build
 *   layers/mmoe.py:14-20 MMOE without is_training, default False): the MOVING statistics normalise and are not
 *   updated; save_mean / save_invstd receive them for er_bn_act_bwd*, which with the same flag drops the two
 *   batch-statistics terms of dx (dx = gamma * invstd * g) and gives the bias its gradient (column sum of dx).
 * -------------------------------------------------------------------------------------------- */
enum { ER_ACT_NONE = 0, ER_ACT_RELU = 1 };
enum { ER_BN_NONE = 0, ER_BN_BATCH = 1, ER_BN_FROZEN = 2 };
int er_bn_act_fwd(const float* x, const float* bias, const float* gamma, const float* beta,
                  int32_t B, int32_t N, int use_bn, float eps, float momentum,
                  float* moving_mean, float* moving_var, int act, float* y, float* save_mean,
                  float* save_invstd, er_stream_t stream);
/* The same normalisation from ready-made column statistics: col_stats = `chunks` Welford triples per column
 * ([chunks][N][3]: count, mean, M2), e.g. written by er_gemm_*'s epilogue (chunks = er_gemm_row_tiles(B)).
 * One launch: every workgroup merges the partials of its 64 columns, then normalises its tile. */
int er_bn_apply_from_stats(const float* x, const float* bias, const float* col_stats, int32_t chunks,
                           const float* gamma, const float* beta, int32_t B, int32_t N, float eps,
                           float momentum, float* moving_mean, float* moving_var, int act, float* y,
                           float* save_mean, float* save_invstd, er_stream_t stream);
/* ... that also writes y_bf16 [B][ld_bf16] (NULL: no), the bf16 copy the next bf16 contraction reads (dense_dtype 'bf16':
 * the producer writes its consumer's operand - no cast launch); er_bn_act_bwd_ld_b16 / er_bn_act_bwd_from_partials_ld_b16
 * likewise leave dx_bf16 for the input-gradient contraction. */
int er_bn_apply_from_stats_b16(const float* x, const float* bias, const float* col_stats, int32_t chunks,
                               const float* gamma, const float* beta, int32_t B, int32_t N, float eps,
                               float momentum, float* moving_mean, float* moving_var, int act, float* y,
                               float* save_mean, float* save_invstd, uint16_t* y_bf16, int32_t ld_bf16, er_stream_t stream);
/* The FINALIZE half of er_bn_apply_from_stats alone: the column statistics are merged (in the same order, to the same bits)
 * into save_mean / save_invstd and the moving statistics; nothing is normalised.  For a TALL layer (DIN's attention MLP over
 * B x L rows, reference model/multi_tower_din.py:62-80 -> layers/dnn.py:57-79) whose apply runs inside the next
 * contraction's staging: er_gemm_f32_bn_a. */
int er_bn_finalize_from_stats(const float* col_stats, int32_t chunks, int32_t B, int32_t N, float eps, float momentum,
                              float* moving_mean, float* moving_var, float* save_mean, float* save_invstd,
                              er_stream_t stream);
int er_bn_act_bwd_ld_b16(const float* x, const float* bias, const float* gamma, const float* y,
                         const float* save_mean, const float* save_invstd, const float* dy, int32_t dy_ld,
                         int32_t B, int32_t N, int use_bn, int act, float* dx, float* dbias, float* dgamma,
                         float* dbeta, int accumulate, uint16_t* dx_bf16, int32_t ld_bf16, er_stream_t stream);
int er_bn_act_bwd_from_partials_ld_b16(const float* x, const float* bias, const float* gamma, const float* y,
                                       const float* save_mean, const float* save_invstd, const float* dy, int32_t dy_ld,
                                       int32_t B, int32_t N, int use_bn, int act, const float* partial, int32_t chunks,
                                       float* dx, float* dbias, float* dgamma, float* dbeta, int accumulate,
                                       uint16_t* dx_bf16, int32_t ld_bf16, er_stream_t stream);
/* dx [B,N]; dbias/dgamma/dbeta [N] (NULL to skip): overwritten, or += when accumulate != 0 (they then
 * point into the flat gradient buffer).  y is the forward output (relu mask); x the forward input. */
int er_bn_act_bwd(const float* x, const float* bias, const float* gamma, const float* y,
                  const float* save_mean, const float* save_invstd, const float* dy, int32_t B,
                  int32_t N, int use_bn, int act, float* dx, float* dbias, float* dgamma,
                  float* dbeta, int accumulate, er_stream_t stream);
/* er_bn_act_bwd with dy as a column block of a wider buffer (row stride dy_ld >= N): the gradient slice a concat's
 * backward hands over, read in place. */
int er_bn_act_bwd_ld(const float* x, const float* bias, const float* gamma, const float* y,
                     const float* save_mean, const float* save_invstd, const float* dy, int32_t dy_ld,
                     int32_t B, int32_t N, int use_bn, int act, float* dx, float* dbias, float* dgamma,
                     float* dbeta, int accumulate, er_stream_t stream);
/* The second half of er_bn_act_bwd, fed with column-sum partials [chunks][N][2] that the dgrad GEMM's epilogue
 * already produced (er_gemm_f32_bn_bwd). */
int er_bn_act_bwd_from_partials(const float* x, const float* bias, const float* gamma, const float* y,
                                const float* save_mean, const float* save_invstd, const float* dy,
                                int32_t B, int32_t N, int use_bn, int act, const float* partial,
                                int32_t chunks, float* dx, float* dbias, float* dgamma, float* dbeta,
                                int accumulate, er_stream_t stream);
/* ... with dy a column block of a wider gradient (dy_ld floats between its rows: er_gemm_f32_bn_bwd_cols' output) */
int er_bn_act_bwd_from_partials_ld(const float* x, const float* bias, const float* gamma, const float* y,
                                   const float* save_mean, const float* save_invstd, const float* dy, int32_t dy_ld,
                                   int32_t B, int32_t N, int use_bn, int act, const float* partial, int32_t chunks,
                                   float* dx, float* dbias, float* dgamma, float* dbeta, int accumulate,
                                   er_stream_t stream);
/* out[j] = sum_i x[i, j]  (bias gradients, partial reductions); er_colsum_acc: out[j] += ... when accumulate
 * (straight into the variable's slice of the flat gradient buffer) */
int er_colsum_acc(const float* x, int32_t rows, int32_t cols, int32_t x_stride, float* out, int accumulate,
                  er_stream_t stream);
int er_colsum(const float* x, int32_t rows, int32_t cols, int32_t x_stride, float* out,
              er_stream_t stream);
/* Dice: p = sigmoid(BN_noaffine(x, eps)); y = alpha*(1-p)*x + p*x */
int er_dice_fwd(const float* x, const float* alpha, int32_t B, int32_t N, float eps,
                float momentum, float* moving_mean, float* moving_var, float* y,
                float* save_mean, float* save_invstd, er_stream_t stream);
int er_dice_bwd(const float* x, const float* alpha, const float* save_mean,
                const float* save_invstd, const float* dy, int32_t B, int32_t N, float* dx,
                float* dalpha, er_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * K10 loss.  Replaces SigmoidCrossEntropyWithLogits + Mean of
 *     loss_builder.build builders/loss_builder.py:35-39 (tf.losses.sigmoid_cross_entropy,
 *     reduction SUM_BY_NONZERO_WEIGHTS) and its gradient.
 * loss_out[0] = sum_i w_i * ce(z_i, y_i) / #(w_i != 0);  dlogits[i] = w_i*(sigmoid(z_i)-y_i)/#nz
 * weights: NULL = all ones.  loss_scale multiplies both (task weight in multi-task models).
 * -------------------------------------------------------------------------------------------- */
int er_sigmoid_ce_fwd_bwd(const float* logits, const float* labels, const float* weights,
                          int32_t B, float loss_scale, float* loss_out, float* dlogits,
                          float* probs_out, er_stream_t stream);
/* ... of several heads (the towers of model/multi_task_model.py:228-275, one tf.losses call each) in one launch. */
typedef struct er_ce_head {
  const float* logits; const float* labels; const float* weights; /* weights may be NULL */
  int32_t B; float loss_scale;
  float* loss_out; float* dlogits; float* probs_out;              /* any of them may be NULL */
} er_ce_head;
int er_sigmoid_ce_multi(const er_ce_head* heads_host, int n, er_stream_t stream);
/* reg_out[0] = reg_emb[0] + reg_dense[0]; total_out[0] = reg_out[0] + sum_i losses[i][0]; report[i][0] = losses[i][0]
 * (report may be NULL).  losses_host / report_host: HOST arrays of n <= 8 DEVICE pointers.  The add_n over the
 * loss dict and REGULARIZATION_LOSSES of model/easy_rec_estimator.py:166-184 in one launch. */
int er_total_loss(const float* reg_emb, const float* reg_dense, const float* const* losses_host,
                  float* const* report_host, int32_t n, float* reg_out, float* total_out, er_stream_t stream);
/* The scalar tail of the loss in ONE launch: reg_out[0] = emb_scale * sum(emb_partials[0..n_partials)) [embedding-output
 * L2 from er_emb_fwd's per-block sums of squares; n_partials may be 0] + sum(dense_partials[0..n_dense)) [the kernels'
 * L2, sum_i 0.5 * coef[i] * w[i]^2, as per-256-weight block sums: er_l2_partials once, then kept current by
 * er_dense_opt_step_l2; n_dense may be 0]; total_out[0] = reg_out[0] + sum_i losses[i][0]; report[i][0] = losses[i][0]
 * (report may be NULL).  losses / report: HOST arrays of n_losses <= 8 DEVICE pointers.
 * = er_reduce_sum + er_l2_loss + er_total_loss. */
int er_reg_total_loss(const float* emb_partials, int32_t n_partials, float emb_scale, const float* dense_partials,
                      int32_t n_dense, const float* const* losses_host, float* const* report_host, int32_t n_losses,
                      float* reg_out, float* total_out, er_stream_t stream);
/* The binary head of a rank model in ONE launch (round 4).  Replaces the TF-op chain of the `output` projection and its
 * loss - tf.layers.dense(units = 1) (model/deepfm.py:84-88, model/dcn.py:66, model/multi_tower.py, model/dlrm.py) ->
 * tf.losses.sigmoid_cross_entropy (builders/loss_builder.py:35-39; no sample weights: the mean over the batch) - AND the
 * gradient of both: logits[r] = x[r, :] . w + b[0]; probs = sigmoid(logits); dlogits[r] = loss_scale * (probs - labels) / B;
 * dx[r, :] = dlogits[r] * w (dense [B, K]); loss = loss_scale * sum(loss_partials) / B over the ceil(B / 64) row tiles;
 * wb_partials[tile][0..K) = sum over the tile's rows of x[r][c] * dlogits[r] (dW), [tile][K] = sum of dlogits (db) - summed
 * into the gradient buffers by er_loss_tail's column-sum jobs.  When x is the activation output of a dense + BatchNorm +
 * ReLU layer (src_z = that layer's pre-normalisation values incl. bias, src_mean / src_invstd its batch statistics, NULL
 * for a layer without BatchNorm; src_act ER_ACT_*), bn_partials [tile][K][2] receives that layer's BatchNorm-backward
 * column sums (sum g, sum g * xhat; g = dx masked by the ReLU) exactly as er_gemm_f32_bn_bwd leaves them.  K % 4 == 0,
 * K <= 256; x, w, dx (and src_z) 16-byte aligned; probs / dlogits / b may be NULL.  Deterministic (fixed reduction trees). */
int er_head_sigmoid_ce(const float* x, int32_t ldx, const float* w, const float* b, const float* labels, int32_t B, int32_t K,
                       float loss_scale, float* logits, float* probs, float* dlogits, float* dx, float* loss_partials,
                       float* wb_partials, const float* src_z, int32_t src_ld, const float* src_mean, const float* src_invstd,
                       int32_t src_act, float* bn_partials, er_stream_t stream);
/* er_reg_total_loss for steps whose head ran as er_head_sigmoid_ce: loss i may be given as loss_parts[i] > 0 partial sums
 * (losses[i] points at them; its value loss_scales[i] * their sum / loss_divs[i] is also stored to loss_values[i][0] when that is
 * non-NULL), and up to 4 column-sum jobs dst[j] += sum_p partial[p * ld + j] (j < n_cols) ride along (the head's dW / db, straight
 * into the flat gradient buffer that the estimator's add_n of IndexedSlices-free dense gradients lives in).  HOST arrays. */
typedef struct er_tail_job {
  const float* partial;
  float* dst;
  int32_t n_parts, n_cols;
  int32_t ld; /* floats between consecutive partial rows (>= n_cols) */
} er_tail_job;
int er_loss_tail(const float* emb_partials, int32_t n_partials, float emb_scale, const float* dense_partials, int32_t n_dense,
                 const float* const* losses_host, float* const* report_host, const int32_t* loss_parts, const float* loss_scales,
                 const float* loss_divs, float* const* loss_values_host, int32_t n_losses, const er_tail_job* jobs_host, int32_t n_jobs, float* reg_out,
                 float* total_out, er_stream_t stream);
/* partials[b] = sum over weights [256 b, 256 b + 256) of 0.5 * coef * w^2 (ceil(n / 256) floats); er_dense_opt_step_l2 =
 * er_dense_opt_step that also leaves these sums of the UPDATED weights (l2_partials may be NULL): the next step's
 * kernel-L2 term costs no pass over the weights. */
int er_l2_partials(const float* w, const float* coef, int64_t n, float* partials, er_stream_t stream);
int er_dense_opt_step_l2(float* w, float* m, float* v, const float* grad, const float* l2coef, int64_t n,
                         int opt_kind, const er_opt_hyper* hyper, float* l2_partials, er_stream_t stream);
/* er_loss_tail's and er_dense_opt_step_l2's arguments as records (HOST structs, the arrays inside HOST arrays as there):
 * what er_emb_bwd_fused_tail (below) takes to run the two inside the step's tail launches. */
typedef struct er_loss_tail_job {
  const float* emb_partials; int32_t n_partials; float emb_scale;
  const float* dense_partials; int32_t n_dense;
  const float* const* losses; float* const* report; const int32_t* loss_parts; const float* loss_scales;
  const float* loss_divs; float* const* loss_values; int32_t n_losses;
  const er_tail_job* jobs; int32_t n_jobs;
  float* reg_out; float* total_out;
} er_loss_tail_job;
typedef struct er_dense_opt_job {
  float* w; float* m; float* v; float* grad; const float* l2coef; int64_t n; int32_t opt_kind; const er_opt_hyper* hyper;
  float* l2_partials;
} er_dense_opt_job;
/* er_hyper_select that also zeroes zero_floats floats at `zero` (the dense variables' flat gradient buffer): the
 * step's prologue as one launch. */
int er_step_prologue(const float* table, int64_t* counter, int32_t n_slots, int32_t floats_per_slot, float* out,
                     float* history, int64_t history_capacity, int32_t history_index, float* zero, int64_t zero_floats,
                     er_stream_t stream);
/* the same + the closed-form replay's per-step table (er_decay_tables_create; NULL: plain er_step_prologue) */
int er_step_prologue_decay(const float* table, int64_t* counter, int32_t n_slots, int32_t floats_per_slot, float* out,
                           float* history, int64_t history_capacity, int32_t history_index, float* zero,
                           int64_t zero_floats, er_decay_tables* decay_tables, er_stream_t stream);
/* the same + string_to_hash_bucket_fast of the batch's id strings (er_hash_bucket_fast's arguments; n_strings == 0: none) as
 * further workgroups of the SAME launch: the hash depends on nothing the prologue writes, and a launch of its own costs
 * what a kernel boundary costs (feature_column_v2.py:3915-3921 issues it per column before the lookups). */
int er_step_prologue_hash(const float* table, int64_t* counter, int32_t n_slots, int32_t floats_per_slot, float* out,
                          float* history, int64_t history_capacity, int32_t history_index, float* zero, int64_t zero_floats,
                          er_decay_tables* decay_tables, const uint8_t* str_bytes, const int64_t* str_offsets, int64_t n_strings,
                          int64_t n_per_col, const uint64_t* num_buckets, int drop_empty, int64_t* ids_out, er_stream_t stream);
/* out[0] = scale * sum of all n partials (deterministic single-block tree) */
int er_reduce_sum(const float* partials, int32_t n, float scale, float* out, int accumulate,
                  er_stream_t stream);
/* out[0] (+)= sum_i 0.5 * coef[i] * w[i]^2   (tf.nn.l2_loss terms, compat/regularizers.py:106) */
int er_l2_loss(const float* w, const float* coef, int64_t n, float* out, int accumulate,
               er_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * K11 MMoE mixing.  Replaces Pack/Softmax/Mul/Sum of MMOE.__call__ layers/mmoe.py:73-82:
 *     out[t][b,:] = sum_e softmax_e(gate_logits[t][b,:]) * experts[e][b,:]
 * experts [E, B, H] ; gate_logits [T, B, E]; out [T, B, H]; gates_out [T, B, E] (softmax saved).
 * -------------------------------------------------------------------------------------------- */
int er_mmoe_mix_fwd(const float* experts, const float* gate_logits, int32_t T, int32_t E,
                    int32_t B, int32_t H, float* gates_out, float* out, er_stream_t stream);
int er_mmoe_mix_bwd(const float* experts, const float* gates, const float* dout, int32_t T,
                    int32_t E, int32_t B, int32_t H, float* dexperts, float* dgate_logits,
                    er_stream_t stream);


/* --------------------------------------------------------------------------------------------
 * K13 dense contractions on the matrix cores.  Replaces the MatMul of tf.layers.dense
 *     (layers/dnn.py:57-62, model/deepfm.py:84-104, layers/mmoe.py:53-58, keras Dense
 *     layers/keras/blocks.py:84-89, Cross W.x layers/keras/interaction.py:249-286) and its gradients.
 *   C[M,N] (+)= op(A) . op(B) (+ bias[N]);  fp32 in HBM, row-major, leading dimensions in floats.
 *     ER_GEMM_NN: A[M,K] B[K,N]   forward   y  = x . W
 *     ER_GEMM_NT: A[M,K] B[N,K]   backward  dx = dy . W^T
 *     ER_GEMM_TN: A[K,M] B[K,N]   backward  dW = x^T . dy   (split-K + deterministic reduce)
 *   er_gemm_f32 : v_mfma_f32_32x32x2_f32, exact fp32 (an fmaf chain in k order).
 *   er_gemm_bf16: operands rounded to bf16 (RNE) while staged into LDS, v_mfma_f32_32x32x16_bf16,
 *                 fp32 accumulate (BASELINE config 3: bf16 dense, fp32 embeddings and master weights).
 *   accumulate != 0: C += result (gradients accumulate straight into the flat gradient buffer).
 *   col_stats != NULL: the epilogue also writes, per 64-row tile t (t < er_gemm_row_tiles(M)) and output
 *     column j, the Welford triple (count, mean, M2) of C[64t : 64t+64, j] to col_stats[(t*N + j)*3 ..]:
 *     the batch statistics of a following BatchNorm without another pass over C (er_bn_apply_from_stats
 *     consumes them: FusedBatchNorm / moments of layers/dnn.py:63-69).
 *   er_gemm_reserve pre-sizes the split-K workspace (floats) before hipGraph capture.
 * -------------------------------------------------------------------------------------------- */
enum { ER_GEMM_NN = 0, ER_GEMM_NT = 1, ER_GEMM_TN = 2 };
int er_gemm_reserve(int64_t floats);
int er_gemm_row_tiles(int32_t M);
/* dgrad GEMM with the BatchNorm-backward column sums in its epilogue.  C = op(A).op(B) is the gradient dy of the
 * activations y = act(BN(z + z_bias)) of the layer below (reference layers/dnn.py:57-79, its tf.layers.dense ->
 * batch_normalization -> relu chain); per 64-row tile of C the epilogue writes partial[tile][col][0..1] =
 * (sum g, sum g * xhat), g = dy masked by the activation - what er_bn_act_bwd otherwise computes in a pass of its
 * own over dy, y and z.  er_bn_act_bwd_from_partials(partial, chunks = er_gemm_row_tiles(M)) then finishes the
 * BatchNorm backward in one launch.  No split-K, bias or accumulate. */
int er_gemm_f32_bn_bwd(int layout, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B,
                       int32_t ldb, float* C, int32_t ldc, const float* z, const float* z_bias, const float* y,
                       const float* save_mean, const float* save_invstd, int32_t ld_zy, int use_bn, int act,
                       float* partial, er_stream_t stream);
/* ... when the layer below produced only a COLUMN BLOCK of what C is the gradient of: output columns [col0, col0 + n_src)
 * of C are the gradient of that layer's n_src activations (z, y, statistics: its own, n_src wide; partial [tiles][n_src]
 * [2]), the other columns carry no epilogue.  DeepFM's final DNN reads [reduce_sum(wide) | FM | deep] (reference
 * model/deepfm.py:75-83): the input gradient of its first layer holds the deep tower's 64 columns at col0 = 1 + D, and
 * the tower's last BatchNorm backward needs no column-sum pass of its own (one launch less per step). */
int er_gemm_f32_bn_bwd_cols(int layout, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B,
                            int32_t ldb, float* C, int32_t ldc, const float* z, const float* z_bias, const float* y,
                            const float* save_mean, const float* save_invstd, int32_t ld_zy, int use_bn, int act,
                            float* partial, int32_t col0, int32_t n_src, er_stream_t stream);
/* Grouped launch: n independent fp32 problems of ONE layout in one grid (+ one grid for their split-K reduces).
 * The use: the weight gradients dW_l = x_l^T . dz_l of every dense layer of a step (reference: the MatMul
 * gradients TF schedules for layers/dnn.py:57-62, one per layer) - each a small M x N with K = batch - queued
 * during the backward pass and contracted together once it is over.  Same kernel body as er_gemm_f32; the
 * k-split of a problem (hence its summation order) is a fixed function of the group's shapes; no column
 * statistics. */
typedef struct er_gemm_problem {
  int32_t M, N, K;
  const float* A; int32_t lda;
  const float* B; int32_t ldb;
  float* C; int32_t ldc;
  const float* bias;
  int32_t accumulate;
  /* col_stats != NULL: per-row-tile Welford statistics of this problem's output columns, as er_gemm_f32's col_stats
   * ([er_gemm_row_tiles(M)][N][3]; the problem is then not k-split and must not accumulate): the same-depth layers of
   * parallel stacks (MMoE's experts, the task towers: layers/mmoe.py:62-83, model/multi_task_model.py:33-100) run as ONE
   * launch and still feed their BatchNorms */
  float* col_stats;
  /* bn_partial != NULL: the epilogue of er_gemm_f32_bn_bwd for this problem - the BatchNorm-backward column sums of the
   * layer that produced the problem's OUTPUT position (z, y, statistics, leading dimension of z / y, partial
   * [er_gemm_row_tiles(M)][N][2]); not k-split, no accumulate */
  const float* bn_z; const float* bn_zbias; const float* bn_y; const float* bn_mean; const float* bn_invstd;
  int32_t bn_ld, bn_use_bn, bn_act;
  float* bn_partial;
  /* fz_y != NULL (layout NN, fp32; no col_stats / bn_partial / accumulate): the problem's output ALSO goes through
   * bias + BatchNorm on the MOVING statistics + activation in the epilogue - the experts of the reference's MMoE / DBMTL
   * (layers/mmoe.py:62-83 builds them with batch_normalization(training=False) inside the training graph; layers/dnn.py:57-79):
   * C keeps z = A . B, fz_y [M][ldc] receives y = act((((z + fz_bias) - fz_mean) * (1 / sqrt(fz_var + fz_eps))) * fz_gamma +
   * fz_beta) - er_bn_act_fwd's frozen arithmetic, operation by operation - and fz_save [2][N] the mean and 1 / sqrt(var + eps)
   * the backward reads.  fz_bias / fz_gamma / fz_beta may be NULL (0 / 1 / 0). */
  const float* fz_bias; const float* fz_gamma; const float* fz_beta; const float* fz_mean; const float* fz_var;
  float fz_eps; int32_t fz_act;
  float* fz_y; float* fz_save;
  /* bn_dz_out != 0 (with bn_partial; the producing layer normalises with the MOVING statistics): the launch stores
   * dz = bn_gamma * bn_invstd * g - g the accumulator masked by the layer's activation - instead of the accumulator: the
   * layer's whole elementwise backward (er_bn_act_bwd's frozen form) in the epilogue of the contraction that produces its dy;
   * its parameter gradients follow from bn_partial (er_bn_bwd_multi with dx == NULL) */
  const float* bn_gamma; int32_t bn_dz_out;
} er_gemm_problem;
int er_gemm_grouped_f32(int layout, const er_gemm_problem* problems_host, int n, er_stream_t stream);
/* ... with the operands rounded to bf16 while staged (er_gemm_bf16's arithmetic: v_mfma_f32_32x32x16_bf16, fp32
 * accumulation; the weight gradients of a bf16 step in one launch).  No A transform, no BatchNorm-backward epilogue. */
int er_gemm_grouped_bf16(int layout, const er_gemm_problem* problems_host, int n, er_stream_t stream);
/* How a grouped launch's grid is laid over the chip (host code, no device needed; the launch itself uses exactly these).
 * A workgroup runs on XCD (block % 8) and every XCD has its own L2; all tiles of one k-split read the same rows of both
 * operands.  er_gemm_grouped_layout: from the problems' tile and k-split counts, the XCD region - problems with
 * by_xcd[p] != 0: with >= 8 splits the first xsplits[p] = 8 * (splits[p] / 8) of them, dealt to the XCDs whole (xper[p] = 1:
 * XCD x takes splits x, x + 8, ...); with 4 or 2 splits every split on xper[p] = 2 or 4 neighbouring XCDs, its tiles dealt
 * round robin among them (a slot past the last tile idles: split -1); per-XCD offsets xstart[0..n] - and the legacy region
 * behind it (all other splits, offsets start[0..n]); by_xcd NULL: legacy region only (the round-4 grid).  Returns the grid
 * size (negative: error).  er_gemm_grouped_coords: what workgroup `block` of that grid computes - problem, tile (plain 1:
 * the tile itself; 0: a slot of the XCD-aware tile order inside the problem) and k-split. */
int er_gemm_grouped_layout(const int32_t* tiles, const int32_t* splits, int n, const int32_t* by_xcd, int32_t* start,
                           int32_t* xstart, int32_t* xsplits, int32_t* xper);
int er_gemm_grouped_coords(const int32_t* tiles, const int32_t* start, const int32_t* xstart, const int32_t* xsplits,
                           const int32_t* xper, int n, int32_t block, int32_t* problem, int32_t* tile, int32_t* split,
                           int32_t* plain);
/* The step's TAIL as two launches instead of four: er_emb_bwd_fused (above) with the weight gradients of the step's dense
 * layers - `wgrads`: n_wgrads <= 16 plain ER_GEMM_TN problems dW_l (+)= x_l^T . dz_l, as er_gemm_grouped_f32 takes them
 * (reference: the gradients tf.gradients builds for tf.layers.dense, layers/dnn.py:57-62, which the optimizer consumes
 * beside the IndexedSlices of the embedding variables, compat/optimizers.py:285-345) - in the SAME grid: the
 * contraction's workgroups first (its XCD region keeps block % 8), the embedding gradient finish + segmented reduce +
 * row update behind them; then the cross-tile fix next to the split-K reduce.  The two halves depend only on the last
 * input-gradient GEMM, not on each other, and are bound by different units (matrix cores / LDS against the latency of
 * random row records).  wgrad_blocks: the number of workgroups the contraction's k-splits aim at; 0 = the stand-alone
 * launch's (512), and then the results are bit-identical to er_gemm_grouped_f32(ER_GEMM_TN, wgrads) followed by
 * er_emb_bwd_fused: same bodies, same k-splits, same reduce order.  Fewer, longer contraction workgroups suit the shared
 * grid better (they hold CU workgroup slots next to the row update's tiles for the whole launch): another split count
 * sums the batch in another - equally fixed, launch-to-launch deterministic - order. */
/* The embedding-parallel REQUESTER's tail of the compute segment as two launches (reference
 * compat/feature_column/feature_column.py:248-357 backward + compat/optimizers.py:285-345: the IndexedSlices of the local
 * lookups are de-duplicated before hvd.alltoall returns them to the owners, next to the dense gradients' all-reduce): the
 * local gradient reductions of up to 4 table groups - modes[i] 1: a sharded group's per-(owner, id) sums into outs[i]
 * (er_emb_bwd_reduce_routed; this step's er_emb_route ran), 2: a replicated group's dense row sums [rows][ld] with the count
 * in column dim (er_emb_bwd_reduce_dense) - in ONE tile grid, with the step's weight gradients (wgrads: n_wgrads <= 16 plain
 * ER_GEMM_TN problems, as er_emb_bwd_fused_tail takes them; 0: none) contracted in the same grid and the loss tail
 * (er_loss_tail's job, may be NULL) as one more workgroup; then the cross-tile fix beside the split-K reduce.  Same bodies
 * as the launches apart: bit-identical results at the same wgrad_blocks (0: er_gemm_grouped_f32's own k-splits). */
int er_emb_reduce_local_tail(er_emb_group* const* groups, const int32_t* modes_host, float* const* outs_host,
                             const int32_t* ld_host, int n, const er_gemm_problem* wgrads_host, int n_wgrads,
                             int32_t wgrad_blocks, const er_loss_tail_job* loss_tail, er_stream_t stream);
int er_emb_bwd_fused_wgrad(er_emb_group* const* groups, int n, const er_grad_group* finish_host, int n_finish, int opt_kind,
                           const er_opt_hyper* hyper, const er_gemm_problem* wgrads_host, int n_wgrads,
                           int32_t wgrad_blocks, er_stream_t stream);
/* ... and with the rest of the step's tail riding in the same two launches (either job may be NULL):
 *   loss_tail: er_loss_tail (the add_n over the loss dict + REGULARIZATION_LOSSES, model/easy_rec_estimator.py:166-184,
 *     and the head's dW / db column sums) as ONE more workgroup of the first launch - it reads what the forward pass left
 *     and is read by nothing before the optimizer, so it needs no launch between them;
 *   dense_opt: er_dense_opt_step_l2 over the flat dense variables (builders/optimizer_builder.py:33-97 applied to the dense
 *     half of the two-optimizer split, model/easy_rec_estimator.py:120-160) as the workgroups behind the cross-tile fix of
 *     the second launch, with the split-K reduce folded into its gradient read: an element of a k-split weight gradient
 *     (every wgrads[i].C must lie inside dense_opt->grad with ldc == N) is summed from the workspace - er_gemm's reduce
 *     order, split by split - stored to grad[] and applied at once.
 * Bit-identical to er_loss_tail, er_gemm_grouped_f32 (at the same wgrad_blocks), er_emb_bwd_fused and
 * er_dense_opt_step_l2 one after the other: six launches become two. */
int er_emb_bwd_fused_tail(er_emb_group* const* groups, int n, const er_grad_group* finish_host, int n_finish, int opt_kind,
                          const er_opt_hyper* hyper, const er_gemm_problem* wgrads_host, int n_wgrads, int32_t wgrad_blocks,
                          const er_loss_tail_job* loss_tail, const er_dense_opt_job* dense_opt, er_stream_t stream);
/* The bias / BatchNorm / activation kernels of SEVERAL layer outputs in one launch (forward) or two (backward: column
 * sums, then finalize + apply): the same-depth layers of parallel stacks - MMoE's experts and task towers
 * (layers/mmoe.py:62-83, model/multi_task_model.py:33-100) - each own launches of a few microseconds otherwise.  Same
 * arithmetic as er_bn_apply_from_stats / er_bn_act_fwd / er_bn_act_bwd(_from_partials), layer by layer.
 *   use_bn 0: y = act(x + bias); 1: batch statistics from `col_stats` ([chunks][N][3], chunks = er_gemm_row_tiles(B) <=
 *   256; moving statistics updated when given); ER_BN_FROZEN: the moving statistics normalise.
 *   backward: dy (leading dimension dy_ld), y_in = the forward's y, `partial` ([chunks][N][2] from er_gemm_f32_bn_bwd /
 *   er_gemm_problem.bn_partial) or NULL (computed here); dx, and dbias / dgamma / dbeta written or accumulated. */
typedef struct er_bn_layer {
  const float* x; const float* bias; const float* gamma; const float* beta;
  float* moving_mean; float* moving_var;
  const float* col_stats; int32_t chunks;
  int32_t B, N, use_bn, act;
  float eps, momentum;
  float* y; float* save_mean; float* save_invstd;
  const float* y_in; const float* dy; int32_t dy_ld; const float* partial;
  float* dx; float* dbias; float* dgamma; float* dbeta; int32_t accumulate;
} er_bn_layer;
int er_bn_fwd_multi(const er_bn_layer* layers_host, int n, er_stream_t stream);
int er_bn_bwd_multi(const er_bn_layer* layers_host, int n, er_stream_t stream);
int er_gemm_f32(int layout, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B,
                int32_t ldb, float* C, int32_t ldc, const float* bias, int accumulate, float* col_stats,
                er_stream_t stream);
int er_gemm_bf16(int layout, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B,
                 int32_t ldb, float* C, int32_t ldc, const float* bias, int accumulate, float* col_stats,
                 er_stream_t stream);
/* bf16 operands IN HBM (BASELINE config 3), both k-contiguous: C[M,N] (+)= A[M,K] . Bt[N,K]^T (+ bias[N]), fp32
 * accumulate on v_mfma_f32_32x32x16_bf16.  Replaces the MatMul of the DCN-v2 cross layer (layers/keras/interaction.py:
 * 249-286) / tf.layers.dense (layers/dnn.py:57-62) and their input gradient (dx = dy . W^T: A = dy, Bt = W [K_in][N_out])
 * when the dense part runs in bf16: 128 x 128 tiles, operands DMA'd HBM -> LDS (global_load_lds, 16 B per lane) in a
 * ring of 4 k-tiles kept in flight across the workgroup barrier.  K, lda, ldb: multiples of 8 (16-byte chunks), A / Bt
 * 16-byte aligned.  C (fp32, optional C +=) and / or C_bf16 (the next layer's operand) are written in one epilogue.
 * er_gemm_bf16_nt_prepare allocates the zero chunk and raises the kernel's LDS limit: call once, outside graph capture.
 * er_cast_bf16: fp32 -> bf16 (round to nearest even) copies of up to any number of matrices in ceil(n / 16) launches:
 * dst[r][c] = src[r][c], or transposed dst[c][r] = src[r][c]; the padding columns of dst up to ld_dst are zeroed
 * (they are read as the k-tail).  Used for the weights' shadows after an optimizer step and for an fp32 activation
 * matrix whose producer does not write bf16 itself. */
typedef struct {
  const float* src;
  uint16_t* dst;
  int64_t rows;      /* of src */
  int32_t cols;      /* of src */
  int32_t ld_src, ld_dst;
  int32_t transpose;
} er_cast_desc;
int er_gemm_bf16_nt_prepare(void);
int er_gemm_bf16_nt(int32_t M, int32_t N, int32_t K, const uint16_t* A, int32_t lda, const uint16_t* Bt, int32_t ldb,
                    float* C, int32_t ldc, uint16_t* C_bf16, int32_t ldc_bf16, const float* bias, int accumulate,
                    er_stream_t stream);
/* What a contraction's epilogue does with its accumulators besides + bias / C += (a HOST record; er_gemm_bf16_nt_epi,
 * er_gemm_f32_cross).  The fusions north_star names for the DCN-v2 cross layer (reference layers/keras/interaction.py:
 * 249-286: x_{l+1} = x0 * (W x_l + b + diag * x_l) + x_l) and for the dense + BatchNorm layers of layers/dnn.py:57-79:
 *   ER_EPI_STATS     col_stats [er_gemm_row_tiles(M)][N][3]: per 64-row tile the Welford triple (count, mean, M2) of the
 *                    output columns (value = acc + bias), as er_gemm_f32's col_stats: the following BatchNorm's batch
 *                    statistics without a pass over C (er_bn_apply_from_stats consumes them).
 *   ER_EPI_BN_BWD    the output is the gradient dy of the activations y = act(BN(z + zbias)) of the layer below: per 64-row
 *                    tile the column sums (sum g, sum g * xhat), g = dy masked by the activation, into bn_partial
 *                    [tiles][bn_n_src][2] (er_gemm_f32_bn_bwd(_cols)' epilogue; output columns [bn_col0, bn_col0 +
 *                    bn_n_src) belong to that layer; bn_n_src 0: all N).
 *   ER_EPI_CROSS_FWD C / C_bf16 = x0 * (acc + bias + diag * xl) + xl with x0, xl fp32 [M][ld]; u (optional) keeps
 *                    acc (the W x_l product without the bias: what the backward's d/dx0 needs).  One launch per cross
 *                    layer where the reference issues MatMul, BiasAdd, Mul, Mul, Add.
 *   ER_EPI_CROSS_BWD the contraction is du_l . W_l^T of cross layer l (du_l = dout_l * x0): v = acc + dout + diag * du_in
 *                    is the whole gradient of x_{l-1}; C (+)= v.  With prev_u != NULL (x_{l-1} is itself the output of a
 *                    cross layer) the elementwise backward of THAT layer happens here too: du_out / du_out_bf16 = v * x0,
 *                    dx0 (+)= v * (prev_u + prev_bias + diag * xl), partial [er_gemm_row_tiles(M)][N] = per-tile column
 *                    sums of du_out (the lower layer's bias gradient: er_colsum_partials_multi finishes them). */
enum { ER_EPI_PLAIN = 0, ER_EPI_STATS = 1, ER_EPI_BN_BWD = 2, ER_EPI_CROSS_FWD = 3, ER_EPI_CROSS_BWD = 4 };
typedef struct er_gemm_epilogue {
  int32_t kind;
  float diag;
  float* col_stats;
  const float* bn_z; const float* bn_zbias; const float* bn_y; const float* bn_mean; const float* bn_invstd;
  float* bn_partial;
  int32_t bn_ld, bn_use_bn, bn_act, bn_col0, bn_n_src;
  int32_t ld_x0, ld_xl, ld_u, ld_dout, ld_du_in, ld_prev_u, ld_dx0, ld_du_out, ld_du_out_bf16, accumulate_dx0;
  const float* x0; const float* xl; float* u;
  const float* dout; const float* du_in; const float* prev_u; const float* prev_bias;
  float* dx0; float* du_out; uint16_t* du_out_bf16; float* partial;
} er_gemm_epilogue;
/* er_gemm_bf16_nt with an epilogue record (epi NULL or kind ER_EPI_PLAIN: er_gemm_bf16_nt).  Epilogues other than PLAIN
 * need N % 4 == 0 and 16-byte aligned rows of every fp32 matrix they touch (8-byte for bf16). */
int er_gemm_bf16_nt_epi(int32_t M, int32_t N, int32_t K, const uint16_t* A, int32_t lda, const uint16_t* Bt, int32_t ldb,
                        float* C, int32_t ldc, uint16_t* C_bf16, int32_t ldc_bf16, const float* bias, int accumulate,
                        const er_gemm_epilogue* epi, er_stream_t stream);
/* The fp32 contraction (er_gemm_f32's kernel, no k-split) with the ER_EPI_CROSS_FWD / ER_EPI_CROSS_BWD epilogue: the
 * values the epilogue needs at a lane's 16 output positions are requested before the k loop. */
int er_gemm_f32_cross(int layout, int32_t M, int32_t N, int32_t K, const float* A, int32_t lda, const float* B, int32_t ldb,
                      float* C, int32_t ldc, const float* bias, int accumulate, const er_gemm_epilogue* epi,
                      er_stream_t stream);
/* C [M][N] = act(BatchNorm(z)) . W [K][N] (+ bias), col_stats as er_gemm_f32's: the A operand is the PRODUCING layer's
 * pre-normalisation output z [M][ldz] (K columns) and its normalisation + activation - ((z - mean) * invstd) * gamma + beta,
 * ReLU: reference layers/dnn.py:62-79 (tf.layers.batch_normalization, then the activation, then the next tf.layers.dense) -
 * is applied while the tile is staged, operation by operation as er_bn_apply_from_stats applies it; y [M][ldy] (NULL: not
 * kept) receives the activations, written once per row tile, for the backward pass.  mean / invstd: er_bn_finalize_from_stats.
 * K % 4 == 0, K <= 256, 16-byte aligned rows of z / y (W: any alignment, e.g. a [K, 1] projection). */
int er_gemm_f32_bn_a(int32_t M, int32_t N, int32_t K, const float* z, int32_t ldz, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, int act, float* y, int32_t ldy, const float* W, int32_t ldw, float* C,
                     int32_t ldc, const float* bias, float* col_stats, er_stream_t stream);
/* C [M][N] = op(x) . W [K][N] (+ bias) for a TALL projection onto N <= 4 columns (DIN's attention score layer [B x L, 32] ->
 * [B x L, 1]: reference model/multi_tower_din.py:80-84, the last tf.layers.dense of the attention DNN, layers/dnn.py:57-62):
 * rows are read whole, 16 bytes per lane, no 64-column MFMA tile.  op(x) = x (mean == NULL), or the producing layer's
 * BatchNorm + activation of x = z as in er_gemm_f32_bn_a (y [M][ldy] keeps the activations, NULL: not kept).  K % 4 == 0. */
int er_gemv_f32_bn_a(int32_t M, int32_t N, int32_t K, const float* x, int32_t ldx, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, int act, float* y, int32_t ldy, const float* W, int32_t ldw, float* C,
                     int32_t ldc, const float* bias, er_stream_t stream);
/* The weight gradient of such a projection: dW [K][N <= 4] (+)= x^T . dz, x [rows][ldx] (K columns), dz [rows][lddz] (the
 * MatMul gradient TF schedules for that tf.layers.dense, reference layers/dnn.py:57-62) - rows in ~512 chunks, per-chunk sums
 * in a fixed lane order, then the chunks in order; dbias [N] (NULL: no) (+)= the column sums of dz (the BiasAdd gradient), from
 * the same pass.  scratch: at least 513 * (K + 1) * N floats, the caller's (stream-ordered). */
int er_wgrad_tall_narrow(int32_t rows, int32_t K, int32_t N, const float* x, int32_t ldx, const float* dz, int32_t lddz,
                         float* dW, int32_t lddw, float* dbias, int accumulate, float* scratch, int64_t scratch_floats,
                         er_stream_t stream);
/* DIN's first attention layer WITHOUT the [B, L, 4E] block (north_star: "DIN-attention as fused HIP kernels"; reference
 * model/multi_tower_din.py:62-80 builds tf.concat([q, h, q - h, q * h], axis=-1) and feeds it to the attention DNN,
 * layers/dnn.py:57-79): the three contractions of the layer take q [B][ldq] and h [B * L][ldh] (E columns each) and form
 * the 4E concat columns while staging (forward, weight gradient), or reduce the gradient of the block to dh and dq in
 * the epilogue (input gradient) - the block exists neither forward nor backward.  fp32 MFMA, 16-byte aligned rows, E % 4 == 0.
 *   er_din_gemm_fwd    z [B * L][N] = [q, h, q - h, q * h] . W [4E][N] (+ bias); col_stats as er_gemm_f32's
 *   er_din_gemm_wgrad  dW [4E][N] (+)= [q, h, q - h, q * h]^T . dz [B * L][N]   (k-splits of <= 2048 rows + fixed-order reduce)
 *   er_din_gemm_dgrad  with dcat = dz . W^T ([B * L][4E], never stored; W's rows visited in an order that puts the four
 *                      segments of 16 embedding positions into one 64-column tile, E % 16 == 0):
 *                      dh [B * L][E] (+)= dcat1 - dcat2 + q * dcat3;  dq [B][E] = sum_l (dcat0 + dcat2 + h * dcat3), summed
 *                      per 64-row tile into dq_partial (er_din_dq_partial_floats(B, L, E) floats) and finished in tile order. */
int er_din_gemm_fwd(const float* q, int32_t ldq, const float* h, int32_t ldh, int32_t B, int32_t L, int32_t E, const float* W,
                    int32_t ldw, int32_t N, const float* bias, float* z, int32_t ldz, float* col_stats, er_stream_t stream);
int er_din_gemm_wgrad(const float* q, int32_t ldq, const float* h, int32_t ldh, int32_t B, int32_t L, int32_t E, const float* dz,
                      int32_t lddz, int32_t N, float* dW, int32_t lddw, int accumulate, er_stream_t stream);
int64_t er_din_dq_partial_floats(int32_t B, int32_t L, int32_t E);
int er_din_gemm_dgrad(const float* dz, int32_t lddz, int32_t N, const float* W, int32_t ldw, const float* q, int32_t ldq,
                      const float* h, int32_t ldh, int32_t B, int32_t L, int32_t E, float* dq, int32_t lddq, float* dh,
                      int32_t lddh, int accumulate_dh, float* dq_partial, er_stream_t stream);
/* out[j] (+)= sum_i x[i * x_stride + j] of SEVERAL narrow matrices (cols <= 64; meant for cols <= 8) in ONE launch: the bias
 * gradients of a multi-task model's tower heads and gates (reference model/mmoe.py:56-68, model/multi_task_model.py:33-100 -
 * tf.layers.dense per tower; each was an er_colsum_acc launch).  Column by column the same sums, in the same order, as
 * er_colsum_acc's narrow form.  jobs: HOST array. */
typedef struct er_colsum_job {
  const float* x;
  int32_t rows, cols, x_stride;
  float* out;
} er_colsum_job;
int er_colsum_narrow_multi(const er_colsum_job* jobs_host, int32_t n_jobs, int accumulate, er_stream_t stream);
/* dst[j] (+)= sum_p partial[p * ld + j] for up to 16 jobs in ONE launch (er_tail_job records, HOST array): the bias
 * gradients of a stack of cross layers from the per-tile column sums their fused backward left. */
int er_colsum_partials_multi(const er_tail_job* jobs_host, int32_t n_jobs, int accumulate, er_stream_t stream);
int er_cast_bf16(const er_cast_desc* descs_host, int n, er_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * K12 embedding-parallel (row-sharded tables, one process per GPU).  Replaces
 *     embedding_parallel_lookup / _get_logits_embedding_parallel
 *       compat/feature_column/feature_column.py:248-357, 416-625
 *     (Unique -> owner = id % world, local row = id // world (:296,317,461-463) -> hvd.alltoall of ids ->
 *      owner-side gather -> hvd.alltoall of rows -> segment sums) and the hvd branch of optimize_loss
 *     compat/optimizers.py:285-345 (embedding gradients divided by the world size :315-316).
 *     The collectives themselves are RCCL all-to-alls issued by the host (torch.distributed); these
 *     entry points are the device work on either side of them.
 *
 * Requester side (a group created from the rank's LOCAL lookups; var may be any [>=1, dim] buffer):
 *   er_emb_group_set_routing: keys become owner * shard_stride + local_base[lookup] + id / world, so
 *     ONE ascending sort groups a step's entries by owner rank and de-duplicates them.
 *   er_emb_route: build + sort; writes the unique routed keys (ascending), *n_unique, for every entry
 *     (source order; lookup l's entries start at the prefix sum of the lookups' capacities) the index
 *     of its unique key or -1, and owner_counts[world] = unique keys per owner (the a2a send splits);
 *     entry_unique_index and owner_counts may be NULL when only the unique keys are wanted.
 *     The lookup itself then runs through er_emb_fwd with table = the rows received from the owners
 *     ([n_unique, dim]) and ids = entry_unique_index.
 *   er_emb_bwd_reduce_routed: reuses the sort of this step's er_emb_route: unique_grads[u * ld + 0..dim) (ld = 0:
 *     dim; larger: dim groups that share a route keep their rows side by side in one exchange buffer) = sum of
 *     the upstream gradients of the entries of unique key u (in-order, deterministic).
 * Owner side:
 *   er_gather_rows: out[i, :] = table[keys[i] - key_sub, :] (key_sub = rank * shard_stride).
 *   er_emb_group_set_active: a group of ONE dense-mode lookup (ids = received local rows, out =
 *     received gradients) processes only its first n_rows entries in er_emb_bwd_update /
 *     er_emb_mark_touched (the received count changes every step; capacity = the lookup's n_rows).
 * Replicated (small) tables, trained data-parallel (their gradient is summed over the ranks like a dense
 *   parameter's - the same math as sharding them):
 *   er_emb_bwd_reduce_dense: the per-row gradient sums of up to 4 groups (one launch; followers of a shared
 *     sort after their leader) go straight into zero-initialised dense buffers dense[i][key * ld[i] + 0..dim),
 *     with dense[i][key * ld[i] + dim] = 1 marking a touched row (ld >= dim + 1) - no run heads or unique list.
 *   er_emb_dense_apply: after the all-reduce of that buffer, one pass over each table: rows with a count > 0
 *     take the optimizer step on grad * grad_scale; under ER_OPT_ADAM (TF-exact) the other rows take the
 *     decay-only step of er_adam_decay_sweep.  Same arithmetic as er_emb_bwd_update over ids = touched rows.
 *   er_scatter_unique (the two-step form): writes the n_unique (device scalar) de-duplicated gradient rows of
 *     er_emb_bwd_reduce into the same kind of buffer.
 * -------------------------------------------------------------------------------------------- */
typedef struct er_dense_apply_desc {
  float* var;         /* [rows, dim] */
  float* m;           /* Adam first moment, or NULL */
  float* v;           /* Adam second moment / Adagrad accumulator, or NULL */
  const float* dense; /* [rows, ld]: gradient (dim floats), count */
  int32_t ld;
  int32_t dim;
  int64_t rows;
  int64_t table_ld;   /* floats between consecutive rows of var / m / v (0 = dim: plain [rows, dim] arrays) */
} er_dense_apply_desc;
int er_emb_group_set_routing(er_emb_group* group, int32_t world, int64_t shard_stride,
                             const int64_t* local_base_host);
int er_emb_group_set_active(er_emb_group* group, int64_t n_rows);
/* Shared sort.  Wide and deep feature groups (DeepFM / WideAndDeep: model/deepfm.py:36-60,
 * layers/input_layer.py:101-125 build both from the SAME input columns) look the same ids up in tables of
 * different width, so the two table groups see identical keys every step.  After
 * er_emb_group_share_sort(group, leader) - accepted only when every lookup i of both groups reads the same
 * ids / offsets with the same rows, key_base, n_rows, max_nnz and routing - er_emb_route and er_emb_bwd_update
 * on `group` reuse the sorted keys, entry permutation and run heads the same call on `leader` left for this
 * step instead of sorting again (the leader must be processed first in every step and outlive the follower;
 * a later er_emb_group_update that breaks the equality silently falls back to an own sort).  On a follower
 * er_emb_route may be called with unique_keys = n_unique = NULL: they would equal the leader's.
 * leader = NULL removes the link. */
int er_emb_group_share_sort(er_emb_group* group, er_emb_group* leader);
/* The same work for up to 4 table groups in ONE launch each ("horizontal fusion": workgroup ranges of one grid run
 * the groups' kernels side by side).  The per-group arithmetic - and every bit of the result - is that of n separate
 * er_emb_catch_up / er_emb_bwd_update calls in the given order of the groups; what changes is that the groups'
 * latency-bound kernels (dependent random accesses that leave most CUs idle) overlap, and the launch count.
 * Groups that share a sort (er_emb_group_share_sort) must come after their leader. */
int er_emb_catch_up_multi(er_emb_group* const* groups_host, const uint32_t* const* unique_keys_host,
                          const int32_t* const* n_unique_host, int n, const er_opt_hyper* hyper,
                          er_stream_t stream);
int er_emb_bwd_update_multi(er_emb_group* const* groups_host, int n, int opt_kind, const er_opt_hyper* hyper,
                            er_stream_t stream);
int er_emb_route(er_emb_group* group, uint32_t* unique_keys, int32_t* n_unique,
                 int64_t* entry_unique_index, int32_t* owner_counts, er_stream_t stream);
int er_emb_bwd_reduce_routed(er_emb_group* group, float* unique_grads, int32_t ld, er_stream_t stream);
int er_gather_rows(const float* table, int64_t table_rows, int32_t dim, const uint32_t* keys, int64_t n,
                   int64_t key_sub, float* out, er_stream_t stream);
/* ... from a table whose rows lie table_ld >= dim floats apart */
int er_gather_rows_ld(const float* table, int64_t table_ld, int64_t table_rows, int32_t dim, const uint32_t* keys,
                      int64_t n, int64_t key_sub, float* out, er_stream_t stream);
int er_scatter_unique(const uint32_t* keys, const float* grads, const int32_t* n_unique,
                      int64_t capacity, int32_t dim, float* dense, int32_t dense_stride,
                      er_stream_t stream);
/* Owner side without a sort.  The keys a rank receives are `n_runs` runs (one per requester, lengths
 * run_counts_host, in rank order), each ascending and duplicate-free - what the requesters' er_emb_route sent.
 *   er_emb_owner_merge: on a group of ONE dense-mode lookup over the received rows (er_emb_group_set_active = the
 *     received count): builds the entries and MERGES the runs into the stable by-key order the reduction needs
 *     (one launch; the following er_emb_bwd_update(_multi) of the group - and of the groups sharing its sort -
 *     reuses it, as after er_emb_route).
 *   er_emb_owner_serve: for up to 4 such groups (a leader before its followers) in one launch: every distinct
 *     received row is caught up (lazy dense decay, if enabled on the group) and - hyper == NULL: no catch-up, the
 *     rows are served as they are (inference after er_emb_flush_decay: no row update follows to advance last_step) -
 *     written to rows_out[i][entry * ld[i] + 0..dim) of every entry that asked for it (ld_host NULL or 0: dim) -
 *     er_emb_route + er_emb_catch_up + er_gather_rows of the sorted form. */
int er_emb_owner_merge(er_emb_group* group, const int32_t* run_counts_host, int n_runs, er_stream_t stream);
/* The same exchange with NO host-visible sizes (no host synchronisation in the step; every launch is static, so
 * the segments between the collectives replay as hipGraphs and the host runs ahead of the device):
 *   er_emb_group_set_peer_capacity: on a routed requester group (any lookups: the per-lookup sort places the keys
 *     directly, the device-wide sort re-lays its ascending key list out in one more launch): er_emb_route then
 *     writes the keys of owner w at
 *     unique_keys[w * peer_cap ...] and entry_unique_index points into that padded layout (rows and row gradients
 *     of owner w at [w * peer_cap, ...)), so keys, rows and row gradients travel in equal-split all-to-alls.
 *     count_header = 1: owner w's segment of unique_keys is peer_cap + 1 slots, [count, keys ...] - the counts ride
 *     in the key all-to-all; 0: they travel as their own [world] int32 all-to-all (owner_counts).  An owner with
 *     more than peer_cap keys sets a device flag read back by er_emb_route_overflow (a blocking copy: poll it off
 *     the critical path); that step's results are void.
 *   er_emb_owner_ids: ids[q * peer_cap + j] = key j of run q - key_sub for j < count[q], else -1 (padding).
 *     counts != NULL: keys at recv_keys[q * peer_cap + j]; counts == NULL: the header form, run q =
 *     recv_keys[q * (peer_cap + 1)] = count, then keys.  counts_out (may be NULL) receives the counts either way.
 *   er_emb_owner_merge_padded: er_emb_owner_merge for runs at q * peer_cap with device-side lengths, on a group
 *     whose lookup spans all n_runs * peer_cap received slots; padding sorts behind the real keys. */
int er_emb_group_set_peer_capacity(er_emb_group* group, int64_t peer_cap, int32_t count_header);
int er_emb_route_overflow(er_emb_group* group, int32_t* overflow_host);
int er_emb_owner_ids(const uint32_t* recv_keys, const int32_t* counts, int n_runs, int64_t peer_cap, int64_t key_sub,
                     int64_t* ids, int32_t* counts_out, er_stream_t stream);
int er_emb_owner_merge_padded(er_emb_group* group, const int32_t* counts, int n_runs, int64_t peer_cap,
                              er_stream_t stream);
/* er_emb_owner_ids + er_emb_owner_merge_padded (with the entry build in front of it) as ONE launch: the key of any received
 * slot follows from the received runs themselves, so the merge's searches read those and nothing waits for another
 * workgroup's build; every array the three launches leave (ids, counts_out, the groups' entry arrays, the merged keys / entry
 * indices / head flags) is left bit for bit.  The group's one lookup must read `ids`.  build_lag1_tables != 0: workgroups
 * behind the merge's build the closed-form replay's lag-1 table, and the er_emb_owner_serve that follows on the stream does
 * not launch its own build. */
int er_emb_owner_ids_merge(er_emb_group* group, const uint32_t* recv_keys, const int32_t* counts, int n_runs,
                           int64_t peer_cap, int64_t key_sub, int64_t* ids, int32_t* counts_out, int build_lag1_tables,
                           er_stream_t stream);
int er_emb_owner_serve(er_emb_group* const* groups_host, float* const* rows_out_host, const int32_t* ld_host,
                       int n, const er_opt_hyper* hyper, er_stream_t stream);
int er_emb_bwd_reduce_dense(er_emb_group* const* groups_host, float* const* dense_host, const int32_t* ld_host,
                            int n, er_stream_t stream);
int er_emb_dense_apply(const er_dense_apply_desc* descs_host, int n, int opt_kind, const er_opt_hyper* hyper,
                       er_stream_t stream);
/* The end of an embedding-parallel step as TWO launches: er_emb_bwd_update_multi(groups) - the owners' row update (reference
 * compat/optimizers.py:285-345: the optimizer applied to the summed IndexedSlices an owner received) - whose second launch
 * (the cross-tile fix) also carries er_emb_dense_apply(descs, n_apply) - the replicated tables after the all-reduce - and
 * er_dense_opt_step_l2(*dense_opt) - the dense variables - as further workgroup ranges: owned rows, replicated tables and
 * dense variables are disjoint, the bodies are the three launches' own, every bit of the result is theirs.  n_apply may be
 * 0; both row optimizers take (opt_kind, hyper), the dense one what its job says. */
int er_emb_owner_update_tail(er_emb_group* const* groups_host, int n, int opt_kind, const er_opt_hyper* hyper,
                             const er_dense_apply_desc* descs_host, int n_apply, const er_dense_opt_job* dense_opt,
                             er_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Dense-variable optimizer: one launch over the flat parameter buffer.  Replaces ApplyAdam /
 *     ApplyAdagrad / ApplyGradientDescent issued per variable by optimize_loss
 *     compat/optimizers.py:412-416, plus the kernel L2 gradient l2*W
 *     (kernel_regularizer of layers/dnn.py:57-62).   g = grad_scale*grad + l2coef*w
 * -------------------------------------------------------------------------------------------- */
int er_dense_opt_step(float* w, float* m, float* v, const float* grad, const float* l2coef,
                      int64_t n, int opt_kind, const er_opt_hyper* hyper, er_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * Gradient clipping by global norm (train_config.gradient_clipping_by_norm).  Replaces `_get_grad_norm` +
 *     clip_ops.clip_by_global_norm of optimize_loss, compat/optimizers.py:365-376 and :453-481: norm = sqrt of the
 *     sum of squares of every gradient (dense: as the optimizer sees it, grad_scale*grad + l2coef*w; embeddings: the
 *     de-duplicated row sums of a lookup, times grad_scale), multiplier = clip_norm * min(1/norm, 1/clip_norm).
 * er_gradsq_rows : acc[0] (+)= weight * sum of x[r*ld + c]^2 over c < cols and the valid rows r < max_rows; with
 *     seg_counts (device int32 [n_seg]) row r is valid iff (r % seg_stride) < seg_counts[r / seg_stride] (a prefix
 *     count: n_seg = 1, seg_stride = max_rows; the fixed-capacity exchange: per-owner counts, seg_stride = capacity).
 * er_gradsq_dense: acc[0] (+)= sum_i (hyper->grad_scale * grad[i] + l2coef[i] * w[i])^2   (l2coef may be NULL)
 * er_clip_scale  : norm = sqrt(normsq[0]); records[i].clip_scale = multiplier for i < n_records (records: DEVICE
 *     er_opt_hyper array, e.g. the step's [embedding, dense] pair written by er_hyper_select); norm_out optional.
 * er_emb_apply_unique: the row-wise optimizer of er_emb_bwd_update applied to ready-made row sums (keys / grads /
 *     n_unique as produced by er_emb_bwd_reduce or er_emb_route + er_emb_bwd_reduce_routed; ld = floats per grads
 *     row, 0 = dim): what the fused reduce+apply does once the norm of ALL gradients is known.
 * Fixed summation order: deterministic. */
int er_gradsq_rows(const float* x, int64_t max_rows, int32_t cols, int32_t ld, const int32_t* seg_counts,
                   int32_t n_seg, int64_t seg_stride, float weight, float* acc, int accumulate, er_stream_t stream);
int er_gradsq_dense(const float* w, const float* grad, const float* l2coef, int64_t n, const er_opt_hyper* hyper,
                    float* acc, int accumulate, er_stream_t stream);
int er_clip_scale(const float* normsq, float clip_norm, er_opt_hyper* records, int32_t n_records, float* norm_out,
                  er_stream_t stream);
int er_emb_apply_unique(er_emb_group* g, const uint32_t* unique_keys, const float* unique_grads, int32_t ld,
                        const int32_t* n_unique, int opt_kind, const er_opt_hyper* hyper, er_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * K16 Streaming AUC.  Replaces the per-threshold confusion-matrix update of tf.metrics.auc
 *     (RankModel.build_metric_graph, easy_rec/python/model/rank_model.py:358-373; default 200 thresholds).
 * counts: uint64 [2][num_thresholds + 1], zero before the first batch; counts[label][k] += 1 for every example
 * whose prediction is greater than exactly k of the (ascending) thresholds; examples with weight <= 0 are
 * skipped (weights may be NULL).  tp(t) = sum_{k > t} counts[1][k], fp(t) likewise from counts[0]; the area is
 * finished on the host (easyrec_amd/core/metrics.py).
 * -------------------------------------------------------------------------------------------- */
int er_auc_update(const float* probs, const float* labels, const float* weights, int64_t n,
                  const float* thresholds, int32_t num_thresholds, uint64_t* counts, er_stream_t stream);
/* K16b grouped AUC: gAUC / session AUC (reference core/metrics.py:59-108 `_separated_auc_impl`, :260-297; the reference
 * gathers (label, prediction, key) in a tf.py_func and averages sklearn's roc_auc_score per key on the host).
 * keys / preds / labels: the n accumulated rows SORTED by (key, prediction) ascending; work: 3 * n doubles, zeroed;
 * out: 3 doubles, zeroed: out[0] = sum over the keys with both classes of w * AUC(key), out[1] = sum of w, out[2] = the
 * number of such keys; AUC(key) = the Mann-Whitney statistic with tied predictions at their average rank (what
 * roc_auc_score returns); w by reduction: 0 'mean' 1, 1 'mean_by_sample_num' rows, 2 'mean_by_positive_num' positives. */
int er_grouped_auc(const int64_t* keys, const float* preds, const float* labels, int64_t n, int reduction, double* work,
                   double* out, er_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * K15 DLRM dot interaction.  Replaces einsum('bne,bme->bnm') + the upper-triangle slicing / concat of
 *     DLRM.build_predict_graph easy_rec/python/model/dlrm.py:44-57.
 * x: [B, F*D] (row stride x_stride), the F feature vectors of an example back to back;
 * out[b, p] = <x[b,i,:], x[b,j,:]> for the pairs in the reference's order: i = 0..F-1, j = i + offset..F-1,
 * offset = 0 with self_interaction (arch_interaction_itself), else 1; P = F(F-1)/2 (+ F).
 * bwd: dx[b,f,:] (+)= sum over pairs containing f of g[b,p] * the partner vector (twice for a self pair).
 * -------------------------------------------------------------------------------------------- */
int er_dot_interaction_fwd(const float* x, int32_t B, int32_t F, int32_t D, int32_t x_stride,
                           int self_interaction, float* out, int32_t out_stride, er_stream_t stream);
int er_dot_interaction_bwd(const float* x, const float* g, int32_t B, int32_t F, int32_t D,
                           int32_t x_stride, int self_interaction, int32_t g_stride, float* dx,
                           int32_t dx_stride, int accumulate, er_stream_t stream);

/* ----------------------------------------------------------------------------------------------
 * K14  Sharded embedding checkpoint files (host code, no device work).
 * Replaces the reference's two native TF ops and the python writer around them:
 *   ops/src/load_dense_embed.cc:54-135 (LoadEmbedOp), ops/src/load_kv_embed.cc:60-163 (LoadKVEmbedOp),
 *   compat/embedding_parallel_saver.py:99-127 (_save_dense_embedding).
 * Files live in "<ckpt_path>-embedding/": "<var_name>-part-<k>.bin" = raw float32 rows of worker k's shard
 * (rows k, k + P, k + 2P, ... of the table, P = number of parts); var_name is "embed-" + the TF variable
 * name with '/' replaced by "__" (e.g. "embed-input_layer__user_id_embedding__embedding_weights:0").
 *   er_save_dense_embed : write this worker's shard; worker 0 deletes parts of workers >= task_num.
 *   er_load_dense_embed : fill out_vals [embed_part_size, embed_dim] (host) with the rows id % task_num ==
 *     task_index at local position id / task_num, from parts written by ANY number of workers; rows the files
 *     do not hold are zero.  Fails unless embed_part_size or embed_part_size - 1 rows were found (the check of
 *     load_dense_embed.cc:121-126).
 *   er_load_kv_embed    : key/value tables ("-part-<k>.key" int64, ".val" float32 [n, embed_dim]): the keys
 *     with key mod task_num == task_index.  Call with out_keys = NULL to get the count in *n_keys, then with
 *     buffers of that capacity.  (The reference shuffles the result; here it is in file order.)
 * -------------------------------------------------------------------------------------------- */
/* CRC-32C (Castagnoli) of n bytes continuing from crc (0 to start): TensorFlow's checksum for tensor-bundle entries and
 * table blocks (tensorflow/core/lib/hash/crc32c.h) - the dense variables' checkpoint files (utils/tensor_bundle.py). Host. */
uint32_t er_crc32c(uint32_t crc, const void* data, int64_t n);
int er_save_dense_embed(const char* ckpt_path, const char* var_name, int32_t task_index, int32_t task_num,
                        const float* vals_host, int64_t rows, int32_t embed_dim);
int er_load_dense_embed(const char* ckpt_path, const char* var_name, int32_t task_index, int32_t task_num,
                        int32_t embed_dim, int64_t embed_part_size, float* out_vals_host,
                        int64_t* rows_loaded);
int er_load_kv_embed(const char* ckpt_path, const char* var_name, int32_t task_index, int32_t task_num,
                     int32_t embed_dim, int64_t capacity, int64_t* out_keys_host, float* out_vals_host,
                     int64_t* n_keys);

#ifdef __cplusplus
}
#endif
#endif /* EASYREC_HIP_H_ */
