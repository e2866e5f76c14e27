/*
 * mi355_paged.h -- C ABI of libmi355_paged.so: paged-attention decode, KV slot
 * write, causal prefill and token sampling as hand-written HIP kernels for
 * MI355X (gfx950 / CDNA4).
 *
 * The reference this library sits under is pure Python; it has no FFI of its
 * own.  Each entry point below replaces the body of one Python function or
 * loop of the reference (cited per function, paths relative to the reference
 * checkout) and is what a binding for that function would call.
 *
 * Conventions
 *  - Every pointer except `stream` and the `const char*` returns is a DEVICE
 *    pointer owned by the caller (e.g. a torch tensor's data_ptr()).
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *    All work is enqueued on it and the call returns without synchronising.
 *  - The library allocates no device memory.  Scratch space comes in through
 *    the `workspace` argument, sized by mi355_paged_decode_workspace_bytes()
 *    (mi355_paged_decode_static keeps its scratch in the library's device image).
 *  - Strides are in ELEMENTS of the tensor's dtype.  Head rows (the innermost
 *    D elements) must be contiguous and 16-byte aligned.
 *  - `dtype`: MI355_F32 = 0, MI355_BF16 = 1, MI355_F16 = 2.  Accumulation is
 *    always fp32.
 *  - Return value: 0 on success, a negative MI355_E* code for a rejected
 *    argument, a positive value = hipError_t from the launch.  A message for
 *    the last failure on the calling thread is available from
 *    mi355_last_error().
 *  - Thread safety: the library keeps no mutable global state besides the
 *    thread-local error string (and the shared scratch array of
 *    mi355_paged_decode_static, see there); calls may be made from any thread.
 */
#ifndef MI355_PAGED_H
#define MI355_PAGED_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_ABI_VERSION 1

#define MI355_F32 0
#define MI355_BF16 1
#define MI355_F16 2

#define MI355_OK 0
#define MI355_EINVAL (-1)      /* bad shape / stride / null pointer            */
#define MI355_EUNSUPPORTED (-2) /* shape outside the compiled kernel set       */
#define MI355_EWORKSPACE (-3)   /* workspace missing or too small              */
#define MI355_ECOLLECTIVE (-4)  /* the collective library reported an error     */

/* Library ABI version (== MI355_ABI_VERSION the library was built with). */
int mi355_abi_version(void);

/* Message describing the last non-zero return on this thread ("" if none). */
const char* mi355_last_error(void);

/*
 * Paged decode attention for a batch of sequences, one new query token each.
 * Replaces the body of paged_attention_decode
 * (01_paged_attention/paged_attention.py:67-101) and the decode branch of
 * SelfAttention.forward (05_mini_vllm/mini_vllm.py:460-481), batched:
 *
 *   for b in [0,B):  for kv_h in [0,Hkv):
 *     rows i in [0, context_lens[b]) of K and V are fetched from physical block
 *     block_tables[b][i / block_size], row i % block_size, head kv_h;
 *     out[b, kv_h*g .. (kv_h+1)*g) = softmax(q . K^T * scale) . V,  g = Hq / Hkv
 *
 *   out, q        [B, Hq, D]; q rows may be strided (q_stride_seq elements
 *                 between sequences; heads are D apart), out is contiguous
 *   k_cache,
 *   v_cache       element (block, head, row, d) at
 *                 block*kv_stride_block + head*kv_stride_head + row*kv_stride_tok + d.
 *                 [NB,Hkv,BS,D] (reference 01 layout) and [NB,BS,H,D]
 *                 (reference 05 layout) are both expressible.
 *   block_tables  [B, max_blocks] int32; only the first
 *                 ceil(context_lens[b]/block_size) entries of a row are read;
 *                 ids are clamped to [0, num_blocks)
 *   context_lens  [B] int32; a row with 0 yields zeros; values are clamped to
 *                 min(max_context, max_blocks*block_size)
 *   max_context   host-side upper bound on context_lens (sizes the launch)
 *   D in {16,32,64,128,256}; Hq % Hkv == 0.
 *   workspace     >= mi355_paged_decode_workspace_bytes(...) bytes, 16-B aligned
 *                 (may be NULL when that function returns 0).  Its contents on entry may be
 *                 anything and nothing in it is kept between calls: a split launch writes its
 *                 fp32 partials there and, when it merges them itself, per-group arrival
 *                 counters that carry a tag of the launch (so leftovers never count).  Calls
 *                 whose kernels may overlap on the GPU must not share a workspace.
 */
int mi355_paged_decode(void* out, const void* q, int64_t q_stride_seq,
                       const void* k_cache, const void* v_cache,
                       const int32_t* block_tables, const int32_t* context_lens,
                       int32_t B, int32_t Hq, int32_t Hkv, int32_t D,
                       int32_t block_size, int32_t max_blocks, int32_t num_blocks,
                       int32_t max_context,
                       int64_t kv_stride_block, int64_t kv_stride_head, int64_t kv_stride_tok,
                       float scale, int32_t dtype,
                       void* workspace, size_t workspace_bytes, void* stream);

/* Scratch bytes mi355_paged_decode needs for this shape (split-KV partials and arrival counters).
 * The launch plan also depends on the cache's strides, which this query does not see: the answer
 * covers whichever plan a launch of this shape may pick. */
size_t mi355_paged_decode_workspace_bytes(int32_t B, int32_t Hq, int32_t Hkv, int32_t D,
                                          int32_t block_size, int32_t max_context, int32_t dtype);

/*
 * mi355_paged_decode without the workspace argument, for callers that cannot allocate device
 * memory: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * It has no counterpart in the reference.  Arguments, kernels and results are those of
 * mi355_paged_decode (bit for bit); the split-KV partials live in an array of
 * MI355_STATIC_SCRATCH_BYTES that is part of the library's device image, so every GPU the
 * library runs on holds that much memory until the process ends.  The call performs no
 * hipMalloc / hipFree; its first split call on a device looks the array's address up once
 * (hipGetSymbolAddress), with that device current: make that first call outside a capture (one
 * eager call per device), later calls may be captured and replayed.
 *   - A shape whose partials (mi355_paged_decode_workspace_bytes) exceed
 *     MI355_STATIC_SCRATCH_BYTES is rejected with MI355_EWORKSPACE before any launch.
 *   - All calls on one GPU share the array: calls whose kernels may overlap on that GPU
 *     (different streams, different threads) race on it.  Calls on one stream are safe, and
 *     nothing in the array needs clearing between calls: every record a call reads, it wrote,
 *     and arrival counters only count under the tag of the launch that reads them.
 */
#define MI355_STATIC_SCRATCH_BYTES (16u << 20)
int mi355_paged_decode_static(void* out, const void* q, int64_t q_stride_seq,
                              const void* k_cache, const void* v_cache,
                              const int32_t* block_tables, const int32_t* context_lens,
                              int32_t B, int32_t Hq, int32_t Hkv, int32_t D,
                              int32_t block_size, int32_t max_blocks, int32_t num_blocks,
                              int32_t max_context,
                              int64_t kv_stride_block, int64_t kv_stride_head, int64_t kv_stride_tok,
                              float scale, int32_t dtype, void* stream);

/*
 * One decode step of an attention layer in one call: store every sequence's newest K/V row in
 * its cache, then attend.  Replaces the decode branch of SelfAttention.forward as a whole --
 * the indexed cache assignment 05_mini_vllm/mini_vllm.py:421-427 followed by the per-sequence
 * attention :460-481 (01_paged_attention/paged_attention.py:48-51 + :67-101) -- and equals
 * mi355_kv_write_slots followed by mi355_paged_decode bit for bit, caches included:
 *
 *   for b in [0,B) with 1 <= context_lens[b] <= min(max_context, max_blocks*block_size):
 *     position context_lens[b] - 1 of sequence b (block block_tables[b][(ctx-1) / block_size],
 *     row (ctx-1) % block_size, every kv head) := k_new[b], v_new[b]
 *   then out = mi355_paged_decode(q, caches, block_tables, context_lens)
 *
 * i.e. context_lens already counts the new token, as it does after write_kv in the reference.
 * Rows with context_lens[b] < 1 or beyond the bound write nothing.
 *   k_new, v_new  [B, Hkv, D], new_stride_seq elements between sequences, 16-B aligned
 *   k_cache,
 *   v_cache       written at B * Hkv rows, otherwise read; all other arguments as in
 *                 mi355_paged_decode (same workspace query)
 * On shapes the VALU decode kernel serves, the wave that reads a sequence's last position
 * stores the row there first: no launch of its own for the write.  Grouped-query shapes on the
 * matrix-core kernel get the write as a small launch ahead of the attention launch.
 * A slot written for one sequence must not lie within another sequence's context in the same
 * call: that reader may see the slot before or after the write.
 */
int mi355_paged_decode_append(void* out, const void* q, int64_t q_stride_seq,
                              const void* k_new, const void* v_new, int64_t new_stride_seq,
                              void* k_cache, void* v_cache,
                              const int32_t* block_tables, const int32_t* context_lens,
                              int32_t B, int32_t Hq, int32_t Hkv, int32_t D,
                              int32_t block_size, int32_t max_blocks, int32_t num_blocks,
                              int32_t max_context,
                              int64_t kv_stride_block, int64_t kv_stride_head, int64_t kv_stride_tok,
                              float scale, int32_t dtype,
                              void* workspace, size_t workspace_bytes, void* stream);

/*
 * mi355_paged_decode_append with the rotary embedding of the step applied on the way: q and
 * k_new arrive UN-rotated, as the fused projection left them, and the kernel rotates both for
 * position context_lens[b] - 1 (the new token's) before use; the rotated K row is what the cache
 * receives.  The whole decode branch of SelfAttention.forward after the projection --
 * 05_mini_vllm/mini_vllm.py:415-419 (cos/sin gather, rotate_half), :421-427 (cache write) and
 * :460-481 (attention) -- in one launch; equals mi355_rope_kv_write (positions = context - 1,
 * slots = that position's) followed by mi355_paged_decode bit for bit, caches included (the
 * rotation is two fp32 products and a sum rounded once, in both).  q itself is left un-rotated.
 *   cos_table, sin_table   [table_len, D] in the cache's dtype (cos[t][i] == cos[t][i + D/2]);
 *                          positions beyond the table use its last row
 * Shapes that mi355_paged_decode_append serves with a separate cache-write launch (the
 * matrix-core kernel's grouped-query shapes, plans whose LDS is full) return MI355_EUNSUPPORTED
 * before anything is launched: use mi355_rope_kv_write + mi355_paged_decode there.
 */
int mi355_paged_decode_append_rope(void* out, const void* q, int64_t q_stride_seq,
                                   const void* k_new, const void* v_new, int64_t new_stride_seq,
                                   void* k_cache, void* v_cache,
                                   const int32_t* block_tables, const int32_t* context_lens,
                                   const void* cos_table, const void* sin_table, int32_t table_len,
                                   int32_t B, int32_t Hq, int32_t Hkv, int32_t D,
                                   int32_t block_size, int32_t max_blocks, int32_t num_blocks,
                                   int32_t max_context,
                                   int64_t kv_stride_block, int64_t kv_stride_head, int64_t kv_stride_tok,
                                   float scale, int32_t dtype,
                                   void* workspace, size_t workspace_bytes, void* stream);

/*
 * Scatter T new K/V rows into the paged caches by flat slot id
 * (slot = block*block_size + row).  Replaces the per-token write loops
 * 01_paged_attention/paged_attention.py:118-123, KVCachePool.write_kv (:48-51)
 * and the indexed assignment 05_mini_vllm/mini_vllm.py:421-427.
 *   k, v   [T, Hkv, D] with src_stride_tok elements between tokens
 *   slots  [T] int64; a negative slot or one >= num_blocks*block_size is skipped
 */
int mi355_kv_write_slots(void* k_cache, void* v_cache, const void* k, const void* v,
                         const int64_t* slots, int32_t T, int32_t Hkv, int32_t D,
                         int64_t src_stride_tok, int32_t block_size, int32_t num_blocks,
                         int64_t kv_stride_block, int64_t kv_stride_head, int64_t kv_stride_tok,
                         int32_t dtype, void* stream);

/*
 * Causal self-attention over packed variable-length sequences (prompt
 * processing).  Replaces 01_paged_attention/paged_attention.py:125-136 and the
 * prefill branch 05_mini_vllm/mini_vllm.py:437-458.
 *   q [T, Hq, D], k/v [T, Hkv, D] (token strides given), out [T, Hq, D] contiguous
 *   cu_seqlens [nseq+1] int32: sequence i owns tokens cu_seqlens[i]..cu_seqlens[i+1]-1,
 *               cu_seqlens[0] = 0, cu_seqlens[nseq] = total_tokens
 *   total_tokens T (host copy of cu_seqlens[nseq]); max_seqlen: host-side upper
 *               bound on any sequence length
 */
int mi355_prefill_causal(void* out, const void* q, const void* k, const void* v,
                         int64_t q_stride_tok, int64_t kv_stride_tok,
                         const int32_t* cu_seqlens, int32_t nseq, int32_t total_tokens,
                         int32_t max_seqlen, int32_t Hq, int32_t Hkv, int32_t D, float scale,
                         int32_t dtype, void* stream);

/*
 * Greedy sampling: out_ids[b] = argmax(logits[b, :vocab]).  Replaces
 * 05_mini_vllm/mini_vllm.py:611-613 (torch.argmax per request + .item()).
 * Ties give the lowest index; NaN counts as the maximum (first NaN wins),
 * matching torch.argmax.  logits rows are row_stride elements apart.
 */
int mi355_sample_argmax(int32_t* out_ids, const void* logits, int32_t B, int32_t vocab,
                        int64_t row_stride, int32_t dtype, void* stream);

/*
 * Stochastic sampling: temperature -> softmax -> optional top-k -> optional
 * top-p (nucleus) -> one multinomial draw per row.  Replaces
 * 05_mini_vllm/mini_vllm.py:615-630.  Nucleus rule as in the reference: with
 * tokens ranked by descending probability, a token is dropped when the mass
 * ranked strictly before it is >= top_p (the top token always survives).
 *   temperature [B] float (> 0), top_p [B] float or NULL (= 1.0),
 *   top_k [B] int32 or NULL (<= 0 = off; not in the reference, additive)
 *   seed/offset select the counter-based random stream: row b draws from
 *   philox(seed, offset + b).
 *   out_probs: optional [B, vocab] float; when non-NULL receives the
 *   renormalised post-filter distribution (for tests / inspection).  A row
 *   without a cut (top_p >= 1, no top-k) keeps the softmax as it is, like the
 *   reference, which renormalises only after a cut (:619-626).
 *   Probabilities below the smallest normal float are taken as 0; a NaN logit gets
 *   probability 0 (it is never drawn), unlike mi355_sample_argmax above.
 */
int mi355_sample_topk_topp(int32_t* out_ids, const void* logits, const float* temperature,
                           const float* top_p, const int32_t* top_k, uint64_t seed,
                           uint64_t offset, int32_t B, int32_t vocab, int64_t row_stride,
                           int32_t dtype, float* out_probs, void* stream);

/*
 * Batched slot mapping on the device: for sequence b and i in [0, counts[b]):
 *   out_slots[cu_tokens[b] + i] =
 *       block_tables[b][(starts[b]+i) / block_size] * block_size + (starts[b]+i) % block_size
 * Replaces compute_slot_mapping (01_paged_attention/paged_attention.py:54-64)
 * and Scheduler._compute_slot_mapping (05_mini_vllm/mini_vllm.py:209-219).
 */
int mi355_slot_mapping(int64_t* out_slots, const int32_t* block_tables, const int32_t* starts,
                       const int32_t* cu_tokens, int32_t B, int32_t max_blocks,
                       int32_t block_size, int32_t total_tokens, void* stream);

/*
 * Latent-space MLA attention (weight-absorbed form) for `rows` query rows.  Replaces steps
 * 3-7 of MultiHeadLatentAttention.forward (04_mla/mini_mla.py:234-299): instead of expanding
 * every cached c_kv row to per-head K and V, heads attend directly to the cached rows
 *
 *   score[h][t] = scale * ( q[h][:R] . c_kv[t]  +  q[h][R:] . rope(k_rope[t], position t) )
 *   out[h]      = softmax_t(score[h]) . c_kv[t]                    -> [rows, H, R] fp32
 *
 * where the caller has folded the key up-projection into q[:, :, :R] and applies the value
 * up-projection to `out`.  rope() is the rotate-half form with cos/sin tables [table_len, P]
 * (fp32; cos[t][i] == cos[t][i + P/2]); cached k_rope is stored un-rotated, as the reference
 * caches it, and token t of a sequence sits at position t.
 *   q         [rows, H, R+P] (q_stride_row elements between rows), bf16 / f16 -- or fp32 (dtype
 *             MI355_F32, then c_kv and k_rope are fp32 too): the fp32 matrix instruction keeps
 *             every product exact, for callers that compare with an fp32 reference; it serves
 *             R in {32,64,128} with P in {8,16,32,64} and R in {256,512} with P in {32,64}
 *   c_kv      row of token t of sequence s at  s*ckv_stride_seq + t*ckv_stride_tok   (R elements)
 *   k_rope    likewise with the krope strides                                           (P elements)
 *   context_lens [rows]; row_seq [rows] = cache sequence each row reads (NULL: row i reads i) --
 *             lets T query tokens of one sequence be T rows with context past + t + 1 (causal)
 *   R in {32,64,128,256,512}; P a multiple of 8, <= 64
 */
int mi355_mla_decode(float* out, const void* q, int64_t q_stride_row,
                     const void* c_kv, int64_t ckv_stride_seq, int64_t ckv_stride_tok,
                     const void* k_rope, int64_t krope_stride_seq, int64_t krope_stride_tok,
                     const float* cos_table, const float* sin_table, int32_t table_len,
                     const int32_t* context_lens, const int32_t* row_seq,
                     int32_t rows, int32_t H, int32_t R, int32_t P, int32_t max_context,
                     float scale, int32_t dtype, void* workspace, size_t workspace_bytes, void* stream);

/*
 * mi355_mla_decode over a BLOCK cache: the latent rows of a sequence live in fixed-size blocks of a
 * shared pool, named by the sequence's block table, exactly as the K/V rows of
 * mi355_paged_decode do -- so that latent caches can use the same block allocator, block tables
 * and prefix sharing.  The reference keeps dense per-sequence tensors (MLAKVCache,
 * 04_mla/mini_mla.py:79-132, and the past_c_kv / past_k_rope arguments of forward, :203-232); the
 * attention computed is the same (:234-299), and a dense cache copied into any set of blocks
 * gives the dense call's result bit for bit.
 *   c_kv, k_rope  row `r` of block `blk`: blk * *_stride_block + r * *_stride_tok.  A pool
 *                 [num_blocks, block_size, R + P] holding [c_kv | k_rope] per row is c_kv = pool,
 *                 k_rope = pool + R, both with strides (block_size * (R + P), R + P).
 *   block_tables  [sequences, max_blocks] int32; row_seq[row] (or row) selects the table row;
 *                 token t sits in block block_tables[s][t / block_size] (ids clamped to
 *                 [0, num_blocks)), row t % block_size.  block_size must be a power of two.
 *   context_lens are clamped to min(max_context, max_blocks * block_size).
 * Everything else (q, out, tables, workspace from mi355_mla_decode_workspace_bytes) as in
 * mi355_mla_decode.
 */
int mi355_mla_decode_paged(float* out, const void* q, int64_t q_stride_row,
                           const void* c_kv, int64_t ckv_stride_block, int64_t ckv_stride_tok,
                           const void* k_rope, int64_t krope_stride_block, int64_t krope_stride_tok,
                           const int32_t* block_tables, int32_t block_size, int32_t max_blocks,
                           int32_t num_blocks,
                           const float* cos_table, const float* sin_table, int32_t table_len,
                           const int32_t* context_lens, const int32_t* row_seq,
                           int32_t rows, int32_t H, int32_t R, int32_t P, int32_t max_context,
                           float scale, int32_t dtype, void* workspace, size_t workspace_bytes, void* stream);

/* Scratch bytes mi355_mla_decode (and mi355_mla_decode_paged) needs for this shape. */
size_t mi355_mla_decode_workspace_bytes(int32_t rows, int32_t H, int32_t R, int32_t P,
                                        int32_t max_context, int32_t dtype);

/*
 * Rotate-half RoPE applied in place to the query and key rows of T tokens, plus the write of
 * the rotated key rows and of the value rows into the paged caches by flat slot id -- the
 * three steps 05_mini_vllm/mini_vllm.py:415-427 (cos/sin gather by position, q/k rotation,
 * indexed cache assignment) in one launch:
 *
 *   x'[i]       = x[i]       * cos[pos][i]       - x[i + D/2] * sin[pos][i]         (i < D/2)
 *   x'[i + D/2] = x[i + D/2] * cos[pos][i + D/2] + x[i]       * sin[pos][i + D/2]
 *
 *   q [T, Hq, D], k / v [T, Hkv, D] with their own token strides (views into one fused
 *   projection output are fine); cos_table / sin_table [table_len, D] in the same dtype;
 *   positions [T] int64 (negative values index from the end of the table, as in torch);
 *   slots [T] int64 or NULL (NULL: rotate only, nothing is written to the caches; a slot
 *   outside [0, num_blocks*block_size) skips that token's cache write);
 *   caches addressed by (block, head, row) strides as in mi355_paged_decode.
 *   D must be a multiple of 16 (16-bit dtypes) or 8 (fp32).
 */
int mi355_rope_kv_write(void* q, void* k, const void* v, int64_t q_stride_tok, int64_t k_stride_tok,
                        int64_t v_stride_tok, const void* cos_table, const void* sin_table,
                        int32_t table_len, const int64_t* positions, const int64_t* slots,
                        void* k_cache, void* v_cache, int32_t T, int32_t Hq, int32_t Hkv, int32_t D,
                        int32_t block_size, int32_t num_blocks, int64_t kv_stride_block,
                        int64_t kv_stride_head, int64_t kv_stride_tok, int32_t dtype, void* stream);

/*
 * RMS normalisation of T rows, optionally preceded by a residual add
 * (RMSNorm.forward, 05_mini_vllm/mini_vllm.py:344-352, and the `x + attn(...)` of
 * TransformerLayer.forward :511-514):
 *   s = x + delta (when delta != NULL; s is also stored to sum_out [T, hidden]) else x
 *   out[t] = weight * (s[t] / sqrt(mean(s[t]^2) + eps))            out [T, hidden] contiguous
 * hidden must be a multiple of 8 (16-bit) / 4 (fp32) and at most 8192 / 4096.
 */
int mi355_rmsnorm(void* out, void* sum_out, const void* x, int64_t x_stride, const void* delta,
                  int64_t delta_stride, const void* weight, float eps, int32_t T, int32_t hidden,
                  int32_t dtype, void* stream);

/*
 * SwiGLU gating, out = silu(gate) * up elementwise over [T, width]
 * (FFN.forward, 05_mini_vllm/mini_vllm.py:490-499); out is contiguous.
 */
int mi355_silu_mul(void* out, const void* gate, int64_t gate_stride, const void* up, int64_t up_stride,
                   int32_t T, int32_t width, int32_t dtype, void* stream);

/*
 * Device-side inputs of a pure decode step (every sequence feeds one token), replacing the
 * host flattening of MiniVLLM.step (05_mini_vllm/mini_vllm.py:722-739) and
 * Scheduler._compute_slot_mapping (:209-219) for that case.  The host uploads one int32
 * array packed[3][rows] = (last token, position, row of the sequence in table_pool); for
 * every r the kernel writes
 *   token_ids[r] = token, positions[r] = max(position, 0)                         (int64)
 *   context_lens[r] = position + 1                                                (int32)
 *   block_tables[r][:] = table_pool[row][:]                                       (int32, [rows, width])
 *   slots[r] = table_pool[row][position / block_size] * block_size + position % block_size
 * A negative position marks a padding row: context 0 and slot -1 (no cache write).
 * A negative token entry -1-j stands for prev_ids[j], the id the sampler wrote for row j of the
 * previous step (the reference appends it on the host, :312-333, before it can schedule the next
 * step; reading it on the device lets the next step be issued before the host has seen it).
 * prev_ids is int32 [prev_rows] and may be NULL when prev_rows is 0.
 * table_pool is [pool_rows, width] int32.
 * block_tables may be NULL: the rows' tables are then not gathered (a step whose rows are the
 * tokens of prompt chunks, which share their sequence's table, gathers per sequence instead).
 */
int mi355_decode_step_inputs(int64_t* token_ids, int64_t* positions, int64_t* slots,
                             int32_t* context_lens, int32_t* block_tables, const int32_t* packed,
                             const int32_t* table_pool, const int32_t* prev_ids, int32_t prev_rows,
                             int32_t rows, int32_t pool_rows, int32_t width, int32_t block_size,
                             void* stream);

/*
 * Attention of several NEW query tokens per sequence over the paged cache, each token attending
 * to cache positions 0 .. its own position -- what a later chunk of a chunked prompt needs
 * (the reference's chunked prefill attends within the chunk only, 05_mini_vllm/mini_vllm.py:437-458;
 * this is the same causal attention with the earlier chunks read through the block table).  The
 * K/V rows of the new tokens must already be in the cache (mi355_kv_write_slots /
 * mi355_rope_kv_write).
 *   segment s (0 <= s < nseg): query tokens are rows seg_q_start[s] .. +seg_len[s]-1 of q / out,
 *   the first of them sits at position seg_pos0[s] of the sequence whose block table is
 *   block_tables[s][:]; query j attends to positions 0 .. seg_pos0[s] + j.
 *   q [T, Hq, D] (q_stride_tok elements between tokens), out [T, Hq, D] contiguous;
 *   max_q_len >= every seg_len, max_context >= every seg_pos0 + seg_len (host-side bounds).
 * bf16 / fp16 at head_dim 64 or 128 run the matrix-core kernels; fp32 and head dims 16 / 32 / 256
 * run the VALU kernel with eight consecutive query rows per wave.  Other head dims return
 * MI355_EUNSUPPORTED.
 */
int mi355_paged_prefill(void* out, const void* q, int64_t q_stride_tok,
                        const void* k_cache, const void* v_cache, const int32_t* block_tables,
                        const int32_t* seg_q_start, const int32_t* seg_len, const int32_t* seg_pos0,
                        int32_t nseg, int32_t max_q_len, int32_t Hq, int32_t Hkv, int32_t D,
                        int32_t block_size, int32_t max_blocks, int32_t num_blocks, int32_t max_context,
                        int64_t kv_stride_block, int64_t kv_stride_head, int64_t kv_stride_tok,
                        float scale, int32_t dtype, void* stream);

/*
 * The longest max_context mi355_paged_prefill serves for a dtype, head_dim, block size and longest
 * segment (max_q_len), so that no caller has to learn a limit from MI355_EUNSUPPORTED: INT32_MAX
 * where a flash-attention kernel takes the launch (16-bit head_dim 64 / 128 and fp32 head_dim
 * 32 / 64 / 128 from MI355_PAGED_PREFILL_MIN_Q = 48 query tokens on: they read the block table from
 * memory), otherwise the context whose block table one workgroup can stage in LDS (8190 blocks,
 * rounded down to the kernels' token quantum), and 0 for a dtype / head_dim no kernel serves.
 * Longer contexts go row by row through mi355_paged_decode, which splits them.  Host arithmetic
 * only (no device call); no counterpart in the reference, whose chunked prefill never reads the
 * cache (05_mini_vllm/mini_vllm.py:437-458).
 */
int32_t mi355_paged_prefill_max_context(int32_t dtype, int32_t D, int32_t block_size, int32_t max_q_len);

/*
 * Data-parallel exchange of the sampled token ids, one process per GPU, over RCCL (xGMI inside a
 * node).  No counterpart in the reference, which is single-process and single-device (its only
 * collective is a commented-out all-reduce, 03_moe/mini_moe.py:356-357): sequences are sharded
 * over ranks and each rank's ids are all-gathered so that every rank holds the whole batch's.
 * RCCL is bound at run time; without librccl.so.1 these return MI355_EUNSUPPORTED.
 *   mi355_dp_unique_id: rank 0 fills id_out[128]; the caller passes those bytes to every rank.
 *   mi355_dp_init: every rank, with its HIP device current, joins; *comm_out is the communicator.
 *   mi355_dp_allgather_i32: recv[r * count .. +count) = rank r's send[0 .. count), on `stream`
 *     (asynchronous; buffers are device memory and stay valid until the stream reaches the end).
 *   mi355_dp_destroy: releases the communicator.
 */
int mi355_dp_unique_id(void* id_out);
int mi355_dp_init(const void* id, int32_t nranks, int32_t rank, void** comm_out);
int mi355_dp_allgather_i32(void* comm, const int32_t* send, int32_t* recv, int64_t count, void* stream);
int mi355_dp_destroy(void* comm);

/*
 * The same exchange started by the GPU, so that the stream that samples pays nothing for it (no
 * counterpart in the reference).  An exchange owns a side stream; mi355_dp_exchange_launch puts on
 * it a one-wave "watcher" kernel that returns once every entry of `send` has been written, then
 * the all-gather, then a kernel that marks `send` as not written again.  The stream that produces
 * the ids carries no event record, no wait and no extra launch, its sampler is the plain one, and
 * ranks do not wait for each other on it.  (Measured on this chip: an event record + stream wait
 * per step costs the producing stream 3.9 us, 1.5 us without the system fence; a wait for a value
 * in signal memory, hipStreamWaitValue64, runs as a polling kernel of the runtime's own.)  The side
 * stream is created with the lowest priority the device offers: a second queue of the default
 * priority that is busy or waiting slows the workgroup dispatch of the producing stream (a
 * 2048-workgroup decode launch by 3 us); behind a low-priority queue it costs 0.3-0.5 us per step.
 * Token ids are non-negative; "not written" is -1.
 *   mi355_dp_exchange_create: with the communicator's device current.
 *   mi355_dp_exchange_arm: marks send[0 .. count) as not written (synchronous); once per send
 *     buffer, before the first launch that reads it.
 *   mi355_dp_exchange_launch: recv[r * count .. +count) = rank r's send[0 .. count) as soon as the
 *     producer's kernel has written every entry (the watcher polls device-coherent memory, where a
 *     kernel's stores are when it ends; a watcher gives up after 10 s and the exchange reports
 *     MI355_ECOLLECTIVE from then on).  `slot` in [0, 8) names the completion this call arms.  The
 *     producer may write `send` again once that completion is reached, not before.
 *   mi355_dp_exchange_done: 1 when the slot's last exchange has finished (or none was launched),
 *     0 while it is in flight.
 *   mi355_dp_exchange_wait: host != 0: the calling thread waits for the slot's exchange;
 *     host == 0: `stream` waits for it.
 *   mi355_dp_exchange_destroy: waits for the side stream, releases everything.  Before
 *     mi355_dp_destroy of the communicator.
 */
int mi355_dp_exchange_create(void* comm, void** exchange_out);
int mi355_dp_exchange_arm(void* exchange, int32_t* send, int64_t count);
int mi355_dp_exchange_launch(void* exchange, int32_t* send, int32_t* recv, int64_t count, int32_t slot);
int mi355_dp_exchange_launch_after(void* exchange, void* producer_stream, const int32_t* send,
                                   int32_t* recv, int64_t count, int32_t slot);
int mi355_dp_exchange_done(void* exchange, int32_t slot);
int mi355_dp_exchange_wait(void* exchange, int32_t slot, void* stream, int32_t host);
int mi355_dp_exchange_destroy(void* exchange);

/*
 * Debugging / measurement aid, not part of the replaced path (no counterpart in the reference).
 * The launch plans have a few integer knobs kept switchable so that two variants can be compared
 * inside one process: MI355_DECODE_GQA_MFMA, MI355_PAGED_TARGET_WGS, MI355_PAGED_HPW,
 * MI355_PAGED_PREFILL_MIN_Q, MI355_PREFILL_VALU, MI355_PREFILL_F32_TILED, MI355_PREFILL_ROW_GROUPS,
 * MI355_PREFILL_WAVES, MI355_MLA_MANY_HEADS, MI355_MLA_TARGET_WGS, MI355_MLA_ROW_GROUPS, MI355_MLA_WIDE, MI355_MLA_WIDE_SPLIT_TOKENS, MI355_MLA_WIDE_FUSED_MERGE, MI355_MLA_WIDE_SKEW_TOKENS, MI355_PREFILL_F32_MFMA,
 * MI355_DECODE_FUSED_MERGE, MI355_PAGED_BALANCE_WGS, MI355_DP_EXCHANGE_PROBE.  Each takes its default, or the
 * value of the environment variable of the same name, which the library reads ONCE per process;
 * afterwards only this call changes it (value INT32_MIN restores the process's start-up value).
 * Returns MI355_EINVAL for an unknown name.  A knob must not change between a workspace-size
 * query and the launch it sizes.
 */
int mi355_debug_set_knob(const char* name, int32_t value);

#ifdef __cplusplus
}
#endif
#endif /* MI355_PAGED_H */
