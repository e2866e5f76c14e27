// Declarations shared by the 16-bit latent-attention (MLA) kernels: the launch parameter block,
// the operand vector types and the cache addressing modes.
#pragma once

#include "common.h"

namespace mi355 {
namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kTok = 16;        // tokens per step (one-wave kernel)
constexpr int kTok2 = 32;       // tokens per step (four-wave kernel)
constexpr int kHeadTile = 16;   // heads per workgroup

struct MlaParams {
  float* out;            // [rows, H, R] fp32
  const void* q;         // [rows, H, R + P]
  const void* ckv;       // c_kv rows
  const void* krope;     // k_rope rows (un-rotated)
  const float* cos_t;    // [table_len, P]
  const float* sin_t;
  const int32_t* ctx;    // [rows]
  const int32_t* row_seq;  // [rows] cache sequence of each query row, or null (= row)
  float* ws_acc;         // [rows, H, nsplit, R]
  float* ws_ml;          // [rows, H, nsplit, 2]
  int64_t q_stride_row, ckv_stride_seq, ckv_stride_tok, krope_stride_seq, krope_stride_tok;
  int rows, H, P, table_len, max_ctx, n_ht, nsplit, tps;
  float scale_log2e;
  // Paged caches (the PAGED instantiations): token t of sequence s sits in block bt[s][t >> bs_shift]
  // (clamped to the pool) at row t & bs_mask; *_stride_seq are then unused.
  const int32_t* bt = nullptr;   // [sequences, max_blocks]
  int64_t ckv_stride_blk = 0, krope_stride_blk = 0;
  int bs_shift = 0, bs_mask = 0, max_blocks = 0, num_blocks = 0;
  // Two token splits merged inside the launch (128-heads-per-workgroup kernel): an arrival cell per
  // (row, head tile) behind the records, counting while it holds (epoch << 16); null: merge kernel.
  unsigned long long* counters = nullptr;
  unsigned long long epoch = 0;
  unsigned long long* stamps = nullptr;   // timing side builds only: 65 x 8 clock stamps at the end of the workspace
};

// Addressing modes of the kernels below (template parameter PAGED).
constexpr int kDense = 0;        // rows of one sequence at a fixed stride from the sequence's base
constexpr int kPaged = 1;        // rows in blocks named by a block table, any power-of-two block size
constexpr int kPagedGroups = 2;  // ... blocks of >= 8 tokens, kv_lora_rank 512, rope slot 64: see GroupAt
constexpr bool is_paged(int mode) { return mode == kPaged || mode == kPagedGroups; }
constexpr bool is_grouped(int mode) { return mode == kPagedGroups; }

}  // namespace
}  // namespace mi355
