// Launch description shared by the paged-decode kernels (paged_decode.hip: VALU kernel, merge
// kernel, planning; paged_decode_gqa.hip: matrix-core kernel for grouped-query shapes).
#pragma once

#include "common.h"

namespace mi355 {

constexpr int kMaxBtEntries = 8192;  // block-table entries one workgroup may stage in LDS
constexpr float kNegBig = -1.0e30f;  // running-max start: finite so that exp2(m - m) never sees inf - inf

struct DecodeParams {
  void* out;
  const void* q;
  const void* kc;
  const void* vc;
  const int32_t* bt;
  const int32_t* ctx;
  const int32_t* cu;  // non-null selects causal mode over packed sequences (VALU kernel only)
  int nseq;
  int row_tiles;      // causal mode with row groups: tiles of consecutive query rows per sequence
  float* ws_acc;  // [B, Hq, nsplit, D]
  float* ws_ml;   // [B, Hq, nsplit, 2]  (max in log2 units, sum)
  int64_t q_stride_seq, s_blk, s_head, s_tok;
  int B, Hq, Hkv, gq;
  int block_size, bs_shift, max_blocks, num_blocks, max_ctx;
  int hpw, sub, hgroups, qchunks, tps, nsplit;
  int lds_bt_bytes;
  float scale_log2e;
  // Append mode of the matrix-core kernel (several query tokens per sequence, each attending to
  // the cache up to and including its own position): per segment the first row of q / out, the
  // number of query tokens and the position of the first one; n_qt query tiles per segment of
  // tq tokens each; a tile's 16 columns are tq tokens x (1 << hgp_shift) heads of the kv group.
  const int32_t* seg_q0;
  const int32_t* seg_len;
  const int32_t* seg_pos0;
  int n_qt, tq, hgp_shift;
  int max_q;   // host-side bound on seg_len the grid was sized for; device values are clamped to it
  // Token splits merged inside the VALU kernel (null: a second launch merges them): one 64-bit
  // arrival cell per (sequence, kv-head group, q-head chunk), holding (launch tag << 16) | arrivals.
  unsigned long long* counters = nullptr;
  unsigned long long epoch = 0;
  // Append mode of the VALU kernel: the K/V row of every sequence's newest token ([B, Hkv, D],
  // new_stride_seq elements between sequences) belongs at position context - 1 of its cache; the
  // wave whose token range holds that position attends to the row from LDS (lds_new_off: where
  // the waves' copies start in dynamic LDS) and stores it to the cache at its end.
  const void* k_new = nullptr;
  const void* v_new = nullptr;
  int64_t new_stride_seq = 0;
  int lds_new_off = 0;
  // ... with rotary embedding applied on the way (non-null tables): q and the new K row arrive
  // un-rotated and are rotated for position context - 1 (rotate-half form, tables [rope_len, D] in
  // the cache's dtype) before use; the rotated K row is what the cache receives.
  const void* rope_cos = nullptr;
  const void* rope_sin = nullptr;
  int rope_len = 0;
};

struct DecodePlan {
  int nw, hpw, sub, hgroups, G, qchunks, tps, nsplit;
  bool mfma;  // grouped-query shape served by the matrix-core kernel
  bool fused_merge = false;  // token splits are merged by the last workgroup of each group to finish
  size_t lds_bytes, lds_bt_bytes, ws_bytes;
};

// Matrix-core decode kernel for grouped-query attention (paged_decode_gqa.hip).
constexpr int kGqaWaves = 4;       // waves per workgroup
constexpr int kGqaTokens = 32;     // tokens per wave per pipeline step
constexpr int kGqaHeads = 16;      // query heads per tile (one MFMA N dimension)
bool decode_gqa_supported(int dtype, int D, int gq);
size_t decode_gqa_lds_bytes(int D, int sub, size_t lds_bt_bytes);
int decode_gqa_launch(const DecodeParams& p, const DecodePlan& pl, int D, int dtype, hipStream_t stream);
int append_gqa_launch(const DecodeParams& p, const DecodePlan& pl, int D, int dtype, hipStream_t stream);

}  // namespace mi355
