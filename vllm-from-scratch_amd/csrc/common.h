// Shared device/host helpers for the mi355_paged kernels (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mi355_paged.h"

namespace mi355 {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;

// Records `code` and a printf-style message for mi355_last_error(); returns code.
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Integer tuning knobs (launch-plan choices kept switchable for A/B measurements).  A knob's
// value is its default, overridden by the environment variable of the same name -- read ONCE
// per process, at the first use of any knob -- and after that only by mi355_debug_set_knob.
// Reading a knob on the launch path is one relaxed atomic load.
enum Knob {
  kKnobDecodeGqaMfma,     // MI355_DECODE_GQA_MFMA     1: grouped-query decode on the matrix cores
  kKnobPagedTargetWgs,    // MI355_PAGED_TARGET_WGS    workgroups a decode launch aims for
  kKnobPagedHpw,          // MI355_PAGED_HPW           0: planned; else kv heads per decode workgroup
  kKnobPagedPrefillMinQ,  // MI355_PAGED_PREFILL_MIN_Q chunk length from which the flash kernel serves paged chunks
  kKnobPrefillValu,       // MI355_PREFILL_VALU        1: packed prompts stay off the matrix cores
  kKnobPrefillF32Tiled,   // MI355_PREFILL_F32_TILED   1: fp32 prompts on the one-row-per-lane kernel
  kKnobPrefillRowGroups,  // MI355_PREFILL_ROW_GROUPS  1: VALU prompts in groups of eight query rows
  kKnobPrefillWaves,      // MI355_PREFILL_WAVES       0: planned; 4 or 8 waves per prompt workgroup
  kKnobMlaManyHeads,      // MI355_MLA_MANY_HEADS      1: 64 heads per MLA workgroup above 16 heads
  kKnobMlaTargetWgs,      // MI355_MLA_TARGET_WGS      0: planned; else workgroups an MLA launch aims for
  kKnobPrefillF32Mfma,    // MI355_PREFILL_F32_MFMA    1: fp32 prompts (head_dim 32 / 64 / 128) on the fp32 matrix instruction
  kKnobDecodeFusedMerge,  // MI355_DECODE_FUSED_MERGE  most token splits the VALU decode kernel merges in its own launch (0: never)
  kKnobPagedBalanceWgs,   // MI355_PAGED_BALANCE_WGS   workgroups a decode launch spreads its kv heads over for ragged batches (0: off)
  kKnobDpExchangeProbe,   // MI355_DP_EXCHANGE_PROBE   measurement only, bit mask: 1 no collective, 2 no completion event, 4 no watcher
  kKnobMlaRowGroups,      // MI355_MLA_ROW_GROUPS      1: paged rank-512 MLA tiles staged as 8 consecutive rows per wave
  kKnobMlaWide,           // MI355_MLA_WIDE            fewest heads served by the 128-heads-per-workgroup MLA kernel (rank 512, rope 64); 0: off
  kKnobMlaWideSplitTokens,  // MI355_MLA_WIDE_SPLIT_TOKENS  fewest tokens per workgroup at which a launch with token splits takes the 128-head kernel (0: planned)
  kKnobMlaWideFusedMerge,   // MI355_MLA_WIDE_FUSED_MERGE   1: two token splits of the 128-head kernel are merged by the later workgroup of the pair (0: by the merge kernel)
  kKnobMlaWideSkewTokens,   // MI355_MLA_WIDE_SKEW_TOKENS   tokens split 0 of such a pair is given less than half (so that only one record is written); 0: even splits
  kKnobCount
};
int knob(Knob k);
// Returns 0 or records + returns the pending hipError_t of the last launch.
int check_launch(const char* what);

// Matrix-core causal prefill (prefill_mfma.hip): covers 16-bit dtypes at head_dim 64 / 128.
bool prefill_mfma_supported(int dtype, int D);
int prefill_mfma_launch(void* out, const void* q, const void* k, const void* v, int64_t q_stride_tok,
                        int64_t kv_stride_tok, const int32_t* cu_seqlens, int nseq, int total_tokens,
                        int max_seqlen, int Hq, int Hkv, int D, float scale, int dtype, hipStream_t stream);
// The same kernel over block caches: segments of new query tokens behind a cached context.
int prefill_mfma_paged_launch(void* out, const void* q, int64_t q_stride_tok, const void* k_cache,
                              const void* v_cache, const int32_t* block_tables, const int32_t* seg_q0,
                              const int32_t* seg_len, const int32_t* seg_pos0, int nseg, int max_q_len, int Hq,
                              int Hkv, int D, int block_size, int max_blocks, int num_blocks, int max_context,
                              int64_t s_blk, int64_t s_head, int64_t s_tok, float scale, int dtype,
                              hipStream_t stream);

// fp32 latent-space MLA attention on the fp32 matrix instruction (mla_decode_f32.hip).
bool mla_f32_supported(int R, int P);
int mla_f32_launch(float* out, const float* q, int64_t q_stride_row, const float* c_kv, int64_t ckv_stride_seq,
                   int64_t ckv_stride_tok, const float* k_rope, int64_t krope_stride_seq, int64_t krope_stride_tok,
                   const float* cos_table, const float* sin_table, int table_len, const int32_t* context_lens,
                   const int32_t* row_seq, int rows, int H, int R, int P, int max_context, float scale,
                   void* workspace, size_t workspace_bytes, hipStream_t stream,
                   // block-table addressing (null table: dense per-sequence rows)
                   const int32_t* block_tables = nullptr, int block_size = 0, int max_blocks = 0, int num_blocks = 0,
                   int64_t ckv_stride_blk = 0, int64_t krope_stride_blk = 0);
void mla_f32_plan(int rows, int H, int max_context, int* nsplit, int* tps, size_t* ws_bytes, int R);
// 128 heads per workgroup on the 32 x 32 x 16 matrix instruction (mla_decode_wide.hip): kv_lora_rank 512,
// rope 64, more than 64 heads, 16-bit dtypes.  `params` is the MlaParams block of mla_common.h.
bool mla_wide_supported(int H, int R, int P, int dtype);
int mla_wide_max_paged_tokens(int block_size);
size_t mla_wide_stamp_bytes();
int mla_wide_launch(const void* params, int dtype, bool paged, hipStream_t stream);
// A 48-bit tag that differs for every launch of this process (paged_decode.hip): arrival cells in a
// caller's workspace count only while they carry the launch's tag, so the workspace needs no clearing.
unsigned long long next_launch_tag();
// Combines token-split partials [rows, H, nsplit, R] + (max, sum) records into out (mla_decode.hip).
int mla_merge_launch(float* out, float* ws_acc, float* ws_ml, const int32_t* ctx, int rows, int H, int R,
                     int max_ctx, int nsplit, int tps, hipStream_t stream);

// fp32 causal prefill on the matrix cores (prefill_mfma_f32.hip): head_dim 32 / 64 / 128, packed and paged.
bool prefill_mfma_f32_supported(int dtype, int D);
int prefill_mfma_f32_launch(void* out, const void* q, const void* k, const void* v, int64_t q_stride_tok,
                            int64_t kv_stride_tok, const int32_t* cu_seqlens, int nseq, int max_seqlen, int Hq,
                            int Hkv, int D, float scale, hipStream_t stream);
int prefill_mfma_f32_paged_launch(void* out, const void* q, int64_t q_stride_tok, const void* k_cache,
                                  const void* v_cache, const int32_t* block_tables, const int32_t* seg_q0,
                                  const int32_t* seg_len, const int32_t* seg_pos0, int nseg, int max_q_len, int Hq,
                                  int Hkv, int D, int block_size, int max_blocks, int num_blocks, int max_context,
                                  int64_t s_blk, int64_t s_head, int64_t s_tok, float scale, hipStream_t stream);

// fp32 causal prefill with one query row per lane (prefill_f32.hip): head_dim 32 / 64 / 128.
bool prefill_f32_supported(int dtype, int D);
int prefill_f32_launch(void* out, const void* q, const void* k, const void* v, int64_t q_stride_tok,
                       int64_t kv_stride_tok, const int32_t* cu_seqlens, int nseq, int max_seqlen, int Hq, int Hkv,
                       int D, float scale, hipStream_t stream);

static inline int ilog2_exact(int v) {  // log2(v) if v is a power of two, else -1
  if (v <= 0 || (v & (v - 1))) return -1;
  int s = 0;
  while ((1 << s) < v) ++s;
  return s;
}
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Bit reinterpretation helpers that take their argument BY VALUE.  Applying
// __builtin_bit_cast directly to an element of an ext_vector lvalue (v[i]) reads
// element 0 for every i with this compiler; going through a by-value scalar
// parameter avoids that.
__device__ __forceinline__ float u2f(uint32_t x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ uint32_t f2u(float x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ bf16x2_t u2bf2(uint32_t x) { return __builtin_bit_cast(bf16x2_t, x); }
__device__ __forceinline__ f16x2_t u2h2(uint32_t x) { return __builtin_bit_cast(f16x2_t, x); }

// ---------------------------------------------------------------------------
// dtype tags.  A lane always moves 16 bytes: kEPL elements.
// ---------------------------------------------------------------------------
struct F32 {
  static constexpr int kBytes = 4, kEPL = 4, kCode = MI355_F32;
  typedef float elem_t;
  struct QFrag { float f[4]; };
  static __device__ __forceinline__ QFrag make_q(u32x4 raw) {
    QFrag q;
#pragma unroll
    for (int i = 0; i < 4; ++i) q.f[i] = u2f(raw[i]);
    return q;
  }
  static __device__ __forceinline__ float dot(const QFrag& q, u32x4 k) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s = __builtin_fmaf(q.f[i], u2f(k[i]), s);
    return s;
  }
  static __device__ __forceinline__ void unpack(u32x4 raw, float (&f)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = u2f(raw[i]);
  }
  static __device__ __forceinline__ u32x4 pack(const float (&f)[4]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = f2u(f[i]);
    return r;
  }
  static __device__ __forceinline__ float to_float(elem_t v) { return v; }
};

struct BF16 {
  static constexpr int kBytes = 2, kEPL = 8, kCode = MI355_BF16;
  typedef uint16_t elem_t;
  struct QFrag { uint32_t w[4]; };
  static __device__ __forceinline__ QFrag make_q(u32x4 raw) {
    QFrag q;
#pragma unroll
    for (int i = 0; i < 4; ++i) q.w[i] = raw[i];
    return q;
  }
  // bf16 x bf16 products are exact in fp32; v_dot2c_f32_bf16 adds two of them
  // into the fp32 accumulator per instruction.
  static __device__ __forceinline__ float dot(const QFrag& q, u32x4 k) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      s = __builtin_amdgcn_fdot2_f32_bf16(u2bf2(k[i]), u2bf2(q.w[i]), s, false);
    return s;
  }
  static __device__ __forceinline__ void unpack(u32x4 raw, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = u2f(raw[i] << 16);
      f[2 * i + 1] = u2f(raw[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ uint32_t round1(float x) {  // RNE, NaN stays NaN
    __bf16 b = (__bf16)x;
    return (uint32_t)__builtin_bit_cast(uint16_t, b);
  }
  static __device__ __forceinline__ u32x4 pack(const float (&f)[8]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = round1(f[2 * i]) | (round1(f[2 * i + 1]) << 16);
    return r;
  }
  static __device__ __forceinline__ float to_float(elem_t v) {
    return __builtin_bit_cast(float, (uint32_t)v << 16);
  }
};

struct F16 {
  static constexpr int kBytes = 2, kEPL = 8, kCode = MI355_F16;
  typedef uint16_t elem_t;
  struct QFrag { uint32_t w[4]; };
  static __device__ __forceinline__ QFrag make_q(u32x4 raw) {
    QFrag q;
#pragma unroll
    for (int i = 0; i < 4; ++i) q.w[i] = raw[i];
    return q;
  }
  static __device__ __forceinline__ float dot(const QFrag& q, u32x4 k) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      s = __builtin_amdgcn_fdot2(u2h2(k[i]), u2h2(q.w[i]), s, false);
    return s;
  }
  static __device__ __forceinline__ void unpack(u32x4 raw, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f16x2_t h = u2h2(raw[i]);
      f[2 * i] = (float)h[0];
      f[2 * i + 1] = (float)h[1];
    }
  }
  static __device__ __forceinline__ u32x4 pack(const float (&f)[8]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f16x2_t h;
      h[0] = (_Float16)f[2 * i];
      h[1] = (_Float16)f[2 * i + 1];
      r[i] = __builtin_bit_cast(uint32_t, h);
    }
    return r;
  }
  static __device__ __forceinline__ float to_float(elem_t v) {
    return (float)__builtin_bit_cast(_Float16, v);
  }
};

// ---------------------------------------------------------------------------
// lane-group reductions.  Groups are aligned runs of N lanes (N a power of 2).
// Steps up to 16 lanes ride on DPP modifiers of the add/max itself; wider
// steps go through ds_bpermute.
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}

// Sum over aligned groups of N lanes; every lane of a group ends with the total.
template <int N>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (N >= 2) v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
  if constexpr (N >= 4) v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
  if constexpr (N >= 8) v += dpp_f<0x141>(v);   // row_half_mirror
  if constexpr (N >= 16) v += dpp_f<0x140>(v);  // row_mirror
  if constexpr (N >= 32) v += __shfl_xor(v, 16);
  if constexpr (N >= 64) v += __shfl_xor(v, 32);
  return v;
}

// Butterfly over the lane bits ABOVE a group of N lanes: combines lanes
// l, l^N, l^2N, ... so that lanes with equal (l % N) share the result.
template <int N, class Op>
__device__ __forceinline__ float cross_group(float v, Op op) {
#pragma unroll
  for (int off = N; off < kWave; off <<= 1) v = op(v, __shfl_xor(v, off));
  return v;
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Requests every 64-byte line of a kernel's argument block at once: the fields a prologue reads
// one after the other (each read a dependent trip through the scalar cache when its line is cold)
// then hit the cache.  Returns zero through an opaque register; adding it to a value the kernel
// needs early (the thread index) keeps the requests at the top of the kernel.  The asm statement
// is not volatile and touches no memory, so uniform loads after it stay scalar loads.
// OFFSET: where the block starts within the kernel's arguments (scalar arguments before it).
template <int BYTES, int OFFSET = 0>
__device__ __forceinline__ int warm_kernel_arguments() {
  typedef const __attribute__((address_space(4))) uint32_t* karg_t;
  karg_t ka = (karg_t)__builtin_amdgcn_kernarg_segment_ptr() + OFFSET / 4;
  constexpr int kLines = (BYTES + 63) / 64;
  static_assert(kLines <= 5, "argument block longer than 320 bytes");
  uint32_t w[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int line = i < kLines ? i : kLines - 1;
    w[i] = ka[(line * 64 < BYTES - 4 ? line * 64 : BYTES - 4) / 4];
  }
  int zero;
  if constexpr (kLines <= 4) asm("s_mov_b32 %0, 0" : "=s"(zero) : "s"(w[0]), "s"(w[1]), "s"(w[2]), "s"(w[3]));
  else asm("s_mov_b32 %0, 0" : "=s"(zero) : "s"(w[0]), "s"(w[1]), "s"(w[2]), "s"(w[3]), "s"(w[4]));
  return zero;
}

// Block index -> (head tile, token split, row).  The head tiles of one (row, split) read the same
// cached rows.  Workgroups b and b + 8 share an XCD and its L2, so within every run of 8 * n_ht
// consecutive blocks the head tile is taken from b / 8 and the (row, split) from b % 8: the tiles of
// a (row, split) sit on one XCD and the second reader finds the rows in that L2 (taken as b % n_ht,
// they would sit on neighbouring XCDs and each fetch the rows from memory).  The blocks past the
// last full run keep the plain order.
__device__ __forceinline__ void split_block_index(int b, int n_ht, int nsplit, int rows, int& ht, int& sp, int& row) {
  const int items = rows * nsplit;                 // (row, split) pairs
  const int run = 8 * n_ht, full = items / 8;      // `full` runs of 8 pairs x n_ht tiles
  int item;
  if (n_ht > 1 && b < full * run) {
    const int r = b / run, i = b - r * run;
    item = r * 8 + (i & 7);
    ht = i >> 3;
  } else {
    const int t = n_ht > 1 ? b - full * run : b;
    item = (n_ht > 1 ? full * 8 : 0) + t / n_ht;
    ht = t % n_ht;
  }
  sp = item % nsplit;
  row = item / nsplit;
}

// Waits until every store this wave has issued has been acknowledged by the memory system
// (s_waitcnt vmcnt(0); stores count in vmcnt until they are).  After it, a value the wave wrote
// with an agent-scope atomic store (written through the XCD's L2) is in device-coherent memory.
// The immediate is the gfx9-family encoding of vmcnt(0) with the other counters left alone.
__device__ __forceinline__ void wait_stores_acknowledged() {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "s_waitcnt immediate below encodes vmcnt(0) for gfx942 / gfx950 only"
#endif
  __builtin_amdgcn_s_waitcnt(0x0F70);
}

template <bool NT>
__device__ __forceinline__ u32x4 load16(const void* p) {
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  else return *reinterpret_cast<const u32x4*>(p);
}

}  // namespace mi355
