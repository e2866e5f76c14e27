// Latent-space MLA attention for many heads: 128 heads per workgroup on v_mfma_f32_32x32x16.
//
// With the up-projections absorbed (see mla_decode.hip) every cached row [c_kv | rope(k_rope)] of
// kv_lora_rank + rope = 512 + 64 dims serves all H heads.  At 128 heads the work per cached byte is
// ~220 flop, so the kernel is bound by the matrix pipe and by what feeds it: the 16-heads-per-wave
// kernel reads every 36 KiB tile from LDS once per 16 heads (an LDS byte per 16 flop-pairs), which
// keeps the matrix pipe a quarter busy.  Here
//
//   workgroup = one query row (x token split) x 128 heads, four waves of 32 heads each;
//   a wave holds its heads' queries (36 fragments = 144 registers) and latent outputs
//   (32 heads x 512 dims fp32 = 256 accumulator registers) for the whole launch: one wave per SIMD;
//   S^T[32 tokens][32 heads] = K . Q^T is 36 matrix instructions whose A operands are 16-byte LDS
//   reads, O^T[512 dims][32 heads] += V^T . P^T is 32 instructions whose A operands are transposing
//   reads of the same rows: an LDS operand feeds twice the flops it does in the 16 x 16 form.
//
// With every register taken nothing can be staged through registers, so the cached rows travel
// global -> LDS directly (global_load_lds_dwordx4), two 32-token tiles ahead, into three buffers;
// the cos / sin rows of the tile's positions take the same road, which keeps ordinary loads (whose
// waits would drain the prefetch) out of the loop.  A wave rotates the rope part of the eight rows
// it fetched itself, behind its own wait and ahead of the tile's one barrier.
#include <type_traits>

#include "common.h"
#include "mla_common.h"

namespace mi355 {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <class Tag>
struct Mma32;
template <>
struct Mma32<BF16> {
  static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                   c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));   // one v_cvt_pk_bf16_f32
  }
};
template <>
struct Mma32<F16> {
  static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c,
                                                  0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    f16x2_t h;
    h[0] = (_Float16)lo;
    h[1] = (_Float16)hi;
    return __builtin_bit_cast(uint32_t, h);
  }
};

// The score product's 32 x 32 accumulator in VGPRs, as asm.  In a kernel that uses AccVGPRs at all
// this compiler gives EVERY matrix instruction an AccVGPR destination; the 256 of them hold O^T, so
// a builtin here would spill sixteen of those around every tile.  The compiler does not see these
// as matrix instructions: the last one of a chain carries the wait states a VALU read of the
// result needs (16 passes + 2), and dependent steps of the chain reuse the same registers, which
// the matrix pipe forwards without wait states.
template <class Tag, bool FIRST, bool LAST>
__device__ __forceinline__ void score_mma(f32x16& acc, u32x4 a, u32x4 b) {
  if constexpr (std::is_same<Tag, BF16>::value) {
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
    else if constexpr (LAST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+v"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  } else {
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
    else if constexpr (LAST) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+v"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  }
}

#ifndef WIDE_KO
#define WIDE_KO 0   // timing-only side builds: bit mask of parts left out (results are wrong)
#endif
#ifndef WIDE_KA
#define WIDE_KA 4
#endif
#ifndef WIDE_KAV
#define WIDE_KAV 4
#endif
constexpr int kR = 512, kP = 64;          // the one shape this kernel serves
constexpr int kTile = 32;                 // tokens per tile
constexpr int kKS = (kR + kP) / 16;       // k-steps of the score product (36)
constexpr int kDB = kR / 32;              // 32-dim blocks of the latent output (16)
// LDS image of a tile.  c_kv rows at a pitch of 1088 B (= 16 banks mod 64: the four rows a
// transposing read serves at a time tile the 64 banks) with rows 4i .. 4i+3 shifted by (i mod 4) x 16 B
// (the sixteen rows a 16-byte read serves at a time then start on sixteen different 4-bank groups).
// The rope parts sit apart, 128 B per row in 16-byte chunks XOR-ed with (row / 2) mod 8, because one
// direct-to-LDS load writes 1 KiB contiguously: eight rows' rope parts, or one row's c_kv.
constexpr int kPitch = 1088;
constexpr int kRopeOff = kTile * kPitch + 64;         // 34880
constexpr int kCosOff = kRopeOff + kTile * 128;
constexpr int kSinOff = kCosOff + kTile * 128;
constexpr int kBufBytes = kSinOff + kTile * 128;      // 47168
constexpr int kBuffers = 3;
constexpr int kIdsOff = kBuffers * kBufBytes;         // block ids of the workgroup's token range (paged caches)
constexpr int kIdsMax = 4096;
constexpr int kLoadsPerTile = 8 + 3;                  // direct-to-LDS loads a wave issues per tile

__device__ __forceinline__ constexpr int row_off(int r) { return r * kPitch + ((r >> 2) & 3) * 16; }

typedef __attribute__((address_space(3))) unsigned char lds_byte_t;

// Direct-to-LDS loads: 64 lanes x 16 bytes from global memory to 1 KiB of LDS at byte address `dst`
// (wave-uniform).  Written as asm: given the builtin, the compiler waits for EVERY such load in
// flight (vmcnt(0)) before the next LDS read it cannot prove independent, which would drain the two
// tiles of prefetch on every tile.  The compiler therefore does not count these loads either; the
// kernel waits for them itself (s_waitcnt vmcnt(N) below).  M0 carries the LDS address and is put
// back, since the compiler does not expect a statement to change it.
//   to_lds16_rows: every lane reads base + voff (base wave-uniform, in SGPRs)
//   to_lds16_lanes: every lane reads its own 64-bit address
__device__ __forceinline__ void to_lds16_rows(const void* base, uint32_t voff, uint32_t dst) {
  if (WIDE_KO & 16) return;
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(base), "s"(dst)
               : "memory");
}
__device__ __forceinline__ void to_lds16_lanes(const void* src, uint32_t dst) {
  if (WIDE_KO & 16) return;
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(dst)
               : "memory");
}

__device__ __forceinline__ void pair_values(float x, float& lo, float& hi) {
  float a = x, b = x;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  lo = a;
  hi = b;
}

// The latent output's accumulators, 16 tuples of 16 = all 256 AccVGPRs of the wave, are only ever
// named by asm statements with "a" (AccVGPR) operands, so the register allocator has no choice about
// where they live.  (Given builtins it treats accumulators as movable between the two register
// files, parks some in VGPRs, and spills the query fragments to make room.)
template <class Tag>
__device__ __forceinline__ void out_mma(f32x16& acc, u32x4 a, u32x4 b) {
  if constexpr (std::is_same<Tag, BF16>::value) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// acc *= f: sixteen copies out, sixteen multiplies, sixteen copies back, fenced by two empty
// statements so that the copies stay inside the (rarely taken) branch that holds them and one
// tuple is done before the next begins.
__device__ __forceinline__ void scale_acc(f32x16& acc, float f) {
  asm volatile("" : "+a"(acc));
  acc *= f;
  asm volatile("" : "+a"(acc));
}

template <class Tag, int PAGED>
__global__ void __launch_bounds__(256) mla_decode_kernel_wide(const MlaParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_generic[];
  lds_byte_t* const lds = (lds_byte_t*)lds_generic;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  int ht, sp, row;
  split_block_index(blockIdx.x, p.n_ht, p.nsplit, p.rows, ht, sp, row);
  const int seq = p.row_seq ? p.row_seq[row] : row;
  const int ctx = max(0, min(p.ctx[row], p.max_ctx));
  const int t_begin = sp * p.tps;
  if (p.nsplit > 1 && t_begin >= ctx) return;
  const int t_end = min(ctx, t_begin + p.tps);
  const int head = ht * 128 + wave * 32 + c;          // this lane's head: a column of S^T and of O^T
  const bool head_ok = head < p.H;
  const bool wave_live = ht * 128 + wave * 32 < p.H;  // else the wave only helps staging
  const int ntiles = t_end > t_begin ? (t_end - t_begin + kTile - 1) / kTile : 0;

  // Query fragments (B operands): lane (c, h) holds Q[head c][16 ks + 8 h .. + 8).
  u32x4 qf[kKS];
  {
    const char* qrow = static_cast<const char*>(p.q) + ((int64_t)row * p.q_stride_row + (int64_t)min(head, p.H - 1) * (kR + kP)) * 2;   // heads past H: a copy of the last one, never stored
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) qf[ks] = *reinterpret_cast<const u32x4*>(qrow + (16 * ks + 8 * h) * 2);
    // Have them arrive here: left to their first use inside the loop, the compiler's waits for them
    // (it counts only its own loads) would sit among the direct-to-LDS loads and drain those.
#pragma unroll
    for (int ks = 0; ks < kKS; ++ks) asm volatile("" ::"v"(qf[ks]));
  }
  f32x16 o[kDB];
#pragma unroll
  for (int i = 0; i < kDB; ++i) o[i] = f32x16(0.f);
  float m_run = -1.0e30f, l_run = 0.f;   // l_run: this lane's half of the head's sum (halves meet at the end)

  const char* ckv = static_cast<const char*>(p.ckv) + (is_paged(PAGED) ? 0 : (int64_t)seq * p.ckv_stride_seq * 2);
  const char* krope = static_cast<const char*>(p.krope) + (is_paged(PAGED) ? 0 : (int64_t)seq * p.krope_stride_seq * 2);
  const int cstride2 = (int)p.ckv_stride_tok * 2, rstride2 = (int)p.krope_stride_tok * 2;   // bytes; host checks the range

  // Paged caches: the block ids of this workgroup's token range, clamped into the pool, are staged
  // in LDS once (ordinary loads, before the first direct-to-LDS load is in flight).
  if constexpr (is_paged(PAGED)) {
    volatile __attribute__((address_space(3))) int* ids = (volatile __attribute__((address_space(3))) int*)(lds + kIdsOff);
    const int32_t* bt_row = p.bt + (int64_t)seq * p.max_blocks;
    const int b0 = t_begin >> p.bs_shift, b1 = (max(t_end, t_begin + 1) - 1) >> p.bs_shift;
    for (int b = b0 + tid; b <= b1; b += 256) ids[b - b0] = min(max(bt_row[min(b, p.max_blocks - 1)], 0), p.num_blocks - 1);
    __syncthreads();
  }
  const int blk0 = t_begin >> p.bs_shift;

  // One tile's loads.  c_kv: wave w fetches rows w, w + 4, ... (one row = one 64-lane load); rope
  // parts and the cos / sin rows of the same tokens: wave w fetches rows 8w .. 8w + 7, eight lanes a
  // row.  Rows past the end of the range re-read its last row (their scores are masked).
  const int rr = 8 * wave + (lane >> 3);             // the rope row this lane fetches and rotates
  const int rslot = lane & 7, rswz = (rr >> 1) & 7;
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;   // LDS byte address of the image
  // piece i of a tile's loads: 0 .. 7 the wave's c_kv rows, 8 its rope rows, 9 / 10 their cos / sin rows
  auto issue_piece = [&](int i, int t0, int bufoff) __attribute__((always_inline)) {
    const int last = t_end - 1;
    if (i < 8) {
      const int r = wave + 4 * i;
      const int tok = min(t0 + r, last);                 // wave-uniform
      const char* src;
      if constexpr (is_paged(PAGED)) {
        const int id = *(const __attribute__((address_space(3))) int*)(lds + kIdsOff + 4 * ((tok >> p.bs_shift) - blk0));
        src = ckv + ((int64_t)__builtin_amdgcn_readfirstlane(id) * p.ckv_stride_blk * 2 + (int64_t)(tok & p.bs_mask) * cstride2);
      } else {
        src = ckv + (uint32_t)(tok * cstride2);
      }
      to_lds16_rows(src, lane * 16, __builtin_amdgcn_readfirstlane(lds0 + bufoff + row_off(r)));
      return;
    }
    const int tokr = min(t0 + rr, last);
    if (i == 8) {
      const char* rsrc;
      if constexpr (is_paged(PAGED)) {
        const int id = *(const __attribute__((address_space(3))) int*)(lds + kIdsOff + 4 * ((tokr >> p.bs_shift) - blk0));
        rsrc = krope + ((int64_t)id * p.krope_stride_blk * 2 + (int64_t)(tokr & p.bs_mask) * rstride2);
      } else {
        rsrc = krope + (uint32_t)(tokr * rstride2);
      }
      to_lds16_lanes(rsrc + ((rslot ^ rswz) << 4), __builtin_amdgcn_readfirstlane(lds0 + bufoff + kRopeOff + wave * 1024));
      return;
    }
    const int pos = min(tokr, p.table_len - 1);
    if (i == 9) to_lds16_lanes(p.cos_t + (int64_t)pos * kP + 4 * rslot, __builtin_amdgcn_readfirstlane(lds0 + bufoff + kCosOff + wave * 1024));
    else to_lds16_lanes(p.sin_t + (int64_t)pos * kP + 4 * rslot, __builtin_amdgcn_readfirstlane(lds0 + bufoff + kSinOff + wave * 1024));
  };
  auto issue_tile = [&](int t0, int bufoff) {
#pragma unroll
    for (int i = 0; i < kLoadsPerTile; ++i) issue_piece(i, t0, bufoff);
  };

  // Rotation of this lane's four (lo, hi) pairs of rope row rr: dims 4 rslot .. + 4 and their
  // partners 32 dims on, with cos / sin columns 4 rslot .. + 4 of the row's position.
  auto rotate = [&](int bufoff) {
    lds_byte_t* lo_p = lds + bufoff + kRopeOff + rr * 128 + (((rslot >> 1) ^ rswz) << 4) + (rslot & 1) * 8;
    lds_byte_t* hi_p = lds + bufoff + kRopeOff + rr * 128 + (((4 + (rslot >> 1)) ^ rswz) << 4) + (rslot & 1) * 8;
    const f32x4 cv = *(const __attribute__((address_space(3))) f32x4*)(lds + bufoff + kCosOff + rr * 128 + rslot * 16);
    const f32x4 sv = *(const __attribute__((address_space(3))) f32x4*)(lds + bufoff + kSinOff + rr * 128 + rslot * 16);
    const u32x2 lo2 = *(const __attribute__((address_space(3))) u32x2*)lo_p;
    const u32x2 hi2 = *(const __attribute__((address_space(3))) u32x2*)hi_p;
    float lo[4], hi[4], nlo[4], nhi[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      lo[2 * k] = Tag::to_float((typename Tag::elem_t)(lo2[k] & 0xffffu));
      lo[2 * k + 1] = Tag::to_float((typename Tag::elem_t)(lo2[k] >> 16));
      hi[2 * k] = Tag::to_float((typename Tag::elem_t)(hi2[k] & 0xffffu));
      hi[2 * k + 1] = Tag::to_float((typename Tag::elem_t)(hi2[k] >> 16));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      nlo[k] = lo[k] * cv[k] - hi[k] * sv[k];
      nhi[k] = hi[k] * cv[k] + lo[k] * sv[k];
    }
    *(__attribute__((address_space(3))) u32x2*)lo_p = u32x2{Mma32<Tag>::pack2(nlo[0], nlo[1]), Mma32<Tag>::pack2(nlo[2], nlo[3])};
    *(__attribute__((address_space(3))) u32x2*)hi_p = u32x2{Mma32<Tag>::pack2(nhi[0], nhi[1]), Mma32<Tag>::pack2(nhi[2], nhi[3])};
  };

  // Per-lane LDS offsets of the operands (buffer offset added per tile).
  const int k_lane = row_off(c) + h * 16;                                  // K operand, latent k-steps: + 32 ks
  const int kr_lane = kRopeOff + c * 128, kr_swz = (c >> 1) & 7;           // rope k-steps: chunk (2 (ks - 32) + h) ^ swz
  const int grp = (lane >> 4) & 1, qi = (lane & 15) >> 2, pi = lane & 3;
  const int v_lane = (4 * h + qi) * kPitch + h * 16 + (16 * grp + 4 * pi) * 2;   // V^T operand

  // One tile of a live wave, in the order the instructions should issue (the asm statements keep
  // their order): the score product with the loads of tile t + 2 dealt out between its matrix
  // instructions; maximum and the first half of the weights; the first half of the value product
  // with the second half of the weights in its shadow; the second half of the value product with,
  // in its shadow, the wait for this wave's loads of tile t + 1 and the rotation of their rope rows.
  auto tile_step = [&](int t0, int bufoff, int next_t0, int next2_t0, int bufoff_next, int bufoff_next2) {
    // ---- S^T = K . Q^T
    f32x16 st;
    {
      constexpr int kAhead = WIDE_KA;
      auto a_at = [&](int n) __attribute__((always_inline)) {
        if (WIDE_KO & 2) return u32x4{(uint32_t)lane, 0u, 0u, 0u};
        if (n < kR / 16) return *(const __attribute__((address_space(3))) u32x4*)(lds + bufoff + k_lane + 32 * n);
        return *(const __attribute__((address_space(3))) u32x4*)(lds + bufoff + kr_lane + (((2 * (n - kR / 16) + h) ^ kr_swz) << 4));
      };
      u32x4 abuf[kAhead];
#pragma unroll
      for (int n = 0; n < kAhead; ++n) abuf[n] = a_at(n);
#pragma unroll
      for (int n = 0; n < kKS; ++n) {
        const u32x4 a = abuf[n % kAhead];
        if (n + kAhead < kKS) abuf[n % kAhead] = a_at(n + kAhead);
        if (WIDE_KO & 1) {
          if (n == 0) st = f32x16(0.f);
          asm volatile("" ::"v"(a));
        } else if (n == 0) score_mma<Tag, true, false>(st, a, qf[n]);
        else if (n == kKS - 1) score_mma<Tag, false, true>(st, a, qf[n]);
        else score_mma<Tag, false, false>(st, a, qf[n]);
        if (n % 3 == 1 && n / 3 < kLoadsPerTile) issue_piece(n / 3, next2_t0, bufoff_next2);
      }
    }
    // ---- online softmax; register i of lane (c, h) is token t0 + 8 (i / 4) + 4 h + i % 4 of head c
    float mx = -__builtin_inff();
    if (t0 + kTile > t_end) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        st[i] = (t0 + 8 * (i >> 2) + 4 * h + (i & 3) < t_end) ? st[i] * p.scale_log2e : -__builtin_inff();
    } else {
      st = st * p.scale_log2e;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[i]);
    {
      float a, b;
      pair_values(mx, a, b);
      mx = fmaxf(a, b);
    }
    // Deferred rescale (as in the 16-heads-per-wave kernel): the reference m_run moves only when
    // some head's maximum outgrows it by 2^8, so the 256 accumulator registers are rescaled on a
    // handful of tiles per sequence; sums, outputs and split records are relative to the same m_run.
    constexpr float kDefer = 8.f;
    if (__builtin_amdgcn_ballot_w64(mx > m_run + kDefer) != 0) {
      const float m_new = fmaxf(m_run, mx);
      const float f = fast_exp2(m_run - m_new);
      m_run = m_new;
      l_run *= f;
#pragma unroll
      for (int i = 0; i < kDB; ++i) scale_acc(o[i], f);
    }
    // ---- O^T += V^T . P^T.  k-slot j of lane half h in step u is token 16 u + 8 (j / 4) + 4 h + j % 4:
    // registers 8 u .. 8 u + 7 of P as they stand, two transposing 4-token reads of V.
    constexpr int NV = 2 * kDB, kAheadV = WIDE_KAV;
    auto v_at = [&](int n) __attribute__((always_inline)) {   // n-th operand: step u = n / kDB, dim block n % kDB
      if (WIDE_KO & 8) return u32x4{(uint32_t)lane, 0u, 0u, 0u};
      const int off = bufoff + v_lane + (16 * (n / kDB)) * kPitch + (n % kDB) * 64;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(lds + off + 8 * kPitch + 32));
      const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
      return u32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
    };
    u32x4 vbuf[kAheadV];
#pragma unroll
    for (int n = 0; n < kAheadV; ++n) vbuf[n] = v_at(n);
    float ps = 0.f;
    u32x4 pb[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      st[i] = fast_exp2(st[i] - m_run);
      ps += st[i];
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) pb[0][w] = Mma32<Tag>::pack2(st[2 * w], st[2 * w + 1]);
    asm volatile("s_nop 3");   // the packed weights come straight from the VALU; the compiler does not see their consumer
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      const u32x4 va = vbuf[n % kAheadV];
      if (n + kAheadV < NV) vbuf[n % kAheadV] = v_at(n + kAheadV);
      if (WIDE_KO & 4) asm volatile("" ::"v"(va), "v"(pb[n / kDB]));
      else out_mma<Tag>(o[n % kDB], va, pb[n / kDB]);
      if (n < 8) {   // the second half of the weights, two per matrix instruction of the first half
        st[8 + n] = fast_exp2(st[8 + n] - m_run);
        ps += st[8 + n];
      }
      if (n == 9) {
#pragma unroll
        for (int w = 0; w < 4; ++w) pb[1][w] = Mma32<Tag>::pack2(st[8 + 2 * w], st[8 + 2 * w + 1]);
        l_run += ps;
      }
      if (n == kDB - 1) asm volatile("s_nop 1");
      if (n == kDB + 1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoadsPerTile) : "memory");
        rotate(bufoff_next);
      }
    }
    (void)next_t0;
  };

  // Tile t lives in buffer t mod 3.  A tile's loads go out two tiles ahead (during tile t - 2's
  // score product), are waited for by the wave that issued them one tile ahead (which then rotates
  // the rope rows among them), and the barrier at the top of tile t makes everybody's part visible
  // to everybody -- and says that everybody is done reading tile t - 1, whose buffer the loads of
  // tile t + 2 then overwrite.  Loads past the last tile are issued all the same (clamped to the
  // last row), so that the count of loads in flight is the same at every wait.
  if (ntiles > 0) {
    issue_tile(t_begin, 0);
    issue_tile(t_begin + kTile, kBufBytes);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoadsPerTile) : "memory");
    rotate(0);
    int cur = 0, nxt = kBufBytes, nxt2 = 2 * kBufBytes;
    for (int t = 0; t < ntiles; ++t) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int t0 = t_begin + t * kTile;
      if (wave_live) {
        tile_step(t0, cur, t0 + kTile, t0 + 2 * kTile, nxt, nxt2);
      } else {   // a wave without heads only stages
        issue_tile(t0 + 2 * kTile, nxt2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoadsPerTile) : "memory");
        rotate(nxt);
      }
      const int c0 = cur;
      cur = nxt;
      nxt = nxt2;
      nxt2 = c0;
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 3" ::: "memory");   // (+ the last matrix results are readable)
  }
#pragma unroll
  for (int i = 0; i < kDB; ++i) asm volatile("" : "+a"(o[i]));

  {
    float a, b;
    pair_values(l_run, a, b);
    l_run = a + b;
  }
  if (!head_ok) return;
  // lane (c, h) holds O[head c][32 db + 8 (i / 4) + 4 h + i % 4] in o[db][i]
  float* dst;
  float inv = 1.f;
  if (p.nsplit == 1) {
    inv = l_run > 0.f ? 1.f / l_run : 0.f;
    dst = p.out + ((int64_t)row * p.H + head) * kR + 4 * h;
  } else {
    const int64_t rec = ((int64_t)row * p.H + head) * p.nsplit + sp;
    dst = p.ws_acc + rec * kR + 4 * h;
    if (h == 0) {
      p.ws_ml[rec * 2] = m_run;
      p.ws_ml[rec * 2 + 1] = l_run;
    }
  }
#pragma unroll
  for (int db = 0; db < kDB; ++db)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 v = {o[db][4 * g4], o[db][4 * g4 + 1], o[db][4 * g4 + 2], o[db][4 * g4 + 3]};
      *reinterpret_cast<f32x4*>(dst + 32 * db + 8 * g4) = v * inv;
    }
}

template <class Tag, int PAGED>
int launch_wide(const MlaParams& p, hipStream_t s) {
  constexpr int kLds = kIdsOff + (is_paged(PAGED) ? kIdsMax * 4 : 0);
  static bool prepared[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(MI355_EINVAL, "mla_decode: device index");
  if (!prepared[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&mla_decode_kernel_wide<Tag, PAGED>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess)
      return fail(MI355_EUNSUPPORTED, "mla_decode: %d bytes of LDS per workgroup refused", kLds);
    prepared[dev] = true;
  }
  const int64_t grid = (int64_t)p.rows * p.nsplit * p.n_ht;
  hipLaunchKernelGGL((mla_decode_kernel_wide<Tag, PAGED>), dim3((unsigned)grid), dim3(256), kLds, s, p);
  return check_launch("mla_decode_kernel_wide");
}

}  // namespace

bool mla_wide_supported(int H, int R, int P, int dtype) {
  return R == kR && P == kP && H > 64 && (dtype == MI355_BF16 || dtype == MI355_F16);
}

// Most tokens per workgroup a paged launch can hold block ids for (the plan splits longer ranges).
int mla_wide_max_paged_tokens(int block_size) { return kIdsMax * block_size; }

int mla_wide_launch(const void* params, int dtype, bool paged, hipStream_t stream) {
  const MlaParams& p = *static_cast<const MlaParams*>(params);
  if (paged) return dtype == MI355_BF16 ? launch_wide<BF16, kPaged>(p, stream) : launch_wide<F16, kPaged>(p, stream);
  return dtype == MI355_BF16 ? launch_wide<BF16, kDense>(p, stream) : launch_wide<F16, kDense>(p, stream);
}

}  // namespace mi355
