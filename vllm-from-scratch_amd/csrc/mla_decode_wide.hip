// Latent-space MLA attention for more than 16 heads: 128 heads per workgroup on v_mfma_f32_32x32x16.
//
// Replaces the attention core of the reference's MLA module (04_mla/mini_mla.py:234-299) for
// kv_lora_rank 512 and a rope slot of 64, like the other kernels of mla_decode.hip.
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
// global -> LDS directly (global_load_lds_dwordx4), two 32-token tiles ahead, into four buffers.
// One wave per SIMD also means that nothing but the wave's own instruction stream can keep the
// matrix pipe busy, so the tiles are software-pipelined: while the score product of tile t + 1
// runs, the wave does the softmax arithmetic of tile t between its matrix instructions (and issues
// the loads of tile t + 3); while the value product of tile t runs, it waits for its own loads of
// tile t + 2 and rotates the rope part of the eight rows among them.  One barrier per tile.
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
// as matrix instructions: where the VALU reads a chain's result right away (the first tile's, LAST)
// the last instruction carries the wait states that read needs (16 passes + 2) -- in the loop a whole
// value product lies between a chain and its readers --, and dependent steps of the chain reuse the
// same registers, which the matrix pipe forwards without wait states.
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
#ifndef WIDE_TIMING
#define WIDE_TIMING 0
#endif
#ifndef WIDE_PACKED
#define WIDE_PACKED 0
#endif
#ifndef WIDE_KG
#define WIDE_KG 3    // LDS operands per group of the score product (two groups in flight)
#endif
#ifndef WIDE_KGV
#define WIDE_KGV 2   // ... of the value product
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
constexpr int kBufBytes = kRopeOff + kTile * 128;     // 38976
constexpr int kBuffers = 4;
constexpr int kIdsOff = kBuffers * kBufBytes;         // block ids of the workgroup's token range (paged caches)
constexpr int kIdsMax = (160 * 1024 - kIdsOff) / 4;   // 1984: what is left of the CU's 160 KiB
constexpr int kLdsLoads = 8 + 1;                      // direct-to-LDS loads a wave issues per tile: 8 c_kv rows, 1 KiB of rope parts

__device__ __forceinline__ constexpr int row_off(int r) { return r * kPitch + ((r >> 2) & 3) * 16; }

typedef __attribute__((address_space(3))) unsigned char lds_byte_t;

// f(integral_constant<int, B>), ..., f(integral_constant<int, E - 1>): a loop whose index is a
// compile-time constant in the body (the pieces dealt out between matrix instructions are chosen
// by it; `#pragma unroll` leaves that to the optimiser, which gave up on the larger bodies).
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// Direct-to-LDS loads: 64 lanes x 16 bytes from global memory to 1 KiB of LDS at byte address `dst`
// (wave-uniform).  Written as asm: given the builtin, the compiler waits for EVERY such load in
// flight (vmcnt(0)) before the next LDS read it cannot prove independent, which would drain the two
// tiles of prefetch on every tile.  The compiler therefore does not count these loads either; the
// kernel waits for them itself (s_waitcnt vmcnt(N) below).
//   to_lds16_rows: every lane reads base + voff (base wave-uniform, in SGPRs); LDS address dst + DST_CONST
//   to_lds16_lanes: every lane reads its own 64-bit address
// M0 carries the LDS address.  It is written and left: nothing the compiler generates for this
// kernel uses M0 (it would set it itself before a use of its own).
template <int DST_CONST>
__device__ __forceinline__ void to_lds16_rows(const void* base, uint32_t voff, uint32_t dst) {
  if (WIDE_KO & 16) return;
  asm volatile("s_add_i32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(voff), "s"(base), "s"(dst), "n"(DST_CONST)
               : "memory", "scc");
}
__device__ __forceinline__ void to_lds16_lanes(const void* src, uint32_t dst) {
  if (WIDE_KO & 16) return;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "memory");
}

// A 16-byte load into registers that the compiler does not count either (its own wait for the
// value would be sized without the direct-to-LDS loads around it): the value may be used only
// behind wait_loads, which names it.
__device__ __forceinline__ void load16_uncounted(f32x4& dst, const float* src) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(src) : "memory");
}
// 16 bytes per lane written through to / read from device-coherent memory (sc1: neither left dirty
// in, nor served from a stale line of, the XCD's L2): what two workgroups of one launch hand each
// other.  Address = base (wave-uniform, in SGPRs) + voff.  Uncounted like the loads above; a loaded
// value may be used only behind a wait that names it.
__device__ __forceinline__ void store16_coherent(float* base, uint32_t voff, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(voff), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void load16_coherent(f32x4& dst, const float* base, uint32_t voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=&v"(dst) : "v"(voff), "s"(base) : "memory");
}
template <int LEFT_IN_FLIGHT>
__device__ __forceinline__ void wait_loads(f32x4& a, f32x4& b) {
  // (a side build without the direct-to-LDS loads has nothing behind the register loads to count)
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"((WIDE_KO & 16) ? 0 : LEFT_IN_FLIGHT) : "memory");
}

// x of lane `l` (wave-uniform), in every lane.
__device__ __forceinline__ float lane_value(float x, int l) {
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(x), l));
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

// Packed f32 arithmetic (v_pk_fma_f32, v_pk_mul_f32) issued beside matrix instructions stalls the SIMD
// far longer than the two plain instructions it replaces; the kernel is compiled without it.
#if defined(__HIP_DEVICE_COMPILE__) && !WIDE_PACKED
#define MI355_NO_PACKED_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define MI355_NO_PACKED_F32
#endif

template <class Tag, int PAGED>
__global__ void __launch_bounds__(256) MI355_NO_PACKED_F32 mla_decode_kernel_wide(const MlaParams p) {
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
  // Two token splits merged in this launch (p.counters): the workgroup of a pair that finishes later
  // combines its registers with the other's record.  Split 0 marks the pair's arrival cell as this
  // launch's when it starts, and writes the row itself when split 1 holds no tokens.
  const bool fused = p.counters != nullptr;
  const int nvalid = min(p.nsplit, (ctx + p.tps - 1) / p.tps);   // splits that hold tokens
  if (p.nsplit > 1 && t_begin >= ctx && !(fused && sp == 0)) return;
  if (fused && sp == 0 && tid == 0)
    (void)__hip_atomic_fetch_max(p.counters + (int64_t)row * p.n_ht + ht, p.epoch << 16, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
  // (merged in the launch, split 1 takes what split 0 leaves: the plan gives split 0 fewer tokens, so
  // that its record is out when split 1 is done and only one of the pair writes one)
  const int t_end = fused && sp == 1 ? ctx : min(ctx, t_begin + p.tps);
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
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the header's __syncthreads would stay out of line here)
    __builtin_amdgcn_s_barrier();
  }
  const int blk0 = t_begin >> p.bs_shift;

  // One tile's loads.  c_kv: one row = one 64-lane load straight into LDS, eight rows per wave;
  // rope parts: wave w fetches those of rows 8w .. 8w + 7 with one such load, eight lanes a row, and
  // every lane the cos / sin columns of its four pairs of its row, into registers.  Rows past the
  // end of the range re-read its last rows (their scores are masked).
  const int rr = 8 * wave + (lane >> 3);             // the rope row this lane fetches and rotates
  const int rslot = lane & 7, rswz = (rr >> 1) & 7;
  const uint32_t lds0 = (uint32_t)(uintptr_t)lds;   // LDS byte address of the image
  // piece i of the direct-to-LDS loads of the tile at t0: 0 .. 7 the wave's c_kv rows, 8 its rope rows.
  // Dense rows: the wave takes rows w, w + 4, ...; the byte offset of row w + 4i from the sequence's
  // base is min(t0 + w, last) * stride + min(4i * stride, rows left behind it * stride): two scalar
  // instructions per row once the two products are known (TileSrc, once per tile).  Blocks of >= 8
  // tokens (kPagedGroups): the wave takes rows 8w .. 8w + 7, which lie in one block -- one table
  // lookup and one 64-bit base per wave and tile, a row adds min(i, rows left) * stride; a group past
  // the end of the range is pulled back to the last live one (its rows are masked either way).
  // Smaller blocks (kPaged): a lookup per row.
  struct TileSrc {
    uint32_t first, room;
    const char* ckv_base;
    const char* krope_base;
    int n_last;
  };
  const uint32_t lane16 = lane * 16;
  // (LDS addresses of the wave's first row; row 8w + i sits i pitches on, shifted by ((2w + i / 4) mod 4) x 16 B)
  const uint32_t dst_rows = __builtin_amdgcn_readfirstlane(lds0 + wave * kPitch);
  const uint32_t dst_grp_a = __builtin_amdgcn_readfirstlane(lds0 + 8 * wave * kPitch + ((2 * wave) & 3) * 16);
  const uint32_t dst_grp_b = __builtin_amdgcn_readfirstlane(lds0 + 8 * wave * kPitch + ((2 * wave + 1) & 3) * 16);
  auto tile_src = [&](int t0) __attribute__((always_inline)) {
    const int last = t_end - 1;
    TileSrc ts = {};
    if constexpr (is_grouped(PAGED)) {
      const int gs = min(t0 + 8 * wave, last & ~7);
      ts.n_last = min(last - gs, 7);
      ts.room = (uint32_t)(ts.n_last * cstride2);
      const int id = __builtin_amdgcn_readfirstlane(
          *(const __attribute__((address_space(3))) int*)(lds + kIdsOff + 4 * ((gs >> p.bs_shift) - blk0)));
      const int r0 = gs & p.bs_mask;
      ts.ckv_base = ckv + ((int64_t)id * p.ckv_stride_blk * 2 + (int64_t)r0 * cstride2);
      ts.krope_base = krope + ((int64_t)id * p.krope_stride_blk * 2 + (int64_t)r0 * rstride2);
    } else {
      const int r0 = min(t0 + wave, last);
      ts.first = (uint32_t)(r0 * cstride2);
      ts.room = (uint32_t)((last - r0) * cstride2);
    }
    return ts;
  };
  auto issue_piece = [&](auto ic, int t0, const TileSrc& ts, int bufoff) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    const int last = t_end - 1;
    if constexpr (i < 8) {
      if constexpr (is_grouped(PAGED)) {
        const uint32_t off = min((uint32_t)i * (uint32_t)cstride2, ts.room);
        to_lds16_rows<i * kPitch>(ts.ckv_base, lane16 + off, (i < 4 ? dst_grp_a : dst_grp_b) + bufoff);
      } else if constexpr (is_paged(PAGED)) {
        const int tok = min(t0 + wave + 4 * i, last);                 // wave-uniform
        const int id = *(const __attribute__((address_space(3))) int*)(lds + kIdsOff + 4 * ((tok >> p.bs_shift) - blk0));
        const char* src = ckv + ((int64_t)__builtin_amdgcn_readfirstlane(id) * p.ckv_stride_blk * 2 + (int64_t)(tok & p.bs_mask) * cstride2);
        to_lds16_rows<row_off(4 * i)>(src, lane16, dst_rows + bufoff);
      } else {
        const uint32_t off = ts.first + min((uint32_t)(4 * i) * (uint32_t)cstride2, ts.room);
        to_lds16_rows<row_off(4 * i)>(ckv, lane16 + off, dst_rows + bufoff);
      }
    } else {
      const char* rsrc;
      if constexpr (is_grouped(PAGED)) {
        rsrc = ts.krope_base + (uint32_t)(min(lane >> 3, ts.n_last) * rstride2);
      } else if constexpr (is_paged(PAGED)) {
        const int tokr = min(t0 + rr, last);
        const int id = *(const __attribute__((address_space(3))) int*)(lds + kIdsOff + 4 * ((tokr >> p.bs_shift) - blk0));
        rsrc = krope + ((int64_t)id * p.krope_stride_blk * 2 + (int64_t)(tokr & p.bs_mask) * rstride2);
      } else {
        rsrc = krope + (uint32_t)(min(t0 + rr, last) * rstride2);
      }
      to_lds16_lanes(rsrc + ((rslot ^ rswz) << 4), __builtin_amdgcn_readfirstlane(lds0 + bufoff + kRopeOff + wave * 1024));
    }
  };
  auto issue_tile = [&](int t0, int bufoff) __attribute__((always_inline)) {
    const TileSrc ts = tile_src(t0);
    static_for<0, kLdsLoads>([&](auto ic) __attribute__((always_inline)) { issue_piece(ic, t0, ts, bufoff); });
  };
  // cos / sin columns 4 rslot .. + 4 of the position of rope row rr of the tile at t0, into registers
  f32x4 cv, sv;
  auto issue_factor = [&](auto which, int t0) __attribute__((always_inline)) {
    const int pos = min(min(t0 + rr, t_end - 1), p.table_len - 1);
    if constexpr (decltype(which)::value == 0) load16_uncounted(cv, p.cos_t + (int64_t)pos * kP + 4 * rslot);
    else load16_uncounted(sv, p.sin_t + (int64_t)pos * kP + 4 * rslot);
  };
  auto issue_factors = [&](int t0) __attribute__((always_inline)) {
    issue_factor(std::integral_constant<int, 0>{}, t0);
    issue_factor(std::integral_constant<int, 1>{}, t0);
  };

  // Rotation of this lane's four (lo, hi) pairs of rope row rr: dims 4 rslot .. + 4 and their
  // partners 32 dims on, by the factors in cv / sv.  In pieces (0: read, 1 / 2: unpack, 3 .. 6: one
  // pair each, 7 / 8: pack, 9: write), so that they can be dealt out between matrix instructions.
  constexpr int kRotatePieces = 10;
  u32x2 rot_lo2, rot_hi2;
  float rot_lo[4], rot_hi[4], rot_nlo[4], rot_nhi[4];
  auto rotate_piece = [&](auto kc, int bufoff) __attribute__((always_inline)) {
    constexpr int k = decltype(kc)::value;
    if constexpr ((WIDE_KO & 64) != 0) return;
    lds_byte_t* lo_p = lds + bufoff + kRopeOff + rr * 128 + (((rslot >> 1) ^ rswz) << 4) + (rslot & 1) * 8;
    lds_byte_t* hi_p = lds + bufoff + kRopeOff + rr * 128 + (((4 + (rslot >> 1)) ^ rswz) << 4) + (rslot & 1) * 8;
    if constexpr (k == 0) {
      rot_lo2 = *(const __attribute__((address_space(3))) u32x2*)lo_p;
      rot_hi2 = *(const __attribute__((address_space(3))) u32x2*)hi_p;
    } else if constexpr (k == 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        rot_lo[2 * j] = Tag::to_float((typename Tag::elem_t)(rot_lo2[j] & 0xffffu));
        rot_lo[2 * j + 1] = Tag::to_float((typename Tag::elem_t)(rot_lo2[j] >> 16));
      }
    } else if constexpr (k == 2) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        rot_hi[2 * j] = Tag::to_float((typename Tag::elem_t)(rot_hi2[j] & 0xffffu));
        rot_hi[2 * j + 1] = Tag::to_float((typename Tag::elem_t)(rot_hi2[j] >> 16));
      }
    } else if constexpr (k < 7) {
      constexpr int j = k - 3;
      rot_nlo[j] = rot_lo[j] * cv[j] - rot_hi[j] * sv[j];
      rot_nhi[j] = rot_hi[j] * cv[j] + rot_lo[j] * sv[j];
    } else if constexpr (k == 7) {
      rot_lo2 = u32x2{Mma32<Tag>::pack2(rot_nlo[0], rot_nlo[1]), Mma32<Tag>::pack2(rot_nlo[2], rot_nlo[3])};
    } else if constexpr (k == 8) {
      rot_hi2 = u32x2{Mma32<Tag>::pack2(rot_nhi[0], rot_nhi[1]), Mma32<Tag>::pack2(rot_nhi[2], rot_nhi[3])};
    } else {
      *(__attribute__((address_space(3))) u32x2*)lo_p = rot_lo2;
      *(__attribute__((address_space(3))) u32x2*)hi_p = rot_hi2;
    }
  };
  auto rotate = [&](int bufoff) __attribute__((always_inline)) {
    static_for<0, kRotatePieces>([&](auto kc) __attribute__((always_inline)) { rotate_piece(kc, bufoff); });
  };

#if WIDE_TIMING
  // side build: shader-clock time of wave 0 of workgroup 0 spent up to the barrier's release, in
  // the score phase and in the value phase of its steps (written over the head of the output)
  unsigned long long tm_prev = 0, tm_bar = 0, tm_s = 0, tm_pv = 0;
#define WIDE_STAMP(acc)                                              \
  {                                                                  \
    const unsigned long long now = __builtin_readcyclecounter();     \
    acc += now - tm_prev;                                            \
    tm_prev = now;                                                   \
  }
#else
#define WIDE_STAMP(acc)
#endif
#if WIDE_TIMING
  unsigned long long te[6] = {};   // the epilogue's phases on the 100 MHz clock ([5]: when the workgroup started)
  te[5] = __builtin_amdgcn_s_memrealtime();
#define WIDE_STAMP_END(k) te[k] = __builtin_amdgcn_s_memrealtime();
#else
#define WIDE_STAMP_END(k)
#endif
  // Per-lane LDS offsets of the operands (buffer offset added per tile).
  const int k_lane = row_off(c) + h * 16;                                  // K operand, latent k-steps: + 32 ks
  const int kr_lane = kRopeOff + c * 128, kr_swz = (c >> 1) & 7;           // rope k-steps: chunk (2 (ks - 32) + h) ^ swz
  const int grp = (lane >> 4) & 1, qi = (lane & 15) >> 2, pi = lane & 3;
  const int v_lane = (4 * h + qi) * kPitch + h * 16 + (16 * grp + 4 * pi) * 2;   // V^T operand
  const float scale = p.scale_log2e;                                       // > 0 (checked by the host)

  // S^T = K . Q^T of the tile in buffer `bufoff` into st; `between(n)` is issued after the n-th
  // matrix instruction.
  auto score_product = [&](f32x16& st, int bufoff, auto final_wait, auto between) __attribute__((always_inline)) {
    auto a_at = [&](int n) __attribute__((always_inline)) {
      if (WIDE_KO & 2) return u32x4{(uint32_t)lane, 0u, 0u, 0u};
      if (n < kR / 16) return *(const __attribute__((address_space(3))) u32x4*)(lds + bufoff + k_lane + 32 * n);
      return *(const __attribute__((address_space(3))) u32x4*)(lds + bufoff + kr_lane + (((2 * (n - kR / 16) + h) ^ kr_swz) << 4));
    };
    // Operands are read in groups of kGroup, two groups ahead: the wait ahead of a group's first
    // matrix instruction then covers the whole group (one s_waitcnt per kGroup matrix
    // instructions instead of one each; at one wave per SIMD every instruction issued counts).
    constexpr int kGroup = WIDE_KG, kGroups = kKS / kGroup, kAhead = 2 * kGroup;
    static_assert(kKS % kGroup == 0, "operand groups");
    u32x4 abuf[kAhead];
#pragma unroll
    for (int n = 0; n < kAhead; ++n) abuf[n] = a_at(n);
    static_for<0, kGroups>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      static_for<0, kGroup>([&](auto jc) __attribute__((always_inline)) {
        constexpr int n = g * kGroup + decltype(jc)::value;
        const u32x4 a = abuf[n % kAhead];
        if constexpr ((WIDE_KO & 1) != 0) {
          if constexpr (n == 0) st = f32x16(0.f);
          asm volatile("" ::"v"(a));
        } else if constexpr (n == 0) score_mma<Tag, true, false>(st, a, qf[n]);
        else if constexpr (n == kKS - 1) score_mma<Tag, false, decltype(final_wait)::value>(st, a, qf[n]);
        else score_mma<Tag, false, false>(st, a, qf[n]);
        between(std::integral_constant<int, n>{});
      });
      if constexpr (g + 2 < kGroups) {
#pragma unroll
        for (int j = 0; j < kGroup; ++j) abuf[(g * kGroup + j) % kAhead] = a_at((g + 2) * kGroup + j);
      }
    });
  };

  // One step of a live wave: the score product of tile t + 1 (into st_next) with the softmax
  // arithmetic of tile t (from st_cur) and the loads of tile t + 3 between its matrix instructions,
  // then the value product of tile t with the wait for and rotation of tile t + 2 in its shadow.
  // b0 .. b3: the buffers of tiles t .. t + 3.
  auto tile_step = [&](int t0, f32x16& st_cur, f32x16& st_next, int b0, int b1, int b2, int b3) __attribute__((always_inline)) {
    // What goes where.  A matrix instruction occupies the pipe for 32 cycles; measured with one
    // wave per SIMD (tools/micro/mfma_valu_coissue.hip), about five plain VALU instructions, or two
    // to three exponentials, issued behind it are free, and whatever exceeds that in ONE slot delays
    // the next matrix instruction -- slack in a later slot does not pay it back.  So every slot of
    // the score product gets one LDS operand read and at most an exponential with its multiply-add
    // and its addition, or four plain instructions, or one load with its address arithmetic.
    float mxa, mxb, mxc, mxd, ps = 0.f;
    u32x4 pb[2];
    auto softmax_piece = [&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;   // 0 .. 3 maximum, 4 .. 19 weights, 20 / 21 packing, 22 sum
      if constexpr ((WIDE_KO & 32) != 0) {   // no softmax arithmetic: the weights are the raw scores' bits
        if constexpr (k == 20 || k == 21) {
          constexpr int u = k - 20;
#pragma unroll
          for (int w = 0; w < 4; ++w) pb[u][w] = f2u(st_cur[8 * u + 2 * w]);
        }
        return;
      }
      // register i of lane (c, h) is token t0 + 8 (i / 4) + 4 h + i % 4 of head c
      if constexpr (k == 0) {
        if (t0 + kTile > t_end) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            if (t0 + 8 * (i >> 2) + 4 * h + (i & 3) >= t_end) st_cur[i] = -__builtin_inff();
        }
        mxa = __builtin_fmaxf(__builtin_fmaxf(st_cur[0], st_cur[1]), st_cur[2]);
        mxb = __builtin_fmaxf(__builtin_fmaxf(st_cur[3], st_cur[4]), st_cur[5]);
      } else if constexpr (k == 1) {
        mxc = __builtin_fmaxf(__builtin_fmaxf(st_cur[6], st_cur[7]), st_cur[8]);
        mxd = __builtin_fmaxf(__builtin_fmaxf(st_cur[9], st_cur[10]), st_cur[11]);
      } else if constexpr (k == 2) {
        mxa = __builtin_fmaxf(__builtin_fmaxf(mxa, mxb), mxc);
        mxb = __builtin_fmaxf(__builtin_fmaxf(st_cur[12], st_cur[13]), st_cur[14]);
        mxa = __builtin_fmaxf(__builtin_fmaxf(mxa, mxb), mxd);
        mxa = __builtin_fmaxf(mxa, st_cur[15]);
      } else if constexpr (k == 3) {
        float a, b;
        pair_values(mxa, a, b);
        const float mx = __builtin_fmaxf(a, b) * scale;
        // Deferred rescale (as in the 16-heads-per-wave kernel): the reference m_run moves only when
        // some head's maximum outgrows it by 2^8, so the 256 accumulator registers are rescaled on a
        // handful of tiles per sequence; sums, outputs and split records are relative to the same
        // m_run.  (The value product that last wrote the accumulators has left the matrix pipe: the
        // score product's instructions issued since then went in behind it.)
        constexpr float kDefer = 8.f;
        if (__builtin_amdgcn_ballot_w64(mx > m_run + kDefer) != 0) {
          const float m_new = fmaxf(m_run, mx);
          const float f = fast_exp2(m_run - m_new);
          m_run = m_new;
          l_run *= f;
#pragma unroll
          for (int i = 0; i < kDB; ++i) scale_acc(o[i], f);
        }
      } else if constexpr (k < 20) {
        constexpr int i = k - 4;
        st_cur[i] = fast_exp2(__builtin_fmaf(st_cur[i], scale, -m_run));
        ps += st_cur[i];
      } else if constexpr (k < 22) {           // k-slot j of lane half h in step u is token 16 u + 8 (j / 4) + 4 h + j % 4:
        constexpr int u = k - 20;              // registers 8 u .. 8 u + 7 of the weights as they stand
#pragma unroll
        for (int w = 0; w < 4; ++w) pb[u][w] = Mma32<Tag>::pack2(st_cur[8 * u + 2 * w], st_cur[8 * u + 2 * w + 1]);
      } else {
        l_run += ps;
      }
    };
    constexpr int NV = 2 * kDB;
    auto v_at = [&](int n) __attribute__((always_inline)) {   // n-th operand: step u = n / kDB, dim block n % kDB
      if (WIDE_KO & 8) return u32x4{(uint32_t)lane, 0u, 0u, 0u};
      const int off = b0 + v_lane + (16 * (n / kDB)) * kPitch + (n % kDB) * 64;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + off));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(lds + off + 8 * kPitch + 32));
      const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
      return u32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
    };
    constexpr int kGroupV = WIDE_KGV, kGroupsV = NV / kGroupV, kAheadV = 2 * kGroupV;
    static_assert(NV % kGroupV == 0, "operand groups");
    u32x4 vbuf[kAheadV];
    const TileSrc ts3 = tile_src(t0 + 3 * kTile);
    score_product(st_next, b1, std::false_type{}, [&](auto nc) __attribute__((always_inline)) {
      constexpr int n = decltype(nc)::value;
      if constexpr (n < 2) issue_factor(nc, t0 + 2 * kTile);                                  // (ahead of this step's direct-to-LDS loads)
      else if constexpr (n < 25) softmax_piece(std::integral_constant<int, n - 2>{});
      else if constexpr (n < 25 + kLdsLoads) issue_piece(std::integral_constant<int, n - 25>{}, t0 + 3 * kTile, ts3, b3);
      else if constexpr (n == 25 + kLdsLoads) {   // the value product's first operands: their LDS round trip passes here
#pragma unroll
        for (int j = 0; j < kAheadV; ++j) vbuf[j] = v_at(j);
      }
    });
    WIDE_STAMP(tm_s)
    // ---- O^T += V^T . P^T of tile t
    static_for<0, kGroupsV>([&](auto gc) __attribute__((always_inline)) {
      constexpr int g = decltype(gc)::value;
      static_for<0, kGroupV>([&](auto jc) __attribute__((always_inline)) {
        constexpr int n = g * kGroupV + decltype(jc)::value;
        const u32x4 va = vbuf[n % kAheadV];
        out_mma<Tag>(o[n % kDB], va, pb[n / kDB]);
        if constexpr (n + 2 * kGroupV < NV && kGroupV == 1) vbuf[n % kAheadV] = v_at(n + kAheadV);
        // the rope rows of tile t + 2: this wave's part of the tile and its factors are in once only
        // tile t + 3's loads are left in flight
        if constexpr (n == 6) wait_loads<kLdsLoads>(cv, sv);
        if constexpr (n >= 6 && n < 6 + 2 * kRotatePieces && n % 2 == 0) rotate_piece(std::integral_constant<int, (n - 6) / 2>{}, b2);
      });
      if constexpr (g + 2 < kGroupsV && kGroupV > 1) {
#pragma unroll
        for (int j = 0; j < kGroupV; ++j) vbuf[(g * kGroupV + j) % kAheadV] = v_at((g + 2) * kGroupV + j);
      }
    });
  };

  // Tile t lives in buffer t mod 4.  Its loads go out during step t - 3, are waited for by the wave
  // that issued them during step t - 2 (which then rotates the rope rows among them), and the
  // barrier at the top of step t - 1 makes everybody's part visible to everybody before the score
  // product of tile t runs in that step -- and says that everybody is done with tile t - 2, whose
  // buffer the loads of tile t + 2 then overwrite.  Loads past the last tile are issued all the same
  // (clamped to the last row), so that the count of loads in flight is the same at every wait; the
  // score product of the tile past the last is computed and dropped.
  if (ntiles > 0) {
    f32x16 st_a, st_b;
    issue_tile(t_begin, 0);
    issue_factors(t_begin);
    issue_tile(t_begin + kTile, kBufBytes);
    issue_tile(t_begin + 2 * kTile, 2 * kBufBytes);
    wait_loads<2 * kLdsLoads>(cv, sv);
    rotate(0);
    issue_factors(t_begin + kTile);
    wait_loads<0>(cv, sv);
    rotate(kBufBytes);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave_live) score_product(st_a, 0, std::true_type{}, [](auto) __attribute__((always_inline)) {});
    int b0 = 0, b1 = kBufBytes, b2 = 2 * kBufBytes, b3 = 3 * kBufBytes;
#if WIDE_TIMING
    tm_prev = __builtin_readcyclecounter();
#endif
    auto step = [&](int t, f32x16& st_cur, f32x16& st_next) __attribute__((always_inline)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      WIDE_STAMP(tm_bar)
      const int t0 = t_begin + t * kTile;
      if (wave_live) {
        tile_step(t0, st_cur, st_next, b0, b1, b2, b3);
      } else {   // a wave without heads only stages
        issue_factors(t0 + 2 * kTile);
        issue_tile(t0 + 3 * kTile, b3);
        wait_loads<kLdsLoads>(cv, sv);
        rotate(b2);
      }
      WIDE_STAMP(tm_pv)
      const int f = b0;
      b0 = b1;
      b1 = b2;
      b2 = b3;
      b3 = f;
    };
    for (int t = 0; t < ntiles; t += 2) {
      step(t, st_a, st_b);
      if (t + 1 < ntiles) step(t + 1, st_b, st_a);
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
  WIDE_STAMP_END(0)
  // ---- Output.  Lane (c, h) holds O[head c][32 db + 8 (i / 4) + 4 h + i % 4] in o[db][i]: stored from
  // there, one instruction would touch 32 rows with 32 bytes each.  The tile buffers are free now, so
  // each wave turns its 32 heads x 512 dims around in its own buffer, 256 dims at a time (rows at a
  // pitch of 1040 bytes: the sixteen lanes of a write phase hit sixteen different bank quads), and
  // stores whole rows: 1 KiB contiguous per instruction.
  const bool direct = p.nsplit == 1 || (fused && nvalid < 2);
  const bool through = fused && !direct;            // records another workgroup of this launch reads
  // Has the other workgroup of the pair been counted already?  Then this one combines, and its own
  // record is never read: it is not written.  (Asked here, read after half a staging pass.)
  unsigned long long cell_seen = 0;
  if (through && tid == 0)
    cell_seen = __hip_atomic_load(p.counters + (int64_t)row * p.n_ht + ht, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_s_barrier();   // every wave is done with the tiles
  constexpr int kStagePitch = 1040;
  static_assert(32 * kStagePitch <= kBufBytes - 16, "a wave's half output fits in one tile buffer");
  lds_byte_t* const stage = lds + wave * kBufBytes;
  typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
  auto stage_half = [&](auto hc) __attribute__((always_inline)) {
    constexpr int hf = decltype(hc)::value;
#pragma unroll
    for (int dbl = 0; dbl < kDB / 2; ++dbl)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int db = hf * (kDB / 2) + dbl;
        *(lds_f32x4*)(stage + c * kStagePitch + (32 * dbl + 8 * g4 + 4 * h) * 4) =
            f32x4{o[db][4 * g4], o[db][4 * g4 + 1], o[db][4 * g4 + 2], o[db][4 * g4 + 3]};
      }
  };
  auto staged_row = [&](int r) __attribute__((always_inline)) { return *(lds_f32x4*)(stage + r * kStagePitch + lane * 16); };
  const int head0 = ht * 128 + wave * 32;           // this wave's heads: head0 .. head0 + 31
  const int nheads = min(32, p.H - head0);          // (<= 0: a wave without heads)
  volatile __attribute__((address_space(3))) int* flag =
      (volatile __attribute__((address_space(3))) int*)(lds + 4 * kBufBytes - 16);   // (past wave 3's staged rows)
  bool other_in = false;
  if (direct) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* dst = p.out + ((int64_t)row * p.H + head0) * kR + 4 * lane;
    static_for<0, 2>([&](auto hc) __attribute__((always_inline)) {
      if (nheads > 0) stage_half(hc);
#pragma unroll
      for (int r = 0; r < 32; ++r)
        if (r < nheads)
          *reinterpret_cast<f32x4*>(dst + (int64_t)r * kR + 256 * decltype(hc)::value) = staged_row(r) * lane_value(inv, r);
    });
  } else {
    if (through) {
      if (nheads > 0) stage_half(std::integral_constant<int, 0>{});
      if (tid == 0) {
        // (asking again for a bounded while when split 0 is not in yet -- up to 32 times, ~0.7 us
        // each -- measured no gain: 163.3 against 163.9 us)
        const unsigned long long counted_once = (p.epoch << 16) | 1ull;   // the launch's tag, one arrival
        flag[1] = cell_seen == counted_once;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      other_in = flag[1] != 0;
    }
    if (!other_in) {
      const int64_t rec0 = ((int64_t)row * p.H + head0) * p.nsplit + sp;
      if (head_ok && h == 0) {
        float* ml = p.ws_ml + (rec0 + (int64_t)c * p.nsplit) * 2;
        if (through) {
          const unsigned long long v = ((unsigned long long)__float_as_uint(l_run) << 32) | __float_as_uint(m_run);
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(ml), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          ml[0] = m_run;
          ml[1] = l_run;
        }
      }
      float* dst = p.ws_acc + rec0 * kR;
      const int64_t pitch = (int64_t)p.nsplit * kR;
      static_for<0, 2>([&](auto hc) __attribute__((always_inline)) {
        if (nheads > 0 && !(through && decltype(hc)::value == 0)) stage_half(hc);
#pragma unroll
        for (int r = 0; r < 32; ++r)
          if (r < nheads) {
            float* at = dst + r * pitch + 256 * decltype(hc)::value;
            if (through) store16_coherent(at, 16 * lane, staged_row(r));
            else *reinterpret_cast<f32x4*>(at + 4 * lane) = staged_row(r);
          }
      });
    }
  }
  WIDE_STAMP_END(1)
  if (through) {
    // The pair's records are written through the XCD's L2 and acknowledged before the workgroup is
    // counted (plain accesses between a device-scope release and acquire instead -- a write-back and
    // an invalidation of the whole L2 per wave -- measured 195-200 against 166-170 us).  A cell counts while its upper 48
    // bits are the launch's tag (decode_merge.h has the protocol: the workspace is never cleared);
    // the second arriver resets it for a graph replay.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WIDE_STAMP_END(2)
    __builtin_amdgcn_s_barrier();
    if (tid == 0) {
      unsigned long long* cell = p.counters + (int64_t)row * p.n_ht + ht;
      unsigned long long seen = __hip_atomic_fetch_add(cell, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned long long next = seen + 1;
      if ((seen >> 16) != p.epoch) {   // got here before split 0's mark, or the cell held something above every tag
        seen = next;
        do {
          next = (seen >> 16) == p.epoch ? seen + 1 : ((p.epoch << 16) | 1ull);
        } while (!__hip_atomic_compare_exchange_strong(cell, &seen, next, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT));
      }
      const int last = (int)(next & 0xffffu) == 2;
      if (last) __hip_atomic_store(cell, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      flag[0] = last;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool last = flag[0] != 0;
    WIDE_STAMP_END(3)
    if (last && nheads > 0) {
      // out = (O_0 f_0 + O_1 f_1) / (l_0 f_0 + l_1 f_1) with the merge kernel's arithmetic, in split
      // order whichever of the two is in registers.  Lane c works out head c's factors; the rows
      // are combined in the staged layout (a row per instruction), whose factors come by readlane.
      const int64_t rec_o = ((int64_t)row * p.H + head0) * 2 + (1 - sp);
      const float* src = p.ws_acc + rec_o * kR;
      float* dst = p.out + ((int64_t)row * p.H + head0) * kR + 4 * lane;
      // The other record, half a row set (32 rows x 1 KiB) in flight at a time; this wave's own half
      // is turned around in LDS, and the factors are worked out, while it travels.
      f32x4 x[32];
      auto fetch_half = [&](auto hc) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 32; ++r) load16_coherent(x[r], src + (int64_t)min(r, nheads - 1) * (2 * kR) + 256 * decltype(hc)::value, 16 * lane);
      };
      auto await_half = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                       "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
                     :
                     : "memory");
        asm volatile(""
                     : "+v"(x[16]), "+v"(x[17]), "+v"(x[18]), "+v"(x[19]), "+v"(x[20]), "+v"(x[21]), "+v"(x[22]), "+v"(x[23]),
                       "+v"(x[24]), "+v"(x[25]), "+v"(x[26]), "+v"(x[27]), "+v"(x[28]), "+v"(x[29]), "+v"(x[30]), "+v"(x[31])
                     :
                     : "memory");
      };
      fetch_half(std::integral_constant<int, 0>{});
      float m_o = -1.0e30f, l_o = 0.f;
      if (head_ok) {
        const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p.ws_ml + (rec_o + 2 * c) * 2),
                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        m_o = __uint_as_float((unsigned)v);
        l_o = __uint_as_float((unsigned)(v >> 32));
      }
      if (!other_in) stage_half(std::integral_constant<int, 0>{});   // (else still there)
      const float m0 = sp == 0 ? m_run : m_o, m1 = sp == 0 ? m_o : m_run;
      const float l0 = sp == 0 ? l_run : l_o, l1 = sp == 0 ? l_o : l_run;
      const float M = fmaxf(fmaxf(-1.0e30f, m0), m1);
      const float f0 = fast_exp2(m0 - M), f1 = fast_exp2(m1 - M);
      const float den = __builtin_fmaf(l1, f1, l0 * f0);
      const float inv = den > 0.f ? 1.f / den : 0.f;
      auto combine = [&](auto own_first) __attribute__((always_inline)) {
        static_for<0, 2>([&](auto hc) __attribute__((always_inline)) {
          constexpr int hf = decltype(hc)::value;
          if constexpr (hf == 1) {
            fetch_half(hc);
            stage_half(hc);
          }
          await_half();
#pragma unroll
          for (int r = 0; r < 32; ++r) {
            if (r < nheads) {
              const f32x4 own = staged_row(r);
              const float g0 = lane_value(f0, r), g1 = lane_value(f1, r), gi = lane_value(inv, r);
              f32x4 res;
#pragma unroll
              for (int e = 0; e < 4; ++e)
                res[e] = (decltype(own_first)::value ? __builtin_fmaf(x[r][e], g1, own[e] * g0) : __builtin_fmaf(own[e], g1, x[r][e] * g0)) * gi;
              *reinterpret_cast<f32x4*>(dst + (int64_t)r * kR + 256 * hf) = res;
            }
          }
        });
      };
      if (sp == 0) combine(std::true_type{});
      else combine(std::false_type{});
    }
  }
#if WIDE_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  WIDE_STAMP_END(4)
  if (blockIdx.x == 0 && tid == 0 && p.stamps) {
    unsigned long long* t = p.stamps + 8 * 64;
    t[0] = tm_bar;
    t[1] = tm_s;
    t[2] = tm_pv;
    t[3] = (unsigned long long)ntiles;
  }
  if (blockIdx.x < 64 && tid == 0 && p.stamps) {   // the first workgroups' epilogues: (split, loop end, records issued, acknowledged, counted, done)
    unsigned long long* t = p.stamps + 8 * blockIdx.x;
    t[0] = (unsigned long long)sp | ((unsigned long long)row << 8) | ((unsigned long long)other_in << 40);
#pragma unroll
    for (int k = 0; k < 5; ++k) t[1 + k] = te[k];
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    t[6] = te[5];
    t[7] = xcc & 15;
  }
#endif
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

// Bytes a -DWIDE_TIMING=1 side build appends to the workspace for its epilogue stamps (else 0).
size_t mla_wide_stamp_bytes() { return WIDE_TIMING ? 65 * 8 * sizeof(unsigned long long) : 0; }

bool mla_wide_supported(int H, int R, int P, int dtype) {
  const int least = knob(kKnobMlaWide);   // 0: off
  return least > 0 && R == kR && P == kP && H >= least && (dtype == MI355_BF16 || dtype == MI355_F16);
}

// Most tokens per workgroup a paged launch can hold block ids for (the plan splits longer ranges).
// (a range that does not start on a block boundary touches up to two blocks more than tokens / block_size)
int mla_wide_max_paged_tokens(int block_size) { return (kIdsMax - 2) * block_size; }

int mla_wide_launch(const void* params, int dtype, bool paged, hipStream_t stream) {
  const MlaParams& p = *static_cast<const MlaParams*>(params);
  if (paged && p.bs_shift >= 3 && knob(kKnobMlaRowGroups) != 0)   // blocks of >= 8 tokens: a wave's 8 rows of a tile share a block
    return dtype == MI355_BF16 ? launch_wide<BF16, kPagedGroups>(p, stream) : launch_wide<F16, kPagedGroups>(p, stream);
  if (paged) return dtype == MI355_BF16 ? launch_wide<BF16, kPaged>(p, stream) : launch_wide<F16, kPaged>(p, stream);
  return dtype == MI355_BF16 ? launch_wide<BF16, kDense>(p, stream) : launch_wide<F16, kDense>(p, stream);
}

}  // namespace mi355
