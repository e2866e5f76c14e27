// Latent-space MLA attention in fp32 on the fp32 matrix instruction.
//
// The reference's MultiHeadLatentAttention computes in fp32 (04_mla/mini_mla.py:203-306); the
// 16-bit kernels (mla_decode.hip) are the serving path, this one keeps fp32 operands end to end so
// that an fp32 module matches the reference to fp32 rounding: v_mfma_f32_16x16x4_f32 multiplies
// fp32 values exactly and accumulates in fp32, weights stay fp32.
//
//   score[h][t] = scale * ( q[h][:R] . c_kv[t]  +  q[h][R:] . rope(k_rope[t], t) )
//   out[h]      = softmax_t(score[h]) . c_kv[t]
//
// Workgroup = one query row x 16 heads; a tile = 16 cached tokens.  kv_lora_rank >= 256: four waves
// (below); smaller ranks: one wave that owns every dim (no exchange, no barrier).  Wave w owns the
// latent dims [w R/4, (w+1) R/4) -- as contraction dims of the scores AND as output dims, so it
// reads only its own quarter of every cached row, straight from memory into the A-operand layout
// (lane = token row, 16 bytes per lane per 16 dims) -- and an eighth of the rope pairs
// (dims i and i + P/2 rotate together): the lane quarter g holds a run of consecutive rope dims --
// g = 0, 1 from the low half, g = 2, 3 their partners from the high half -- fetched, like their
// cos / sin entries, in 16-byte pieces when the rows are aligned (VEC), and lanes g and g ^ 2
// trade their values for the rotation.  The four partial score tiles meet in LDS (one barrier
// per tile, two buffers); every wave then holds the full 16 x 16 scores with the head on the lane
// and four tokens in its registers: register r is, unchanged, the B operand of the instruction
// that contracts tokens {r, 4 + r, 8 + r, 12 + r}.  The A operand of that product is the wave's
// quarter tile transposed, read back from a wave-private LDS image (row pitch R/4 + 4 words: the
// four token rows a read touches land on four disjoint groups of 16 banks).
// A workgroup's walk over its tokens is one dependent chain (loads -> scores -> exchange ->
// softmax -> value product), so the chip is filled by running several workgroups per CU: when
// rows x head tiles leave CUs idle the context is cut into token splits whose fp32 partials are
// combined by the merge kernel of the 16-bit path (same record format).
#include "common.h"

namespace mi355 {
namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

#ifndef MI355_MLA_F32_TILES
#define MI355_MLA_F32_TILES 3
#endif

struct MlaF32Params {
  float* out;
  const float* q;
  const float* ckv;
  const float* krope;
  const float* cos_t;
  const float* sin_t;
  const int32_t* ctx;
  const int32_t* row_seq;
  int64_t q_stride_row, ckv_stride_seq, ckv_stride_tok, krope_stride_seq, krope_stride_tok;
  int rows, H, n_ht, table_len, max_ctx, nsplit, tps;
  float* ws_acc;   // [rows, H, nsplit, R] un-normalised partial outputs (nsplit > 1)
  float* ws_ml;    // [rows, H, nsplit, 2] (running maximum in log2 units, sum)
  float scale_log2e;
  // Paged caches (the PAGED instantiations): token t of sequence s sits in block bt[s][t >> bs_shift]
  // (clamped to the pool) at row t & bs_mask; the per-sequence strides are then unused.
  const int32_t* bt = nullptr;
  int64_t ckv_stride_blk = 0, krope_stride_blk = 0;
  int bs_shift = 0, bs_mask = 0, max_blocks = 0, num_blocks = 0;
};

__device__ __forceinline__ f32x4v mma4(float a, float b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// The value of x in both 16-lane rows of a row pair (rows 0|1 and 2|3) / in both halves of the wave:
// one v_permlane16_swap / v_permlane32_swap (it exchanges the odd rows [upper half] of its first
// register with the even rows [lower half] of its second), in the vector ALU instead of a trip
// through the LDS crossbar.  Written as asm because this compiler's builtins return the first
// result for both; the s_nop covers the wait states after a vector write of the operands.
__device__ __forceinline__ void row_pair_values(float x, float& even_row, float& odd_row) {
  float a = x, b = x;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  even_row = a;
  odd_row = b;
}
__device__ __forceinline__ void half_values(float x, float& lower, float& upper) {
  float a = x, b = x;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  lower = a;
  upper = b;
}
// Reductions over the four lanes c, c + 16, c + 32, c + 48 (fixed order: results repeat run to run).
__device__ __forceinline__ float quarters_max(float x) {
  float a, b;
  row_pair_values(x, a, b);
  x = fmaxf(a, b);
  half_values(x, a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float quarters_sum(float x) {
  float a, b;
  row_pair_values(x, a, b);
  x = a + b;
  half_values(x, a, b);
  return a + b;
}

// RS consecutive floats: 16-byte pieces (or one 8-byte piece) when VEC, else element by element.
template <int RS, bool VEC>
__device__ __forceinline__ void load_run(const float* src, float (&dst)[RS]) {
  if constexpr (VEC && RS % 4 == 0) {
#pragma unroll
    for (int i = 0; i < RS / 4; ++i) {
      const f32x4v v = *reinterpret_cast<const f32x4v*>(src + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[4 * i + e] = v[e];
    }
  } else if constexpr (VEC && RS == 2) {
    typedef float f32x2v __attribute__((ext_vector_type(2)));
    const f32x2v v = *reinterpret_cast<const f32x2v*>(src);
    dst[0] = v[0];
    dst[1] = v[1];
  } else {
#pragma unroll
    for (int e = 0; e < RS; ++e) dst[e] = src[e];
  }
}

template <int R, int P, int WAVES, bool VEC, bool PAGED>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 4 ? 2 : 1) mla_decode_f32_kernel(const MlaF32Params p) {
  constexpr int Q = R / WAVES;            // latent dims per wave
  constexpr int NJ = Q / 16;              // 16-dim groups per wave
  constexpr int PR = P / (2 * WAVES);     // rope dims per wave in each half
  constexpr int RS = 2 * PR / 4;          // rope dims per lane quarter = rope contraction steps per wave
  constexpr int PITCH = Q + 4;            // words per token row of the transposed-read image
  static_assert(Q % 16 == 0 && PR >= 2 && PR == 2 * RS, "shape");
  __shared__ __attribute__((aligned(16))) float s_part[2][WAVES][WAVES > 1 ? 256 : 4];   // (unused by one wave)
  __shared__ __attribute__((aligned(16))) float s_v[WAVES][16 * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  int ht, sp, row;
  split_block_index(blockIdx.x, p.n_ht, p.nsplit, p.rows, ht, sp, row);
  const int seq = p.row_seq ? p.row_seq[row] : row;
  const int ctx_all = max(0, min(p.ctx[row], p.max_ctx));
  const int tok0 = sp * p.tps;                             // this split's tokens: [tok0, ctx)
  if (p.nsplit > 1 && tok0 >= ctx_all) return;             // the merge skips splits past the context
  const int ctx = min(ctx_all, tok0 + p.tps);
  const int head = ht * 16 + c, head_c = min(head, p.H - 1);

  // rope dims of this lane: rbase .. rbase + RS - 1 (step s of the contraction takes dim rbase + s);
  // lane quarter g ^ 2 holds their rotation partners
  const bool rlo = g < 2;
  const int rbase = (rlo ? 0 : P / 2) + PR * wave + (g & 1) * RS;
  const float rsign = rlo ? -1.f : 1.f;

  // query fragments (B operands): lane (c = head, g) holds q[head][Q w + 16 j + 4 g + e] and the rope dims above
  const float* qrow = p.q + (int64_t)row * p.q_stride_row + (int64_t)head_c * (R + P);
  f32x4v ql[NJ];
  float qr[RS];
#pragma unroll
  for (int j = 0; j < NJ; ++j) ql[j] = *reinterpret_cast<const f32x4v*>(qrow + Q * wave + 16 * j + 4 * g);
  load_run<RS, VEC>(qrow + R + rbase, qr);

  f32x4v o[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) o[j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  float m_run = -1.0e30f, l_run = 0.f;

  const float* ckv_seq = p.ckv + (PAGED ? 0 : (int64_t)seq * p.ckv_stride_seq) + Q * wave;
  const float* kr_seq = p.krope + (PAGED ? 0 : (int64_t)seq * p.krope_stride_seq);
  const int32_t* bt_row = PAGED ? p.bt + (int64_t)seq * p.max_blocks : nullptr;

  // A tile in registers: this lane's token row (clamped to the context; rows past it are masked below)
  struct Tile {
    f32x4v k[NJ];
    float v[RS], cs[RS], sn[RS];
  };
  // Paged caches: the block id of this lane's row of the NEXT tile to be loaded (load_tile is
  // called for consecutive tiles), fetched a call ahead and before that call's data loads, so that
  // waiting for the id never waits for a tile in flight: loads return in order, and a table read
  // in the address path of a tile would drain the tiles prefetched before it.
  const int last_key = max(ctx - 1, 0);
  int blk_ahead = 0;
  if constexpr (PAGED) blk_ahead = bt_row[min(tok0 + c, last_key) >> p.bs_shift];
  auto load_tile = [&](int t0, Tile& x) __attribute__((always_inline)) {
    const int key = min(t0 + c, last_key);                  // (t0 counts from token 0 of the sequence)
    int64_t ckv_at = (int64_t)key * p.ckv_stride_tok, kr_at = (int64_t)key * p.krope_stride_tok;
    if constexpr (PAGED) {
      const int blk = min(max(blk_ahead, 0), p.num_blocks - 1), r = key & p.bs_mask;
      blk_ahead = bt_row[min(t0 + 16 + c, last_key) >> p.bs_shift];
      ckv_at = (int64_t)blk * p.ckv_stride_blk + (int64_t)r * p.ckv_stride_tok;
      kr_at = (int64_t)blk * p.krope_stride_blk + (int64_t)r * p.krope_stride_tok;
    }
    const float* krow = ckv_seq + ckv_at;
#pragma unroll
    for (int j = 0; j < NJ; ++j) x.k[j] = *reinterpret_cast<const f32x4v*>(krow + 16 * j + 4 * g);
    const float* rrow = kr_seq + kr_at;
    const int trow = min(key, p.table_len - 1) * P;
    load_run<RS, VEC>(rrow + rbase, x.v);
    load_run<RS, VEC>(p.cos_t + trow + rbase, x.cs);
    load_run<RS, VEC>(p.sin_t + trow + rbase, x.sn);
  };

  const int ntiles = ctx > tok0 ? (ctx - tok0 + 15) / 16 : 0;
  // register tiles in rotation: three for the four-wave kernels (44 registers per tile at rank 512 / rope 64:
  // 256 VGPRs, no scratch); the one-wave kernels hold whole rows (up to 80 registers per tile) and keep two
  constexpr int kTiles = WAVES == 4 ? MI355_MLA_F32_TILES : 2;
  float* my_v = s_v[wave];
  f32x4v sc_ahead = {0.f, 0.f, 0.f, 0.f};    // one wave: the scores computed one tile ahead stay in registers

  // This wave's share of a tile's scores: 16 tokens x 16 heads over its latent quarter and rope
  // dims, on two accumulators (one chain of dependent matrix instructions would run at their
  // latency, not at their issue rate).  Four waves: the share goes to the tile's exchange buffer.
  auto scores = [&](int t, const Tile& x) __attribute__((always_inline)) {
    f32x4v sp = {0.f, 0.f, 0.f, 0.f}, sp2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; j += 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sp = mma4(x.k[j][e], ql[j][e], sp);
        if (j + 1 < NJ) sp2 = mma4(x.k[j + 1][e], ql[j + 1][e], sp2);
      }
    }
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      float lower, upper;
      half_values(x.v[s], lower, upper);
      const float partner = rlo ? upper : lower;       // the same token's dim rbase + s -+ P/2
      const float roped = __builtin_fmaf(rsign * partner, x.sn[s], x.v[s] * x.cs[s]);
      if (s & 1) sp2 = mma4(roped, qr[s], sp2);
      else sp = mma4(roped, qr[s], sp);
    }
    sp += sp2;
    if constexpr (WAVES > 1) *reinterpret_cast<f32x4v*>(&s_part[t & 1][wave][lane * 4]) = sp;
    else sc_ahead = sp;
  };
  // a quarter tile, row-major, for the transposed reads of the value product
  auto stash = [&](const Tile& x) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4v*>(my_v + c * PITCH + 16 * j + 4 * g) = x.k[j];
  };

  // Software pipeline: the scores of tile t + 1 are issued before the softmax and the value product
  // of tile t, so the exchange of a tile's partial scores (LDS write, barrier, LDS reads) has a whole
  // value product between its two ends, and after every barrier a wave has matrix work ready.
  // Entering step t: `ready` = tile t + 1, the other register tile = tile t + 2 (in flight), my_v =
  // tile t, exchange buffer t & 1 = the partial scores of tile t (behind a barrier).  At its end the
  // step refills `ready` with tile t + 3, and the two register tiles swap roles: a tile's loads have
  // a whole step before they are waited for, and nothing is copied (with a copy at the loop's back
  // edge the compiler loads into temporaries and waits for the loads it has just issued).
  // A step has no branches: tiles past the last one are loaded (their rows clamp to the last token),
  // scored and stashed like the others; a step on such a tile sees every token masked and adds
  // nothing, which lets the loop run in pairs of steps.
  auto step = [&](int t, Tile& ready) __attribute__((always_inline)) {
    const int t0 = tok0 + t * 16;
    f32x4v sc = sc_ahead;
    if constexpr (WAVES > 1) {
      sc = *reinterpret_cast<const f32x4v*>(&s_part[t & 1][0][lane * 4]);
#pragma unroll
      for (int w = 1; w < WAVES; ++w) sc += *reinterpret_cast<const f32x4v*>(&s_part[t & 1][w][lane * 4]);
    }
    scores(t + 1, ready);
    // ---- online softmax; lane (c, g) holds tokens t0 + 4 g + r of head c
    float v[4], mx = -__builtin_inff();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = (t0 + 4 * g + r < ctx) ? sc[r] * p.scale_log2e : -__builtin_inff();
      mx = fmaxf(mx, v[r]);
    }
    mx = quarters_max(mx);
    const float m_new = fmaxf(m_run, mx);
    const float f = fast_exp2(m_run - m_new);
    m_run = m_new;
    float pr[4], ps = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pr[r] = fast_exp2(v[r] - m_new);
      ps += pr[r];
    }
    ps = quarters_sum(ps);
    l_run = l_run * f + ps;
#pragma unroll
    for (int j = 0; j < NJ; ++j) o[j] *= f;
    // ---- out^T += V^T . P^T over this wave's latent quarter.  The A operands come from the LDS
    // image kVAhead matrix instructions ahead of their use and the order is pinned: left alone, the
    // compiler reads two values into the same registers, waits, issues their two instructions, and
    // so on -- every pair of matrix instructions behind a full LDS round trip.
    {
      constexpr int NV = 4 * NJ, kVAhead = NV < 8 ? NV : 8;
      auto a_at = [&](int n) __attribute__((always_inline)) {
        return my_v[(4 * g + n / NJ) * PITCH + 16 * (n % NJ) + c];
      };
      float abuf[kVAhead];
#pragma unroll
      for (int n = 0; n < kVAhead; ++n) abuf[n] = a_at(n);
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        const float a = abuf[n % kVAhead];
        if (n + kVAhead < NV) abuf[n % kVAhead] = a_at(n + kVAhead);
        o[n % NJ] = mma4(a, pr[n / NJ], o[n % NJ]);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, kVAhead, 0);
#pragma unroll
      for (int n = 0; n < NV - kVAhead; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, kVAhead, 0);
    }
    stash(ready);                      // tile t + 1 (after this wave's reads of tile t: LDS runs a wave's accesses in order)
    load_tile(t0 + 16 * (kTiles + 1), ready);   // tile t + kTiles + 1
    if constexpr (WAVES > 1) __syncthreads();
  };
  if (ntiles > 0) {
    Tile tl[kTiles];
    load_tile(tok0, tl[kTiles - 1]);
#pragma unroll
    for (int i = 0; i < kTiles - 1; ++i) load_tile(tok0 + 16 * (i + 1), tl[i]);
    scores(0, tl[kTiles - 1]);
    stash(tl[kTiles - 1]);
    load_tile(tok0 + 16 * kTiles, tl[kTiles - 1]);
    if constexpr (WAVES > 1) __syncthreads();
    for (int t = 0; t < ntiles; t += kTiles) {
#pragma unroll
      for (int i = 0; i < kTiles; ++i) step(t + i, tl[i]);
    }
  }

  if (head < p.H) {
    if (p.nsplit == 1) {
      const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
      float* dst = p.out + ((int64_t)row * p.H + head) * R + Q * wave;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        *reinterpret_cast<f32x4v*>(dst + 16 * j + 4 * g) = o[j] * inv;   // rows (dims) 16 j + 4 g + r of head c
    } else {
      const int64_t rec = ((int64_t)row * p.H + head) * p.nsplit + sp;
      float* dst = p.ws_acc + rec * R + Q * wave;
#pragma unroll
      for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4v*>(dst + 16 * j + 4 * g) = o[j];
      if (wave == 0 && g == 0) {
        p.ws_ml[rec * 2] = m_run;
        p.ws_ml[rec * 2 + 1] = l_run;
      }
    }
  }
}

template <int R, int P, int WAVES>
void launch_rp(const MlaF32Params& p, dim3 g, bool vec, hipStream_t s) {
  if (p.bt) {
    if (vec) hipLaunchKernelGGL((mla_decode_f32_kernel<R, P, WAVES, true, true>), g, dim3(WAVES * 64), 0, s, p);
    else hipLaunchKernelGGL((mla_decode_f32_kernel<R, P, WAVES, false, true>), g, dim3(WAVES * 64), 0, s, p);
  } else if (vec) hipLaunchKernelGGL((mla_decode_f32_kernel<R, P, WAVES, true, false>), g, dim3(WAVES * 64), 0, s, p);
  else hipLaunchKernelGGL((mla_decode_f32_kernel<R, P, WAVES, false, false>), g, dim3(WAVES * 64), 0, s, p);
}

// vec: rope rows, q's rope part and the cos / sin tables may be read in 16-byte pieces
template <int R>
int launch_r(const MlaF32Params& p, int P, bool vec, hipStream_t s) {
  const int64_t grid = (int64_t)p.rows * p.nsplit * p.n_ht;
  if (grid >= (1LL << 31)) return fail(MI355_EINVAL, "mla_decode (fp32): grid too large");
  const dim3 g((unsigned)grid);
  if constexpr (R >= 256) {
    if (P == 32) launch_rp<R, 32, 4>(p, g, vec, s);
    else launch_rp<R, 64, 4>(p, g, vec, s);
  } else {
    switch (P) {
      case 8: launch_rp<R, 8, 1>(p, g, vec, s); break;
      case 16: launch_rp<R, 16, 1>(p, g, vec, s); break;
      case 32: launch_rp<R, 32, 1>(p, g, vec, s); break;
      default: launch_rp<R, 64, 1>(p, g, vec, s); break;
    }
  }
  if (int rc = check_launch("mla_decode_f32_kernel")) return rc;
  if (p.nsplit > 1)
    return mla_merge_launch(p.out, p.ws_acc, p.ws_ml, p.ctx, p.rows, p.H, R, p.max_ctx, p.nsplit, p.tps, s);
  return 0;
}

}  // namespace

// Token splits of the fp32 kernel: aim for two workgroups per CU, at least 256 tokens per split.
void mla_f32_plan(int rows, int H, int max_context, int* nsplit, int* tps, size_t* ws_bytes, int R) {
  const int64_t base = (int64_t)rows * ((H + 15) / 16);
  const int ctx = max_context > 0 ? max_context : 1;
  int64_t want = ceil_div64(512, base > 0 ? base : 1);
  const int64_t most = ctx / 256 > 0 ? ctx / 256 : 1;
  if (want > most) want = most;
  if (want < 1) want = 1;
  *tps = (int)(ceil_div64(ceil_div64(ctx, want), 16) * 16);
  *nsplit = (int)ceil_div64(ctx, *tps);
  *ws_bytes = *nsplit > 1 ? (size_t)rows * H * *nsplit * (R + 2) * sizeof(float) : 0;
}

// kv_lora_rank 32 / 64 / 128 with rope dim 8 / 16 / 32 / 64 (one wave), 256 / 512 with 32 / 64 (four waves)
bool mla_f32_supported(int R, int P) {
  if (R == 32 || R == 64 || R == 128) return P == 8 || P == 16 || P == 32 || P == 64;
  return (R == 256 || R == 512) && (P == 32 || P == 64);
}

int mla_f32_launch(float* out, const float* q, int64_t q_stride_row, const float* c_kv, int64_t ckv_stride_seq,
                   int64_t ckv_stride_tok, const float* k_rope, int64_t krope_stride_seq, int64_t krope_stride_tok,
                   const float* cos_table, const float* sin_table, int table_len, const int32_t* context_lens,
                   const int32_t* row_seq, int rows, int H, int R, int P, int max_context, float scale,
                   void* workspace, size_t workspace_bytes, hipStream_t stream, const int32_t* block_tables,
                   int block_size, int max_blocks, int num_blocks, int64_t ckv_stride_blk, int64_t krope_stride_blk) {
  MlaF32Params p;
  if (block_tables) {   // (block_size is a power of two: checked by the caller)
    p.bt = block_tables;
    p.bs_shift = ilog2_exact(block_size);
    p.bs_mask = block_size - 1;
    p.max_blocks = max_blocks;
    p.num_blocks = num_blocks;
    p.ckv_stride_blk = ckv_stride_blk;
    p.krope_stride_blk = krope_stride_blk;
  }
  size_t need = 0;
  mla_f32_plan(rows, H, max_context, &p.nsplit, &p.tps, &need, R);
  if (need && (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)))
    return fail(MI355_EWORKSPACE, "mla_decode (fp32): need %zu workspace bytes (16-B aligned), got %zu", need,
                workspace ? workspace_bytes : (size_t)0);
  p.ws_acc = static_cast<float*>(workspace);
  p.ws_ml = p.ws_acc ? p.ws_acc + (size_t)rows * H * p.nsplit * R : nullptr;
  p.out = out;
  p.q = q;
  p.ckv = c_kv;
  p.krope = k_rope;
  p.cos_t = cos_table;
  p.sin_t = sin_table;
  p.ctx = context_lens;
  p.row_seq = row_seq;
  p.q_stride_row = q_stride_row;
  p.ckv_stride_seq = ckv_stride_seq;
  p.ckv_stride_tok = ckv_stride_tok;
  p.krope_stride_seq = krope_stride_seq;
  p.krope_stride_tok = krope_stride_tok;
  p.rows = rows;
  p.H = H;
  p.n_ht = (H + 15) / 16;
  p.table_len = table_len;
  p.max_ctx = max_context;
  p.scale_log2e = scale * 1.4426950408889634f;
  // (q rows are 16-byte aligned by the caller's checks and R is a multiple of 4, so q's rope part is too)
  const bool vec = !(((uintptr_t)k_rope | (uintptr_t)cos_table | (uintptr_t)sin_table) & 15) &&
                   krope_stride_tok % 4 == 0 && (block_tables ? krope_stride_blk % 4 == 0 : krope_stride_seq % 4 == 0);
  switch (R) {
    case 32: return launch_r<32>(p, P, vec, stream);
    case 64: return launch_r<64>(p, P, vec, stream);
    case 128: return launch_r<128>(p, P, vec, stream);
    case 256: return launch_r<256>(p, P, vec, stream);
    case 512: return launch_r<512>(p, P, vec, stream);
  }
  return fail(MI355_EUNSUPPORTED, "mla_decode (fp32): kv_lora_rank %d", R);
}

}  // namespace mi355
