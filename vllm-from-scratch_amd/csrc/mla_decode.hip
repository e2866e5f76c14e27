// Latent-space (weight-absorbed) MLA attention on the matrix cores.
//
// Multi-head Latent Attention caches, per token, one low-rank vector c_kv [R] shared by
// all heads plus a small un-rotated RoPE key k_rope [P].  With the up-projections
// absorbed into the query and the output (done by the caller with dense GEMMs), the
// attention core is multi-query attention over rows [c_kv | rope(k_rope, pos)]:
//
//   score[h][t] = scale * ( q_lat[h] . c_kv[t]  +  q_rope[h] . rope(k_rope[t], t) )
//   o_lat[h]    = softmax_t(score[h]) . c_kv[t]                     (value = the same c_kv)
//
// Every cached row is read ONCE for all heads, and H heads x (R + P) dims of work per row
// is a real dense contraction, so this runs on v_mfma_f32_16x16x32:
//
//   workgroup = one wave = (query row, token split, tile of 16 heads); per step 16 tokens:
//   rows are staged in LDS (k_rope rotated in place with the cos/sin of the token's
//   position), S^T[token][head] = K . Q^T takes (R+P)/32 MFMAs with the 16 token rows as
//   A and the query fragments (registers) as B; the softmax statistics of a head are
//   lane-local plus two cross-lane steps; P^T goes straight from the accumulator registers
//   into the B operand of O^T[d][head] += V^T . P^T, whose A operand (4 tokens x 16 dims
//   per 16-lane group) is read from the same LDS rows with the transposing
//   ds_read_b64_tr_b16.  Memory latency is hidden by occupancy (8 one-wave workgroups per
//   CU, ~18 KiB of loads each), not by software pipelining.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "mla_common.h"

namespace mi355 {
namespace {

// Element offset of token `tok`'s row from the cache base: dense rows of one sequence (the base
// already points at the sequence), or a row of the block the sequence's table names.
template <int PAGED>
struct RowAt {
  const MlaParams& p;
  const int32_t* bt_row;
  __device__ __forceinline__ RowAt(const MlaParams& p_, int seq) : p(p_), bt_row(is_paged(PAGED) ? p_.bt + (int64_t)seq * p_.max_blocks : nullptr) {}
  __device__ __forceinline__ int block(int tok) const {
    return min(max(bt_row[tok >> p.bs_shift], 0), p.num_blocks - 1);
  }
  __device__ __forceinline__ int64_t ckv(int tok) const {
    if constexpr (is_paged(PAGED)) return (int64_t)block(tok) * p.ckv_stride_blk + (int64_t)(tok & p.bs_mask) * p.ckv_stride_tok;
    else return (int64_t)tok * p.ckv_stride_tok;
  }
  __device__ __forceinline__ int64_t krope(int tok) const {
    if constexpr (is_paged(PAGED)) return (int64_t)block(tok) * p.krope_stride_blk + (int64_t)(tok & p.bs_mask) * p.krope_stride_tok;
    else return (int64_t)tok * p.krope_stride_tok;
  }
};

// kPagedGroups: each wave of the four-wave kernels stages 8 CONSECUTIVE rows of a 32-token tile
// (one row = one 64-lane load), and with blocks of >= 8 tokens those 8 rows lie in one block.
// The table lookup, the clamp and the 64-bit address of the group's first row are then computed
// once per wave and tile on the scalar unit, and a row adds a 32-bit offset.  (With one wave per SIMD nothing hides
// address arithmetic: per-row addressing, ~15 scalar instructions for each of 8 rows, measured
// +15 % on the whole kernel at 256 rows x 4096 tokens; grouped: +3 %.)  Dense caches keep the
// other row order (wave w stages rows w, w + 4, ...: the four waves' loads of one moment cover
// 4 adjacent rows), which measures 0.5 % (16 heads) to 3 % (128 heads) faster than groups there.
// Rows past the sequence's end re-read its last row, as in the other modes.
struct GroupAt {
  const char* ckv_base = nullptr;
  const char* krope_base = nullptr;
  int n_last = 0;   // rows of the group before its last valid one
  __device__ __forceinline__ GroupAt(const MlaParams& p, int mode, const char* ckv, const char* krope, int ids, int t0,
                                     int t_end, int wave_u) {
    if (!is_grouped(mode)) return;
    const int last = t_end - 1;
    const int gs = min(t0 + 8 * wave_u, last & ~7);      // first row of the group, pulled back to the last live group
    n_last = min(last - gs, 7);
    const int id = min(max(__builtin_amdgcn_readlane(ids, (gs >> p.bs_shift) - (t0 >> p.bs_shift)), 0), p.num_blocks - 1);
    const int r0 = gs & p.bs_mask;
    ckv_base = ckv + ((int64_t)id * p.ckv_stride_blk + (int64_t)r0 * p.ckv_stride_tok) * 2;
    krope_base = krope + ((int64_t)id * p.krope_stride_blk + (int64_t)r0 * p.krope_stride_tok) * 2;
  }
  __device__ __forceinline__ int row_bytes(int i, int stride_tok) const { return min(i, n_last) * (stride_tok * 2); }
};

template <class Tag>
struct Mma16;
template <>
struct Mma16<BF16> {
  static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                   c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    return BF16::round1(lo) | (BF16::round1(hi) << 16);
  }
  static __device__ __forceinline__ uint16_t from_f(float x) { return (uint16_t)BF16::round1(x); }
};
template <>
struct Mma16<F16> {
  static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c,
                                                  0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    f16x2_t h;
    h[0] = (_Float16)lo;
    h[1] = (_Float16)hi;
    return __builtin_bit_cast(uint32_t, h);
  }
  static __device__ __forceinline__ uint16_t from_f(float x) {
    return __builtin_bit_cast(uint16_t, (_Float16)x);
  }
};

template <class Tag, int R, int PP, int PAGED = kDense>
__global__ void __launch_bounds__(kWave) mla_decode_kernel(const MlaParams p) {
  constexpr int DK = R + PP;              // key dims incl. zero padding of the rope part
  constexpr int KS = DK / 32;             // MFMA k-steps of the score product
  constexpr int OT = R / 16;              // 16-dim tiles of the latent output
  constexpr int CPR = DK / 8;             // 16-byte chunks per staged row
  constexpr int ROWB = DK * 2 + 16;       // LDS row pitch (16 B of padding staggers the banks)
  constexpr int NCH = kTok * CPR;         // chunks per step
  __shared__ __attribute__((aligned(16))) unsigned char tile[kTok * ROWB];

  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  int ht, sp, row;
  split_block_index(blockIdx.x, p.n_ht, p.nsplit, p.rows, ht, sp, row);
  const int seq = p.row_seq ? p.row_seq[row] : row;
  const int ctx = max(0, min(p.ctx[row], p.max_ctx));
  const int t_begin = sp * p.tps;
  if (p.nsplit > 1 && t_begin >= ctx) return;
  const int t_end = min(ctx, t_begin + p.tps);
  const int head = ht * kHeadTile + c;          // this lane's query head (column of S^T / O^T)
  const bool head_ok = head < p.H;

  // Query fragments: B operand, lane (c,g) holds Q[head c][32*ks + 8*g .. +8); zero beyond R+P and for padded heads.
  u32x4 qf[KS];
  {
    const char* qrow = static_cast<const char*>(p.q) + ((int64_t)row * p.q_stride_row + (int64_t)head * (R + p.P)) * 2;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d = 32 * ks + 8 * g;
      const bool ok = head_ok && (d < R || d - R < p.P);
      qf[ks] = ok ? *reinterpret_cast<const u32x4*>(qrow + d * 2) : u32x4{0u, 0u, 0u, 0u};
    }
  }

  f32x4 o[OT];
#pragma unroll
  for (int ot = 0; ot < OT; ++ot) o[ot] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1.0e30f, l_run = 0.f;

  const char* ckv = static_cast<const char*>(p.ckv) + (is_paged(PAGED) ? 0 : (int64_t)seq * p.ckv_stride_seq * 2);
  const char* krope = static_cast<const char*>(p.krope) + (is_paged(PAGED) ? 0 : (int64_t)seq * p.krope_stride_seq * 2);
  const RowAt<PAGED> at(p, seq);
  const int half = p.P >> 1;

  for (int t0 = t_begin; t0 < t_end; t0 += kTok) {
    // ---- stage 16 rows [c_kv | k_rope | 0-pad] in LDS
#pragma unroll 6
    for (int ch = lane; ch < NCH; ch += kWave) {
      const int tt = ch / CPR, dc = ch - tt * CPR;
      const int tok = min(t0 + tt, t_end - 1);        // rows past the end re-read the last one (masked below)
      const int d = dc * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (d < R) v = load16<true>(ckv + (at.ckv(tok) + d) * 2);
      else if (d - R < p.P) v = load16<true>(krope + (at.krope(tok) + (d - R)) * 2);
      *reinterpret_cast<u32x4*>(tile + tt * ROWB + dc * 16) = v;
    }
    __syncthreads();
    // ---- rotate the rope part in place: token c, position = its index in the sequence
    {
      const int pos = min(min(t0 + c, t_end - 1), p.table_len - 1);
      typename Tag::elem_t* kr = reinterpret_cast<typename Tag::elem_t*>(tile + c * ROWB + R * 2);
      const float* cs = p.cos_t + (int64_t)pos * p.P;
      const float* sn = p.sin_t + (int64_t)pos * p.P;
      for (int i0 = 8 * g; i0 < half; i0 += 32) {
        for (int e = 0; e < 8 && i0 + e < half; ++e) {
          const int i = i0 + e;
          const float lo = Tag::to_float(kr[i]), hi = Tag::to_float(kr[i + half]);
          const float co = cs[i], si = sn[i];
          kr[i] = Mma16<Tag>::from_f(lo * co - hi * si);
          kr[i + half] = Mma16<Tag>::from_f(hi * co + lo * si);
        }
      }
    }
    __syncthreads();
    // ---- S^T[token][head] = K . Q^T : A = 16 token rows, lane (r=c, g) reads row c, dims 32*ks + 8*g .. +8
    f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(tile + c * ROWB + (32 * ks + 8 * g) * 2);
      st = Mma16<Tag>::run(a, qf[ks], st);
    }
    // lane (c,g) now holds S^T[token 4g + j][head c], j = 0..3
    float s[4], mx = -__builtin_inff();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = (t0 + 4 * g + j < t_end) ? st[j] * p.scale_log2e : -__builtin_inff();
      mx = fmaxf(mx, s[j]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float f = fast_exp2(m_run - m_new);
    m_run = m_new;
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = fast_exp2(s[j] - m_new);
      ps += s[j];
    }
    ps += __shfl_xor(ps, 16);
    ps += __shfl_xor(ps, 32);
    l_run = l_run * f + ps;
    if (!__all(f == 1.0f)) {
#pragma unroll
      for (int ot = 0; ot < OT; ++ot) o[ot] *= f;
    }
    // ---- O^T[d][head] += V^T . P^T.  MFMA k-slot (g, j) carries token 4g + j for j < 4 and
    // nothing for j >= 4 (both operands zero there), so P^T is the accumulator registers as
    // they stand and V^T is one transposing 4-row read per 16-dim tile.
    const u32x4 pb = {Mma16<Tag>::pack2(s[0], s[1]), Mma16<Tag>::pack2(s[2], s[3]), 0u, 0u};
    const int qi = c >> 2, pi = c & 3;
    const unsigned char* vrow = tile + (4 * g + qi) * ROWB + (4 * pi) * 2;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(vrow + ot * 32));
      const u32x2 t2 = __builtin_bit_cast(u32x2, t);
      const u32x4 va = {t2[0], t2[1], 0u, 0u};
      o[ot] = Mma16<Tag>::run(va, pb, o[ot]);
    }
    __syncthreads();   // the rows are overwritten by the next step
  }

  if (!head_ok) return;
  // lane (c,g) holds O[head c][16*ot + 4*g + j]
  if (p.nsplit == 1) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* dst = p.out + ((int64_t)row * p.H + head) * R + 4 * g;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) *reinterpret_cast<f32x4*>(dst + 16 * ot) = o[ot] * inv;
  } else {
    const int64_t rec = ((int64_t)row * p.H + head) * p.nsplit + sp;
    float* dst = p.ws_acc + rec * R + 4 * g;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) *reinterpret_cast<f32x4*>(dst + 16 * ot) = o[ot];
    if (g == 0) {
      p.ws_ml[rec * 2] = m_run;
      p.ws_ml[rec * 2 + 1] = l_run;
    }
  }
}

// Four-wave variant for kv_lora_rank >= 128.  The workgroup shares 32-token tiles through a
// double-buffered LDS image fed from two register stages (the global loads of the next TWO
// tiles are in flight while the current one is reduced).  The score product is split four
// ways: a wave multiplies one half of the R + P key dims for one 16-token half of the tile,
// and the two partial 16 x 16 tiles of each half are summed through LDS, so a wave keeps half
// of the query fragments in registers and reads a quarter of the tile; each wave then owns a
// quarter of the latent output dims for all 32 tokens.  Registers and LDS stay low enough for
// 2 workgroups (8 waves) per CU.
template <class Tag, int R, int PP, int PAGED = kDense>
__global__ void __launch_bounds__(256, 2) mla_decode_kernel_w4(const MlaParams p) {
  constexpr int DK = R + PP;
  constexpr int KS = DK / 32;
  constexpr int KSW = (KS + 1) / 2;       // k-steps of the score product per wave
  constexpr int OTW = R / 64;             // 16-dim output tiles per wave
  constexpr int CPR = DK / 8;
  // 16 B of padding staggers the banks; the conflict-free pitch of the many-heads kernel below
  // (+32 B) would push two workgroups' tiles past the CU's LDS and measures slower here.
  constexpr int ROWB = DK * 2 + 16;
  constexpr int CCH = R / 8;              // 16-byte chunks of the c_kv part of a row
  constexpr int RSTEP = 256 / CCH;        // rows covered by one pass of the 256 threads over c_kv chunks
  constexpr int NCKV = kTok2 / RSTEP;     // c_kv loads per thread per tile
  constexpr int RCH = PP / 8;             // chunks of the (padded) k_rope part of a row
  constexpr int NLOAD = NCKV + 1;         // + one k_rope chunk per thread
  constexpr int MAXPAIR = 4;              // rope pairs per thread: 4 consecutive dims (P <= 64 -> half <= 32 = 8 x 4)
  static_assert(kTok2 * RCH <= 256 && CPR == CCH + RCH, "tile staging layout");
  __shared__ __attribute__((aligned(16))) unsigned char tiles[2][kTok2 * ROWB];
  __shared__ f32x4 part[4][kWave];        // per-wave partial scores of the current tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
  int ht, sp, row;
  split_block_index(blockIdx.x, p.n_ht, p.nsplit, p.rows, ht, sp, row);
  const int seq = p.row_seq ? p.row_seq[row] : row;
  const int ctx = max(0, min(p.ctx[row], p.max_ctx));
  const int t_begin = sp * p.tps;
  if (p.nsplit > 1 && t_begin >= ctx) return;
  const int t_end = min(ctx, t_begin + p.tps);
  const int head = ht * kHeadTile + c;
  const bool head_ok = head < p.H;

  // Query fragments of this wave's slice [ks0, ks1) of the key dims; it scores tile rows 16*su .. +16.
  const int su = wave >> 1;
  const int ks0 = ((wave & 1) * KS) / 2, ks1 = (((wave & 1) + 1) * KS) / 2;
  u32x4 qf[KSW];
  {
    const char* qrow = static_cast<const char*>(p.q) + ((int64_t)row * p.q_stride_row + (int64_t)head * (R + p.P)) * 2;
#pragma unroll
    for (int i = 0; i < KSW; ++i) {
      const int d = 32 * (ks0 + i) + 8 * g;
      const bool ok = head_ok && ks0 + i < ks1 && (d < R || d - R < p.P);
      qf[i] = ok ? *reinterpret_cast<const u32x4*>(qrow + d * 2) : u32x4{0u, 0u, 0u, 0u};
    }
  }
  f32x4 o[OTW];
#pragma unroll
  for (int i = 0; i < OTW; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1.0e30f, l_run = 0.f;

  const char* ckv = static_cast<const char*>(p.ckv) + (is_paged(PAGED) ? 0 : (int64_t)seq * p.ckv_stride_seq * 2);
  const char* krope = static_cast<const char*>(p.krope) + (is_paged(PAGED) ? 0 : (int64_t)seq * p.krope_stride_seq * 2);
  const RowAt<PAGED> at(p, seq);
  const int half = p.P >> 1;
  // rope rotation: thread = (token of the tile, group of 4 consecutive pairs), so that the 8 lanes
  // of a token read one contiguous 128-byte run of the cos / sin row
  const int rtok = tid >> 3, rgrp = tid & 7;
  const bool r_live = 4 * rgrp < half;
  const int cstride = (int)p.ckv_stride_tok, rstride = (int)p.krope_stride_tok;   // host checks S * stride < 2^30

  // Staging layout: thread -> one fixed c_kv column chunk for rows crow, crow + RSTEP, ... and
  // one k_rope chunk (row krow, chunk kcol), so the per-load state is one row index.
  const int ccol = tid % CCH, crow = tid / CCH;
  const int kcol = tid % RCH, krow = tid / RCH;
  const bool k_live = kcol * 8 < p.P;
  const int kcol_ld = k_live ? kcol : 0;
  u32x4 stage_a[NLOAD], stage_b[NLOAD];
  float rc[MAXPAIR], rs[MAXPAIR];
  // Paged caches: the block ids of a 32-token tile, lane l holding the id of the tile's l-th block
  // (at most 32 of them), fetched a tile before the loads that need them so that no load of the
  // steady state waits for a table entry (a dependent table read in the address path would sit
  // behind the prefetched tiles in the in-order load queue and drain it).
  auto load_ids = [&](int t0) {
    if constexpr (is_paged(PAGED)) return at.bt_row[min((t0 >> p.bs_shift) + (lane & 31), p.max_blocks - 1)];
    else return 0;
  };
  auto issue_loads = [&](int t0, u32x4 (&stage)[NLOAD], int ids) {
    const int blk0 = t0 >> p.bs_shift;
    const GroupAt grp(p, PAGED, ckv, krope, ids, t0, t_end, __builtin_amdgcn_readfirstlane(wave));
#pragma unroll
    for (int i = 0; i < NCKV; ++i) {
      const int tok = min(t0 + crow + RSTEP * i, t_end - 1);   // rows past the end re-read the last one (masked later)
      if constexpr (is_grouped(PAGED)) {
        stage[i] = load16<true>(grp.ckv_base + (uint32_t)(grp.row_bytes(i, cstride) + ccol * 16));
      } else if constexpr (is_paged(PAGED) && CCH % kWave == 0) {
        // a wave's load is one token row (kv_lora_rank 512: 64 lanes x 16 bytes): the row, its block id
        // and the row's base address are wave-uniform, i.e. scalar work, and the load takes the
        // scalar base plus the lane's fixed column offset
        const int tok_u = __builtin_amdgcn_readfirstlane(tok);
        const int id = min(max(__builtin_amdgcn_readlane(ids, (tok_u >> p.bs_shift) - blk0), 0), p.num_blocks - 1);
        const char* rowp = ckv + ((int64_t)id * p.ckv_stride_blk + (int64_t)(tok_u & p.bs_mask) * cstride) * 2;
        stage[i] = load16<true>(rowp + (uint32_t)(ccol * 16));
      } else if constexpr (is_paged(PAGED)) {
        const int id = min(max(__shfl(ids, (tok >> p.bs_shift) - blk0), 0), p.num_blocks - 1);
        stage[i] = load16<true>(ckv + ((int64_t)id * p.ckv_stride_blk + (int64_t)((tok & p.bs_mask) * cstride + ccol * 8)) * 2);
      } else {
        // 32-bit byte offsets from the (wave-uniform) sequence base keep the per-load address state small
        stage[i] = load16<true>(ckv + (uint32_t)((tok * cstride + ccol * 8) * 2));
      }
    }
    // unconditional (address clamped into the row) so the compiler can count outstanding loads
    // exactly; lanes beyond the rope width are zeroed when the stage is written to LDS
    const int tok = min(t0 + min(krow, kTok2 - 1), t_end - 1);
    if constexpr (is_grouped(PAGED)) {
      // (rope slot 64: thread tid stages chunk tid % 8 of row tid / 8, a row of its own wave's group)
      stage[NCKV] = load16<true>(grp.krope_base + (uint32_t)(min(lane >> 3, grp.n_last) * (rstride * 2) + kcol_ld * 16));
    } else if constexpr (is_paged(PAGED)) {
      const int id = min(max(__shfl(ids, (tok >> p.bs_shift) - blk0), 0), p.num_blocks - 1);
      stage[NCKV] = load16<true>(krope + ((int64_t)id * p.krope_stride_blk + (int64_t)((tok & p.bs_mask) * rstride + kcol_ld * 8)) * 2);
    } else {
      stage[NCKV] = load16<true>(krope + (uint32_t)((tok * rstride + kcol_ld * 8) * 2));
    }
  };
  // cos / sin of the rope pairs this thread rotates in tile t0.  Issued BEFORE the data loads
  // of the tile after it, so waiting for them never waits for those.
  auto issue_rope_factors = [&](int t0) {
    const int pos = min(min(t0 + rtok, t_end - 1), p.table_len - 1);
    const int col = r_live ? 4 * rgrp : 0;   // unconditional loads; groups beyond the rope width are never rotated
    const f32x4 cv = *reinterpret_cast<const f32x4*>(p.cos_t + (int64_t)pos * p.P + col);
    const f32x4 sv = *reinterpret_cast<const f32x4*>(p.sin_t + (int64_t)pos * p.P + col);
#pragma unroll
    for (int k = 0; k < MAXPAIR; ++k) {
      rc[k] = cv[k];
      rs[k] = sv[k];
    }
  };

  const int ntiles = t_end > t_begin ? (t_end - t_begin + kTok2 - 1) / kTok2 : 0;

  // One tile: registers -> LDS, rotate the rope part, refill the same registers with the tile
  // two steps ahead, then scores / softmax / weighted sum from LDS.
  auto process = [&](int t, u32x4 (&stage)[NLOAD], int& ids_mine, int& ids_other, auto steady) {
    const int t0 = t_begin + t * kTok2;
    unsigned char* tile = tiles[t & 1];
#pragma unroll
    for (int i = 0; i < NCKV; ++i)
      *reinterpret_cast<u32x4*>(tile + (is_grouped(PAGED) ? crow * NCKV + i : crow + RSTEP * i) * ROWB + ccol * 16) = stage[i];
    if (krow < kTok2)
      *reinterpret_cast<u32x4*>(tile + krow * ROWB + R * 2 + kcol * 16) = k_live ? stage[NCKV] : u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    if (r_live) {   // rotate the rope part of the 32 rows in place: 4 (lo, hi) pairs per thread, 8-byte accesses
      unsigned char* kr = tile + rtok * ROWB + R * 2 + rgrp * 8;
      const u32x2 lo2 = *reinterpret_cast<const u32x2*>(kr), hi2 = *reinterpret_cast<const u32x2*>(kr + half * 2);
      float lo[4], hi[4], nlo[4], nhi[4];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        lo[2 * k] = Tag::to_float((typename Tag::elem_t)(lo2[k] & 0xffffu));
        lo[2 * k + 1] = Tag::to_float((typename Tag::elem_t)(lo2[k] >> 16));
        hi[2 * k] = Tag::to_float((typename Tag::elem_t)(hi2[k] & 0xffffu));
        hi[2 * k + 1] = Tag::to_float((typename Tag::elem_t)(hi2[k] >> 16));
      }
#pragma unroll
      for (int k = 0; k < MAXPAIR; ++k) {
        nlo[k] = lo[k] * rc[k] - hi[k] * rs[k];
        nhi[k] = hi[k] * rc[k] + lo[k] * rs[k];
      }
      *reinterpret_cast<u32x2*>(kr) = u32x2{Mma16<Tag>::pack2(nlo[0], nlo[1]), Mma16<Tag>::pack2(nlo[2], nlo[3])};
      *reinterpret_cast<u32x2*>(kr + half * 2) = u32x2{Mma16<Tag>::pack2(nhi[0], nhi[1]), Mma16<Tag>::pack2(nhi[2], nhi[3])};
    }
    __syncthreads();
    // (ids_mine: the ids of tile t + 2, fetched a step ago; the ids of tile t + 3 go out ahead of
    // tile t + 2's loads, so that the next step's wait for them does not wait for those loads)
    if constexpr (decltype(steady)::value) {   // main loop: both refills exist, no branches around the loads
      issue_rope_factors(t0 + kTok2);
      ids_other = load_ids(t0 + 3 * kTok2);
      issue_loads(t0 + 2 * kTok2, stage, ids_mine);
    } else {
      if (t + 1 < ntiles) issue_rope_factors(t0 + kTok2);
      if (t + 3 < ntiles) ids_other = load_ids(t0 + 3 * kTok2);
      if (t + 2 < ntiles) issue_loads(t0 + 2 * kTok2, stage, ids_mine);
    }

    // partial S^T (16 tokens x 16 heads over half of the key dims), exchanged through LDS
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < KSW; ++i) {
        // a step past this wave's slice (odd KS) multiplies by a zero query fragment
        const int ks = min(ks0 + i, KS - 1);
        const u32x4 a = *reinterpret_cast<const u32x4*>(tile + (16 * su + c) * ROWB + (32 * ks + 8 * g) * 2);
        acc = Mma16<Tag>::run(a, qf[i], acc);
      }
      part[wave][lane] = acc;
    }
    __syncthreads();
    f32x4 st[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) st[u] = part[2 * u][lane] + part[2 * u + 1][lane];

    float s[8], mx = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = (t0 + 16 * u + 4 * g + j < t_end) ? st[u][j] * p.scale_log2e : -__builtin_inff();
        s[4 * u + j] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float f = fast_exp2(m_run - m_new);
    m_run = m_new;
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s[j] = fast_exp2(s[j] - m_new);
      ps += s[j];
    }
    ps += __shfl_xor(ps, 16);
    ps += __shfl_xor(ps, 32);
    l_run = l_run * f + ps;
    if (!__all(f == 1.0f)) {
#pragma unroll
      for (int i = 0; i < OTW; ++i) o[i] *= f;
    }
    // k-slot (g, j): token 4g + j of the first 16 rows for j < 4, token 16 + 4g + (j - 4) for j >= 4
    const u32x4 pb = {Mma16<Tag>::pack2(s[0], s[1]), Mma16<Tag>::pack2(s[2], s[3]),
                      Mma16<Tag>::pack2(s[4], s[5]), Mma16<Tag>::pack2(s[6], s[7])};
    const int qi = c >> 2, pi = c & 3;
    const unsigned char* vrow = tile + (4 * g + qi) * ROWB + (4 * pi) * 2 + wave * OTW * 32;
#pragma unroll
    for (int i = 0; i < OTW; ++i) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(vrow + i * 32));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(vrow + i * 32 + 16 * ROWB));
      const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
      const u32x4 va = {lo2[0], lo2[1], hi2[0], hi2[1]};
      o[i] = Mma16<Tag>::run(va, pb, o[i]);
    }
  };

  // Issue order in the steady state is data(t+2), factors(t+3), data(t+3), ...; the prologue
  // follows the same order so the compiler's outstanding-load counts agree on both loop edges.
  int t = 0;
  int ids_a = load_ids(t_begin), ids_b = load_ids(t_begin + kTok2);   // (paged caches; see load_ids)
  if (ntiles >= 4) {
    issue_loads(t_begin, stage_a, ids_a);
    ids_a = load_ids(t_begin + 2 * kTok2);
    issue_rope_factors(t_begin);
    issue_loads(t_begin + kTok2, stage_b, ids_b);
    do {
      process(t, stage_a, ids_a, ids_b, std::true_type{});
      process(t + 1, stage_b, ids_b, ids_a, std::true_type{});
      t += 2;
    } while (t + 3 < ntiles);
  } else if (ntiles > 0) {
    issue_loads(t_begin, stage_a, ids_a);
    if (ntiles > 2) ids_a = load_ids(t_begin + 2 * kTok2);
    issue_rope_factors(t_begin);
    if (ntiles > 1) issue_loads(t_begin + kTok2, stage_b, ids_b);
  }
  for (; t < ntiles; t += 2) {   // last one to three tiles
    process(t, stage_a, ids_a, ids_b, std::false_type{});
    if (t + 1 < ntiles) process(t + 1, stage_b, ids_b, ids_a, std::false_type{});
  }

  if (!head_ok) return;
  const int d_base = wave * OTW * 16 + 4 * g;       // lane holds O[head][d_base + 16*i + j]
  if (p.nsplit == 1) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* dst = p.out + ((int64_t)row * p.H + head) * R + d_base;
#pragma unroll
    for (int i = 0; i < OTW; ++i) *reinterpret_cast<f32x4*>(dst + 16 * i) = o[i] * inv;
  } else {
    const int64_t rec = ((int64_t)row * p.H + head) * p.nsplit + sp;
    float* dst = p.ws_acc + rec * R + d_base;
#pragma unroll
    for (int i = 0; i < OTW; ++i) *reinterpret_cast<f32x4*>(dst + 16 * i) = o[i];
    if (wave == 0 && g == 0) {
      p.ws_ml[rec * 2] = m_run;
      p.ws_ml[rec * 2 + 1] = l_run;
    }
  }
}

// Many-heads variant (more than 16 heads, kv_lora_rank >= 128): the four waves share the staged
// tiles exactly as above, but each wave owns its own 16 heads -- all of their query fragments,
// scores and latent outputs -- so a cached row is read once per 64 heads instead of once per 16.
// A wave then holds 72 + 128 registers of fragments and accumulators at R = 512: one workgroup
// (one wave per SIMD) per CU, with the two-tile register prefetch covering the memory latency.
// Per 32-token tile a wave issues 2 x KS + R / 16 matrix instructions (68 at R = 512, about a
// third of the time the tile's 36 KiB take to arrive at the CU's share of HBM bandwidth).
template <class Tag, int R, int PP, int PAGED = kDense>
__global__ void __launch_bounds__(256) mla_decode_kernel_mh(const MlaParams p) {
  constexpr int DK = R + PP;
  constexpr int KS = DK / 32;
  constexpr int OT = R / 16;              // 16-dim tiles of the latent output
  constexpr int CPR = DK / 8;
  // Row pitch = an odd multiple of 32 bytes.  The transposing 8-byte reads of V^T are served 32
  // lanes = 8 rows x 32 bytes at a time: the rows' 8-bank windows then tile the 64 banks.  The
  // 16-byte reads of K are served 16 lanes at a time, 8 rows at one chunk and 8 other rows at the
  // next chunk: with this pitch the first 8 land on the even 4-bank groups, the others on the odd.
  constexpr int ROWB = DK * 2 + 32;
  static_assert((ROWB / 32) % 2 == 1, "row pitch must be an odd multiple of 32 bytes");
  constexpr int CCH = R / 8;              // 16-byte chunks of the c_kv part of a row
  constexpr int RSTEP = 256 / CCH;        // rows covered by one pass of the 256 threads over c_kv chunks
  constexpr int NCKV = kTok2 / RSTEP;     // c_kv loads per thread per tile
  constexpr int RCH = PP / 8;             // chunks of the (padded) k_rope part of a row
  constexpr int NLOAD = NCKV + 1;         // + one k_rope chunk per thread
  constexpr int MAXPAIR = 4;              // rope pairs per thread: 4 consecutive dims (P <= 64 -> half <= 32 = 8 x 4)
  static_assert(kTok2 * RCH <= 256 && CPR == CCH + RCH, "tile staging layout");
  __shared__ __attribute__((aligned(16))) unsigned char tiles[2][kTok2 * ROWB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
  int ht, sp, row;
  split_block_index(blockIdx.x, p.n_ht, p.nsplit, p.rows, ht, sp, row);
  const int seq = p.row_seq ? p.row_seq[row] : row;
  const int ctx = max(0, min(p.ctx[row], p.max_ctx));
  const int t_begin = sp * p.tps;
  if (p.nsplit > 1 && t_begin >= ctx) return;
  const int t_end = min(ctx, t_begin + p.tps);
  const int head = (ht * 4 + wave) * kHeadTile + c;          // a wave's 16 heads
  const bool head_ok = head < p.H;
  const bool wave_live = (ht * 4 + wave) * kHeadTile < p.H;  // else: the wave only helps staging the tiles

  // Query fragments: B operand, lane (c,g) holds Q[head c][32*ks + 8*g .. +8); zero beyond R+P and for padded heads.
  u32x4 qf[KS];
  {
    const char* qrow = static_cast<const char*>(p.q) + ((int64_t)row * p.q_stride_row + (int64_t)head * (R + p.P)) * 2;
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int d = 32 * i + 8 * g;
      const bool ok = head_ok && (d < R || d - R < p.P);
      qf[i] = ok ? *reinterpret_cast<const u32x4*>(qrow + d * 2) : u32x4{0u, 0u, 0u, 0u};
    }
  }
  f32x4 o[OT];
#pragma unroll
  for (int i = 0; i < OT; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1.0e30f, l_run = 0.f;

  const char* ckv = static_cast<const char*>(p.ckv) + (is_paged(PAGED) ? 0 : (int64_t)seq * p.ckv_stride_seq * 2);
  const char* krope = static_cast<const char*>(p.krope) + (is_paged(PAGED) ? 0 : (int64_t)seq * p.krope_stride_seq * 2);
  const RowAt<PAGED> at(p, seq);
  const int half = p.P >> 1;
  // rope rotation: thread = (token of the tile, group of 4 consecutive pairs), so that the 8 lanes
  // of a token read one contiguous 128-byte run of the cos / sin row
  const int rtok = tid >> 3, rgrp = tid & 7;
  const bool r_live = 4 * rgrp < half;
  const int cstride = (int)p.ckv_stride_tok, rstride = (int)p.krope_stride_tok;   // host checks S * stride < 2^30

  // Staging layout: thread -> one fixed c_kv column chunk for rows crow, crow + RSTEP, ... and
  // one k_rope chunk (row krow, chunk kcol), so the per-load state is one row index.
  const int ccol = tid % CCH, crow = tid / CCH;
  const int kcol = tid % RCH, krow = tid / RCH;
  const bool k_live = kcol * 8 < p.P;
  const int kcol_ld = k_live ? kcol : 0;
  u32x4 stage_a[NLOAD], stage_b[NLOAD];
  float rc[MAXPAIR], rs[MAXPAIR];
  // Paged caches: the block ids of a 32-token tile, lane l holding the id of the tile's l-th block
  // (at most 32 of them), fetched a tile before the loads that need them so that no load of the
  // steady state waits for a table entry (a dependent table read in the address path would sit
  // behind the prefetched tiles in the in-order load queue and drain it).
  auto load_ids = [&](int t0) {
    if constexpr (is_paged(PAGED)) return at.bt_row[min((t0 >> p.bs_shift) + (lane & 31), p.max_blocks - 1)];
    else return 0;
  };
  auto issue_loads = [&](int t0, u32x4 (&stage)[NLOAD], int ids) {
    const int blk0 = t0 >> p.bs_shift;
    const GroupAt grp(p, PAGED, ckv, krope, ids, t0, t_end, __builtin_amdgcn_readfirstlane(wave));
#pragma unroll
    for (int i = 0; i < NCKV; ++i) {
      const int tok = min(t0 + crow + RSTEP * i, t_end - 1);   // rows past the end re-read the last one (masked later)
      if constexpr (is_grouped(PAGED)) {
        stage[i] = load16<true>(grp.ckv_base + (uint32_t)(grp.row_bytes(i, cstride) + ccol * 16));
      } else if constexpr (is_paged(PAGED) && CCH % kWave == 0) {
        // a wave's load is one token row (kv_lora_rank 512: 64 lanes x 16 bytes): the row, its block id
        // and the row's base address are wave-uniform, i.e. scalar work, and the load takes the
        // scalar base plus the lane's fixed column offset
        const int tok_u = __builtin_amdgcn_readfirstlane(tok);
        const int id = min(max(__builtin_amdgcn_readlane(ids, (tok_u >> p.bs_shift) - blk0), 0), p.num_blocks - 1);
        const char* rowp = ckv + ((int64_t)id * p.ckv_stride_blk + (int64_t)(tok_u & p.bs_mask) * cstride) * 2;
        stage[i] = load16<true>(rowp + (uint32_t)(ccol * 16));
      } else if constexpr (is_paged(PAGED)) {
        const int id = min(max(__shfl(ids, (tok >> p.bs_shift) - blk0), 0), p.num_blocks - 1);
        stage[i] = load16<true>(ckv + ((int64_t)id * p.ckv_stride_blk + (int64_t)((tok & p.bs_mask) * cstride + ccol * 8)) * 2);
      } else {
        // 32-bit byte offsets from the (wave-uniform) sequence base keep the per-load address state small
        stage[i] = load16<true>(ckv + (uint32_t)((tok * cstride + ccol * 8) * 2));
      }
    }
    // unconditional (address clamped into the row) so the compiler can count outstanding loads
    // exactly; lanes beyond the rope width are zeroed when the stage is written to LDS
    const int tok = min(t0 + min(krow, kTok2 - 1), t_end - 1);
    if constexpr (is_grouped(PAGED)) {
      // (rope slot 64: thread tid stages chunk tid % 8 of row tid / 8, a row of its own wave's group)
      stage[NCKV] = load16<true>(grp.krope_base + (uint32_t)(min(lane >> 3, grp.n_last) * (rstride * 2) + kcol_ld * 16));
    } else if constexpr (is_paged(PAGED)) {
      const int id = min(max(__shfl(ids, (tok >> p.bs_shift) - blk0), 0), p.num_blocks - 1);
      stage[NCKV] = load16<true>(krope + ((int64_t)id * p.krope_stride_blk + (int64_t)((tok & p.bs_mask) * rstride + kcol_ld * 8)) * 2);
    } else {
      stage[NCKV] = load16<true>(krope + (uint32_t)((tok * rstride + kcol_ld * 8) * 2));
    }
  };
  // cos / sin of the rope pairs this thread rotates in tile t0.  Issued BEFORE the data loads
  // of the tile after it, so waiting for them never waits for those.
  auto issue_rope_factors = [&](int t0) {
    const int pos = min(min(t0 + rtok, t_end - 1), p.table_len - 1);
    const int col = r_live ? 4 * rgrp : 0;   // unconditional loads; groups beyond the rope width are never rotated
    const f32x4 cv = *reinterpret_cast<const f32x4*>(p.cos_t + (int64_t)pos * p.P + col);
    const f32x4 sv = *reinterpret_cast<const f32x4*>(p.sin_t + (int64_t)pos * p.P + col);
#pragma unroll
    for (int k = 0; k < MAXPAIR; ++k) {
      rc[k] = cv[k];
      rs[k] = sv[k];
    }
  };

  const int ntiles = t_end > t_begin ? (t_end - t_begin + kTok2 - 1) / kTok2 : 0;

  // One tile: registers -> LDS, rotate the rope part, refill the same registers with the tile
  // two steps ahead, then scores / softmax / weighted sum from LDS.
  auto process = [&](int t, u32x4 (&stage)[NLOAD], int& ids_mine, int& ids_other, auto steady) {
    const int t0 = t_begin + t * kTok2;
    unsigned char* tile = tiles[t & 1];
#pragma unroll
    for (int i = 0; i < NCKV; ++i)
      *reinterpret_cast<u32x4*>(tile + (is_grouped(PAGED) ? crow * NCKV + i : crow + RSTEP * i) * ROWB + ccol * 16) = stage[i];
    if (krow < kTok2)
      *reinterpret_cast<u32x4*>(tile + krow * ROWB + R * 2 + kcol * 16) = k_live ? stage[NCKV] : u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    if (r_live) {   // rotate the rope part of the 32 rows in place: 4 (lo, hi) pairs per thread, 8-byte accesses
      unsigned char* kr = tile + rtok * ROWB + R * 2 + rgrp * 8;
      const u32x2 lo2 = *reinterpret_cast<const u32x2*>(kr), hi2 = *reinterpret_cast<const u32x2*>(kr + half * 2);
      float lo[4], hi[4], nlo[4], nhi[4];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        lo[2 * k] = Tag::to_float((typename Tag::elem_t)(lo2[k] & 0xffffu));
        lo[2 * k + 1] = Tag::to_float((typename Tag::elem_t)(lo2[k] >> 16));
        hi[2 * k] = Tag::to_float((typename Tag::elem_t)(hi2[k] & 0xffffu));
        hi[2 * k + 1] = Tag::to_float((typename Tag::elem_t)(hi2[k] >> 16));
      }
#pragma unroll
      for (int k = 0; k < MAXPAIR; ++k) {
        nlo[k] = lo[k] * rc[k] - hi[k] * rs[k];
        nhi[k] = hi[k] * rc[k] + lo[k] * rs[k];
      }
      *reinterpret_cast<u32x2*>(kr) = u32x2{Mma16<Tag>::pack2(nlo[0], nlo[1]), Mma16<Tag>::pack2(nlo[2], nlo[3])};
      *reinterpret_cast<u32x2*>(kr + half * 2) = u32x2{Mma16<Tag>::pack2(nhi[0], nhi[1]), Mma16<Tag>::pack2(nhi[2], nhi[3])};
    }
    __syncthreads();
    // (ids_mine: the ids of tile t + 2, fetched a step ago; the ids of tile t + 3 go out ahead of
    // tile t + 2's loads, so that the next step's wait for them does not wait for those loads)
    if constexpr (decltype(steady)::value) {   // main loop: both refills exist, no branches around the loads
      issue_rope_factors(t0 + kTok2);
      ids_other = load_ids(t0 + 3 * kTok2);
      issue_loads(t0 + 2 * kTok2, stage, ids_mine);
    } else {
      if (t + 1 < ntiles) issue_rope_factors(t0 + kTok2);
      if (t + 3 < ntiles) ids_other = load_ids(t0 + 3 * kTok2);
      if (t + 2 < ntiles) issue_loads(t0 + 2 * kTok2, stage, ids_mine);
    }

    if (!wave_live) return;
    // S^T[token][head] = K . Q^T for the two 16-token halves of the tile.  The A operands (token
    // rows from LDS) are fetched kAhead matrix instructions ahead of their use; left to itself
    // the compiler, short of registers in this kernel, reuses one register quad and waits for
    // every 16-byte read right before the instruction that consumes it.
    f32x4 st[2];
    {
      constexpr int kAhead = 6, NA = 2 * KS;
      const unsigned char* arow = tile + c * ROWB + 8 * g * 2;
      auto a_at = [&](int n) __attribute__((always_inline)) {   // n-th operand: half n / KS, k-step n % KS
        return *reinterpret_cast<const u32x4*>(arow + (n / KS) * 16 * ROWB + (n % KS) * 64);
      };
      u32x4 abuf[kAhead];
#pragma unroll
      for (int n = 0; n < kAhead; ++n) abuf[n] = a_at(n);
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int n = 0; n < NA; ++n) {
        const u32x4 a = abuf[n % kAhead];
        if (n + kAhead < NA) abuf[n % kAhead] = a_at(n + kAhead);
        acc[n / KS] = Mma16<Tag>::run(a, qf[n % KS], acc[n / KS]);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
      for (int n = 0; n < NA - kAhead; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, kAhead, 0);
      st[0] = acc[0];
      st[1] = acc[1];
    }

    float s[8], mx = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = (t0 + 16 * u + 4 * g + j < t_end) ? st[u][j] * p.scale_log2e : -__builtin_inff();
        s[4 * u + j] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    // Deferred rescale.  m_run is the reference the exponentials are taken against, not
    // necessarily the running maximum: it only moves when some head's maximum has outgrown it by
    // more than kDefer (in log2 units), so weights stay below 2^kDefer and the 128 accumulator
    // registers of a wave (parked in AccVGPRs) are rescaled on a few tiles per sequence instead
    // of on every tile in which any of the 16 heads sees a new maximum.  Sums, outputs and the
    // split records are all relative to the same m_run, so the result is the exact softmax.
    constexpr float kDefer = 8.f;
    if (__builtin_amdgcn_ballot_w64(mx > m_run + kDefer) != 0) {
      const float m_new = fmaxf(m_run, mx);
      const float f = fast_exp2(m_run - m_new);
      m_run = m_new;
      l_run *= f;
#pragma unroll
      for (int i = 0; i < OT; ++i) o[i] *= f;
    }
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s[j] = fast_exp2(s[j] - m_run);
      ps += s[j];
    }
    ps += __shfl_xor(ps, 16);
    ps += __shfl_xor(ps, 32);
    l_run += ps;
    // k-slot (g, j): token 4g + j of the first 16 rows for j < 4, token 16 + 4g + (j - 4) for j >= 4
    const u32x4 pb = {Mma16<Tag>::pack2(s[0], s[1]), Mma16<Tag>::pack2(s[2], s[3]),
                      Mma16<Tag>::pack2(s[4], s[5]), Mma16<Tag>::pack2(s[6], s[7])};
    const int qi = c >> 2, pi = c & 3;
    const unsigned char* vrow = tile + (4 * g + qi) * ROWB + (4 * pi) * 2;
#pragma unroll
    for (int i = 0; i < OT; ++i) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(vrow + i * 32));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(vrow + i * 32 + 16 * ROWB));
      const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
      const u32x4 va = {lo2[0], lo2[1], hi2[0], hi2[1]};
      o[i] = Mma16<Tag>::run(va, pb, o[i]);
    }
  };

  // Issue order in the steady state is data(t+2), factors(t+3), data(t+3), ...; the prologue
  // follows the same order so the compiler's outstanding-load counts agree on both loop edges.
  int t = 0;
  int ids_a = load_ids(t_begin), ids_b = load_ids(t_begin + kTok2);   // (paged caches; see load_ids)
  if (ntiles >= 4) {
    issue_loads(t_begin, stage_a, ids_a);
    ids_a = load_ids(t_begin + 2 * kTok2);
    issue_rope_factors(t_begin);
    issue_loads(t_begin + kTok2, stage_b, ids_b);
    do {
      process(t, stage_a, ids_a, ids_b, std::true_type{});
      process(t + 1, stage_b, ids_b, ids_a, std::true_type{});
      t += 2;
    } while (t + 3 < ntiles);
  } else if (ntiles > 0) {
    issue_loads(t_begin, stage_a, ids_a);
    if (ntiles > 2) ids_a = load_ids(t_begin + 2 * kTok2);
    issue_rope_factors(t_begin);
    if (ntiles > 1) issue_loads(t_begin + kTok2, stage_b, ids_b);
  }
  for (; t < ntiles; t += 2) {   // last one to three tiles
    process(t, stage_a, ids_a, ids_b, std::false_type{});
    if (t + 1 < ntiles) process(t + 1, stage_b, ids_b, ids_a, std::false_type{});
  }

  if (!head_ok) return;
  const int d_base = 4 * g;                         // lane holds O[head][d_base + 16*i + j]
  if (p.nsplit == 1) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* dst = p.out + ((int64_t)row * p.H + head) * R + d_base;
#pragma unroll
    for (int i = 0; i < OT; ++i) *reinterpret_cast<f32x4*>(dst + 16 * i) = o[i] * inv;
  } else {
    const int64_t rec = ((int64_t)row * p.H + head) * p.nsplit + sp;
    float* dst = p.ws_acc + rec * R + d_base;
#pragma unroll
    for (int i = 0; i < OT; ++i) *reinterpret_cast<f32x4*>(dst + 16 * i) = o[i];
    if (g == 0) {
      p.ws_ml[rec * 2] = m_run;
      p.ws_ml[rec * 2 + 1] = l_run;
    }
  }
}

template <int R>
__global__ void __launch_bounds__(256) mla_merge_kernel(const MlaParams p) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)p.rows * p.H * (R / 4);
  if (idx >= total) return;
  const int dv = (int)(idx % (R / 4));
  const int64_t rh = idx / (R / 4);
  const int row = (int)(rh / p.H);
  const int ctx = max(0, min(p.ctx[row], p.max_ctx));
  const int nvalid = min(p.nsplit, (ctx + p.tps - 1) / p.tps);
  const float* ml = p.ws_ml + rh * p.nsplit * 2;
  const float* pa = p.ws_acc + rh * p.nsplit * R + dv * 4;
  float M = -1.0e30f;
  for (int s = 0; s < nvalid; ++s) M = fmaxf(M, ml[2 * s]);
  float den = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < nvalid; ++s) {
    const float f = fast_exp2(ml[2 * s] - M);
    den = __builtin_fmaf(ml[2 * s + 1], f, den);
    acc += *reinterpret_cast<const f32x4*>(pa + (size_t)s * R) * f;
  }
  const float inv = den > 0.f ? 1.f / den : 0.f;
  *reinterpret_cast<f32x4*>(p.out + rh * R + dv * 4) = acc * inv;
}

// The same combination for many splits of few rows (one long sequence spread over the chip): the
// kernel above gives a thread all the splits of its four dims, one after the other -- 16 384 threads
// walking 256 records each took 43 of the 77 us of 1 row x 128 heads x 32 768 tokens.  Here a
// workgroup owns (row, head, 128 dims) and its eight groups of 32 lanes each take every eighth
// split; the groups' sums meet in LDS in a fixed order.  Factors exp2(m_s - M) are worked out once per
// workgroup and kept in LDS (up to kMergeManyCap splits).
constexpr int kMergeManyCap = 2048;
constexpr int kMergeManyMin = 8;     // fewer splits: the kernel above

template <int R>
__global__ void __launch_bounds__(256) mla_merge_many_kernel(const MlaParams p) {
  constexpr int kChunks = R / 128;
  __shared__ float s_f[kMergeManyCap];
  __shared__ float s_wave[2][4];
  __shared__ f32x4 s_acc[8][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int dvl = tid & 31, grp = tid >> 5;
  const int chunk = (int)(blockIdx.x % kChunks);
  const int64_t rh = blockIdx.x / kChunks;
  const int row = (int)(rh / p.H);
  const int ctx = max(0, min(p.ctx[row], p.max_ctx));
  const int nvalid = min(p.nsplit, (ctx + p.tps - 1) / p.tps);
  const float* ml = p.ws_ml + rh * p.nsplit * 2;
  float m = -1.0e30f;
  for (int s = tid; s < nvalid; s += 256) m = fmaxf(m, ml[2 * s]);
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) m = fmaxf(m, __shfl_xor(m, off));
  if (lane == 0) s_wave[0][wave] = m;
  __syncthreads();
  const float M = fmaxf(fmaxf(s_wave[0][0], s_wave[0][1]), fmaxf(s_wave[0][2], s_wave[0][3]));
  float d = 0.f;
  for (int s = tid; s < nvalid; s += 256) {
    const float f = fast_exp2(ml[2 * s] - M);
    s_f[s] = f;
    d = __builtin_fmaf(ml[2 * s + 1], f, d);
  }
  d = group_sum<64>(d);
  if (lane == 0) s_wave[1][wave] = d;
  __syncthreads();
  const float den = (s_wave[1][0] + s_wave[1][1]) + (s_wave[1][2] + s_wave[1][3]);
  const float* pa = p.ws_acc + rh * p.nsplit * R + chunk * 128 + dvl * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int s = grp;
  for (; s + 24 < nvalid; s += 32) {      // four records in flight per thread
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(pa + (size_t)s * R);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(pa + (size_t)(s + 8) * R);
    const f32x4 a2 = *reinterpret_cast<const f32x4*>(pa + (size_t)(s + 16) * R);
    const f32x4 a3 = *reinterpret_cast<const f32x4*>(pa + (size_t)(s + 24) * R);
    acc += a0 * s_f[s];
    acc += a1 * s_f[s + 8];
    acc += a2 * s_f[s + 16];
    acc += a3 * s_f[s + 24];
  }
  for (; s < nvalid; s += 8) acc += *reinterpret_cast<const f32x4*>(pa + (size_t)s * R) * s_f[s];
  s_acc[grp][dvl] = acc;
  __syncthreads();
  if (grp == 0) {
    f32x4 total = s_acc[0][dvl];
#pragma unroll
    for (int g = 1; g < 8; ++g) total += s_acc[g][dvl];
    const float inv = den > 0.f ? 1.f / den : 0.f;
    *reinterpret_cast<f32x4*>(p.out + rh * R + chunk * 128 + dvl * 4) = total * inv;
  }
}

// Launches the combination of a launch's split records on `s`.
template <int R>
void launch_merge(const MlaParams& p, hipStream_t s) {
  if constexpr (R >= 128) {
    // (from about 512 workgroups of the kernel above on, its threads are enough to keep memory busy
    // and its single pass is faster: 32 rows x 128 heads x 8 splits 70.0 against 75.0 us)
    if (p.nsplit >= kMergeManyMin && p.nsplit <= kMergeManyCap && (int64_t)p.rows * p.H * (R / 4) <= 512 * 256) {
      hipLaunchKernelGGL((mla_merge_many_kernel<R>), dim3((unsigned)((int64_t)p.rows * p.H * (R / 128))), dim3(256), 0, s, p);
      return;
    }
  }
  const int64_t threads = (int64_t)p.rows * p.H * (R / 4);
  hipLaunchKernelGGL((mla_merge_kernel<R>), dim3((unsigned)ceil_div64(threads, 256)), dim3(256), 0, s, p);
}

struct MlaPlan {
  bool many;
  bool wide;   // 128 heads per workgroup (mla_decode_wide.hip)
  bool fused;  // ... and its two token splits merged inside the launch
  int skew;    // ... of which split 0 then takes this many tokens fewer than tps, split 1 the rest
  int n_ht, nsplit, tps;
  size_t ws_bytes;
};

int mla_plan(int rows, int H, int R, int P, int max_context, int dtype, MlaPlan* out) {
  if (rows <= 0 || H <= 0) return fail(MI355_EINVAL, "mla_decode: rows/H (%d,%d)", rows, H);
  if (dtype != MI355_BF16 && dtype != MI355_F16)
    return fail(MI355_EUNSUPPORTED, "mla_decode: dtype %d (bf16 / f16 only)", dtype);
  if (R != 32 && R != 64 && R != 128 && R != 256 && R != 512)
    return fail(MI355_EUNSUPPORTED, "mla_decode: kv_lora_rank %d not in {32,64,128,256,512}", R);
  if (P <= 0 || P > 64 || P % 8) return fail(MI355_EUNSUPPORTED, "mla_decode: rope dim %d (multiple of 8, <= 64)", P);
  if (max_context < 0) return fail(MI355_EINVAL, "mla_decode: max_context %d", max_context);
  MlaPlan pl;
  pl.many = R >= 128 && H > kHeadTile && knob(kKnobMlaManyHeads) != 0;   // 64 heads per workgroup
  pl.wide = pl.many && mla_wide_supported(H, R, P, dtype);
  const int ctx = max_context > 0 ? max_context : 1;
  // one-wave kernel (R < 128): 8 workgroups per CU.  Four-wave kernel: one workgroup per CU
  // before tokens are split -- with two tiles of prefetch per workgroup, few long workgroups
  // beat many short ones (measured at 256 rows x 4096 tokens: 187 us unsplit, 193 us with 2
  // splits, 212 us with 4).
  const int target = knob(kKnobMlaTargetWgs) > 0 ? knob(kKnobMlaTargetWgs) : (R >= 128 ? 256 : 2048);
  auto split_tokens = [&]() {
    pl.n_ht = pl.wide ? (H + 127) / 128 : pl.many ? (H + 4 * kHeadTile - 1) / (4 * kHeadTile) : (H + kHeadTile - 1) / kHeadTile;
    int64_t want = ceil_div64(target, (int64_t)rows * pl.n_ht);
    // The four-wave kernels run one workgroup per CU: a 257th workgroup waits for a whole one to
    // finish, so the split count is rounded down (96 rows x 128 heads x 4096 tokens: 195 us in 3 splits,
    // 138 in 2; 64-head kernel, 24 rows: 109 -> 69 us; 16 heads, 48 / 96 rows: 65.5 -> 48.6 / 111.8 ->
    // 92.0).  The 16-head kernel is bound by memory, where a short second round costs less than
    // CUs left idle: it keeps two splits where rounding down would leave none (130 rows: 154.5
    // against 166.1 us unsplit; the 128-head kernel at 160 rows: 274 against 225 unsplit).
    if (R >= 128 && knob(kKnobMlaTargetWgs) <= 0) {
      const int64_t down = target / ((int64_t)rows * pl.n_ht);
      if (pl.many || down >= 2) want = down;
    }
    const int64_t most = ctx / 128 > 0 ? ctx / 128 : 1;           // at least 4 steps per workgroup
    if (want > most) want = most;
    if (want < 1) want = 1;
    pl.tps = (int)(ceil_div64(ceil_div64(ctx, want), kTok2) * kTok2);
    pl.nsplit = (int)ceil_div64(ctx, pl.tps);
  };
  split_tokens();
  // The 128-heads-per-workgroup kernel starts with three tiles of loads and the queries of 128 heads,
  // and with one head tile where the 64-head kernel has two it needs twice the token splits (and
  // twice the records to merge) to fill the chip: the smallest batches are the 64-head kernel's.
  // Measured at 128 heads x 4096 tokens, wide against 64-head (us): 24 rows (416 tokens per
  // workgroup) 58.4 / 69.3, 20 rows (352) 55.3 / 62.2, 16 rows (256) 52.3 / 52.1, 8 rows 46.2 / 41.3,
  // 1 row x 32768 tokens 123 / 77; 32 heads, 8 rows x 8192 (256 tokens per workgroup, the same splits
  // for both kernels): 39.8 / 41.9.
  if (pl.wide && pl.nsplit > 1) {
    const int least = knob(kKnobMlaWideSplitTokens) > 0 ? knob(kKnobMlaWideSplitTokens) : (H > 64 ? 320 : 256);
    if (pl.tps < least) {
      pl.wide = false;
      split_tokens();
    }
  }
  pl.ws_bytes = pl.nsplit > 1 ? (size_t)rows * H * pl.nsplit * (R + 2) * sizeof(float) : 0;
  // Two splits of the wide kernel: the later workgroup of a pair still holds its own partial in
  // registers, reads the other's record and writes the output row -- a third of the merge kernel's
  // traffic and no second launch.  (R + 2 is even: the cells behind the records are 8-byte aligned.)
  pl.fused = pl.wide && pl.nsplit == 2 && knob(kKnobMlaWideFusedMerge) != 0;
  if (pl.fused) pl.ws_bytes += (size_t)rows * pl.n_ht * sizeof(unsigned long long);
  if (pl.wide) pl.ws_bytes += mla_wide_stamp_bytes();   // (0 outside timing side builds)
  // Split 0 is given fewer tokens: its record is then in memory by the time split 1 is done, which
  // finds the pair's cell counted and combines without writing a record of its own.  All 256
  // workgroups writing records at once are bound by HBM (67 MB at 128 rows x 128 heads: 12 us).
  pl.skew = 0;
  if (pl.fused) {
    const int skew = (knob(kKnobMlaWideSkewTokens) + kTok2 - 1) / kTok2 * kTok2;
    if (skew > 0 && pl.tps >= 8 * skew) pl.skew = skew;
  }
  *out = pl;
  return 0;
}

template <int R>
int launch_tail(const MlaParams& p, hipStream_t s) {
  if (int rc = check_launch("mla_decode_kernel")) return rc;
  if (p.nsplit > 1) {
    launch_merge<R>(p, s);
    return check_launch("mla_merge_kernel");
  }
  return 0;
}

int launch_wide_then_merge(MlaParams& p, int R, int dtype, bool paged, bool fused, int skew, void* workspace, size_t ws_bytes, hipStream_t s) {
  if (fused) {
    p.tps -= skew;
    p.counters = reinterpret_cast<unsigned long long*>(p.ws_ml + (size_t)p.rows * p.H * p.nsplit * 2);
    p.epoch = next_launch_tag();
  }
  if (mla_wide_stamp_bytes()) p.stamps = reinterpret_cast<unsigned long long*>(static_cast<char*>(workspace) + ws_bytes - mla_wide_stamp_bytes());
  if (int rc = mla_wide_launch(&p, dtype, paged, s)) return rc;
  if (p.counters) return 0;   // merged inside the launch
  return R == 512 ? launch_tail<512>(p, s) : fail(MI355_EUNSUPPORTED, "mla_decode: wide kernel at rank %d", R);
}

template <class Tag, int R, int PAGED>
int launch_r(const MlaParams& p, bool many, hipStream_t s) {
  const int64_t grid = (int64_t)p.rows * p.nsplit * p.n_ht;
  if constexpr (R == 512 && PAGED == kPaged) {
    if (p.P > 32 && p.bs_shift >= 3 && p.ckv_stride_tok < (1 << 26) && p.krope_stride_tok < (1 << 26) &&
        knob(kKnobMlaRowGroups) != 0) {
      if (many) hipLaunchKernelGGL((mla_decode_kernel_mh<Tag, R, 64, kPagedGroups>), dim3((unsigned)grid), dim3(256), 0, s, p);
      else hipLaunchKernelGGL((mla_decode_kernel_w4<Tag, R, 64, kPagedGroups>), dim3((unsigned)grid), dim3(256), 0, s, p);
      return launch_tail<R>(p, s);
    }
  }
  if constexpr (R >= 128) {
    if (many) {
      if (p.P <= 32) hipLaunchKernelGGL((mla_decode_kernel_mh<Tag, R, 32, PAGED>), dim3((unsigned)grid), dim3(256), 0, s, p);
      else hipLaunchKernelGGL((mla_decode_kernel_mh<Tag, R, 64, PAGED>), dim3((unsigned)grid), dim3(256), 0, s, p);
    } else if (p.P <= 32) hipLaunchKernelGGL((mla_decode_kernel_w4<Tag, R, 32, PAGED>), dim3((unsigned)grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((mla_decode_kernel_w4<Tag, R, 64, PAGED>), dim3((unsigned)grid), dim3(256), 0, s, p);
  } else {
    if (p.P <= 32) hipLaunchKernelGGL((mla_decode_kernel<Tag, R, 32, PAGED>), dim3((unsigned)grid), dim3(kWave), 0, s, p);
    else hipLaunchKernelGGL((mla_decode_kernel<Tag, R, 64, PAGED>), dim3((unsigned)grid), dim3(kWave), 0, s, p);
  }
  return launch_tail<R>(p, s);
}

template <class Tag, int PAGED>
int launch_t(int R, const MlaParams& p, bool many, hipStream_t s) {
  switch (R) {
    case 32: return launch_r<Tag, 32, PAGED>(p, many, s);
    case 64: return launch_r<Tag, 64, PAGED>(p, many, s);
    case 128: return launch_r<Tag, 128, PAGED>(p, many, s);
    case 256: return launch_r<Tag, 256, PAGED>(p, many, s);
    case 512: return launch_r<Tag, 512, PAGED>(p, many, s);
  }
  return fail(MI355_EUNSUPPORTED, "mla_decode: kv_lora_rank %d", R);
}

}  // namespace

int mla_merge_launch(float* out, float* ws_acc, float* ws_ml, const int32_t* ctx, int rows, int H, int R,
                     int max_ctx, int nsplit, int tps, hipStream_t stream) {
  MlaParams p = {};
  p.out = out;
  p.ws_acc = ws_acc;
  p.ws_ml = ws_ml;
  p.ctx = ctx;
  p.rows = rows;
  p.H = H;
  p.max_ctx = max_ctx;
  p.nsplit = nsplit;
  p.tps = tps;
  switch (R) {
    case 32: launch_merge<32>(p, stream); break;
    case 64: launch_merge<64>(p, stream); break;
    case 128: launch_merge<128>(p, stream); break;
    case 256: launch_merge<256>(p, stream); break;
    case 512: launch_merge<512>(p, stream); break;
    default: return fail(MI355_EUNSUPPORTED, "mla merge: kv_lora_rank %d", R);
  }
  return check_launch("mla_merge_kernel");
}

}  // namespace mi355

using namespace mi355;

extern "C" size_t mi355_mla_decode_workspace_bytes(int32_t rows, int32_t H, int32_t R, int32_t P,
                                                   int32_t max_context, int32_t dtype) {
  if (dtype == MI355_F32) {
    int nsplit, tps;
    size_t need = 0;
    if (rows <= 0 || H <= 0 || !mla_f32_supported(R, P)) return 0;
    mla_f32_plan(rows, H, max_context, &nsplit, &tps, &need, R);
    return need;
  }
  MlaPlan pl;
  if (mla_plan(rows, H, R, P, max_context, dtype, &pl)) return 0;
  return pl.ws_bytes;
}

namespace mi355 {
namespace {
// Block-table addressing of mi355_mla_decode_paged (null table: dense per-sequence rows).
struct MlaPaging {
  const int32_t* bt = nullptr;
  int block_size = 0, max_blocks = 0, num_blocks = 0;
  int64_t ckv_stride_blk = 0, krope_stride_blk = 0;
};

int mla_decode_impl(const char* what, float* out, const void* q, int64_t q_stride_row, const void* c_kv,
                    int64_t ckv_stride_seq, int64_t ckv_stride_tok, const void* k_rope, int64_t krope_stride_seq,
                    int64_t krope_stride_tok, const float* cos_table, const float* sin_table, int32_t table_len,
                    const int32_t* context_lens, const int32_t* row_seq, int32_t rows, int32_t H, int32_t R,
                    int32_t P, int32_t max_context, float scale, int32_t dtype, void* workspace,
                    size_t workspace_bytes, void* stream, const MlaPaging& pg) {
  if (rows == 0) return 0;
  if (!out || !q || !c_kv || !k_rope || !cos_table || !sin_table || !context_lens)
    return fail(MI355_EINVAL, "%s: null pointer argument", what);
  const bool paged = pg.bt != nullptr;
  const int epl = dtype == MI355_F32 ? 4 : 8;
  if (paged) {
    if (pg.block_size <= 0 || ilog2_exact(pg.block_size) < 0)
      return fail(MI355_EUNSUPPORTED, "%s: block_size %d is not a power of two", what, pg.block_size);
    if (pg.max_blocks <= 0 || pg.num_blocks <= 0) return fail(MI355_EINVAL, "%s: max_blocks / num_blocks", what);
    if (pg.ckv_stride_blk <= 0 || pg.krope_stride_blk <= 0 || pg.ckv_stride_blk % epl || pg.krope_stride_blk % epl)
      return fail(MI355_EINVAL, "%s: block strides must keep rows 16-byte aligned", what);
    if ((int64_t)max_context > (int64_t)pg.max_blocks * pg.block_size) max_context = pg.max_blocks * pg.block_size;
  }
  if (dtype == MI355_F32) {
    if (rows < 0 || H <= 0 || max_context < 0 || table_len <= 0) return fail(MI355_EINVAL, "%s: bad shape", what);
    if (!mla_f32_supported(R, P))
      return fail(MI355_EUNSUPPORTED, "%s (fp32): kv_lora_rank %d / rope dim %d outside {32,64,128} x {8,16,32,64} and {256,512} x {32,64}",
                  what, R, P);
    if (q_stride_row % 4 || ckv_stride_tok % 4 || ckv_stride_seq % 4 || q_stride_row < (int64_t)H * (R + P) ||
        ckv_stride_tok < R || krope_stride_tok < P)
      return fail(MI355_EINVAL, "%s: strides must keep rows 16-byte aligned", what);
    if (((uintptr_t)out | (uintptr_t)q | (uintptr_t)c_kv) & 15)
      return fail(MI355_EINVAL, "%s: tensors must be 16-byte aligned", what);
    return mla_f32_launch(out, static_cast<const float*>(q), q_stride_row, static_cast<const float*>(c_kv),
                          ckv_stride_seq, ckv_stride_tok, static_cast<const float*>(k_rope), krope_stride_seq,
                          krope_stride_tok, cos_table, sin_table, table_len, context_lens, row_seq, rows, H, R, P,
                          max_context, scale, workspace, workspace_bytes, static_cast<hipStream_t>(stream), pg.bt,
                          pg.block_size, pg.max_blocks, pg.num_blocks, pg.ckv_stride_blk, pg.krope_stride_blk);
  }
  MlaPlan pl;
  if (int rc = mla_plan(rows, H, R, P, max_context, dtype, &pl)) return rc;
  if (table_len <= 0) return fail(MI355_EINVAL, "%s: empty cos/sin table", what);
  if (q_stride_row % 8 || ckv_stride_tok % 8 || ckv_stride_seq % 8 || krope_stride_tok % 8 || krope_stride_seq % 8 ||
      q_stride_row < (int64_t)H * (R + P) || ckv_stride_tok < R || krope_stride_tok < P)
    return fail(MI355_EINVAL, "%s: strides must keep rows 16-byte aligned", what);
  if (((uintptr_t)out | (uintptr_t)q | (uintptr_t)c_kv | (uintptr_t)k_rope | (uintptr_t)cos_table |
       (uintptr_t)sin_table) & 15)
    return fail(MI355_EINVAL, "%s: tensors must be 16-byte aligned", what);
  // (dense rows are addressed with 32-bit offsets from the sequence's base; blocks with 64-bit ones)
  if (!paged && ((int64_t)max_context * ckv_stride_tok >= (1LL << 30) || (int64_t)max_context * krope_stride_tok >= (1LL << 30)))
    return fail(MI355_EUNSUPPORTED, "%s: one sequence's cache must span < 2^30 elements", what);
  if (pl.ws_bytes && (!workspace || workspace_bytes < pl.ws_bytes || ((uintptr_t)workspace & 15)))
    return fail(MI355_EWORKSPACE, "%s: need %zu workspace bytes (16-B aligned), got %zu", what, pl.ws_bytes,
                workspace ? workspace_bytes : (size_t)0);
  MlaParams p;
  p.out = out;
  p.q = q;
  p.ckv = c_kv;
  p.krope = k_rope;
  p.cos_t = cos_table;
  p.sin_t = sin_table;
  p.ctx = context_lens;
  p.row_seq = row_seq;
  p.ws_acc = static_cast<float*>(workspace);
  p.ws_ml = p.ws_acc ? p.ws_acc + (size_t)rows * H * pl.nsplit * R : nullptr;
  p.q_stride_row = q_stride_row;
  p.ckv_stride_seq = ckv_stride_seq;
  p.ckv_stride_tok = ckv_stride_tok;
  p.krope_stride_seq = krope_stride_seq;
  p.krope_stride_tok = krope_stride_tok;
  p.rows = rows;
  p.H = H;
  p.P = P;
  p.table_len = table_len;
  p.max_ctx = max_context;
  p.n_ht = pl.n_ht;
  p.nsplit = pl.nsplit;
  p.tps = pl.tps;
  p.scale_log2e = scale * 1.4426950408889634f;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (paged) {
    p.bt = pg.bt;
    p.ckv_stride_blk = pg.ckv_stride_blk;
    p.krope_stride_blk = pg.krope_stride_blk;
    p.bs_shift = ilog2_exact(pg.block_size);
    p.bs_mask = pg.block_size - 1;
    p.max_blocks = pg.max_blocks;
    p.num_blocks = pg.num_blocks;
    // The wide kernel holds its token range's block ids in LDS and takes its row maxima on the
    // unscaled scores; a longer range (tiny blocks) or a scale that is not positive goes to the
    // 64-head kernel, with the same token splits and so the same workspace.
    if (pl.wide && scale > 0.f && p.tps + pl.skew < mla_wide_max_paged_tokens(pg.block_size) && ckv_stride_tok < (1 << 26) &&
        krope_stride_tok < (1 << 26))
      return launch_wide_then_merge(p, R, dtype, true, pl.fused, pl.skew, workspace, pl.ws_bytes, s);
    if (pl.wide) p.n_ht = (H + 4 * kHeadTile - 1) / (4 * kHeadTile);
    return dtype == MI355_BF16 ? launch_t<BF16, kPaged>(R, p, pl.many, s) : launch_t<F16, kPaged>(R, p, pl.many, s);
  }
  if (pl.wide && scale > 0.f) return launch_wide_then_merge(p, R, dtype, false, pl.fused, pl.skew, workspace, pl.ws_bytes, s);
  if (pl.wide) p.n_ht = (H + 4 * kHeadTile - 1) / (4 * kHeadTile);
  return dtype == MI355_BF16 ? launch_t<BF16, kDense>(R, p, pl.many, s) : launch_t<F16, kDense>(R, p, pl.many, s);
}
}  // namespace
}  // namespace mi355

extern "C" int mi355_mla_decode(float* out, const void* q, int64_t q_stride_row, const void* c_kv,
                                int64_t ckv_stride_seq, int64_t ckv_stride_tok, const void* k_rope,
                                int64_t krope_stride_seq, int64_t krope_stride_tok, const float* cos_table,
                                const float* sin_table, int32_t table_len, const int32_t* context_lens,
                                const int32_t* row_seq, int32_t rows, int32_t H, int32_t R, int32_t P,
                                int32_t max_context, float scale, int32_t dtype, void* workspace,
                                size_t workspace_bytes, void* stream) {
  return mla_decode_impl("mla_decode", out, q, q_stride_row, c_kv, ckv_stride_seq, ckv_stride_tok, k_rope,
                         krope_stride_seq, krope_stride_tok, cos_table, sin_table, table_len, context_lens, row_seq,
                         rows, H, R, P, max_context, scale, dtype, workspace, workspace_bytes, stream, MlaPaging{});
}

extern "C" int mi355_mla_decode_paged(float* out, const void* q, int64_t q_stride_row, const void* c_kv,
                                      int64_t ckv_stride_block, int64_t ckv_stride_tok, const void* k_rope,
                                      int64_t krope_stride_block, int64_t krope_stride_tok,
                                      const int32_t* block_tables, int32_t block_size, int32_t max_blocks,
                                      int32_t num_blocks, const float* cos_table, const float* sin_table,
                                      int32_t table_len, const int32_t* context_lens, const int32_t* row_seq,
                                      int32_t rows, int32_t H, int32_t R, int32_t P, int32_t max_context,
                                      float scale, int32_t dtype, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  if (rows == 0) return 0;
  if (!block_tables) return fail(MI355_EINVAL, "mla_decode_paged: null block table");
  MlaPaging pg;
  pg.bt = block_tables;
  pg.block_size = block_size;
  pg.max_blocks = max_blocks;
  pg.num_blocks = num_blocks;
  pg.ckv_stride_blk = ckv_stride_block;
  pg.krope_stride_blk = krope_stride_block;
  // (the per-sequence strides of the dense form are unused; 8 keeps their alignment checks quiet)
  return mla_decode_impl("mla_decode_paged", out, q, q_stride_row, c_kv, 8, ckv_stride_tok, k_rope, 8, krope_stride_tok,
                         cos_table, sin_table, table_len, context_lens, row_seq, rows, H, R, P, max_context, scale,
                         dtype, workspace, workspace_bytes, stream, pg);
}
