// Paged decode attention for MI355X (gfx950).
//
// One new query token per sequence attends to that sequence's cached K/V rows,
// which live in fixed-size physical blocks addressed through a block table.
// The operation is HBM-bound (about one FLOP per byte), so the kernel is built
// around keeping many 16-byte-per-lane loads in flight; all arithmetic is VALU
// (v_dot2c for bf16/f16 scores, fp32 fma for the weighted V sum).
//
// Work decomposition
//   workgroup = (sequence b, token split sp, kv-head group hg, q-head chunk qc)
//   wave      = one kv head of the group (x one sub-range of the split when the
//               model has fewer than 8 kv heads)
//   One wave-load covers 1 KiB: 64 lanes x 16 B = TPL consecutive tokens of one
//   head, LPT lanes per token row.  With the [block][head][row][d] layout the
//   8 waves of a workgroup walk the same physical blocks in step, so the 8 head
//   tiles of a block (contiguous in memory) are fetched together.
//   Every aligned group of LPT lanes keeps its own online-softmax state
//   (running max, sum, weighted V) for the tokens it sees, so the loop has no
//   communication beyond the LPT-lane score reduction; the groups are merged
//   once at the end, then sub-ranges through LDS, then token splits by a small
//   second kernel reading fp32 partials from the caller's workspace.
#include <stdlib.h>

#include <unistd.h>

#include <atomic>
#include <mutex>
#include <random>

#include "common.h"
#include "decode_merge.h"
#include "decode_params.h"

namespace mi355 {
namespace {

constexpr int kMaxThreads = 512;
constexpr int kLeadBytes = 64;   // the decode kernel's leading scalar arguments (16 dwords, before DecodeParams)

#ifndef MI355_DECODE_EARLY_BT
#define MI355_DECODE_EARLY_BT 1
#endif
#ifndef MI355_DECODE_WARM
#define MI355_DECODE_WARM 1
#endif

// MI355_DECODE_TIMING (side builds only): thread 0 of every workgroup of a launch without token
// splits writes the constant 100 MHz clock and the shader clock at five points of the kernel into
// the caller's workspace (16 64-bit words per workgroup): tools/probe_decode_timeline.py.
#ifdef MI355_DECODE_TIMING
#define DECODE_STAMP(n)                                                                             \
  do {                                                                                              \
    if (threadIdx.x == 0 && p.nsplit == 1 && p.ws_acc) {                                            \
      unsigned long long* st_ = reinterpret_cast<unsigned long long*>(p.ws_acc) + (size_t)blockIdx.x * 16; \
      st_[2 * (n)] = __builtin_amdgcn_s_memrealtime();                                              \
      st_[2 * (n) + 1] = __builtin_readcyclecounter();                                              \
    }                                                                                               \
  } while (0)
#else
#define DECODE_STAMP(n) do { } while (0)
#endif

template <class Tag, int D>
struct Geo {
  static constexpr int EPL = Tag::kEPL;      // elements per 16-byte lane load
  static constexpr int LPT = D / EPL;        // lanes per token row
  static constexpr int TPL = kWave / LPT;    // tokens per wave load
  static_assert(LPT >= 1 && LPT <= kWave && (LPT & (LPT - 1)) == 0, "unsupported head_dim");
};

// ROWS (causal mode only): the G slots of a wave are G consecutive query rows of one sequence and
// one query head instead of G query heads of one row.  They share every K/V load -- row r reads
// keys 0..r, so the group streams keys 0..last row once and masks, per slot, the few keys past
// that slot's own position -- which divides the K/V traffic of prompt processing by G.
// The fields a decode launch needs before its first loads, as leading scalar arguments: the first
// 16 dwords of a kernel's arguments arrive in SGPRs with the wave (kernel-argument preload), so the
// requests for the context length, the block-table slice and q go out without a first trip to the
// argument block.  The same values are in DecodeParams; everything after the prologue reads those.
struct DecodeLead {
  const int32_t* bt;
  const int32_t* ctx;    // null: packed causal launch (no table, no context lengths)
  const void* q;
  int q_stride_seq;      // (host: < 2^31)
  int qchunks, hgroups, nsplit, tps, bs_shift, max_blocks, max_ctx, hpw, gq;
};

// Rotate-half rotary embedding of one 16-byte piece: x = this lane's elements, y = the elements
// D/2 away (the partner lane's), c / s = cos and sin of the piece's own dims.  Two products and a
// sum, uncontracted, then one rounding by the caller's pack -- the arithmetic of
// rope_kv_write_kernel (layer_ops.hip), so both paths give the same bits.
template <int N>
__device__ __forceinline__ void rope_combine(const float (&x)[N], const float (&y)[N], const float (&c)[N],
                                             const float (&s)[N], bool low_half, float (&o)[N]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const float r = low_half ? -y[e] : y[e];
    o[e] = x[e] * c[e] + r * s[e];
  }
}

// APPEND (decode mode only): every sequence's newest K/V row arrives as an argument instead of
// sitting in the cache already.  The wave whose token range holds the last position stores the
// row there before it reads it, so the step's cache write needs no launch of its own.
template <class Tag, int D, int G, int U, bool NT, bool ROWS = false, bool APPEND = false>
__global__ void __launch_bounds__(kMaxThreads)
paged_decode_kernel(const int32_t* a_bt, const int32_t* a_ctx, const void* a_q, int a_q_stride_seq, int a_qchunks,
                    int a_hgroups, int a_nsplit, int a_tps, int a_bs_shift, int a_max_blocks, int a_max_ctx,
                    int a_hpw, int a_gq, const DecodeParams p) {
  constexpr int EPL = Geo<Tag, D>::EPL, LPT = Geo<Tag, D>::LPT, TPL = Geo<Tag, D>::TPL;
  constexpr int TS = TPL * U;  // tokens per pipeline step of one wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int32_t* s_bt = reinterpret_cast<int32_t*>(smem);
  float* s_part = reinterpret_cast<float*>(smem + p.lds_bt_bytes);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#if MI355_DECODE_EARLY_BT && MI355_DECODE_WARM
  // (the new-row fields at the end of DecodeParams are only read in append mode)
  int bid = blockIdx.x + warm_kernel_arguments<APPEND ? sizeof(DecodeParams) : offsetof(DecodeParams, k_new), kLeadBytes>();
#else
  int bid = blockIdx.x;
#endif
  DECODE_STAMP(0);
  const int qc = bid % a_qchunks;
  bid /= a_qchunks;
  const int hg = bid % a_hgroups;
  bid /= a_hgroups;
  const int sp = bid % a_nsplit;
  const int b = bid / a_nsplit;
  const int t0 = sp * a_tps;
  auto block_of = [&](int t) { return a_bs_shift >= 0 ? (t >> a_bs_shift) : (t / p.block_size); };
  const int lb0 = block_of(t0);

  // Decode mode: the first blockDim entries of this split's block-table slice are requested
  // before the context length is known (the slice up to max_ctx always exists in the table, and
  // ids are clamped before use), so the trips for the length, the table and q overlap instead of
  // following each other.
  int bt_first = 0;
#if MI355_DECODE_EARLY_BT
  if constexpr (!ROWS) {
    // (no branch around the load, so that nothing waits for it here: packed causal launches, which
    // have no table, read the first entry of their sequence offsets instead and ignore it)
    const int end_max = min(a_max_ctx, t0 + a_tps);
    const int nlb_max = end_max > t0 ? block_of(end_max - 1) - lb0 + 1 : 1;
    const int32_t* src = a_ctx ? a_bt + ((int64_t)b * a_max_blocks + lb0 + min(tid, nlb_max - 1)) : p.cu;
    bt_first = *src;
  }
#endif

  const int hslot = wave % a_hpw, sb = wave / a_hpw;
  const int kvh = hg * a_hpw + hslot;
  const int tl = lane / LPT, dg = lane % LPT;

  // Query fragments: this lane's EPL dims of each of the G query heads.  Their loads are issued
  // before the context length and the block table have arrived, so these trips to memory overlap
  // (row groups first need their rows' offsets, read below).
  int row0 = b, nrows = 1;             // ROWS: first q / out row of the group, rows in it
  const int hq0 = ROWS ? kvh * a_gq + qc : kvh * a_gq + qc * G;
  u32x4 qraw[G];
  auto load_q = [&]() {
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      const int hq = ROWS ? hq0 : (qc * G + gi < a_gq) ? hq0 + gi : hq0;
      const int qrow = ROWS ? row0 + min(gi, nrows - 1) : b;
      const char* src = static_cast<const char*>(a_q) +
                        ((int64_t)qrow * a_q_stride_seq + (int64_t)hq * D + dg * EPL) * Tag::kBytes;
      qraw[gi] = load16<false>(src);
    }
  };
  if constexpr (!ROWS) load_q();
  // Append mode: this lane's piece of the head's new K and V row (every token group of the wave
  // loads the same 16 bytes: no branch around the loads, and any lane group can store them)
  u32x4 new_k = {0u, 0u, 0u, 0u}, new_v = {0u, 0u, 0u, 0u};
  if constexpr (APPEND) {
    const int64_t src = ((int64_t)b * p.new_stride_seq + (int64_t)kvh * D + dg * EPL) * Tag::kBytes;
    new_k = load16<false>(static_cast<const char*>(p.k_new) + src);
    new_v = load16<false>(static_cast<const char*>(p.v_new) + src);
  }

  // Causal mode (prompt processing): "sequence" b is query token b of a packed
  // batch; it attends to tokens start..b of its own prompt, which sit in k/v rows
  // start..b (block_size 1, physical id = token row).  No block table is read.
  int ctx, lin_start = 0;
  int ctx_given = 0;                   // decode mode: the context length as the caller gave it
  int ctx0 = 0;                        // ROWS: context of the group's first row
  int bt_row = b;                      // row of the block table this workgroup reads
  if constexpr (ROWS) {
    // packed prompts (p.cu): sequence `seq` owns rows cu[seq] .. cu[seq + 1] - 1 and starts at position 0;
    // segments over the paged cache: rows seg_q0 .. + seg_len - 1, the first one at position seg_pos0
    // index order = dispatch order: the last (longest) row tiles of every sequence first, so that the
    // short ones fill the tail instead of one long wave finishing alone
    const int seq = b % p.nseq, first = (p.row_tiles - 1 - b / p.nseq) * G;
    int len, pos0 = 0;
    if (p.cu) {
      lin_start = p.cu[seq];
      len = p.cu[seq + 1] - lin_start;
      row0 = lin_start + first;
    } else {
      len = min(p.seg_len[seq], p.max_q);             // device values, clamped to the host-side bounds
      pos0 = min(max(p.seg_pos0[seq], 0), p.max_ctx);
      row0 = p.seg_q0[seq] + first;
      bt_row = seq;
    }
    if (first >= len) return;          // the grid is sized for the longest sequence / segment
    nrows = min(G, len - first);
    ctx0 = pos0 + first + 1;
    ctx = p.cu ? first + nrows : min(pos0 + first + nrows, p.max_ctx);
  } else if (!a_ctx) {   // packed causal launch (p.cu is set)
    int lo = 0, hi = p.nseq - 1;
    while (lo < hi) {  // largest s with cu[s] <= b
      const int mid = (lo + hi + 1) >> 1;
      if (p.cu[mid] <= b) lo = mid; else hi = mid - 1;
    }
    lin_start = p.cu[lo];
    ctx = b - lin_start + 1;
  } else {
    ctx_given = a_ctx[b];
    ctx = max(0, min(ctx_given, a_max_ctx));
  }
  // Append mode with rotary embedding: cos / sin of the new token's position for this lane's dims
  // (requested as soon as the length is known; used behind the table barrier)
  u32x4 rope_c = {0u, 0u, 0u, 0u}, rope_s = {0u, 0u, 0u, 0u};
  if constexpr (APPEND) {
    if (p.rope_cos) {
      const int pos = min(max(ctx_given - 1, 0), p.rope_len - 1);
      const int64_t off = ((int64_t)pos * D + dg * EPL) * Tag::kBytes;
      rope_c = load16<false>(static_cast<const char*>(p.rope_cos) + off);
      rope_s = load16<false>(static_cast<const char*>(p.rope_sin) + off);
    }
  }
  // Splits past the context do nothing and the merge skips them; with the merge in this launch
  // split 0 always takes part, so that an empty sequence still gets its (zero) output row.
  const bool fused = !ROWS && p.counters != nullptr;
  if (p.nsplit > 1 && t0 >= ctx && !(fused && sp == 0)) return;
  if (fused && sp == 0 && tid == 0) tag_arrival_cell(p, b, hg, qc);
  const int wg_end = min(ctx, t0 + p.tps);

  if constexpr (ROWS) load_q();

  // Stage this split's slice of the block table in LDS (ids clamped for memory safety).
  const int nlb = (wg_end > t0 && !p.cu) ? block_of(wg_end - 1) - lb0 + 1 : 0;
#if MI355_DECODE_EARLY_BT
  constexpr bool kEarly = !ROWS;
#else
  constexpr bool kEarly = false;
#endif
  if (kEarly && tid < nlb) s_bt[tid] = min(max(bt_first, 0), p.num_blocks - 1);
  for (int i = kEarly ? tid + (int)blockDim.x : tid; i < nlb; i += blockDim.x) {
    int id = p.bt[(int64_t)bt_row * p.max_blocks + lb0 + i];
    s_bt[i] = min(max(id, 0), p.num_blocks - 1);
  }
  __syncthreads();
  // Append mode: the new row's pieces have arrived by now (requested with q, ahead of the table
  // entry the barrier waited for).  Taking them here, in every wave, keeps a load that most waves
  // never use from standing in the way of the stream's own loads: a register whose load might
  // still be pending cannot be reused without waiting for everything issued after it.
  if constexpr (APPEND) asm("" : "+v"(new_k), "+v"(new_v));
  if constexpr (APPEND) {
    if (p.rope_cos) {
      // q and the new K row arrive un-rotated: a lane holds EPL dims of a row, the dims D/2 away sit
      // LPT/2 lanes further (same token group, and every token group holds the same pieces)
      static_assert(LPT >= 2, "a row's halves are on different lanes");
      float c[EPL], sn[EPL];
      Tag::unpack(rope_c, c);
      Tag::unpack(rope_s, sn);
      const bool low_half = dg < LPT / 2;
      auto rotated = [&](u32x4 raw) {
        u32x4 other;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t w = raw[i];
          other[i] = (uint32_t)__shfl_xor((int)w, LPT / 2);
        }
        float x[EPL], y[EPL], o[EPL];
        Tag::unpack(raw, x);
        Tag::unpack(other, y);
        rope_combine<EPL>(x, y, c, sn, low_half, o);
        return Tag::pack(o);
      };
#pragma unroll
      for (int gi = 0; gi < G; ++gi) qraw[gi] = rotated(qraw[gi]);
      new_k = rotated(new_k);
    }
  }
  DECODE_STAMP(1);
  // ROWS: the group's own context is shared out between the sub-range waves (early groups are short)
  const int len_w = ROWS ? (ctx + p.sub * TS - 1) / (p.sub * TS) * TS : p.tps / p.sub;
  const int tw0 = t0 + sb * len_w;
  const int tw1 = min(wg_end, tw0 + len_w);

  typename Tag::QFrag qf[G];
#pragma unroll
  for (int gi = 0; gi < G; ++gi) qf[gi] = Tag::make_q(qraw[gi]);

  float m[G], l[G], acc[G][EPL];
#pragma unroll
  for (int gi = 0; gi < G; ++gi) {
    m[gi] = kNegBig;
    l[gi] = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[gi][e] = 0.f;
  }

  const char* kbase = static_cast<const char*>(p.kc);
  const char* vbase = static_cast<const char*>(p.vc);
  const uint32_t lane_off = (uint32_t)kvh * (uint32_t)p.s_head + dg * EPL;  // host checks it fits

  // Append mode.  The new token sits at position ctx - 1 (a length beyond the launch's bound
  // appends nothing), and of this workgroup only one wave per head reads that position: the one
  // whose token range holds it, in the range's last pipeline step.  That wave never takes the row
  // from the cache: it parks the new row's pieces in LDS now, puts them in place of whatever the
  // loads of that step returned for the position, and stores them to the cache when its stream is
  // done -- so no load of this launch depends on the store, and the store's way through the
  // memory system is not on the path of the wave's loads (stores and loads of a wave complete in
  // the order issued: a store ahead of the stream made every later wait of the wave wait for its
  // acknowledgement too, 1.4 us on a 164 us launch).
  bool append_here = false;
  u32x4* s_new = nullptr;
  if constexpr (APPEND) {
    append_here = ctx_given == ctx && ctx - 1 >= tw0 && ctx - 1 < tw1;
    s_new = reinterpret_cast<u32x4*>(smem + p.lds_new_off) + wave * 2 * LPT;
    if (append_here && lane < LPT) {
      s_new[dg] = new_k;
      s_new[LPT + dg] = new_v;
    }
    __builtin_amdgcn_wave_barrier();   // (the wave's other lanes read these pieces later)
  }
  auto patch_step = [&](int step, u32x4 (&kb)[U], u32x4 (&vb)[U]) {
    if constexpr (APPEND) {
      if (append_here) {   // (wave-uniform; the pieces were written by this wave's own lanes)
        const u32x4 nk = s_new[dg], nv = s_new[LPT + dg];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int t = min(tw0 + step * TS + u * TPL + tl, tw1 - 1);   // the row load_step fetched
          if (t == ctx - 1) {
            kb[u] = nk;
            vb[u] = nv;
          }
        }
      }
    }
  };

  auto load_step = [&](int step, u32x4 (&kb)[U], u32x4 (&vb)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = min(tw0 + step * TS + u * TPL + tl, tw1 - 1);  // rows past the end re-read the last row
      const int lb = block_of(t);
      const int row = t - lb * p.block_size;
      const uint32_t pid = p.cu ? (uint32_t)(lin_start + lb) : (uint32_t)s_bt[lb - lb0];
      const uint64_t off = ((uint64_t)pid * (uint64_t)p.s_blk + (uint32_t)row * (uint32_t)p.s_tok + lane_off) *
                           Tag::kBytes;
      kb[u] = load16<NT>(kbase + off);
      vb[u] = load16<NT>(vbase + off);
    }
  };

  auto compute_step = [&](int step, const u32x4 (&kb)[U], const u32x4 (&vb)[U]) {
    float sc[G][U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tk = tw0 + step * TS + u * TPL + tl;
      const bool valid = tk < tw1;
#pragma unroll
      for (int gi = 0; gi < G; ++gi) {
        const float s = group_sum<LPT>(Tag::dot(qf[gi], kb[u]));
        const bool seen = ROWS ? (valid && tk < ctx0 + gi) : valid;   // ROWS: slot gi sits at position ctx0 + gi - 1
        sc[gi][u] = seen ? s * p.scale_log2e : -__builtin_inff();
      }
    }
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      float mx = sc[gi][0];
#pragma unroll
      for (int u = 1; u < U; ++u) mx = fmaxf(mx, sc[gi][u]);
      const float m_new = fmaxf(m[gi], mx);
      const float f = fast_exp2(m[gi] - m_new);
      m[gi] = m_new;
      l[gi] *= f;
#pragma unroll
      for (int e = 0; e < EPL; ++e) acc[gi][e] *= f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float vf[EPL];
      Tag::unpack(vb[u], vf);
#pragma unroll
      for (int gi = 0; gi < G; ++gi) {
        const float pr = fast_exp2(sc[gi][u] - m[gi]);
        l[gi] += pr;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[gi][e] = __builtin_fmaf(pr, vf[e], acc[gi][e]);
      }
    }
  };

  // Two register buffers: the loads of step s+1 are in flight while step s is reduced.
  const int nsteps = tw1 > tw0 ? (tw1 - tw0 + TS - 1) / TS : 0;
  u32x4 ka[U], va[U], kb[U], vb[U];
  if (nsteps > 0) load_step(0, ka, va);
  // (append mode: the position of the new row is the last one of its wave's range, so the step to
  // patch is the range's last -- the one after which nothing is left to load)
  if constexpr (APPEND) {
    for (int s = 0; s < nsteps; s += 2) {
      if (s + 1 < nsteps) load_step(s + 1, kb, vb);
      else patch_step(s, ka, va);
      compute_step(s, ka, va);
      if (s + 2 < nsteps) load_step(s + 2, ka, va);
      else if (s + 1 < nsteps) patch_step(s + 1, kb, vb);
      if (s + 1 < nsteps) compute_step(s + 1, kb, vb);
    }
  } else {
    for (int s = 0; s < nsteps; s += 2) {
      if (s + 1 < nsteps) load_step(s + 1, kb, vb);
      compute_step(s, ka, va);
#ifdef MI355_DECODE_TIMING
      if (s == 0) DECODE_STAMP(2);
#endif
      if (s + 2 < nsteps) load_step(s + 2, ka, va);
      if (s + 1 < nsteps) compute_step(s + 1, kb, vb);
    }
  }
  DECODE_STAMP(3);

  // Merge the TPL lane groups of the wave (same dims, different tokens).
  auto fmax_op = [](float a, float c) { return fmaxf(a, c); };
  auto add_op = [](float a, float c) { return a + c; };
#pragma unroll
  for (int gi = 0; gi < G; ++gi) {
    const float M = cross_group<LPT>(m[gi], fmax_op);
    const float f = fast_exp2(m[gi] - M);
    m[gi] = M;
    l[gi] = cross_group<LPT>(l[gi] * f, add_op);
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[gi][e] = cross_group<LPT>(acc[gi][e] * f, add_op);
  }

  // Merge the sub-ranges handled by different waves of this workgroup.
  if (p.sub > 1) {
    constexpr int kRec = D + 2;
    float* mine = s_part + (size_t)wave * G * kRec;
    if (lane < LPT) {
#pragma unroll
      for (int gi = 0; gi < G; ++gi) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) mine[gi * kRec + dg * EPL + e] = acc[gi][e];
        if (lane == 0) {
          mine[gi * kRec + D] = m[gi];
          mine[gi * kRec + D + 1] = l[gi];
        }
      }
    }
    __syncthreads();
    if (sb == 0 && lane < LPT) {
      for (int j = 1; j < p.sub; ++j) {
        const float* other = s_part + (size_t)(hslot + j * p.hpw) * G * kRec;
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          const float mo = other[gi * kRec + D], lo = other[gi * kRec + D + 1];
          const float M = fmaxf(m[gi], mo);
          const float fa = fast_exp2(m[gi] - M), fb = fast_exp2(mo - M);
          m[gi] = M;
          l[gi] = l[gi] * fa + lo * fb;
#pragma unroll
          for (int e = 0; e < EPL; ++e)
            acc[gi][e] = acc[gi][e] * fa + other[gi * kRec + dg * EPL + e] * fb;
        }
      }
    }
  }

  if constexpr (APPEND) {
    // the new row goes to the cache (behind the workgroup's barrier, which waits for a wave's stores)
    if (append_here && lane < LPT) {
      const int last = ctx - 1;
      const int lb = block_of(last);
      const uint64_t off = ((uint64_t)(uint32_t)s_bt[lb - lb0] * (uint64_t)p.s_blk +
                            (uint32_t)(last - lb * p.block_size) * (uint32_t)p.s_tok + lane_off) * Tag::kBytes;
      *reinterpret_cast<u32x4*>(const_cast<char*>(kbase) + off) = s_new[dg];
      *reinterpret_cast<u32x4*>(const_cast<char*>(vbase) + off) = s_new[LPT + dg];
    }
  }

  DECODE_STAMP(4);
  const bool owner = sb == 0 && lane < LPT;
  if (!owner && !fused) return;
#pragma unroll
  for (int gi = 0; gi < G; ++gi) {
    if (!owner) break;
    if (ROWS ? gi >= nrows : qc * G + gi >= p.gq) continue;
    const int hq = ROWS ? hq0 : hq0 + gi;
    const int orow = ROWS ? row0 + gi : b;
    if (ROWS || p.nsplit == 1) {
      const float inv = l[gi] > 0.f ? 1.f / l[gi] : 0.f;
      float o[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] = acc[gi][e] * inv;
      char* dst = static_cast<char*>(p.out) + (((int64_t)orow * p.Hq + hq) * D + dg * EPL) * Tag::kBytes;
      *reinterpret_cast<u32x4*>(dst) = Tag::pack(o);
    } else {
      const int64_t rec = ((int64_t)b * p.Hq + hq) * p.nsplit + sp;
      float* dst = p.ws_acc + rec * D + dg * EPL;
      if (fused) {
#pragma unroll
        for (int e = 0; e < EPL; e += 2) store_coherent2(dst + e, acc[gi][e], acc[gi][e + 1]);
        if (lane == 0) store_coherent2(p.ws_ml + rec * 2, m[gi], l[gi]);
      } else {
#pragma unroll
        for (int e = 0; e < EPL; e += 4)
          *reinterpret_cast<f32x4*>(dst + e) = f32x4{acc[gi][e], acc[gi][e + 1], acc[gi][e + 2], acc[gi][e + 3]};
        if (lane == 0) {
          p.ws_ml[rec * 2] = m[gi];
          p.ws_ml[rec * 2 + 1] = l[gi];
        }
      }
    }
  }

  // Token splits merged in this launch (decode_merge.h): the workgroup that arrives last
  // combines the group's partials.
  if constexpr (!ROWS) {
    if (!fused) return;
    const int nvalid = max(1, min(p.nsplit, (ctx + p.tps - 1) / p.tps));
    if (!arrived_last(p, b, hg, qc, nvalid, tid)) return;
    merge_splits_in_workgroup<Tag, D, kMaxThreads>(p, b, hg, qc, G, nvalid, tid);
  }
}

// Combine the token splits of every (sequence, query head): one thread per 4 output dims.
template <class Tag, int D>
__global__ void __launch_bounds__(256) paged_decode_merge_kernel(const DecodeParams p) {
  constexpr int kVec = 4;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)p.B * p.Hq * (D / kVec);
  if (idx >= total) return;
  const int dv = (int)(idx % (D / kVec));
  const int64_t bh = idx / (D / kVec);
  const int b = (int)(bh / p.Hq);
  const int ctx = max(0, min(p.ctx[b], p.max_ctx));
  const int nvalid = min(p.nsplit, (ctx + p.tps - 1) / p.tps);
  const float* ml = p.ws_ml + bh * p.nsplit * 2;
  const float* pa = p.ws_acc + bh * p.nsplit * D + dv * kVec;
  float M = kNegBig;
  for (int s = 0; s < nvalid; ++s) M = fmaxf(M, ml[2 * s]);
  float den = 0.f, o[kVec] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < nvalid; ++s) {
    const float f = fast_exp2(ml[2 * s] - M);
    den = __builtin_fmaf(ml[2 * s + 1], f, den);
    const f32x4 a = *reinterpret_cast<const f32x4*>(pa + (size_t)s * D);
#pragma unroll
    for (int e = 0; e < kVec; ++e) o[e] = __builtin_fmaf(a[e], f, o[e]);
  }
  const float inv = den > 0.f ? 1.f / den : 0.f;
  typename Tag::elem_t* dst = static_cast<typename Tag::elem_t*>(p.out) + bh * D + dv * kVec;
  if constexpr (Tag::kBytes == 4) {
    *reinterpret_cast<f32x4*>(dst) = f32x4{o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
  } else {
    float lo[8] = {o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv, 0.f, 0.f, 0.f, 0.f};
    const u32x4 packed = Tag::pack(lo);
    *reinterpret_cast<uint2*>(dst) = make_uint2(packed[0], packed[1]);
  }
}

// The same combination for many splits (small batches of long contexts): one workgroup per
// (sequence, query head); its threads are (split group, 4 output dims), so the chain of dependent
// loads a thread walks is nsplit / groups long instead of nsplit, and the groups meet in LDS.
template <class Tag, int D>
__global__ void __launch_bounds__(256) paged_decode_merge_wide_kernel(const DecodeParams p) {
  constexpr int kVec = 4, kDv = D / kVec, kGroups = 256 / kDv;
  static_assert(kDv <= 64 && 256 % kDv == 0, "head_dim");
  __shared__ float s_red[kGroups][kDv][kVec + 1];
  __shared__ float s_max[4];
  const int tid = threadIdx.x, dv = tid % kDv, grp = tid / kDv;
  const int64_t bh = blockIdx.x;
  const int b = (int)(bh / p.Hq);
  const int ctx = max(0, min(p.ctx[b], p.max_ctx));
  const int nvalid = min(p.nsplit, (ctx + p.tps - 1) / p.tps);
  const float* ml = p.ws_ml + bh * p.nsplit * 2;
  const float* pa = p.ws_acc + bh * p.nsplit * D + dv * kVec;
  float M = kNegBig;
  for (int s = tid; s < nvalid; s += 256) M = fmaxf(M, ml[2 * s]);
  for (int off = 32; off > 0; off >>= 1) M = fmaxf(M, __shfl_xor(M, off));
  if ((tid & 63) == 0) s_max[tid >> 6] = M;
  __syncthreads();
  M = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
  float den = 0.f, o[kVec] = {0.f, 0.f, 0.f, 0.f};
  for (int s = grp; s < nvalid; s += kGroups) {
    const float f = fast_exp2(ml[2 * s] - M);
    den = __builtin_fmaf(ml[2 * s + 1], f, den);
    const f32x4 a = *reinterpret_cast<const f32x4*>(pa + (size_t)s * D);
#pragma unroll
    for (int e = 0; e < kVec; ++e) o[e] = __builtin_fmaf(a[e], f, o[e]);
  }
#pragma unroll
  for (int e = 0; e < kVec; ++e) s_red[grp][dv][e] = o[e];
  s_red[grp][dv][kVec] = den;
  __syncthreads();
  if (grp != 0) return;
  // fixed order over the groups: the result does not depend on timing
  for (int g = 1; g < kGroups; ++g) {
#pragma unroll
    for (int e = 0; e < kVec; ++e) o[e] += s_red[g][dv][e];
    den += s_red[g][dv][kVec];
  }
  const float inv = den > 0.f ? 1.f / den : 0.f;
  typename Tag::elem_t* dst = static_cast<typename Tag::elem_t*>(p.out) + bh * D + dv * kVec;
  if constexpr (Tag::kBytes == 4) {
    *reinterpret_cast<f32x4*>(dst) = f32x4{o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
  } else {
    float lo[8] = {o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv, 0.f, 0.f, 0.f, 0.f};
    const u32x4 packed = Tag::pack(lo);
    *reinterpret_cast<uint2*>(dst) = make_uint2(packed[0], packed[1]);
  }
}

constexpr int kBalanceSteps = 8;      // pipeline steps every wave keeps when heads are spread for balance
constexpr int64_t kSpreadTileBytes = 2048;   // smallest contiguous (block, head) tile for which heads are spread
constexpr int kWideMergeSplits = 8;   // from this many splits on, one workgroup per (sequence, head) merges

int elems_per_lane(int dtype) { return dtype == MI355_F32 ? 4 : 8; }

#ifndef MI355_DECODE_U
#define MI355_DECODE_U 4
#endif
constexpr int kU = MI355_DECODE_U;  // 16-byte loads of K (and of V) per lane per pipeline step
// With 8 query heads per wave the accumulators and scores take most of the register file;
// a shallower step keeps the kernel free of scratch spills (a spill reload shares the
// memory counter with the prefetch loads and would serialise them).
constexpr int steps_for_group(int G) { return G >= 8 ? (kU > 2 ? 2 : kU) : kU; }

// Launch geometry for a problem shape.  Deterministic in its arguments (plus the launch-plan
// knobs, which only mi355_debug_set_knob changes after start-up), so the workspace query and the
// launch agree.
// What the planner knows about the cache's strides (the workspace query knows nothing):
//   kLayoutHeadTiles    a (block, kv head) tile is one contiguous piece of >= kSpreadTileBytes
//                       ([block][head][row][d]): kv heads may be spread over workgroups (below)
//   kLayoutInterleaved  the rows of all heads alternate ([block][row][head][d]): a workgroup of h
//                       heads reads pieces of h * D elements per token
enum PlanLayout { kLayoutUnknown = 0, kLayoutHeadTiles = 1, kLayoutInterleaved = 2 };
constexpr int64_t kInterleavedPieceBytes = 512;        // smallest piece a bandwidth-bound launch should read ...
constexpr int64_t kInterleavedBigBytes = 256ll << 20;  // ... from this many bytes of K/V on

int make_plan(int B, int Hq, int Hkv, int D, int block_size, int max_context, int dtype,
              DecodePlan* plan, bool causal = false, int layout = kLayoutUnknown) {
  if (B <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv) return fail(MI355_EINVAL, "bad B/Hq/Hkv (%d,%d,%d)", B, Hq, Hkv);
  if (dtype != MI355_F32 && dtype != MI355_BF16 && dtype != MI355_F16)
    return fail(MI355_EINVAL, "unknown dtype code %d", dtype);
  if (D != 16 && D != 32 && D != 64 && D != 128 && D != 256)
    return fail(MI355_EUNSUPPORTED, "head_dim %d not in {16,32,64,128,256}", D);
  if (block_size <= 0) return fail(MI355_EINVAL, "block_size %d", block_size);
  if (max_context < 0) return fail(MI355_EINVAL, "max_context %d", max_context);
  DecodePlan pl;
  const int gq = Hq / Hkv;
  // Grouped-query shapes (>= 4 query heads per kv head, 16-bit, head_dim 64 / 128) go to the
  // matrix-core kernel; MI355_DECODE_GQA_MFMA=0 keeps them on the VALU kernel (A/B measurements).
  pl.mfma = !causal && decode_gqa_supported(dtype, D, gq) && knob(kKnobDecodeGqaMfma) != 0;
  pl.nw = pl.mfma ? kGqaWaves : 8;
  pl.G = pl.mfma ? kGqaHeads : gq >= 8 ? 8 : gq > 2 ? 4 : gq;  // query heads per wave
  pl.qchunks = (gq + pl.G - 1) / pl.G;
  const int lpt = D / elems_per_lane(dtype);
  const int ts = pl.mfma ? kGqaTokens : (kWave / lpt) * steps_for_group(pl.G);  // tokens per wave per pipeline step
  const int ctx = max_context > 0 ? max_context : 1;
  // one 8-wave workgroup per CU, or two 4-wave ones, before tokens are split
  const int target = knob(kKnobPagedTargetWgs);
  // Heads per workgroup.  Eight kv heads per workgroup fetch the eight head tiles of a block
  // (contiguous in the [block][head][row][d] layout) together.  When that leaves CUs idle,
  // fewer heads per workgroup with the freed waves taking token sub-ranges of the same head
  // fills the chip without a second launch: sub-ranges merge through LDS inside the
  // workgroup, token splits across workgroups need the merge kernel and a workspace trip.
  pl.hpw = pl.nw;
  while (pl.hpw > 1 && Hkv % pl.hpw) pl.hpw /= 2;
  const int forced_hpw = knob(kKnobPagedHpw);
  if (forced_hpw > 0 && forced_hpw <= pl.hpw && pl.hpw % forced_hpw == 0) {
    pl.hpw = forced_hpw;
  } else if (!causal) {
    // Interleaved rows, bandwidth-bound sizes: heads stay together until a token's piece is
    // >= 512 bytes, and the chip is filled with token splits instead (merged in the launch).
    // Measured, 8 kv heads: head_dim 64 B=16 ctx 8192: 52.0 us with one head per workgroup (128-byte
    // pieces), 45.1 with four; B=16 ctx 32768: 188.8 -> 161.4; head_dim 128 (matrix-core kernel)
    // B=16 ctx 32768: 334.8 -> 312.7 with two.  Small launches are latency-bound and keep the finer plan.
    const int ebytes = dtype == MI355_F32 ? 4 : 2;
    int min_hpw = 1;
    if (layout == kLayoutInterleaved && (int64_t)B * ctx * Hkv * D * 2 * ebytes >= kInterleavedBigBytes)
      while (min_hpw < pl.hpw && (int64_t)min_hpw * D * ebytes < kInterleavedPieceBytes) min_hpw *= 2;
    while (pl.hpw > min_hpw && (int64_t)B * (Hkv / pl.hpw) * pl.qchunks < target &&
           ctx >= 2 * ts * (pl.nw / pl.hpw) * 2)  // every sub-range keeps >= 2 pipeline steps
      pl.hpw /= 2;
    // Beyond filling the chip: fewer heads per workgroup still (its waves sharing a head's tokens)
    // costs a uniform batch 0-1 % while every wave keeps a long enough stream (metric shape:
    // 164.5 us with 256 workgroups of 8 heads, 164.6 with 2048 of one; head_dim 128 and fp32: +1 %)
    // and evens out a ragged one, whose launch otherwise ends with the longest sequences'
    // workgroups alone on the chip (contexts U[S/2, S]: 132 -> 128 us, U[1, S]: 104 -> 90 us,
    // exponential lengths: 88 -> 54 us).  Only for caches whose (block, head) tiles are contiguous
    // pieces of >= 2 KiB: with 1 KiB tiles (head_dim 32) it costs 7 %, and with rows of all heads
    // interleaved ([block][row][head][d]) one head per workgroup reads 128-byte pieces (+15 %).
    const int balance = layout == kLayoutHeadTiles ? knob(kKnobPagedBalanceWgs) : 0;
    while (pl.hpw > 1 && (int64_t)B * (Hkv / pl.hpw) * pl.qchunks < balance &&
           ctx >= kBalanceSteps * ts * (pl.nw / pl.hpw) * 2)
      pl.hpw /= 2;
  }
  pl.sub = pl.nw / pl.hpw;
  pl.hgroups = Hkv / pl.hpw;
  const int quantum = pl.sub * ts;  // tokens one workgroup consumes per pipeline step
  const int64_t base_wgs = (int64_t)B * pl.hgroups * pl.qchunks;
  int64_t want = ceil_div64(target, base_wgs);
  const int fuse_up_to = knob(kKnobDecodeFusedMerge);
  const int64_t most = ctx / (2 * quantum) > 0 ? ctx / (2 * quantum) : 1;  // >= 2 steps per wave
  if (want > most) want = most;
  if (want < 1 || causal) want = 1;  // causal mode: one workgroup per query row, no token split
  int64_t tps = ceil_div64(ceil_div64(ctx, want), quantum) * quantum;
  if (!causal) {
    const int64_t tps_cap = ((int64_t)(kMaxBtEntries - 2) * block_size / quantum) * quantum;
    if (tps_cap < quantum)
      return fail(MI355_EUNSUPPORTED, "block_size %d too small for head_dim %d", block_size, D);
    if (tps > tps_cap) tps = tps_cap;
  }
  pl.tps = (int)tps;
  pl.nsplit = (int)ceil_div64(ctx, tps);
  pl.lds_bt_bytes = causal ? 16 : (size_t)((tps / block_size + 2) * 4 + 15) / 16 * 16;
  pl.lds_bytes = pl.mfma ? decode_gqa_lds_bytes(D, pl.sub, pl.lds_bt_bytes)
                         : pl.lds_bt_bytes + (pl.sub > 1 ? (size_t)pl.nw * pl.G * (D + 2) * 4 : 0);
  pl.ws_bytes = pl.nsplit > 1 ? (size_t)B * Hq * pl.nsplit * (D + 2) * sizeof(float) : 0;
  // Up to MI355_DECODE_FUSED_MERGE splits (0: never) the decode kernels merge their token splits
  // themselves (decode_merge.h; arrival cells behind the partials): with the host queued ahead that saves 1-2 us of
  // 17-47 at 2-8 splits; from 16 splits on, the last workgroup's serial tail (arrival, then
  // reading every partial) takes longer than the wide merge kernel's launch.
  pl.fused_merge = pl.nsplit > 1 && pl.nsplit <= (fuse_up_to < 65535 ? fuse_up_to : 65535) && !causal;
  if (pl.fused_merge) pl.ws_bytes += (size_t)B * pl.hgroups * pl.qchunks * sizeof(unsigned long long);
  *plan = pl;
  return 0;
}

#ifndef MI355_DECODE_NT
#define MI355_DECODE_NT 1
#endif

template <class Tag, int D>
int launch_merge(const DecodeParams& p, hipStream_t stream) {
  if constexpr (D >= 16) {
    if (p.nsplit >= kWideMergeSplits) {
      hipLaunchKernelGGL((paged_decode_merge_wide_kernel<Tag, D>), dim3((unsigned)((int64_t)p.B * p.Hq)), dim3(256), 0,
                         stream, p);
      return check_launch("paged_decode_merge_wide_kernel");
    }
  }
  const int64_t threads = (int64_t)p.B * p.Hq * (D / 4);
  hipLaunchKernelGGL((paged_decode_merge_kernel<Tag, D>), dim3((unsigned)ceil_div64(threads, 256)), dim3(256), 0,
                     stream, p);
  return check_launch("paged_decode_merge_kernel");
}

#define MI355_DECODE_LEAD_ARGS(p)                                                                         \
  (p).bt, (p).cu ? nullptr : (p).ctx, (p).q, (int)(p).q_stride_seq, (p).qchunks, (p).hgroups, (p).nsplit, (p).tps, \
      (p).bs_shift, (p).max_blocks, (p).max_ctx, (p).hpw, (p).gq

template <class Tag, int D, int G>
int launch_decode(const DecodeParams& p, const DecodePlan& pl, hipStream_t stream) {
  const int64_t grid = (int64_t)p.B * p.nsplit * p.hgroups * p.qchunks;
  if (p.q_stride_seq >= (1LL << 31)) return fail(MI355_EINVAL, "paged_decode: q row stride out of range");
  if (p.k_new)
    hipLaunchKernelGGL((paged_decode_kernel<Tag, D, G, steps_for_group(G), MI355_DECODE_NT != 0, false, true>),
                       dim3((unsigned)grid), dim3(pl.nw * kWave), pl.lds_bytes, stream, MI355_DECODE_LEAD_ARGS(p), p);
  else
    hipLaunchKernelGGL((paged_decode_kernel<Tag, D, G, steps_for_group(G), MI355_DECODE_NT != 0>), dim3((unsigned)grid),
                       dim3(pl.nw * kWave), pl.lds_bytes, stream, MI355_DECODE_LEAD_ARGS(p), p);
  if (int rc = check_launch("paged_decode_kernel")) return rc;
  if (p.nsplit > 1 && !p.counters) return launch_merge<Tag, D>(p, stream);
  return 0;
}

// Prompt processing on the VALU kernel: row groups of kRowGroup consecutive query rows per wave.
constexpr int kRowGroup = 8;
constexpr size_t kRowsMaxLds = 144 * 1024;   // of the 160 KB a CU has
template <class Tag, int D>
int launch_causal_rows(const DecodeParams& p, size_t lds_bytes, hipStream_t stream) {
  const int64_t grid = (int64_t)p.B * p.hgroups * p.qchunks;
  auto kernel = paged_decode_kernel<Tag, D, kRowGroup, steps_for_group(kRowGroup), false, true>;
  if (lds_bytes > 64 * 1024) {   // beyond the default dynamic-LDS limit (head_dim 256 with waves splitting the keys)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess)
      return fail(MI355_EUNSUPPORTED, "row-group kernel: %zu bytes of LDS refused", lds_bytes);
  }
  if (p.q_stride_seq >= (1LL << 31)) return fail(MI355_EINVAL, "prefill: q row stride out of range");
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(8 * kWave), lds_bytes, stream, MI355_DECODE_LEAD_ARGS(p), p);
  return check_launch("paged_decode_kernel (causal row groups)");
}

template <class Tag>
int dispatch_causal_rows(int D, const DecodeParams& p, size_t lds_bytes, hipStream_t s) {
  switch (D) {
    case 16: return launch_causal_rows<Tag, 16>(p, lds_bytes, s);
    case 32: return launch_causal_rows<Tag, 32>(p, lds_bytes, s);
    case 64: return launch_causal_rows<Tag, 64>(p, lds_bytes, s);
    case 128: return launch_causal_rows<Tag, 128>(p, lds_bytes, s);
    case 256: return launch_causal_rows<Tag, 256>(p, lds_bytes, s);
  }
  return fail(MI355_EUNSUPPORTED, "head_dim %d", D);
}

// Matrix-core kernel for grouped-query shapes (+ the shared merge kernel when tokens are split).
int launch_gqa(const DecodeParams& p, const DecodePlan& pl, int D, int dtype, hipStream_t s) {
  if (int rc = decode_gqa_launch(p, pl, D, dtype, s)) return rc;
  if (p.nsplit == 1 || p.counters) return 0;   // (token splits merged inside the launch)
  if (dtype == MI355_BF16) return D == 64 ? launch_merge<BF16, 64>(p, s) : launch_merge<BF16, 128>(p, s);
  return D == 64 ? launch_merge<F16, 64>(p, s) : launch_merge<F16, 128>(p, s);
}

template <class Tag, int D>
int dispatch_g(const DecodeParams& p, const DecodePlan& pl, hipStream_t s) {
  switch (pl.G) {
    case 1: return launch_decode<Tag, D, 1>(p, pl, s);
    case 2: return launch_decode<Tag, D, 2>(p, pl, s);
    case 4: return launch_decode<Tag, D, 4>(p, pl, s);
    case 8: return launch_decode<Tag, D, 8>(p, pl, s);
  }
  return fail(MI355_EUNSUPPORTED, "query group %d", pl.G);
}

template <class Tag>
int dispatch_d(int D, const DecodeParams& p, const DecodePlan& pl, hipStream_t s) {
  switch (D) {
    case 16: return dispatch_g<Tag, 16>(p, pl, s);
    case 32: return dispatch_g<Tag, 32>(p, pl, s);
    case 64: return dispatch_g<Tag, 64>(p, pl, s);
    case 128: return dispatch_g<Tag, 128>(p, pl, s);
    case 256: return dispatch_g<Tag, 256>(p, pl, s);
  }
  return fail(MI355_EUNSUPPORTED, "head_dim %d", D);
}

// Chunks of new query tokens over the paged cache for the shapes the matrix-core kernels do not
// take: the VALU kernel with kRowGroup consecutive query rows of a segment per wave.
int paged_prefill_rows(void* out, const void* q, int64_t q_stride_tok, const void* k_cache, const void* v_cache,
                       const int32_t* block_tables, const int32_t* seg_q0, const int32_t* seg_len,
                       const int32_t* seg_pos0, int nseg, int max_q_len, int Hq, int Hkv, int D, int block_size,
                       int max_blocks, int num_blocks, int max_context, int64_t s_blk, int64_t s_head,
                       int64_t s_tok, float scale, int dtype, hipStream_t stream) {
  if (dtype != MI355_F32 && dtype != MI355_BF16 && dtype != MI355_F16)
    return fail(MI355_EINVAL, "paged_prefill: unknown dtype code %d", dtype);
  if (D != 16 && D != 32 && D != 64 && D != 128 && D != 256)
    return fail(MI355_EUNSUPPORTED, "paged_prefill: head_dim %d not in {16,32,64,128,256}", D);
  const int epl = elems_per_lane(dtype);
  if (s_blk <= 0 || s_head < 0 || s_tok <= 0 || s_blk >= (1LL << 31) || s_tok >= (1LL << 31) ||
      (int64_t)Hkv * s_head + D >= (1LL << 31))
    return fail(MI355_EINVAL, "paged_prefill: kv strides out of range");
  if (s_blk % epl || s_head % epl || s_tok % epl || q_stride_tok % epl || q_stride_tok < (int64_t)Hq * D)
    return fail(MI355_EINVAL, "paged_prefill: strides must keep head rows 16-byte aligned");
  if (((uintptr_t)out | (uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache) & 15)
    return fail(MI355_EINVAL, "paged_prefill: tensors must be 16-byte aligned");
  DecodeParams p;
  p.out = out;
  p.q = q;
  p.kc = k_cache;
  p.vc = v_cache;
  p.bt = block_tables;
  p.ctx = nullptr;
  p.cu = nullptr;
  p.nseq = nseg;
  p.ws_acc = nullptr;
  p.ws_ml = nullptr;
  p.q_stride_seq = q_stride_tok;
  p.s_blk = s_blk;
  p.s_head = s_head;
  p.s_tok = s_tok;
  p.Hq = Hq;
  p.Hkv = Hkv;
  p.gq = Hq / Hkv;
  p.block_size = block_size;
  p.bs_shift = ilog2_exact(block_size);
  p.max_blocks = max_blocks;
  p.num_blocks = num_blocks;
  p.max_ctx = max_context;
  p.hpw = 8;
  while (p.hpw > 1 && Hkv % p.hpw) p.hpw /= 2;
  p.sub = 8 / p.hpw;
  p.hgroups = Hkv / p.hpw;
  p.qchunks = p.gq;
  const int ts = (kWave / (D / epl)) * steps_for_group(kRowGroup);
  const int quantum = p.sub * ts;
  p.tps = (int)(ceil_div64(max_context, quantum) * quantum);
  p.nsplit = 1;
  if ((int64_t)p.tps / block_size + 2 > kMaxBtEntries)
    return fail(MI355_EUNSUPPORTED, "paged_prefill: context %d with block_size %d exceeds the staged block table",
                max_context, block_size);
  p.lds_bt_bytes = (int)((size_t)((p.tps / block_size + 2) * 4 + 15) / 16 * 16);
  const size_t lds = (size_t)p.lds_bt_bytes + (p.sub > 1 ? (size_t)8 * kRowGroup * (D + 2) * sizeof(float) : 0);
  if (lds > kRowsMaxLds)
    return fail(MI355_EUNSUPPORTED, "paged_prefill: context %d at head_dim %d needs %zu bytes of LDS", max_context, D, lds);
  p.scale_log2e = scale * 1.4426950408889634f;
  p.seg_q0 = seg_q0;
  p.seg_len = seg_len;
  p.seg_pos0 = seg_pos0;
  p.n_qt = p.tq = p.hgp_shift = 0;
  p.max_q = max_q_len;
  p.row_tiles = (max_q_len + kRowGroup - 1) / kRowGroup;
  const int64_t groups = (int64_t)nseg * p.row_tiles;
  if (groups * p.hgroups * p.qchunks >= (1LL << 31)) return fail(MI355_EINVAL, "paged_prefill: grid too large");
  p.B = (int)groups;
  switch (dtype) {
    case MI355_F32: return dispatch_causal_rows<F32>(D, p, lds, stream);
    case MI355_BF16: return dispatch_causal_rows<BF16>(D, p, lds, stream);
    default: return dispatch_causal_rows<F16>(D, p, lds, stream);
  }
}

}  // namespace
}  // namespace mi355

namespace mi355 {
namespace {
// The launch picks its plan by the cache's strides, which the workspace query does not see: the
// query answers for whichever plan needs more.
size_t decode_workspace_bound(int B, int Hq, int Hkv, int D, int block_size, int max_context, int dtype) {
  DecodePlan a, b;
  if (make_plan(B, Hq, Hkv, D, block_size, max_context, dtype, &a)) return 0;
  if (make_plan(B, Hq, Hkv, D, block_size, max_context, dtype, &b, false, kLayoutInterleaved)) return a.ws_bytes;
  return a.ws_bytes > b.ws_bytes ? a.ws_bytes : b.ws_bytes;
}
}  // namespace
}  // namespace mi355

using namespace mi355;

extern "C" size_t mi355_paged_decode_workspace_bytes(int32_t B, int32_t Hq, int32_t Hkv, int32_t D,
                                                     int32_t block_size, int32_t max_context,
                                                     int32_t dtype) {
  return decode_workspace_bound(B, Hq, Hkv, D, block_size, max_context, dtype);
}

namespace mi355 {

// A 48-bit tag, different for every launch of this process and starting from a random value, by
// which a launch tells its own arrival counts from whatever the workspace held before.  The top
// four bits are set: a tagged cell compares above most leftovers (an atomic max installs the tag).
unsigned long long next_launch_tag() {
  static std::atomic<unsigned long long> counter{[] {
    std::random_device rd;
    return ((unsigned long long)rd() << 32) ^ rd() ^ (unsigned long long)getpid();
  }()};
  return (counter.fetch_add(1, std::memory_order_relaxed) & ((1ull << 44) - 1)) | (0xFull << 44);
}

namespace {

// The cache write of mi355_paged_decode_append as a launch of its own, for the shapes the
// matrix-core decode kernel serves: thread = (sequence, kv head, 16-byte piece) stores that piece
// of the sequence's new K and V row at position context - 1 of its block table.
struct AppendParams {
  char* kc;
  char* vc;
  const char* k;
  const char* v;
  const int32_t* bt;
  const int32_t* ctx;
  int64_t src_stride_seq, s_blk, s_head, s_tok;   // elements
  int B, Hkv, lpt, ebytes, block_size, bs_shift, max_blocks, num_blocks, max_ctx;
};

__global__ void __launch_bounds__(256) kv_append_kernel(const AppendParams p) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_seq = p.Hkv * p.lpt;
  if (idx >= (int64_t)p.B * per_seq) return;
  const int b = (int)(idx / per_seq), rem = (int)(idx - (int64_t)b * per_seq);
  const int h = rem / p.lpt, piece = rem - h * p.lpt;
  const int ctx = p.ctx[b];
  if (ctx < 1 || ctx > p.max_ctx) return;
  const int last = ctx - 1;
  const int lb = p.bs_shift >= 0 ? (last >> p.bs_shift) : (last / p.block_size);
  const int id = min(max(p.bt[(int64_t)b * p.max_blocks + lb], 0), p.num_blocks - 1);
  const int epl = 16 / p.ebytes;
  const int64_t src = ((int64_t)b * p.src_stride_seq + (int64_t)h * p.lpt * epl + piece * epl) * p.ebytes;
  const int64_t dst = ((int64_t)id * p.s_blk + (int64_t)h * p.s_head + (int64_t)(last - lb * p.block_size) * p.s_tok +
                       piece * epl) * p.ebytes;
  *reinterpret_cast<u32x4*>(p.kc + dst) = *reinterpret_cast<const u32x4*>(p.k + src);
  *reinterpret_cast<u32x4*>(p.vc + dst) = *reinterpret_cast<const u32x4*>(p.v + src);
}

// Body of mi355_paged_decode; `what` names the entry point in error messages.  k_new / v_new
// non-null: append mode (mi355_paged_decode_append), the caches are written before they are read.
int paged_decode_impl(const char* what, void* out, const void* q, int64_t q_stride_seq, const void* k_cache,
                      const void* v_cache, const int32_t* block_tables, const int32_t* context_lens, int32_t B,
                      int32_t Hq, int32_t Hkv, int32_t D, int32_t block_size, int32_t max_blocks,
                      int32_t num_blocks, int32_t max_context, int64_t kv_stride_block, int64_t kv_stride_head,
                      int64_t kv_stride_tok, float scale, int32_t dtype, void* workspace, size_t workspace_bytes,
                      void* stream, const void* k_new = nullptr, const void* v_new = nullptr,
                      int64_t new_stride_seq = 0, const void* rope_cos = nullptr, const void* rope_sin = nullptr,
                      int32_t rope_len = 0) {
  if (B == 0) return 0;
  if (!out || !q || !k_cache || !v_cache || !block_tables || !context_lens)
    return fail(MI355_EINVAL, "%s: null pointer argument", what);
  if (max_blocks <= 0 || num_blocks <= 0) return fail(MI355_EINVAL, "%s: max_blocks/num_blocks", what);
  if ((int64_t)max_context > (int64_t)max_blocks * block_size) max_context = max_blocks * block_size;
  DecodePlan pl;
  const bool interleaved = kv_stride_tok != D;   // rows of one head are not adjacent
  if (int rc = make_plan(B, Hq, Hkv, D, block_size, max_context, dtype, &pl, false,
                         interleaved ? kLayoutInterleaved : kLayoutUnknown))
    return rc;
  // A launch without token splits over contiguous head tiles may spread its kv heads over more
  // workgroups (ragged batches, see make_plan), provided that plan needs no workspace either.
  if (!interleaved && pl.nsplit == 1 && (int64_t)block_size * D * (dtype == MI355_F32 ? 4 : 2) >= kSpreadTileBytes) {
    DecodePlan spread;
    if (make_plan(B, Hq, Hkv, D, block_size, max_context, dtype, &spread, false, kLayoutHeadTiles) == 0 &&
        spread.nsplit == 1)
      pl = spread;
  }
  const int epl = elems_per_lane(dtype);
  if (kv_stride_block <= 0 || kv_stride_head < 0 || kv_stride_tok <= 0 || kv_stride_block >= (1LL << 31) ||
      kv_stride_tok >= (1LL << 31) || (int64_t)Hkv * kv_stride_head + D >= (1LL << 31))
    return fail(MI355_EINVAL, "%s: kv strides out of range", what);
  if (kv_stride_block % epl || kv_stride_head % epl || kv_stride_tok % epl || q_stride_seq % epl ||
      q_stride_seq < (int64_t)Hq * D)
    return fail(MI355_EINVAL, "%s: strides must keep head rows 16-byte aligned", what);
  if (((uintptr_t)out | (uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache) & 15)
    return fail(MI355_EINVAL, "%s: tensors must be 16-byte aligned", what);
  if ((int64_t)B * pl.nsplit * pl.hgroups * pl.qchunks >= (1LL << 31) || (int64_t)B * Hq >= (1LL << 31))
    return fail(MI355_EINVAL, "%s: grid too large (B %d)", what, B);
  if (pl.ws_bytes && (!workspace || workspace_bytes < pl.ws_bytes || ((uintptr_t)workspace & 15)))
    return fail(MI355_EWORKSPACE, "%s: need %zu workspace bytes (16-B aligned), got %zu", what,
                pl.ws_bytes, workspace ? workspace_bytes : (size_t)0);

  DecodeParams p;
  p.out = out;
  p.q = q;
  p.kc = k_cache;
  p.vc = v_cache;
  p.bt = block_tables;
  p.ctx = context_lens;
  p.cu = nullptr;
  p.nseq = 0;
  p.ws_acc = static_cast<float*>(workspace);
  p.ws_ml = p.ws_acc ? p.ws_acc + (size_t)B * Hq * pl.nsplit * D : nullptr;
  if (pl.fused_merge) {   // (D + 2 is even: the cells behind the partials are 8-byte aligned)
    p.counters = reinterpret_cast<unsigned long long*>(p.ws_ml + (size_t)B * Hq * pl.nsplit * 2);
    p.epoch = next_launch_tag();
  }
  p.q_stride_seq = q_stride_seq;
  p.s_blk = kv_stride_block;
  p.s_head = kv_stride_head;
  p.s_tok = kv_stride_tok;
  p.B = B;
  p.Hq = Hq;
  p.Hkv = Hkv;
  p.gq = Hq / Hkv;
  p.block_size = block_size;
  p.bs_shift = ilog2_exact(block_size);
  p.max_blocks = max_blocks;
  p.num_blocks = num_blocks;
  p.max_ctx = max_context;
  p.hpw = pl.hpw;
  p.sub = pl.sub;
  p.hgroups = pl.hgroups;
  p.qchunks = pl.qchunks;
  p.tps = pl.tps;
  p.nsplit = pl.nsplit;
  p.lds_bt_bytes = (int)pl.lds_bt_bytes;
  p.scale_log2e = scale * 1.4426950408889634f;
  p.seg_q0 = p.seg_len = p.seg_pos0 = nullptr;
  p.n_qt = p.tq = p.hgp_shift = p.max_q = 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (k_new) {
    if (!v_new) return fail(MI355_EINVAL, "%s: null pointer argument", what);
    if (new_stride_seq % epl || new_stride_seq < (int64_t)Hkv * D || (((uintptr_t)k_new | (uintptr_t)v_new) & 15))
      return fail(MI355_EINVAL, "%s: new K/V rows must be 16-byte aligned, [B, Hkv, D]", what);
    // per wave of the VALU kernel: the K and the V row of its head, parked in LDS
    const size_t stash = (size_t)pl.nw * 2 * D * (dtype == MI355_F32 ? 4 : 2);
    // The matrix-core kernel reads finished caches, and a workgroup's LDS has its limit: there
    // the write goes first, as a launch of its own.
    if (pl.mfma || pl.lds_bytes + stash > 64 * 1024) {
      // (with rotary embedding q is rotated inside the VALU kernel only: nothing has been launched yet)
      if (rope_cos)
        return fail(MI355_EUNSUPPORTED, "%s: this shape is served by the matrix-core kernel or fills the workgroup's "
                                        "LDS; rotate with mi355_rope_kv_write and call mi355_paged_decode", what);
      AppendParams ap;
      ap.kc = static_cast<char*>(const_cast<void*>(k_cache));
      ap.vc = static_cast<char*>(const_cast<void*>(v_cache));
      ap.k = static_cast<const char*>(k_new);
      ap.v = static_cast<const char*>(v_new);
      ap.bt = block_tables;
      ap.ctx = context_lens;
      ap.src_stride_seq = new_stride_seq;
      ap.s_blk = kv_stride_block;
      ap.s_head = kv_stride_head;
      ap.s_tok = kv_stride_tok;
      ap.B = B;
      ap.Hkv = Hkv;
      ap.lpt = D / epl;
      ap.ebytes = 16 / epl;
      ap.block_size = block_size;
      ap.bs_shift = p.bs_shift;
      ap.max_blocks = max_blocks;
      ap.num_blocks = num_blocks;
      ap.max_ctx = max_context;
      const int64_t threads = (int64_t)B * Hkv * ap.lpt;
      hipLaunchKernelGGL(kv_append_kernel, dim3((unsigned)ceil_div64(threads, 256)), dim3(256), 0, s, ap);
      if (int rc = check_launch("kv_append_kernel")) return rc;
    } else {
      p.k_new = k_new;
      p.v_new = v_new;
      p.new_stride_seq = new_stride_seq;
      p.lds_new_off = (int)pl.lds_bytes;
      pl.lds_bytes += stash;
      p.rope_cos = rope_cos;
      p.rope_sin = rope_sin;
      p.rope_len = rope_len;
    }
  }
  if (pl.mfma) return launch_gqa(p, pl, D, dtype, s);
  switch (dtype) {
    case MI355_F32: return dispatch_d<F32>(D, p, pl, s);
    case MI355_BF16: return dispatch_d<BF16>(D, p, pl, s);
    case MI355_F16: return dispatch_d<F16>(D, p, pl, s);
  }
  return fail(MI355_EINVAL, "%s: dtype %d", what, dtype);
}

// Split-KV partials of mi355_paged_decode_static: one array per GPU, held until the process ends.
// The name below is only a host-side handle; kernels get the per-device address that
// hipGetSymbolAddress reports.
static __device__ __align__(256) unsigned char scratch[MI355_STATIC_SCRATCH_BYTES];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices];   // device address of `scratch` on each GPU, filled on first use

// Address of `scratch` on the current device (looked up once per device), or nullptr + error.
void* static_scratch(int* rc) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    *rc = fail(e != hipSuccess ? (int)e : MI355_EUNSUPPORTED, "paged_decode_static: device %d (%s)", dev,
               hipGetErrorString(e));
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void* ptr = nullptr;
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) {
      *rc = fail(e != hipSuccess ? (int)e : MI355_EUNSUPPORTED,
                 "paged_decode_static: hipGetSymbolAddress on device %d: %s", dev, hipGetErrorString(e));
      return nullptr;
    }
    g_scratch_ptr[dev] = ptr;
  }
  return g_scratch_ptr[dev];
}

}  // namespace
}  // namespace mi355

extern "C" int mi355_paged_decode(void* out, const void* q, int64_t q_stride_seq, const void* k_cache,
                                  const void* v_cache, const int32_t* block_tables,
                                  const int32_t* context_lens, int32_t B, int32_t Hq, int32_t Hkv,
                                  int32_t D, int32_t block_size, int32_t max_blocks, int32_t num_blocks,
                                  int32_t max_context, int64_t kv_stride_block, int64_t kv_stride_head,
                                  int64_t kv_stride_tok, float scale, int32_t dtype, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  return paged_decode_impl("paged_decode", out, q, q_stride_seq, k_cache, v_cache, block_tables, context_lens, B, Hq,
                           Hkv, D, block_size, max_blocks, num_blocks, max_context, kv_stride_block, kv_stride_head,
                           kv_stride_tok, scale, dtype, workspace, workspace_bytes, stream);
}

extern "C" int mi355_paged_decode_append(void* out, const void* q, int64_t q_stride_seq, const void* k_new,
                                         const void* v_new, int64_t new_stride_seq, void* k_cache, void* v_cache,
                                         const int32_t* block_tables, const int32_t* context_lens, int32_t B,
                                         int32_t Hq, int32_t Hkv, int32_t D, int32_t block_size,
                                         int32_t max_blocks, int32_t num_blocks, int32_t max_context,
                                         int64_t kv_stride_block, int64_t kv_stride_head, int64_t kv_stride_tok,
                                         float scale, int32_t dtype, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  if (B == 0) return 0;
  if (!k_new || !v_new) return fail(MI355_EINVAL, "paged_decode_append: null pointer argument");
  return paged_decode_impl("paged_decode_append", out, q, q_stride_seq, k_cache, v_cache, block_tables, context_lens,
                           B, Hq, Hkv, D, block_size, max_blocks, num_blocks, max_context, kv_stride_block,
                           kv_stride_head, kv_stride_tok, scale, dtype, workspace, workspace_bytes, stream, k_new,
                           v_new, new_stride_seq);
}

extern "C" int mi355_paged_decode_append_rope(void* out, const void* q, int64_t q_stride_seq, const void* k_new,
                                              const void* v_new, int64_t new_stride_seq, void* k_cache,
                                              void* v_cache, const int32_t* block_tables,
                                              const int32_t* context_lens, const void* cos_table,
                                              const void* sin_table, int32_t table_len, int32_t B, int32_t Hq,
                                              int32_t Hkv, int32_t D, int32_t block_size, int32_t max_blocks,
                                              int32_t num_blocks, int32_t max_context, int64_t kv_stride_block,
                                              int64_t kv_stride_head, int64_t kv_stride_tok, float scale,
                                              int32_t dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (B == 0) return 0;
  if (!k_new || !v_new || !cos_table || !sin_table)
    return fail(MI355_EINVAL, "paged_decode_append_rope: null pointer argument");
  if (table_len <= 0 || (((uintptr_t)cos_table | (uintptr_t)sin_table) & 15))
    return fail(MI355_EINVAL, "paged_decode_append_rope: cos / sin tables must be [table_len, D], 16-byte aligned");
  return paged_decode_impl("paged_decode_append_rope", out, q, q_stride_seq, k_cache, v_cache, block_tables,
                           context_lens, B, Hq, Hkv, D, block_size, max_blocks, num_blocks, max_context,
                           kv_stride_block, kv_stride_head, kv_stride_tok, scale, dtype, workspace, workspace_bytes,
                           stream, k_new, v_new, new_stride_seq, cos_table, sin_table, table_len);
}

extern "C" int mi355_paged_decode_static(void* out, const void* q, int64_t q_stride_seq, const void* k_cache,
                                         const void* v_cache, const int32_t* block_tables,
                                         const int32_t* context_lens, int32_t B, int32_t Hq, int32_t Hkv,
                                         int32_t D, int32_t block_size, int32_t max_blocks, int32_t num_blocks,
                                         int32_t max_context, int64_t kv_stride_block, int64_t kv_stride_head,
                                         int64_t kv_stride_tok, float scale, int32_t dtype, void* stream) {
  if (B == 0) return 0;
  if (max_blocks > 0 && block_size > 0 && (int64_t)max_context > (int64_t)max_blocks * block_size)
    max_context = max_blocks * block_size;   // the plan is made for the clamped context, as in the body
  DecodePlan pl;
  if (int rc = make_plan(B, Hq, Hkv, D, block_size, max_context, dtype, &pl)) return rc;   // (argument errors)
  const size_t need = decode_workspace_bound(B, Hq, Hkv, D, block_size, max_context, dtype);
  if (need > (size_t)MI355_STATIC_SCRATCH_BYTES)
    return fail(MI355_EWORKSPACE,
                "paged_decode_static: this shape needs %zu bytes of split-KV partials, the built-in scratch holds %zu; "
                "use mi355_paged_decode with a caller workspace",
                need, (size_t)MI355_STATIC_SCRATCH_BYTES);
  void* ptr = nullptr;
  if (need) {   // unsplit shapes never touch the array
    int rc = 0;
    ptr = static_scratch(&rc);
    if (!ptr) return rc;
  }
  return paged_decode_impl("paged_decode_static", out, q, q_stride_seq, k_cache, v_cache, block_tables, context_lens,
                           B, Hq, Hkv, D, block_size, max_blocks, num_blocks, max_context, kv_stride_block,
                           kv_stride_head, kv_stride_tok, scale, dtype, ptr, ptr ? (size_t)MI355_STATIC_SCRATCH_BYTES : 0,
                           stream);
}

extern "C" int32_t mi355_paged_prefill_max_context(int32_t dtype, int32_t D, int32_t block_size, int32_t max_q_len) {
  if (dtype != MI355_F32 && dtype != MI355_BF16 && dtype != MI355_F16) return 0;
  if (D != 16 && D != 32 && D != 64 && D != 128 && D != 256) return 0;
  if (block_size <= 0 || max_q_len < 0) return 0;
  const bool many = max_q_len >= knob(kKnobPagedPrefillMinQ);
  if (many && (prefill_mfma_supported(dtype, D) || (prefill_mfma_f32_supported(dtype, D) && knob(kKnobPrefillF32Mfma))))
    return INT32_MAX;
  // the staging kernels round the context up to their token quantum (a power of two <= 1024) before
  // they size the table slice: kMaxBtEntries - 2 blocks, rounded down to a multiple of every quantum
  const int64_t staged = ((int64_t)(kMaxBtEntries - 2) * block_size / 1024) * 1024;
  return (int32_t)(staged < INT32_MAX ? staged : INT32_MAX);
}

extern "C" int mi355_paged_prefill(void* out, const void* q, int64_t q_stride_tok, const void* k_cache,
                                   const void* v_cache, const int32_t* block_tables, const int32_t* seg_q_start,
                                   const int32_t* seg_len, const int32_t* seg_pos0, int32_t nseg,
                                   int32_t max_q_len, int32_t Hq, int32_t Hkv, int32_t D, int32_t block_size,
                                   int32_t max_blocks, int32_t num_blocks, int32_t max_context,
                                   int64_t kv_stride_block, int64_t kv_stride_head, int64_t kv_stride_tok,
                                   float scale, int32_t dtype, void* stream) {
  if (nseg == 0 || max_q_len == 0) return 0;
  if (!out || !q || !k_cache || !v_cache || !block_tables || !seg_q_start || !seg_len || !seg_pos0)
    return fail(MI355_EINVAL, "paged_prefill: null pointer argument");
  if (nseg < 0 || max_q_len < 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv || block_size <= 0 || max_blocks <= 0 ||
      num_blocks <= 0 || max_context <= 0)
    return fail(MI355_EINVAL, "paged_prefill: bad shape");
  const int gq = Hq / Hkv;
  if ((int64_t)max_context > (int64_t)max_blocks * block_size) max_context = max_blocks * block_size;
  // fp32 chunks of many query tokens at head_dim 32 / 64 / 128: the fp32 matrix-core flash kernel
  if (prefill_mfma_f32_supported(dtype, D) && scale > 0.f && max_q_len >= knob(kKnobPagedPrefillMinQ) &&
      knob(kKnobPrefillF32Mfma) && kv_stride_block > 0 && kv_stride_head >= 0 && kv_stride_tok > 0 &&
      kv_stride_block % 4 == 0 && kv_stride_head % 4 == 0 && kv_stride_tok % 4 == 0 && q_stride_tok % 4 == 0 &&
      q_stride_tok >= (int64_t)Hq * D && kv_stride_block < (1LL << 31) &&
      (kv_stride_tok * (int64_t)block_size + (int64_t)Hkv * kv_stride_head + D) * 4 < (1LL << 32) &&
      !(((uintptr_t)out | (uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache) & 15))
    return prefill_mfma_f32_paged_launch(out, q, q_stride_tok, k_cache, v_cache, block_tables, seg_q_start, seg_len,
                                         seg_pos0, nseg, max_q_len, Hq, Hkv, D, block_size, max_blocks, num_blocks,
                                         max_context, kv_stride_block, kv_stride_head, kv_stride_tok, scale,
                                         static_cast<hipStream_t>(stream));
  if (!decode_gqa_supported(dtype, D, 4))   // fp32, head dims off the matrix-core path: VALU kernel, row groups
    return paged_prefill_rows(out, q, q_stride_tok, k_cache, v_cache, block_tables, seg_q_start, seg_len, seg_pos0,
                              nseg, max_q_len, Hq, Hkv, D, block_size, max_blocks, num_blocks, max_context,
                              kv_stride_block, kv_stride_head, kv_stride_tok, scale, dtype,
                              static_cast<hipStream_t>(stream));
  if (kv_stride_block <= 0 || kv_stride_head < 0 || kv_stride_tok <= 0 || kv_stride_block >= (1LL << 31) ||
      kv_stride_tok >= (1LL << 31) || (int64_t)Hkv * kv_stride_head + D >= (1LL << 31))
    return fail(MI355_EINVAL, "paged_prefill: kv strides out of range");
  if (kv_stride_block % 8 || kv_stride_head % 8 || kv_stride_tok % 8 || q_stride_tok % 8 ||
      q_stride_tok < (int64_t)Hq * D)
    return fail(MI355_EINVAL, "paged_prefill: strides must keep head rows 16-byte aligned");
  if (((uintptr_t)out | (uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache) & 15)
    return fail(MI355_EINVAL, "paged_prefill: tensors must be 16-byte aligned");

  // Chunks of many query tokens are dense work: the flash-attention kernel shares every K/V tile
  // among 128 or 256 query rows through LDS.  Short chunks (a few tokens per sequence) keep the
  // decode-shaped kernel below, whose tiles are 16 query columns wide.
  // MI355_PAGED_PREFILL_MIN_Q moves the switch (A/B measurements).
  if (prefill_mfma_supported(dtype, D) && scale > 0.f && max_q_len >= knob(kKnobPagedPrefillMinQ) &&
      kv_stride_tok * 2 * (int64_t)block_size < (1LL << 32))
    return prefill_mfma_paged_launch(out, q, q_stride_tok, k_cache, v_cache, block_tables, seg_q_start, seg_len,
                                     seg_pos0, nseg, max_q_len, Hq, Hkv, D, block_size, max_blocks, num_blocks,
                                     max_context,
                                     kv_stride_block, kv_stride_head, kv_stride_tok, scale, dtype,
                                     static_cast<hipStream_t>(stream));

  // Tile columns: tq query tokens x hgp (>= heads of the kv group, power of two, <= 16) heads.
  int hgp_shift = 0;
  while ((1 << hgp_shift) < gq && hgp_shift < 4) ++hgp_shift;
  DecodePlan pl;
  pl.mfma = true;
  pl.nw = kGqaWaves;
  pl.G = kGqaHeads;
  pl.qchunks = gq > kGqaHeads ? (gq + kGqaHeads - 1) / kGqaHeads : 1;
  const int tq = kGqaHeads >> hgp_shift;
  const int n_qt = (max_q_len + tq - 1) / tq;
  const int64_t tiles = (int64_t)nseg * n_qt;
  pl.hpw = pl.nw;
  while (pl.hpw > 1 && Hkv % pl.hpw) pl.hpw /= 2;
  while (pl.hpw > 1 && tiles * (Hkv / pl.hpw) * pl.qchunks < 512 &&
         max_context >= 2 * kGqaTokens * (pl.nw / pl.hpw) * 2)
    pl.hpw /= 2;
  pl.sub = pl.nw / pl.hpw;
  pl.hgroups = Hkv / pl.hpw;
  const int quantum = pl.sub * kGqaTokens;
  pl.tps = (int)(ceil_div64(max_context, quantum) * quantum);   // no token splits: the tiles already fill the chip
  pl.nsplit = 1;
  if ((int64_t)pl.tps / block_size + 2 > kMaxBtEntries)
    return fail(MI355_EUNSUPPORTED, "paged_prefill: context %d with block_size %d exceeds the staged block table",
                max_context, block_size);
  if (tiles * pl.hgroups * pl.qchunks >= (1LL << 31)) return fail(MI355_EINVAL, "paged_prefill: grid too large");
  pl.lds_bt_bytes = (size_t)((pl.tps / block_size + 2) * 4 + 15) / 16 * 16;
  pl.lds_bytes = decode_gqa_lds_bytes(D, pl.sub, pl.lds_bt_bytes);
  pl.ws_bytes = 0;

  DecodeParams p;
  p.out = out;
  p.q = q;
  p.kc = k_cache;
  p.vc = v_cache;
  p.bt = block_tables;
  p.ctx = nullptr;
  p.cu = nullptr;
  p.nseq = 0;
  p.ws_acc = nullptr;
  p.ws_ml = nullptr;
  p.q_stride_seq = q_stride_tok;
  p.s_blk = kv_stride_block;
  p.s_head = kv_stride_head;
  p.s_tok = kv_stride_tok;
  p.B = (int)tiles;
  p.Hq = Hq;
  p.Hkv = Hkv;
  p.gq = gq;
  p.block_size = block_size;
  p.bs_shift = ilog2_exact(block_size);
  p.max_blocks = max_blocks;
  p.num_blocks = num_blocks;
  p.max_ctx = max_context;
  p.hpw = pl.hpw;
  p.sub = pl.sub;
  p.hgroups = pl.hgroups;
  p.qchunks = pl.qchunks;
  p.tps = pl.tps;
  p.nsplit = 1;
  p.lds_bt_bytes = (int)pl.lds_bt_bytes;
  p.scale_log2e = scale * 1.4426950408889634f;
  p.seg_q0 = seg_q_start;
  p.seg_len = seg_len;
  p.seg_pos0 = seg_pos0;
  p.n_qt = n_qt;
  p.tq = tq;
  p.hgp_shift = hgp_shift;
  p.max_q = max_q_len;
  return append_gqa_launch(p, pl, D, dtype, static_cast<hipStream_t>(stream));
}

extern "C" int mi355_prefill_causal(void* out, const void* q, const void* k, const void* v,
                                    int64_t q_stride_tok, int64_t kv_stride_tok, const int32_t* cu_seqlens,
                                    int32_t nseq, int32_t total_tokens, int32_t max_seqlen, int32_t Hq,
                                    int32_t Hkv, int32_t D, float scale, int32_t dtype, void* stream) {
  if (total_tokens == 0 || nseq == 0) return 0;
  if (!out || !q || !k || !v || !cu_seqlens) return fail(MI355_EINVAL, "prefill_causal: null pointer");
  if (total_tokens < 0 || nseq < 0 || max_seqlen <= 0) return fail(MI355_EINVAL, "prefill_causal: bad sizes");
  DecodePlan pl;
  if (int rc = make_plan(total_tokens, Hq, Hkv, D, 1, max_seqlen, dtype, &pl, /*causal=*/true)) return rc;
  const int epl = elems_per_lane(dtype);
  if (q_stride_tok % epl || kv_stride_tok % epl || q_stride_tok < (int64_t)Hq * D ||
      kv_stride_tok < (int64_t)Hkv * D || kv_stride_tok >= (1LL << 31))
    return fail(MI355_EINVAL, "prefill_causal: strides must keep head rows 16-byte aligned");
  if (((uintptr_t)out | (uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15)
    return fail(MI355_EINVAL, "prefill_causal: tensors must be 16-byte aligned");
  // 16-bit inputs at head_dim 64/128 go to the matrix-core kernel; everything else (fp32,
  // other head dims) runs the VALU decode kernel in causal mode below.
  // MI355_PREFILL_VALU=1 forces the latter (A/B measurements).  The matrix-core kernel takes
  // the row maximum before scaling, which needs a positive scale, and addresses a tile's rows
  // with 32-bit byte offsets, which bounds the token stride.
  if (prefill_mfma_supported(dtype, D) && scale > 0.f && kv_stride_tok < (1LL << 24) &&
      !knob(kKnobPrefillValu))
    return prefill_mfma_launch(out, q, k, v, q_stride_tok, kv_stride_tok, cu_seqlens, nseq, total_tokens,
                               max_seqlen, Hq, Hkv, D, scale, dtype, static_cast<hipStream_t>(stream));
  // fp32 at head_dim 32 / 64 / 128: the fp32 matrix instruction (prefill_mfma_f32.hip; exact fp32
  // products).  MI355_PREFILL_F32_MFMA=0 keeps the vector-ALU kernels below (A/B measurements).
  if (prefill_mfma_f32_supported(dtype, D) && scale > 0.f && kv_stride_tok < (1LL << 24) &&
      knob(kKnobPrefillF32Mfma) && !knob(kKnobPrefillValu))
    return prefill_mfma_f32_launch(out, q, k, v, q_stride_tok, kv_stride_tok, cu_seqlens, nseq, max_seqlen, Hq, Hkv,
                                   D, scale, static_cast<hipStream_t>(stream));
  // fp32 at head_dim 32 / 64 / 128: one query row per lane (prefill_f32.hip).
  // MI355_PREFILL_F32_TILED=0 keeps the decode-shaped kernel below (A/B measurements).
  if (prefill_f32_supported(dtype, D) && knob(kKnobPrefillF32Tiled))
    return prefill_f32_launch(out, q, k, v, q_stride_tok, kv_stride_tok, cu_seqlens, nseq, max_seqlen, Hq, Hkv, D,
                              scale, static_cast<hipStream_t>(stream));
  DecodeParams p;
  p.out = out;
  p.q = q;
  p.kc = k;
  p.vc = v;
  p.bt = nullptr;
  p.ctx = nullptr;
  p.cu = cu_seqlens;
  p.nseq = nseq;
  p.ws_acc = nullptr;
  p.ws_ml = nullptr;
  p.q_stride_seq = q_stride_tok;
  p.s_blk = kv_stride_tok;  // "block" = one token row
  p.s_head = D;
  p.s_tok = 0;
  p.B = total_tokens;
  p.Hq = Hq;
  p.Hkv = Hkv;
  p.gq = Hq / Hkv;
  p.block_size = 1;
  p.bs_shift = 0;
  p.max_blocks = max_seqlen;
  p.num_blocks = total_tokens;
  p.max_ctx = max_seqlen;
  p.hpw = pl.hpw;
  p.sub = pl.sub;
  p.hgroups = pl.hgroups;
  p.qchunks = pl.qchunks;
  p.tps = pl.tps;
  p.nsplit = 1;
  p.lds_bt_bytes = (int)pl.lds_bt_bytes;
  p.scale_log2e = scale * 1.4426950408889634f;
  p.row_tiles = 0;
  p.seg_q0 = p.seg_len = p.seg_pos0 = nullptr;
  p.n_qt = p.tq = p.hgp_shift = p.max_q = 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // Row groups: one workgroup per (sequence, tile of kRowGroup query rows, kv-head group, query head of
  // the kv group).  MI355_PREFILL_ROW_GROUPS=0 keeps one query row per workgroup (A/B measurements).
  const size_t rows_lds = 16 + (pl.sub > 1 ? (size_t)pl.nw * kRowGroup * (D + 2) * sizeof(float) : 0);
  const int64_t row_tiles = ((int64_t)max_seqlen + kRowGroup - 1) / kRowGroup;
  const int64_t rows_grid = (int64_t)nseq * row_tiles * pl.hgroups * p.gq;
  if (knob(kKnobPrefillRowGroups) && rows_lds <= kRowsMaxLds && rows_grid < (1LL << 31)) {
    p.row_tiles = (int)row_tiles;
    p.B = (int)(nseq * row_tiles);
    p.qchunks = p.gq;
    p.lds_bt_bytes = 16;
    switch (dtype) {
      case MI355_F32: return dispatch_causal_rows<F32>(D, p, rows_lds, s);
      case MI355_BF16: return dispatch_causal_rows<BF16>(D, p, rows_lds, s);
      case MI355_F16: return dispatch_causal_rows<F16>(D, p, rows_lds, s);
    }
  }
  switch (dtype) {
    case MI355_F32: return dispatch_d<F32>(D, p, pl, s);
    case MI355_BF16: return dispatch_d<BF16>(D, p, pl, s);
    case MI355_F16: return dispatch_d<F16>(D, p, pl, s);
  }
  return fail(MI355_EINVAL, "prefill_causal: dtype %d", dtype);
}
