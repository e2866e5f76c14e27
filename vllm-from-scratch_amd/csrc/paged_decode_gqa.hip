// Paged decode attention for grouped-query shapes on the matrix cores (bf16 / fp16,
// head_dim 64 or 128, 4 or more query heads per kv head).
//
// With G query heads sharing one kv head, every cached K/V row takes part in G dot products
// and G weighted sums: on the VALU that work grows with G and the kernel stops being
// bandwidth-bound (8 query heads per kv head ran at 2.8 TB/s).  Here the G heads are the N
// dimension (padded to 16) of v_mfma_f32_16x16x32:
//
//   workgroup = 4 waves = (sequence, token split, group of up to 4 kv heads, tile of 16 q heads)
//   wave      = one kv head (x a token sub-range when the group has fewer than 4 heads);
//               waves do not synchronise inside the token loop
//   per step (32 tokens = two 16-row MFMA tiles):
//     K and V rows are loaded with fully coalesced 16-byte-per-lane loads (a wave instruction
//       covers 1 KiB of consecutive row bytes) and stored row-major into wave-private LDS
//       tiles.  Loading rows straight into the MFMA operand layout (lane = token row) was
//       measured first: every instruction then touches 64 separate 16-byte pieces and the
//       address/L1 path, not HBM, sets the pace (3.5 TB/s);
//     S^T[token][head] = K . Q^T, D/32 MFMAs per tile with the A operand read back from LDS
//       (ds_read_b128, lane = token row); softmax statistics of a head are lane-local plus
//       two cross-lane steps (heads sit on lanes 0..15 of each 16-lane row);
//     V comes back transposed through ds_read_b64_tr_b16 as the A operand of
//       O^T[d][head] += V^T . P^T, whose B operand is the score accumulator converted to
//       16 bits in place;
//   two register stages keep the loads of the next two steps in flight.
// Sub-ranges of a workgroup merge through LDS, token splits through the caller's workspace
// and the merge kernel of paged_decode.hip (same record layout as the VALU kernel).
#include <type_traits>

#include "common.h"
#include "decode_merge.h"
#include "decode_params.h"

namespace mi355 {
namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <class Tag>
struct Mma16;
template <>
struct Mma16<BF16> {
  static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                   c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
  }
};
template <>
struct Mma16<F16> {
  static __device__ __forceinline__ f32x4 run(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c,
                                                  0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
  }
};

// APPEND = false: one query token per sequence, a tile's 16 columns are query heads of one kv
// head (decode).  APPEND = true: a tile's columns are tq query tokens of one segment x the
// (padded) heads of the kv group, column by column limited to its own causal context; the
// segment's new K/V rows are already in the cache (chunked prompt processing).
template <class Tag, int D, bool APPEND>
__global__ void __launch_bounds__(kGqaWaves* kWave, 2) paged_decode_gqa_kernel(const DecodeParams p) {
  constexpr int KS = D / 32;          // MFMA k-steps over head_dim
  constexpr int OT = D / 16;          // 16-dim tiles of the output
  constexpr int CPR = D / 8;          // 16-byte chunks per row
  constexpr int RPI = kWave / CPR;    // rows one wave load instruction covers
  constexpr int NL = kGqaTokens / RPI;  // load instructions per tile (K, and again V)
  constexpr int ROWB = D * 2 + 16;    // tile row pitch in LDS (16 B of padding staggers the banks)
  constexpr int kRec = D + 2;         // floats of one (head) partial record: D sums, max, sum
  constexpr int kTileB = kGqaTokens * ROWB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // block-table slice
  __shared__ __attribute__((aligned(16))) unsigned char tiles[kGqaWaves * 2 * kTileB];
  int32_t* s_bt = reinterpret_cast<int32_t*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
  unsigned char* kt = tiles + (wave * 2) * kTileB;          // this wave's K tile
  unsigned char* vt = kt + kTileB;                          // and V tile
  // after the token loop a wave's K tile holds its partial record for the sub-range merge
  static_assert(kGqaHeads * kRec * sizeof(float) <= (size_t)kTileB, "partial record must fit a tile");

  int bid = blockIdx.x;
  const int qc = bid % p.qchunks;
  bid /= p.qchunks;
  const int hg = bid % p.hgroups;
  bid /= p.hgroups;
  const int sp = bid % p.nsplit;
  const int b = bid / p.nsplit;

  // Tile geometry: context the workgroup walks, and this lane's column (query row, head, limit).
  int ctx, qrow = b, bt_row = b, head_in_group = qc * kGqaHeads + c, col_limit;
  bool col_ok;
  if constexpr (APPEND) {
    const int seg = b / p.n_qt, qt = b - seg * p.n_qt;
    // descriptors come from device memory: clamped to the host-side bounds the launch was sized for
    const int slen = min(p.seg_len[seg], p.max_q), tok0 = qt * p.tq;
    if (tok0 >= slen) return;
    const int ntok = min(p.tq, slen - tok0), pos0 = min(max(p.seg_pos0[seg], 0), p.max_ctx);
    const int ctok = c >> p.hgp_shift, chd = c & ((1 << p.hgp_shift) - 1);
    ctx = max(0, min(pos0 + tok0 + ntok, p.max_ctx));          // context of the tile's last query token
    col_limit = min(pos0 + tok0 + ctok + 1, ctx);               // causal: a query sees positions <= its own
    head_in_group = p.hgp_shift == 4 ? qc * kGqaHeads + chd : chd;
    col_ok = ctok < ntok && head_in_group < p.gq;
    qrow = p.seg_q0[seg] + tok0 + min(ctok, ntok - 1);
    bt_row = seg;
  } else {
    ctx = max(0, min(p.ctx[b], p.max_ctx));
    col_limit = ctx;
    col_ok = head_in_group < p.gq;
  }
  const int t0 = sp * p.tps;
  // Splits past the context do nothing and the merge skips them; with the merge in this launch
  // (decode_merge.h) split 0 always takes part, so that an empty sequence still gets its zero row.
  const bool fused = !APPEND && p.counters != nullptr;
  if (p.nsplit > 1 && t0 >= ctx && !(fused && sp == 0)) return;
  if (fused && sp == 0 && tid == 0) tag_arrival_cell(p, b, hg, qc);
  const int wg_end = min(ctx, t0 + p.tps);
  auto block_of = [&](int t) { return p.bs_shift >= 0 ? (t >> p.bs_shift) : (t / p.block_size); };

  // Stage this split's slice of the block table in LDS (ids clamped for memory safety).
  const int lb0 = block_of(t0);
  const int nlb = wg_end > t0 ? block_of(wg_end - 1) - lb0 + 1 : 0;
  for (int i = tid; i < nlb; i += blockDim.x) {
    const int id = p.bt[(int64_t)bt_row * p.max_blocks + lb0 + i];
    s_bt[i] = min(max(id, 0), p.num_blocks - 1);
  }
  __syncthreads();

  const int hslot = wave % p.hpw, sb = wave / p.hpw;
  const int kvh = hg * p.hpw + hslot;
  const int len_w = p.tps / p.sub;
  const int tw0 = t0 + sb * len_w;
  const int tw1 = min(wg_end, tw0 + len_w);
  const int hq = kvh * p.gq + min(head_in_group, p.gq - 1);   // this lane's query head (column c of the tile)
  const bool head_ok = col_ok;

  // Query fragments (B operand): lane (c, g), k-step s holds Q[column c][32s + 8g .. +8).
  u32x4 qf[KS];
  {
    const char* src = static_cast<const char*>(p.q) +
                      ((int64_t)qrow * p.q_stride_seq + (int64_t)hq * D + 8 * g) * 2;
#pragma unroll
    for (int s = 0; s < KS; ++s)
      qf[s] = head_ok ? *reinterpret_cast<const u32x4*>(src + s * 64) : u32x4{0u, 0u, 0u, 0u};
  }

  f32x4 o[OT];
#pragma unroll
  for (int i = 0; i < OT; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = kNegBig, l_run = 0.f;

  const char* kbase = static_cast<const char*>(p.kc);
  const char* vbase = static_cast<const char*>(p.vc);
  const int lrow = lane / CPR, lchunk = lane % CPR;   // this lane's (row within an instruction, chunk)
  const uint32_t lane_off = (uint32_t)kvh * (uint32_t)p.s_head + lchunk * 8;  // host checks it fits
  const int nsteps = tw1 > tw0 ? (tw1 - tw0 + kGqaTokens - 1) / kGqaTokens : 0;
  const int lim = min(tw1, col_limit);   // first token this lane's column does not attend to

  // Load instruction i of a step covers rows i*RPI .. i*RPI + RPI - 1 of the 32-token tile,
  // 16 bytes per lane, consecutive lanes on consecutive bytes of a row.
  auto issue_loads = [&](int step, u32x4 (&kr)[NL], u32x4 (&vr)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int t = min(tw0 + step * kGqaTokens + i * RPI + lrow, tw1 - 1);  // rows past the end re-read the last row
      const int lb = block_of(t);
      const int row = t - lb * p.block_size;
      const uint32_t pid = (uint32_t)s_bt[lb - lb0];
      const uint64_t off = ((uint64_t)pid * (uint64_t)p.s_blk + (uint32_t)row * (uint32_t)p.s_tok + lane_off) * 2;
      kr[i] = load16<true>(kbase + off);
      vr[i] = load16<true>(vbase + off);
    }
  };

  auto process = [&](int step, u32x4 (&kr)[NL], u32x4 (&vr)[NL], auto steady) {
    const int ts0 = tw0 + step * kGqaTokens;
    // ---- rows to the wave's LDS tiles (row-major), registers refilled with the step after next
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      *reinterpret_cast<u32x4*>(kt + (i * RPI + lrow) * ROWB + lchunk * 16) = kr[i];
      *reinterpret_cast<u32x4*>(vt + (i * RPI + lrow) * ROWB + lchunk * 16) = vr[i];
    }
    if constexpr (decltype(steady)::value) {
      issue_loads(step + 2, kr, vr);
    } else {
      if (step + 2 < nsteps) issue_loads(step + 2, kr, vr);
    }
    // ---- S^T[token][head] for the two tiles; lane (c, g) ends with tokens 4g..4g+3 of head c
    f32x4 st[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      st[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(kt + (16 * u + c) * ROWB + (32 * s + 8 * g) * 2);
        st[u] = Mma16<Tag>::run(a, qf[s], st[u]);
      }
    }
    // ---- online softmax
    float s[8], mx = -__builtin_inff();
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = (ts0 + 16 * u + 4 * g + j < lim) ? st[u][j] * p.scale_log2e : -__builtin_inff();
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
      for (int i = 0; i < OT; ++i) o[i] *= f;
    }
    // ---- O^T[d][head] += V^T . P^T.  MFMA k-slot (g, j) is token 4g + j of the first tile for
    // j < 4 and token 16 + 4g + (j - 4) of the second for j >= 4: P^T is the two accumulators as
    // they stand, V^T two transposing 4-row reads per 16-dim tile.
    const u32x4 pb = {Mma16<Tag>::pack2(s[0], s[1]), Mma16<Tag>::pack2(s[2], s[3]),
                      Mma16<Tag>::pack2(s[4], s[5]), Mma16<Tag>::pack2(s[6], s[7])};
    const int qi = c >> 2, pi = c & 3;
    const unsigned char* vrow = vt + (4 * g + qi) * ROWB + (4 * pi) * 2;
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

  // The loads of the steady-state loop are unconditional and issued in a fixed order, so the
  // compiler's count of outstanding loads is exact and a wait never covers the newest stage.
  u32x4 ka[NL], va[NL], kb[NL], vb[NL];
  int step = 0;
  if (nsteps >= 4) {
    issue_loads(0, ka, va);
    issue_loads(1, kb, vb);
    do {
      process(step, ka, va, std::true_type{});
      process(step + 1, kb, vb, std::true_type{});
      step += 2;
    } while (step + 3 < nsteps);
  } else if (nsteps > 0) {
    issue_loads(0, ka, va);
    if (nsteps > 1) issue_loads(1, kb, vb);
  }
  for (; step < nsteps; step += 2) {   // last one to three steps
    process(step, ka, va, std::false_type{});
    if (step + 1 < nsteps) process(step + 1, kb, vb, std::false_type{});
  }

  // lane (c, g) holds O[head c][16i + 4g + j] (unnormalised), m_run / l_run of head c.
  if (p.sub > 1) {  // merge the token sub-ranges handled by different waves of this workgroup
    float* mine = reinterpret_cast<float*>(kt) + (size_t)c * kRec;
#pragma unroll
    for (int i = 0; i < OT; ++i) *reinterpret_cast<f32x4*>(mine + 16 * i + 4 * g) = o[i];
    if (g == 0) {
      mine[D] = m_run;
      mine[D + 1] = l_run;
    }
    __syncthreads();
    if (sb == 0) {
      for (int j = 1; j < p.sub; ++j) {
        const float* other =
            reinterpret_cast<const float*>(tiles + ((hslot + j * p.hpw) * 2) * kTileB) + (size_t)c * kRec;
        const float mo = other[D], lo = other[D + 1];
        const float M = fmaxf(m_run, mo);
        const float fa = fast_exp2(m_run - M), fb = fast_exp2(mo - M);
        m_run = M;
        l_run = l_run * fa + lo * fb;
#pragma unroll
        for (int i = 0; i < OT; ++i) o[i] = o[i] * fa + *reinterpret_cast<const f32x4*>(other + 16 * i + 4 * g) * fb;
      }
    }
  }
  const bool owner = sb == 0 && head_ok;
  if (!owner && !fused) return;
  if (!owner) {
    // (a lane without an output column of its own still takes part in the merge below)
  } else if (p.nsplit == 1) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    char* dst = static_cast<char*>(p.out) + (((int64_t)qrow * p.Hq + hq) * D + 4 * g) * 2;
#pragma unroll
    for (int i = 0; i < OT; ++i) {
      const f32x4 v = o[i] * inv;
      *reinterpret_cast<u32x2*>(dst + i * 32) = u32x2{Mma16<Tag>::pack2(v[0], v[1]), Mma16<Tag>::pack2(v[2], v[3])};
    }
  } else {
    const int64_t rec = ((int64_t)qrow * p.Hq + hq) * p.nsplit + sp;
    float* dst = p.ws_acc + rec * D + 4 * g;
    if (fused) {
#pragma unroll
      for (int i = 0; i < OT; ++i) {
        store_coherent2(dst + 16 * i, o[i][0], o[i][1]);
        store_coherent2(dst + 16 * i + 2, o[i][2], o[i][3]);
      }
      if (g == 0) store_coherent2(p.ws_ml + rec * 2, m_run, l_run);
    } else {
#pragma unroll
      for (int i = 0; i < OT; ++i) *reinterpret_cast<f32x4*>(dst + 16 * i) = o[i];
      if (g == 0) {
        p.ws_ml[rec * 2] = m_run;
        p.ws_ml[rec * 2 + 1] = l_run;
      }
    }
  }
  if constexpr (!APPEND) {
    if (!fused) return;
    const int nvalid = max(1, min(p.nsplit, (ctx + p.tps - 1) / p.tps));
    if (!arrived_last(p, b, hg, qc, nvalid, tid)) return;
    merge_splits_in_workgroup<Tag, D, kGqaWaves * kWave>(p, b, hg, qc, kGqaHeads, nvalid, tid);
  }
}

template <class Tag, int D, bool APPEND>
int launch(const DecodeParams& p, const DecodePlan& pl, hipStream_t stream) {
  const int64_t grid = (int64_t)p.B * p.nsplit * p.hgroups * p.qchunks;
  hipLaunchKernelGGL((paged_decode_gqa_kernel<Tag, D, APPEND>), dim3((unsigned)grid), dim3(kGqaWaves * kWave),
                     pl.lds_bytes, stream, p);
  return check_launch(APPEND ? "paged_decode_gqa_kernel (append)" : "paged_decode_gqa_kernel");
}

template <bool APPEND>
int launch_any(const DecodeParams& p, const DecodePlan& pl, int D, int dtype, hipStream_t stream) {
  if (dtype == MI355_BF16)
    return D == 64 ? launch<BF16, 64, APPEND>(p, pl, stream) : launch<BF16, 128, APPEND>(p, pl, stream);
  return D == 64 ? launch<F16, 64, APPEND>(p, pl, stream) : launch<F16, 128, APPEND>(p, pl, stream);
}

}  // namespace

bool decode_gqa_supported(int dtype, int D, int gq) {
  return (dtype == MI355_BF16 || dtype == MI355_F16) && (D == 64 || D == 128) && gq >= 4;
}

size_t decode_gqa_lds_bytes(int D, int sub, size_t lds_bt_bytes) {
  (void)sub;  // the K / V tiles are static; the sub-range partials reuse them
  (void)D;
  return lds_bt_bytes;
}

int decode_gqa_launch(const DecodeParams& p, const DecodePlan& pl, int D, int dtype, hipStream_t stream) {
  return launch_any<false>(p, pl, D, dtype, stream);
}

int append_gqa_launch(const DecodeParams& p, const DecodePlan& pl, int D, int dtype, hipStream_t stream) {
  return launch_any<true>(p, pl, D, dtype, stream);
}

}  // namespace mi355
