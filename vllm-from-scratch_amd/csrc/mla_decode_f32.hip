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
};

__device__ __forceinline__ f32x4v mma4(float a, float b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
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

template <int R, int P, int WAVES, bool VEC>
__global__ void __launch_bounds__(WAVES * 64) mla_decode_f32_kernel(const MlaF32Params p) {
  constexpr int Q = R / WAVES;            // latent dims per wave
  constexpr int NJ = Q / 16;              // 16-dim groups per wave
  constexpr int PR = P / (2 * WAVES);     // rope dims per wave in each half
  constexpr int RS = 2 * PR / 4;          // rope dims per lane quarter = rope contraction steps per wave
  constexpr int PITCH = Q + 4;            // words per token row of the transposed-read image
  static_assert(Q % 16 == 0 && PR >= 2 && PR == 2 * RS, "shape");
  __shared__ __attribute__((aligned(16))) float s_part[WAVES > 1 ? 2 : 1][WAVES][256];
  __shared__ __attribute__((aligned(16))) float s_v[WAVES][16 * PITCH];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  int bid = blockIdx.x;
  const int ht = bid % p.n_ht;
  bid /= p.n_ht;
  const int sp = bid % p.nsplit, row = bid / p.nsplit;
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

  const float* ckv_seq = p.ckv + (int64_t)seq * p.ckv_stride_seq + Q * wave;
  const float* kr_seq = p.krope + (int64_t)seq * p.krope_stride_seq;

  // A tile's loads: this lane's token row (clamped to the context; rows past it are masked below)
  f32x4v kk[NJ], kk_n[NJ];
  float rv[RS], rc[RS], rn[RS], rv_n[RS], rc_n[RS], rn_n[RS];
  auto load_tile = [&](int t0, f32x4v (&k4)[NJ], float (&v)[RS], float (&cs)[RS], float (&sn)[RS]) {
    const int key = min(t0 + c, max(ctx - 1, 0));           // (t0 counts from token 0 of the sequence)
    const float* krow = ckv_seq + (int64_t)key * p.ckv_stride_tok;
#pragma unroll
    for (int j = 0; j < NJ; ++j) k4[j] = *reinterpret_cast<const f32x4v*>(krow + 16 * j + 4 * g);
    const float* rrow = kr_seq + (int64_t)key * p.krope_stride_tok;
    const int trow = min(key, p.table_len - 1) * P;
    load_run<RS, VEC>(rrow + rbase, v);
    load_run<RS, VEC>(p.cos_t + trow + rbase, cs);
    load_run<RS, VEC>(p.sin_t + trow + rbase, sn);
  };

  const int ntiles = ctx > tok0 ? (ctx - tok0 + 15) / 16 : 0;
  if (ntiles > 0) load_tile(tok0, kk, rv, rc, rn);
  float* my_v = s_v[wave];
  for (int t = 0; t < ntiles; ++t) {
    const int t0 = tok0 + t * 16;
    if (t + 1 < ntiles) load_tile(t0 + 16, kk_n, rv_n, rc_n, rn_n);
    // ---- this wave's share of the scores: 16 tokens x 16 heads over its latent quarter and rope dims,
    // on two accumulators (one chain of dependent matrix instructions would run at their latency, not
    // at their issue rate)
    f32x4v sp = {0.f, 0.f, 0.f, 0.f}, sp2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; j += 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sp = mma4(kk[j][e], ql[j][e], sp);
        if (j + 1 < NJ) sp2 = mma4(kk[j + 1][e], ql[j + 1][e], sp2);
      }
    }
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      const float partner = __shfl_xor(rv[s], 32);     // the same token's dim rbase + s -+ P/2
      const float roped = __builtin_fmaf(rsign * partner, rn[s], rv[s] * rc[s]);
      if (s & 1) sp2 = mma4(roped, qr[s], sp2);
      else sp = mma4(roped, qr[s], sp);
    }
    sp += sp2;
    // the quarter tile, row-major, for the transposed reads of the value product
#pragma unroll
    for (int j = 0; j < NJ; ++j) *reinterpret_cast<f32x4v*>(my_v + c * PITCH + 16 * j + 4 * g) = kk[j];
    f32x4v sc = sp;
    if constexpr (WAVES > 1) {
      *reinterpret_cast<f32x4v*>(&s_part[t & 1][wave][lane * 4]) = sp;
      __syncthreads();
      sc = *reinterpret_cast<const f32x4v*>(&s_part[t & 1][0][lane * 4]);
#pragma unroll
      for (int w = 1; w < WAVES; ++w) sc += *reinterpret_cast<const f32x4v*>(&s_part[t & 1][w][lane * 4]);
    }
    // ---- online softmax; lane (c, g) holds tokens t0 + 4 g + r of head c
    float v[4], mx = -__builtin_inff();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = (t0 + 4 * g + r < ctx) ? sc[r] * p.scale_log2e : -__builtin_inff();
      mx = fmaxf(mx, v[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float f = fast_exp2(m_run - m_new);
    m_run = m_new;
    float pr[4], ps = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pr[r] = fast_exp2(v[r] - m_new);
      ps += pr[r];
    }
    ps += __shfl_xor(ps, 16);
    ps += __shfl_xor(ps, 32);
    l_run = l_run * f + ps;
#pragma unroll
    for (int j = 0; j < NJ; ++j) o[j] *= f;
    // ---- out^T += V^T . P^T over this wave's latent quarter
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < NJ; ++j) o[j] = mma4(my_v[(4 * g + r) * PITCH + 16 * j + c], pr[r], o[j]);
    if (t + 1 < ntiles) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) kk[j] = kk_n[j];
#pragma unroll
      for (int s = 0; s < RS; ++s) {
        rv[s] = rv_n[s];
        rc[s] = rc_n[s];
        rn[s] = rn_n[s];
      }
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
  if (vec) hipLaunchKernelGGL((mla_decode_f32_kernel<R, P, WAVES, true>), g, dim3(WAVES * 64), 0, s, p);
  else hipLaunchKernelGGL((mla_decode_f32_kernel<R, P, WAVES, false>), g, dim3(WAVES * 64), 0, s, p);
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
                   void* workspace, size_t workspace_bytes, hipStream_t stream) {
  MlaF32Params p;
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
                   krope_stride_tok % 4 == 0 && krope_stride_seq % 4 == 0;
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
