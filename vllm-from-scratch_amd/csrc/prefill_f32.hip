// Causal prompt attention in fp32 on the vector ALUs: one query row per lane.
//
// The reference's default engine runs fp32 at head_dim 32 (05_mini_vllm/mini_vllm.py:437-458 is the
// prompt branch this serves) and its tests compare at 1e-5, so this path keeps fp32 products and
// sums.  Layout: a wave owns 64 consecutive query rows of one (sequence, query head); lane l holds
// row l's query and output accumulator in registers.  Every lane needs the same key row at the same
// time: a key row is fetched by ONE coalesced load that leaves element c in lane c, and each element
// is handed to all lanes as a scalar operand of the FMA (v_readlane), so a (row, key) pair costs 2 D
// FMAs + 2 D lane reads, no LDS traffic and no shuffle tree -- the decode-shaped kernel, whose lanes
// split a row's dims, pays a cross-lane reduction per score.  Two other ways of sharing a key row
// were measured first and both stopped at 16-17 TFLOP/s: rows staged in LDS and read back as
// broadcasts (a broadcast still returns 16 bytes to each of 64 lanes; the LDS return path, shared by
// the CU's four SIMDs, is the bound) and scalar loads through the constant address space (one row
// fills the scalar registers, so every row's load latency is exposed, and the scalar cache misses
// on every row).  Scores are taken kBlock keys at a time: one running-maximum update and one
// rescale of the accumulator per block, then the kBlock weighted value rows.
// head_dim 128 keeps 2 x 128 values per lane: the compiler parks part of them in the accumulation
// registers and the kernel runs one wave per SIMD (29 TFLOP/s against 42 at head_dim 64).
#include "common.h"

namespace mi355 {
namespace {

constexpr int kBlock = 8;    // keys per softmax update
constexpr float kNegBigF = -1.0e30f;

__device__ __forceinline__ float lane_value(float v, int lane) {   // v of lane `lane` (a constant), as a scalar
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

struct PrefillF32Params {
  float* out;
  const float* q;
  const float* k;
  const float* v;
  const int32_t* cu;
  int64_t q_stride_tok, kv_stride_tok;
  int Hq, gq, row_tiles, nseq;
  float scale_log2e;
};

template <int D>
__global__ void __launch_bounds__(kWave) prefill_f32_kernel(const PrefillF32Params p) {
  const int lane = threadIdx.x;
  // Workgroups are dispatched in index order and a row tile's work grows with its position in the
  // sequence, so the index runs over the tiles from the last (longest) to the first, all sequences
  // and heads of one tile position together: the long waves start first and the short ones fill
  // the tail.  (In (sequence, head, tile ascending) order the chip averaged 0.8 waves per SIMD.)
  const int per_tile = p.nseq * p.Hq;
  const int tile = p.row_tiles - 1 - (int)(blockIdx.x / per_tile);
  const int seq = (int)(blockIdx.x % per_tile) / p.Hq;
  const int hq = (int)(blockIdx.x % per_tile) % p.Hq, hk = hq / p.gq;
  const int start = p.cu[seq], len = p.cu[seq + 1] - start;
  const int r0 = tile * kWave;
  if (r0 >= len) return;                       // the grid is sized for the longest sequence
  const int row = r0 + lane;                   // this lane's position in its sequence
  const bool live = row < len;
  const int row_c = live ? row : len - 1;      // lanes past the end compute a duplicate that is not stored

  float qv[D], acc[D];
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(p.q + (int64_t)(start + row_c) * p.q_stride_tok + (int64_t)hq * D);
#pragma unroll
    for (int c = 0; c < D / 4; ++c) {
      const f32x4 t = src[c];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qv[c * 4 + e] = t[e];
        acc[c * 4 + e] = 0.f;
      }
    }
  }
  float m = kNegBigF, l = 0.f;

  const int nkeys = min(len, r0 + kWave);      // keys any row of this wave may see
  // one load instruction fetches kWave / D whole rows -- lane (r * D + c) gets element c of row r -- or,
  // at head_dim 128, half a row (two loads per row: elements lane and 64 + lane)
  constexpr int KPL = D <= kWave ? kWave / D : 1, LPR = D <= kWave ? 1 : D / kWave, NL = kBlock / KPL * LPR;
  constexpr int DL = D <= kWave ? D : kWave;    // elements of one row held by one load
  const int lrow = lane / DL, lcol = lane % DL;
  const float* kbase = p.k + (int64_t)start * p.kv_stride_tok + (int64_t)hk * D + lcol;
  const float* vbase = p.v + (int64_t)start * p.kv_stride_tok + (int64_t)hk * D + lcol;
  auto fetch = [&](int b0, float (&kr)[NL], float (&vr)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int64_t off = (int64_t)min(b0 + (i / LPR) * KPL + lrow, len - 1) * p.kv_stride_tok + (i % LPR) * kWave;   // rows past the end: masked below
      kr[i] = kbase[off];
      vr[i] = vbase[off];
    }
  };
  float kcur[NL], vcur[NL], knext[NL], vnext[NL];
  fetch(0, kcur, vcur);
  for (int b0 = 0; b0 < nkeys; b0 += kBlock) {
    if (b0 + kBlock < nkeys) fetch(b0 + kBlock, knext, vnext);   // in flight during this block's arithmetic
    float s[kBlock];
#pragma unroll
    for (int j = 0; j < kBlock; ++j) {
      float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;   // four chains keep the FMA pipe busy
#pragma unroll
      for (int c = 0; c < D; c += 4) {
        const int reg = (j / KPL) * LPR + c / kWave, at = (j % KPL) * DL + c % kWave;
        d0 = __builtin_fmaf(qv[c], lane_value(kcur[reg], at), d0);
        d1 = __builtin_fmaf(qv[c + 1], lane_value(kcur[reg], at + 1), d1);
        d2 = __builtin_fmaf(qv[c + 2], lane_value(kcur[reg], at + 2), d2);
        d3 = __builtin_fmaf(qv[c + 3], lane_value(kcur[reg], at + 3), d3);
      }
      s[j] = b0 + j <= row_c ? ((d0 + d1) + (d2 + d3)) * p.scale_log2e : -__builtin_inff();
    }
    float mx = s[0];
#pragma unroll
    for (int j = 1; j < kBlock; ++j) mx = fmaxf(mx, s[j]);
    const float m_new = fmaxf(m, mx);          // stays finite: m starts at -1e30
    const float f = __builtin_amdgcn_exp2f(m - m_new);
    m = m_new;
    l *= f;
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] *= f;
#pragma unroll
    for (int j = 0; j < kBlock; ++j) {
      const float w = __builtin_amdgcn_exp2f(s[j] - m);   // masked keys: exp2(-inf) = 0
      l += w;
#pragma unroll
      for (int c = 0; c < D; ++c)
        acc[c] = __builtin_fmaf(w, lane_value(vcur[(j / KPL) * LPR + c / kWave], (j % KPL) * DL + c % kWave), acc[c]);
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      kcur[i] = knext[i];
      vcur[i] = vnext[i];
    }
  }
  if (!live) return;
  const float inv = 1.0f / l;                  // l >= 1: every row sees at least its own key
  f32x4* dst = reinterpret_cast<f32x4*>(p.out + ((int64_t)(start + row) * p.Hq + hq) * D);
#pragma unroll
  for (int c = 0; c < D / 4; ++c)
    dst[c] = f32x4{acc[c * 4] * inv, acc[c * 4 + 1] * inv, acc[c * 4 + 2] * inv, acc[c * 4 + 3] * inv};
}

template <int D>
int launch(const PrefillF32Params& p, int nseq, hipStream_t stream) {
  hipLaunchKernelGGL(prefill_f32_kernel<D>, dim3((unsigned)((int64_t)nseq * p.row_tiles * p.Hq)), dim3(kWave), 0,
                     stream, p);
  return check_launch("prefill_f32_kernel");
}

}  // namespace

bool prefill_f32_supported(int dtype, int D) { return dtype == MI355_F32 && (D == 32 || D == 64 || D == 128); }

int prefill_f32_launch(void* out, const void* q, const void* k, const void* v, int64_t q_stride_tok,
                       int64_t kv_stride_tok, const int32_t* cu_seqlens, int nseq, int max_seqlen, int Hq, int Hkv,
                       int D, float scale, hipStream_t stream) {
  PrefillF32Params p;
  p.out = static_cast<float*>(out);
  p.q = static_cast<const float*>(q);
  p.k = static_cast<const float*>(k);
  p.v = static_cast<const float*>(v);
  p.cu = cu_seqlens;
  p.q_stride_tok = q_stride_tok;
  p.kv_stride_tok = kv_stride_tok;
  p.Hq = Hq;
  p.gq = Hq / Hkv;
  p.row_tiles = (max_seqlen + kWave - 1) / kWave;
  p.scale_log2e = scale * 1.4426950408889634f;
  p.nseq = nseq;
  if ((int64_t)nseq * p.row_tiles * Hq >= (1LL << 31))
    return fail(MI355_EINVAL, "prefill_causal: grid too large (%d sequences x %d row tiles, %d heads)", nseq,
                p.row_tiles, Hq);
  return D == 32 ? launch<32>(p, nseq, stream) : D == 64 ? launch<64>(p, nseq, stream) : launch<128>(p, nseq, stream);
}

}  // namespace mi355
