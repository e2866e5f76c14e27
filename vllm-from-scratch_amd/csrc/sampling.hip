// Token sampling kernels: greedy argmax and temperature / top-k / top-p sampling.
//
// One workgroup owns one row of logits.  Row maxima, argmax pairs, probability
// sums and counts are reduced with wave shuffles first and a small LDS exchange
// across the waves of the workgroup second.
#include <type_traits>
#include <math.h>

#include "common.h"

namespace mi355 {
namespace {

constexpr int kMaxWaves = 16;
constexpr int kHistShift = 19;   // histogram key = probability bit pattern >> 19: exponent + 4 mantissa bits
constexpr int kHistBins = 2048;  // keys of values in [0, 1] stay below 0x3F800000 >> 19 = 2032

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ float lane_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// Wave-wide reductions, result in every lane.  Four DPP steps fold each row of 16 lanes
// (xor 1, xor 2, half-row mirror, row mirror), four v_readlane combine the rows: no LDS,
// no ds_bpermute.  All 64 lanes must be active.
template <class Op>
__device__ __forceinline__ float wave_reduce(float v, Op op) {
  v = op(v, dpp_f<0xB1>(v));
  v = op(v, dpp_f<0x4E>(v));
  v = op(v, dpp_f<0x141>(v));
  v = op(v, dpp_f<0x140>(v));
  return op(op(lane_f(v, 0), lane_f(v, 16)), op(lane_f(v, 32), lane_f(v, 48)));
}
template <class Op>
__device__ __forceinline__ int wave_reduce(int v, Op op) {
  v = op(v, dpp_i<0xB1>(v));
  v = op(v, dpp_i<0x4E>(v));
  v = op(v, dpp_i<0x141>(v));
  v = op(v, dpp_i<0x140>(v));
  return op(op(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
            op(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// Inclusive prefix sum of a 64-bit value over the lanes of a wave (lane l: lanes 0 .. l) by DPP
// row shifts and row broadcasts: both halves are moved with the same control word (lanes without
// a source get 0) and added as one 64-bit integer, so carries are kept.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_move_u64(unsigned long long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROW_MASK, 0xf, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROW_MASK, 0xf, false);
  return ((unsigned long long)hi << 32) | lo;
}
// The same for a float (lane l: v of lanes 0 .. l, added in the fixed order of the six steps).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_prefix_sum(float v) {
  v += dpp_move_f<0x111, 0xf>(v);
  v += dpp_move_f<0x112, 0xf>(v);
  v += dpp_move_f<0x114, 0xf>(v);
  v += dpp_move_f<0x118, 0xf>(v);
  v += dpp_move_f<0x142, 0xa>(v);
  v += dpp_move_f<0x143, 0xc>(v);
  return v;
}
__device__ __forceinline__ int wave_prefix_max_int(int v) {      // inclusive prefix maximum over the lanes, v >= 0
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
  return v;
}
__device__ __forceinline__ float lane_shift_up(float v) {         // lane l gets lane l - 1's value (lane 0: its own)
  return __shfl_up(v, 1);
}
__device__ __forceinline__ unsigned long long lane_u64(unsigned long long v, int l) {   // l wave-uniform
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_prefix_sum_u64(unsigned long long v) {
  v += dpp_move_u64<0x111, 0xf>(v);   // row_shr:1
  v += dpp_move_u64<0x112, 0xf>(v);   // row_shr:2
  v += dpp_move_u64<0x114, 0xf>(v);   // row_shr:4
  v += dpp_move_u64<0x118, 0xf>(v);   // row_shr:8   -> prefix within each row of 16
  v += dpp_move_u64<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
  v += dpp_move_u64<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
  return v;
}

// Workgroup-wide reductions through a two-slot LDS exchange: call n uses slot n & 1, so one
// barrier per call is enough (a wave can be at most one call ahead of the slowest reader).
struct Exchange {
  float f[2][kMaxWaves];
  int c[2][kMaxWaves];
};

template <class T, class Op>
__device__ __forceinline__ T block_reduce(T v, Op op, Exchange& x, int& slot) {
  v = wave_reduce(v, op);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  T* buf;
  if constexpr (sizeof(T) == 4 && __is_same(T, float)) buf = reinterpret_cast<T*>(x.f[slot]);
  else buf = reinterpret_cast<T*>(x.c[slot]);
  if (lane == 0) buf[wave] = v;
  __syncthreads();
  T r = buf[0];
  for (int w = 1; w < nw; ++w) r = op(r, buf[w]);
  slot ^= 1;
  return r;
}

// ------------------------------------------------------------------ argmax
// Candidate = (value, index).  Order: NaN beats every number, larger beats
// smaller, and between equals (or two NaNs) the lower index wins -- the result
// torch.argmax returns.
struct Cand {
  float v;
  int i;
};

__device__ __forceinline__ bool cand_better(float av, int ai, float bv, int bi) {
  const bool an = av != av, bn = bv != bv;
  if (an != bn) return an;
  if (!an && av != bv) return av > bv;
  return ai < bi;
}

template <class Tag>
__global__ void __launch_bounds__(1024)
argmax_kernel(int32_t* out, const typename Tag::elem_t* logits, int vocab, int64_t row_stride, int vec_ok) {
  constexpr int EPL = Tag::kEPL;
  __shared__ float s_v[kMaxWaves];
  __shared__ int s_i[kMaxWaves];
  const typename Tag::elem_t* row = logits + (int64_t)blockIdx.x * row_stride;
  const int tid = threadIdx.x, nt = blockDim.x;
  // Per thread: running maximum over numbers (first occurrence) and first NaN seen.
  float best = -__builtin_inff();
  int best_i = 0x7fffffff, nan_i = 0x7fffffff;
  auto visit = [&](float x, int idx) {
    if (x != x) {
      nan_i = min(nan_i, idx);
    } else if (x > best || best_i == 0x7fffffff) {
      best = x;
      best_i = idx;
    }
  };
  // 16-byte pieces, kFly independent loads in flight per thread: a row of up to 32768 fp32 logits
  // (1024 threads x 8 pieces x 4 values) is requested in one go, so the row costs one trip to
  // memory instead of one per group of loads.  The index is clamped rather than branched around
  // (loads under a branch wait for their data one by one); pieces are visited in increasing index
  // order so "first occurrence" stays well defined per thread.
  // (512 threads x 16 loads and 256 x 32 measured 12 % and 2x slower: the per-element compares of
  // a row then run on fewer waves)
  const int nvec = vec_ok ? vocab / EPL : 0;
  constexpr int kFly = 8;
  for (int c0 = tid; c0 < nvec; c0 += nt * kFly) {
    u32x4 raw[kFly];
#pragma unroll
    for (int j = 0; j < kFly; ++j) {
      const int c = min(c0 + j * nt, nvec - 1);
      raw[j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(row + (int64_t)c * EPL));
    }
#pragma unroll
    for (int j = 0; j < kFly; ++j) {
      const int c = c0 + j * nt;
      if (c < nvec) {
        float f[EPL];
        Tag::unpack(raw[j], f);
#pragma unroll
        for (int e = 0; e < EPL; ++e) visit(f[e], c * EPL + e);
      }
    }
  }
  for (int idx = nvec * EPL + tid; idx < vocab; idx += nt) visit(Tag::to_float(row[idx]), idx);

  float cv = nan_i != 0x7fffffff ? __builtin_nanf("") : best;
  int ci = nan_i != 0x7fffffff ? nan_i : best_i;
  auto fold = [&](int off) {
    const float ov = __shfl_xor(cv, off);
    const int oi = __shfl_xor(ci, off);
    if (cand_better(ov, oi, cv, ci)) {
      cv = ov;
      ci = oi;
    }
  };
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) fold(off);
  const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  if (nw > 1) {
    if (lane == 0) {
      s_v[wave] = cv;
      s_i[wave] = ci;
    }
    __syncthreads();
    if (wave != 0) return;
    // the waves' candidates meet in the first lanes of wave 0 (one LDS read each, four folds)
    cv = lane < nw ? s_v[lane] : -__builtin_inff();
    ci = lane < nw ? s_i[lane] : 0x7fffffff;
#pragma unroll
    for (int off = kMaxWaves / 2; off >= 1; off >>= 1) fold(off);
  }
  if (tid == 0) out[blockIdx.x] = ci == 0x7fffffff ? 0 : ci;
}

// ------------------------------------------------------------------ philox
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (uint32_t)p1;
  c[3] = (uint32_t)p0;
  c[0] = n0;
  c[2] = n2;
}

// Philox4x32-10: uniform in [0,1) from (seed, counter).
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t counter) {
  uint32_t c[4] = {(uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return (float)(c[0] >> 8) * (1.0f / 16777216.0f);
}

// ------------------------------------------------------------------ sampling
struct SampleParams {
  int32_t* out;
  const void* logits;
  const float* temperature;
  const float* top_p;
  const int32_t* top_k;
  float* out_probs;
  uint64_t seed, offset;
  int vocab;
  int64_t row_stride;
};

struct FC {  // (probability mass, count) pair reduced together
  float f;
  int c;
};

// SAMPLER_TIMING (side builds only): thread 0 of every workgroup writes the shader clock at the
// phase boundaries of sample_kernel into out_probs (as 64-bit words, 32 per row) instead of
// probabilities.
#ifdef SAMPLER_TIMING
#define STAMP(n)                                                                                     \
  do {                                                                                               \
    if (threadIdx.x == 0 && p.out_probs)                                                             \
      reinterpret_cast<unsigned long long*>(p.out_probs)[(size_t)blockIdx.x * 32 + (n)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define STAMP(n) do { } while (0)
#endif
#ifndef SAMPLER_LEAN_MATH
#define SAMPLER_LEAN_MATH 1
#endif

// a / b for a whole row that shares b, given y = 1.0f / b (one correctly rounded division per
// row): the product a * y is within an ulp of the quotient, and one residual step
// q + (a - b q) y, with the residual exact in an FMA, lands on the correctly rounded quotient --
// the sequence the compiler emits for `/` without its scaling and fix-up steps, which only
// matter for denormal or near-overflow operands.  Infinite a passes through (the residual
// would be inf - inf).
__device__ __forceinline__ float div_shared(float a, float b, float y) {
#if SAMPLER_LEAN_MATH
  const float q = a * y;
  const float r = __builtin_fmaf(-b, q, a);
  const float c = __builtin_fmaf(r, y, q);
  return __builtin_fabsf(q) < __builtin_inff() ? c : q;
#else
  (void)y;
  return a / b;
#endif
}

// exp(d) for d <= 0 (softmax numerators).  d * log2(e) is formed as a rounded product plus its
// rounding error (FMA) plus the low word of log2(e); the hardware exp2 takes the rounded part and
// the rest is applied as a first-order factor: about an ulp, like the library routine, in six
// instructions.  Results below the smallest normal float (d < -87.3, -inf included) are 0.
__device__ __forceinline__ float exp_nonpos(float d) {
#if SAMPLER_LEAN_MATH
  constexpr float kL2eHi = 1.44269502162933349609375f, kL2eLo = 1.925963033500011e-8f, kLn2 = 0.693147182464599609375f;
  const float t = d * kL2eHi;
  float lo = __builtin_fmaf(d, kL2eHi, -t);
  lo = __builtin_fmaf(d, kL2eLo, lo);
  const float e = __builtin_amdgcn_exp2f(t);
  const float r = __builtin_fmaf(e, lo * kLn2, e);
  return d > -87.3f ? r : 0.f;
#else
  return expf(d);
#endif
}

// A probability as the cut search tallies it: count 1 in the bits from 41 up, the value in units of
// 2^-40 below them.  Sums of such words are exact and do not depend on the order of addition.
__device__ __forceinline__ unsigned long long packed_prob(float q) {
  return (1ull << 41) + (unsigned long long)(q * 1099511627776.0f);
}
// mass field (41 bits, units of 2^-40) -> float, correctly rounded: the upper 24 bits and the
// lower 17 are each exact as floats and their sum is rounded once -- what the generic 64-bit
// conversion returns, in a third of its instructions.
__device__ __forceinline__ float packed_mass(unsigned long long packed) {
  const unsigned long long m = packed & ((1ull << 41) - 1);
  return __builtin_fmaf((float)(uint32_t)(m >> 17), 131072.0f, (float)((uint32_t)m & 0x1FFFFu)) * 0x1p-40f;
}

// VPT > 0: the row's probabilities live in registers (VPT per thread).
// VPT == 0: they are recomputed from the logits on every pass (any vocab size).
// VEC: the row is read in 16-byte pieces (row base and stride 16-byte aligned, vocab a multiple of
// the piece): a thread owns the elements of pieces tid, tid + nt, ... instead of single elements.
template <class Tag, int VPT, int NT, bool VEC>
__global__ void __launch_bounds__(NT) sample_kernel(const SampleParams p) {
  constexpr int EPL = Tag::kEPL;
  static_assert(!VEC || VPT % EPL == 0, "register rows hold whole pieces");
  __shared__ Exchange xch;
  __shared__ float s_scan[kMaxWaves];
  __shared__ int s_pick;
  // cut search state (see the filter section)
  __shared__ unsigned long long s_hist[kHistBins];
  __shared__ unsigned long long s_above;
  __shared__ unsigned long long s_wsum[kMaxWaves];
  __shared__ int s_cutbin, s_cutcount;
  __shared__ uint32_t s_cutbits;
  __shared__ float s_wbound[kMaxWaves];
  constexpr int kListCap = 4096;
  __shared__ float s_list[kListCap];
  __shared__ int s_list_n;
  // streamed rows under a top-k limit: the few elements at or above the bound (x / T and index), as
  // they arrive and ordered by index
  constexpr int kCandCap = VPT == 0 ? NT : 1;
  __shared__ float s_cand_x[kCandCap], s_cand_sx[kCandCap];
  __shared__ int s_cand_i[kCandCap], s_cand_si[kCandCap];
  __shared__ int s_cand_n;
  // register-resident rows under a top-k limit: the candidates at or above a lower bound of the
  // k-th largest weight (weight and index as collected, then probability), and the k largest of
  // them in descending order (probability, index, rank of the first copy of the value, and what
  // lies above each rank as a packed count | mass)
  constexpr int kTopCap = VPT == 0 ? 4 : (NT < 512 ? NT : 512);
  constexpr int kTopK = 128;
  __shared__ __attribute__((aligned(16))) unsigned long long s_top_key[kTopCap + 2];
  __shared__ int s_top_rank[kTopCap];
  __shared__ float s_top_bound[kMaxWaves];
  __shared__ unsigned long long s_srt_key[kTopK];
  __shared__ unsigned long long s_srt_above[kTopK + 1];
  constexpr int kTieCap = 64;
  __shared__ int s_tie_i[kTieCap];
  __shared__ int s_tie_n;
  __shared__ float s_walk[VPT > 0 ? VPT : 1];   // the picking thread's elements, handed to wave 0
  __shared__ float s_walk_base;
  __shared__ float s_uniform;
  int slot = 0;
  constexpr int nt = NT;              // (every launch of this kernel has NT threads: indices are tid + constants)
  const int tid = threadIdx.x, b = blockIdx.x;
  STAMP(0);
  const int lane = tid & 63, wave = tid >> 6;
  // The row's one random number, by one thread while the row's first loads are in the air (its ten
  // rounds of 64-bit products otherwise sit in every thread's way right before the draw); read back
  // behind the barriers of the reductions below.
  auto draw_uniform = [&]() {
    if (tid == 0) s_uniform = philox_uniform(p.seed, p.offset + (uint64_t)b);
  };
  // Wave 0 finds the first of the picker's elements (s_walk, running sum starting at s_walk_base)
  // whose running sum passes the target, with a prefix scan, and writes its index.
  auto walk_pick = [&](int picker, float target) {
    const float q = lane < VPT ? s_walk[lane < VPT ? lane : 0] : 0.f;
    const float run = wave_prefix_sum(q) + s_walk_base;
    const unsigned long long hit = __ballot(q > 0.f && run > target);
    const unsigned long long any = __ballot(q > 0.f);          // not empty: the picker's mass is > 0
    const int j = hit ? __ffsll((long long)hit) - 1 : 63 - __clzll((long long)any);
    if (lane == 0) p.out[b] = VEC ? ((j / EPL) * nt + picker) * EPL + (j % EPL) : j * nt + picker;
  };
  const typename Tag::elem_t* row = static_cast<const typename Tag::elem_t*>(p.logits) + (int64_t)b * p.row_stride;
  const int vocab = p.vocab;
  const float temp = p.temperature[b];
  const float tp = p.top_p ? p.top_p[b] : 1.0f;
  const int tk = p.top_k ? p.top_k[b] : 0;
  auto fmaxop = [](float a, float c) { return fmaxf(a, c); };
  auto addf = [](float a, float c) { return a + c; };
  auto addi = [](int a, int c) { return a + c; };
  auto maxi = [](int a, int c) { return max(a, c); };

  // p = softmax(x / T): the division comes first, as in the reference.  Register-resident
  // rows are divided and exponentiated exactly once.
  constexpr int NR = VPT > 0 ? VPT : 1;
  float pr[NR];
  float mx = -__builtin_inff(), z = 0.f;
  float thread_top = -__builtin_inff();   // streamed rows: the largest x / T among this thread's elements
  const float inv_t = 1.0f / temp;
  const bool unit_temp = temp == 1.0f;   // x / 1 is x, exactly
  // a row without a cut whose distribution nobody asked for only needs its draw (the short paths below)
  const bool draw_only = !(tp < 1.0f || tk > 0) && !p.out_probs;
  auto by_temp = [&](float x) { return unit_temp ? x : draw_only ? x * inv_t : div_shared(x, temp, inv_t); };
  // Rows that do not fit the registers are streamed from memory on every pass: fn(logit, index)
  // for the thread's elements tid, tid + nt, ... in that order, kStreamU loads in flight per thread
  // (the index is clamped instead of branching around the load, so that the loads of a group
  // issue back to back; a load under a branch waits for its data before the next one starts).
  constexpr int kStreamU = VEC ? 4 : 8;
  auto stream_row = [&](auto fn) {
    if constexpr (VEC) {
      const int nvec = vocab / EPL;
      for (int c0 = tid; c0 < nvec; c0 += nt * kStreamU) {
        u32x4 raw[kStreamU];
#pragma unroll
        for (int u = 0; u < kStreamU; ++u)
          raw[u] = *reinterpret_cast<const u32x4*>(row + (int64_t)min(c0 + u * nt, nvec - 1) * EPL);
#pragma unroll
        for (int u = 0; u < kStreamU; ++u)
          if (c0 + u * nt < nvec) {
            float f[EPL];
            Tag::unpack(raw[u], f);
#pragma unroll
            for (int e = 0; e < EPL; ++e) fn(f[e], (c0 + u * nt) * EPL + e);
          }
      }
    } else {
      for (int i0 = tid; i0 < vocab; i0 += nt * kStreamU) {
        float x[kStreamU];
#pragma unroll
        for (int u = 0; u < kStreamU; ++u) x[u] = Tag::to_float(row[min(i0 + u * nt, vocab - 1)]);
#pragma unroll
        for (int u = 0; u < kStreamU; ++u)
          if (i0 + u * nt < vocab) fn(x[u], i0 + u * nt);
      }
    }
  };
  // index of the thread's j-th register-resident element (ascending in j)
  auto eidx = [&](int j) {
    if constexpr (VEC) return ((j / EPL) * nt + tid) * EPL + (j % EPL);
    else return j * nt + tid;
  };
  if constexpr (VPT > 0) {
    // all of the thread's loads first (clamped index, no branch), then the arithmetic
    if constexpr (VEC) {
      const int nvec = vocab / EPL;
      u32x4 raw[VPT / EPL];
#pragma unroll
      for (int k = 0; k < VPT / EPL; ++k)
        raw[k] = __builtin_nontemporal_load(
            reinterpret_cast<const u32x4*>(row + (int64_t)min(k * nt + tid, nvec - 1) * EPL));
      draw_uniform();
#pragma unroll
      for (int k = 0; k < VPT / EPL; ++k) {
        float f[EPL];
        Tag::unpack(raw[k], f);
#pragma unroll
        for (int e = 0; e < EPL; ++e) pr[k * EPL + e] = f[e];
      }
    } else {
#pragma unroll
      for (int j = 0; j < VPT; ++j) pr[j] = Tag::to_float(row[min(j * nt + tid, vocab - 1)]);
      draw_uniform();
    }
    if (!unit_temp) {   // (wave-uniform: a row at temperature 1 skips 32 divisions per thread)
      if (draw_only) {  // no cut to pin: the product with 1 / T is within an ulp of the quotient
#pragma unroll
        for (int j = 0; j < VPT; ++j) pr[j] *= inv_t;
      } else {
#pragma unroll
        for (int j = 0; j < VPT; ++j) pr[j] = div_shared(pr[j], temp, inv_t);
      }
    }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      pr[j] = (eidx(j) < vocab) ? pr[j] : -__builtin_inff();
      mx = fmaxf(mx, pr[j]);
    }
    // Under a top-k limit of up to 128 the cut, the tie rule, the kept mass and the draw are taken
    // from a short list of candidates (see the list section below).  Its lower bound is known here
    // already, in x / T: every wave takes the k'-th largest of its lanes' largest values, k' =
    // ceil(k / waves) -- at least k' different elements are >= it, so the smallest of the waves'
    // values has at least k elements at or above it.  (The reduction below publishes the bounds.)
    const int list_per_wave = tk > 0 ? (tk + (nt >> 6) - 1) / (nt >> 6) : 0;
    const bool list_try = list_per_wave >= 1 && list_per_wave <= 8;   // (the same for the whole row)
    if (list_try) {
      float top = mx, kth = -__builtin_inff();
      for (int r = 0; r < list_per_wave; ++r) {
        kth = wave_reduce(top, fmaxop);
        const unsigned long long holders = __ballot(top == kth);
        if (lane == __ffsll((long long)holders) - 1) top = -__builtin_inff();   // one lane gives its value up per round
      }
      if (lane == 0) s_top_bound[wave] = kth;
      if (tid == 0) s_cand_n = 0;
      if (tid < kTopCap) {
        s_top_rank[tid] = 0;
        s_top_key[tid] = 0ull;                             // (entries past the list: behind every candidate)
      }
    }
    mx = block_reduce(mx, fmaxop, xch, slot);
    STAMP(1);
    // A row without a cut whose distribution nobody asked for only needs its draw: weights
    // e = exp2((x/T - max) log2 e) (a plain exp2: the extra digits of exp_nonpos pin the kept set of
    // a cut, here they would only move a draw by an ulp of the target), no division by their sum --
    // the target is scaled instead -- and the sum itself falls out of the scan over threads.
    if (draw_only) {
      float mine = 0.f;
#pragma unroll
      for (int j = 0; j < VPT; ++j) {
        pr[j] = __builtin_amdgcn_exp2f((pr[j] - mx) * 1.44269504088896340736f);   // absent elements: 0
        mine += pr[j];
      }
      const float incl = wave_prefix_sum(mine);
      if (lane == 63) s_scan[wave] = incl;
      if (tid == 0) s_pick = 0x7fffffff;
      __syncthreads();
      float base = 0.f, total = 0.f;
      const int nw = nt >> 6;
#pragma unroll
      for (int w = 0; w < kMaxWaves; ++w) {
        const float v = w < nw ? s_scan[w] : 0.f;
        total += v;
        base += w < wave ? v : 0.f;
      }
      const float target = s_uniform * total;
      const float excl = base + incl - mine;
      if (mine > 0.f && excl <= target && target < excl + mine) atomicMin(&s_pick, tid);
      __syncthreads();
      const int picker = s_pick;
      if (picker == 0x7fffffff) {      // rounding pushed the target past the walked total: the last weighted element
        int last = -1;
#pragma unroll
        for (int j = 0; j < VPT; ++j) last = (pr[j] > 0.f && eidx(j) < vocab) ? max(last, eidx(j)) : last;
        last = block_reduce(last, maxi, xch, slot);
        if (tid == 0) p.out[b] = max(last, 0);
        return;
      }
      if (tid == picker) {
#pragma unroll
        for (int j = 0; j < VPT; ++j) s_walk[j] = eidx(j) < vocab ? pr[j] : 0.f;
        s_walk_base = excl;
      }
      __syncthreads();
      if (wave == 0) walk_pick(picker, target);
      return;
    }
    // The candidates of the list path are picked up in the pass that exponentiates: x / T at or above
    // the bound -- a hundred or two of 32000 -- goes to LDS as one 64-bit key, weight bits above the
    // complemented index; most rounds of a wave find none and cost a comparison and a scalar branch.
    // A bound whose weight underflows to 0 (fewer than k positive weights) means no list.
    float lowest_x = __builtin_inff();                   // nothing is a candidate
    if (list_try) {
      float lo = s_top_bound[0];
#pragma unroll
      for (int w = 1; w < kMaxWaves; ++w) lo = fminf(lo, s_top_bound[w < (nt >> 6) ? w : 0]);
      if (exp_nonpos(lo - mx) > 0.f) lowest_x = lo;      // (-inf: a wave without k' values)
    }
    const bool list_on = lowest_x < __builtin_inff();
    if (list_on) {                                       // (wave-uniform; rows without a list keep the bare loop)
#pragma unroll
      for (int j = 0; j < VPT; ++j) {
        const float x = pr[j];
        const float e = exp_nonpos(x - mx);              // absent elements: exp(-inf) = 0
        pr[j] = e;
        z += e;
        const bool hit = x >= lowest_x;
        const unsigned long long hits = __ballot(hit);
        if (hits) {                                      // (wave-uniform)
          int at = 0;
          if (lane == 0) at = atomicAdd(&s_cand_n, __popcll(hits));
          at = __builtin_amdgcn_readfirstlane(at) +
               (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(hits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hits, 0u));
          // (the element's index is formed here, from a thread id the compiler cannot match with the
          // one the load pass used: shared with that pass, all 32 indices stay live across the loop
          // and push the kernel past its 128 registers)
          int t = tid;
          asm volatile("" : "+v"(t));
          const int idx = VEC ? ((j / EPL) * nt + t) * EPL + (j % EPL) : j * nt + t;
          if (hit && at < kTopCap)
            s_top_key[at] = ((unsigned long long)__builtin_bit_cast(uint32_t, e) << 32) | (uint32_t)~idx;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < VPT; ++j) {
        pr[j] = exp_nonpos(pr[j] - mx);                  // absent elements: exp(-inf) = 0
        z += pr[j];
      }
    }
    z = block_reduce(z, addf, xch, slot);
    const float inv_z = 1.0f / z;
    STAMP(2);
    // The list: only the candidates are normalised (by the expression every other path uses, so
    // probabilities are the same bits), each finds its place in the descending order (ties: lower
    // index first, the stable sort of the reference, 05_mini_vllm/mini_vllm.py:620) by counting the
    // candidates ahead of it, and the first k in that order are everything the cut and the draw
    // need: no histogram over the row, no normalisation of the row, no pass for the kept mass.
    // What lies above a value is summed as the histogram bins would sum it (packed_prob: exact,
    // order-free), so the kept set is the one the bin search finds.  A larger key comes first in
    // the order (probabilities are positive floats, whose bit patterns order as the values do; of
    // two equal ones the lower index wins).  The list is trusted only while the k-th value in it
    // lies strictly above the probability of the bound: everything outside the list is at or below
    // that, so nothing outside can tie with a kept value and claim its place by a lower index.  A
    // row whose candidates do not fit (long constant stretches), or whose k-th value meets the
    // bound, takes the general path below.
    {
      if (list_on) {
        const int n = s_cand_n;
        STAMP(24);
        if (n <= kTopCap) {
          if (tid < n) {
            const unsigned long long key = s_top_key[tid];
            const float q = div_shared(__builtin_bit_cast(float, (uint32_t)(key >> 32)), z, inv_z);
            s_top_key[tid] = ((unsigned long long)__builtin_bit_cast(uint32_t, q) << 32) | (uint32_t)key;
          }
          __syncthreads();
          STAMP(25);
          // Place of every candidate in the descending order = the number of larger keys.  The
          // threads of the workgroup share the comparisons: thread = (candidate, slice of the list).
          const int n2 = (n + 1) & ~1;
          const int parts = nt / n;                      // >= 1: 1 <= n <= kTopCap <= nt
          const int part = tid / n, c = tid - part * n;
          const int per_part = (((n2 >> 1) + parts - 1) / parts) << 1;
          if (part < parts) {
            const unsigned long long key = s_top_key[c];
            const int m0 = part * per_part, m1 = min(m0 + per_part, n2);
            int ahead = 0;
            for (int m = m0; m < m1; m += 2) {           // two candidates per read
              const ulonglong2 km = *reinterpret_cast<const ulonglong2*>(&s_top_key[m]);
              ahead += (km.x > key) + (km.y > key);
            }
            if (ahead) atomicAdd(&s_top_rank[c], ahead);
          }
#ifndef SAMPLER_TIMING
          if (p.out_probs) {                             // every entry is 0 except the kept ones, written below
            float* dst = p.out_probs + (int64_t)b * vocab;
#pragma unroll
            for (int j = 0; j < VPT; ++j)
              if (eidx(j) < vocab) dst[eidx(j)] = 0.f;
          }
#endif
          __syncthreads();
          if (tid < n) {
            const int rank = s_top_rank[tid];
            if (rank < kTopK) s_srt_key[rank] = s_top_key[tid];
          }
          __syncthreads();
          STAMP(26);
          const int nk = min(min(tk, n), kTopK);
          const float q_kth = __builtin_bit_cast(float, (uint32_t)(s_srt_key[nk - 1] >> 32));
          if (q_kth > div_shared(exp_nonpos(lowest_x - mx), z, inv_z)) {   // (wave-uniform: see above)
          if (wave != 0) return;
          // Wave 0 alone takes it from here, two ranks per lane (rank = lane, lane + 64), the first
          // nk = min(k, n) of the order.  first = rank of the first copy of the entry's value (equal
          // values are neighbours in the order); above[r] = packed (count | mass) of ranks 0 .. r - 1.
          const bool live0 = lane < nk, live1 = lane + kWave < nk;
          const unsigned long long k0 = live0 ? s_srt_key[lane] : 0ull, k1 = live1 ? s_srt_key[lane + kWave] : 0ull;
          const float q0 = __builtin_bit_cast(float, (uint32_t)(k0 >> 32)), q1 = __builtin_bit_cast(float, (uint32_t)(k1 >> 32));
          const int i0 = (int)~(uint32_t)k0, i1 = (int)~(uint32_t)k1;
          // a value starts where it differs from the entry before it; the start's rank is carried forward
          const float before0 = lane > 0 ? lane_shift_up(q0) : -1.0f;
          const float before1 = lane > 0 ? lane_shift_up(q1) : lane_f(q0, kWave - 1);
          const int f0 = wave_prefix_max_int(q0 != before0 ? lane : 0);
          const int f1 = max(wave_prefix_max_int(q1 != before1 ? lane + kWave : 0), __builtin_amdgcn_readlane(f0, kWave - 1));
          const unsigned long long in0 = wave_prefix_sum_u64(live0 ? packed_prob(q0) : 0ull);
          const unsigned long long in1 = wave_prefix_sum_u64(live1 ? packed_prob(q1) : 0ull) + lane_u64(in0, kWave - 1);
          if (lane == 0) s_srt_above[0] = 0ull;
          s_srt_above[lane + 1] = in0;
          s_srt_above[lane + 1 + kWave] = in1;
          __builtin_amdgcn_wave_barrier();               // (written and read by this wave only)
          // copy j of a value is kept iff (mass above the value) + j * value < top_p and its rank < k;
          // the top token always stays (05_mini_vllm/mini_vllm.py:624)
          const bool use_p = tp < 1.0f;
          const float a0 = packed_mass(s_srt_above[f0]), a1 = packed_mass(s_srt_above[f1]);
          const bool kept0 = live0 && (lane == 0 || !use_p || a0 + (float)(lane - f0) * q0 < tp);
          const bool kept1 = live1 && (!use_p || a1 + (float)(lane + kWave - f1) * q1 < tp);
          // kept entries are a prefix of the order: its length, and its mass as the general path
          // forms it (mass above the last kept value + its kept copies times the value)
          const int kept_n = __popcll(__ballot(kept0)) + __popcll(__ballot(kept1));   // >= 1
          const int last = kept_n - 1;
          const float m0v = a0 + (float)(lane - f0 + 1) * q0, m1v = a1 + (float)(lane + kWave - f1 + 1) * q1;
          const float kept_mass_l = last < kWave ? lane_f(m0v, last) : lane_f(m1v, last - kWave);
          const float target = s_uniform * kept_mass_l;
          const float run0 = wave_prefix_sum(lane < kept_n ? q0 : 0.f);
          const float run1 = wave_prefix_sum(lane + kWave < kept_n ? q1 : 0.f) + lane_f(run0, kWave - 1);
          const unsigned long long hit0 = __ballot(lane < kept_n && run0 > target);
          const unsigned long long hit1 = __ballot(lane + kWave < kept_n && run1 > target);
          // (rounding may push the target past the total: the last kept entry takes it)
          const int pick = hit0 ? __ffsll((long long)hit0) - 1 : hit1 ? kWave + __ffsll((long long)hit1) - 1 : last;
          const int pick_i = pick < kWave ? __builtin_amdgcn_readlane(i0, pick) : __builtin_amdgcn_readlane(i1, pick - kWave);
          if (lane == 0) p.out[b] = pick_i;
          STAMP(27);
#ifndef SAMPLER_TIMING
          if (p.out_probs) {
            float* dst = p.out_probs + (int64_t)b * vocab;
            if (lane < kept_n) dst[i0] = q0 / kept_mass_l;
            if (lane + kWave < kept_n) dst[i1] = q1 / kept_mass_l;
          }
#endif
          return;
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < VPT; ++j) pr[j] = (eidx(j) < vocab) ? div_shared(pr[j], z, inv_z) : -1.0f;   // -1 marks "no element"
  } else {
    draw_uniform();
    // A streamed row without a cut whose distribution nobody asked for: ONE trip through memory.
    // Every thread keeps a running maximum and the sum of its weights exp2((x/T - maximum) log2 e)
    // relative to it (rescaled when a group of loads raises the maximum: one extra exponential per
    // group, not per element); the threads' sums, brought to the row's maximum, are scanned for the
    // thread the target falls to, and wave 0 alone re-reads that thread's hundred-odd elements to
    // find the token.  (A row with a cut shares the first trip -- online_max_sum below -- and then
    // makes further ones for the cut search and the kept mass.)
    constexpr float kL2e = 1.44269504088896340736f;
    constexpr int GE = VEC ? EPL : 1;                  // elements per load
    constexpr int kU = VEC ? 32 / EPL : 8;             // loads in flight per thread: 32 elements (8) per group
    const int nload = VEC ? vocab / EPL : vocab;       // loads in the row
    // One trip through the row: the thread's running maximum of x / T and the sum of its weights relative
    // to it.  `once`: the row is not read again (loads that bypass the caches, weights by the bare
    // exp2 instruction); otherwise the weights are formed as the probabilities will be.
    auto online_max_sum = [&](auto once, float& m_t, float& z_t) {
      constexpr bool kOnce = decltype(once)::value;
      auto load_piece = [&](const u32x4* at) {
        if constexpr (kOnce) return __builtin_nontemporal_load(at); else return *at;
      };
      auto weight = [&](float d) {
        if constexpr (kOnce) return __builtin_amdgcn_exp2f(d * kL2e); else return exp_nonpos(d);
      };
      m_t = -__builtin_inff();
      z_t = 0.f;
      for (int c0 = tid; c0 < nload; c0 += nt * kU) {
        float xg[kU * GE];
        if constexpr (VEC) {
          u32x4 raw[kU];
#pragma unroll
          for (int u = 0; u < kU; ++u)
            raw[u] = load_piece(reinterpret_cast<const u32x4*>(row + (int64_t)min(c0 + u * nt, nload - 1) * EPL));
#pragma unroll
          for (int u = 0; u < kU; ++u) {
            float f[EPL];
            Tag::unpack(raw[u], f);
#pragma unroll
            for (int e = 0; e < EPL; ++e) xg[u * GE + e] = c0 + u * nt < nload ? by_temp(f[e]) : -__builtin_inff();
          }
        } else {
          float x[kU];
#pragma unroll
          for (int u = 0; u < kU; ++u) x[u] = Tag::to_float(row[min(c0 + u * nt, nload - 1)]);
#pragma unroll
          for (int u = 0; u < kU; ++u) xg[u] = c0 + u * nt < nload ? by_temp(x[u]) : -__builtin_inff();
        }
        float gmax = xg[0];
#pragma unroll
        for (int i = 1; i < kU * GE; ++i) gmax = fmaxf(gmax, xg[i]);   // (a NaN logit never wins)
        if (gmax > m_t) {
          z_t *= weight(m_t - gmax);                                         // first group: 0 * 0
          m_t = gmax;
        }
        const float ref = m_t > -__builtin_inff() ? m_t : 0.f;               // (nothing finite yet: every weight 0)
#pragma unroll
        for (int i = 0; i < kU * GE; ++i)
          z_t += fmaxf(weight(xg[i] - ref), 0.f);                            // NaN logit: weight 0
      }
    };
    if (draw_only) {
      float m_t, z_t;
      online_max_sum(std::true_type{}, m_t, z_t);
      mx = block_reduce(m_t, fmaxop, xch, slot);
      const float row_ref = mx > -__builtin_inff() ? mx : 0.f;
      const float mine = m_t > -__builtin_inff() ? z_t * __builtin_amdgcn_exp2f((m_t - row_ref) * kL2e) : 0.f;
      const float incl = wave_prefix_sum(mine);
      if (lane == 63) s_scan[wave] = incl;
      if (tid == 0) s_pick = 0x7fffffff;
      __syncthreads();
      float base = 0.f, total = 0.f;
#pragma unroll
      for (int w = 0; w < kMaxWaves; ++w) {
        const float v = w < (nt >> 6) ? s_scan[w] : 0.f;
        total += v;
        base += w < wave ? v : 0.f;
      }
      const float target = s_uniform * total;
      const float excl = base + incl - mine;
      if (mine > 0.f && excl <= target && target < excl + mine) atomicMin(&s_pick, tid);
      __syncthreads();
      int picker = s_pick;
      if (picker == 0x7fffffff) {      // rounding pushed the target past the total: the last thread with any weight
        picker = block_reduce(mine > 0.f ? tid : -1, maxi, xch, slot);
        if (picker < 0) {              // no finite logit in the row
          if (tid == 0) p.out[b] = 0;
          return;
        }
      }
      if (tid == picker) s_walk_base = excl;
      __syncthreads();
      if (wave != 0) return;
      // the picker's elements, in its own order, 64 at a time
      const float tgt = target - s_walk_base;
      const int owned = picker < nload ? ((nload - 1 - picker) / nt + 1) * GE : 0;
      float carry = 0.f;
      int last_weighted = -1;
      for (int j0 = 0; j0 < owned; j0 += kWave) {
        const int j = j0 + lane;
        const int idx = ((j / GE) * nt + picker) * GE + (j % GE);
        const float x = j < owned ? by_temp(Tag::to_float(row[j < owned ? idx : 0])) : -__builtin_inff();
        const float w = fmaxf(__builtin_amdgcn_exp2f((x - row_ref) * kL2e), 0.f);
        const float run = wave_prefix_sum(w) + carry;
        const unsigned long long weighted = __ballot(w > 0.f);
        const unsigned long long hit = __ballot(w > 0.f && run > tgt);
        if (hit) {
          const int l = __ffsll((long long)hit) - 1;
          if (lane == l) p.out[b] = idx;
          return;
        }
        if (weighted) last_weighted = __builtin_amdgcn_readlane(idx, 63 - __clzll((long long)weighted));
        carry = lane_f(run, 63);
      }
      if (lane == 0) p.out[b] = max(last_weighted, 0);   // rounding again: the picker's last weighted element
      return;
    }
    // with a cut: maximum and sum in one trip as well, the threads' sums brought to the row's maximum
    float m_t, z_t;
    online_max_sum(std::false_type{}, m_t, z_t);
    thread_top = m_t;
    mx = block_reduce(m_t, fmaxop, xch, slot);
    z = block_reduce(m_t > -__builtin_inff() ? z_t * exp_nonpos(m_t - mx) : 0.f, addf, xch, slot);
  }
  const float inv_z_row = 1.0f / z;
  auto prob_of = [&](float x) { return div_shared(exp_nonpos(by_temp(x) - mx), z, inv_z_row); };
  // Visits every element this thread owns: fn(probability, index).
  // Streamed rows under a top-k limit: once a lower bound of the k-th largest value is known (below),
  // elements whose x / T lies under x_floor are passed over without computing their probability --
  // two divisions and an exponential each, which is what a pass over such a row spends its time on.
  // When no more of them than threads pass that test (the usual case: a few hundred of 128256), one
  // trip through the row collects them in LDS, ordered by index so that the draw does not depend on
  // the order of arrival, and every later visit is one element per thread from that list instead of
  // another trip through memory.
  bool floor_on = false, cand_on = false;
  float x_floor = 0.f;
  int cand_n = 0;
  auto for_each = [&](auto fn) {
    if constexpr (VPT > 0) {
#pragma unroll
      for (int j = 0; j < VPT; ++j)
        if (pr[j] >= 0.f) fn(pr[j], eidx(j));
    } else if (cand_on) {
      if (tid < cand_n) fn(div_shared(exp_nonpos(s_cand_sx[tid] - mx), z, inv_z_row), s_cand_si[tid]);
    } else {
      stream_row([&](float x, int i) {
        const float xs = by_temp(x);
        if (!floor_on || xs >= x_floor) fn(div_shared(exp_nonpos(xs - mx), z, inv_z_row), i);
      });
    }
  };

  // Cut: elements with probability > vstar are kept, and of those equal to vstar
  // the ones with index < tie_end.  Default keeps everything.
  float vstar = -0.5f, kept_mass;   // (below every probability, above the -1 that marks "no element")
  int tie_end = vocab;
  const bool filter = tp < 1.0f || tk > 0;
  if (!filter) {
    // nothing is cut: the distribution is the softmax itself, as in the reference, which renormalises
    // only after a cut (05_mini_vllm/mini_vllm.py:619-626).  The walked total can differ from 1 by
    // rounding; a target beyond it falls to the last element (see the draw).
    kept_mass = 1.0f;
  } else {
    const int kk = tk > 0 ? tk : 0x7fffffff;
    // A value v survives (its first copy in descending order is kept) iff the mass strictly
    // above it is < top_p and fewer than k elements are above it: ok(mass, count).  That
    // predicate is monotone in v, so the smallest surviving bit pattern is found by a radix
    // search over the probability's bit pattern, most significant field first: 12 bits
    // (exponent + 4 mantissa bits), then 11, then 8.  Each level is one pass of the whole
    // workgroup over the row -- one LDS atomic per element that shares the prefix fixed so far,
    // carrying its count and its mass in 2^-40 fixed point in one 64-bit add, so bin sums are
    // exact and independent of the order of arrival -- and one scan of the bins by wave 0 for the
    // lowest bin above which everything still passes ok().  After the last level the bin IS the
    // smallest kept value, with the exact (mass, count) above it and the number of its copies.
    // top_p >= 1 switches the mass test off altogether (05_mini_vllm/mini_vllm.py:619 applies the
    // nucleus cut only `if top_p < 1.0`): with a peaked row the mass above the small entries
    // rounds to 1.0f, and `mass < 1.0f` would drop them from a row that only has a top-k limit.
    const bool use_p = tp < 1.0f;
    auto ok = [&](float mass, int count) { return (!use_p || mass < tp) && count < kk; };
    auto mass_of = [&](unsigned long long packed) { return packed_mass(packed); };
    constexpr int kLevelShift[3] = {kHistShift, 8, 0};
    constexpr int kLevelBits[3] = {11, 11, 8};        // level 0: bit 31 is the sign (0), keys < kHistBins
    static_assert(kHistShift == 19 && kHistBins == 2048, "the three fields must tile the 32 bits");
    unsigned long long acc = 0ull;     // packed (count | mass) of everything above the prefix's range
    uint32_t prefix = 0;               // the bits fixed so far
    // After the first level only the elements of its cut bin matter.  When they fit (the usual
    // case: a few hundred of 32000) they are copied to an LDS list once, and the other two levels
    // walk the list -- one element per thread -- instead of the whole row again (for a
    // register-resident row that is 32 compares per thread per level, for a streamed row another
    // trip through memory).
    int listed = -1;                   // >= 0: the candidates are s_list[0 .. listed)
    // Under a top-k limit nothing below the k-th largest probability matters: not for the cut, not
    // for what lies above it.  A lower bound on that value lets the passes below skip most of the
    // row, and the first level's 32 LDS atomics per thread are its largest single cost.  Every
    // wave takes the k'-th largest of its lanes' largest values, k' = ceil(k / waves): at least k'
    // different elements of the row are >= it, so the smallest of the waves' values has at least k
    // elements at or above it and cannot exceed the k-th largest.  Bins below the bound's then stay
    // empty, but everything tallied lies above them -- at least k elements -- so they fail the count
    // test as they would with the whole row in; from the bound's bin upwards nothing changes.
    float tally_from = 0.f;
    bool bounded = false;
    if constexpr (VPT > 0) {
      const int nw = nt >> 6;
      const int per_wave = tk > 0 ? (tk + nw - 1) / nw : 0;
      bounded = per_wave >= 1 && per_wave <= 8;          // (wave-uniform; larger k: the whole row is tallied)
      if (bounded) {
        float top = -1.0f;
#pragma unroll
        for (int j = 0; j < VPT; ++j) top = fmaxf(top, pr[j]);
        float kth = -1.0f;
        for (int r = 0; r < per_wave; ++r) {
          kth = wave_reduce(top, fmaxop);
          const unsigned long long holders = __ballot(top == kth);
          if (lane == __ffsll((long long)holders) - 1) top = -2.0f;   // one lane gives its value up per round
        }
        if (lane == 0) s_wbound[wave] = kth;             // (read behind the first level's first barrier)
      }
    } else {
      // Streamed row: the same bound from the threads' largest x / T (kept from the first pass).
      // Its probability, lowered by a few units in the last place, is the floor of the tallies --
      // the computed probability need not be monotone in x / T to the last bit -- and elements more
      // than 0.01 below it in x / T (a probability ratio of 0.99) are not even evaluated.
      const int nw = nt >> 6;
      const int per_wave = tk > 0 ? (tk + nw - 1) / nw : 0;
      bounded = per_wave >= 1 && per_wave <= 8;
      if (bounded) {
        float top = thread_top;
        float kth = -__builtin_inff();
        for (int r = 0; r < per_wave; ++r) {
          kth = wave_reduce(top, fmaxop);
          const unsigned long long holders = __ballot(top == kth);
          if (holders && lane == __ffsll((long long)holders) - 1) top = -__builtin_inff();
        }
        if (lane == 0) s_wbound[wave] = kth;
      }
    }
    auto tally = [&](float q, int shift, int nbins) {
      atomicAdd(&s_hist[(__builtin_bit_cast(uint32_t, q) >> shift) & (uint32_t)(nbins - 1)],
                (1ull << 41) + (unsigned long long)(q * 1099511627776.0f));   // count | q * 2^40
    };
    // A cut bin of up to 64 values (the usual case under a top-k limit: a handful) is finished by
    // wave 0 alone, one value per lane, in place of the two lower levels and their ten workgroup
    // phases.  What lies above a lane's value is summed exactly as the bins would (count | mass
    // in one 64-bit word), over one broadcast per listed value.  The passing bit patterns of the
    // bin's range form an upper range that can only start at the range's lowest pattern or at one
    // of the listed values -- between two listed values nothing changes -- so the lowest passing
    // one of those is the pattern the radix levels would end on; when nothing passes (top_p <= 0)
    // both follow the largest value.
    constexpr int kWaveFinishMax = 64;
    auto finish_in_wave0 = [&]() {
      __syncthreads();                       // the list is complete
      if (wave == 0) {
        const bool live = lane < listed;
        const float q = live ? s_list[live ? lane : 0] : 0.f;
        const uint32_t vb = __builtin_bit_cast(uint32_t, q);
        const unsigned long long w = live ? (1ull << 41) + (unsigned long long)(q * 1099511627776.0f) : 0ull;
        const uint32_t lo = prefix << kHistShift;
        unsigned long long above_own = acc, above_lo = acc;
        for (int j = 0; j < listed; ++j) {   // (wave-uniform trip count)
          const uint32_t vj = (uint32_t)__builtin_amdgcn_readlane((int)vb, j);
          const unsigned long long wj =
              ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(w >> 32), j) << 32) |
              (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)w, j);
          above_own += vj > vb ? wj : 0ull;
          above_lo += vj > lo ? wj : 0ull;
        }
        auto passes = [&](unsigned long long a) { return ok(mass_of(a), (int)(a >> 41)); };
        uint32_t v = lo;
        unsigned long long a_final = above_lo;
        if (!passes(above_lo) && listed > 0) {
          const bool pass_own = live && passes(above_own);
          int cand = wave_reduce(pass_own ? (int)vb : 0x7fffffff, [](int x, int y) { return min(x, y); });
          if (cand == 0x7fffffff) cand = wave_reduce(live ? (int)vb : 0, maxi);   // nothing passes: the largest value
          v = (uint32_t)cand;
          const unsigned long long holders = __ballot(live && vb == v);           // not empty: v is a listed value
          const int src = __ffsll((long long)holders) - 1;
          a_final = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(above_own >> 32), src) << 32) |
                    (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)above_own, src);
        }
        const int same = __popcll(__ballot(live && vb == v));
        if (lane == 0) {
          s_cutbits = v;
          s_above = a_final;
          s_cutcount = same;
        }
      }
      __syncthreads();
      prefix = s_cutbits;
      acc = s_above;
    };
    if constexpr (VPT == 0) {
      if (bounded) {
        if (tid == 0) s_cand_n = 0;
        __syncthreads();                                 // the waves' bounds are in
        float lowest = s_wbound[0];
#pragma unroll
        for (int w = 1; w < kMaxWaves; ++w) lowest = fminf(lowest, s_wbound[w < (nt >> 6) ? w : 0]);
        if (lowest > -__builtin_inff()) {                // (a wave without k' finite values: no bound)
          tally_from = div_shared(exp_nonpos(lowest - mx), z, inv_z_row) * (1.0f - 0x1p-20f);
          x_floor = lowest - 0.01f;                      // (where 0.01 is below the spacing of floats at `lowest`,
          floor_on = true;                               //  this is `lowest` itself: the test is >=)
          stream_row([&](float x, int i) {
            const float xs = by_temp(x);
            if (xs >= x_floor) {
              const int at = atomicAdd(&s_cand_n, 1);
              if (at < kCandCap) {
                s_cand_x[at] = xs;
                s_cand_i[at] = i;
              }
            }
          });
          __syncthreads();
          cand_n = s_cand_n;
          if (cand_n <= kCandCap) {                      // (else: the row is walked as before)
            if (tid < cand_n) {
              const int mine_i = s_cand_i[tid];
              int rank = 0;
              for (int m = 0; m < cand_n; ++m) rank += s_cand_i[m] < mine_i ? 1 : 0;
              s_cand_sx[rank] = s_cand_x[tid];
              s_cand_si[rank] = mine_i;
            }
            __syncthreads();
            cand_on = true;
          }
        }
      }
    }
#pragma unroll
    for (int L = 0; L < 3; ++L) {
      if (L > 0 && listed >= 0 && listed <= kWaveFinishMax) continue;   // (finished below, by wave 0)
      const int shift = kLevelShift[L], nbins = 1 << kLevelBits[L];
      for (int i = tid; i < nbins; i += nt) s_hist[i] = 0ull;
      __syncthreads();
      STAMP(3 + 5 * L);
      if constexpr (VPT > 0) {
        if (L == 0 && bounded) {
          float lowest = s_wbound[0];
#pragma unroll
          for (int w = 1; w < kMaxWaves; ++w) lowest = fminf(lowest, s_wbound[w < (nt >> 6) ? w : 0]);
          tally_from = fmaxf(lowest, 0.f);               // (a wave without k' values reports -1: no bound)
        }
      }
      if (L > 0 && listed >= 0) {
        for (int i = tid; i < listed; i += nt) {
          const float q = s_list[i];
          if (L == 1 || (__builtin_bit_cast(uint32_t, q) >> (shift + kLevelBits[L])) == prefix) tally(q, shift, nbins);
        }
      } else {
        for_each([&](float q, int) {
          if (q >= tally_from && (L == 0 || (__builtin_bit_cast(uint32_t, q) >> (shift + kLevelBits[L])) == prefix))
            tally(q, shift, nbins);
        });
      }
      __syncthreads();
      STAMP(4 + 5 * L);
      {
        // Every thread owns nbins / nt consecutive bins (at least one; threads past the last bin
        // idle), thread 0 the highest ones: what lies above a thread's bins is then a prefix sum
        // over the threads before it -- within the wave by DPP shifts (no LDS round trips), across
        // waves through LDS.  `run` = everything above the bin.  Passing bins form an upper range,
        // and whether a bin's lower neighbour passes follows from the bin's own content, so the
        // one thread that owns the lowest passing bin sees that by itself and publishes the cut:
        // no workgroup-wide minimum.  (One wave scanning all 2048 bins alone, 32 per lane, took
        // 3-4 us of a level's 6-9; suffix sums by wave shuffles and a minimum over the workgroup,
        // 2.3 us per level.)
        const int bpt = nbins >= nt ? nbins / nt : 1;
        const int hi_bin = nbins - 1 - tid * bpt;          // this thread's highest bin; its others follow downwards
        const bool has = tid * bpt < nbins;
        unsigned long long own = 0ull;
        if (has)
          for (int i = 0; i < bpt; ++i) own += s_hist[hi_bin - i];
        const unsigned long long incl = wave_prefix_sum_u64(own);   // lanes <= this one
        if (lane == kWave - 1) s_wsum[wave] = incl;
        if (tid == 0) s_cutbin = -1;
        __syncthreads();
        // what the waves before this one hold: one entry per lane, a prefix sum over the first row of
        // lanes, the entry of wave - 1 broadcast (sixteen reads and selected additions per thread before)
        const unsigned long long mine_w = lane < (nt >> 6) ? s_wsum[lane < kMaxWaves ? lane : 0] : 0ull;
        const unsigned long long upto = wave_prefix_sum_u64(mine_w);
        unsigned long long run = acc + (incl - own);
        if (wave > 0) run += lane_u64(upto, wave - 1);
        if (has) {
          bool pass_here = ok(mass_of(run), (int)(run >> 41));
          for (int i = 0; i < bpt; ++i) {
            const int bin = hi_bin - i;
            const unsigned long long h = s_hist[bin];
            const unsigned long long below = run + h;      // what lies above the next lower bin
            const bool pass_below = bin > 0 && ok(mass_of(below), (int)(below >> 41));
            if (pass_here && !pass_below) {                // the lowest passing bin
              s_cutbin = bin;
              s_above = run;
              s_cutcount = (int)(h >> 41);
            }
            pass_here = pass_below;
            run = below;
          }
        }
        __syncthreads();
        STAMP(5 + 5 * L);
        if (s_cutbin < 0) {
          // nothing passes (top_p <= 0): the largest value alone survives -> follow the highest bin in use
          int top = -1;
          if (has)
            for (int i = bpt - 1; i >= 0; --i) top = s_hist[hi_bin - i] != 0ull ? hi_bin - i : top;
          const int sel = max(block_reduce(top, maxi, xch, slot), 0);
          if (has && sel <= hi_bin && sel > hi_bin - bpt) {
            unsigned long long r = acc;                    // everything above the bin: this level's bins above it are empty
            s_cutbin = sel;
            s_above = r;
            s_cutcount = (int)(s_hist[sel] >> 41);
          }
        }
      }
      __syncthreads();
      prefix = (prefix << kLevelBits[L]) | (uint32_t)s_cutbin;
      acc = s_above;
      STAMP(6 + 5 * L);
      if (L == 0 && s_cutcount <= kListCap) {     // (the bin's exact population: the list cannot overflow)
        listed = s_cutcount;
        if (tid == 0) s_list_n = 0;
        __syncthreads();
        for_each([&](float q, int) {
          if (q >= tally_from && (__builtin_bit_cast(uint32_t, q) >> kHistShift) == prefix)
            s_list[atomicAdd(&s_list_n, 1)] = q;
        });
        // (the next level's barrier after zeroing the bins orders these writes before its reads)
      }
      STAMP(7 + 5 * L);
    }
    if (listed >= 0 && listed <= kWaveFinishMax) finish_in_wave0();
    STAMP(18);
    vstar = __builtin_bit_cast(float, prefix);
    const int ties = s_cutcount;
    const FC a = {mass_of(acc), (int)(acc >> 41)};   // strictly above vstar
    // Of the `ties` copies of vstar, copy j (0-based, lowest index first -- the stable descending order)
    // is kept iff mass_above + j * v < top_p and count_above + j < k.
    int keep = ties;
    if (use_p && vstar > 0.f) {
      int jp = (int)fminf(ceilf((tp - a.f) / vstar), 2.0e9f);
      while (jp > 0 && !(a.f + (float)(jp - 1) * vstar < tp)) --jp;
      while (jp < ties && (a.f + (float)jp * vstar < tp)) ++jp;
      keep = min(keep, jp);
    }
    if (tk > 0) keep = min(keep, kk - a.c);
    keep = max(keep, a.c == 0 ? 1 : 0);          // the top token always stays (05:624)
    if (keep < ties && vstar > 0.f) {            // copies of an underflowed 0 carry no mass either way
      // tie_end = 1 + the index of the keep-th copy in index order
      if (tid == 0) s_tie_n = 0;
      __syncthreads();
      for_each([&](float q, int i) {
        if (q == vstar) {
          const int at = atomicAdd(&s_tie_n, 1);
          if (at < kTieCap) s_tie_i[at] = i;
        }
      });
      __syncthreads();
      if (keep <= 0) {
        tie_end = 0;
      } else if (ties <= kTieCap) {
        // few copies (the usual case: two or three values that happen to coincide): rank them in wave 0
        if (wave == 0) {
          const int mine_i = lane < ties ? s_tie_i[lane] : 0x7fffffff;
          int rank = 0;
          for (int m = 0; m < ties; ++m) rank += s_tie_i[m] < mine_i ? 1 : 0;
          if (lane < ties && rank == keep - 1) s_cutbin = mine_i + 1;
        }
        __syncthreads();
        tie_end = s_cutbin;
      } else {
        // many equal values (constant stretches of the row): bisect on the index over the whole row
        int xl = 1, xh = vocab;
        while (xl < xh) {
          const int xm = xl + ((xh - xl) >> 1);
          int c = 0;
          for_each([&](float q, int i) { c += (q == vstar && i < xm); });
          c = block_reduce(c, addi, xch, slot);
          if (c >= keep) xh = xm; else xl = xm + 1;
        }
        tie_end = xl;
      }
    }
    kept_mass = a.f + (float)keep * vstar;
  }
  STAMP(19);
  // Kept: q > vstar, and of the copies of vstar those with index < tie_end.  Almost always either all
  // copies stay (q >= vstar) or none does (q > vstar); the choice is the same for the whole row, and
  // the passes below run over 32 elements per thread, so they take the one-comparison forms.
  const int keep_mode = tie_end >= vocab ? 1 : tie_end <= 0 ? 0 : 2;
  auto is_kept = [&](float q, int i) { return q > vstar || (q == vstar && i < tie_end); };
  // kept ? value : 0 for the thread's j-th register-resident element, one comparison in modes 0 and 1
  // (vstar > -1: the "no element" marker fails both)
  auto kept_value = [&](int j, int mode) {
    const float q = pr[j < NR ? j : 0];
    if (mode == 1) return q >= vstar ? q : 0.f;
    if (mode == 0) return q > vstar ? q : 0.f;
    return (q >= 0.f && is_kept(q, eidx(j))) ? q : 0.f;
  };

#ifndef SAMPLER_TIMING
  if (p.out_probs) {
    float* dst = p.out_probs + (int64_t)b * vocab;
    const bool floored = floor_on, from_list = cand_on;
    floor_on = cand_on = false;                           // every element gets its entry
    for_each([&](float q, int i) { dst[i] = is_kept(q, i) ? q / kept_mass : 0.f; });
    floor_on = floored;
    cand_on = from_list;
  }
#endif

  // One draw.  Kept mass is walked thread by thread (any fixed order gives the
  // same distribution): exclusive scan of per-thread kept sums, the thread whose
  // interval contains the target walks its own elements.
  STAMP(20);
  const float target = s_uniform * kept_mass;
  float mine = 0.f;
  if constexpr (VPT > 0) {
    if (keep_mode == 1) {
#pragma unroll
      for (int j = 0; j < VPT; ++j) mine += kept_value(j, 1);
    } else if (keep_mode == 0) {
#pragma unroll
      for (int j = 0; j < VPT; ++j) mine += kept_value(j, 0);
    } else {
#pragma unroll
      for (int j = 0; j < VPT; ++j) mine += kept_value(j, 2);
    }
  } else {
    for_each([&](float q, int i) { mine += is_kept(q, i) ? q : 0.f; });
  }
  const float incl = wave_prefix_sum(mine);
  STAMP(21);
  if (lane == 63) s_scan[wave] = incl;
  if (tid == 0) s_pick = 0x7fffffff;
  __syncthreads();
  float base = 0.f;
#pragma unroll
  for (int w = 0; w < kMaxWaves; ++w) base += w < wave ? s_scan[w] : 0.f;   // all reads issue together
  const float excl = base + incl - mine;
  if (mine > 0.f && excl <= target && target < excl + mine) atomicMin(&s_pick, tid);
  __syncthreads();
  STAMP(22);
  const int picker = s_pick;
  if (picker == 0x7fffffff) {
    // rounding pushed the target past the walked total (or between two threads' intervals): the
    // last kept element takes it.  Rare, so its reduction stays off the common path.
    int last_kept = -1;
    for_each([&](float q, int i) { last_kept = is_kept(q, i) ? max(last_kept, i) : last_kept; });
    last_kept = block_reduce(last_kept, maxi, xch, slot);
    if (tid == 0) p.out[b] = max(last_kept, 0);
    return;
  }
  if constexpr (VPT > 0) {
    // The picking thread's elements go through LDS to the lanes of wave 0, which finds the first one
    // whose running sum passes the target with a prefix scan (one thread walking its 32 registers
    // alone costs more than the rest of the draw).
    if (tid == picker) {
      if (keep_mode == 1) {
#pragma unroll
        for (int j = 0; j < VPT; ++j) s_walk[j] = kept_value(j, 1);
      } else if (keep_mode == 0) {
#pragma unroll
        for (int j = 0; j < VPT; ++j) s_walk[j] = kept_value(j, 0);
      } else {
#pragma unroll
        for (int j = 0; j < VPT; ++j) s_walk[j] = kept_value(j, 2);
      }
      s_walk_base = excl;
    }
    __syncthreads();
    STAMP(23);
    if (wave == 0) walk_pick(picker, target);
  } else {
    // Streamed row: wave 0 re-reads the picking thread's elements, 64 at a time in that thread's
    // order, and finds the first kept one whose running sum passes the target (one thread walking
    // its hundred-odd elements alone, two divisions and an exponential each, took as long as a
    // whole pass of the workgroup over the row).
    if (cand_on) {                     // the picking thread's one element
      if (tid == picker) p.out[b] = s_cand_si[tid];
      return;
    }
    if (tid == picker) s_walk_base = excl;
    __syncthreads();
    if (wave != 0) return;
    constexpr int GE = VEC ? EPL : 1;
    const int nload = VEC ? vocab / EPL : vocab;
    const int owned = picker < nload ? ((nload - 1 - picker) / nt + 1) * GE : 0;
    float carry = s_walk_base;
    int last_kept = -1;
    for (int j0 = 0; j0 < owned; j0 += kWave) {
      const int j = j0 + lane;
      const int idx = ((j / GE) * nt + picker) * GE + (j % GE);
      const float q = j < owned ? prob_of(Tag::to_float(row[j < owned ? idx : 0])) : 0.f;
      const bool kept = j < owned && is_kept(q, idx);
      const float run = wave_prefix_sum(kept ? q : 0.f) + carry;
      const unsigned long long kept_lanes = __ballot(kept);
      const unsigned long long hit = __ballot(kept && run > target);
      if (hit) {
        if (lane == __ffsll((long long)hit) - 1) p.out[b] = idx;
        return;
      }
      if (kept_lanes) last_kept = __builtin_amdgcn_readlane(idx, 63 - __clzll((long long)kept_lanes));
      carry = lane_f(run, 63);
    }
    if (lane == 0) p.out[b] = max(last_kept, 0);
  }
}

template <class Tag>
int launch_sample(const SampleParams& p, int B, hipStream_t s) {
  const int v = p.vocab;
  // 16-byte loads when every row starts on a 16-byte boundary and holds whole pieces
  const bool vec = (((uintptr_t)p.logits | (uintptr_t)(p.row_stride * Tag::kBytes)) & 15) == 0 && v % Tag::kEPL == 0;
  // Register-resident rows up to 32768 entries (1024 threads x 32 values: 16 waves per CU hide
  // the latency of the dependent reduction chains better than 512 x 64); longer rows are
  // recomputed from the logits on every pass.
  if (v <= 4 * 256) hipLaunchKernelGGL((sample_kernel<Tag, 4, 256, false>), dim3(B), dim3(256), 0, s, p);
  else if (v <= 16 * 1024) {
    if (vec) hipLaunchKernelGGL((sample_kernel<Tag, 16, 1024, true>), dim3(B), dim3(1024), 0, s, p);
    else hipLaunchKernelGGL((sample_kernel<Tag, 16, 1024, false>), dim3(B), dim3(1024), 0, s, p);
  } else if (v <= 32 * 1024) {
    if (vec) hipLaunchKernelGGL((sample_kernel<Tag, 32, 1024, true>), dim3(B), dim3(1024), 0, s, p);
    else hipLaunchKernelGGL((sample_kernel<Tag, 32, 1024, false>), dim3(B), dim3(1024), 0, s, p);
  } else {
    if (vec) hipLaunchKernelGGL((sample_kernel<Tag, 0, 1024, true>), dim3(B), dim3(1024), 0, s, p);
    else hipLaunchKernelGGL((sample_kernel<Tag, 0, 1024, false>), dim3(B), dim3(1024), 0, s, p);
  }
  return check_launch("sample_kernel");
}

}  // namespace
}  // namespace mi355

using namespace mi355;

extern "C" int mi355_sample_argmax(int32_t* out_ids, const void* logits, int32_t B, int32_t vocab,
                                   int64_t row_stride, int32_t dtype, void* stream) {
  if (B == 0) return 0;
  if (!out_ids || !logits) return fail(MI355_EINVAL, "sample_argmax: null pointer");
  if (B < 0 || vocab <= 0 || row_stride < vocab) return fail(MI355_EINVAL, "sample_argmax: bad shape");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int threads = vocab >= 16384 ? 1024 : vocab >= 4096 ? 512 : vocab >= 1024 ? 256 : 64;
  const int ebytes = dtype == MI355_F32 ? 4 : 2;
  const int vec_ok = (((uintptr_t)logits & 15) == 0) && ((row_stride * ebytes) % 16 == 0);
  switch (dtype) {
    case MI355_F32:
      hipLaunchKernelGGL(argmax_kernel<F32>, dim3(B), dim3(threads), 0, s, out_ids,
                         static_cast<const float*>(logits), vocab, row_stride, vec_ok);
      break;
    case MI355_BF16:
      hipLaunchKernelGGL(argmax_kernel<BF16>, dim3(B), dim3(threads), 0, s, out_ids,
                         static_cast<const uint16_t*>(logits), vocab, row_stride, vec_ok);
      break;
    case MI355_F16:
      hipLaunchKernelGGL(argmax_kernel<F16>, dim3(B), dim3(threads), 0, s, out_ids,
                         static_cast<const uint16_t*>(logits), vocab, row_stride, vec_ok);
      break;
    default:
      return fail(MI355_EINVAL, "sample_argmax: dtype %d", dtype);
  }
  return check_launch("argmax_kernel");
}

extern "C" int mi355_sample_topk_topp(int32_t* out_ids, const void* logits, const float* temperature,
                                      const float* top_p, const int32_t* top_k, uint64_t seed,
                                      uint64_t offset, int32_t B, int32_t vocab, int64_t row_stride,
                                      int32_t dtype, float* out_probs, void* stream) {
  if (B == 0) return 0;
  if (!out_ids || !logits || !temperature) return fail(MI355_EINVAL, "sample_topk_topp: null pointer");
  if (B < 0 || vocab <= 0 || row_stride < vocab) return fail(MI355_EINVAL, "sample_topk_topp: bad shape");
  SampleParams p;
  p.out = out_ids;
  p.logits = logits;
  p.temperature = temperature;
  p.top_p = top_p;
  p.top_k = top_k;
  p.out_probs = out_probs;
  p.seed = seed;
  p.offset = offset;
  p.vocab = vocab;
  p.row_stride = row_stride;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case MI355_F32: return launch_sample<F32>(p, B, s);
    case MI355_BF16: return launch_sample<BF16>(p, B, s);
    case MI355_F16: return launch_sample<F16>(p, B, s);
  }
  return fail(MI355_EINVAL, "sample_topk_topp: dtype %d", dtype);
}
