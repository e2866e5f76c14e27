// Token sampling kernels: greedy argmax and temperature / top-k / top-p sampling.
//
// One workgroup owns one row of logits.  Row maxima, argmax pairs, probability
// sums and counts are reduced with wave shuffles first and a small LDS exchange
// across the waves of the workgroup second.
#include <math.h>

#include "common.h"

namespace mi355 {
namespace {

constexpr int kMaxWaves = 16;
constexpr int kCandCapMax = 1024;   // cut-search candidates one workgroup may park in LDS
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
  // 16-byte pieces, four independent loads in flight per thread; pieces are visited in
  // increasing index order so "first occurrence" stays well defined per thread.
  const int nvec = vec_ok ? vocab / EPL : 0;
  constexpr int kFly = 4;
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
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ov = __shfl_xor(cv, off);
    const int oi = __shfl_xor(ci, off);
    if (cand_better(ov, oi, cv, ci)) {
      cv = ov;
      ci = oi;
    }
  }
  const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  if (lane == 0) {
    s_v[wave] = cv;
    s_i[wave] = ci;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < nw; ++w)
      if (cand_better(s_v[w], s_i[w], cv, ci)) {
        cv = s_v[w];
        ci = s_i[w];
      }
    out[blockIdx.x] = ci == 0x7fffffff ? 0 : ci;
  }
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

// VPT > 0: the row's probabilities live in registers (VPT per thread).
// VPT == 0: they are recomputed from the logits on every pass (any vocab size).
template <class Tag, int VPT, int NT>
__global__ void __launch_bounds__(NT) sample_kernel(const SampleParams p) {
  __shared__ Exchange xch;
  __shared__ float s_scan[kMaxWaves];
  __shared__ int s_pick;
  // cut search state (see the filter section)
  __shared__ unsigned long long s_hist[kHistBins];
  __shared__ unsigned long long s_above;
  constexpr int kCandCap = kCandCapMax;
  __shared__ float s_cand[kCandCap];
  __shared__ int s_cand_i[kCandCap];
  __shared__ int s_cutbin;
  __shared__ float s_res_f[2];
  __shared__ int s_res_i[3];
  int slot = 0;
  const int tid = threadIdx.x, nt = blockDim.x, b = blockIdx.x;
  const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  const typename Tag::elem_t* row = static_cast<const typename Tag::elem_t*>(p.logits) + (int64_t)b * p.row_stride;
  const int vocab = p.vocab;
  const float temp = p.temperature[b];
  const float tp = p.top_p ? p.top_p[b] : 1.0f;
  const int tk = p.top_k ? p.top_k[b] : 0;
  auto fmaxop = [](float a, float c) { return fmaxf(a, c); };
  auto fminop = [](float a, float c) { return fminf(a, c); };
  auto addf = [](float a, float c) { return a + c; };
  auto addi = [](int a, int c) { return a + c; };
  auto maxi = [](int a, int c) { return max(a, c); };
  auto mini = [](int a, int c) { return min(a, c); };

  // p = softmax(x / T): the division comes first, as in the reference.  Register-resident
  // rows are divided and exponentiated exactly once.
  constexpr int NR = VPT > 0 ? VPT : 1;
  float pr[NR];
  float mx = -__builtin_inff(), z = 0.f;
  const float inv_t = 1.0f / temp;
  // Rows that do not fit the registers are streamed from memory on every pass: fn(logit, index)
  // for the thread's elements tid, tid + nt, ... in that order, kStreamU loads in flight per thread
  // (the index is clamped instead of branching around the load, so that the loads of a group
  // issue back to back; a load under a branch waits for its data before the next one starts).
  constexpr int kStreamU = 8;
  auto stream_row = [&](auto fn) {
    for (int i0 = tid; i0 < vocab; i0 += nt * kStreamU) {
      float x[kStreamU];
#pragma unroll
      for (int u = 0; u < kStreamU; ++u) x[u] = Tag::to_float(row[min(i0 + u * nt, vocab - 1)]);
#pragma unroll
      for (int u = 0; u < kStreamU; ++u)
        if (i0 + u * nt < vocab) fn(x[u], i0 + u * nt);
    }
  };
  if constexpr (VPT > 0) {
    // all of the thread's loads first (clamped index, no branch), then the arithmetic
#pragma unroll
    for (int j = 0; j < VPT; ++j) pr[j] = Tag::to_float(row[min(j * nt + tid, vocab - 1)]);
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      pr[j] = (j * nt + tid < vocab) ? div_shared(pr[j], temp, inv_t) : -__builtin_inff();
      mx = fmaxf(mx, pr[j]);
    }
    mx = block_reduce(mx, fmaxop, xch, slot);
#if KO_STOP == 1
    if (tid == 0) p.out[b] = (int)mx; return;
#endif
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
      pr[j] = exp_nonpos(pr[j] - mx);    // absent elements: exp(-inf) = 0
      z += pr[j];
    }
    z = block_reduce(z, addf, xch, slot);
#if KO_STOP == 2
    if (tid == 0) p.out[b] = (int)z; return;
#endif
    const float inv_z = 1.0f / z;
#pragma unroll
    for (int j = 0; j < VPT; ++j) pr[j] = (j * nt + tid < vocab) ? div_shared(pr[j], z, inv_z) : -1.0f;   // -1 marks "no element"
  } else {
    stream_row([&](float x, int) { mx = fmaxf(mx, div_shared(x, temp, inv_t)); });
    mx = block_reduce(mx, fmaxop, xch, slot);
    stream_row([&](float x, int) { z += exp_nonpos(div_shared(x, temp, inv_t) - mx); });
    z = block_reduce(z, addf, xch, slot);
  }
  const float inv_z_row = 1.0f / z;
  auto prob_of = [&](float x) { return div_shared(exp_nonpos(div_shared(x, temp, inv_t) - mx), z, inv_z_row); };
  // Visits every element this thread owns: fn(probability, index).
  auto for_each = [&](auto fn) {
    if constexpr (VPT > 0) {
#pragma unroll
      for (int j = 0; j < VPT; ++j)
        if (pr[j] >= 0.f) fn(pr[j], j * nt + tid);
    } else {
      stream_row([&](float x, int i) { fn(prob_of(x), i); });
    }
  };
  // (mass, count) of elements strictly above v, over the whole row.
  auto above = [&](float v) {
    FC a = {0.f, 0};
    if constexpr (VPT > 0) {
      // v >= 0 and absent elements are -1, so no presence test is needed here
#pragma unroll
      for (int j = 0; j < VPT; ++j) {
        const bool up = pr[j] > v;
        a.f += up ? pr[j] : 0.f;
        a.c += up ? 1 : 0;
      }
    } else {
      for_each([&](float q, int) {
        if (q > v) {
          a.f += q;
          a.c += 1;
        }
      });
    }
    a.f = wave_reduce(a.f, addf);
    a.c = wave_reduce(a.c, addi);
    if (lane == 0) {
      xch.f[slot][wave] = a.f;
      xch.c[slot][wave] = a.c;
    }
    __syncthreads();
    FC r = {xch.f[slot][0], xch.c[slot][0]};
    for (int w = 1; w < nw; ++w) {
      r.f += xch.f[slot][w];
      r.c += xch.c[slot][w];
    }
    slot ^= 1;
    return r;
  };

  // Cut: elements with probability > vstar are kept, and of those equal to vstar
  // the ones with index < tie_end.  Default keeps everything.
  float vstar = -1.0f, kept_mass;
  int tie_end = vocab;
  const bool filter = tp < 1.0f || tk > 0;
  if (!filter) {
    // nothing is cut: the distribution is the softmax itself, as in the reference, which renormalises
    // only after a cut (05_mini_vllm/mini_vllm.py:619-626).  The walked total can differ from 1 by
    // rounding; a target beyond it falls to the last element (see the draw).
    kept_mass = 1.0f;
#if KO_STOP == 3
    if (tid == 0) p.out[b] = (int)kept_mass; return;
#endif
  } else {
    const int kk = tk > 0 ? tk : 0x7fffffff;
    // A value v survives (its first copy in descending order is kept) iff the mass strictly
    // above it is < top_p and fewer than k elements are above it: ok(mass, count).  That
    // predicate is monotone in v, so the smallest surviving bit pattern can be bracketed and
    // bisected.  Four stages, every sum in a fixed order (results repeat run to run):
    //  1. histogram of the row over the top 12 bits of the probability's bit pattern (one
    //     LDS atomic per element: count and fixed-point mass packed in 64 bits) -> the bin
    //     that holds the cut, with the exact (mass, count) above that bin;
    //  2. whole-row bisection passes only while more than kCandCap elements sit in the bracket;
    //  3. those elements (value, index) are parked in LDS in (thread, element) order;
    //  4. one wave finishes on the candidates alone: bisection, the smallest kept value, the
    //     ties at that value and which of them (lowest indices first) stay.
    // top_p >= 1 switches the mass test off altogether (05_mini_vllm/mini_vllm.py:619 applies the
    // nucleus cut only `if top_p < 1.0`): with a peaked row the mass above the small entries
    // rounds to 1.0f, and `mass < 1.0f` would drop them from a row that only has a top-k limit.
    const bool use_p = tp < 1.0f;
    auto ok = [&](float mass, int count) { return (!use_p || mass < tp) && count < kk; };
    float pmax = 0.f;
    for_each([&](float q, int) { pmax = fmaxf(pmax, q); });
    pmax = block_reduce(pmax, fmaxop, xch, slot);
    const uint32_t pmax_bits = __builtin_bit_cast(uint32_t, pmax);

    // ---- 1. histogram
    for (int i = tid; i < kHistBins; i += nt) s_hist[i] = 0ull;
    __syncthreads();
    for_each([&](float q, int) {
      const uint32_t key = __builtin_bit_cast(uint32_t, q) >> kHistShift;     // q in [0, 1] -> key < kHistBins
      atomicAdd(&s_hist[key], (1ull << 41) + (unsigned long long)(q * 1099511627776.0f));   // count | q * 2^40
    });
    __syncthreads();
    if (wave == 0) {
      // lane l owns bins [l * kPer, (l + 1) * kPer); suffix sums over the lanes above it, then a walk
      // down its own bins for the lowest one whose "everything above this bin" still passes ok()
      constexpr int kPer = kHistBins / kWave;
      unsigned long long own = 0ull;
#pragma unroll 4
      for (int i = 0; i < kPer; ++i) own += s_hist[lane * kPer + i];
      unsigned long long incl = own;    // inclusive suffix scan: lanes >= l
#pragma unroll
      for (int off = 1; off < kWave; off <<= 1) {
        const unsigned long long o = __shfl_down(incl, off);
        if (lane + off < kWave) incl += o;
      }
      unsigned long long run = incl - own;   // sum of the bins above this lane's range
      int best = 0x7fffffff;
#pragma unroll 2
      for (int i = kPer - 1; i >= 0; --i) {
        const int b = lane * kPer + i;
        const float mass = (float)(run & ((1ull << 41) - 1)) * 0x1p-40f;
        if (ok(mass, (int)(run >> 41))) best = b;       // everything above bin b passes: b is a candidate cut bin
        run += s_hist[b];
      }
      best = wave_reduce(best, mini);
      if (lane == 0) s_cutbin = best;
    }
    __syncthreads();
    int cutbin = s_cutbin;
    if (cutbin == 0x7fffffff) cutbin = (int)(pmax_bits >> kHistShift);   // nothing passes (top_p <= 0): top bin
    cutbin = min(cutbin, (int)(pmax_bits >> kHistShift));
    // (mass, count) strictly above the cut bin, exact: integer sums of the histogram
    if (wave == 0) {
      constexpr int kPer = kHistBins / kWave;
      unsigned long long part = 0ull;
#pragma unroll 4
      for (int i = 0; i < kPer; ++i) {
        const int b = lane * kPer + i;
        part += b > cutbin ? s_hist[b] : 0ull;
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
      if (lane == 0) s_above = part;
    }
    __syncthreads();
    const unsigned long long packed_above = s_above;
    uint32_t lo = (uint32_t)cutbin << kHistShift;
    uint32_t hi = min((((uint32_t)cutbin + 1u) << kHistShift) - 1u, pmax_bits);
    FC a_hi = {(float)(packed_above & ((1ull << 41) - 1)) * 0x1p-40f, (int)(packed_above >> 41)};   // strictly above hi
    int n_ge_lo = a_hi.c + (int)(s_hist[cutbin] >> 41);                                            // elements >= lo

    // ---- 2. whole-row passes while the bracket holds too many elements
    while (lo < hi && n_ge_lo - a_hi.c > kCandCap) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      const FC a = above(__builtin_bit_cast(float, mid));
      if (ok(a.f, a.c)) {
        hi = mid;
        a_hi = a;
      } else {
        lo = mid + 1;
        n_ge_lo = a.c;
      }
    }

    // ---- 3. candidates = elements with value in [lo, hi], in (thread, element) order
    const float flo = __builtin_bit_cast(float, lo), fhi = __builtin_bit_cast(float, hi);
    int mine_n = 0;
    for_each([&](float q, int) { mine_n += (q >= flo && q <= fhi) ? 1 : 0; });
    int incl_n = mine_n;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int o = __shfl_up(incl_n, off);
      if (lane >= off) incl_n += o;
    }
    if (lane == kWave - 1) xch.c[slot][wave] = incl_n;
    __syncthreads();
    int base_n = 0, total_n = 0;
    for (int w = 0; w < nw; ++w) {
      const int c = xch.c[slot][w];
      base_n += w < wave ? c : 0;
      total_n += c;
    }
    slot ^= 1;
    {
      int at = base_n + incl_n - mine_n;
      for_each([&](float q, int i) {
        if (q >= flo && q <= fhi) {
          if (at < kCandCap) {
            s_cand[at] = q;
            s_cand_i[at] = i;
          }
          ++at;
        }
      });
    }
    __syncthreads();
    const int ncand = min(total_n, kCandCap);   // total_n <= kCandCap unless lo == hi with many equal values

    // ---- 4. one wave finishes on the candidates
    if (wave == 0) {
      // The candidates stay in LDS and are re-read on every use (a fixed number of independent reads
      // per lane that pipeline; holding them in registers next to the thread's slice of the row
      // overflows the 128-register budget of 1024 threads).  -1 stands for an empty slot.
      constexpr int kPerLane = kCandCap / kWave;
      auto cand = [&](int i) {
        asm volatile("" ::: "memory");   // keep the read where it is used
        const int at = lane + i * kWave;
        return at < ncand ? s_cand[at] : -1.0f;
      };
      auto cand_above = [&](float v) {   // (mass, count) of all row elements strictly above v, v inside the bracket
        float mf = 0.f;
        int mc = 0;
#pragma unroll
        for (int i = 0; i < kPerLane; ++i) {
          const float c = cand(i);
          const bool up = c > v;
          mf += up ? c : 0.f;
          mc += up ? 1 : 0;
        }
        FC r;
        r.f = wave_reduce(mf, addf) + a_hi.f;
        r.c = wave_reduce(mc, addi) + a_hi.c;
        return r;
      };
      while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const FC a = cand_above(__builtin_bit_cast(float, mid));
        if (ok(a.f, a.c)) hi = mid; else lo = mid + 1;
      }
      const float u = __builtin_bit_cast(float, lo);
      // smallest kept value = smallest candidate >= u (when the bracket collapsed onto one value with more
      // copies than the candidate buffer holds, every copy equals that value)
      float vmin = total_n > kCandCap ? fhi : __builtin_inff();
      int ties_n = 0;
#pragma unroll
      for (int i = 0; i < kPerLane; ++i) {
        const float c = cand(i);
        if (c >= u) vmin = fminf(vmin, c);
      }
      vmin = wave_reduce(vmin, fminop);
      if (vmin == __builtin_inff()) vmin = fhi;    // no candidate at or above u: cut at the bracket's top
      const FC a = cand_above(vmin);
#pragma unroll
      for (int i = 0; i < kPerLane; ++i) ties_n += (cand(i) == vmin) ? 1 : 0;
      ties_n = wave_reduce(ties_n, addi);
      // Number of tied copies kept: copy j (0-based) needs mass_above + j*v < top_p and count_above + j < k.
      const bool whole_row = total_n > kCandCap;      // more copies of one value than the buffer holds
      int keep_n = ties_n, cut_index = vocab;
      if (!whole_row) {
        if (tp < 1.0f && vmin > 0.f) {
          int jp = (int)fminf(ceilf((tp - a.f) / vmin), 2.0e9f);
          while (jp > 0 && !(a.f + (float)(jp - 1) * vmin < tp)) --jp;
          while (jp < ties_n && (a.f + (float)jp * vmin < tp)) ++jp;
          keep_n = min(keep_n, jp);
        }
        if (tk > 0) keep_n = min(keep_n, kk - a.c);
        keep_n = max(keep_n, a.c == 0 ? 1 : 0);
        if (keep_n < ties_n) {
          // keep the keep_n lowest-index copies: smallest X with #(ties at index < X) >= keep_n
          int xl = 1, xh = vocab;
          while (xl < xh) {
            const int xm = xl + ((xh - xl) >> 1);
            int c = 0;
#pragma unroll
            for (int i = 0; i < kPerLane; ++i) {
              const bool tied = cand(i) == vmin;
              c += (tied && s_cand_i[lane + i * kWave] < xm) ? 1 : 0;
            }
            c = wave_reduce(c, addi);
            if (c >= keep_n) xh = xm; else xl = xm + 1;
          }
          cut_index = xl;
        }
      }
      if (lane == 0) {
        s_res_f[0] = vmin;
        s_res_f[1] = a.f;
        s_res_i[0] = a.c;
        s_res_i[1] = whole_row ? -1 : keep_n;   // -1: ties are counted and cut over the whole row
        s_res_i[2] = cut_index;
      }
    }
    __syncthreads();
    vstar = s_res_f[0];
    const FC a = {s_res_f[1], s_res_i[0]};
    int keep = s_res_i[1];
    tie_end = s_res_i[2];
    if (keep < 0) {
      int t = 0;
      for_each([&](float q, int) { t += (q == vstar); });
      const int ties = block_reduce(t, addi, xch, slot);
      keep = ties;
      if (tp < 1.0f && vstar > 0.f) {
        int jp = (int)fminf(ceilf((tp - a.f) / vstar), 2.0e9f);
        while (jp > 0 && !(a.f + (float)(jp - 1) * vstar < tp)) --jp;
        while (jp < ties && (a.f + (float)jp * vstar < tp)) ++jp;
        keep = min(keep, jp);
      }
      if (tk > 0) keep = min(keep, kk - a.c);
      keep = max(keep, a.c == 0 ? 1 : 0);
      if (keep < ties) {
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
  auto is_kept = [&](float q, int i) { return q > vstar || (q == vstar && i < tie_end); };

  if (p.out_probs) {
    float* dst = p.out_probs + (int64_t)b * vocab;
    for_each([&](float q, int i) { dst[i] = is_kept(q, i) ? q / kept_mass : 0.f; });
  }

  // One draw.  Kept mass is walked thread by thread (any fixed order gives the
  // same distribution): exclusive scan of per-thread kept sums, the thread whose
  // interval contains the target walks its own elements.
  const float target = philox_uniform(p.seed, p.offset + (uint64_t)b) * kept_mass;
  float mine = 0.f;
  for_each([&](float q, int i) { mine += is_kept(q, i) ? q : 0.f; });
  float incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) s_scan[wave] = incl;
  if (tid == 0) s_pick = 0x7fffffff;
  __syncthreads();
  float base = 0.f;
#pragma unroll
  for (int w = 0; w < kMaxWaves; ++w) base += w < wave ? s_scan[w] : 0.f;   // all reads issue together
  const float excl = base + incl - mine;
  if (mine > 0.f && excl <= target && target < excl + mine) atomicMin(&s_pick, tid);
  __syncthreads();
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
    __shared__ float s_walk[VPT];
    __shared__ float s_walk_base;
    if (tid == picker) {
#pragma unroll
      for (int j = 0; j < VPT; ++j) s_walk[j] = (pr[j] >= 0.f && is_kept(pr[j], j * nt + tid)) ? pr[j] : 0.f;
      s_walk_base = excl;
    }
    __syncthreads();
    if (wave == 0) {
      const float q = lane < VPT ? s_walk[lane] : 0.f;
      float run = q;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const float o = __shfl_up(run, off);
        if (lane >= off) run += o;
      }
      run += s_walk_base;
      const unsigned long long hit = __ballot(q > 0.f && run > target);
      const unsigned long long any = __ballot(q > 0.f);          // not empty: the picker's mass is > 0
      const int j = hit ? __ffsll((long long)hit) - 1 : 63 - __clzll((long long)any);
      if (lane == 0) p.out[b] = j * nt + picker;
    }
  } else {
    if (tid == picker) {
      float run = excl;
      int choice = -1, fallback = -1;
      for_each([&](float q, int i) {
        if (is_kept(q, i)) {
          fallback = i;
          run += q;
          if (choice < 0 && run > target) choice = i;
        }
      });
      p.out[b] = choice >= 0 ? choice : fallback;
    }
  }
}

template <class Tag>
int launch_sample(const SampleParams& p, int B, hipStream_t s) {
  const int v = p.vocab;
  // Register-resident rows up to 32768 entries (1024 threads x 32 values: 16 waves per CU hide
  // the latency of the dependent reduction chains better than 512 x 64); longer rows are
  // recomputed from the logits on every pass.
  if (v <= 4 * 256) hipLaunchKernelGGL((sample_kernel<Tag, 4, 256>), dim3(B), dim3(256), 0, s, p);
  else if (v <= 16 * 1024) hipLaunchKernelGGL((sample_kernel<Tag, 16, 1024>), dim3(B), dim3(1024), 0, s, p);
  else if (v <= 32 * 1024) hipLaunchKernelGGL((sample_kernel<Tag, 32, 1024>), dim3(B), dim3(1024), 0, s, p);
  else hipLaunchKernelGGL((sample_kernel<Tag, 0, 1024>), dim3(B), dim3(1024), 0, s, p);
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
