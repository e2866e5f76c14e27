// Token splits of a decode launch merged inside that launch (VALU kernel and matrix-core kernel).
//
// The workgroups of a (sequence, kv-head group, q-head chunk) write their partial records
// (unnormalised sums, running maximum, running sum per query head), count their arrivals in a
// 64-bit cell of the workspace, and the one that arrives last combines the group's partials.
// Partials travel through device-coherent accesses (store_coherent2 / load_coherent2) instead
// of device-scope fences, whose write-back and invalidation of the whole L2 cost several times
// what the merge saves.
// The workspace is not required to be zeroed: a cell counts only while its upper 48 bits equal
// this launch's tag.  Split 0 raises the cell to (tag << 16) when it starts (an atomic max: it
// cannot undo arrivals), and an arrival is normally one fetch-add.  A workgroup that finds
// another tag -- it got there before split 0's max did, or the cell held something larger than
// any tag -- has added 1 to a value nobody reads and arrives again by compare-and-swap, which
// installs the tag itself.  The last arriver clears the cell, so that a replay of the same
// launch (a captured graph) starts over.
#pragma once

#include "common.h"
#include "decode_params.h"

namespace mi355 {

// Two floats written through to / read from device-coherent memory (relaxed 8-byte atomics at
// agent scope: sc1 accesses that neither leave dirty lines in, nor hit stale lines of, the
// per-XCD L2), for data that workgroups of one launch hand to each other.
__device__ inline void store_coherent2(float* dst, float a, float b) {
  const unsigned long long v = ((unsigned long long)__float_as_uint(b) << 32) | __float_as_uint(a);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void load_coherent2(const float* src, float& a, float& b) {
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
  a = __uint_as_float((unsigned)v);
  b = __uint_as_float((unsigned)(v >> 32));
}

__device__ inline unsigned long long* arrival_cell(const DecodeParams& p, int b, int hg, int qc) {
  return p.counters + ((int64_t)b * p.hgroups + hg) * p.qchunks + qc;
}

// Split 0's workgroup, one thread, when it starts: nothing waits for the result.
__device__ inline void tag_arrival_cell(const DecodeParams& p, int b, int hg, int qc) {
  (void)__hip_atomic_fetch_max(arrival_cell(p, b, hg, qc), p.epoch << 16, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
}

// Called by every thread of a workgroup after its partial stores (store_coherent2); true in the
// workgroup that is the last of its group's `nvalid` to get here.
__device__ inline bool arrived_last(const DecodeParams& p, int b, int hg, int qc, int nvalid, int tid) {
  __shared__ int s_last;
  // every wave's partials have been written through (vmcnt counts stores until they are
  // acknowledged) before the workgroup's arrival is counted
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  wait_stores_acknowledged();
  __syncthreads();
  if (tid == 0) {
    unsigned long long* cell = arrival_cell(p, b, hg, qc);
    unsigned long long seen = __hip_atomic_fetch_add(cell, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long next = seen + 1;
    if ((seen >> 16) != p.epoch) {
      seen = next;
      do {   // every failed exchange means another workgroup of the group has arrived: at most nsplit rounds
        next = (seen >> 16) == p.epoch ? seen + 1 : ((p.epoch << 16) | 1ull);
      } while (!__hip_atomic_compare_exchange_strong(cell, &seen, next, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT));
    }
    const int last = (int)(next & 0xffffu) == nvalid;
    if (last) __hip_atomic_store(cell, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}

// The last workgroup of a group: combines the `nvalid` partial records of each of the group's
// query heads (hpw kv heads x up to heads_per_kv query heads of chunk qc) and writes the output
// rows.  A thread owns (query head, 4 output dims) and walks the splits itself, in order: the
// merge is only taken for a handful of splits (see make_plan), so there is nothing to gain from
// spreading one head's splits over threads, and no barrier or LDS traffic is needed.
template <class Tag, int D, int NT>
__device__ inline void merge_splits_in_workgroup(const DecodeParams& p, int b, int hg, int qc, int heads_per_kv,
                                                 int nvalid, int tid) {
  constexpr int kVec = 4, kDv = D / kVec;
  const int items = p.hpw * heads_per_kv * kDv;
  for (int item = tid; item < items; item += NT) {
    const int h = item / kDv, dv = item - h * kDv;
    const int slot = h / heads_per_kv, gi = h - slot * heads_per_kv;
    if (qc * heads_per_kv + gi >= p.gq) continue;
    const int64_t bh = (int64_t)b * p.Hq + (int64_t)(hg * p.hpw + slot) * p.gq + qc * heads_per_kv + gi;
    const float* ml = p.ws_ml + bh * p.nsplit * 2;
    const float* pa = p.ws_acc + bh * p.nsplit * D + dv * kVec;
    float M = kNegBig, den = 0.f, o[kVec] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s = 0; s < nvalid; ++s) {
      float ms, ls, a[kVec];
      load_coherent2(ml + 2 * s, ms, ls);
      load_coherent2(pa + (size_t)s * D, a[0], a[1]);
      load_coherent2(pa + (size_t)s * D + 2, a[2], a[3]);
      const float Mn = fmaxf(M, ms);
      const float fo = fast_exp2(M - Mn), fn = fast_exp2(ms - Mn);
      M = Mn;
      den = __builtin_fmaf(ls, fn, den * fo);
#pragma unroll
      for (int e = 0; e < kVec; ++e) o[e] = __builtin_fmaf(a[e], fn, o[e] * fo);
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
}

}  // namespace mi355
