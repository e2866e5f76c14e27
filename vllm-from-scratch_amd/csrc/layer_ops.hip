// Elementwise steps that sit directly before and after the attention kernels in a
// transformer layer, fused so that a decode step is a short chain of launches:
//
//   rope_kv_write_kernel  rotate-half RoPE on q and k in place + scatter of the
//                         rotated k and of v into the paged caches by slot id
//   rmsnorm_kernel        (optional residual add) + RMS normalisation, one wave per row
//   silu_mul_kernel       SwiGLU gating: silu(gate) * up
//
// All three are HBM/launch-latency bound byte movers; arithmetic is fp32 with one
// rounding on store.
#include "common.h"

namespace mi355 {
namespace {

// ---------------------------------------------------------------------------
// RoPE + KV write
// ---------------------------------------------------------------------------
struct RopeParams {
  char* q;
  char* k;
  const char* v;
  const char* cos;
  const char* sin;
  const int64_t* positions;
  const int64_t* slots;  // may be null: rotate only
  char* kc;
  char* vc;
  int64_t q_stride_tok, k_stride_tok, v_stride_tok, s_blk, s_head, s_tok;  // elements
  int64_t num_slots;
  int T, Hq, Hkv, D, half_pieces, table_len, block_size, bs_shift;
};

// Work item = (token, head slot, piece of the first half of the head row).  Head slots
// 0..Hq-1 are query heads, the next Hkv are key heads (rotated, then also written to the
// cache), the last Hkv are value heads (copied to the cache).  A work item owns elements
// [i, i+EPL) and [i + D/2, i + D/2 + EPL) of its row: exactly the pairs RoPE mixes.
template <class Tag>
__global__ void __launch_bounds__(256) rope_kv_write_kernel(const RopeParams p) {
#pragma clang fp contract(off)  // (x*cos) + (rot*sin) as two products and a sum, like the torch expression
  constexpr int EPL = Tag::kEPL, EB = Tag::kBytes;
  const int heads = p.Hq + 2 * p.Hkv;
  const int64_t per_tok = (int64_t)heads * p.half_pieces;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)p.T * per_tok) return;
  const int t = (int)(idx / per_tok);
  const int rem = (int)(idx - (int64_t)t * per_tok);
  const int hs = rem / p.half_pieces, piece = rem - hs * p.half_pieces;
  const int lo = piece * EPL, hi = lo + p.D / 2;

  int64_t dst = -1;  // element offset of this row in the caches, -1 = no cache write
  if (hs >= p.Hq && p.slots) {
    const int64_t slot = p.slots[t];
    if (slot >= 0 && slot < p.num_slots) {
      const int64_t blk = p.bs_shift >= 0 ? (slot >> p.bs_shift) : (slot / p.block_size);
      const int64_t row = slot - blk * p.block_size;
      const int h = hs < p.Hq + p.Hkv ? hs - p.Hq : hs - p.Hq - p.Hkv;
      dst = blk * p.s_blk + h * p.s_head + row * p.s_tok;
    }
  }

  if (hs >= p.Hq + p.Hkv) {  // value head: plain copy into the cache
    if (dst < 0) return;
    const char* src = p.v + ((int64_t)t * p.v_stride_tok + (int64_t)(hs - p.Hq - p.Hkv) * p.D) * EB;
    *reinterpret_cast<u32x4*>(p.vc + (dst + lo) * EB) = *reinterpret_cast<const u32x4*>(src + lo * EB);
    *reinterpret_cast<u32x4*>(p.vc + (dst + hi) * EB) = *reinterpret_cast<const u32x4*>(src + hi * EB);
    return;
  }

  char* row = hs < p.Hq ? p.q + ((int64_t)t * p.q_stride_tok + (int64_t)hs * p.D) * EB
                        : p.k + ((int64_t)t * p.k_stride_tok + (int64_t)(hs - p.Hq) * p.D) * EB;
  int64_t pos = p.positions[t];
  if (pos < 0) pos += p.table_len;  // torch indexing semantics for negative indices
  pos = pos < 0 ? 0 : (pos >= p.table_len ? p.table_len - 1 : pos);
  const char* crow = p.cos + pos * p.D * EB;
  const char* srow = p.sin + pos * p.D * EB;
  float xl[EPL], xh[EPL], cl[EPL], ch[EPL], sl[EPL], sh[EPL], ol[EPL], oh[EPL];
  Tag::unpack(*reinterpret_cast<const u32x4*>(row + lo * EB), xl);
  Tag::unpack(*reinterpret_cast<const u32x4*>(row + hi * EB), xh);
  Tag::unpack(*reinterpret_cast<const u32x4*>(crow + lo * EB), cl);
  Tag::unpack(*reinterpret_cast<const u32x4*>(crow + hi * EB), ch);
  Tag::unpack(*reinterpret_cast<const u32x4*>(srow + lo * EB), sl);
  Tag::unpack(*reinterpret_cast<const u32x4*>(srow + hi * EB), sh);
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    ol[e] = xl[e] * cl[e] + (-xh[e]) * sl[e];
    oh[e] = xh[e] * ch[e] + xl[e] * sh[e];
  }
  const u32x4 pl = Tag::pack(ol), ph = Tag::pack(oh);
  *reinterpret_cast<u32x4*>(row + lo * EB) = pl;
  *reinterpret_cast<u32x4*>(row + hi * EB) = ph;
  if (dst >= 0) {
    *reinterpret_cast<u32x4*>(p.kc + (dst + lo) * EB) = pl;
    *reinterpret_cast<u32x4*>(p.kc + (dst + hi) * EB) = ph;
  }
}

// ---------------------------------------------------------------------------
// RMSNorm (+ residual add)
// ---------------------------------------------------------------------------
struct NormParams {
  const char* x;      // [T, hidden]
  const char* delta;  // [T, hidden] or null
  const char* w;      // [hidden]
  char* sum_out;      // x + delta (written only when delta is given)
  char* out;          // normalised rows
  int64_t x_stride, delta_stride;
  int T, hidden;
  float eps;
};

constexpr int kNormMaxPieces = 16;  // 16-byte pieces a lane holds: hidden <= 64 * 16 * EPL

template <class Tag>
__global__ void __launch_bounds__(256) rmsnorm_kernel(const NormParams p) {
#pragma clang fp contract(off)
  constexpr int EPL = Tag::kEPL, EB = Tag::kBytes;
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (t >= p.T) return;
  const int pieces = p.hidden / EPL;
  const char* xr = p.x + (int64_t)t * p.x_stride * EB;
  const char* dr = p.delta ? p.delta + (int64_t)t * p.delta_stride * EB : nullptr;
  float v[kNormMaxPieces][EPL];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < kNormMaxPieces; ++i) {
    const int pc = lane + i * kWave;
    if (pc < pieces) {
      Tag::unpack(*reinterpret_cast<const u32x4*>(xr + (int64_t)pc * 16), v[i]);
      if (dr) {
        float d[EPL];
        Tag::unpack(*reinterpret_cast<const u32x4*>(dr + (int64_t)pc * 16), d);
#pragma unroll
        for (int e = 0; e < EPL; ++e) v[i][e] += d[e];
        const u32x4 packed = Tag::pack(v[i]);
        *reinterpret_cast<u32x4*>(p.sum_out + ((int64_t)t * p.hidden + pc * EPL) * EB) = packed;
        Tag::unpack(packed, v[i]);  // normalise the value that was stored (rounded to the dtype)
      }
#pragma unroll
      for (int e = 0; e < EPL; ++e) ss += v[i][e] * v[i][e];
    }
  }
  ss = group_sum<64>(ss);
  const float rms = __builtin_sqrtf(ss / (float)p.hidden + p.eps);
#pragma unroll
  for (int i = 0; i < kNormMaxPieces; ++i) {
    const int pc = lane + i * kWave;
    if (pc < pieces) {
      float w[EPL], o[EPL];
      Tag::unpack(*reinterpret_cast<const u32x4*>(p.w + (int64_t)pc * 16), w);
#pragma unroll
      for (int e = 0; e < EPL; ++e) o[e] = w[e] * (v[i][e] / rms);
      *reinterpret_cast<u32x4*>(p.out + ((int64_t)t * p.hidden + pc * EPL) * EB) = Tag::pack(o);
    }
  }
}

// ---------------------------------------------------------------------------
// SwiGLU gating
// ---------------------------------------------------------------------------
template <class Tag>
__global__ void __launch_bounds__(256)
silu_mul_kernel(char* out, const char* gate, const char* up, int64_t gate_stride, int64_t up_stride, int T,
                int width) {
  constexpr int EPL = Tag::kEPL, EB = Tag::kBytes;
  const int pieces = width / EPL;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)T * pieces) return;
  const int t = (int)(idx / pieces), pc = (int)(idx - (int64_t)t * pieces);
  float g[EPL], u[EPL], o[EPL];
  Tag::unpack(*reinterpret_cast<const u32x4*>(gate + ((int64_t)t * gate_stride + pc * EPL) * EB), g);
  Tag::unpack(*reinterpret_cast<const u32x4*>(up + ((int64_t)t * up_stride + pc * EPL) * EB), u);
#pragma unroll
  for (int e = 0; e < EPL; ++e) o[e] = (g[e] / (1.f + expf(-g[e]))) * u[e];
  *reinterpret_cast<u32x4*>(out + ((int64_t)t * width + pc * EPL) * EB) = Tag::pack(o);
}

bool dtype_ok(int dtype) { return dtype == MI355_F32 || dtype == MI355_BF16 || dtype == MI355_F16; }
int epl_of(int dtype) { return dtype == MI355_F32 ? 4 : 8; }

}  // namespace
}  // namespace mi355

using namespace mi355;

extern "C" int mi355_rope_kv_write(void* q, void* k, const void* v, int64_t q_stride_tok,
                                   int64_t k_stride_tok, int64_t v_stride_tok, const void* cos_table,
                                   const void* sin_table, int32_t table_len, const int64_t* positions,
                                   const int64_t* slots, void* k_cache, void* v_cache, int32_t T, int32_t Hq,
                                   int32_t Hkv, int32_t D, int32_t block_size, int32_t num_blocks,
                                   int64_t kv_stride_block, int64_t kv_stride_head, int64_t kv_stride_tok,
                                   int32_t dtype, void* stream) {
  if (T == 0) return 0;
  if (!q || !k || !cos_table || !sin_table || !positions) return fail(MI355_EINVAL, "rope_kv_write: null pointer");
  if (!dtype_ok(dtype)) return fail(MI355_EINVAL, "rope_kv_write: dtype %d", dtype);
  const int epl = epl_of(dtype);
  if (T < 0 || Hq <= 0 || Hkv <= 0 || D <= 0 || D % (2 * epl) || table_len <= 0)
    return fail(MI355_EINVAL, "rope_kv_write: bad shape (T=%d Hq=%d Hkv=%d D=%d); D must be a multiple of %d", T, Hq,
                Hkv, D, 2 * epl);
  if (q_stride_tok % epl || k_stride_tok % epl || q_stride_tok < (int64_t)Hq * D || k_stride_tok < (int64_t)Hkv * D)
    return fail(MI355_EINVAL, "rope_kv_write: q/k strides must keep head rows 16-byte aligned");
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)cos_table | (uintptr_t)sin_table) & 15)
    return fail(MI355_EINVAL, "rope_kv_write: tensors must be 16-byte aligned");
  if (slots) {
    if (!v || !k_cache || !v_cache) return fail(MI355_EINVAL, "rope_kv_write: slots given without v / caches");
    if (block_size <= 0 || num_blocks <= 0) return fail(MI355_EINVAL, "rope_kv_write: block_size/num_blocks");
    if (v_stride_tok % epl || v_stride_tok < (int64_t)Hkv * D || kv_stride_block % epl || kv_stride_head % epl ||
        kv_stride_tok % epl)
      return fail(MI355_EINVAL, "rope_kv_write: v / cache strides must keep head rows 16-byte aligned");
    if (((uintptr_t)v | (uintptr_t)k_cache | (uintptr_t)v_cache) & 15)
      return fail(MI355_EINVAL, "rope_kv_write: tensors must be 16-byte aligned");
  }
  RopeParams p;
  p.q = static_cast<char*>(q);
  p.k = static_cast<char*>(k);
  p.v = static_cast<const char*>(v);
  p.cos = static_cast<const char*>(cos_table);
  p.sin = static_cast<const char*>(sin_table);
  p.positions = positions;
  p.slots = slots;
  p.kc = static_cast<char*>(k_cache);
  p.vc = static_cast<char*>(v_cache);
  p.q_stride_tok = q_stride_tok;
  p.k_stride_tok = k_stride_tok;
  p.v_stride_tok = v_stride_tok;
  p.s_blk = kv_stride_block;
  p.s_head = kv_stride_head;
  p.s_tok = kv_stride_tok;
  p.num_slots = slots ? (int64_t)num_blocks * block_size : 0;
  p.T = T;
  p.Hq = Hq;
  p.Hkv = Hkv;
  p.D = D;
  p.half_pieces = D / (2 * epl);
  p.table_len = table_len;
  p.block_size = block_size > 0 ? block_size : 1;
  p.bs_shift = ilog2_exact(p.block_size);
  const int64_t items = (int64_t)T * (Hq + 2 * Hkv) * p.half_pieces;
  const dim3 grid((unsigned)ceil_div64(items, 256)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case MI355_F32: hipLaunchKernelGGL(rope_kv_write_kernel<F32>, grid, block, 0, s, p); break;
    case MI355_BF16: hipLaunchKernelGGL(rope_kv_write_kernel<BF16>, grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL(rope_kv_write_kernel<F16>, grid, block, 0, s, p); break;
  }
  return check_launch("rope_kv_write_kernel");
}

extern "C" int mi355_rmsnorm(void* out, void* sum_out, const void* x, int64_t x_stride, const void* delta,
                             int64_t delta_stride, const void* weight, float eps, int32_t T, int32_t hidden,
                             int32_t dtype, void* stream) {
  if (T == 0) return 0;
  if (!out || !x || !weight) return fail(MI355_EINVAL, "rmsnorm: null pointer");
  if (delta && !sum_out) return fail(MI355_EINVAL, "rmsnorm: delta given without sum_out");
  if (!dtype_ok(dtype)) return fail(MI355_EINVAL, "rmsnorm: dtype %d", dtype);
  const int epl = epl_of(dtype);
  if (T < 0 || hidden <= 0 || hidden % epl) return fail(MI355_EINVAL, "rmsnorm: hidden %d must be a multiple of %d", hidden, epl);
  if (hidden > kWave * kNormMaxPieces * epl)
    return fail(MI355_EUNSUPPORTED, "rmsnorm: hidden %d above %d", hidden, kWave * kNormMaxPieces * epl);
  if (x_stride % epl || x_stride < hidden || (delta && (delta_stride % epl || delta_stride < hidden)))
    return fail(MI355_EINVAL, "rmsnorm: row strides must keep rows 16-byte aligned");
  if (((uintptr_t)out | (uintptr_t)sum_out | (uintptr_t)x | (uintptr_t)delta | (uintptr_t)weight) & 15)
    return fail(MI355_EINVAL, "rmsnorm: tensors must be 16-byte aligned");
  NormParams p;
  p.x = static_cast<const char*>(x);
  p.delta = static_cast<const char*>(delta);
  p.w = static_cast<const char*>(weight);
  p.sum_out = static_cast<char*>(sum_out);
  p.out = static_cast<char*>(out);
  p.x_stride = x_stride;
  p.delta_stride = delta_stride;
  p.T = T;
  p.hidden = hidden;
  p.eps = eps;
  const dim3 grid((unsigned)ceil_div64(T, 4)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case MI355_F32: hipLaunchKernelGGL(rmsnorm_kernel<F32>, grid, block, 0, s, p); break;
    case MI355_BF16: hipLaunchKernelGGL(rmsnorm_kernel<BF16>, grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL(rmsnorm_kernel<F16>, grid, block, 0, s, p); break;
  }
  return check_launch("rmsnorm_kernel");
}

extern "C" int mi355_silu_mul(void* out, const void* gate, int64_t gate_stride, const void* up,
                              int64_t up_stride, int32_t T, int32_t width, int32_t dtype, void* stream) {
  if (T == 0) return 0;
  if (!out || !gate || !up) return fail(MI355_EINVAL, "silu_mul: null pointer");
  if (!dtype_ok(dtype)) return fail(MI355_EINVAL, "silu_mul: dtype %d", dtype);
  const int epl = epl_of(dtype);
  if (T < 0 || width <= 0 || width % epl || gate_stride % epl || up_stride % epl || gate_stride < width ||
      up_stride < width)
    return fail(MI355_EINVAL, "silu_mul: width / strides must keep rows 16-byte aligned");
  if (((uintptr_t)out | (uintptr_t)gate | (uintptr_t)up) & 15)
    return fail(MI355_EINVAL, "silu_mul: tensors must be 16-byte aligned");
  const int64_t items = (int64_t)T * (width / epl);
  const dim3 grid((unsigned)ceil_div64(items, 256)), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* o = static_cast<char*>(out);
  const char* g = static_cast<const char*>(gate);
  const char* u = static_cast<const char*>(up);
  switch (dtype) {
    case MI355_F32: hipLaunchKernelGGL(silu_mul_kernel<F32>, grid, block, 0, s, o, g, u, gate_stride, up_stride, T, width); break;
    case MI355_BF16: hipLaunchKernelGGL(silu_mul_kernel<BF16>, grid, block, 0, s, o, g, u, gate_stride, up_stride, T, width); break;
    default: hipLaunchKernelGGL(silu_mul_kernel<F16>, grid, block, 0, s, o, g, u, gate_stride, up_stride, T, width); break;
  }
  return check_launch("silu_mul_kernel");
}
