// KV slot write and slot-mapping kernels (integer / byte movement, HBM-bound).
#include "common.h"

namespace mi355 {
namespace {

struct KvWriteParams {
  char* kc;
  char* vc;
  const char* k;
  const char* v;
  const int64_t* slots;
  int64_t src_stride_tok, s_blk, s_head, s_tok;  // elements
  int T, Hkv, lpt /* 16-byte pieces per head row */, block_size, bs_shift;
  int64_t num_slots;
  int ebytes;
};

// One thread moves one 16-byte piece of K and the matching piece of V.
// Thread order = (token, head, piece): a wave writes whole contiguous head rows.
__global__ void __launch_bounds__(256) kv_write_slots_kernel(const KvWriteParams p) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per_tok = (int64_t)p.Hkv * p.lpt;
  if (idx >= (int64_t)p.T * per_tok) return;
  const int t = (int)(idx / per_tok);
  const int rem = (int)(idx - (int64_t)t * per_tok);
  const int h = rem / p.lpt, piece = rem - h * p.lpt;
  const int64_t slot = p.slots[t];
  if (slot < 0 || slot >= p.num_slots) return;
  const int64_t blk = p.bs_shift >= 0 ? (slot >> p.bs_shift) : (slot / p.block_size);
  const int64_t row = slot - blk * p.block_size;
  const int epl = 16 / p.ebytes;
  const int64_t src = ((int64_t)t * p.src_stride_tok + (int64_t)h * p.lpt * epl + piece * epl) * p.ebytes;
  const int64_t dst = (blk * p.s_blk + h * p.s_head + row * p.s_tok + piece * epl) * p.ebytes;
  *reinterpret_cast<u32x4*>(p.kc + dst) = *reinterpret_cast<const u32x4*>(p.k + src);
  *reinterpret_cast<u32x4*>(p.vc + dst) = *reinterpret_cast<const u32x4*>(p.v + src);
}

__global__ void __launch_bounds__(256)
slot_mapping_kernel(int64_t* out, const int32_t* bt, const int32_t* starts, const int32_t* cu, int B,
                    int max_blocks, int block_size, int bs_shift, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  // Find the owning sequence: largest b with cu[b] <= i.
  int lo = 0, hi = B - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (cu[mid] <= i) lo = mid; else hi = mid - 1;
  }
  const int pos = starts[lo] + (i - cu[lo]);
  const int lb = bs_shift >= 0 ? (pos >> bs_shift) : (pos / block_size);
  const int row = pos - lb * block_size;
  const int id = lb < max_blocks ? bt[(int64_t)lo * max_blocks + lb] : 0;
  out[i] = (int64_t)id * block_size + row;
}

// Per-step inputs of a pure decode batch, derived on the device from the three vectors the
// host uploads (last token, position, block-table row of every sequence).  Thread (r, c)
// copies entry c of sequence r's block table; thread (r, 0) also writes the scalars.
// A negative token entry -1-j means "the id the previous step sampled for its row j"
// (prev_ids[j]): the step can then be issued before the host has seen that id.
__global__ void __launch_bounds__(256)
decode_inputs_kernel(int64_t* tok_out, int64_t* pos_out, int64_t* slot_out, int32_t* ctx_out, int32_t* bt_out,
                     const int32_t* packed, const int32_t* tables, const int32_t* prev_ids, int prev_rows,
                     int rows, int table_rows, int width, int block_size, int bs_shift) {
  const int gw = bt_out ? width : 1;   // threads per row: one per gathered table entry, or one when nothing is gathered
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)rows * gw) return;
  const int r = (int)(idx / gw), c = (int)(idx - (int64_t)r * gw);
  const int pos = packed[rows + r];
  int src = packed[2 * rows + r];
  src = src < 0 ? 0 : (src >= table_rows ? table_rows - 1 : src);
  const int32_t* row = tables + (int64_t)src * width;
  if (bt_out) bt_out[idx] = row[c];
  if (c != 0) return;
  int tok = packed[r];
  if (tok < 0) {
    const int j = -1 - tok;
    tok = (prev_ids && j < prev_rows) ? prev_ids[j] : 0;
  }
  tok_out[r] = tok;
  pos_out[r] = pos < 0 ? 0 : pos;
  ctx_out[r] = pos < 0 ? 0 : pos + 1;
  int64_t slot = -1;  // padding row (position < 0): no cache write, empty context
  if (pos >= 0) {
    const int lb = bs_shift >= 0 ? (pos >> bs_shift) : (pos / block_size);
    if (lb < width) slot = (int64_t)row[lb] * block_size + (pos - lb * block_size);
  }
  slot_out[r] = slot;
}

}  // namespace
}  // namespace mi355

using namespace mi355;

extern "C" int mi355_decode_step_inputs(int64_t* token_ids, int64_t* positions, int64_t* slots,
                                        int32_t* context_lens, int32_t* block_tables, const int32_t* packed,
                                        const int32_t* table_pool, const int32_t* prev_ids,
                                        int32_t prev_rows, int32_t rows, int32_t pool_rows,
                                        int32_t width, int32_t block_size, void* stream) {
  if (rows == 0) return 0;
  if (!token_ids || !positions || !slots || !context_lens || !packed || !table_pool)
    return fail(MI355_EINVAL, "decode_step_inputs: null pointer");
  if (rows < 0 || pool_rows <= 0 || width <= 0 || block_size <= 0 || prev_rows < 0 || (prev_rows > 0 && !prev_ids))
    return fail(MI355_EINVAL, "decode_step_inputs: bad shape");
  const int64_t threads = (int64_t)rows * (block_tables ? width : 1);
  hipLaunchKernelGGL(decode_inputs_kernel, dim3((unsigned)ceil_div64(threads, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), token_ids, positions, slots, context_lens, block_tables,
                     packed, table_pool, prev_ids, prev_rows, rows, pool_rows, width, block_size,
                     ilog2_exact(block_size));
  return check_launch("decode_inputs_kernel");
}

extern "C" int mi355_kv_write_slots(void* k_cache, void* v_cache, const void* k, const void* v,
                                    const int64_t* slots, int32_t T, int32_t Hkv, int32_t D,
                                    int64_t src_stride_tok, int32_t block_size, int32_t num_blocks,
                                    int64_t kv_stride_block, int64_t kv_stride_head, int64_t kv_stride_tok,
                                    int32_t dtype, void* stream) {
  if (T == 0) return 0;
  if (!k_cache || !v_cache || !k || !v || !slots) return fail(MI355_EINVAL, "kv_write_slots: null pointer");
  if (dtype != MI355_F32 && dtype != MI355_BF16 && dtype != MI355_F16)
    return fail(MI355_EINVAL, "kv_write_slots: dtype %d", dtype);
  const int ebytes = dtype == MI355_F32 ? 4 : 2, epl = 16 / ebytes;
  if (T < 0 || Hkv <= 0 || D <= 0 || D % epl || block_size <= 0 || num_blocks <= 0)
    return fail(MI355_EINVAL, "kv_write_slots: bad shape (T=%d Hkv=%d D=%d)", T, Hkv, D);
  if (src_stride_tok % epl || kv_stride_block % epl || kv_stride_head % epl || kv_stride_tok % epl ||
      src_stride_tok < (int64_t)Hkv * D)
    return fail(MI355_EINVAL, "kv_write_slots: strides must keep head rows 16-byte aligned");
  if (((uintptr_t)k_cache | (uintptr_t)v_cache | (uintptr_t)k | (uintptr_t)v) & 15)
    return fail(MI355_EINVAL, "kv_write_slots: tensors must be 16-byte aligned");
  KvWriteParams p;
  p.kc = static_cast<char*>(k_cache);
  p.vc = static_cast<char*>(v_cache);
  p.k = static_cast<const char*>(k);
  p.v = static_cast<const char*>(v);
  p.slots = slots;
  p.src_stride_tok = src_stride_tok;
  p.s_blk = kv_stride_block;
  p.s_head = kv_stride_head;
  p.s_tok = kv_stride_tok;
  p.T = T;
  p.Hkv = Hkv;
  p.lpt = D / epl;
  p.block_size = block_size;
  p.bs_shift = ilog2_exact(block_size);
  p.num_slots = (int64_t)num_blocks * block_size;
  p.ebytes = ebytes;
  const int64_t threads = (int64_t)T * Hkv * p.lpt;
  hipLaunchKernelGGL(kv_write_slots_kernel, dim3((unsigned)ceil_div64(threads, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  return check_launch("kv_write_slots_kernel");
}

extern "C" int mi355_slot_mapping(int64_t* out_slots, const int32_t* block_tables, const int32_t* starts,
                                  const int32_t* cu_tokens, int32_t B, int32_t max_blocks,
                                  int32_t block_size, int32_t total_tokens, void* stream) {
  if (total_tokens == 0 || B == 0) return 0;
  if (!out_slots || !block_tables || !starts || !cu_tokens)
    return fail(MI355_EINVAL, "slot_mapping: null pointer");
  if (B < 0 || total_tokens < 0 || max_blocks <= 0 || block_size <= 0)
    return fail(MI355_EINVAL, "slot_mapping: bad shape");
  hipLaunchKernelGGL(slot_mapping_kernel, dim3((unsigned)ceil_div64(total_tokens, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), out_slots, block_tables, starts, cu_tokens, B,
                     max_blocks, block_size, ilog2_exact(block_size), total_tokens);
  return check_launch("slot_mapping_kernel");
}
