// Causal prompt attention in fp32 on the matrix cores (head_dim 32, 64 or 128).
//
// The reference engine computes in fp32 by default (05_mini_vllm/mini_vllm.py:437-458 is the
// prompt branch; its tests compare at 1e-5), so this kernel keeps fp32 operands: the matrix
// instruction v_mfma_f32_32x32x2_f32 multiplies fp32 values exactly and accumulates in fp32.
// Same flash-attention structure as the 16-bit kernel (prefill_mfma.hip):
//
//   workgroup = 4 waves = 128 consecutive query rows of one (sequence, query head);
//   wave      = 32 query rows; K/V walked in tiles of kKT keys shared by the waves through
//               double-buffered LDS (register-staged: the next tile's global loads are issued
//               before the current tile is reduced; one barrier per tile).
//   S^T = K . Q^T   one 32 x 32 tile per 32 keys; a matrix instruction contracts 2 head dims.
//               Lane (c, h) = (lane & 31, lane >> 5) supplies K[key c][d] as the A operand and
//               Q[query c][d] as the B operand for d = 8j + 4h + e: one 16-byte LDS read (j) serves
//               four instructions (e), and the query fragments are plain 16-byte global loads.
//               The result has the QUERY on the lane and 16 keys in its registers, so softmax
//               row statistics are lane-local plus one exchange with lane ^ 32.
//   O^T += V^T . P^T   accumulator register r of lane half h holds key (r & 3) + 8 (r >> 2) + 4 h:
//               taken as it is, register r is the B operand of the instruction that contracts that
//               pair of keys (h = 0 / 1) -- P never moves.  The A operand V^T[d][key] is one 4-byte
//               LDS read per instruction: 32 lanes read 32 consecutive dims of one key row, the
//               other 32 those of the row four keys further; a row pitch of head_dim + 8 words puts
//               the two rows on disjoint halves of the 64 banks.
//
// Two addressing modes share the kernel: packed (K/V rows of the prompt itself) and paged (query
// tokens of a chunk attending to the block cache through the sequence's block table).
#include "common.h"

namespace mi355 {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kQW = 32;              // query rows per wave
constexpr int kWaves = 4;            // waves per workgroup
constexpr int kQT = kQW * kWaves;    // query rows per workgroup

struct PrefillF32MmaParams {
  float* out;
  const float* q;
  const float* k;
  const float* v;
  const int32_t* cu;                   // packed mode: sequence boundaries [nseq + 1]
  const int32_t* bt;                   // paged mode: [nseq, max_blocks]
  const int32_t* seg_q0;
  const int32_t* seg_len;
  const int32_t* seg_pos0;
  int64_t s_blk, s_head, s_tok;        // cache strides in elements
  int max_blocks, block_size, bs_shift, num_blocks, max_ctx;
  int64_t q_stride_tok, kv_stride_tok;
  int nseq, n_qt_max, max_seqlen, Hq, Hkv;
  float scale_log2e;
};

#if defined(__HIP_DEVICE_COMPILE__)
#define MI355_NO_PACKED_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define MI355_NO_PACKED_F32
#endif

__device__ __forceinline__ void wg_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// Both values of the lane pair (c, c + 32), in every lane of the pair.
__device__ __forceinline__ void pair_values(float x, float& lo, float& hi) {
  float a = x, b = x;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  lo = a;
  hi = b;
}
__device__ __forceinline__ f32x16 mma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// kKT keys per tile: 64 at head_dim <= 64, 32 at head_dim 128 (two workgroups per CU either way).
template <int D, bool PAGED>
__global__ void __launch_bounds__(kWaves * kWave, 2) MI355_NO_PACKED_F32
prefill_mfma_f32_kernel(const PrefillF32MmaParams p) {
  constexpr int kKT = D <= 64 ? 64 : 32;
  constexpr int HALVES = kKT / 32;
  constexpr int DT = D / 32;            // 32-row tiles of O^T (over head_dim)
  constexpr int ROWB = D * 4;           // bytes per K row in LDS (16-byte chunks XOR-swizzled)
  constexpr int ROWBV = (D + 8) * 4;    // V row pitch: + 8 words (see the header)
  constexpr int CPR = D / 4;            // 16-byte chunks per row
  constexpr int LOADS = kKT * CPR / (kWaves * kWave);
  static_assert(LOADS >= 1 && kKT * CPR % (kWaves * kWave) == 0, "tile does not split over the threads");
  __shared__ __attribute__((aligned(16))) unsigned char sK2[2][kKT * ROWB];
  __shared__ __attribute__((aligned(16))) unsigned char sV2[2][kKT * ROWBV];

  // 16-byte reads are served 16 lanes at a time; those lanes hold 16 consecutive keys and the same
  // chunk index, so XOR-ing the chunk index with the key's index within a 256-byte span of rows
  // puts them on 16 different 4-bank groups
  auto kswz = [](int key) __attribute__((always_inline)) {
    return (ROWB >= 256 ? key : key / (256 / (ROWB >= 256 ? 256 : ROWB))) & (CPR < 16 ? CPR - 1 : 15);
  };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  int bid = blockIdx.x;
  // Workgroups b and b + 8 share an XCD (and its L2).  With the head index taken as b % Hq, the
  // query heads of one kv group (consecutive heads) would sit on different XCDs and each K/V tile
  // would be fetched into several L2s; slot b % 8 takes Hq / 8 consecutive heads instead, so a kv
  // group's heads -- and all query tiles of a head -- share one L2.
  const int hslot = bid % p.Hq;
  const int hq = (p.Hq & 7) == 0 ? (hslot & 7) * (p.Hq >> 3) + (hslot >> 3) : hslot;
  bid /= p.Hq;
  const int qt = p.n_qt_max - 1 - (bid % p.n_qt_max);   // long (late) tiles are issued first
  const int seq = bid / p.n_qt_max;
  const int q_row0 = PAGED ? p.seg_q0[seq] : p.cu[seq];
  // paged mode: descriptors come from device memory and are clamped to the host-side bounds
  const int pos0 = PAGED ? min(max(p.seg_pos0[seq], 0), p.max_ctx) : 0;
  const int nq = PAGED ? min(min(p.seg_len[seq], p.max_seqlen), p.max_ctx - pos0) : p.cu[seq + 1] - q_row0;
  const int seq_start = q_row0;
  const int len = pos0 + nq;
  const int q0 = qt * kQT;
  if (q0 >= nq) return;
  const int kvh = hq / (p.Hq / p.Hkv);
  const int qw0 = pos0 + q0 + wave * kQW;               // position of this wave's first query
  const int my_q = qw0 + c;                             // this lane's query position (may be >= len)

  // Q fragments: lane (c, h) holds Q[query c][8 j + 4 h + e], e = 0..3, in qf[j][e]
  f32x4 qf[D / 8];
  {
    const int row = q_row0 + min(my_q, len - 1) - pos0;
    const float* src = p.q + (int64_t)row * p.q_stride_tok + (int64_t)hq * D;
#pragma unroll
    for (int j = 0; j < D / 8; ++j) qf[j] = *reinterpret_cast<const f32x4*>(src + 8 * j + 4 * h);
  }

  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  const int kv_end = min(len, pos0 + q0 + kQT);         // causal: no key past the tile's last query
  const int ntiles = (kv_end + kKT - 1) / kKT;
  const char* kbase = reinterpret_cast<const char*>(p.k);
  const char* vbase = reinterpret_cast<const char*>(p.v);

  u32x4 kreg[LOADS], vreg[LOADS];
  const uint32_t rowbytes = (uint32_t)p.kv_stride_tok * 4u;
  uint32_t voff[LOADS];
  int32_t blk[LOADS];
  const int32_t* bt_row = PAGED ? p.bt + (int64_t)seq * p.max_blocks : nullptr;
#pragma unroll
  for (int i = 0; i < LOADS; ++i) {
    const int ch = tid + i * (kWaves * kWave);
    voff[i] = PAGED ? (uint32_t)((int64_t)kvh * p.s_head + (ch % CPR) * 4) * 4u
                    : (uint32_t)(ch / CPR) * rowbytes + (uint32_t)(kvh * D + (ch % CPR) * 4) * 4u;
    blk[i] = 0;
  }
  auto block_of = [&](int key) __attribute__((always_inline)) {
    return p.bs_shift >= 0 ? key >> p.bs_shift : key / p.block_size;
  };
  auto fetch_blocks = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int key = min(tile * kKT + (tid + i * (kWaves * kWave)) / CPR, len - 1);
      blk[i] = min(max(bt_row[block_of(key)], 0), p.num_blocks - 1);   // a bad id must not leave the pool
    }
  };
  auto issue_loads = [&](int tile) __attribute__((always_inline)) {
    if constexpr (PAGED) {
      const uint32_t tokbytes = (uint32_t)p.s_tok * 4u;
      const int64_t blkbytes = p.s_blk * 4;
#pragma unroll
      for (int i = 0; i < LOADS; ++i) {
        const int key = min(tile * kKT + (tid + i * (kWaves * kWave)) / CPR, len - 1);
        const int within = key - block_of(key) * p.block_size;
        const int64_t off = (int64_t)blk[i] * blkbytes + (uint32_t)within * tokbytes + voff[i];
        kreg[i] = *reinterpret_cast<const u32x4*>(kbase + off);
        vreg[i] = *reinterpret_cast<const u32x4*>(vbase + off);
      }
      if ((tile + 1) * kKT < len) fetch_blocks(tile + 1);
    } else {
      const int64_t base = (int64_t)(seq_start + tile * kKT) * rowbytes;
      const char* kb = kbase + base;
      const char* vb = vbase + base;
      const int last = len - 1 - tile * kKT;                          // tile-relative index of the last valid row
#pragma unroll
      for (int i = 0; i < LOADS; ++i) {
        const int ch = tid + i * (kWaves * kWave);
        const uint32_t off = last >= kKT - 1 ? voff[i]
                                             : (uint32_t)min(ch / CPR, last) * rowbytes +
                                                   (uint32_t)(kvh * D + (ch % CPR) * 4) * 4u;
        kreg[i] = *reinterpret_cast<const u32x4*>(kb + off);
        vreg[i] = *reinterpret_cast<const u32x4*>(vb + off);
      }
    }
  };
  auto write_lds = [&](int buf) __attribute__((always_inline)) {
    unsigned char* sK = sK2[buf];
    unsigned char* sV = sV2[buf];
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
      const int ch = tid + i * (kWaves * kWave);
      const int key = ch / CPR, dc = ch % CPR;
      *reinterpret_cast<u32x4*>(sK + key * ROWB + ((dc ^ kswz(key)) << 4)) = kreg[i];
      *reinterpret_cast<u32x4*>(sV + key * ROWBV + (dc << 4)) = vreg[i];
    }
  };

  if constexpr (PAGED) fetch_blocks(0);
  issue_loads(0);
  write_lds(0);
  if (ntiles > 1) issue_loads(1);
  wg_barrier();
  for (int t = 0; t < ntiles; ++t) {
    const unsigned char* sK = sK2[t & 1];
    const unsigned char* sV = sV2[t & 1];
    const int kv0 = t * kKT;
    if (kv0 <= qw0 + kQW - 1) {          // else: tile entirely above this wave's diagonal (wave-uniform)
      // ---- S^T = K . Q^T per 32-key half
      f32x16 st[HALVES];
#pragma unroll
      for (int kt = 0; kt < HALVES; ++kt) {
        const int row = 32 * kt + c;
#pragma unroll
        for (int j = 0; j < D / 8; ++j) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(sK + row * ROWB + (((2 * j + h) ^ kswz(row)) << 4));
#pragma unroll
          for (int e = 0; e < 4; ++e) st[kt] = mma(a[e], qf[j][e], (j == 0 && e == 0) ? f32x16(0.f) : st[kt]);
        }
      }
      // ---- causal mask, online softmax (the lane's query is column c); scale > 0, so the row
      // maximum is taken on raw scores and scale and shift fold into one fma per score
      const bool diag = kv0 + kKT - 1 > qw0;
      if (diag) {
#pragma unroll
        for (int kt = 0; kt < HALVES; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = kv0 + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h;
            st[kt][r] = key <= my_q ? st[kt][r] : -__builtin_inff();
          }
      }
      float mx = st[0][0];
#pragma unroll
      for (int kt = 0; kt < HALVES; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = __builtin_fmaxf(mx, st[kt][r]);
      {
        float a, b;
        pair_values(mx, a, b);
        mx = __builtin_fmaxf(a, b);
      }
      // Deferred rescale: the reference the exponentials are taken against moves (and the
      // accumulators shrink) only when some query's maximum outgrows it by more than 2^kDefer;
      // until then weights may exceed 1 (below 2^kDefer), which fp32 sums hold without loss.
      constexpr float kDefer = 8.f;
      const float mxs = mx * p.scale_log2e;
      if (__builtin_amdgcn_ballot_w64(mxs > m_run + kDefer) != 0) {
        const float m_new = fmaxf(m_run, mxs);
        const float f = fast_exp2(m_run - m_new);
        m_run = m_new;
        l_run *= f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] *= f;
      }
      float psum = 0.f;
#pragma unroll
      for (int kt = 0; kt < HALVES; ++kt) {
        st[kt] = st[kt] * p.scale_log2e - m_run;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          st[kt][r] = fast_exp2(st[kt][r]);
          psum += st[kt][r];
        }
      }
      {
        float a, b;
        pair_values(psum, a, b);
        psum = a + b;
      }
      l_run += psum;

      // ---- O^T += V^T . P^T: instruction (kt, r) contracts keys 32 kt + (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
      for (int kt = 0; kt < HALVES; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h;
          const unsigned char* vrow = sV + key * ROWBV + c * 4;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
            o[dt] = mma(*reinterpret_cast<const float*>(vrow + dt * 128), st[kt][r], o[dt]);
        }
    }
    if (t + 1 < ntiles) {
      write_lds((t + 1) & 1);
      if (t + 2 < ntiles) issue_loads(t + 2);
    }
    wg_barrier();
  }

  // ---- normalise and store: lane (c, h) holds O[my_q][32 dt + 8 (r >> 2) + 4 h + (r & 3)]
  if (my_q < len) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* dst = p.out + ((int64_t)(q_row0 + my_q - pos0) * p.Hq + hq) * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const f32x4 w = {o[dt][4 * rg] * inv, o[dt][4 * rg + 1] * inv, o[dt][4 * rg + 2] * inv,
                         o[dt][4 * rg + 3] * inv};
        *reinterpret_cast<f32x4*>(dst + 32 * dt + 8 * rg + 4 * h) = w;
      }
  }
}

template <int D, bool PAGED>
int launch(PrefillF32MmaParams p, hipStream_t s) {
  p.n_qt_max = (p.max_seqlen + kQT - 1) / kQT;
  const int64_t grid = (int64_t)p.nseq * p.n_qt_max * p.Hq;
  if (grid >= (1LL << 31)) return fail(MI355_EINVAL, "prefill (fp32 matrix cores): grid too large");
  hipLaunchKernelGGL((prefill_mfma_f32_kernel<D, PAGED>), dim3((unsigned)grid), dim3(kWaves * kWave), 0, s, p);
  return check_launch("prefill_mfma_f32_kernel");
}

template <bool PAGED>
int dispatch(int D, const PrefillF32MmaParams& p, hipStream_t s) {
  switch (D) {
    case 32: return launch<32, PAGED>(p, s);
    case 64: return launch<64, PAGED>(p, s);
    case 128: return launch<128, PAGED>(p, s);
  }
  return fail(MI355_EUNSUPPORTED, "prefill (fp32 matrix cores): head_dim %d", D);
}

}  // namespace

bool prefill_mfma_f32_supported(int dtype, int D) { return dtype == MI355_F32 && (D == 32 || D == 64 || D == 128); }

int prefill_mfma_f32_launch(void* out, const void* q, const void* k, const void* v, int64_t q_stride_tok,
                            int64_t kv_stride_tok, const int32_t* cu_seqlens, int nseq, int max_seqlen, int Hq,
                            int Hkv, int D, float scale, hipStream_t stream) {
  PrefillF32MmaParams p = {};
  p.out = static_cast<float*>(out);
  p.q = static_cast<const float*>(q);
  p.k = static_cast<const float*>(k);
  p.v = static_cast<const float*>(v);
  p.cu = cu_seqlens;
  p.q_stride_tok = q_stride_tok;
  p.kv_stride_tok = kv_stride_tok;
  p.nseq = nseq;
  p.max_seqlen = max_seqlen;
  p.Hq = Hq;
  p.Hkv = Hkv;
  p.bs_shift = -1;
  p.scale_log2e = scale * 1.4426950408889634f;
  return dispatch<false>(D, p, stream);
}

int prefill_mfma_f32_paged_launch(void* out, const void* q, int64_t q_stride_tok, const void* k_cache,
                                  const void* v_cache, const int32_t* block_tables, const int32_t* seg_q0,
                                  const int32_t* seg_len, const int32_t* seg_pos0, int nseg, int max_q_len, int Hq,
                                  int Hkv, int D, int block_size, int max_blocks, int num_blocks, int max_context,
                                  int64_t s_blk, int64_t s_head, int64_t s_tok, float scale, hipStream_t stream) {
  PrefillF32MmaParams p = {};
  p.out = static_cast<float*>(out);
  p.q = static_cast<const float*>(q);
  p.k = static_cast<const float*>(k_cache);
  p.v = static_cast<const float*>(v_cache);
  p.bt = block_tables;
  p.seg_q0 = seg_q0;
  p.seg_len = seg_len;
  p.seg_pos0 = seg_pos0;
  p.s_blk = s_blk;
  p.s_head = s_head;
  p.s_tok = s_tok;
  p.max_blocks = max_blocks;
  p.block_size = block_size;
  p.bs_shift = ilog2_exact(block_size);
  p.num_blocks = num_blocks;
  p.max_ctx = (int)((int64_t)max_context < (int64_t)max_blocks * block_size ? (int64_t)max_context
                                                                             : (int64_t)max_blocks * block_size);
  p.q_stride_tok = q_stride_tok;
  p.kv_stride_tok = s_tok;
  p.nseq = nseg;
  p.max_seqlen = max_q_len;
  p.Hq = Hq;
  p.Hkv = Hkv;
  p.scale_log2e = scale * 1.4426950408889634f;
  return dispatch<true>(D, p, stream);
}

}  // namespace mi355
