// Causal prompt attention on the matrix cores (bf16 / fp16, head_dim 64 or 128).
//
// The prompt's QK^T and PV products are real dense contractions (Tq x Tk x D per
// head), so unlike decode they belong on MFMA.  Flash-attention structure:
//
//   workgroup = 4 or 8 waves = 128 or 256 consecutive query rows of one (sequence, query head);
//   wave      = 32 query rows; K/V are walked in 64-key tiles shared by the waves
//               through LDS (register-staged: the next tile's global loads are issued
//               before the current tile is reduced).
//   S^T = K . Q^T  with v_mfma_f32_32x32x16: the 32x32 result has the QUERY on the lane
//               (column) and 16 keys in the lane's registers, so the softmax row
//               statistics are lane-local plus one exchange with lane^32, and the
//               accumulator registers, converted to 16-bit in place, are already the
//               B operand of the next product (no LDS round trip for P).
//   O^T += V^T . P^T: the A operand V^T comes from a row-major V tile in LDS through
//               ds_read_b64_tr_b16 (transposing read).
//   O^T keeps the query on the lane too, so the online-softmax rescale of O is a
//   per-lane scalar multiply.
//
// Two addressing modes share the kernel: packed (K/V rows of the prompt itself, [T, Hkv, D]) and
// paged (query tokens of a later chunk, or of a prompt behind a cached prefix, attending to the
// block cache through the sequence's block table; the chunk's own K/V rows are already in it).
//
// Numerics: fp32 scores / statistics / accumulation; P is rounded to the 16-bit
// input type before the PV product (as every flash-attention kernel does).
#include "common.h"

namespace mi355 {
namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr int kQW = 32;              // query rows per wave
constexpr int kKT = 64;              // keys per tile

struct PrefillParams {
  void* out;
  const void* q;
  const void* k;
  const void* v;
  const int32_t* cu;                   // packed mode: sequence boundaries [nseq + 1]
  // paged mode: K/V are block caches read through a table; a segment is seg_len new query tokens
  // whose first one sits at position seg_pos0 of its sequence (positions before it are cached)
  const int32_t* bt;                   // [nseq, max_blocks]
  const int32_t* seg_q0;
  const int32_t* seg_len;
  const int32_t* seg_pos0;
  int64_t s_blk, s_head, s_tok;        // cache strides in elements
  int max_blocks, block_size, bs_shift, num_blocks;
  int max_ctx;                         // paged mode: no key at or past this position is read (<= max_blocks * block_size)
  int64_t q_stride_tok, kv_stride_tok;
  int nseq, n_qt_max, max_seqlen, Hq, Hkv, total_tokens;
  float scale_log2e;
};

template <class Tag>
struct Mma;
template <>
struct Mma<BF16> {
  static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                   c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));   // one v_cvt_pk_bf16_f32
  }
};
template <>
struct Mma<F16> {
  static __device__ __forceinline__ f32x16 run(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c,
                                                  0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    f16x2_t h;
    h[0] = (_Float16)lo;
    h[1] = (_Float16)hi;
    return __builtin_bit_cast(uint32_t, h);
  }
};

// Packed f32 arithmetic (v_pk_fma_f32, v_pk_add_f32, v_pk_mul_f32) issued beside MFMAs stalls the
// SIMD far longer than the two scalar instructions it replaces, so this kernel is compiled without
// it: the softmax arithmetic stays single-issue VALU that fits between matrix instructions.
#if defined(__HIP_DEVICE_COMPILE__)
#define MI355_NO_PACKED_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define MI355_NO_PACKED_F32
#endif

// The HIP header's __syncthreads / __shfl_xor / __all are ordinary functions compiled with the
// default feature set and would stay out of line in a kernel compiled without packed f32;
// these are the same operations on builtins.
__device__ __forceinline__ void wg_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// Both values of the lane pair (c, c+32), in every lane of the pair: one v_permlane32_swap
// (it exchanges the upper 32 lanes of its first register with the lower 32 of its second).
// Written as asm because this compiler's builtin returns the first result for both; the s_nop
// covers the wait states the instruction needs after a VALU write of its operands.
__device__ __forceinline__ void pair_values(float x, float& lo, float& hi) {
  float a = x, b = x;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  lo = a;
  hi = b;
}

// Two workgroups (8 waves) per CU: with a 256-register budget the 64 accumulator registers of a
// wave stay in VGPRs, where the softmax arithmetic reads and writes them directly (a tighter
// budget parks them in AccVGPRs and pays a copy per register per use).
template <class Tag, int D, int kWaves, bool PAGED>
__global__ void __launch_bounds__(kWaves * kWave, 2) MI355_NO_PACKED_F32 prefill_mfma_kernel(const PrefillParams p) {
  constexpr int kQT = kQW * kWaves;   // query rows per workgroup
  constexpr int KS = D / 16;          // k-steps of the S^T product (over head_dim)
  constexpr int DT = D / 32;          // 32-row tiles of O^T (over head_dim)
  constexpr int ROWB = D * 2;         // bytes per K row in LDS (chunks XOR-swizzled)
  // LDS banking (64 banks of 4 B).  A 16-byte read is served 16 lanes at a time, and those 16 lanes
  // hold keys c with 16 different values of c mod 16; XOR-ing the chunk index with the key's index
  // within a 256-byte span of rows puts them on 16 different 4-bank groups.  A transposing 8-byte
  // read is served 32 lanes at a time = 4 rows x (2 x 4) column pieces of 2 banks; a row pitch of
  // 48 or 80 banks (= +/-16 mod 64) makes the four rows tile all 64 banks.
  constexpr int ROWBV = D * 2 + 64;   // V row pitch
  constexpr int CPR = D / 8;          // 16-byte chunks per row
  constexpr int LOADS = kKT * CPR / (kWaves * kWave);  // 16-byte chunks per thread per tile (K, and again V)
  __shared__ __attribute__((aligned(16))) unsigned char sK2[2][kKT * ROWB];   // double-buffered tiles:
  __shared__ __attribute__((aligned(16))) unsigned char sV2[2][kKT * ROWBV];  // one barrier per tile

  auto kswz = [](int key) __attribute__((always_inline)) { return (key / (256 / ROWB)) & (CPR - 1); };
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
  // nq query rows starting at row q_row0 of q / out; query i sits at position pos0 + i and sees
  // keys 0 .. pos0 + i of the len = pos0 + nq keys
  const int q_row0 = PAGED ? p.seg_q0[seq] : p.cu[seq];
  // paged mode: segment descriptors come from device memory; whatever they hold, no key at or past
  // max_ctx (which the table covers) is read, and queries at or past it are neither read nor written
  const int pos0 = PAGED ? min(max(p.seg_pos0[seq], 0), p.max_ctx) : 0;
  const int nq = PAGED ? min(min(p.seg_len[seq], p.max_seqlen), p.max_ctx - pos0) : p.cu[seq + 1] - q_row0;
  const int seq_start = q_row0;                         // packed mode: K/V rows are the sequence's own
  const int len = pos0 + nq;
  const int q0 = qt * kQT;
  if (q0 >= nq) return;
  const int kvh = hq / (p.Hq / p.Hkv);
  const int qw0 = pos0 + q0 + wave * kQW;               // position of this wave's first query
  const int my_q = qw0 + c;                             // this lane's query position (may be >= len)

  // Q fragments (B operand of S^T = K.Q^T): lane (c,h), k-step s holds Q[c][16s + 8h .. +8).
  u32x4 qf[KS];
  {
    const int row = q_row0 + min(my_q, len - 1) - pos0;
    const char* src = static_cast<const char*>(p.q) + ((int64_t)row * p.q_stride_tok + (int64_t)hq * D) * 2;
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = *reinterpret_cast<const u32x4*>(src + (16 * s + 8 * h) * 2);
  }

  f32x16 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  const int kv_end = min(len, pos0 + q0 + kQT);         // causal: no key past the tile's last query
  const int ntiles = (kv_end + kKT - 1) / kKT;
  const char* kbase = static_cast<const char*>(p.k);
  const char* vbase = static_cast<const char*>(p.v);

  // A thread always stages the same (key, chunk) positions of a tile.  Packed mode: a load address
  // is a wave-uniform tile base plus a per-thread 32-bit offset that only changes in the
  // sequence's last, partial tile (rows past the end re-read the last row; they are masked
  // below).  Paged mode: the key's block id comes from the table (fetched one tile ahead, so the
  // dependent K/V loads do not wait for it) and the address is block * stride + row in block.
  u32x4 kreg[LOADS], vreg[LOADS];
  const uint32_t rowbytes = (uint32_t)p.kv_stride_tok * 2u;
  uint32_t voff[LOADS];
  int32_t blk[LOADS];
  const int32_t* bt_row = PAGED ? p.bt + (int64_t)seq * p.max_blocks : nullptr;
#pragma unroll
  for (int i = 0; i < LOADS; ++i) {
    const int ch = tid + i * (kWaves * kWave);
    voff[i] = PAGED ? (uint32_t)((int64_t)kvh * p.s_head + (ch % CPR) * 8) * 2u
                    : (uint32_t)(ch / CPR) * rowbytes + (uint32_t)(kvh * D + (ch % CPR) * 8) * 2u;
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
      const uint32_t tokbytes = (uint32_t)p.s_tok * 2u;
      const int64_t blkbytes = p.s_blk * 2;
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
      if (last >= kKT - 1) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
          kreg[i] = *reinterpret_cast<const u32x4*>(kb + voff[i]);
          vreg[i] = *reinterpret_cast<const u32x4*>(vb + voff[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
          const int ch = tid + i * (kWaves * kWave);
          const uint32_t off = (uint32_t)min(ch / CPR, last) * rowbytes + (uint32_t)(kvh * D + (ch % CPR) * 8) * 2u;
          kreg[i] = *reinterpret_cast<const u32x4*>(kb + off);
          vreg[i] = *reinterpret_cast<const u32x4*>(vb + off);
        }
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
      *reinterpret_cast<u32x4*>(sK + key * ROWB + ((dc ^ kswz(key)) << 4)) = kreg[i];   // XOR-swizzled chunks
      *reinterpret_cast<u32x4*>(sV + key * ROWBV + (dc << 4)) = vreg[i];
    }
  };

  // Tile t is reduced from buffer t&1 while tile t+1 sits in registers; at the end of the step
  // those registers go to the other buffer (last read in step t-1, which every wave has left)
  // and the loads of tile t+2 are issued.
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

    // ---- S^T = K . Q^T for the two 32-key halves of the tile.  The K operands are fetched
    // kAhead matrix instructions ahead of their use and the order is pinned: left alone, the
    // compiler keeps only one or two 16-byte reads in flight and every MFMA waits out most of an
    // LDS round trip.
    f32x16 st[2];
    {
      constexpr int NA = 2 * KS, kAhead = 4;
      auto a_at = [&](int n) __attribute__((always_inline)) {
        const int row = 32 * (n / KS) + c, ks = n % KS;
        return *reinterpret_cast<const u32x4*>(sK + row * ROWB + (((2 * ks + h) ^ kswz(row)) << 4));
      };
      u32x4 abuf[kAhead];
#pragma unroll
      for (int n = 0; n < kAhead; ++n) abuf[n] = a_at(n);
#pragma unroll
      for (int n = 0; n < NA; ++n) {
        const u32x4 a = abuf[n % kAhead];
        if (n + kAhead < NA) abuf[n % kAhead] = a_at(n + kAhead);
        const int kt = n / KS;
        st[kt] = Mma<Tag>::run(a, qf[n % KS], n % KS == 0 ? f32x16(0.f) : st[kt]);   // first step accumulates onto the constant 0
      }
      __builtin_amdgcn_sched_group_barrier(0x100, kAhead, 0);
#pragma unroll
      for (int n = 0; n < NA - kAhead; ++n) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, kAhead, 0);
    }
    // ---- causal mask, online softmax (the lane's query is column c).  The scale is positive, so
    // the row maximum is taken on the raw scores and scale and shift fold into one fma per score.
    const bool diag = kv0 + kKT - 1 > qw0;           // some key of the tile may exceed some query of the wave
    if (diag) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h;
          st[kt][r] = key <= my_q ? st[kt][r] : -__builtin_inff();
        }
    }
    float mx = fmaxf(st[0][0], st[1][0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = __builtin_fmaxf(__builtin_fmaxf(mx, st[0][r]), st[1][r]);   // v_max3_f32
    {
      float a, b;
      pair_values(mx, a, b);
      mx = __builtin_fmaxf(a, b);
    }
    // Deferred rescale: m_run is the reference the exponentials are taken against.  It moves (and
    // the accumulators shrink by f) only when some query's maximum has outgrown it by more than
    // 2^kDefer; until then weights may exceed 1 (they stay below 2^kDefer), which fp32 sums and
    // 16-bit weights hold without loss.  Row sums and outputs are relative to the same reference,
    // so the quotient at the end is the exact softmax.  On ordinary data that is the first tile of
    // a row and almost never again, instead of every tile in which any of the wave's 32 queries
    // meets a new maximum.
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
    f32x16 psum16;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      st[kt] = st[kt] * p.scale_log2e - m_run;
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = fast_exp2(st[kt][r]);
    }
    psum16 = st[0] + st[1];
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) psum += psum16[r];
    {
      float a, b;
      pair_values(psum, a, b);
      psum = a + b;
    }
    l_run += psum;

    // ---- O^T += V^T . P^T : k-step s4 covers keys 16*s4 .. 16*s4+15 of the tile.
    // Fragment element j of lane half h is key 16*s4 + 8*(j>>2) + 4*h + (j&3): exactly
    // registers 8*(s4&1) .. +7 of st[s4>>1] for P, and two transposing 4-key reads for V.
    const int grp = (lane >> 4) & 1, qi = (lane & 15) >> 2, pi = lane & 3;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int kt = s4 >> 1, r0 = 8 * (s4 & 1);
      u32x4 pb;
#pragma unroll
      for (int w = 0; w < 4; ++w) pb[w] = Mma<Tag>::pack2(st[kt][r0 + 2 * w], st[kt][r0 + 2 * w + 1]);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        // lane 4*qi+pi of each 16-lane group addresses row (key) base+qi, columns d0+4*pi..+3;
        // it receives column (lane&15) of the 4 rows = V[base+0..3][d0 + (lane&15)].
        const int d0 = 32 * dt + 16 * grp;
        const int key_lo = 16 * s4 + 4 * h + qi;
        const unsigned char* a0 = sV + key_lo * ROWBV + (d0 + 4 * pi) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(a0 + 8 * ROWBV));
        const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
        const u32x4 va = {lo2[0], lo2[1], hi2[0], hi2[1]};
        o[dt] = Mma<Tag>::run(va, pb, o[dt]);
      }
    }
    }
    if (t + 1 < ntiles) {
      write_lds((t + 1) & 1);
      if (t + 2 < ntiles) issue_loads(t + 2);
    }
    wg_barrier();
  }

  // ---- normalise and store: lane (c,h) holds O[my_q][32*dt + 8*(r>>2) + 4*h + (r&3)]
  if (my_q < len) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    char* dst = static_cast<char*>(p.out) + ((int64_t)(q_row0 + my_q - pos0) * p.Hq + hq) * D * 2;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d0 = 32 * dt + 8 * rg + 4 * h;
        u32x2 w;
        w[0] = Mma<Tag>::pack2(o[dt][4 * rg] * inv, o[dt][4 * rg + 1] * inv);
        w[1] = Mma<Tag>::pack2(o[dt][4 * rg + 2] * inv, o[dt][4 * rg + 3] * inv);
        *reinterpret_cast<u32x2*>(dst + d0 * 2) = w;
      }
  }
}

template <class Tag, int D, int kWaves, bool PAGED>
int launch_w(PrefillParams p, hipStream_t s) {
  p.n_qt_max = (p.max_seqlen + kQW * kWaves - 1) / (kQW * kWaves);
  const int64_t grid = (int64_t)p.nseq * p.n_qt_max * p.Hq;
  if (grid >= (1LL << 31)) return fail(MI355_EINVAL, "prefill: grid too large");
  hipLaunchKernelGGL((prefill_mfma_kernel<Tag, D, kWaves, PAGED>), dim3((unsigned)grid), dim3(kWaves * kWave), 0, s,
                     p);
  return check_launch("prefill_mfma_kernel");
}

// 256-row workgroups stage every K/V tile once for twice the queries (half the LDS writes and
// global loads per product), but their waves idle more on the diagonal and there are half as
// many of them: they take over for long contexts once the launch still fills the chip.
// max_context = the longest key range a query tile walks.
template <class Tag, int D, bool PAGED>
int launch(const PrefillParams& p, int max_context, hipStream_t s) {
  const int force = knob(kKnobPrefillWaves);
  const int64_t wgs8 = (int64_t)p.nseq * ((p.max_seqlen + 8 * kQW - 1) / (8 * kQW)) * p.Hq;
  const bool big = force ? force == 8 : (max_context >= 4096 && p.max_seqlen >= 8 * kQW && wgs8 >= 2 * 256);
  return big ? launch_w<Tag, D, 8, PAGED>(p, s) : launch_w<Tag, D, 4, PAGED>(p, s);
}

}  // namespace

// Returns 1 when this kernel family covers (dtype, D); the caller then must not fall back.
bool prefill_mfma_supported(int dtype, int D) {
  return (dtype == MI355_BF16 || dtype == MI355_F16) && (D == 64 || D == 128);
}

int prefill_mfma_launch(void* out, const void* q, const void* k, const void* v, int64_t q_stride_tok,
                        int64_t kv_stride_tok, const int32_t* cu_seqlens, int nseq, int total_tokens,
                        int max_seqlen, int Hq, int Hkv, int D, float scale, int dtype, hipStream_t stream) {
  PrefillParams p;
  p.out = out;
  p.q = q;
  p.k = k;
  p.v = v;
  p.cu = cu_seqlens;
  p.q_stride_tok = q_stride_tok;
  p.kv_stride_tok = kv_stride_tok;
  p.nseq = nseq;
  p.max_seqlen = max_seqlen;
  p.Hq = Hq;
  p.Hkv = Hkv;
  p.total_tokens = total_tokens;
  p.scale_log2e = scale * 1.4426950408889634f;
  p.bt = nullptr;
  p.seg_q0 = p.seg_len = p.seg_pos0 = nullptr;
  p.s_blk = p.s_head = p.s_tok = 0;
  p.max_blocks = p.block_size = p.num_blocks = p.max_ctx = 0;
  p.bs_shift = -1;
  if (dtype == MI355_BF16)
    return D == 64 ? launch<BF16, 64, false>(p, max_seqlen, stream) : launch<BF16, 128, false>(p, max_seqlen, stream);
  return D == 64 ? launch<F16, 64, false>(p, max_seqlen, stream) : launch<F16, 128, false>(p, max_seqlen, stream);
}

// Paged mode: nseg segments of new query tokens (rows seg_q0[s] .. +seg_len[s]-1 of q / out, first
// token at position seg_pos0[s]) over block caches with the given element strides.
int prefill_mfma_paged_launch(void* out, const void* q, int64_t q_stride_tok, const void* k_cache,
                              const void* v_cache, const int32_t* block_tables, const int32_t* seg_q0,
                              const int32_t* seg_len, const int32_t* seg_pos0, int nseg, int max_q_len, int Hq,
                              int Hkv, int D, int block_size, int max_blocks, int num_blocks, int max_context,
                              int64_t s_blk, int64_t s_head, int64_t s_tok, float scale, int dtype,
                              hipStream_t stream) {
  PrefillParams p;
  p.num_blocks = num_blocks;
  p.out = out;
  p.q = q;
  p.k = k_cache;
  p.v = v_cache;
  p.cu = nullptr;
  p.bt = block_tables;
  p.seg_q0 = seg_q0;
  p.seg_len = seg_len;
  p.seg_pos0 = seg_pos0;
  p.s_blk = s_blk;
  p.s_head = s_head;
  p.s_tok = s_tok;
  p.max_blocks = max_blocks;
  p.block_size = block_size;
  p.max_ctx = (int)((int64_t)max_context < (int64_t)max_blocks * block_size ? (int64_t)max_context
                                                                             : (int64_t)max_blocks * block_size);
  p.bs_shift = ilog2_exact(block_size);
  p.q_stride_tok = q_stride_tok;
  p.kv_stride_tok = s_tok;
  p.nseq = nseg;
  p.max_seqlen = max_q_len;
  p.Hq = Hq;
  p.Hkv = Hkv;
  p.total_tokens = 0;
  p.scale_log2e = scale * 1.4426950408889634f;
  if (dtype == MI355_BF16)
    return D == 64 ? launch<BF16, 64, true>(p, max_context, stream) : launch<BF16, 128, true>(p, max_context, stream);
  return D == 64 ? launch<F16, 64, true>(p, max_context, stream) : launch<F16, 128, true>(p, max_context, stream);
}

}  // namespace mi355
