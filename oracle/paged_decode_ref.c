/*
 * Plain-C restatement of the reference's paged decode attention, fp32.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY: loaded by tests (checked against the
 * golden vectors captured from the reference) and by the cpu_baseline leg of
 * bench.py.  The product library never links or calls it.
 *
 * Algorithm followed: 01_paged_attention/paged_attention.py:80-101 -- for every
 * kv head, gather the seq_len K and V rows through the block table
 * (row i lives in block block_table[i / block_size] at offset i % block_size),
 * scores = q . K^T / sqrt(d), softmax over the sequence, out = attn . V; query
 * heads kv_h*g .. (kv_h+1)*g-1 share kv head kv_h.
 *
 * Layout: k_cache, v_cache [num_blocks][num_kv_heads][block_size][head_dim];
 * query, out [batch][num_q_heads][head_dim]; block_tables [batch][max_blocks];
 * context_lens [batch].  Rows with context length 0 produce zeros.
 * Parallelised over (sequence, kv head) with OpenMP when built with -fopenmp.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_paged_decode_f32(float* out, const float* query, const float* k_cache, const float* v_cache,
                             const int32_t* block_tables, const int32_t* context_lens, int batch,
                             int num_q_heads, int num_kv_heads, int head_dim, int block_size,
                             int max_blocks) {
  const int group = num_q_heads / num_kv_heads;
  const float scale = 1.0f / sqrtf((float)head_dim);
  const size_t head_stride = (size_t)block_size * head_dim;
  const size_t block_stride = (size_t)num_kv_heads * head_stride;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int b = 0; b < batch; ++b) {
    for (int kvh = 0; kvh < num_kv_heads; ++kvh) {
      const int n = context_lens[b];
      float* o = out + ((size_t)b * num_q_heads + (size_t)kvh * group) * head_dim;
      memset(o, 0, sizeof(float) * (size_t)group * head_dim);
      if (n <= 0) continue;
      float* scores = (float*)malloc(sizeof(float) * (size_t)n);
      const int32_t* table = block_tables + (size_t)b * max_blocks;
      for (int g = 0; g < group; ++g) {
        const float* q = query + ((size_t)b * num_q_heads + (size_t)kvh * group + g) * head_dim;
        float top = -INFINITY;
        for (int i = 0; i < n; ++i) {
          const float* krow = k_cache + (size_t)table[i / block_size] * block_stride + kvh * head_stride +
                              (size_t)(i % block_size) * head_dim;
          float dot = 0.0f;
          for (int d = 0; d < head_dim; ++d) dot += q[d] * krow[d];
          scores[i] = dot * scale;
          if (scores[i] > top) top = scores[i];
        }
        float denom = 0.0f;
        for (int i = 0; i < n; ++i) {
          scores[i] = expf(scores[i] - top);
          denom += scores[i];
        }
        float* og = o + (size_t)g * head_dim;
        for (int i = 0; i < n; ++i) {
          const float w = scores[i] / denom;
          const float* vrow = v_cache + (size_t)table[i / block_size] * block_stride + kvh * head_stride +
                              (size_t)(i % block_size) * head_dim;
          for (int d = 0; d < head_dim; ++d) og[d] += w * vrow[d];
        }
      }
      free(scores);
    }
  }
}
