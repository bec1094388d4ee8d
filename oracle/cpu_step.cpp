// CPU restatement of one training step of the hot path in C++ / OpenMP (fp32): the "build's own C++ CPU restatement"
// of SURVEY 8(d)(ii), timed by bench.py's `cpu_baseline` leg on the GPU box's host cores.
//
// TEST / MEASUREMENT INFRASTRUCTURE ONLY (see oracle/model_np.py header): only tests/, __graft_entry__ and bench.py's
// cpu_baseline leg may load this; the product path never does.  Pinned against the numpy oracle (oracle/model_np.py, itself
// pinned to the reference's own outputs) by tests/test_cpu_step.py: 4 losses, every named gradient, one clip + AdamW step.
//
// `model.py` = /root/reference/notebooks/Training/transformer.model.py, `train.py` = .../transformer.py.  Pretraining
// forward + backward on an already-masked batch (model.py:417-462 is applied by the caller), no LoRA.  The arithmetic follows
// the reference's modules one to one; what differs from the numpy oracle is only HOW it is evaluated: a blocked SGEMM (AVX2 or
// AVX-512 micro kernel, picked at run time) on all cores, attention restricted to each query block's candidate keys (the packed users of a row) with the soft-max
// recomputed in the backward from its log-sum-exp instead of a stored (B, H, T, T) probability tensor.
#include <immintrin.h>
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------------ SGEMM
// C[M,N] (+)= A[M,K] . B[K,N]; element (i,k) of A at A[i*rsa + k*csa], (k,j) of B at B[k*rsb + j*csb].
constexpr int KC = 256;

// micro kernels: C[MR x NR] (+)= Ap[kc][MR] . Bp[kc][NR] on packed panels.  AVX2: 6 x 16 (12 ymm accumulators); AVX-512: 12 x 32
// (24 zmm accumulators), chosen at run time so that the library built in the CPU container uses what the host offers.
struct KernAvx2 {
  static constexpr int MR = 6, NR = 16;
  static void micro(int kc, const float* __restrict ap, const float* __restrict bp, float* __restrict c, int ldc, int mr, int nr, bool acc) {
    __m256 r[MR][2];
#pragma GCC unroll 6
    for (int i = 0; i < MR; ++i) { r[i][0] = _mm256_setzero_ps(); r[i][1] = _mm256_setzero_ps(); }
    for (int k = 0; k < kc; ++k) {
      const __m256 b0 = _mm256_load_ps(bp + k * NR), b1 = _mm256_load_ps(bp + k * NR + 8);
      const float* a = ap + k * MR;
#pragma GCC unroll 6
      for (int i = 0; i < MR; ++i) {
        const __m256 av = _mm256_broadcast_ss(a + i);
        r[i][0] = _mm256_fmadd_ps(av, b0, r[i][0]); r[i][1] = _mm256_fmadd_ps(av, b1, r[i][1]);
      }
    }
    if (mr == MR && nr == NR) {
#pragma GCC unroll 6
      for (int i = 0; i < MR; ++i) {
        float* ci = c + (int64_t)i * ldc;
        if (acc) { r[i][0] = _mm256_add_ps(r[i][0], _mm256_loadu_ps(ci)); r[i][1] = _mm256_add_ps(r[i][1], _mm256_loadu_ps(ci + 8)); }
        _mm256_storeu_ps(ci, r[i][0]); _mm256_storeu_ps(ci + 8, r[i][1]);
      }
      return;
    }
    alignas(64) float t[MR * NR];
#pragma GCC unroll 6
    for (int i = 0; i < MR; ++i) { _mm256_store_ps(t + i * NR, r[i][0]); _mm256_store_ps(t + i * NR + 8, r[i][1]); }
    for (int i = 0; i < mr; ++i)
      for (int j = 0; j < nr; ++j) c[(int64_t)i * ldc + j] = acc ? c[(int64_t)i * ldc + j] + t[i * NR + j] : t[i * NR + j];
  }
};
struct KernAvx512 {
  static constexpr int MR = 12, NR = 32;
  __attribute__((target("avx512f"))) static void micro(int kc, const float* __restrict ap, const float* __restrict bp, float* __restrict c,
                                                       int ldc, int mr, int nr, bool acc) {
    __m512 r[MR][2];
#pragma GCC unroll 12
    for (int i = 0; i < MR; ++i) { r[i][0] = _mm512_setzero_ps(); r[i][1] = _mm512_setzero_ps(); }
    for (int k = 0; k < kc; ++k) {
      const __m512 b0 = _mm512_load_ps(bp + k * NR), b1 = _mm512_load_ps(bp + k * NR + 16);
      const float* a = ap + k * MR;
#pragma GCC unroll 12
      for (int i = 0; i < MR; ++i) {
        const __m512 av = _mm512_set1_ps(a[i]);
        r[i][0] = _mm512_fmadd_ps(av, b0, r[i][0]); r[i][1] = _mm512_fmadd_ps(av, b1, r[i][1]);
      }
    }
    if (mr == MR && nr == NR) {
#pragma GCC unroll 12
      for (int i = 0; i < MR; ++i) {
        float* ci = c + (int64_t)i * ldc;
        if (acc) { r[i][0] = _mm512_add_ps(r[i][0], _mm512_loadu_ps(ci)); r[i][1] = _mm512_add_ps(r[i][1], _mm512_loadu_ps(ci + 16)); }
        _mm512_storeu_ps(ci, r[i][0]); _mm512_storeu_ps(ci + 16, r[i][1]);
      }
      return;
    }
    alignas(64) float t[MR * NR];
#pragma GCC unroll 12
    for (int i = 0; i < MR; ++i) { _mm512_store_ps(t + i * NR, r[i][0]); _mm512_store_ps(t + i * NR + 16, r[i][1]); }
    for (int i = 0; i < mr; ++i)
      for (int j = 0; j < nr; ++j) c[(int64_t)i * ldc + j] = acc ? c[(int64_t)i * ldc + j] + t[i * NR + j] : t[i * NR + j];
  }
};

bool pick_avx512() {
  const char* e = getenv("CPU_STEP_ISA");   // "avx2" forces the narrow kernel (tests)
  if (e && strcmp(e, "avx2") == 0) return false;
  return __builtin_cpu_supports("avx512f");
}
const bool g_avx512 = pick_avx512();

struct PackBuf {
  float *a = nullptr, *b = nullptr;
  size_t na = 0, nb = 0;
  void need(size_t fa, size_t fb) {
    if (fa > na) { free(a); a = (float*)aligned_alloc(64, ((fa * 4 + 63) / 64) * 64); na = fa; }
    if (fb > nb) { free(b); b = (float*)aligned_alloc(64, ((fb * 4 + 63) / 64) * 64); nb = fb; }
  }
  ~PackBuf() { free(a); free(b); }
};
thread_local PackBuf g_pack;

// one thread: C block [m x n] over the full K
template <class KT>
void gemm_block_t(int m, int n, int K, const float* A, int64_t rsa, int64_t csa, const float* B, int64_t rsb, int64_t csb, float* C, int ldc,
                  bool acc) {
  constexpr int MR = KT::MR, NR = KT::NR;
  const int mp = (m + MR - 1) / MR, np_ = (n + NR - 1) / NR;
  g_pack.need((size_t)mp * MR * KC, (size_t)np_ * NR * KC);
  float *ap = g_pack.a, *bp = g_pack.b;
  for (int k0 = 0; k0 < K; k0 += KC) {
    const int kc = std::min(KC, K - k0);
    for (int ip = 0; ip < mp; ++ip) {   // A panel ip: [kc][MR]
      float* d = ap + (size_t)ip * MR * kc;
      const int i0 = ip * MR, mr = std::min(MR, m - i0);
      const float* s = A + (int64_t)i0 * rsa + (int64_t)k0 * csa;
      for (int k = 0; k < kc; ++k) {
        for (int i = 0; i < mr; ++i) d[k * MR + i] = s[(int64_t)i * rsa + (int64_t)k * csa];
        for (int i = mr; i < MR; ++i) d[k * MR + i] = 0.f;
      }
    }
    for (int jp = 0; jp < np_; ++jp) {  // B panel jp: [kc][NR]
      float* d = bp + (size_t)jp * NR * kc;
      const int j0 = jp * NR, nr = std::min(NR, n - j0);
      const float* s = B + (int64_t)k0 * rsb + (int64_t)j0 * csb;
      if (csb == 1 && nr == NR) {
        for (int k = 0; k < kc; ++k) memcpy(d + k * NR, s + (int64_t)k * rsb, NR * 4);
      } else {
        for (int k = 0; k < kc; ++k) {
          for (int j = 0; j < nr; ++j) d[k * NR + j] = s[(int64_t)k * rsb + (int64_t)j * csb];
          for (int j = nr; j < NR; ++j) d[k * NR + j] = 0.f;
        }
      }
    }
    const bool a2 = acc || k0 > 0;
    for (int jp = 0; jp < np_; ++jp)
      for (int ip = 0; ip < mp; ++ip)
        KT::micro(kc, ap + (size_t)ip * MR * kc, bp + (size_t)jp * NR * kc, C + (int64_t)ip * MR * ldc + jp * NR, ldc,
                  std::min(MR, m - ip * MR), std::min(NR, n - jp * NR), a2);
  }
}
void gemm_block(int m, int n, int K, const float* A, int64_t rsa, int64_t csa, const float* B, int64_t rsb, int64_t csb, float* C, int ldc,
                bool acc) {
  if (m <= 0 || n <= 0) return;
  if (K <= 0) {
    if (!acc) for (int i = 0; i < m; ++i) memset(C + (int64_t)i * ldc, 0, (size_t)n * 4);
    return;
  }
  if (g_avx512) gemm_block_t<KernAvx512>(m, n, K, A, rsa, csa, B, rsb, csb, C, ldc, acc);
  else gemm_block_t<KernAvx2>(m, n, K, A, rsa, csa, B, rsb, csb, C, ldc, acc);
}

// all threads
void gemm(int M, int N, int K, const float* A, int64_t rsa, int64_t csa, const float* B, int64_t rsb, int64_t csb, float* C, int ldc, bool acc) {
  int MB = 96, NB = 256;
  const int nth = omp_get_max_threads();
  auto tasks = [&]() { return (int64_t)((M + MB - 1) / MB) * ((N + NB - 1) / NB); };
  while (tasks() < 3 * nth && (MB > 24 || NB > 64)) { if (NB > 64 && NB >= MB) NB /= 2; else MB /= 2; }
  const int tm = (M + MB - 1) / MB, tn = (N + NB - 1) / NB;
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (int im = 0; im < tm; ++im)
    for (int in = 0; in < tn; ++in) {
      const int i0 = im * MB, j0 = in * NB;
      gemm_block(std::min(MB, M - i0), std::min(NB, N - j0), K, A + (int64_t)i0 * rsa, rsa, csa, B + (int64_t)j0 * csb, rsb, csb,
                 C + (int64_t)i0 * ldc + j0, ldc, acc);
    }
}
// y[M,N] = x[M,K] . W[N,K]^T
inline void gemm_nt(int M, int N, int K, const float* X, int ldx, const float* W, int ldw, float* Y, int ldy, bool acc = false) {
  gemm(M, N, K, X, ldx, 1, W, 1, ldw, Y, ldy, acc);
}
// dx[M,K] = dy[M,N] . W[N,K]
inline void gemm_nn(int M, int K, int N, const float* DY, int lddy, const float* W, int ldw, float* DX, int lddx, bool acc = false) {
  gemm(M, K, N, DY, lddy, 1, W, ldw, 1, DX, lddx, acc);
}
// dW[N,K] = dy[M,N]^T . x[M,K]
inline void gemm_tn(int N, int K, int M, const float* DY, int lddy, const float* X, int ldx, float* DW, int lddw, bool acc = false) {
  gemm(N, K, M, DY, 1, lddy, X, ldx, 1, DW, lddw, acc);
}

// ------------------------------------------------------------------------------------------------ model
struct Cfg {
  int32_t L, H, KV, D, I, S, V0, V1, M, K, vs_status, vs_gender, vs_source, rows;
  double min_ts, max_ts;
  float rating_mean, rating_std;
};
enum { P_E = 0, P_META, P_WP, P_BP, P_PCOS, P_PSIN, P_STATUS, P_GENDER, P_SOURCE, P_LINW, P_LINB, P_NORM, P_R0W, P_R0B, P_R2W, P_R2B, P_LAYER0 };
enum { L_WQ = 0, L_WK, L_WV, L_WO, L_W1, L_W2, L_W3, L_SA, L_MLP, L_COUNT };
struct Batch {
  const int32_t *userid, *tmid, *gender, *source, *matchedid, *status;
  const double* time;
  const float *rating, *progress;
  const float* label[4]; const float* weight[4]; const int32_t* position[4];   // tasks (0,watch) (0,rating) (1,watch) (1,rating)
};

// exp over arrays, 8 lanes (Cephes expf polynomial, < 2 ulp; arguments below -87.3 -- the masked scores' -inf -- give exactly 0)
inline __m256 exp256(__m256 x) {
  const __m256 lo = _mm256_set1_ps(-87.3365447504f), hi = _mm256_set1_ps(88.3762626647949f);
  const __m256 under = _mm256_cmp_ps(x, lo, _CMP_LT_OQ);
  x = _mm256_max_ps(_mm256_min_ps(x, hi), lo);
  __m256 fx = _mm256_floor_ps(_mm256_fmadd_ps(x, _mm256_set1_ps(1.44269504088896341f), _mm256_set1_ps(0.5f)));
  x = _mm256_fnmadd_ps(fx, _mm256_set1_ps(0.693359375f), x);
  x = _mm256_fnmadd_ps(fx, _mm256_set1_ps(-2.12194440e-4f), x);
  const __m256 z = _mm256_mul_ps(x, x);
  __m256 y = _mm256_set1_ps(1.9875691500e-4f);
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(1.3981999507e-3f));
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(8.3334519073e-3f));
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(4.1665795894e-2f));
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(1.6666665459e-1f));
  y = _mm256_fmadd_ps(y, x, _mm256_set1_ps(5.0000001201e-1f));
  y = _mm256_add_ps(_mm256_fmadd_ps(y, z, x), _mm256_set1_ps(1.0f));
  const __m256i e = _mm256_slli_epi32(_mm256_add_epi32(_mm256_cvttps_epi32(fx), _mm256_set1_epi32(127)), 23);
  return _mm256_andnot_ps(under, _mm256_mul_ps(y, _mm256_castsi256_ps(e)));
}
// x[i] = exp(x[i] - shift) * mul; returns the sum of the exponentials before `mul`
inline float exp_inplace(float* x, int n, float shift, float mul) {
  const __m256 sh = _mm256_set1_ps(shift), mu = _mm256_set1_ps(mul);
  __m256 acc = _mm256_setzero_ps();
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    const __m256 e = exp256(_mm256_sub_ps(_mm256_loadu_ps(x + i), sh));
    acc = _mm256_add_ps(acc, e);
    _mm256_storeu_ps(x + i, _mm256_mul_ps(e, mu));
  }
  alignas(32) float t[8];
  _mm256_store_ps(t, acc);
  float sum = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  for (; i < n; ++i) { const float e = expf(x[i] - shift); sum += e; x[i] = e * mul; }
  return sum;
}
inline float exp_sum(const float* x, int n, float shift) {
  const __m256 sh = _mm256_set1_ps(shift);
  __m256 acc = _mm256_setzero_ps();
  int i = 0;
  for (; i + 8 <= n; i += 8) acc = _mm256_add_ps(acc, exp256(_mm256_sub_ps(_mm256_loadu_ps(x + i), sh)));
  alignas(32) float t[8];
  _mm256_store_ps(t, acc);
  float sum = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
  for (; i < n; ++i) sum += expf(x[i] - shift);
  return sum;
}
// SwiGLU (model.py:205-213): g = silu(a) * b; and its gradient
void swiglu_fwd(const float* a, const float* b, float* g, int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < (n + 4095) / 4096; ++c) {
    const int64_t i0 = c * 4096, i1 = std::min(n, i0 + 4096);
    int64_t i = i0;
    const __m256 one = _mm256_set1_ps(1.0f), zero = _mm256_setzero_ps();
    for (; i + 8 <= i1; i += 8) {
      const __m256 z = _mm256_loadu_ps(a + i);
      const __m256 sg = _mm256_div_ps(one, _mm256_add_ps(one, exp256(_mm256_sub_ps(zero, z))));
      _mm256_storeu_ps(g + i, _mm256_mul_ps(_mm256_mul_ps(z, sg), _mm256_loadu_ps(b + i)));
    }
    for (; i < i1; ++i) g[i] = a[i] / (1.0f + expf(-a[i])) * b[i];
  }
}
void swiglu_bwd(const float* gg, const float* a, const float* b, float* ga, float* gb, int64_t n) {
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < (n + 4095) / 4096; ++c) {
    const int64_t i0 = c * 4096, i1 = std::min(n, i0 + 4096);
    int64_t i = i0;
    const __m256 one = _mm256_set1_ps(1.0f), zero = _mm256_setzero_ps();
    for (; i + 8 <= i1; i += 8) {
      const __m256 z = _mm256_loadu_ps(a + i), g = _mm256_loadu_ps(gg + i);
      const __m256 sg = _mm256_div_ps(one, _mm256_add_ps(one, exp256(_mm256_sub_ps(zero, z))));
      const __m256 ds = _mm256_mul_ps(sg, _mm256_fmadd_ps(z, _mm256_sub_ps(one, sg), one));   // d silu / dz
      _mm256_storeu_ps(ga + i, _mm256_mul_ps(_mm256_mul_ps(g, _mm256_loadu_ps(b + i)), ds));
      _mm256_storeu_ps(gb + i, _mm256_mul_ps(_mm256_mul_ps(g, z), sg));
    }
    for (; i < i1; ++i) {
      const float zz = a[i], sg = 1.0f / (1.0f + expf(-zz));
      ga[i] = gg[i] * b[i] * (sg * (1.0f + zz * (1.0f - sg)));
      gb[i] = gg[i] * zz * sg;
    }
  }
}

void rmsnorm_fwd(const float* x, const float* sc, float* y, float* r, int64_t n, int D) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float* xi = x + i * D;
    float ss = 0.f;
    for (int j = 0; j < D; ++j) ss += xi[j] * xi[j];
    const float ri = 1.0f / sqrtf(ss / D + 1e-5f);   // model.py:193-202
    r[i] = ri;
    float* yi = y + i * D;
    for (int j = 0; j < D; ++j) yi[j] = xi[j] * ri * sc[j];
  }
}
// dx (+)= ...; dscale += ...
void rmsnorm_bwd(const float* g, const float* x, const float* sc, const float* r, float* dx, bool acc, float* dscale, int64_t n, int D) {
  const int nth = omp_get_max_threads();
  std::vector<float> part((size_t)nth * D, 0.f);
#pragma omp parallel
  {
    float* ds = part.data() + (size_t)omp_get_thread_num() * D;
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
      const float *gi = g + i * D, *xi = x + i * D;
      const float ri = r[i];
      float dot = 0.f;
      for (int j = 0; j < D; ++j) { dot += gi[j] * sc[j] * xi[j]; ds[j] += gi[j] * xi[j] * ri; }
      const float c = ri * ri * ri * dot / D;
      float* di = dx + i * D;
      if (acc) for (int j = 0; j < D; ++j) di[j] += ri * gi[j] * sc[j] - xi[j] * c;
      else for (int j = 0; j < D; ++j) di[j] = ri * gi[j] * sc[j] - xi[j] * c;
    }
  }
  for (int t = 0; t < nth; ++t) for (int j = 0; j < D; ++j) dscale[j] += part[(size_t)t * D + j];
}

// interleaved-pair rotation (model.py:182-190); x rows of `heads` heads of hd, token position t = row % T; sign -1 = transpose
void rope(float* x, int64_t n, int T, int heads, int hd, const float* cs, const float* sn, float sign) {
  const int h2 = hd / 2;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float *c = cs + (int64_t)(i % T) * h2, *s = sn + (int64_t)(i % T) * h2;
    float* xi = x + i * heads * hd;
    for (int h = 0; h < heads; ++h)
      for (int p = 0; p < h2; ++p) {
        const float a = xi[h * hd + 2 * p], b = xi[h * hd + 2 * p + 1], ss = sign * s[p];
        xi[h * hd + 2 * p] = a * c[p] - b * ss;
        xi[h * hd + 2 * p + 1] = a * ss + b * c[p];
      }
  }
}

struct Layer { float *x, *xn, *r1, *q, *k, *v, *lse, *o, *h, *hn, *r2, *a, *b, *g; };

// Work buffers are kept between calls (same configuration and rows -> the same request sequence): a training loop reuses its
// activations' memory, and first-touch page faults of ~20 GB would otherwise be a third of the timed step.
std::vector<std::pair<float*, size_t>> g_pool;
struct Arena {
  size_t at = 0;
  float* get(size_t n) {
    if (at == g_pool.size()) g_pool.emplace_back(nullptr, 0);
    auto& e = g_pool[at++];
    if (e.second < n) { free(e.first); e.first = (float*)aligned_alloc(64, ((n * 4 + 63) / 64) * 64); e.second = n; }
    return e.first;
  }
};

// operand_round = "bf16" (cpu_step_set_operand_round): every array the HIP path stores as a bf16 GEMM operand is rounded to
// bfloat16 (nearest even) at the point the kernels store it, accumulation stays float -- the same rounding points as
// oracle/model_np.py's OracleModel(operand_round="bf16"), so that the benchmarked arithmetic can be checked at its own size
bool g_round = false;
inline float bf16r(float x) {
  uint32_t u; memcpy(&u, &x, 4);
  u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
  memcpy(&x, &u, 4);
  return x;
}
void round_arr(float* x, int64_t n) {
  if (!g_round) return;
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < (n + 4095) / 4096; ++c) {
    const int64_t i1 = std::min(n, c * 4096 + 4096);
    for (int64_t i = c * 4096; i < i1; ++i) x[i] = bf16r(x[i]);
  }
}
// rows x cols block of `src` (row stride ld) as a rounded dense copy; `src` itself when rounding is off and ld == cols
const float* rounded_copy(Arena& ar, const float* src, int64_t rows, int64_t cols, int64_t ld) {
  if (!g_round && ld == cols) return src;
  float* d = ar.get((size_t)(rows * cols));
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < rows; ++r)
    for (int64_t j = 0; j < cols; ++j) d[r * cols + j] = g_round ? bf16r(src[r * ld + j]) : src[r * ld + j];
  return d;
}

inline bool allowed(int uq, int tq, int uk, int tk) { return uq == uk && (tk == 0 || tq == tk); }   // model.py:479-487

constexpr int QB = 48;   // queries (or keys) per attention task

// candidate key range of tokens [i0, i1) of a row: the union of the runs of equal userid they belong to (runs: rs/re per token);
// rows in which a userid occurs in two separate runs use the whole row (flag)
struct Runs { std::vector<int> rs, re; std::vector<char> whole; };

void attention_fwd(const Cfg& c, const float* q, const float* k, const float* v, const int* uid, const int* tm, const Runs& R, float* o,
                   float* lse) {
  const int T = 2 * c.S, H = c.H, KV = c.KV, hd = c.D / c.H, rep = H / KV, ldq = H * hd, ldk = KV * hd;
  const float scale = 1.0f / sqrtf((float)hd);
  const int nqb = (T + QB - 1) / QB;
#pragma omp parallel
  {
    std::vector<float> sbuf((size_t)QB * T);
    float invs[QB];
#pragma omp for schedule(dynamic, 1) collapse(3)
    for (int b = 0; b < c.rows; ++b)
      for (int h = 0; h < H; ++h)
        for (int qb = 0; qb < nqb; ++qb) {
          const int i0 = qb * QB, i1 = std::min(T, i0 + QB), nq = i1 - i0;
          const int64_t base = (int64_t)b * T;
          int ks = T, ke = 0;
          if (R.whole[b]) { ks = 0; ke = T; }
          else for (int i = i0; i < i1; ++i) { ks = std::min(ks, R.rs[base + i]); ke = std::max(ke, R.re[base + i]); }
          const int nk = ke - ks;
          float* s = sbuf.data();
          const float* kh = k + (base + ks) * ldk + (h / rep) * hd;
          const float* vh = v + (base + ks) * ldk + (h / rep) * hd;
          gemm_block(nq, nk, hd, q + (base + i0) * ldq + h * hd, ldq, 1, kh, 1, ldk, s, nk, false);
          for (int i = 0; i < nq; ++i) {
            float* si = s + (size_t)i * nk;
            const int uq = uid[base + i0 + i], tq = tm[base + i0 + i];
            float mx = -INFINITY;
            for (int j = 0; j < nk; ++j) {
              const bool ok = allowed(uq, tq, uid[base + ks + j], tm[base + ks + j]);
              si[j] = ok ? si[j] * scale : -INFINITY;
              mx = std::max(mx, si[j]);
            }
            const float den = exp_inplace(si, nk, mx, 1.0f);
            const float inv = 1.0f / den;
            if (g_round) { for (int j = 0; j < nk; ++j) si[j] = bf16r(si[j]); invs[i] = inv; }   // (the flash kernels round exp(s - max), not the quotient)
            else for (int j = 0; j < nk; ++j) si[j] *= inv;
            lse[((int64_t)b * H + h) * T + i0 + i] = mx + logf(den);
          }
          gemm_block(nq, hd, nk, s, nk, 1, vh, ldk, 1, o + (base + i0) * ldq + h * hd, ldq, false);
          if (g_round)
            for (int i = 0; i < nq; ++i) { float* oi = o + (base + i0 + i) * ldq + h * hd; for (int e = 0; e < hd; ++e) oi[e] = bf16r(oi[e] * invs[i]); }
        }
  }
}

// gq (B,T,H,hd), gk / gv (B,T,KV,hd) = gradients w.r.t. the rotated q, k and v
void attention_bwd(const Cfg& c, const float* q, const float* k, const float* v, const float* o, const float* go, const float* lse, const int* uid,
                   const int* tm, const Runs& R, float* gq, float* gk, float* gv, float* delta) {
  const int T = 2 * c.S, H = c.H, KV = c.KV, hd = c.D / c.H, rep = H / KV, ldq = H * hd, ldk = KV * hd;
  const float scale = 1.0f / sqrtf((float)hd);
  const int nqb = (T + QB - 1) / QB;
  const int64_t NT = (int64_t)c.rows * T;
  // delta[b,h,t] = sum_d go * o
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < NT; ++i)
    for (int h = 0; h < H; ++h) {
      float d = 0.f;
      for (int e = 0; e < hd; ++e) d += go[i * ldq + h * hd + e] * o[i * ldq + h * hd + e];
      delta[((i / T) * H + h) * T + i % T] = d;
    }
#pragma omp parallel
  {
    std::vector<float> sbuf((size_t)QB * T), pbuf((size_t)QB * T);
    // dq: per (row, head, query block)
#pragma omp for schedule(dynamic, 1) collapse(3)
    for (int b = 0; b < c.rows; ++b)
      for (int h = 0; h < H; ++h)
        for (int qb = 0; qb < nqb; ++qb) {
          const int i0 = qb * QB, i1 = std::min(T, i0 + QB), nq = i1 - i0;
          const int64_t base = (int64_t)b * T;
          int ks = T, ke = 0;
          if (R.whole[b]) { ks = 0; ke = T; }
          else for (int i = i0; i < i1; ++i) { ks = std::min(ks, R.rs[base + i]); ke = std::max(ke, R.re[base + i]); }
          const int nk = ke - ks;
          float *s = sbuf.data(), *dp = pbuf.data();
          const float* kh = k + (base + ks) * ldk + (h / rep) * hd;
          const float* vh = v + (base + ks) * ldk + (h / rep) * hd;
          gemm_block(nq, nk, hd, q + (base + i0) * ldq + h * hd, ldq, 1, kh, 1, ldk, s, nk, false);
          gemm_block(nq, nk, hd, go + (base + i0) * ldq + h * hd, ldq, 1, vh, 1, ldk, dp, nk, false);
          for (int i = 0; i < nq; ++i) {
            const int uq = uid[base + i0 + i], tq = tm[base + i0 + i];
            const float l = lse[((int64_t)b * H + h) * T + i0 + i], dl = delta[((int64_t)b * H + h) * T + i0 + i];
            float *si = s + (size_t)i * nk, *di = dp + (size_t)i * nk;
            for (int j = 0; j < nk; ++j) si[j] = allowed(uq, tq, uid[base + ks + j], tm[base + ks + j]) ? si[j] * scale : -INFINITY;
            exp_inplace(si, nk, l, 1.0f);
            // operand rounding: the flash kernels feed dS' = P (dP - delta) to the MFMA as bf16 and apply 1/sqrt(hd) to the sum
            if (g_round) for (int j = 0; j < nk; ++j) si[j] = bf16r(si[j] * (di[j] - dl));
            else for (int j = 0; j < nk; ++j) si[j] = si[j] * (di[j] - dl) * scale;
          }
          gemm_block(nq, hd, nk, s, nk, 1, kh, ldk, 1, gq + (base + i0) * ldq + h * hd, ldq, false);
          if (g_round)
            for (int i = 0; i < nq; ++i) { float* gi = gq + (base + i0 + i) * ldq + h * hd; for (int e = 0; e < hd; ++e) gi[e] *= scale; }
        }
    // dk, dv: per (row, kv head, key block), summed over the query heads of the group
#pragma omp for schedule(dynamic, 1) collapse(3)
    for (int b = 0; b < c.rows; ++b)
      for (int g = 0; g < KV; ++g)
        for (int kb = 0; kb < nqb; ++kb) {
          const int j0 = kb * QB, j1 = std::min(T, j0 + QB), nk = j1 - j0;
          const int64_t base = (int64_t)b * T;
          int qs = T, qe = 0;
          if (R.whole[b]) { qs = 0; qe = T; }
          else for (int j = j0; j < j1; ++j) { qs = std::min(qs, R.rs[base + j]); qe = std::max(qe, R.re[base + j]); }
          const int nq = qe - qs;
          float *st = sbuf.data(), *dpt = pbuf.data();   // transposed blocks [nk][nq]
          const float* kh = k + (base + j0) * ldk + g * hd;
          const float* vh = v + (base + j0) * ldk + g * hd;
          for (int r = 0; r < rep; ++r) {
            const int h = g * rep + r;
            const float* qh = q + (base + qs) * ldq + h * hd;
            const float* goh = go + (base + qs) * ldq + h * hd;
            gemm_block(nk, nq, hd, kh, ldk, 1, qh, 1, ldq, st, nq, false);
            gemm_block(nk, nq, hd, vh, ldk, 1, goh, 1, ldq, dpt, nq, false);
            for (int j = 0; j < nk; ++j) {
              const int uk = uid[base + j0 + j], tk = tm[base + j0 + j];
              float *sj = st + (size_t)j * nq, *dj = dpt + (size_t)j * nq;
              const float* lrow = lse + ((int64_t)b * H + h) * T + qs;
              const float* drow = delta + ((int64_t)b * H + h) * T + qs;
              for (int i = 0; i < nq; ++i)
                sj[i] = allowed(uid[base + qs + i], tm[base + qs + i], uk, tk) ? sj[i] * scale - lrow[i] : -INFINITY;
              exp_inplace(sj, nq, 0.f, 1.0f);                                           // P^T
              if (g_round) for (int i = 0; i < nq; ++i) { dj[i] = bf16r(sj[i] * (dj[i] - drow[i])); sj[i] = bf16r(sj[i]); }   // bf16 MFMA operands; scale below
              else for (int i = 0; i < nq; ++i) dj[i] = sj[i] * (dj[i] - drow[i]) * scale;   // dS^T
            }
            gemm_block(nk, hd, nq, st, nq, 1, goh, ldq, 1, gv + (base + j0) * ldk + g * hd, ldk, r > 0);
            gemm_block(nk, hd, nq, dpt, nq, 1, qh, ldq, 1, gk + (base + j0) * ldk + g * hd, ldk, r > 0);
          }
          if (g_round)
            for (int j = 0; j < nk; ++j) { float* gj = gk + (base + j0 + j) * ldk + g * hd; for (int e = 0; e < hd; ++e) gj[e] *= scale; }
        }
  }
}

inline float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
inline float gelu_grad_f(float x) { return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * expf(-0.5f * x * x) * 0.39894228040143268f; }

void colsum_add(const float* x, int64_t n, int D, int ldx, float* out) {
  const int nth = omp_get_max_threads();
  std::vector<float> part((size_t)nth * D, 0.f);
#pragma omp parallel
  {
    float* p = part.data() + (size_t)omp_get_thread_num() * D;
#pragma omp for schedule(static)
    for (int64_t i = 0; i < n; ++i) for (int j = 0; j < D; ++j) p[j] += x[i * ldx + j];
  }
  for (int t = 0; t < nth; ++t) for (int j = 0; j < D; ++j) out[j] += part[(size_t)t * D + j];
}

}  // namespace

namespace {
struct Phase {   // CPU_STEP_TIMING=1: seconds per phase on stderr
  bool on = getenv("CPU_STEP_TIMING") != nullptr;
  double t0 = omp_get_wtime();
  void mark(const char* what) { if (!on) return; const double t = omp_get_wtime(); fprintf(stderr, "cpu_step %-28s %8.3f s\n", what, t - t0); t0 = t; }
};
}  // namespace

extern "C" {

int cpu_step_threads(void) { return omp_get_max_threads(); }
void cpu_step_set_threads(int n) { if (n >= 1) omp_set_num_threads(n); }
void cpu_step_set_operand_round(int bf16) { g_round = bf16 != 0; }
void cpu_step_release(void) { for (auto& e : g_pool) free(e.first); g_pool.clear(); }

// One forward + backward of sum_i task_w[i] * loss_i (model.py:493-529 and its autograd).  params / grads: pointer tables in
// the order of the P_* / L_* enums (oracle/cpu_step.py builds them from the state-dict names); grads[P_META] is ignored
// (frozen, model.py:113-114).  Gradients are overwritten.  Returns 0, or -1 on a bad index in the batch.
int cpu_step_forward_backward(const Cfg* cfg, const float* const* P, const Batch* bt, const float* task_w, float* losses, float* const* G) {
  const Cfg& c = *cfg;
  const int D = c.D, I = c.I, S = c.S, T = 2 * S, H = c.H, KV = c.KV, hd = D / H, Dk = KV * hd, V = c.V0 + c.V1, M = c.M;
  const int64_t N = (int64_t)c.rows * S, NT = 2 * N;
  Arena ar;
  Phase ph;
  auto LP = [&](int l, int w) { return P[P_LAYER0 + l * L_COUNT + w]; };
  // weights as GEMM operands: the bf16 shadow copies in operand-rounding mode (the metadata table is expected to hold
  // bfloat16-representable values already: 4.9 GB at cfg-3, not copied)
  std::vector<const float*> PWv;
  if (g_round) {
    PWv.assign(P_LAYER0 + c.L * L_COUNT, nullptr);
    PWv[P_WP] = rounded_copy(ar, P[P_WP], D, M, M); PWv[P_LINW] = rounded_copy(ar, P[P_LINW], D, 32, 32); PWv[P_R0W] = rounded_copy(ar, P[P_R0W], D, D, D);
    for (int l = 0; l < c.L; ++l) {
      const int b = P_LAYER0 + l * L_COUNT;
      PWv[b + L_WQ] = rounded_copy(ar, P[b + L_WQ], D, D, D); PWv[b + L_WK] = rounded_copy(ar, P[b + L_WK], Dk, D, D);
      PWv[b + L_WV] = rounded_copy(ar, P[b + L_WV], Dk, D, D); PWv[b + L_WO] = rounded_copy(ar, P[b + L_WO], D, D, D);
      PWv[b + L_W1] = rounded_copy(ar, P[b + L_W1], I, D, D); PWv[b + L_W3] = rounded_copy(ar, P[b + L_W3], I, D, D);
      PWv[b + L_W2] = rounded_copy(ar, P[b + L_W2], D, I, I);
    }
  }
  auto PW = [&](int i) { return g_round ? PWv[i] : P[i]; };
  auto LW = [&](int l, int w) { return PW(P_LAYER0 + l * L_COUNT + w); };
  auto LG = [&](int l, int w) { return G[P_LAYER0 + l * L_COUNT + w]; };
  for (int64_t i = 0; i < N; ++i) {
    if (bt->matchedid[i] < -1 || bt->matchedid[i] >= V) return -1;
    if (bt->status[i] < -1 || bt->status[i] > c.vs_status || bt->gender[i] < -1 || bt->gender[i] > c.vs_gender || bt->source[i] < -1 ||
        bt->source[i] > c.vs_source)
      return -1;
  }
  // ---- action features (model.py:51-93)
  float* feat = ar.get(N * 32);
  float* pcs = ar.get(N * 4);   // cos/sin arguments for the phase gradients
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < N; ++i) {
    float* f = feat + i * 32;
    const double ts = std::max(bt->time[i], c.min_ts);
    const float per0 = (float)(2.0 * M_PI * ts / 86400.0), per1 = (float)(2.0 * M_PI * ts / 604800.0);
    f[0] = (float)((ts - c.min_ts) / (c.max_ts - c.min_ts));
    const float pc0 = per0 + P[P_PCOS][0], pc1 = per1 + P[P_PCOS][1], ps0 = per0 + P[P_PSIN][0], ps1 = per1 + P[P_PSIN][1];
    pcs[i * 4 + 0] = pc0; pcs[i * 4 + 1] = pc1; pcs[i * 4 + 2] = ps0; pcs[i * 4 + 3] = ps1;
    f[1] = cosf(pc0); f[2] = cosf(pc1); f[3] = sinf(ps0); f[4] = sinf(ps1);
    const int gi = bt->gender[i] == -1 ? c.vs_gender : bt->gender[i], si = bt->source[i] == -1 ? c.vs_source : bt->source[i];
    const int sti = bt->status[i] == -1 ? c.vs_status : bt->status[i];
    for (int j = 0; j < 4; ++j) { f[5 + j] = P[P_GENDER][gi * 4 + j]; f[9 + j] = P[P_SOURCE][si * 4 + j]; }
    const float has = bt->rating[i] != 0.f ? 1.f : 0.f;
    f[13] = has; f[14] = has * ((bt->rating[i] - c.rating_mean) / c.rating_std);
    for (int j = 0; j < 16; ++j) f[15 + j] = P[P_STATUS][sti * 16 + j];
    f[31] = bt->progress[i];
  }
  round_arr(feat, N * 32);
  ph.mark("action features");
  // ---- fused item table F = E + Meta Wp^T + bp (model.py:120-133, 143-145)
  float* F = ar.get((size_t)(V + 1) * D);
  gemm_nt(V + 1, D, M, P[P_META], M, PW(P_WP), M, F, D);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)(V + 1); ++i)
    for (int j = 0; j < D; ++j) F[i * D + j] += P[P_E][i * D + j] + P[P_BP][j];
  const float* Fq = rounded_copy(ar, F, V + 1, D, D);   // the tied watch-head operand
  ph.mark("fused table");
  // ---- x0: even tokens items, odd tokens actions (model.py:403-415)
  float* x = ar.get(NT * D);
  gemm_nt((int)N, D, 32, feat, 32, PW(P_LINW), 32, x + D, 2 * D);
  std::vector<int> uid(NT), tm(NT), ids(N);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < N; ++i) {
    const int id = bt->matchedid[i] == -1 ? V : bt->matchedid[i];
    ids[i] = id;
    memcpy(x + 2 * i * D, F + (int64_t)id * D, (size_t)D * 4);
    float* xa = x + (2 * i + 1) * D;
    for (int j = 0; j < D; ++j) xa[j] += P[P_LINB][j];
    uid[2 * i] = uid[2 * i + 1] = bt->userid[i];
    tm[2 * i] = tm[2 * i + 1] = bt->tmid[i];
  }
  Runs R; R.rs.resize(NT); R.re.resize(NT); R.whole.assign(c.rows, 0);
  for (int b = 0; b < c.rows; ++b) {
    const int64_t base = (int64_t)b * T;
    std::vector<int> seen;
    int s = 0;
    while (s < T) {
      int e = s + 1;
      while (e < T && uid[base + e] == uid[base + s]) ++e;
      for (int i = s; i < e; ++i) { R.rs[base + i] = s; R.re[base + i] = e; }
      if (std::find(seen.begin(), seen.end(), uid[base + s]) != seen.end()) R.whole[b] = 1;
      seen.push_back(uid[base + s]);
      s = e;
    }
  }
  // rope tables (model.py:173-179), float32
  const int h2 = hd / 2;
  std::vector<float> cs((size_t)T * h2), sn((size_t)T * h2);
  for (int t = 0; t < T; ++t)
    for (int p = 0; p < h2; ++p) {
      const float fr = 1.0f / powf(500000.0f, (float)(2 * p) / (float)hd);
      const float a = (float)t * fr;
      cs[(size_t)t * h2 + p] = cosf(a); sn[(size_t)t * h2 + p] = sinf(a);
    }
  ph.mark("gather, runs, rope tables");
  // ---- trunk forward (model.py:297-309, 335-343)
  std::vector<Layer> lay(c.L);
  float* ab = ar.get(NT * 2 * (size_t)I);   // scratch for one layer's gate gradients in the backward
  for (int l = 0; l < c.L; ++l) {
    Layer& a = lay[l];
    a.x = x;
    a.xn = ar.get(NT * D); a.r1 = ar.get(NT); a.q = ar.get(NT * D); a.k = ar.get(NT * Dk); a.v = ar.get(NT * Dk);
    a.lse = ar.get((size_t)c.rows * H * T); a.o = ar.get(NT * D); a.h = ar.get(NT * D); a.hn = ar.get(NT * D); a.r2 = ar.get(NT);
    a.a = ar.get(NT * (size_t)I); a.b = ar.get(NT * (size_t)I); a.g = ar.get(NT * (size_t)I);
    rmsnorm_fwd(a.x, LP(l, L_SA), a.xn, a.r1, NT, D);
    round_arr(a.xn, NT * D);
    gemm_nt((int)NT, D, D, a.xn, D, LW(l, L_WQ), D, a.q, D);
    gemm_nt((int)NT, Dk, D, a.xn, D, LW(l, L_WK), D, a.k, Dk);
    gemm_nt((int)NT, Dk, D, a.xn, D, LW(l, L_WV), D, a.v, Dk);
    rope(a.q, NT, T, H, hd, cs.data(), sn.data(), 1.f);
    rope(a.k, NT, T, KV, hd, cs.data(), sn.data(), 1.f);
    round_arr(a.q, NT * D); round_arr(a.k, NT * Dk); round_arr(a.v, NT * Dk);
    attention_fwd(c, a.q, a.k, a.v, uid.data(), tm.data(), R, a.o, a.lse);   // (rounds its output itself in operand-rounding mode)
    memcpy(a.h, a.x, (size_t)NT * D * 4);
    gemm_nt((int)NT, D, D, a.o, D, LW(l, L_WO), D, a.h, D, true);
    rmsnorm_fwd(a.h, LP(l, L_MLP), a.hn, a.r2, NT, D);
    round_arr(a.hn, NT * D);
    gemm_nt((int)NT, I, D, a.hn, D, LW(l, L_W1), D, a.a, I);
    gemm_nt((int)NT, I, D, a.hn, D, LW(l, L_W3), D, a.b, I);
    swiglu_fwd(a.a, a.b, a.g, NT * (int64_t)I);
    round_arr(a.g, NT * (int64_t)I); round_arr(a.a, NT * (int64_t)I); round_arr(a.b, NT * (int64_t)I);   // (the product used the accumulators; the backward reads the stored operands)
    float* out = ar.get(NT * D);
    memcpy(out, a.h, (size_t)NT * D * 4);
    gemm_nt((int)NT, D, I, a.g, I, LW(l, L_W2), I, out, D, true);
    x = out;
  }
  float* xL = x;
  float* y = ar.get(NT * D);
  float* rf = ar.get(NT);
  rmsnorm_fwd(xL, P[P_NORM], y, rf, NT, D);
  round_arr(y, NT * D);
  ph.mark("trunk forward");
  // ---- heads (model.py:499-526)
  float* gy = ar.get(NT * D);
  memset(gy, 0, (size_t)NT * D * 4);
  float* gF = G[P_E];
  memset(gF, 0, (size_t)(V + 1) * D * 4);
  for (int i : {P_R0W, P_R0B, P_R2W, P_R2B}) memset(G[i], 0, (size_t)(i == P_R0W ? D * D : i == P_R2B ? 1 : D) * 4);
  const int KB = c.K * c.rows;
  std::vector<int> order(N), bp(KB);
  float* emb = ar.get((size_t)KB * D);
  float* gemb = ar.get((size_t)KB * D);
  float* logits = ar.get((size_t)KB * std::max(c.V0, c.V1));
  float* z = ar.get((size_t)KB * D);
  float* hact = ar.get((size_t)KB * D);
  for (int ti = 0; ti < 4; ++ti) {
    const int medium = ti / 2, rating = ti % 2;
    const float *w = bt->weight[ti], *lab = bt->label[ti];
    const int32_t* pos = bt->position[ti];
    // model.py:509 torch.topk made deterministic: larger weights first, ties by ascending flat index
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w[a] > w[b]; });
    std::copy(order.begin(), order.begin() + KB, bp.begin());
    double wsum = 0;
    for (int r = 0; r < KB; ++r) wsum += w[bp[r]];
    const float ws = (float)std::max(wsum, 1e-8);
    const float tw = task_w[ti];
#pragma omp parallel for schedule(static)
    for (int r = 0; r < KB; ++r) memcpy(emb + (int64_t)r * D, y + (2 * (int64_t)bp[r] + rating) * D, (size_t)D * 4);
    if (!rating) {
      const int s0 = medium == 0 ? 0 : c.V0, Vm = medium == 0 ? c.V0 : c.V1;
      for (int r = 0; r < KB; ++r) if (pos[bp[r]] < 0 || pos[bp[r]] >= Vm) return -1;
      gemm_nt(KB, Vm, D, emb, D, Fq + (int64_t)s0 * D, D, logits, Vm);
      round_arr(logits, (int64_t)KB * Vm);
      double loss = 0;
#pragma omp parallel for schedule(static) reduction(+ : loss)
      for (int r = 0; r < KB; ++r) {
        float* lr = logits + (int64_t)r * Vm;
        float mx = lr[0];
        for (int j = 1; j < Vm; ++j) mx = std::max(mx, lr[j]);
        const float lse = mx + logf(exp_sum(lr, Vm, mx));
        const int t = pos[bp[r]];
        loss += (double)((lse - lr[t]) * lab[bp[r]] * w[bp[r]]);
        const float coef = tw * lab[bp[r]] * w[bp[r]] / ws;
        exp_inplace(lr, Vm, lse, coef);
        lr[t] -= coef;
        if (g_round) for (int j = 0; j < Vm; ++j) lr[j] = bf16r(lr[j]);
      }
      losses[ti] = (float)(loss / ws);
      if (tw != 0.f) {
        gemm_nn(KB, D, Vm, logits, Vm, Fq + (int64_t)s0 * D, D, gemb, D);
        gemm_tn(Vm, D, KB, logits, Vm, emb, D, gF + (int64_t)s0 * D, D, true);
        for (int r = 0; r < KB; ++r) { float* d = gy + 2 * (int64_t)bp[r] * D; for (int j = 0; j < D; ++j) d[j] += gemb[(int64_t)r * D + j]; }
      }
    } else {
      gemm_nt(KB, D, D, emb, D, PW(P_R0W), D, z, D);
      std::vector<float> gp(KB);
      double loss = 0;
#pragma omp parallel for schedule(static) reduction(+ : loss)
      for (int r = 0; r < KB; ++r) {
        float pr = P[P_R2B][0];
        for (int j = 0; j < D; ++j) {
          const float zz = z[(int64_t)r * D + j] + P[P_R0B][j];
          z[(int64_t)r * D + j] = g_round ? bf16r(zz) : zz;
          const float hh = g_round ? bf16r(gelu_f(zz)) : gelu_f(zz);
          hact[(int64_t)r * D + j] = hh;
          pr += hh * P[P_R2W][j];
        }
        const float tgt = lab[bp[r]] - c.rating_mean, ww = w[bp[r]];
        loss += (double)((pr - tgt) * (pr - tgt) * ww);
        gp[r] = tw * 2.0f * (pr - tgt) * ww / ws;
      }
      losses[ti] = (float)(loss / ws);
      if (tw != 0.f) {
        for (int r = 0; r < KB; ++r) {
          G[P_R2B][0] += gp[r];
          for (int j = 0; j < D; ++j) G[P_R2W][j] += gp[r] * hact[(int64_t)r * D + j];
        }
#pragma omp parallel for schedule(static)
        for (int r = 0; r < KB; ++r)
          for (int j = 0; j < D; ++j) { const float gz = gp[r] * P[P_R2W][j] * gelu_grad_f(z[(int64_t)r * D + j]); z[(int64_t)r * D + j] = g_round ? bf16r(gz) : gz; }
        gemm_tn(D, D, KB, z, D, emb, D, G[P_R0W], D, true);
        colsum_add(z, KB, D, D, G[P_R0B]);
        gemm_nn(KB, D, D, z, D, PW(P_R0W), D, gemb, D);
        for (int r = 0; r < KB; ++r) { float* d = gy + (2 * (int64_t)bp[r] + 1) * D; for (int j = 0; j < D; ++j) d[j] += gemb[(int64_t)r * D + j]; }
      }
    }
  }
  ph.mark("heads forward + backward");
  // ---- trunk backward
  float* gx = ar.get(NT * D);
  memset(G[P_NORM], 0, (size_t)D * 4);
  rmsnorm_bwd(gy, xL, P[P_NORM], rf, gx, false, G[P_NORM], NT, D);
  float* gg = ar.get(NT * (size_t)I);
  float* ghn = ar.get(NT * D);
  float* go = ar.get(NT * D);
  float* gq = ar.get(NT * D); float* gk = ar.get(NT * Dk); float* gv = ar.get(NT * Dk);
  float* delta = ar.get((size_t)c.rows * H * T);
  float *ga = ab, *gb = ab + NT * (size_t)I;
  float* gxq = g_round ? ar.get(NT * D) : nullptr;   // the residual-stream gradient as a (rounded) GEMM operand
  auto operand = [&](const float* g) -> const float* {
    if (!g_round) return g;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < NT; ++i) for (int j = 0; j < D; ++j) gxq[i * D + j] = bf16r(g[i * D + j]);
    return gxq;
  };
  for (int l = c.L - 1; l >= 0; --l) {
    Layer& a = lay[l];
    const float* gxo = operand(gx);
    gemm_tn(D, I, (int)NT, gxo, D, a.g, I, LG(l, L_W2), I);
    gemm_nn((int)NT, I, D, gxo, D, LW(l, L_W2), I, gg, I);
    swiglu_bwd(gg, a.a, a.b, ga, gb, NT * (int64_t)I);
    round_arr(ga, NT * (int64_t)I); round_arr(gb, NT * (int64_t)I);
    gemm_tn(I, D, (int)NT, ga, I, a.hn, D, LG(l, L_W1), D);
    gemm_tn(I, D, (int)NT, gb, I, a.hn, D, LG(l, L_W3), D);
    gemm_nn((int)NT, D, I, ga, I, LW(l, L_W1), D, ghn, D);
    gemm_nn((int)NT, D, I, gb, I, LW(l, L_W3), D, ghn, D, true);
    round_arr(ghn, NT * D);
    memset(LG(l, L_MLP), 0, (size_t)D * 4);
    rmsnorm_bwd(ghn, a.h, LP(l, L_MLP), a.r2, gx, true, LG(l, L_MLP), NT, D);       // gx = gh now
    const float* gho = operand(gx);
    gemm_tn(D, D, (int)NT, gho, D, a.o, D, LG(l, L_WO), D);
    gemm_nn((int)NT, D, D, gho, D, LW(l, L_WO), D, go, D);
    round_arr(go, NT * D);
    attention_bwd(c, a.q, a.k, a.v, a.o, go, a.lse, uid.data(), tm.data(), R, gq, gk, gv, delta);
    rope(gq, NT, T, H, hd, cs.data(), sn.data(), -1.f);
    rope(gk, NT, T, KV, hd, cs.data(), sn.data(), -1.f);
    round_arr(gq, NT * D); round_arr(gk, NT * Dk); round_arr(gv, NT * Dk);
    gemm_tn(D, D, (int)NT, gq, D, a.xn, D, LG(l, L_WQ), D);
    gemm_tn(Dk, D, (int)NT, gk, Dk, a.xn, D, LG(l, L_WK), D);
    gemm_tn(Dk, D, (int)NT, gv, Dk, a.xn, D, LG(l, L_WV), D);
    gemm_nn((int)NT, D, D, gq, D, LW(l, L_WQ), D, ghn, D);
    gemm_nn((int)NT, D, Dk, gk, Dk, LW(l, L_WK), D, ghn, D, true);
    gemm_nn((int)NT, D, Dk, gv, Dk, LW(l, L_WV), D, ghn, D, true);
    round_arr(ghn, NT * D);
    memset(LG(l, L_SA), 0, (size_t)D * 4);
    rmsnorm_bwd(ghn, a.x, LP(l, L_SA), a.r1, gx, true, LG(l, L_SA), NT, D);
  }
  ph.mark("trunk backward");
  // ---- embeddings backward: gx even rows -> item table rows, odd rows -> action embedding
  for (int64_t i = 0; i < N; ++i) { float* d = gF + (int64_t)ids[i] * D; const float* s = gx + 2 * i * D; for (int j = 0; j < D; ++j) d[j] += s[j]; }
  gemm_tn(D, M, V + 1, rounded_copy(ar, gF, V + 1, D, D), D, P[P_META], M, G[P_WP], M);
  memset(G[P_BP], 0, (size_t)D * 4);
  colsum_add(gF, V + 1, D, D, G[P_BP]);
  gemm_tn(D, 32, (int)N, gx + D, 2 * D, feat, 32, G[P_LINW], 32);
  memset(G[P_LINB], 0, (size_t)D * 4);
  colsum_add(gx + D, N, D, 2 * D, G[P_LINB]);
  float* gf = ar.get(N * 32);
  if (g_round) gemm_nn((int)N, 32, D, rounded_copy(ar, gx + D, N, D, 2 * D), D, PW(P_LINW), 32, gf, 32);
  else gemm_nn((int)N, 32, D, gx + D, 2 * D, P[P_LINW], 32, gf, 32);
  memset(G[P_PCOS], 0, 8); memset(G[P_PSIN], 0, 8);
  memset(G[P_STATUS], 0, (size_t)(c.vs_status + 1) * 16 * 4); memset(G[P_GENDER], 0, (size_t)(c.vs_gender + 1) * 4 * 4);
  memset(G[P_SOURCE], 0, (size_t)(c.vs_source + 1) * 4 * 4);
  double dc0 = 0, dc1 = 0, ds0 = 0, ds1 = 0;
  for (int64_t i = 0; i < N; ++i) {
    const float* g = gf + i * 32;
    dc0 += -sinf(pcs[i * 4 + 0]) * g[1]; dc1 += -sinf(pcs[i * 4 + 1]) * g[2];
    ds0 += cosf(pcs[i * 4 + 2]) * g[3]; ds1 += cosf(pcs[i * 4 + 3]) * g[4];
    const int gi = bt->gender[i] == -1 ? c.vs_gender : bt->gender[i], si = bt->source[i] == -1 ? c.vs_source : bt->source[i];
    const int sti = bt->status[i] == -1 ? c.vs_status : bt->status[i];
    for (int j = 0; j < 4; ++j) { G[P_GENDER][gi * 4 + j] += g[5 + j]; G[P_SOURCE][si * 4 + j] += g[9 + j]; }
    for (int j = 0; j < 16; ++j) G[P_STATUS][sti * 16 + j] += g[15 + j];
  }
  ph.mark("embeddings backward");
  G[P_PCOS][0] = (float)dc0; G[P_PCOS][1] = (float)dc1; G[P_PSIN][0] = (float)ds0; G[P_PSIN][1] = (float)ds1;
  return 0;
}

// train.py:273 clip_grad_norm_(1.0) + train.py:285-298 fused AdamW over `n` tensors (decay[i] = 0.1 for dim >= 2, else 0); returns
// the gradient norm before clipping
double cpu_step_clip_adamw(int n, float* const* P, float* const* G, float* const* Mo, float* const* Vo, const int64_t* numel, const float* decay,
                           float lr, float b1, float b2, float eps, int step, float max_norm) {
  double ss = 0;
  for (int t = 0; t < n; ++t) {
    const float* g = G[t];
    double s = 0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (int64_t i = 0; i < numel[t]; ++i) s += (double)g[i] * g[i];
    ss += s;
  }
  const double norm = sqrt(ss);
  const float coef = (float)std::min(1.0, (double)max_norm / (norm + 1e-6));
  const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
  for (int t = 0; t < n; ++t) {
    float *p = P[t], *m = Mo[t], *v = Vo[t];
    const float* g = G[t];
    const float wd = decay[t];
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < numel[t]; ++i) {
      const float gi = g[i] * coef;
      const float mi = b1 * m[i] + (1.0f - b1) * gi, vi = b2 * v[i] + (1.0f - b2) * gi * gi;
      m[i] = mi; v[i] = vi;
      const float pi = p[i] * (1.0f - lr * wd);
      p[i] = pi - (lr / bc1) * (mi / (sqrtf(vi) / sqrtf(bc2) + eps));
    }
  }
  return norm;
}

// SGEMM alone (tests + the GFLOP/s figure quoted beside the baseline): C[M,N] = A[M,K] . B[N,K]^T
void cpu_step_sgemm_nt(int M, int N, int K, const float* A, const float* B, float* C) { gemm_nt(M, N, K, A, K, B, K, C, N); }
int cpu_step_isa(void) { return g_avx512 ? 512 : 256; }

}  // extern "C"
