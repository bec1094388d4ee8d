// Block-sparse bidirectional attention with the reference's two-predicate mask
// (transformer.model.py:479-487): allowed(q,kv) = userid[q]==userid[kv] AND
// (tmid[kv]==0 OR tmid[q]==tmid[kv]); GQA (q head h -> kv head h/(H/KV)); scores
// scaled by 1/sqrt(hd) (flex_attention default, model.py:278-285).
//
// gfx950 design: 64x64 (q x kv) tiles, 4 waves per workgroup, each wave owns 16
// rows; every contraction is an MFMA 16x16 product whose operands are read as
// contiguous rows from LDS -- the QKV / dO GEMM epilogues also emit per-head
// transposed copies (Q^T,K^T,V^T,dO^T: [b][head][hd][T]) so no in-kernel
// transpose is needed.  Tiles with no allowed pair are skipped through per-row
// tile bitmaps built once per step (the reference rebuilds a dense block mask every
// step, model.py:488-490).  Backward is two kernels (dK/dV per kv tile, dQ per q
// tile): no float atomics, bitwise reproducible.
#include "kernels.hpp"

namespace rsys {

template <typename T> struct AMma;
template <> struct AMma<bf16> {
  static constexpr int KS = 32;
  static constexpr int PAD = 8;
  using Frag = bf16x8;
  static __device__ __forceinline__ Frag zero() { Frag f; for (int i = 0; i < 8; ++i) f[i] = (bf16)0.f; return f; }
  // fragment of a row-major tile: row = row0 + (l&15), k = k0 + 8*(l>>4) .. +7
  static __device__ __forceinline__ Frag lds(const bf16* tile, int ld, int row0, int k0, int l) {
    return *(const bf16x8*)(tile + (row0 + (l & 15)) * ld + k0 + 8 * (l >> 4));
  }
  static __device__ __forceinline__ Frag glb(const bf16* rowptr, int k0, int kmax, int l) {
    int k = k0 + 8 * (l >> 4);
    if (k < kmax) return *(const bf16x8*)(rowptr + k);
    return zero();
  }
  static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct AMma<float> {
  static constexpr int KS = 4;
  static constexpr int PAD = 4;
  using Frag = float;
  static __device__ __forceinline__ Frag zero() { return 0.f; }
  static __device__ __forceinline__ Frag lds(const float* tile, int ld, int row0, int k0, int l) {
    return tile[(row0 + (l & 15)) * ld + k0 + (l >> 4)];
  }
  static __device__ __forceinline__ Frag glb(const float* rowptr, int k0, int kmax, int l) {
    int k = k0 + (l >> 4);
    return k < kmax ? rowptr[k] : 0.f;
  }
  static __device__ __forceinline__ f32x4 mma(Frag a, Frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
};

template <typename T, int HD> struct ACfg {
  static constexpr int KS = AMma<T>::KS;
  static constexpr int HDP = HD > KS ? HD : KS;      // contraction over d padded to one k-step
  static constexpr int LDD = HDP + AMma<T>::PAD;      // [64 tokens][HDP] tiles
  static constexpr int LDT = 64 + AMma<T>::PAD;       // [HD][64 tokens] tiles and P tiles
  static constexpr int E = 16 / sizeof(T);
};

// stage a [64 tokens][HD] tile (row-major source, row stride ld) into LDS [64][LDD]
template <typename T, int HD>
__device__ __forceinline__ void stage_rows(T* dst, const T* src, long long ld, int t, int nvalid) {
  using C = ACfg<T, HD>;
  constexpr int CPR = HD / C::E;
  for (int c = t; c < 64 * CPR; c += 256) {
    int row = c / CPR, ch = c % CPR;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < nvalid) v = *(const uint4*)(src + row * ld + ch * C::E);
    *(uint4*)(dst + row * C::LDD + ch * C::E) = v;
  }
}
// stage a [HD][64 tokens] tile from a transposed copy (row stride T tokens) into LDS [HD][LDT]
template <typename T, int HD>
__device__ __forceinline__ void stage_trans(T* dst, const T* src, long long ldT, int t, int nvalid) {
  using C = ACfg<T, HD>;
  constexpr int CPR = 64 / C::E;
  for (int c = t; c < HD * CPR; c += 256) {
    int row = c / CPR, ch = c % CPR;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ch * C::E < nvalid) v = *(const uint4*)(src + row * ldT + ch * C::E);   // T % 8 == 0: chunks never straddle T
    *(uint4*)(dst + row * C::LDT + ch * C::E) = v;
  }
}
// tokens past the end of a row (T % 64 != 0) get ids that match nothing
#define SENT_Q (-2147483647)
#define SENT_K (-2147483646)
template <typename T, int HD>
__device__ __forceinline__ void zero_pad_cols(T* dst, int t) {
  using C = ACfg<T, HD>;
  if constexpr (C::HDP > HD) {
    for (int c = t; c < 64 * (C::HDP - HD); c += 256) {
      int row = c / (C::HDP - HD), col = HD + c % (C::HDP - HD);
      dst[row * C::LDD + col] = from_f32<T>(0.f);
    }
  }
}

__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ------------------------------------------------------------------------ tile maps
__global__ __launch_bounds__(256) void attn_tilemap_kernel(AttnParams p) {
  __shared__ int uq[64], tq[64];
  __shared__ unsigned int bits;
  const int qt = blockIdx.x, b = blockIdx.y, t = threadIdx.x;
  const int nt = (p.T + 63) / 64;
  const long long base = (long long)b * p.T;
  if (t < 64) {
    const bool v = qt * 64 + t < p.T;
    uq[t] = v ? p.uid[base + qt * 64 + t] : SENT_Q; tq[t] = v ? p.tm[base + qt * 64 + t] : 0;
  }
  if (t == 0) bits = 0u;
  __syncthreads();
  for (int kv = t; kv < p.T; kv += 256) {
    const int uk = p.uid[base + kv], tk = p.tm[base + kv];
    bool any = false;
    for (int i = 0; i < 64; ++i) any |= (uq[i] == uk) && (tk == 0 || tq[i] == tk);
    if (any) atomicOr(&bits, 1u << (kv >> 6));
  }
  __syncthreads();
  if (t == 0) p.qmap[b * nt + qt] = bits;
  if (t < nt && ((bits >> t) & 1u)) atomicOr(&p.kmap[b * nt + t], 1u << qt);
}

int launch_attn_tilemap(const AttnParams& p, hipStream_t s) {
  ARG_CHECK(p.T % 8 == 0 && (p.T + 63) / 64 <= 32, "attention: T must be a multiple of 8 and <= 2048");
  HIP_CHECK(hipMemsetAsync(p.kmap, 0, sizeof(unsigned int) * p.B * ((p.T + 63) / 64), s));
  hipLaunchKernelGGL(attn_tilemap_kernel, dim3((p.T + 63) / 64, p.B), dim3(256), 0, s, p);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

// ------------------------------------------------------------------------ forward
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
  using C = ACfg<T, HD>;
  using M = AMma<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ks = (T*)smem_raw;                       // [64][LDD]
  T* Vt = Ks + 64 * C::LDD;                   // [HD][LDT]
  T* Ps = Vt + HD * C::LDT;                   // [4][16][LDT]
  int* uk = (int*)(Ps + 4 * 16 * C::LDT);     // [64]
  int* tk = uk + 64;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int t = threadIdx.x, l = t & 63, w = t >> 6, fq = l >> 4, fr = l & 15;
  const int kvh = h / (p.H / p.KV);
  const int nt = (p.T + 63) / 64;
  const long long tok0 = (long long)b * p.T;
  const float scale = rsqrtf((float)HD);
  const int qr0 = qt * 64 + w * 16;
  constexpr int NQS = C::HDP / C::KS;
  typename M::Frag qf[NQS];
  {
    const T* qrow = (const T*)p.q + (tok0 + min(qr0 + fr, p.T - 1)) * p.ld + h * HD;
#pragma unroll
    for (int s = 0; s < NQS; ++s) qf[s] = M::glb(qrow, s * C::KS, HD, l);
  }
  int uq[4], tq[4];
  float mrow[4], lrow[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int qi = qr0 + 4 * fq + r;
    uq[r] = qi < p.T ? p.uid[tok0 + qi] : SENT_Q;
    tq[r] = qi < p.T ? p.tm[tok0 + qi] : 0;
    mrow[r] = -1e30f; lrow[r] = 0.f;
  }
  f32x4 oacc[HD / 16];
#pragma unroll
  for (int j = 0; j < HD / 16; ++j) oacc[j] = f32x4{0, 0, 0, 0};
  zero_pad_cols<T, HD>(Ks, t);
  T* Pw = Ps + w * 16 * C::LDT;
  const unsigned int bits = p.qmap[b * nt + qt];
  for (int kt = 0; kt < nt; ++kt) {
    if (!((bits >> kt) & 1u)) continue;
    __syncthreads();
    const int nv = min(64, p.T - kt * 64);
    stage_rows<T, HD>(Ks, (const T*)p.k + (tok0 + kt * 64) * p.ld + kvh * HD, p.ld, t, nv);
    stage_trans<T, HD>(Vt, (const T*)p.vT + ((long long)(b * p.KV + kvh) * HD) * p.T + kt * 64, p.T, t, nv);
    if (t < 64) { uk[t] = t < nv ? p.uid[tok0 + kt * 64 + t] : SENT_K; tk[t] = t < nv ? p.tm[tok0 + kt * 64 + t] : 0; }
    __syncthreads();
    f32x4 sacc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sacc[j] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < NQS; ++s) sacc[j] = M::mma(qf[s], M::lds(Ks, C::LDD, j * 16, s * C::KS, l), sacc[j]);
    }
    bool ok[4][4];
    float rmax[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ukv = uk[j * 16 + fr], tkv = tk[j * 16 + fr];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ok[j][r] = (uq[r] == ukv) && (tkv == 0 || tq[r] == tkv);
        float sv = ok[j][r] ? sacc[j][r] * scale : -1e30f;
        sacc[j][r] = sv;
        rmax[r] = fmaxf(rmax[r], sv);
      }
    }
    float alpha[4], rsum[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mx = group16_max(rmax[r]);
      float mn = fmaxf(mrow[r], mx);
      alpha[r] = __expf(mrow[r] - mn);
      mrow[r] = mn;
      rsum[r] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pv = ok[j][r] ? __expf(sacc[j][r] - mrow[r]) : 0.f;
        rsum[r] += pv;
        Pw[(4 * fq + r) * C::LDT + j * 16 + fr] = from_f32<T>(pv);
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      lrow[r] = lrow[r] * alpha[r] + group16_sum(rsum[r]);
#pragma unroll
      for (int j = 0; j < HD / 16; ++j) oacc[j][r] *= alpha[r];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 64 / C::KS; ++s) {
      typename M::Frag pf = M::lds(Pw, C::LDT, 0, s * C::KS, l);
#pragma unroll
      for (int j = 0; j < HD / 16; ++j) oacc[j] = M::mma(pf, M::lds(Vt, C::LDT, j * 16, s * C::KS, l), oacc[j]);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (qr0 + 4 * fq + r >= p.T) continue;
    const long long row = tok0 + qr0 + 4 * fq + r;
    const float inv = 1.f / lrow[r];
#pragma unroll
    for (int j = 0; j < HD / 16; ++j) ((T*)p.o)[row * p.ldo + h * HD + j * 16 + fr] = from_f32<T>(oacc[j][r] * inv);
    if (fr == 0) p.lse[((long long)b * p.H + h) * p.T + qr0 + 4 * fq + r] = mrow[r] + logf(lrow[r]);
  }
}

template <typename T, int HD>
static size_t fwd_smem() {
  using C = ACfg<T, HD>;
  return sizeof(T) * (64 * C::LDD + HD * C::LDT + 4 * 16 * C::LDT) + 128 * sizeof(int);
}

template <typename T, int HD>
static int attn_fwd_hd(const AttnParams& p, hipStream_t s) {
  size_t sm = fwd_smem<T, HD>();
  static bool set = false;
  if (!set) {
    HIP_CHECK(hipFuncSetAttribute((const void*)attn_fwd_kernel<T, HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    set = true;
  }
  hipLaunchKernelGGL((attn_fwd_kernel<T, HD>), dim3((p.T + 63) / 64, p.H, p.B), dim3(256), sm, s, p);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

static int check_attn(const AttnParams& p, size_t esz) {
  ARG_CHECK(p.T % 8 == 0 && (p.T + 63) / 64 <= 32, "attention: T must be a multiple of 8 and <= 2048");
  ARG_CHECK(p.H % p.KV == 0, "attention: H % KV");
  ARG_CHECK((p.ld * esz) % 16 == 0 && (p.hd * esz) % 16 == 0, "attention: 16-byte row alignment");
  return RSYS_OK;
}

template <typename T>
int launch_attn_fwd(const AttnParams& p, hipStream_t s) {
  int rc = check_attn(p, sizeof(T));
  if (rc) return rc;
  switch (p.hd) {
    case 16: return attn_fwd_hd<T, 16>(p, s);
    case 32: return attn_fwd_hd<T, 32>(p, s);
    case 64: return attn_fwd_hd<T, 64>(p, s);
    case 128: return attn_fwd_hd<T, 128>(p, s);
  }
  set_error("attention: head_dim must be 16, 32, 64 or 128");
  return RSYS_ERR_ARG;
}
template int launch_attn_fwd<bf16>(const AttnParams&, hipStream_t);
template int launch_attn_fwd<float>(const AttnParams&, hipStream_t);

// ------------------------------------------------------------------------ delta = rowsum(dO * O) per head
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnParams p) {
  // one wave per token row; each lane owns 16-byte chunks; lanes of one head reduce with shuffles
  constexpr int E = 16 / sizeof(T);
  const long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int l = threadIdx.x & 63;
  if (row >= (long long)p.B * p.T) return;
  const int D = p.H * p.hd;
  const int lph = p.hd / E;                 // lanes per head (2..32, power of two)
  const long long b = row / p.T; const int t = (int)(row % p.T);
  for (int c0 = 0; c0 < D / E; c0 += 64) {
    const int c = c0 + l;
    float acc = 0.f;
    if (c < D / E) {
      uint4 ra = *(const uint4*)((const T*)p.dO + row * p.ldo + c * E);
      uint4 rb = *(const uint4*)((const T*)p.o + row * p.ldo + c * E);
      const T* a = (const T*)&ra; const T* o = (const T*)&rb;
#pragma unroll
      for (int k = 0; k < E; ++k) acc += to_f32(a[k]) * to_f32(o[k]);
    }
    for (int off = 1; off < lph; off <<= 1) acc += __shfl_xor(acc, off, 64);
    if (c < D / E && (l & (lph - 1)) == 0) {
      const int h = (c * E) / p.hd;
      ((float*)p.delta)[(b * p.H + h) * p.T + t] = acc;
    }
  }
}
template <typename T>
int launch_attn_delta(const AttnParams& p, hipStream_t s) {
  long long rows = (long long)p.B * p.T;
  hipLaunchKernelGGL((attn_delta_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, p);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}
template int launch_attn_delta<bf16>(const AttnParams&, hipStream_t);
template int launch_attn_delta<float>(const AttnParams&, hipStream_t);

// inverse rotation of an accumulator tile (rows = tokens, cols = d on the lane), transformer.model.py:182-190 transposed
__device__ __forceinline__ float rope_inv(float x, float c, float s, int l) {
  float partner = __shfl_xor(x, 1, 64);
  return (l & 1) ? (x * c - partner * s) : (x * c + partner * s);
}

// ------------------------------------------------------------------------ backward: dK, dV (one workgroup per kv tile and kv head)
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(AttnParams p) {
  using C = ACfg<T, HD>;
  using M = AMma<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Qs = (T*)smem_raw;                 // [64 q][LDD]
  T* dOs = Qs + 64 * C::LDD;            // [64 q][LDD]
  T* QTs = dOs + 64 * C::LDD;           // [HD][LDT]
  T* dOTs = QTs + HD * C::LDT;          // [HD][LDT]
  T* PTs = dOTs + HD * C::LDT;          // [4][16 kv][LDT]
  T* dSTs = PTs + 4 * 16 * C::LDT;      // [4][16 kv][LDT]
  float* lse_q = (float*)(dSTs + 4 * 16 * C::LDT);
  float* del_q = lse_q + 64;
  int* uqs = (int*)(del_q + 64);
  int* tqs = uqs + 64;
  const int kvt = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
  const int t = threadIdx.x, l = t & 63, w = t >> 6, fq = l >> 4, fr = l & 15;
  const int rep = p.H / p.KV, nt = (p.T + 63) / 64;
  const long long tok0 = (long long)b * p.T;
  const float scale = rsqrtf((float)HD);
  const int kr0 = kvt * 64 + w * 16;
  constexpr int NDS = C::HDP / C::KS;
  typename M::Frag kf[NDS], vf[NDS];
  {
    const T* krow = (const T*)p.k + (tok0 + min(kr0 + fr, p.T - 1)) * p.ld + kvh * HD;
    const T* vrow = (const T*)p.v + (tok0 + min(kr0 + fr, p.T - 1)) * p.ld + kvh * HD;
#pragma unroll
    for (int s = 0; s < NDS; ++s) { kf[s] = M::glb(krow, s * C::KS, HD, l); vf[s] = M::glb(vrow, s * C::KS, HD, l); }
  }
  int ukv[4], tkv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int ki = kr0 + 4 * fq + r;
    ukv[r] = ki < p.T ? p.uid[tok0 + ki] : SENT_K; tkv[r] = ki < p.T ? p.tm[tok0 + ki] : 0;
  }
  f32x4 dK[HD / 16], dV[HD / 16];
#pragma unroll
  for (int j = 0; j < HD / 16; ++j) { dK[j] = f32x4{0, 0, 0, 0}; dV[j] = f32x4{0, 0, 0, 0}; }
  zero_pad_cols<T, HD>(Qs, t);
  zero_pad_cols<T, HD>(dOs, t);
  T* PTw = PTs + w * 16 * C::LDT;
  T* dSTw = dSTs + w * 16 * C::LDT;
  const unsigned int bits = p.kmap[b * nt + kvt];
  for (int hh = 0; hh < rep; ++hh) {
    const int h = kvh * rep + hh;
    for (int qt = 0; qt < nt; ++qt) {
      if (!((bits >> qt) & 1u)) continue;
      __syncthreads();
      const int nv = min(64, p.T - qt * 64);
      stage_rows<T, HD>(Qs, (const T*)p.q + (tok0 + qt * 64) * p.ld + h * HD, p.ld, t, nv);
      stage_rows<T, HD>(dOs, (const T*)p.dO + (tok0 + qt * 64) * p.ldo + h * HD, p.ldo, t, nv);
      stage_trans<T, HD>(QTs, (const T*)p.qT + ((long long)(b * p.H + h) * HD) * p.T + qt * 64, p.T, t, nv);
      stage_trans<T, HD>(dOTs, (const T*)p.dOT + ((long long)(b * p.H + h) * HD) * p.T + qt * 64, p.T, t, nv);
      if (t < 64) {
        const bool v = t < nv;
        lse_q[t] = v ? p.lse[((long long)b * p.H + h) * p.T + qt * 64 + t] : 0.f;
        del_q[t] = v ? p.delta[((long long)b * p.H + h) * p.T + qt * 64 + t] : 0.f;
        uqs[t] = v ? p.uid[tok0 + qt * 64 + t] : SENT_Q;
        tqs[t] = v ? p.tm[tok0 + qt * 64 + t] : 0;
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 sacc = f32x4{0, 0, 0, 0}, dpacc = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < NDS; ++s) {
          sacc = M::mma(kf[s], M::lds(Qs, C::LDD, j * 16, s * C::KS, l), sacc);
          dpacc = M::mma(vf[s], M::lds(dOs, C::LDD, j * 16, s * C::KS, l), dpacc);
        }
        const int qc = j * 16 + fr;
        const float lse = lse_q[qc], dl = del_q[qc];
        const int uqv = uqs[qc], tqv = tqs[qc];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          bool ok = (uqv == ukv[r]) && (tkv[r] == 0 || tqv == tkv[r]);
          float pv = ok ? __expf(sacc[r] * scale - lse) : 0.f;
          float ds = pv * (dpacc[r] - dl) * scale;
          PTw[(4 * fq + r) * C::LDT + qc] = from_f32<T>(pv);
          dSTw[(4 * fq + r) * C::LDT + qc] = from_f32<T>(ds);
        }
      }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < 64 / C::KS; ++s) {
        typename M::Frag pf = M::lds(PTw, C::LDT, 0, s * C::KS, l);
        typename M::Frag df = M::lds(dSTw, C::LDT, 0, s * C::KS, l);
#pragma unroll
        for (int j = 0; j < HD / 16; ++j) {
          dV[j] = M::mma(pf, M::lds(dOTs, C::LDT, j * 16, s * C::KS, l), dV[j]);
          dK[j] = M::mma(df, M::lds(QTs, C::LDT, j * 16, s * C::KS, l), dK[j]);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int tk = kr0 + 4 * fq + r;
    const bool rv = tk < p.T;
    const long long row = tok0 + (rv ? tk : p.T - 1);
    const int pos = p.rope_pos ? p.rope_pos[row] : (rv ? tk : 0);
#pragma unroll
    for (int j = 0; j < HD / 16; ++j) {
      const int d = j * 16 + fr;
      const float c = p.rope_cos[pos * (HD / 2) + (d >> 1)], sn = p.rope_sin[pos * (HD / 2) + (d >> 1)];
      float gk = rope_inv(dK[j][r], c, sn, l);
      if (rv) {
        ((T*)p.dk)[row * p.ldg + kvh * HD + d] = from_f32<T>(gk);
        ((T*)p.dv)[row * p.ldg + kvh * HD + d] = from_f32<T>(dV[j][r]);
      }
    }
  }
}

// ------------------------------------------------------------------------ backward: dQ (one workgroup per q tile and head)
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(AttnParams p) {
  using C = ACfg<T, HD>;
  using M = AMma<T>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Ks = (T*)smem_raw;                 // [64 kv][LDD]
  T* Vs = Ks + 64 * C::LDD;             // [64 kv][LDD]
  T* KTs = Vs + 64 * C::LDD;            // [HD][LDT]
  T* dSs = KTs + HD * C::LDT;           // [4][16 q][LDT]
  int* uk = (int*)(dSs + 4 * 16 * C::LDT);
  int* tk = uk + 64;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int t = threadIdx.x, l = t & 63, w = t >> 6, fq = l >> 4, fr = l & 15;
  const int kvh = h / (p.H / p.KV), nt = (p.T + 63) / 64;
  const long long tok0 = (long long)b * p.T;
  const float scale = rsqrtf((float)HD);
  const int qr0 = qt * 64 + w * 16;
  constexpr int NDS = C::HDP / C::KS;
  typename M::Frag qf[NDS], dof[NDS];
  {
    const T* qrow = (const T*)p.q + (tok0 + min(qr0 + fr, p.T - 1)) * p.ld + h * HD;
    const T* drow = (const T*)p.dO + (tok0 + min(qr0 + fr, p.T - 1)) * p.ldo + h * HD;
#pragma unroll
    for (int s = 0; s < NDS; ++s) { qf[s] = M::glb(qrow, s * C::KS, HD, l); dof[s] = M::glb(drow, s * C::KS, HD, l); }
  }
  int uq[4], tq[4];
  float lse[4], dl[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int tq_i = qr0 + 4 * fq + r;
    const bool v = tq_i < p.T;
    uq[r] = v ? p.uid[tok0 + tq_i] : SENT_Q; tq[r] = v ? p.tm[tok0 + tq_i] : 0;
    lse[r] = v ? p.lse[((long long)b * p.H + h) * p.T + tq_i] : 0.f;
    dl[r] = v ? p.delta[((long long)b * p.H + h) * p.T + tq_i] : 0.f;
  }
  f32x4 dQ[HD / 16];
#pragma unroll
  for (int j = 0; j < HD / 16; ++j) dQ[j] = f32x4{0, 0, 0, 0};
  zero_pad_cols<T, HD>(Ks, t);
  zero_pad_cols<T, HD>(Vs, t);
  T* dSw = dSs + w * 16 * C::LDT;
  const unsigned int bits = p.qmap[b * nt + qt];
  for (int kt = 0; kt < nt; ++kt) {
    if (!((bits >> kt) & 1u)) continue;
    __syncthreads();
    const int nv = min(64, p.T - kt * 64);
    stage_rows<T, HD>(Ks, (const T*)p.k + (tok0 + kt * 64) * p.ld + kvh * HD, p.ld, t, nv);
    stage_rows<T, HD>(Vs, (const T*)p.v + (tok0 + kt * 64) * p.ld + kvh * HD, p.ld, t, nv);
    stage_trans<T, HD>(KTs, (const T*)p.kT + ((long long)(b * p.KV + kvh) * HD) * p.T + kt * 64, p.T, t, nv);
    if (t < 64) { uk[t] = t < nv ? p.uid[tok0 + kt * 64 + t] : SENT_K; tk[t] = t < nv ? p.tm[tok0 + kt * 64 + t] : 0; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 sacc = f32x4{0, 0, 0, 0}, dpacc = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int s = 0; s < NDS; ++s) {
        sacc = M::mma(qf[s], M::lds(Ks, C::LDD, j * 16, s * C::KS, l), sacc);
        dpacc = M::mma(dof[s], M::lds(Vs, C::LDD, j * 16, s * C::KS, l), dpacc);
      }
      const int kc = j * 16 + fr;
      const int ukv = uk[kc], tkv = tk[kc];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bool ok = (uq[r] == ukv) && (tkv == 0 || tq[r] == tkv);
        float pv = ok ? __expf(sacc[r] * scale - lse[r]) : 0.f;
        float ds = pv * (dpacc[r] - dl[r]) * scale;
        dSw[(4 * fq + r) * C::LDT + kc] = from_f32<T>(ds);
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 64 / C::KS; ++s) {
      typename M::Frag df = M::lds(dSw, C::LDT, 0, s * C::KS, l);
#pragma unroll
      for (int j = 0; j < HD / 16; ++j) dQ[j] = M::mma(df, M::lds(KTs, C::LDT, j * 16, s * C::KS, l), dQ[j]);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int tq_i = qr0 + 4 * fq + r;
    const bool rv = tq_i < p.T;
    const long long row = tok0 + (rv ? tq_i : p.T - 1);
    const int pos = p.rope_pos ? p.rope_pos[row] : (rv ? tq_i : 0);
#pragma unroll
    for (int j = 0; j < HD / 16; ++j) {
      const int d = j * 16 + fr;
      const float c = p.rope_cos[pos * (HD / 2) + (d >> 1)], sn = p.rope_sin[pos * (HD / 2) + (d >> 1)];
      const float gq = rope_inv(dQ[j][r], c, sn, l);
      if (rv) ((T*)p.dq)[row * p.ldg + h * HD + d] = from_f32<T>(gq);
    }
  }
}

template <typename T, int HD>
static int attn_bwd_hd(const AttnParams& p, hipStream_t s) {
  using C = ACfg<T, HD>;
  size_t sm_kv = sizeof(T) * (2 * 64 * C::LDD + 2 * HD * C::LDT + 8 * 16 * C::LDT) + 256 * 4;
  size_t sm_q = sizeof(T) * (2 * 64 * C::LDD + HD * C::LDT + 4 * 16 * C::LDT) + 128 * 4;
  static bool set = false;
  if (!set) {
    HIP_CHECK(hipFuncSetAttribute((const void*)attn_bwd_kv_kernel<T, HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_kv));
    HIP_CHECK(hipFuncSetAttribute((const void*)attn_bwd_q_kernel<T, HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm_q));
    set = true;
  }
  hipLaunchKernelGGL((attn_bwd_kv_kernel<T, HD>), dim3((p.T + 63) / 64, p.KV, p.B), dim3(256), sm_kv, s, p);
  HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL((attn_bwd_q_kernel<T, HD>), dim3((p.T + 63) / 64, p.H, p.B), dim3(256), sm_q, s, p);
  HIP_CHECK(hipGetLastError());
  return RSYS_OK;
}

template <typename T>
int launch_attn_bwd(const AttnParams& p, hipStream_t s) {
  int rc = check_attn(p, sizeof(T));
  if (rc) return rc;
  switch (p.hd) {
    case 16: return attn_bwd_hd<T, 16>(p, s);
    case 32: return attn_bwd_hd<T, 32>(p, s);
    case 64: return attn_bwd_hd<T, 64>(p, s);
    case 128: return attn_bwd_hd<T, 128>(p, s);
  }
  set_error("attention: head_dim must be 16, 32, 64 or 128");
  return RSYS_ERR_ARG;
}
template int launch_attn_bwd<bf16>(const AttnParams&, hipStream_t);
template int launch_attn_bwd<float>(const AttnParams&, hipStream_t);

}  // namespace rsys
