// attention_fp8.hip — MXFP8 attention forward for long sequences (BASELINE config 5: 1024 protein sites).
//
// O = softmax(scale * Q K^T) V with BOTH matrix products on v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 operands, E8M0
// block scales, fp32 accumulate: twice the bf16 MFMA rate and half the LDS / HBM bytes per operand), softmax statistics
// and the running output in fp32.  Same problem / segment addressing as dl_attn_fwd (include/druglamp_hip.h).
//
// Two steps, both launched by dl_attn_fwd_fp8 into a caller-provided workspace:
//   1. quantisation (attn_quant_rows_kernel, attn_quant_vt_kernel): bf16 Q / K / V -> e4m3 bytes + one E8M0 scale per
//      32-element block (scale = 2^(floor(log2 amax) - 7), so block values land in [128, 256) at the top).  The block
//      partition follows what the instruction does, measured by tools/micro/mfma_f8_probe.hip: lane (row, h = lane >> 5)
//      supplies 32 bytes of a 64-byte row group, and the scale byte of lane-half h applies to bytes 16h..16h+15 of BOTH
//      lane halves.  Rows therefore keep their natural element order (byte m of a 64-byte group = element m) and block
//      b of a group is elements {16b..16b+15} U {32+16b..32+16b+15}.
//      V is stored TRANSPOSED per 64-key tile, [head_dim][64 bytes], with the keys of a tile permuted so that the 32
//      bytes a lane reads are exactly the keys whose probabilities that lane holds after the score MFMAs (below): no
//      LDS transpose read, no cross-lane traffic for P.
//   2. attn_fwd_fp8_kernel: S^T = K Q^T per 32-key x 32-query tile (A = K rows from LDS, B = Q kept in registers);
//      the 32x32 accumulator layout gives lane (q = lane & 31, h = lane >> 5) the scores of query q against keys
//      (r & 3) + 8 (r >> 2) + 4 h, r = 0..15, of each tile, so per-query statistics are a register reduction plus ONE
//      cross-half exchange; P^T (e4m3, scale 2^-5) is built in registers straight from the accumulators and is the B
//      operand of O^T = V^T P^T.  K / V^T tiles of 64 keys stream through two LDS stages by LDS-DMA (4 KB per operand and
//      tile at head_dim 64: one 16-byte DMA per thread), one barrier per tile.  The running maximum is only raised when a
//      tile exceeds it by more than 2^3 (then, and only then, the output accumulators are rescaled): P stays <= 8, which
//      times the 2^5 operand scale fits e4m3.
#include "tiles.cuh"

namespace {
using namespace dltile;

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int F8_KVB = 64;            // keys per streamed tile
constexpr int F8_PSHIFT = 5;          // P is quantised as p * 2^5 (<= 256), operand scale 2^-5
constexpr float F8_TAU = 3.0f;        // lazy-rescale threshold in the log2 domain (p <= 2^3)

struct AttnF8P {
  const uint8_t *Q8, *K8, *V8T, *sQ, *sK, *sV;
  char* Out; float* LSE;
  int64_t o_ps, o_hs, o_rs, o_ss;
  int P, H, S, shift, Lq, Lk, LqP, LkP, NT;
  float scale;
};

struct QuantP {
  const char* X; int64_t ps, hs, rs;      // bf16 source, element strides
  uint8_t* X8; uint8_t* sX;               // [P][LP][H][HD], [P][LP][H][HD/32]   (rows)   or V8T / sV (transposed)
  int P, H, L, LP, NT;
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// E8M0 byte for a block with maximum magnitude amax: values / 2^(byte - 127) land in [128, 256) for the largest
__device__ __forceinline__ uint32_t e8m0_for(float amax) {
  const int be = (int)((__builtin_bit_cast(uint32_t, amax) >> 23) & 0xffu);       // biased exponent of amax
  int sb = be - 7;
  return (uint32_t)(sb < 1 ? 1 : (sb > 253 ? 253 : sb));
}
__device__ __forceinline__ float inv_scale_of(uint32_t sb) { return __builtin_bit_cast(float, (254u - sb) << 23); }

// ---- quantisation of row operands (Q, K): one thread per 16-byte chunk (8 bf16) of a row ----------------------------
// Within a 64-element group (8 consecutive lanes, chunk c = lane & 7) block b = (c >> 1) & 1 holds chunks {2b, 2b+1, 4+2b,
// 5+2b} (elements 16b..16b+15 and 32+16b..32+16b+15): the block maximum is two lane exchanges (c ^ 1, c ^ 4).  Loads are
// 1 KB and stores 512 B contiguous per wave.
template <int HD>
__global__ void attn_quant_rows_kernel(const QuantP p) {
  constexpr int CPRW = HD / 8;                       // 16-byte bf16 chunks per row
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)p.P * p.H * p.LP * CPRW;
  const bool live = idx < total;
  const int64_t id = live ? idx : total - 1;         // (all lanes take part in the exchanges)
  const int j = (int)(id % CPRW);
  int64_t t = id / CPRW;
  const int h = (int)(t % p.H); t /= p.H;            // heads fastest: a wave walks contiguous source memory
  const int row = (int)(t % p.LP);
  const int pr = (int)(t / p.LP);
  const int c = j & 7, kf = j >> 3, b = (c >> 1) & 1;
  f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = {0.f, 0.f, 0.f, 0.f};
  if (row < p.L) {
    const bf16_t* src = reinterpret_cast<const bf16_t*>(p.X) + (int64_t)pr * p.ps + (int64_t)h * p.hs + (int64_t)row * p.rs + j * 8;
    const u32x4 w = *reinterpret_cast<const u32x4*>(src);
    x0 = f32x4{bf16lo(w[0]), bf16hi(w[0]), bf16lo(w[1]), bf16hi(w[1])};
    x1 = f32x4{bf16lo(w[2]), bf16hi(w[2]), bf16lo(w[3]), bf16hi(w[3])};
  }
  float amax = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(x0[e]), fabsf(x1[e])));
  amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
  amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
  const uint32_t sb = row < p.L ? e8m0_for(amax) : 127u;
  const float inv = inv_scale_of(sb);
  int w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x0[0] * inv, x0[1] * inv, w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(x0[2] * inv, x0[3] * inv, w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x1[0] * inv, x1[1] * inv, w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(x1[2] * inv, x1[3] * inv, w1, true);
  if (!live) return;
  const int64_t rbase = ((int64_t)pr * p.LP + row) * p.H + h;      // [P][LP][H][HD]: rows of one head are H * HD bytes apart
  *reinterpret_cast<u32x2*>(p.X8 + rbase * HD + j * 8) = u32x2{(uint32_t)w0, (uint32_t)w1};
  if ((c & 5) == 0) p.sX[rbase * (HD / 32) + kf * 2 + b] = (uint8_t)sb;
}

// ---- quantisation of V, transposed and key-permuted: one workgroup per (problem, head, 64-key tile) ------------------
// V8T[(tile*HD + d)*64 + h*32 + kt*16 + r] = V[tile*64 + kt*32 + (r & 3) + 8 (r >> 2) + 4 h][d];  block kt = keys kt*32..+31.
// The 64 x HD bf16 tile is read with coalesced 16-byte loads into LDS; item (d, kt) then walks its 32 keys down a column.
template <int HD>
__global__ __launch_bounds__(256) void attn_quant_vt_kernel(const QuantP p) {
  __shared__ __attribute__((aligned(16))) bf16_t tile_s[F8_KVB][HD + 8];     // +8: column walks of adjacent d stay conflict-free
  int64_t t = blockIdx.x;
  const int tile = (int)(t % p.NT); t /= p.NT;
  const int h = (int)(t % p.H);
  const int pr = (int)(t / p.H);
  const bf16_t* src = reinterpret_cast<const bf16_t*>(p.X) + (int64_t)pr * p.ps + (int64_t)h * p.hs;
  constexpr int CPRW = HD / 8;
  for (int c = threadIdx.x; c < F8_KVB * CPRW; c += 256) {
    const int row = c / CPRW, ch = c % CPRW, key = tile * F8_KVB + row;
    u32x4 w = {0u, 0u, 0u, 0u};
    if (key < p.L) w = *reinterpret_cast<const u32x4*>(src + (int64_t)key * p.rs + ch * 8);
    *reinterpret_cast<u32x4*>(&tile_s[row][ch * 8]) = w;
  }
  __syncthreads();
  for (int item = threadIdx.x; item < 2 * HD; item += 256) {
    const int d = item % HD, kt = item / HD;
    float v[32];                                     // v[hh*16 + r]
    float amax = 0.f;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = to_f32(tile_s[kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh][d]);
        v[hh * 16 + r] = x;
        amax = fmaxf(amax, fabsf(x));
      }
    const uint32_t sb = e8m0_for(amax);
    const float inv = inv_scale_of(sb);
    uint8_t* base = p.X8 + ((((int64_t)pr * p.H + h) * p.NT + tile) * HD + d) * 64;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      u32x4 w;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        int word = 0;
        word = __builtin_amdgcn_cvt_pk_fp8_f32(v[hh * 16 + c * 4 + 0] * inv, v[hh * 16 + c * 4 + 1] * inv, word, false);
        word = __builtin_amdgcn_cvt_pk_fp8_f32(v[hh * 16 + c * 4 + 2] * inv, v[hh * 16 + c * 4 + 3] * inv, word, true);
        w[c] = (uint32_t)word;
      }
      *reinterpret_cast<u32x4*>(base + hh * 32 + kt * 16) = w;
    }
    p.sX[((((int64_t)pr * p.H + h) * p.NT + tile) * HD + d) * 2 + kt] = (uint8_t)sb;
  }
}

// ---- the attention kernel -------------------------------------------------------------------------------------------
template <int CPR> __device__ __forceinline__ int f8_swz(int row) {      // XOR of a row's 16-byte chunks (see file header)
  if constexpr (CPR == 4) return (row >> 2) & 3;
  else return (row >> 1) & 7;
}
// 32 bytes of row `row`, logical chunks c0 and c0 + 1 (16 bytes each), of a [rows][CPR * 16 bytes] swizzled tile
template <int CPR>
__device__ __forceinline__ i32x8 f8_frag(const char* tile, int row, int c0) {
  const int sw = f8_swz<CPR>(row);
  const u32x4 a = lds_read16(tile, (uint32_t)(row * CPR * 16 + (((c0) ^ sw) << 4)));
  const u32x4 b = lds_read16(tile, (uint32_t)(row * CPR * 16 + (((c0 + 1) ^ sw) << 4)));
  i32x8 r = {(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
  return r;
}
// LDS-DMA of one contiguous tile of ROWS x (CPR * 16) bytes into its swizzled image (256 threads)
template <int ROWS, int CPR>
__device__ __forceinline__ void f8_dma_tile(char* lds, const uint8_t* src, int64_t pitch) {
  constexpr int NCH = ROWS * CPR;
  const int tid = threadIdx.x, wave = tid >> 6;
#pragma unroll
  for (int c0 = 0; c0 < NCH; c0 += 256) {
    const int c = c0 + tid;
    const int row = c / CPR, ch = (c % CPR) ^ f8_swz<CPR>(row);
    const uint8_t* g = src + (int64_t)row * pitch + ch * 16;
    const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)((c0 + wave * 64) * 16));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(lds + off), 16, 0, 0);
  }
}

template <int HD, int QT>
__global__ __launch_bounds__(256, 2) void attn_fwd_fp8_kernel(const AttnF8P p) {
  constexpr int NKF = HD / 64;          // 64-element contraction groups of a Q / K row
  constexpr int NDT = HD / 32;          // 32-row tiles of O^T
  constexpr int CPRK = HD / 16;         // 16-byte chunks per K row
  constexpr int KT_BYTES = F8_KVB * HD, VT_BYTES = HD * 64, STAGE = KT_BYTES + VT_BYTES;
  constexpr int QB = 4 * QT * 32;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, il = lane & 31, hf = lane >> 5;
  const int bps = (p.Lq + QB - 1) / QB;
  const int seg = blockIdx.x / bps, qb = blockIdx.x % bps;
  const int h = blockIdx.y, pr = blockIdx.z;
  const int qprob = seg == 0 ? pr : (pr + p.shift) % p.P;
  // row operands are [problem][row][head][HD]: rows of one head H * HD bytes apart (scales: H * HD / 32)
  const int64_t rp = (int64_t)p.H * HD, sp = (int64_t)p.H * (HD / 32);
  const uint8_t* Q8 = p.Q8 + (int64_t)qprob * p.LqP * rp + h * HD;
  const uint8_t* sQ = p.sQ + (int64_t)qprob * p.LqP * sp + h * (HD / 32);
  const uint8_t* K8 = p.K8 + (int64_t)pr * p.LkP * rp + h * HD;
  const uint8_t* sK = p.sK + (int64_t)pr * p.LkP * sp + h * (HD / 32);
  const uint8_t* V8 = p.V8T + ((int64_t)pr * p.H + h) * p.NT * HD * 64;
  const uint8_t* sV = p.sV + ((int64_t)pr * p.H + h) * p.NT * HD * 2;
  const int qw0 = qb * QB + wave * QT * 32;

  // Q fragments (B operand: column = query il, 32 bytes of lane half hf per 64-element group) and their scales
  i32x8 qf[QT][NKF];
  int qs[QT][NKF];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = qw0 + qt * 32 + il;
    const bool ok = q < p.Lq;
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf) {
      u32x4 a = {0u, 0u, 0u, 0u}, b = {0u, 0u, 0u, 0u};
      int s = 127;
      if (ok) {
        const uint8_t* row = Q8 + (int64_t)q * rp + kf * 64 + hf * 32;
        a = *reinterpret_cast<const u32x4*>(row);
        b = *reinterpret_cast<const u32x4*>(row + 16);
        s = sQ[(int64_t)q * sp + kf * 2 + hf];
      }
      qf[qt][kf] = i32x8{(int)a[0], (int)a[1], (int)a[2], (int)a[3], (int)b[0], (int)b[1], (int)b[2], (int)b[3]};
      qs[qt][kf] = s;
    }
  }
  f32x16 o[QT][NDT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qt][dt][r] = 0.f;
  // Row sums ride the matrix pipe (which idles behind the exponentials): l^T = 1 P^T with an all-ones A operand
  // (e4m3 1.0 = 0x38) — every row of the 32x32 result is the sum over the tile's 64 keys of the QUANTISED p of query
  // il, so the normalisation is consistent with what multiplies V; register 0 is read at the end.
  f32x16 lacc[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt)
#pragma unroll
    for (int r = 0; r < 16; ++r) lacc[qt][r] = 0.f;
  const i32x8 ones = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838};
  float m_run[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) m_run[qt] = -INFINITY;
  const float c = p.scale * LOG2E;

  auto stage = [&](int t, int buf) {
    char* b = smem + buf * STAGE;
    f8_dma_tile<F8_KVB, CPRK>(b, K8 + (int64_t)t * F8_KVB * rp, rp);
    f8_dma_tile<HD, 4>(b + KT_BYTES, V8 + (int64_t)t * VT_BYTES, 64);
  };
  // scale bytes of a tile's fragments (tiny, L2-resident), fetched ONE TILE AHEAD so that no tile waits on them
  int ks[2][NKF], vs[NDT], ks_n[2][NKF], vs_n[NDT];
  auto load_scales = [&](int t, int (&k_)[2][NKF], int (&v_)[NDT]) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int kf = 0; kf < NKF; ++kf) k_[kt][kf] = sK[(int64_t)(t * F8_KVB + kt * 32 + il) * sp + kf * 2 + hf];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) v_[dt] = sV[((int64_t)t * HD + dt * 32 + il) * 2 + hf];
  };
  stage(0, 0);
  load_scales(0, ks, vs);
  for (int t = 0; t < p.NT; ++t) {
    const int k0 = t * F8_KVB;
    __syncthreads();                                    // tile t has landed; everyone is done with the other stage
    if (t + 1 < p.NT) { stage(t + 1, (t + 1) & 1); load_scales(t + 1, ks_n, vs_n); }
    const char* Ks = smem + (t & 1) * STAGE;
    const char* Vs = Ks + KT_BYTES;

    // ---- S^T = K Q^T: s[qt][kt][r] = score(key = k0 + kt*32 + (r&3) + 8(r>>2) + 4hf, query il) ----
    f32x16 s[QT][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[qt][kt][r] = 0.f;
#pragma unroll
      for (int kf = 0; kf < NKF; ++kf) {
        const i32x8 ka = f8_frag<CPRK>(Ks, kt * 32 + il, kf * 4 + hf * 2);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          s[qt][kt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ka, qf[qt][kf], s[qt][kt], 0, 0, 0, ks[kt][kf], 0,
                                                                       qs[qt][kf]);
      }
    }
    if (k0 + F8_KVB > p.Lk) {                           // keys past the end (zero rows of K8): no weight
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (k0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf >= p.Lk) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) s[qt][kt][r] = -INFINITY;
          }
    }
    // ---- online softmax with a lazily raised maximum ----
    float mx[QT];
    bool need = false;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float m = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, s[qt][kt][r]);
      m = fmaxf(m, __shfl_xor(m, 32, 64));              // the other half of the keys of this query
      mx[qt] = m;
      need = need || (m * c > m_run[qt] * c + F8_TAU) || (m_run[qt] == -INFINITY);
    }
    if (__builtin_amdgcn_ballot_w64(need) != 0ull) {     // wave-uniform: rare after the first tiles
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const bool up = (mx[qt] * c > m_run[qt] * c + F8_TAU) || (m_run[qt] == -INFINITY);
        const float m_new = up ? mx[qt] : m_run[qt];
        const float alpha = up ? fast_exp2((m_run[qt] - m_new) * c) : 1.0f;
        lacc[qt][0] *= alpha;
        m_run[qt] = m_new;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qt][dt][r] *= alpha;
      }
    }
    // ---- P^T (e4m3 of p * 2^5) in the B-operand layout: byte kt*16 + r of lane (il, hf) ----
    i32x8 pf[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const float mb = m_run[qt] * c - (float)F8_PSHIFT;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float e0 = fast_exp2(fmaf(s[qt][kt][4 * w + 0], c, -mb));
          const float e1 = fast_exp2(fmaf(s[qt][kt][4 * w + 1], c, -mb));
          const float e2 = fast_exp2(fmaf(s[qt][kt][4 * w + 2], c, -mb));
          const float e3 = fast_exp2(fmaf(s[qt][kt][4 * w + 3], c, -mb));
          int word = 0;
          word = __builtin_amdgcn_cvt_pk_fp8_f32(e0, e1, word, false);
          word = __builtin_amdgcn_cvt_pk_fp8_f32(e2, e3, word, true);
          pf[qt][kt * 4 + w] = word;
        }
      lacc[qt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones, pf[qt], lacc[qt], 0, 0, 0, 127, 0, 127 - F8_PSHIFT);
    }
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const i32x8 va = f8_frag<4>(Vs, dt * 32 + il, hf * 2);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
        o[qt][dt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, pf[qt], o[qt][dt], 0, 0, 0, vs[dt], 0,
                                                                     127 - F8_PSHIFT);
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int kf = 0; kf < NKF; ++kf) ks[kt][kf] = ks_n[kt][kf];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) vs[dt] = vs_n[dt];
  }
  // ---- epilogue: O[q][d] = O^T[d][q] / l ----
  bf16_t* Ob = reinterpret_cast<bf16_t*>(p.Out) + (int64_t)seg * p.o_ss + (int64_t)pr * p.o_ps + (int64_t)h * p.o_hs;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int q = qw0 + qt * 32 + il;
    const float l = lacc[qt][0];                        // sum over ALL keys (both lane halves contract in the MFMA)
    const float fix = 1.0f / l;
    if (q < p.Lq) {
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x4 v = {o[qt][dt][4 * r4 + 0] * fix, o[qt][dt][4 * r4 + 1] * fix, o[qt][dt][4 * r4 + 2] * fix,
                           o[qt][dt][4 * r4 + 3] * fix};
          store4<bf16_t>(Ob + (int64_t)q * p.o_rs + dt * 32 + 8 * r4 + 4 * hf, v);
        }
      if (p.LSE && hf == 0)
        p.LSE[(((int64_t)seg * p.P + pr) * p.H + h) * p.Lq + q] = m_run[qt] * p.scale + logf(l);
    }
  }
}

struct F8Layout { size_t q8, sq, k8, sk, v8, sv, total; int LqP, LkP, NT; };
F8Layout f8_layout(int P, int H, int Lq, int Lk, int HD) {
  F8Layout L;
  L.LqP = (Lq + 31) / 32 * 32; L.NT = (Lk + F8_KVB - 1) / F8_KVB; L.LkP = L.NT * F8_KVB;
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  size_t off = 0;
  const size_t ph = (size_t)P * H;
  L.q8 = off; off = up(off + ph * L.LqP * HD);
  L.sq = off; off = up(off + ph * L.LqP * (HD / 32));
  L.k8 = off; off = up(off + ph * L.LkP * HD);
  L.sk = off; off = up(off + ph * L.LkP * (HD / 32));
  L.v8 = off; off = up(off + ph * L.NT * HD * 64);
  L.sv = off; off = up(off + ph * L.NT * HD * 2);
  L.total = off;
  return L;
}

int f8_check(const dl_attn_fwd_args* a) {
  DL_CHECK_ARG(a && a->Q && a->K && a->V && a->O, DL_ERR_ARG, "dl_attn_fwd_fp8: null pointer");
  DL_CHECK_ARG(a->dtype == DL_BF16, DL_ERR_UNSUPPORTED, "dl_attn_fwd_fp8: operands must be bf16 (they are quantised inside)");
  DL_CHECK_ARG(a->head_dim == 64 || a->head_dim == 128, DL_ERR_UNSUPPORTED, "dl_attn_fwd_fp8: head_dim %d not in {64,128}", a->head_dim);
  DL_CHECK_ARG(a->n_segments == 1 || a->n_segments == 2, DL_ERR_ARG, "dl_attn_fwd_fp8: n_segments must be 1 or 2");
  DL_CHECK_ARG(a->n_problems > 0 && a->n_heads > 0 && a->Lq > 0 && a->Lk > 0 && a->n_problems <= 65535 && a->n_heads <= 65535,
               DL_ERR_SHAPE, "dl_attn_fwd_fp8: bad sizes");
  DL_CHECK_ARG(a->n_segments == 1 || (a->partner_shift >= 0 && a->partner_shift < a->n_problems), DL_ERR_ARG,
               "dl_attn_fwd_fp8: bad partner_shift");
  DL_CHECK_ARG(!a->raw_logits, DL_ERR_UNSUPPORTED, "dl_attn_fwd_fp8: raw logits are not produced by the fp8 form");
  const int64_t st[] = {a->q_ps, a->q_hs, a->q_rs, a->k_ps, a->k_hs, a->k_rs, a->v_ps, a->v_hs, a->v_rs, a->o_ps, a->o_hs, a->o_rs, a->o_ss};
  for (int i = 0; i < 13; ++i)
    DL_CHECK_ARG(st[i] % 8 == 0, DL_ERR_ALIGN, "dl_attn_fwd_fp8: stride #%d (%ld) not a multiple of 8 elements", i, (long)st[i]);
  DL_CHECK_ARG((((uintptr_t)a->Q | (uintptr_t)a->K | (uintptr_t)a->V) & 15) == 0, DL_ERR_ALIGN, "dl_attn_fwd_fp8: Q/K/V must be 16-byte aligned");
  return DL_OK;
}

template <int HD>
int f8_launch(const dl_attn_fwd_args* a, char* ws, hipStream_t s) {
  const F8Layout L = f8_layout(a->n_problems, a->n_heads, a->Lq, a->Lk, HD);
  QuantP q = {};
  q.P = a->n_problems; q.H = a->n_heads;
  // Q rows
  q.X = (const char*)a->Q; q.ps = a->q_ps; q.hs = a->q_hs; q.rs = a->q_rs; q.L = a->Lq; q.LP = L.LqP;
  q.X8 = (uint8_t*)(ws + L.q8); q.sX = (uint8_t*)(ws + L.sq);
  int64_t n = (int64_t)q.P * q.H * q.LP * (HD / 8);
  hipLaunchKernelGGL((attn_quant_rows_kernel<HD>), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, q);
  // K rows
  q.X = (const char*)a->K; q.ps = a->k_ps; q.hs = a->k_hs; q.rs = a->k_rs; q.L = a->Lk; q.LP = L.LkP;
  q.X8 = (uint8_t*)(ws + L.k8); q.sX = (uint8_t*)(ws + L.sk);
  n = (int64_t)q.P * q.H * q.LP * (HD / 8);
  hipLaunchKernelGGL((attn_quant_rows_kernel<HD>), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, q);
  // V transposed
  q.X = (const char*)a->V; q.ps = a->v_ps; q.hs = a->v_hs; q.rs = a->v_rs; q.L = a->Lk; q.NT = L.NT;
  q.X8 = (uint8_t*)(ws + L.v8); q.sX = (uint8_t*)(ws + L.sv);
  n = (int64_t)q.P * q.H * L.NT;
  hipLaunchKernelGGL((attn_quant_vt_kernel<HD>), dim3((uint32_t)n), dim3(256), 0, s, q);

  AttnF8P p = {};
  p.Q8 = (const uint8_t*)(ws + L.q8); p.sQ = (const uint8_t*)(ws + L.sq);
  p.K8 = (const uint8_t*)(ws + L.k8); p.sK = (const uint8_t*)(ws + L.sk);
  p.V8T = (const uint8_t*)(ws + L.v8); p.sV = (const uint8_t*)(ws + L.sv);
  p.Out = (char*)a->O; p.LSE = a->LSE;
  p.o_ps = a->o_ps; p.o_hs = a->o_hs; p.o_rs = a->o_rs; p.o_ss = a->o_ss;
  p.P = a->n_problems; p.H = a->n_heads; p.S = a->n_segments; p.shift = a->partner_shift;
  p.Lq = a->Lq; p.Lk = a->Lk; p.LqP = L.LqP; p.LkP = L.LkP; p.NT = L.NT; p.scale = a->scale;
  constexpr int QT = HD == 64 ? 2 : 1;
  constexpr int QB = 4 * QT * 32;
  const dim3 grid((uint32_t)(p.S * ((p.Lq + QB - 1) / QB)), (uint32_t)p.H, (uint32_t)p.P);
  hipLaunchKernelGGL((attn_fwd_fp8_kernel<HD, QT>), grid, dim3(256), 0, s, p);
  return DL_OK;
}

}  // namespace

extern "C" size_t dl_attn_fwd_fp8_workspace_bytes(const dl_attn_fwd_args* a) {
  if (!a || (a->head_dim != 64 && a->head_dim != 128) || a->n_problems <= 0 || a->n_heads <= 0 || a->Lq <= 0 || a->Lk <= 0) return 0;
  return f8_layout(a->n_problems, a->n_heads, a->Lq, a->Lk, a->head_dim).total;
}

extern "C" int dl_attn_fwd_fp8(const dl_attn_fwd_args* a, void* workspace, size_t workspace_bytes, dl_stream stream) {
  hipStream_t s = (hipStream_t)stream;
  int rc = f8_check(a);
  if (rc != DL_OK) return rc;
  const size_t need = dl_attn_fwd_fp8_workspace_bytes(a);
  DL_CHECK_ARG(workspace && workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0, DL_ERR_WORKSPACE,
               "dl_attn_fwd_fp8: needs %zu workspace bytes (16-byte aligned), got %zu", need, workspace_bytes);
  dl_prof_before(4, s);
  rc = a->head_dim == 64 ? f8_launch<64>(a, (char*)workspace, s) : f8_launch<128>(a, (char*)workspace, s);
  DL_CHECK_LAUNCH("dl_attn_fwd_fp8");
  const double nq = (double)a->n_segments * a->n_problems * a->n_heads * a->Lq;
  dl_prof_after(4, s, 4.0 * nq * a->Lk * a->head_dim,
                (2.0 * nq + 2.0 * a->n_problems * a->n_heads * a->Lk) * a->head_dim * 2.0);
  return rc;
}
