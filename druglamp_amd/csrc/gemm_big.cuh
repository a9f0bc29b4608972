// gemm_big.cuh — large-tile bf16 GEMM for the forward / data-gradient shapes of the path (both operands
// K-contiguous, M in the tens of thousands, K a few hundred to a few thousand).  Included by gemm.hip.
//
// Why a second kernel: at 128x128 the operand stream from L2 (64 FLOP per byte) caps the main loop near
// 800 TFLOP/s; a 256x256 tile halves that traffic (128 FLOP/B) and a wave tile of 128x64 lowers the LDS
// fragment traffic per MFMA by a quarter.
//
//   workgroup : 512 threads = 8 waves; 256(m) x 256(n) tile as 2 x 4 waves of 128 x 64, or 256 x 128 as 4 x 2
//               waves of 64 x 64 (narrow outputs).  One workgroup per CU (LDS-limited), persistent over
//               the tile list with the same XCD-aware order as gemm_kernel.
//   pipeline  : NSTAGE LDS buffers of [BM + BN rows][ROWB bytes] filled by LDS-DMA (global_load_lds,
//               source-side XOR swizzle), prefetch distance NSTAGE-1 k-steps, ONE s_barrier per k-step and
//               counted s_waitcnt vmcnt(n) so that later stages stay in flight across the barrier.
//   epilogue  : the first stages of the NEXT tile are requested before the epilogue; accumulators go
//               through a 4 KB (or 2 KB) wave-private, XOR-swizzled fp32 slice of the stage buffer that was consumed
//               last (no workgroup barrier inside the epilogue) and leave as 16-byte stores, 8 lanes per
//               128-byte line.
#pragma once

// One 8-column piece of an output row.  b0/b1: the lane's bias values (the column is fixed per lane, loaded once per
// tile); ext: the residual (EPI 3) or the saved pre-activation (EPI 4) for this piece, requested several pieces
// ahead by the caller so that its HBM latency is not paid once per piece.
// Addresses (round 6): a piece's row is row_s + (lane >> 3) with row_s WAVE-UNIFORM (tile row + the wave's offset + a compile-time
// constant per piece), so every pointer is a scalar base (row_s * ld + the wave's first column: scalar unit) plus ONE 32-bit lane
// offset per tensor that is the same for every piece and tile (lo_c / lo_p: output, pre-activation copy), and the dropout group
// index is a scalar plus the lane's constant.  Rounds 1-5 multiplied row * ld (and row * N for the dropout key) per piece in
// 64-bit vector arithmetic: five quarter-rate integer multiplies = 4 of the ~27 VALU slots per output element of the GELU +
// dropout epilogue, which is VALU-bound (tools/trickle_parts.py).
template <int EPI>
__device__ __forceinline__ void big_epilogue8(const GemmP& p, uint64_t seed_eff, int row_s, int n0_s, uint32_t lo_c, uint32_t lo_p, uint32_t lo_g,
                                              f32x4 a0, f32x4 a1, f32x4 b0, f32x4 b1, u32x4 ext) {
  typedef bf16_t T;
  if constexpr (EPI != 4) { a0 += b0; a1 += b1; }
  if constexpr (EPI == 2 || EPI == 3 || EPI == 4) { a0 = dl_round_store<T>(a0); a1 = dl_round_store<T>(a1); }   // (study builds only: common.cuh)
  if constexpr (EPI == 2) {
    char* pd = p.pre_out + ((int64_t)row_s * p.ldp + n0_s) * 2 + lo_p;
    u32x4 o = {pack_bf16x2(a0[0], a0[1]), pack_bf16x2(a0[2], a0[3]), pack_bf16x2(a1[0], a1[1]), pack_bf16x2(a1[2], a1[3])};
    if (p.nt_pre) store16_nt(pd, o, p.nt_pre);
    else *reinterpret_cast<u32x4*>(pd) = o;
    a0 = gelu4<T>(a0); a1 = gelu4<T>(a1);
  } else if constexpr (EPI == 5) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { a0[r] = fmaxf(a0[r], 0.f); a1[r] = fmaxf(a1[r], 0.f); }
  } else if constexpr (EPI == 4) {
    a0 *= gelu_grad4<T>(f32x4{bf16lo(ext[0]), bf16hi(ext[0]), bf16lo(ext[1]), bf16hi(ext[1])});
    a1 *= gelu_grad4<T>(f32x4{bf16lo(ext[2]), bf16hi(ext[2]), bf16lo(ext[3]), bf16hi(ext[3])});
  }
  if (EPI != 5 && EPI != 0 && p.drop_thr16) {
    const uint64_t grp = (((uint64_t)row_s * (uint64_t)p.N + (uint64_t)n0_s) >> 2) + lo_g;   // = (row * N + col) >> 2: N % 8 == 0, col % 8 == 0
    a0 = dl_dropout4_idx(a0, seed_eff, grp, p.drop_thr16, p.drop_inv_keep);
    a1 = dl_dropout4_idx(a1, seed_eff, grp + 1, p.drop_thr16, p.drop_inv_keep);
  }
  if constexpr (EPI == 3) {
    a0 += f32x4{bf16lo(ext[0]), bf16hi(ext[0]), bf16lo(ext[1]), bf16hi(ext[1])};
    a1 += f32x4{bf16lo(ext[2]), bf16hi(ext[2]), bf16lo(ext[3]), bf16hi(ext[3])};
  }
  u32x4 o = {pack_bf16x2(a0[0], a0[1]), pack_bf16x2(a0[2], a0[3]), pack_bf16x2(a1[0], a1[1]), pack_bf16x2(a1[2], a1[3])};
  char* dstp = p.C + ((int64_t)row_s * p.ldc + n0_s) * 2 + lo_c;
  if (p.nt_c) store16_nt(dstp, o, p.nt_c);
  else *reinterpret_cast<u32x4*>(dstp) = o;
}

// own DMA of the awaited step has landed (counted: younger steps stay in flight) and own LDS reads retired
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }
// One LDS-DMA instruction (64 lanes x 16 bytes -> 1 KB of LDS at lds_off + 16 * lane) as inline assembly.  Through
// __builtin_amdgcn_global_load_lds the compiler books a FLAT access that may touch LDS: while one is pending in its
// model (always, here: the asm waits above are invisible to it) it drains EVERY fragment read with lgkmcnt(0), the
// read issued one instruction earlier included, and the prefetch distance of the fragment pipeline collapses.
// Invisible DMA leaves it counting (lgkmcnt(2) with two younger reads in flight); ordering against the reads of
// the buffer is by wait_vmcnt + wg_barrier, as before.
__device__ __forceinline__ void dma16(const char* gsrc, uint32_t lds_off) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_off) : "memory");   // M0 is reserved: it cannot be listed as a clobber; nothing else in these kernels lives in it
}
// wave-level ordering point for cross-lane traffic through LDS (no instruction: the LDS queue is in order)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// XF / WF: MFMA 16x16 tiles per wave along m / n (WF is fixed at 4: 64 columns per wave, see the epilogue);
// NWM x NWN waves; ROWB: bytes of contraction per operand row per k-step (128 or 64); NSTAGE LDS buffers.
// (second launch-bound argument: workgroups per CU the register allocation is sized for — one when the stage ring alone
//  takes more than half of the 160 KB LDS, else two, i.e. two waves per SIMD either way for the 8-wave forms)
template <int XF, int NWM, int NWN, int ROWB, int NSTAGE, int EPI>
__global__ __launch_bounds__(64 * NWM * NWN, (NWM * NWN >= 8 ? NWM * NWN / 4 : ((16 * XF * NWM + 64 * NWN) * ROWB * NSTAGE > 81920 ? 1 : 2)))
void gemm_big_kernel(const GemmP p) {
  typedef bf16_t T;
  constexpr int WF = 4;
  constexpr int NT = 64 * NWM * NWN, BM = 16 * XF * NWM, BN = 16 * WF * NWN;
  constexpr int KCH = ROWB / 16, NKF = ROWB / 64, BKE = ROWB / 2;
  constexpr int XB = BM * ROWB, WB = BN * ROWB, STAGE = XB + WB;
  constexpr int XCH = BM * KCH / NT, WCH = BN * KCH / NT, PER = XCH + WCH;
  constexpr int D = NSTAGE - 1;
  static_assert(BM * KCH % NT == 0 && BN * KCH % NT == 0, "whole DMA instructions per thread");
  constexpr int SR = (STAGE >= 4096 * NWM * NWN) ? 16 : 8;     // rows per epilogue staging pass
  static_assert(STAGE >= SR * 256 * NWM * NWN, "epilogue slices must fit one stage buffer");
  static_assert(D >= 1 && D <= 3 && PER * D < 64, "prefetch distance");
  // (+16 bytes: the ticket word of the dynamic tile hand-out lives in the SAME shared object — a second __shared__
  //  variable makes the compiler drain vmcnt before every fragment read of the LDS-DMA pipeline)
  __shared__ __attribute__((aligned(16))) char smem[NSTAGE * STAGE + 16];
  volatile uint32_t* ticket_s = reinterpret_cast<volatile uint32_t*>(smem + NSTAGE * STAGE);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 15, g = lane >> 4;
  const int wm = wave / NWN, wn = wave % NWN;
  const int swz = (ROWB == 128) ? (il & 7) : ((il >> 1) & 3);
  const int xoff = (wm * 16 * XF + il) * ROWB, woff = XB + (wn * 16 * WF + il) * ROWB;

  // the dropout seed of this launch: by-value seed + the device-side step offset, read ONCE here (round 5, late: the
  // epilogue read *seed_off per 8-column piece — behind its own stores, which the load could not be hoisted over: a
  // dependent memory round trip per piece, +0.26 ms per batch-256 step whenever the offset regime was on, i.e. in every
  // graph-replaying trainer)
  const uint64_t seed_eff = (EPI != 5 && EPI != 0 && p.drop_thr16) ? dl_eff_seed(p.seed, p.seed_off) : 0;
  const uint32_t ntiles = (uint32_t)p.mt * p.nt;
  const uint32_t G = gridDim.x;
  const bool dyn = p.tickets != nullptr;
  auto locate = [&](uint32_t it, int& m0, int& n0) {
    const uint32_t round0 = (it / G) * G;
    const uint32_t span = min(G, ntiles - round0);
    const uint32_t t = round0 + xcd_remap(it - round0, span);
    m0 = (int)(t / p.nt) * BM; n0 = (int)(t % p.nt) * BN;
  };
  const char* xs[XCH];
  const char* ws[WCH];
  auto point = [&](int m0, int n0) {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int c = tid + i * NT, row = c / KCH, pc = c % KCH;
      const int kc = pc ^ ((ROWB == 128) ? (row & 7) : ((row >> 1) & 3));
      int gr = m0 + row; gr = gr < p.M ? gr : p.M - 1;
      xs[i] = p.X + ((int64_t)gr * p.ldx) * 2 + kc * 16;
    }
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
      const int c = tid + i * NT, row = c / KCH, pc = c % KCH;
      const int kc = pc ^ ((ROWB == 128) ? (row & 7) : ((row >> 1) & 3));
      int gr = n0 + row; gr = gr < p.N ? gr : p.N - 1;
      ws[i] = p.W + ((int64_t)gr * p.ldw) * 2 + kc * 16;
    }
  };
  // LDS byte address of the stage ring (M0 takes addresses, not pointers)
  const uint32_t smem_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  // one of the PER DMA instructions of a step (piece q: X pieces first)
  auto issue_piece = [&](int kt, uint32_t buf, int q) {
    const uint32_t sb = smem_lds + buf * STAGE;
    const int64_t kb = (int64_t)kt * ROWB;
    if (q < XCH) {
      dma16(xs[q] + kb, __builtin_amdgcn_readfirstlane(sb + (uint32_t)((wave * 64 + q * NT) * 16)));
    } else {
      dma16(ws[q - XCH] + kb, __builtin_amdgcn_readfirstlane(sb + (uint32_t)(XB + (wave * 64 + (q - XCH) * NT) * 16)));
    }
  };
  auto issue = [&](int kt, uint32_t buf) {
#pragma unroll
    for (int q = 0; q < PER; ++q) issue_piece(kt, buf, q);
  };
  const int nk = p.K / BKE;            // whole steps (checked by the host)
  uint32_t it = blockIdx.x;
  if (it >= ntiles) return;
  int m0, n0;
  locate(it, m0, n0);
  point(m0, n0);
  uint32_t gs = 0;                     // global step counter: stage buffer = gs % NSTAGE
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < nk) issue(s, (gs + s) % NSTAGE);

  for (;;) {
    f32x4 acc[XF][WF];
#pragma unroll
    for (int i = 0; i < XF; ++i)
#pragma unroll
      for (int j = 0; j < WF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // One k-step.  FEED: the next step's PER DMA instructions go out one per item between the MFMA groups
    // (issued back to back after the barrier they cost every wave ~100-185 cycles apiece with the matrix
    // pipe idle: ~1200 cycles against the step's 1024 MFMA cycles per wave).
    auto kstep = [&](auto feed, int kt) {
      constexpr bool FEED = decltype(feed)::value;
      // step kt has landed once at most `later` younger steps' DMA instructions are outstanding
      const int later = (kt == 0) ? 0 : min(D - 1, nk - 1 - kt);
      if (later <= 0) wait_vmcnt<0>();
      else if (later == 1) wait_vmcnt<PER>();
      else wait_vmcnt<(D >= 3 ? 2 * PER : PER)>();
      wg_barrier();
      // every wave is past step kt-1: its buffer takes step kt+D
      constexpr int U = NKF * XF;
      constexpr bool SPREAD = FEED && PER <= U;
      if (FEED && !SPREAD) issue(kt + D, (gs + D) % NSTAGE);
      const char* cur = smem + (gs % NSTAGE) * STAGE;
      // Software-pipelined fragment stream: item u = (kf, i) needs X fragment fx[u] and the four W fragments of
      // its kf.  X fragments are read two items ahead, the next kf's W fragments during the last four items of kf 0,
      // and sched_group_barrier pins "reads of this item, then its 4 MFMAs" so that LDS latency hides under
      // the matrix pipe instead of in front of it.
      auto rd_fx = [&](int u) { return lds_read16(cur, xoff + (u % XF) * 16 * ROWB + ((((u / XF) * 4 + g) ^ swz) << 4)); };
      auto rd_fw = [&](int kf, int j) { return lds_read16(cur, woff + j * 16 * ROWB + (((kf * 4 + g) ^ swz) << 4)); };
      u32x4 fx[U], fw[NKF][WF];
#pragma unroll
      for (int j = 0; j < WF; ++j) fw[0][j] = rd_fw(0, j);
      fx[0] = rd_fx(0);
      fx[1] = rd_fx(1);
      __builtin_amdgcn_sched_group_barrier(0x100, WF + 2, 0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int nread = 0;
        if (u + 2 < U) { fx[u + 2] = rd_fx(u + 2); ++nread; }
        if (NKF == 2 && u >= XF - WF && u < XF) { fw[1][u - (XF - WF)] = rd_fw(1, u - (XF - WF)); ++nread; }
        if (SPREAD && u < PER) issue_piece(kt + D, (gs + D) % NSTAGE, u);
#pragma unroll
        for (int j = 0; j < WF; ++j) acc[u % XF][j] = Mma<T>::mma(fw[u / XF][j], fx[u], acc[u % XF][j]);
        if (nread == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        else if (nread == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (SPREAD && u < PER) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, WF, 0);
      }
      ++gs;
    };
    // dynamic hand-out: the ticket of the NEXT tile is drawn while this tile's main loop runs (its latency hides behind
    // the k-steps; every k-step has a workgroup barrier, so the word is visible to all waves long before it is read)
    if (dyn && tid == 0) *ticket_s = G + (uint32_t)__hip_atomic_fetch_add(p.tickets, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (DL_DBG(p) & 2) {                   // timing study: no operand feed after the prologue (results are garbage)
      for (int kt = 0; kt < nk; ++kt) kstep(std::false_type{}, kt);
    } else {
      for (int kt = 0; kt + D < nk; ++kt) kstep(std::true_type{}, kt);
      for (int kt = max(nk - D, 0); kt < nk; ++kt) kstep(std::false_type{}, kt);
    }

    // ---- tile end: request the next tile's first stages, then the epilogue --------------------
    const int cm0 = m0, cn0 = n0;
    const uint32_t itn = dyn ? (uint32_t)__builtin_amdgcn_readfirstlane((int)*ticket_s) : it + G;
    const bool have_next = itn < ntiles;
    if (have_next) {
      locate(itn, m0, n0);
      point(m0, n0);
#pragma unroll
      for (int s = 0; s < D; ++s)
        if (s < nk) issue(s, (gs + s) % NSTAGE);   // every buffer but the one consumed last is free
    }
    wg_barrier();                                   // all waves are done reading the last stage buffer (and the ticket word)
    char* st = smem + ((gs + NSTAGE - 1) % NSTAGE) * STAGE + wave * (SR * 256);
    // pieces of this lane: item = (i, hp, h) -> row m_of(item), columns n_lane .. n_lane + 7
    constexpr int HPI = 16 / SR, HH = SR / 8, NITEM = XF * HPI * HH, PF = 4;
    const int n_lane = cn0 + wn * 16 * WF + (lane & 7) * 8;
    const bool n_ok = n_lane < p.N;
    // rows of this lane's pieces: row_s(item) + (lane >> 3), row_s wave-uniform (ro_of: a compile-time constant per unrolled item)
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int row0_s = cm0 + (wave_s / NWN) * 16 * XF, n0_s = cn0 + (wave_s % NWN) * 16 * WF;
    auto ro_of = [](int item) { return (item / (HPI * HH)) * 16 + ((item / HH) % HPI) * SR + 8 * (item % HH); };
    const int lr = lane >> 3, lc = (lane & 7) * 8;
    const uint32_t lo_c = (uint32_t)((lr * (int)p.ldc + lc) * 2);                         // byte offsets of (row lr, column lc) of a piece
    const uint32_t lo_p = EPI == 2 ? (uint32_t)((lr * (int)p.ldp + lc) * 2) : 0u;
    const int64_t lde = EPI == 3 ? p.ldr : p.lddp;
    const uint32_t lo_e = (EPI == 3 || EPI == 4) ? (uint32_t)((lr * (int)lde + lc) * 2) : 0u;
    const uint32_t lo_g = (uint32_t)((lr * p.N + lc) >> 2);                              // dropout group offset
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (EPI != 4 && p.bias && n_ok) {
      b0 = *reinterpret_cast<const f32x4*>(p.bias + n_lane);
      b1 = *reinterpret_cast<const f32x4*>(p.bias + n_lane + 4);
    }
    auto fetch = [&](int item) -> u32x4 {
      u32x4 v = {0u, 0u, 0u, 0u};
      if constexpr (EPI == 3 || EPI == 4) {
        const int row_s = row0_s + ro_of(item);
        if (row_s + lr < p.M && n_ok) {
          const char* src = (EPI == 3 ? p.res : p.dact_pre) + ((int64_t)row_s * lde + n0_s) * 2 + lo_e;
          v = p.nt_ext ? load16_nt(src) : *reinterpret_cast<const u32x4*>(src);
        }
      }
      return v;
    };
    u32x4 ext[NITEM];
#pragma unroll
    for (int it2 = 0; it2 < PF && it2 < NITEM; ++it2) ext[it2] = fetch(it2);
#pragma unroll
    for (int i = 0; i < XF; ++i) {
#pragma unroll
      for (int hp = 0; hp < HPI; ++hp) {
        if (SR == 16 || (il >> 3) == hp) {
          const int wr = il & (SR - 1);
#pragma unroll
          for (int j = 0; j < WF; ++j) lds_write16(st, wr * 256 + (((j * 4 + g) ^ wr) << 4), __builtin_bit_cast(u32x4, acc[i][j]));
        }
        // lanes exchange rows through the slice: keep the reads out of the (possibly divergent) write region
        wave_sync();
#pragma unroll
        for (int h = 0; h < HH; ++h) {
          const int item = (i * HPI + hp) * HH + h;
          if (item + PF < NITEM) ext[item + PF] = fetch(item + PF);
          const int r = (lane >> 3) + 8 * h, c8 = lane & 7;
          const f32x4 a0 = __builtin_bit_cast(f32x4, lds_read16(st, r * 256 + (((2 * c8) ^ r) << 4)));
          const f32x4 a1 = __builtin_bit_cast(f32x4, lds_read16(st, r * 256 + (((2 * c8 + 1) ^ r) << 4)));
          const int row_s = row0_s + ro_of(item);
          if (row_s + lr < p.M && n_ok && !(DL_DBG(p) & 1))
            big_epilogue8<EPI>(p, seed_eff, row_s, n0_s, lo_c, lo_p, lo_g, a0, a1, b0, b1, ext[item]);
        }
        wave_sync();
      }
    }
    if (!have_next) break;
    it = itn;
  }
  // the last workgroup to leave returns the two ticket words to zero for the next launch on the stream
  if (dyn && tid == 0) {
    const int gone = __hip_atomic_fetch_add(p.tickets + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gone == (int)G - 1) {
      __hip_atomic_store(p.tickets, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(p.tickets + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Weight-gradient form: C[m][n] = sum_k X[k][m] W[k][n], both operands K-SLOW (rows of dY and of the layer
// input), tens of thousands of k-rows, a small output.  Same 8-wave large tile and LDS-DMA ring as above;
// operand tiles are [64 k-rows][BM or BN columns] with the 16-byte chunks of a k-row XOR-swizzled by
// swz(k) = (k & 3) << 1 | ((k >> 3) & 1) << 3 so that the 8-byte pieces the 32 lanes of a ds_read_b64_tr_b16 lane group touch land on 32
// different chunk columns.  The K range is cut into `splits` slabs (fp32 partial outputs, reduced by
// splitk_reduce_kernel in a fixed order); k-rows past the end of a slab's range are fed from a zero page,
// so K needs no alignment.
__device__ __attribute__((aligned(16))) const uint32_t dl_zero_page[4] = {0u, 0u, 0u, 0u};

// ---------------------------------------------------------------------------------------------------------
// Round 3: the weight-gradient tile with a deep feed (round 2's gemm_big_tt_kernel: two 64-row stages, DMA through the
// builtin).  Measured on 2048x512x65536 (tools/tt_study.py): the builtin's DMA made the compiler drain every transposing
// fragment read (compute-only 218 -> 141 us with the inline-assembly form), and with ONE step in flight the loop waits
// for first-touch HBM data every step (every k-row of a weight gradient is read once): 32-row steps x 4 stages =
// three steps in flight, 242 -> 164 us.
//   * LDS-DMA as inline assembly in the SGPR-base + 32-bit VGPR-offset form: a lane's offset inside an operand tile is
//     the same for every step and every piece (the 16-row piece stride and the step advance are uniform), so the
//     per-piece 64-bit pointers of the old kernel (24 VGPRs) become two offsets;
//   * one DMA instruction per item BETWEEN the MFMA groups, counted vmcnt, one barrier per step;
//   * K tail (a slab whose last step is partial): that one step takes per-lane pointers with the zero page selected
//     for rows past the end;
//   * PF (optional): an L2 prefetch D steps beyond the DMA distance, one dword LDS-DMA per 128-byte line into a
//     256-byte scratch behind the ring (no destination register; vmcnt returns in order, so it has D + 1 steps).
__device__ __forceinline__ void dma16s(uint32_t voff, const char* sbase_, uint32_t lds_off) {
  // (the base IS uniform; readfirstlane makes the compiler's divergence analysis agree, else "s" may be handed a VGPR pair)
  const uint64_t b = (uint64_t)sbase_;
  const char* sbase = (const char*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                                    (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b));
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}
// GROUP (round 3, dl_gemm_group): up to DL_GROUP_MAX weight-gradient products of DIFFERENT shapes in one launch — the
// weight gradients of a block's backward at the strong-scaling batches, where each of them alone is a handful of tiles
// over a few thousand k-rows and pays a launch, a prologue and a 16- to 32-way slab round trip for a few microseconds of
// matrix work.  The tile lists of the problems are concatenated (GroupProb::end = running tile count); a workgroup
// re-reads the problem's fields whenever it takes a tile (wave-uniform scalar loads from the kernel arguments).
constexpr int DL_GROUP_MAX = 16;
struct GroupProb {
  const char* X; const char* W; float* slabs; float* cs_slabs;
  int64_t ldx, ldw;
  int M, N, K, k_per_split, mt, nt;
  uint32_t end; int pad;
};
struct GemmGroupP { int n; int dbg; GroupProb q[DL_GROUP_MAX]; };

template <int XF, int NWM, int NWN, bool CS, int KS, int NSTAGE, bool PF, bool GROUP = false>
__global__ __launch_bounds__(64 * NWM * NWN, 2) void gemm_big_tt2_kernel(const std::conditional_t<GROUP, GemmGroupP, GemmP> p) {
  typedef bf16_t T;
  constexpr int WF = 4;
  constexpr int NT = 64 * NWM * NWN, BM = 16 * XF * NWM, BN = 16 * WF * NWN;
  constexpr int PX = BM * 2, PW = BN * 2;                    // bytes per k-row
  constexpr int CPX = PX / 16, CPW = PW / 16;                // 16-byte chunks per k-row
  constexpr int XB = KS * PX, WB = KS * PW, STAGE = XB + WB;
  constexpr int XCH = KS * CPX / NT, WCH = KS * CPW / NT, PER = XCH + WCH;
  constexpr int XRS = NT / CPX, WRS = NT / CPW;              // k-rows between consecutive pieces of a thread
  constexpr int D = NSTAGE - 1, NPF = PF ? 1 : 0;
  constexpr int NKF = KS / 32, U = NKF * XF;
  constexpr int LPR = (PX + PW) / 128;                        // 128-byte lines per k-row of a step (both operands)
  static_assert(KS * CPX % NT == 0 && KS * CPW % NT == 0 && NT % CPX == 0 && NT % CPW == 0, "whole DMA instructions per thread");
  static_assert(XRS % 16 == 0 && WRS % 16 == 0, "pieces of a thread share their swizzle");
  static_assert(STAGE >= 4096 * NWM * NWN, "epilogue slices must fit one stage buffer");
  static_assert(CPX >= 16 && CPW >= 16, "swizzle needs 16 chunk columns");
  static_assert(XF >= WF || NKF == 1, "W fragments of the second half-step are fetched during the last WF items of the first");
  static_assert(PER <= U, "one DMA instruction per item");
  static_assert(D >= 1 && D <= 4 && D * NPF + (D - 1) * PER < 64, "vmcnt immediate");
  static_assert(!PF || KS * LPR <= NT, "one prefetch line per thread");
  __shared__ __attribute__((aligned(16))) char smem[NSTAGE * STAGE + (PF ? 256 : 0)];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int il = lane & 15, g = lane >> 4;
  const int wm = wave / NWN, wn = wave % NWN;
  auto swz = [](int k) { return ((k & 3) << 1) | (((k >> 3) & 1) << 3); };
  const uint32_t smem_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  // the problem of the tile being located (loop-invariant unless GROUP)
  const char* qX; const char* qW; float* qslabs; float* qcs;
  int64_t ldx2, ldw2;
  int qM, qN, qK, qkps, qnt;
  uint32_t per_split, ntiles;
  if constexpr (GROUP) {
    ntiles = p.q[p.n - 1].end;
    qX = qW = nullptr; qslabs = qcs = nullptr; ldx2 = ldw2 = 0; qM = qN = qK = qkps = qnt = 0; per_split = 1;
  } else {
    qX = p.X; qW = p.W; qslabs = p.slabs; qcs = p.cs_slabs; ldx2 = p.ldx * 2; ldw2 = p.ldw * 2;
    qM = p.M; qN = p.N; qK = p.K; qkps = p.k_per_split; qnt = p.nt;
    per_split = (uint32_t)p.mt * p.nt;
    ntiles = per_split * p.splits;
  }
  const uint32_t G = gridDim.x;
  auto locate = [&](uint32_t it, int& split, int& m0, int& n0, int& kbeg, int& kend) {
    const uint32_t round0 = (it / G) * G;
    const uint32_t span = min(G, ntiles - round0);
    uint32_t t = round0 + xcd_remap(it - round0, span);
    if constexpr (GROUP) {
      int i = 0;
      while (t >= p.q[i].end) ++i;
      const GroupProb& q = p.q[i];
      t -= i ? p.q[i - 1].end : 0u;
      qX = q.X; qW = q.W; qslabs = q.slabs; qcs = q.cs_slabs; ldx2 = q.ldx * 2; ldw2 = q.ldw * 2;
      qM = q.M; qN = q.N; qK = q.K; qkps = q.k_per_split; qnt = q.nt;
      per_split = (uint32_t)q.mt * q.nt;
    }
    split = (int)(t / per_split);
    const uint32_t tile = t % per_split;
    m0 = (int)(tile / qnt) * BM; n0 = (int)(tile % qnt) * BN;
    kbeg = split * qkps;
    kend = min(qK, kbeg + qkps);
  };
  // a lane's byte offset inside the operand tile of a step (piece 0); the tile's first byte is uniform
  const int xkrow = tid / CPX, wkrow = tid / CPW;
  uint32_t xvoff = 0, wvoff = 0;
  const char* xtile = nullptr; const char* wtile = nullptr;   // uniform: first byte of the slab's first step
  const char* pf_base = nullptr; int64_t pf_ld = 0; int pf_row = 0;
  auto point = [&](int m0, int n0, int kbeg) {
    {
      const int pc = tid % CPX;
      int col = (pc ^ swz(xkrow)) << 3;
      col = m0 + col < qM ? col : qM - 8 - m0;
      xvoff = (uint32_t)(xkrow * (int)ldx2 + col * 2);
      xtile = qX + (int64_t)kbeg * ldx2 + (int64_t)m0 * 2;
    }
    {
      const int pc = tid % CPW;
      int col = (pc ^ swz(wkrow)) << 3;
      col = n0 + col < qN ? col : qN - 8 - n0;
      wvoff = (uint32_t)(wkrow * (int)ldw2 + col * 2);
      wtile = qW + (int64_t)kbeg * ldw2 + (int64_t)n0 * 2;
    }
    if constexpr (PF) {
      const int krow = min(tid / LPR, KS - 1), part = tid % LPR;
      pf_row = kbeg + krow;
      if (part < PX / 128) {
        const int col = min(m0 + part * 64, qM - 2);
        pf_base = qX + (int64_t)col * 2; pf_ld = ldx2;
      } else {
        const int col = min(n0 + (part - PX / 128) * 64, qN - 2);
        pf_base = qW + (int64_t)col * 2; pf_ld = ldw2;
      }
    }
  };
  const char* zero = reinterpret_cast<const char*>(dl_zero_page);
  // piece q of step kt (X pieces first).  PARTIAL: the slab's last step with only `rows` (< KS) k-rows left
  auto issue_piece = [&](auto partial, int kt, int rows, uint32_t buf, int q) {
    constexpr bool PARTIAL = decltype(partial)::value;
    const uint32_t sb = smem_lds + buf * STAGE;
    if (q < XCH) {
      const char* sbase = xtile + ((int64_t)kt * KS + q * XRS) * ldx2;
      const uint32_t lo = __builtin_amdgcn_readfirstlane(sb + (uint32_t)((wave * 64 + q * NT) * 16));
      if constexpr (!PARTIAL) dma16s(xvoff, sbase, lo);
      else dma16((xkrow + q * XRS < rows) ? sbase + xvoff : zero, lo);
    } else {
      const int i = q - XCH;
      const char* sbase = wtile + ((int64_t)kt * KS + i * WRS) * ldw2;
      const uint32_t lo = __builtin_amdgcn_readfirstlane(sb + (uint32_t)(XB + (wave * 64 + i * NT) * 16));
      if constexpr (!PARTIAL) dma16s(wvoff, sbase, lo);
      else dma16((wkrow + i * WRS < rows) ? sbase + wvoff : zero, lo);
    }
  };
  auto issue = [&](int kt, int rows, uint32_t buf) {
    if (rows >= KS) {
#pragma unroll
      for (int q = 0; q < PER; ++q) issue_piece(std::false_type{}, kt, rows, buf, q);
    } else {
#pragma unroll
      for (int q = 0; q < PER; ++q) issue_piece(std::true_type{}, kt, rows, buf, q);
    }
  };
  // L2 prefetch of step kt's lines (rows past the operand's end are clamped to its last row: always a legal address)
  auto prefetch = [&](int kt) {
    if constexpr (PF) {
      const int row = min(pf_row + kt * KS, qK - 1);
      const char* src = pf_base + (int64_t)row * pf_ld;
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(smem_lds + (uint32_t)(NSTAGE * STAGE)) : "memory");
    }
  };
  auto frag = [&](const char* tile, int pitch, int col0, int kf) -> u32x4 {
    const int k0 = kf * 32 + g * 8 + (il >> 2), k1 = k0 + 4;
    const int cb = (col0 + (il & 3) * 4) * 2;
    const u32x2 a = lds_read_tr16(tile, k0 * pitch + (((cb >> 4) ^ swz(k0)) << 4) + (cb & 15));
    const u32x2 b = lds_read_tr16(tile, k1 * pitch + (((cb >> 4) ^ swz(k1)) << 4) + (cb & 15));
    return u32x4{a[0], a[1], b[0], b[1]};
  };

  uint32_t it = blockIdx.x;
  if (it >= ntiles) return;
  int split, m0, n0, kbeg, kend;
  locate(it, split, m0, n0, kbeg, kend);
  point(m0, n0, kbeg);
  uint32_t gs = 0;
  // prologue of a tile: steps 0 .. D-1 requested, each followed by the prefetch of step D + s (the loop's pattern)
  auto prologue = [&]() {
    const int len = kend - kbeg;
#pragma unroll
    for (int s = 0; s < D; ++s) {
      if (s * KS < len) issue(s, len - s * KS, (gs + s) % NSTAGE);
      prefetch(D + s);
    }
  };
  prologue();

  for (;;) {
    f32x4 acc[XF][WF];
#pragma unroll
    for (int i = 0; i < XF; ++i)
#pragma unroll
      for (int j = 0; j < WF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float cs[XF];
#pragma unroll
    for (int i = 0; i < XF; ++i) cs[i] = 0.f;
    const bool do_cs = CS && n0 == 0 && wn == 0 && (!GROUP || qcs != nullptr);
    const int len = kend - kbeg;
    const int nk = (len + KS - 1) / KS;
    auto kstep = [&](auto with_cs, auto feed, int kt) __attribute__((always_inline)) {
      constexpr bool WCS = decltype(with_cs)::value;
      constexpr int FEED = decltype(feed)::value;           // 0: nothing to request, 1: a whole step, 2: the slab's partial last step
      // step kt has landed once only the younger requests are outstanding: D prefetches and `later` steps of DMA
      const int later = (kt == 0 || nk < D) ? -1 : min(D - 1, nk - 1 - kt);
      if (later < 0) wait_vmcnt<0>();                       // (tile start: also drains the previous epilogue's stores)
      else if (later == 0) wait_vmcnt<D * NPF>();
      else if (later == 1) wait_vmcnt<D * NPF + PER>();
      else if (later == 2) wait_vmcnt<D * NPF + (D >= 3 ? 2 : 1) * PER>();
      else wait_vmcnt<D * NPF + (D >= 4 ? 3 : 1) * PER>();
      wg_barrier();
      const char* cur = smem + (gs % NSTAGE) * STAGE;
      const int rows_next = len - (kt + D) * KS;
      auto rd_fx = [&](int u) { return frag(cur, PX, wm * 16 * XF + (u % XF) * 16, u / XF); };
      auto rd_fw = [&](int kf, int j) { return frag(cur + XB, PW, wn * 16 * WF + j * 16, kf); };
      u32x4 fx[U], fw[NKF][WF];
#pragma unroll
      for (int j = 0; j < WF; ++j) fw[0][j] = rd_fw(0, j);
      fx[0] = rd_fx(0);
      fx[1] = rd_fx(1);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * (WF + 2), 0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int nread = 0;
        if (u + 2 < U) { fx[u + 2] = rd_fx(u + 2); ++nread; }
        if (NKF == 2 && u >= XF - WF && u < XF) { fw[NKF - 1][u - (XF - WF)] = rd_fw(1, u - (XF - WF)); ++nread; }
        if (FEED == 1 && u < PER) issue_piece(std::false_type{}, kt + D, rows_next, (gs + D) % NSTAGE, u);
        if (FEED == 2 && u < PER) issue_piece(std::true_type{}, kt + D, rows_next, (gs + D) % NSTAGE, u);
        if (u == PER) prefetch(kt + 2 * D);
#pragma unroll
        for (int j = 0; j < WF; ++j) acc[u % XF][j] = Mma<T>::mma(fw[u / XF][j], fx[u], acc[u % XF][j]);
        if (nread == 2) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        else if (nread == 1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, WF, 0);
        if constexpr (WCS) cs[u % XF] = frag_slot_sum<T>(fx[u], cs[u % XF]);
      }
      if (U <= PER) prefetch(kt + 2 * D);
      ++gs;
    };
    auto kloop = [&](auto with_cs) __attribute__((always_inline)) {
      typedef std::integral_constant<int, 0> F0; typedef std::integral_constant<int, 1> F1; typedef std::integral_constant<int, 2> F2;
      if (DL_DBG(p) & 2) {
        for (int kt = 0; kt < nk; ++kt) kstep(with_cs, F0{}, kt);
      } else {
        const int nwhole = len / KS;                        // steps [0, nwhole) are whole; step nwhole (if any) is partial
        int kt = 0;
        for (; kt + D < nwhole; ++kt) kstep(with_cs, F1{}, kt);
        if (kt + D < nk) { kstep(with_cs, F2{}, kt); ++kt; }
        for (; kt < nk; ++kt) kstep(with_cs, F0{}, kt);
      }
    };
    if constexpr (CS) {
      if (do_cs) kloop(std::true_type{}); else kloop(std::false_type{});
    } else {
      kloop(std::false_type{});
    }

    const int cm0 = m0, cn0 = n0, csplit = split;
    const int eM = qM, eN = qN;                    // this tile's problem: locate() moves q* on to the next tile's
    float* const eslabs = qslabs; float* const ecs = qcs;
    const uint32_t itn = it + G;
    const bool have_next = itn < ntiles;
    if (have_next) {
      locate(itn, split, m0, n0, kbeg, kend);
      point(m0, n0, kbeg);
      prologue();
    }
    wg_barrier();
    if (do_cs) {
#pragma unroll
      for (int i = 0; i < XF; ++i) {
        const float v = group4_sum(cs[i]);
        const int m = cm0 + wm * 16 * XF + i * 16 + il;
        if (g == 0 && m < eM) ecs[(int64_t)csplit * eM + m] = v;
      }
    }
    char* st = smem + ((gs + NSTAGE - 1) % NSTAGE) * STAGE + wave * 4096;
#pragma unroll
    for (int i = 0; i < XF; ++i) {
#pragma unroll
      for (int j = 0; j < WF; ++j) lds_write16(st, il * 256 + (((j * 4 + g) ^ il) << 4), __builtin_bit_cast(u32x4, acc[i][j]));
      wave_sync();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = (lane >> 3) + 8 * h, c8 = lane & 7;
        const u32x4 a0 = lds_read16(st, r * 256 + (((2 * c8) ^ r) << 4));
        const u32x4 a1 = lds_read16(st, r * 256 + (((2 * c8 + 1) ^ r) << 4));
        const int m = cm0 + wm * 16 * XF + i * 16 + r, n = cn0 + wn * 16 * WF + c8 * 8;
        if (m < eM && n < eN && !(DL_DBG(p) & 1)) {
          float* dst = eslabs + ((int64_t)csplit * eM + m) * eN + n;
          store16_fam<8>(dst, a0);
          store16_fam<8>(dst + 4, a1);
        }
      }
      wave_sync();
    }
    if (!have_next) break;
    it = itn;
  }
}
