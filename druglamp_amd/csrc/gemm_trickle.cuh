// gemm_trickle.cuh — the large-tile forward / data-gradient GEMM whose epilogue runs INSIDE the next tile's main loop
// (round 6).  Included by gemm.hip after gemm_big.cuh (shares its LDS-DMA helpers and epilogue math).
//
// STATUS: STUDY LIBRARY ONLY (-DDL_STUDY, DL_GEMM_TRICKLE=1).  Correct (bit-identical to the other two kernels on every shape and
// epilogue: tools/trickle_bench.py) and NOT faster: 5-25 % slower than gemm_big_kernel on the shapes of the path, a tie at K = 256
// with the heavy epilogues.  The decomposition (tools/trickle_parts.py, profiles/r6_trickle_study.txt; 65536x2048x512 with GELU +
// pre-activation + dropout): the trickled stores cost 27-33 us instead of the burst's 132 us — the mechanism works — but the
// 256x128 main loop that leaves room for the parking area runs its k-steps at 40 % of the matrix pipe against the 256x256 tile's
// 61 % (156 vs 103 us compute-only: half the MFMA work per barrier, the same fixed cost per k-step), and the epilogue arithmetic,
// which the 256x256 kernel hides behind its stalled stores, is on every wave's critical path here (+63 us).  What would be needed
// is the 128x64 wave tile AND a parked tile: 128 accumulator registers + 128 KB of LDS ring + 128 KB of parking do not fit a CU.
//
// Why: gemm_big_kernel (256x256, one workgroup per CU) ends every tile with all eight waves storing at once — the matrix
// pipe idles while 256 workgroups burst 33-66 MB at the chip's write ceiling, and the stores queue in front of the next
// tile's operand loads in each CU's memory pipeline (main loop and store time ADD: 65536x2048x512 with
// GELU + pre-activation 114 us without its stores, 226-268 us with them).  Every attempt to overlap the phases from
// outside (two workgroups per CU, staggered groups, store slots, deeper prefetch) failed because a finished tile has to
// be HELD somewhere while it drains at the rate HBM takes it.  Here it is held in LDS:
//
//   tile      : 256(m) x 128(n), 8 waves as 4 x 2 of 64 x 64 (64 accumulator registers per lane), one workgroup per CU
//   LDS       : two 48 KB operand stages (LDS-DMA ring, one k-step of 64 in flight) + a 64 KB PARKING area = 160 KB:
//               each wave parks its finished 64 x 64 sub-tile as bf16(acc + bias) — 8 KB, wave-private, no workgroup
//               barrier — and starts the next tile's main loop at once
//   trickle   : the parked tile leaves in 8 pieces per lane (8 rows x 128 bytes per wave and piece = whole cache lines),
//               one piece per k-step at K = 512 (two at K = 256, one every K / 512 steps beyond): LDS read, epilogue
//               arithmetic (GELU, gelu', dropout, residual), 16-byte streaming stores.  Waves 0-3 do the arithmetic
//               before their MFMA items, waves 4-7 behind them (waves w and w + 4 share a SIMD: one's vector work runs
//               under the other's matrix work); all stores go out at the END of the step, behind the step's DMA
//               requests, so that the next step's counted wait (`vmcnt(stores)`) covers the DMA and leaves the stores
//               in flight — store latency is never waited for, the operand feed never queues behind a burst
//   pipeline  : the ring runs ACROSS tiles (the last k-step of a tile requests the first stage of the next one)
//   order     : a workgroup keeps its column tile (grid = a multiple of the column-tile count): bias and the lane's
//               columns are fixed for the launch
//
// Numerics: the parked value is the bf16 ROUNDING of acc + bias — what the pre-activation / plain output stores anyway;
// the GELU / gelu' / dropout / residual epilogues therefore act on that rounded value (as a chain of bf16 tensor ops
// would).  gemm_kernel's and gemm_big_kernel's bf16 epilogues round at the same point (dl_round_store<T>), so the three
// kernels stay bit-identical to each other (tests/test_kernels_gpu.py::test_large_tile_gemm_is_bitwise_equal_*).
//
// Waits: vmcnt counts loads, LDS-DMA and stores in issue order (the compiler relies on the same rule for its own counted
// waits on this target); the DMA instructions are inline assembly the compiler does not see, so every compiler-visible
// load whose use lies behind DMA requests is "touched" (empty asm) directly behind the step's own counted wait — the
// compiler then places its wait there, with the same count.
#pragma once

template <int N> __device__ __forceinline__ void touch_regs(u32x4 (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

// one 8-column piece of a parked row: `parked` = 8 bf16 of round(acc + bias); returns the 8 output values packed
template <int EPI>
__device__ __forceinline__ u32x4 trickle_math8(const GemmP& p, uint64_t seed_eff, int m, int n, u32x4 parked, u32x4 ext) {
  typedef bf16_t T;
  if constexpr (EPI == 0) return parked;
  f32x4 a0 = {bf16lo(parked[0]), bf16hi(parked[0]), bf16lo(parked[1]), bf16hi(parked[1])};
  f32x4 a1 = {bf16lo(parked[2]), bf16hi(parked[2]), bf16lo(parked[3]), bf16hi(parked[3])};
  if constexpr (EPI == 2) {
    a0 = gelu4<T>(a0); a1 = gelu4<T>(a1);
  } else if constexpr (EPI == 4) {
    a0 *= gelu_grad4<T>(f32x4{bf16lo(ext[0]), bf16hi(ext[0]), bf16lo(ext[1]), bf16hi(ext[1])});
    a1 *= gelu_grad4<T>(f32x4{bf16lo(ext[2]), bf16hi(ext[2]), bf16lo(ext[3]), bf16hi(ext[3])});
  }
  if (p.drop_thr16) {
    a0 = dl_dropout4(a0, seed_eff, (uint64_t)m, (uint64_t)n, (uint64_t)p.N, p.drop_thr16, p.drop_inv_keep);
    a1 = dl_dropout4(a1, seed_eff, (uint64_t)m, (uint64_t)(n + 4), (uint64_t)p.N, p.drop_thr16, p.drop_inv_keep);
  }
  if constexpr (EPI == 3) {
    a0 += f32x4{bf16lo(ext[0]), bf16hi(ext[0]), bf16lo(ext[1]), bf16hi(ext[1])};
    a1 += f32x4{bf16lo(ext[2]), bf16hi(ext[2]), bf16lo(ext[3]), bf16hi(ext[3])};
  }
  return u32x4{pack_bf16x2(a0[0], a0[1]), pack_bf16x2(a0[2], a0[3]), pack_bf16x2(a1[0], a1[1]), pack_bf16x2(a1[2], a1[3])};
}

// PPK: pieces per k-step (2: K = 256, four steps per tile; 1: K a multiple of 512, a piece every K / 512 steps)
template <int EPI, int PPK>
__global__ __launch_bounds__(512, 2) void gemm_trickle_kernel(const GemmP p) {
  typedef bf16_t T;
  constexpr int XF = 4, WF = 4, NWN = 2, ROWB = 128;
  constexpr int NT = 512, BM = 256, BN = 128, KCH = ROWB / 16, NKF = ROWB / 64, BKE = ROWB / 2;
  constexpr int XB = BM * ROWB, WB = BN * ROWB, STAGE = XB + WB;
  constexpr int XCH = BM * KCH / NT, WCH = BN * KCH / NT, PER = XCH + WCH;        // 4 + 2 DMA instructions per thread and step
  constexpr int PARK = 2 * STAGE, PARKW = 64 * 128;                              // per wave: 64 rows x 64 bf16
  constexpr int NPIECE = 8;                                                      // pieces per lane and tile
  constexpr int NST = (EPI == 2) ? 2 : 1;                                        // stores per piece
  constexpr bool EXT = (EPI == 3 || EPI == 4);
  constexpr int U = NKF * XF;
  static_assert(PER <= U, "one DMA instruction per MFMA item");
  static_assert(PARK + 8 * PARKW <= 163840, "LDS");
  __shared__ __attribute__((aligned(16))) char smem[PARK + 8 * PARKW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (an SGPR: branches on it are scalar, tile pointers stay uniform)
  const int il = lane & 15, g = lane >> 4;
  const int wm = wave / NWN, wn = wave % NWN;
  const int swz = il & 7;
  const int xoff = (wm * 16 * XF + il) * ROWB, woff = XB + (wn * 16 * WF + il) * ROWB;
  char* const park = smem + PARK + wave * PARKW;
  const bool early = wave < 4;                     // (waves w and w + 4 share a SIMD)
  const uint64_t seed_eff = (EPI != 0 && p.drop_thr16) ? dl_eff_seed(p.seed, p.seed_off) : 0;
  const uint32_t ntiles = (uint32_t)p.mt * p.nt;
  const uint32_t G = gridDim.x;                    // a multiple of p.nt (host): a workgroup keeps its column tile in whole rounds
  auto locate = [&](uint32_t it, int& m0, int& n0) {
    const uint32_t round0 = (it / G) * G;
    const uint32_t span = min(G, ntiles - round0);
    const uint32_t t = round0 + xcd_remap(it - round0, span);
    m0 = (int)(t / p.nt) * BM; n0 = (int)(t % p.nt) * BN;
  };
  // LDS-DMA addresses in the SGPR-base + 32-bit VGPR-offset form (gemm_big.cuh: dma16s): piece q of a step covers rows
  // q * 64 + tid / 8 of the operand tile, 16-byte chunk (tid % 8) ^ (row & 7) — the lane's offset inside a piece is the same for
  // every piece, step and tile (M and N are whole tiles: no row clamp), the piece's first byte is wave-uniform
  const int drow = tid / KCH, dkc = (tid % KCH) ^ (drow & 7);
  const uint32_t xvoff = (uint32_t)(drow * (int)(p.ldx * 2) + dkc * 16), wvoff = (uint32_t)(drow * (int)(p.ldw * 2) + dkc * 16);
  const uint32_t smem_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  int fm0 = 0, fn0 = 0;                            // the tile the feed addresses
  auto point = [&](int m0, int n0) { fm0 = m0; fn0 = n0; };
  auto issue_piece = [&](int kt, uint32_t buf, int q) {
    const uint32_t sb = smem_lds + buf * STAGE;
    const int64_t kb = (int64_t)kt * ROWB;
    if (q < XCH) {
      dma16s(xvoff, p.X + (int64_t)(fm0 + q * (NT / KCH)) * p.ldx * 2 + kb,
             __builtin_amdgcn_readfirstlane(sb + (uint32_t)((wave * 64 + q * NT) * 16)));
    } else {
      dma16s(wvoff, p.W + (int64_t)(fn0 + (q - XCH) * (NT / KCH)) * p.ldw * 2 + kb,
             __builtin_amdgcn_readfirstlane(sb + (uint32_t)(XB + (wave * 64 + (q - XCH) * NT) * 16)));
    }
  };
  const int nk = p.K / BKE;                        // 4 (PPK = 2) or a multiple of 8 (PPK = 1): checked by the host
  const int ksp = PPK == 2 ? 1 : nk / NPIECE;      // k-steps per piece step

  uint32_t it = blockIdx.x;
  if (it >= ntiles) return;
  int m0, n0;
  locate(it, m0, n0);
  point(m0, n0);
#pragma unroll
  for (int q = 0; q < PER; ++q) issue_piece(0, 0, q);
  uint32_t gs = 0;                                 // global step counter: stage buffer = gs & 1

  // bias for the parking pass: lane L keeps the value of column n0 + wn * 64 + L (ONE register); a lane's 16 values (columns
  // j * 16 + 4 g .. + 3) come through the LDS crossbar (ds_bpermute) when a tile is parked — 16 registers held across the main
  // loop were spilled to scratch, and a scratch reload is a vmcnt(0) drain per tile
  float pbl = 0.f;
  auto load_bias = [&](int n0_) {
    pbl = (EPI != 4 && p.bias) ? p.bias[n0_ + wn * 16 * WF + lane] : 0.f;
    // (drained here, once: a pending load at the loop's merge points would make the compiler wait in every step)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(pbl));
  };
  load_bias(n0);

  // the tile whose parked outputs are being trickled
  bool have_prev = false;
  int pm0 = 0, pn0 = 0;
  u32x4 ext[PPK];
#pragma unroll
  for (int k = 0; k < PPK; ++k) ext[k] = u32x4{0u, 0u, 0u, 0u};
  const int c8 = lane & 7, r8 = lane >> 3;
  auto piece_row = [&](int pc) { return r8 + 8 * pc; };
  // global addresses of a piece: a wave-uniform base (tile, wave, piece) + ONE 32-bit lane offset per tensor (row r8 of the
  // piece's 8 rows, columns c8 * 8 ..): 64-bit per-lane pointers for four tensors were what spilled
  const uint32_t lo_c = (uint32_t)((r8 * (int)p.ldc + c8 * 8) * 2);
  const uint32_t lo_p = EPI == 2 ? (uint32_t)((r8 * (int)p.ldp + c8 * 8) * 2) : 0u;
  const int64_t ld_e = EPI == 3 ? p.ldr : p.lddp;
  const uint32_t lo_e = EXT ? (uint32_t)((r8 * (int)ld_e + c8 * 8) * 2) : 0u;
  auto ext_load = [&](int tm0, int tn0, int pc) -> u32x4 {
    u32x4 v = {0u, 0u, 0u, 0u};
    if constexpr (EXT) {
      const char* base = (EPI == 3 ? p.res : p.dact_pre) + ((int64_t)(tm0 + wm * 64 + 8 * pc) * ld_e + tn0 + wn * 64) * 2;
      v = *reinterpret_cast<const u32x4*>(base + lo_e);
    }
    return v;
  };
  auto park_read = [&](int pc) -> u32x4 {
    const int r = piece_row(pc);
    return lds_read16(park, r * 128 + ((c8 ^ ((r >> 1) & 7)) << 4));
  };
  auto store_piece = [&](int pc, u32x4 o, u32x4 pre) {
    if (!(DL_DBG(p) & 1)) {
      const int64_t row0 = pm0 + wm * 64 + 8 * pc;
      if constexpr (EPI == 2) {
        char* pd = p.pre_out + (row0 * p.ldp + pn0 + wn * 64) * 2 + lo_p;
        if (p.nt_pre) store16_nt(pd, pre, p.nt_pre); else *reinterpret_cast<u32x4*>(pd) = pre;
      }
      char* dstp = p.C + (row0 * p.ldc + pn0 + wn * 64) * 2 + lo_c;
      if (p.nt_c) store16_nt(dstp, o, p.nt_c); else *reinterpret_cast<u32x4*>(dstp) = o;
    }
  };

  bool prev_stores = false;                        // the previous k-step ended with NST * PPK stores behind its DMA requests
  for (;;) {
    f32x4 acc[XF][WF];
#pragma unroll
    for (int i = 0; i < XF; ++i)
#pragma unroll
      for (int j = 0; j < WF; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t itn = it + G;
    const bool have_next = itn < ntiles;
    int nm0 = m0, nn0 = n0;

    for (int kt = 0; kt < nk; ++kt) {
      // step kt of this tile has landed once only the stores of the previous step are outstanding (they are younger than its
      // DMA requests); every wave is past step kt - 1: its buffer takes the next step
      if (prev_stores) wait_vmcnt<NST * PPK>(); else wait_vmcnt<0>();
      wg_barrier();
      if constexpr (EXT) touch_regs(ext);
      const bool last = kt == nk - 1;
      if (last && have_next) { locate(itn, nm0, nn0); point(nm0, nn0); }
      const bool feed = (!last || have_next) && !(DL_DBG(p) & 2);      // (study bit 2: no operand feed after the prologue)
      const int fkt = last ? 0 : kt + 1;
      const uint32_t fbuf = (gs + 1) & 1;
      const char* cur = smem + (gs & 1) * STAGE;
      // pieces of the parked tile handled in this step: [plo, plo + PPK)
      const bool is_piece = have_prev && (PPK == 2 || kt % ksp == 0);
      const int plo = PPK == 2 ? kt * 2 : kt / ksp;
      u32x4 out_o[PPK], out_p[PPK];
#pragma unroll
      for (int k = 0; k < PPK; ++k) { out_o[k] = u32x4{0u, 0u, 0u, 0u}; out_p[k] = out_o[k]; }
      auto compute = [&]() {
#pragma unroll
        for (int k = 0; k < PPK; ++k) {
          const int pc = plo + k;
          out_p[k] = park_read(pc);
          if (DL_DBG(p) & 8) out_o[k] = out_p[k];      // (study bit 8: no epilogue arithmetic)
          else out_o[k] = trickle_math8<EPI>(p, seed_eff, pm0 + wm * 64 + piece_row(pc), pn0 + wn * 64 + c8 * 8, out_p[k], ext[k]);
        }
      };
      if (is_piece && early) compute();
      __builtin_amdgcn_sched_barrier(0);
      // the MFMA items with the next step's DMA requests one per item (compile-time FEED: a branch inside would cut the
      // scheduling region the sched_group_barrier pattern pins)
      auto mfma_block = [&](auto feed_t) __attribute__((always_inline)) {
        constexpr bool FEED = decltype(feed_t)::value;
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
          if (u < XF) { fw[1][u] = rd_fw(1, u); ++nread; }
          if (FEED && u < PER) issue_piece(fkt, fbuf, u);
#pragma unroll
          for (int j = 0; j < WF; ++j) acc[u % XF][j] = Mma<T>::mma(fw[u / XF][j], fx[u], acc[u % XF][j]);
          if (nread == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          else if (nread == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (FEED && u < PER) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, WF, 0);
        }
      };
      if (feed) mfma_block(std::true_type{}); else mfma_block(std::false_type{});
      __builtin_amdgcn_sched_barrier(0);
      ++gs;
      if (is_piece && !early) compute();
      // end of the step: the NEXT piece step's epilogue operands, then this step's stores — all behind the DMA requests
      if (is_piece) {
        if constexpr (EXT) {
#pragma unroll
          for (int k = 0; k < PPK; ++k) {
            const int pcn = plo + PPK + k;                         // the next piece step's pieces ...
            ext[k] = pcn < NPIECE ? ext_load(pm0, pn0, pcn) : ext_load(m0, n0, pcn - NPIECE);   // ... or the first ones of THIS tile
          }
        }
#pragma unroll
        for (int k = 0; k < PPK; ++k) store_piece(plo + k, out_o[k], out_p[k]);
      }
      prev_stores = is_piece;
    }

    // ---- tile end: park round(acc + bias); the next tile's first stage is already in flight ----------------------
    f32x4 pbq[WF];
#pragma unroll
    for (int j = 0; j < WF; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) pbq[j][r] = __shfl(pbl, j * 16 + 4 * g + r, 64);
#pragma unroll
    for (int i = 0; i < XF; ++i) {
      const int row = i * 16 + il;
#pragma unroll
      for (int j = 0; j < WF; ++j) {
        const f32x4 v = acc[i][j] + pbq[j];
        const u32x2 w = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        const int chunk = j * 2 + (g >> 1);
        *reinterpret_cast<u32x2*>(park + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4) + (g & 1) * 8) = w;
      }
    }
    wave_sync();                                   // lanes exchange rows through the wave's slice
    if (!have_prev) {
      // first tile of this workgroup: nothing was trickled during it, so the epilogue operands of its first pieces have not
      // been requested yet (one drained wait per launch)
      if constexpr (EXT) {
#pragma unroll
        for (int k = 0; k < PPK; ++k) ext[k] = ext_load(m0, n0, k);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        touch_regs(ext);
      }
    }
    pm0 = m0; pn0 = n0; have_prev = true;
    if (!have_next) break;
    if (nn0 != n0) load_bias(nn0);                 // (only in a partial last round: the column tile moved)
    it = itn; m0 = nm0; n0 = nn0;
  }
  // ---- the last tile's outputs: nothing left to hide them behind ---------------------------------------------------
#pragma unroll
  for (int k = 0; k < PPK; ++k) {                  // (their epilogue operands were requested with the last piece step)
    const u32x4 pk = park_read(k);
    const u32x4 o = trickle_math8<EPI>(p, seed_eff, pm0 + wm * 64 + piece_row(k), pn0 + wn * 64 + c8 * 8, pk, ext[k]);
    store_piece(k, o, pk);
  }
#pragma unroll 1
  for (int pc = PPK; pc < NPIECE; ++pc) {
    const u32x4 e = ext_load(pm0, pn0, pc);
    const u32x4 pk = park_read(pc);
    const u32x4 o = trickle_math8<EPI>(p, seed_eff, pm0 + wm * 64 + piece_row(pc), pn0 + wn * 64 + c8 * 8, pk, e);
    store_piece(pc, o, pk);
  }
}
