// The row-team sweep of SHORT row blocks (round 6): gemv_tnp1_kernel -- one WAVE per workgroup holds the whole column of its
// device's block (up to 8 row groups of 1 KiB: 2048 rows of Float32, 1024 of Float64; the headline's block at N = 8).  Included
// inside `namespace pgtn { namespace {` by pg_gemv_tn4.hip after pg_gemv_tnt.h, whose PROTOCOL it speaks unchanged (ring layout,
// tagged 8-byte granules, epochs, PEER_RING slots, the DELAY stamps, the telemetry words): a device running this kernel and
// gemv_tnt_kernel<..., WAVES = 1, PEER> produce the same granules.  What changed is the instruction stream.
//
// Why.  With one wave per SIMD a wave issues at most one instruction every four cycles and nothing hides what it waits for.
// Round 5's sweep spent ~950 instructions per 16 KiB step (profiles/r6_peer_sweep_counters.md): the epilogue of every column ran
// in every lane, one column after the other, with its branches; the member-order sum was a v_readlane + v_add per member and
// column behind a jump table; the column map was evaluated three times; the wave sum crossed LDS twice per column and the
// partial dots once more (five exposed LDS round trips per step).  Here:
//   * dots as packed partial sums (v_pk_fma_f32: two multiply-adds per instruction), reduced ACROSS the C columns at once
//     (cr_stage, pg_lanes.h: permlane swaps and DPP moves, no LDS);
//   * the step's granules are summed over the members by DPP row shifts -- lane c*G of every 16-lane row adds the lanes
//     c*G + k*C*G of its row, k ascending (= member order), rows combined by two permlane swaps -- TM - 1 adds for all columns
//     together, the same order on every device;
//   * ONE epilogue per step, lane-parallel (lane c*G works on column c): one load each of x, z_old (and the per-element
//     parameters), one store per output vector, no per-column branch;
//   * v_j returns through v_readlane (a scalar operand of the multiply-adds), as in gemv_tnw_kernel.
// Reference statements as for gemv_tnt_kernel: benchmark/benchmarks.jl:15-16, fast_forward_backward.jl:135-142
// (forward_backward.jl:113-120).
#pragma once

template <typename T>
struct Pair2;
template <>
struct Pair2<float> {
  typedef float type __attribute__((ext_vector_type(2)));
};
template <>
struct Pair2<double> {
  typedef double type __attribute__((ext_vector_type(2)));
};

// lane l <- lane l + K of the same 16-lane row (0 past the end of the row): DPP row_shl
template <int K, typename T>
__device__ __forceinline__ T row_from_above(T v) {
  static_assert(K >= 1 && K <= 15, "a shift inside one row");
  return pg_dpp_mov<0x100 + K>(v);
}

// acc += sum over k = 1 .. N - 1 of the lanes S * k above (ascending): the members of one row, in member order
template <int S, int K, typename T>
__device__ __forceinline__ void row_member_sum(T& acc, const T val) {
  if constexpr (S * K <= 15) {
    acc += row_from_above<S * K>(val);
    row_member_sum<S, K + 1>(acc, val);
  }
}

// PAIR: one post per TWO steps.  A device keeps the granules of an even step until the odd step behind it has its own and writes
// both with one store of 16 * S bytes per inbox (slot (i / 2) % RING, member p at p * 2 S, the even step's granules first; tag =
// epoch | i / 2 + 1); a consumer polls the pair's slot at both steps and takes its half.  Half the fabric transactions (the first
// knob on real xGMI) for one step less of slack on the even steps (their granules leave one step later).
//
// AHEAD: the poll of a step's totals (and the fetch of its x_j, z_old_j) is issued ONE STEP before its use, ahead of that step's tile
// loads.  Vector loads return in issue order: a poll issued at the start of the step that consumes it sits BEHIND the loads of the
// tile the next step will dot, so reading it waits for that tile too -- one of the two tiles "in flight" is then always complete
// when the multiply-adds run, and the wave streams with half its depth.  Issued a step earlier the poll returns with a tile that
// is needed anyway; the price is a first look one step sooner after the post (LT - 1 steps of slack instead of LT).
#ifndef PG_TNP1_ATTR
#ifdef PG_TNT_EXPERIMENT
// (experiment: the C = 1 geometries are compiled for TWO waves per SIMD -- half the register file each)
#define PG_TNP1_ATTR __attribute__((amdgpu_waves_per_eu((C == 1 ? 2 : 1), (C == 1 ? 2 : 8))))
#else
#define PG_TNP1_ATTR  // (kernel lab: e.g. -DPG_TNP1_ATTR='__attribute__((amdgpu_waves_per_eu(2,2)))')
#endif
#endif
#ifdef PG_TNT_EXPERIMENT
#define PG_TNP1_BLOCK 128  // (timing experiment dbg & 2048: a second wave that does the stores)
#else
#define PG_TNP1_BLOCK 64
#endif
template <typename T, int U, int C, int LAG, int PF, int LAGR, bool DELAY, bool PAIR = false, bool AHEAD = false>
__global__ __launch_bounds__(PG_TNP1_BLOCK) PG_TNP1_ATTR void gemv_tnp1_kernel(TNArgs<T> a) {
  using V = typename VecOf<T>::type;
  using T2 = typename Pair2<T>::type;
  constexpr int VEC = VecOf<T>::N;
  constexpr int G = (int)sizeof(T) / 4;  // granules per value
  constexpr int S = C * G;               // granules a member posts per step
  constexpr int MS = PAIR ? 2 * S : S;   // granules of one member in a ring slot (PAIR: two steps share a slot)
  constexpr int GL = 64 / C;             // lanes that hold the same column after the column-parallel reduction
  constexpr int RING = PEER_RING;
  constexpr int LT = LAG + LAGR;
#ifdef PG_TNP1_LOADS_AFTER_DOTS
  constexpr bool LOADS_FIRST = false;  // (experiment, not kept: 6.15 TB/s against 6.44 with the loads first -- an early issue is worth more than counter headroom)
#else
  constexpr bool LOADS_FIRST = true;
#endif
  static_assert(LT > 0 && 2 * LT + 2 <= RING, "the totals of a step are consumed LT > 0 steps later; the ring holds 2 LT + 2 steps");
  static_assert(MS <= 8 && (C & (C - 1)) == 0 && C >= 1, "C a power of two, at most eight granules per member and ring slot");
  extern __shared__ __attribute__((aligned(16))) unsigned char park_raw[];
  V* const park = reinterpret_cast<V*>(park_raw);  // [LAG][C][U][64]
#ifdef PG_TNT_EXPERIMENT
  const int lane = threadIdx.x & 63;
#else
  const int lane = threadIdx.x;
#endif
  const int team = (int)blockIdx.x;
  const int member = a.peer_rank;
  const int TM = a.peer_n;
  const int npoll = TM * MS;
  const int64_t ncg = (a.n + C - 1) / C;
  const CgMap map(ncg, C, a.line_cols, team, a.nteams);
  const int64_t cnt = map.cnt;
  const size_t ring_off = (size_t)team * RING * (size_t)(TEAM_MAX * MS);
  auto slot_of = [&](int64_t i) __attribute__((always_inline)) -> size_t { return (size_t)((PAIR ? (i >> 1) : i) % RING) * (TEAM_MAX * MS); };
  auto tag_of = [&](int64_t i) __attribute__((always_inline)) -> unsigned { return a.tag_base + (unsigned)((PAIR ? (i >> 1) : i) + 1); };
  unsigned long long* const ring = a.xch + ring_off;

  V rk[U], racc[U];
  int rgc[U];  // row groups past the end of this device's block are clamped to its last one; their r is zero
#pragma unroll
  for (int u = 0; u < U; ++u) {
    rgc[u] = max(0, min(u, a.nrg - 1));
#pragma unroll
    for (int e = 0; e < VEC; ++e) racc[u][e] = T(0);
    if (u < a.nrg) {
      rk[u] = *reinterpret_cast<const V*>(a.r + (int64_t)u * (WAVE * VEC) + lane * VEC);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) rk[u][e] = T(0);
    }
  }
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  bool dead = false;
  unsigned late_steps = 0, late_polls = 0;
  unsigned long long age_sum = 0;
  unsigned age_cnt = 0;
  const bool has_pv = a.p0v != nullptr;

  // where this lane posts: lane q * MS + s2 (q < TM, s2 < MS) writes granule s2 of this device's step (PAIR: of its two steps, the even
  // one's S granules first) into member q's inbox (its own included) -- the granules reach every inbox with ONE store instruction,
  // 8 * MS contiguous bytes per inbox (DELAY: the lanes behind them carry this device's stamp, one per inbox)
  const int npoll_all = DELAY ? npoll + TM : npoll;
  const int post_s = (lane % MS) % S;   // the granule of a step this lane carries: half post_s % G of column post_s / G
  const int post_h = (lane % MS) / S;   // PAIR: 0 = the even step of the pair, 1 = the odd one
  unsigned long long* post_ptr = nullptr;
  {
    const int q_of_lane = lane < npoll ? lane / MS : lane - npoll;
#pragma unroll
    for (int q = 0; q < TEAM_MAX; ++q)
      if (q < TM && q_of_lane == q) post_ptr = a.peer_ring[q];
    post_ptr += ring_off + (lane < npoll ? (size_t)member * MS + (size_t)(lane % MS) : (size_t)TM * MS + (size_t)member);
  }
  unsigned bits_even = 0;  // PAIR: this lane's granule of the even step, kept until the odd step's is there
#ifdef PG_TNT_EXPERIMENT
  if (threadIdx.x >= 64) {
    // timing experiment (dbg & 2048, with 1 | 8 | 16 in the sweeping wave): a FREE-RUNNING second wave issues the sweep's stores -- one post
    // per step, five output lines per chunk, paced by the clock at a.delay_ticks per step (values are garbage; the sweeping wave never
    // waits).  Does the sweep keep the rate it has without stores (then the stores cost QUEUE ORDER in the sweeping wave and a writer wave
    // fed through LDS recovers it), or does it fall back (then they cost on the memory side and nothing in the kernel helps)?
    if (!(a.dbg & 2048)) return;
    const int wl = (int)threadIdx.x - 64;
    unsigned long long* wp = nullptr;
    {
      const int q_of_lane = wl < npoll ? wl / MS : wl - npoll;
#pragma unroll
      for (int q = 0; q < TEAM_MAX; ++q)
        if (q < TM && q_of_lane == q) wp = a.peer_ring[q];
      wp += ring_off + (wl < npoll ? (size_t)member * MS + (size_t)(wl % MS) : (size_t)TM * MS + (size_t)member);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int KCw = 1 << map.shift;
    for (int64_t i = 0; i < cnt; ++i) {
      while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)i * a.delay_ticks) __builtin_amdgcn_s_sleep(8);
      if (a.dbg & 4096) continue;  // (control: the second wave is there and stores nothing)
      if (wl < npoll_all) __hip_atomic_store(wp + slot_of(i), (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const bool chunked = i < (int64_t)map.head;
      const int k = chunked ? (int)(i & (int64_t)(KCw - 1)) : 0, ncols = chunked ? KCw * C : C;
      if (k == ncols / C - 1) {
        const int64_t j = map.at(i - k) * C + min(wl, ncols - 1);
        if (wl < ncols && j < a.n) {
          const T v = (T)(float)i;
          a.g_out[j] = v, a.y[j] = v, a.z_new[j] = v, a.res[j] = v;
          if (a.v_out != nullptr) a.v_out[j] = v;
        }
      }
    }
    return;
  }
#endif

  struct Tile {
    V col[C][U];
  };
  struct Pend {
    unsigned long long w;  // this lane's granule of the awaited step
    T xs, zos, q0, q1;     // x_j, z_old_j and the per-element parameters of this lane's column
  };
  Pend pn{};  // AHEAD: the poll and the small loads of the NEXT step's totals, in flight across a step
  auto load = [&](Tile& t, int64_t i) __attribute__((always_inline)) {
    const int64_t j0 = map.at(i) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t j = (j0 + c < a.n) ? (j0 + c) : (a.n - 1);
      const T* __restrict__ p = a.A + j * a.ld;
#pragma unroll
      for (int u = 0; u < U; ++u)
        t.col[c][u] = nt_load(reinterpret_cast<const V*>(p + (int64_t)rgc[u] * (WAVE * VEC)) + lane);
    }
  };
  // this device's partial dots of step i -> the inbox of every device
  auto dot_post = [&](const Tile& t, int64_t i) __attribute__((always_inline)) {
    T d[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      T2 s2 = {T(0), T(0)};
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < VEC; e += 2) {
          const T2 x2 = {t.col[c][u][e], t.col[c][u][e + 1]}, r2 = {rk[u][e], rk[u][e + 1]};
          s2 = __builtin_elementwise_fma(x2, r2, s2);
        }
      }
      d[c] = s2[0] + s2[1];
    }
    cr_stage<T, C, 0>(d, lane);  // d[0] = the total of column lane / GL, in every lane of that group
    // ONE store for all inboxes: lane q * S + s carries granule s (half s % G of column s / G) to member q
    // (every posting lane takes its column's total from that column's lane group -- its own d[0] is the total of column lane / GL,
    // which is the column it carries only while all posting lanes sit in group 0)
    T mine = pg_readlane(d[0], 0);
#pragma unroll
    for (int c = 1; c < C; ++c) {
      const T tc = pg_readlane(d[0], c * GL);
      if (post_s / G == c) mine = tc;
    }
    unsigned bits;
    if constexpr (G == 1) {
      bits = __builtin_bit_cast(unsigned, mine);
    } else {
      const unsigned long long b = __builtin_bit_cast(unsigned long long, mine);
      bits = (post_s % G) == 0 ? (unsigned)b : (unsigned)(b >> 32);
    }
    if constexpr (PAIR) {
      const bool odd = (i & 1) != 0;
      if (!odd) bits_even = bits;
      if (!odd && i + 1 < cnt) return;  // the even step of a pair: its granules leave with the odd step's (a last step without a partner: now)
      bits = post_h == 0 ? bits_even : (odd ? bits : 0u);
    }
    if constexpr (DELAY) {
      if (lane >= npoll) bits = (unsigned)__builtin_amdgcn_s_memrealtime();  // the stamp granules: behind the TM members' values, one per member
    }
    const unsigned long long word = ((unsigned long long)tag_of(i) << 32) | bits;
#ifdef PG_TNT_EXPERIMENT
    if constexpr (G == 1 && C == 2 && !PAIR && !DELAY) {
      if (a.dbg & 65536) {
        // experiment: the post as SCALAR stores (one s_store_dwordx4 of this device's two granules per inbox, then s_dcache_wb) -- outside the
        // wave's vector-memory queue; scripts/kernel_lab/scalar_store_probe.hip: such a store reaches uncached memory while the kernel runs
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        const unsigned tg = __builtin_amdgcn_readfirstlane(tag_of(i));
        u4 w;
        w.x = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, pg_readlane(d[0], 0)));
        w.y = tg;
        w.z = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, pg_readlane(d[0], GL)));
        w.w = tg;
        for (int q = 0; q < TM; ++q) {
          unsigned long long* dst = a.peer_ring[q] + ring_off + slot_of(i) + (size_t)member * MS;
          asm volatile("s_store_dwordx4 %0, %1, 0x0 glc" ::"s"(w), "s"(dst) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
        return;
      }
    }
    if (a.dbg & (8 | 256)) return;  // timing experiment: no post
    if (a.dbg & 64) {  // timing experiment (solo): a FULL 64-byte line per post instead of 8 * MS bytes
      if (lane < 8) __hip_atomic_store(a.peer_ring[0] + ring_off + slot_of(i) + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    if (a.dbg & 8192) {
      // timing experiment (solo): the LANDING traffic of a team of a.delay_ticks devices -- this device's inbox receives that many 8 * MS-byte
      // pieces per step, each in another line (member q's piece of step i + q, as arrivals spread over time would be); bit 16384: the same
      // bytes as ONE contiguous store (what a merged arrival would be)
      const int fake = (int)a.delay_ticks;
      if (a.dbg & 16384) {
        if (lane < fake * MS) __hip_atomic_store(a.peer_ring[0] + ring_off + slot_of(i) + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
      }
      for (int q = 0; q < fake; ++q)
        if (lane < MS) __hip_atomic_store(a.peer_ring[0] + ring_off + slot_of(i + q) + (size_t)q * MS + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    if (a.dbg & 128) {  // timing experiment: the post as a plain (cached, write-back) store
      if (lane < npoll_all) post_ptr[slot_of(i)] = word;
      return;
    }
#endif
    if (lane < npoll_all)
      __hip_atomic_store(post_ptr + slot_of(i), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  const int poll_lane = lane < npoll_all ? lane : npoll_all - 1;
  const bool adds = lane < npoll && (lane % G) == 0;  // this lane's granule (pair) is a member's value of some column
  auto arrived = [&](unsigned long long w, unsigned tag) __attribute__((always_inline)) -> bool {
    bool ok = (unsigned)(w >> 32) == tag;
    if constexpr (DELAY) {
      const unsigned now = (unsigned)__builtin_amdgcn_s_memrealtime();
      if (lane >= npoll && lane < npoll_all) ok = ok && (now - (unsigned)w) >= a.delay_ticks;
    }
    return __builtin_amdgcn_ballot_w64(ok) == ~0ull;
  };
  auto poll_word = [&](int64_t i) __attribute__((always_inline)) -> unsigned long long {
#ifdef PG_TNT_EXPERIMENT
    if (a.dbg & (8 | 512)) return 0ull;  // timing experiment: no poll
#endif
    return __hip_atomic_load(ring + slot_of(i) + poll_lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  };
  // Whole lines of the output vectors.  A wave works through CHUNKS of line_cols = 32 consecutive columns (CgMap: K = 32 / C steps in
  // a row, then a jump); its epilogue of step k of a chunk runs in lanes k * C + c, which have loaded x, z_old (and the per-element
  // parameters) of THEIR column of the chunk, and what it produces (g, y, z, res, v) is collected lane by lane until the chunk's
  // last step writes five full 128-byte lines with 32 lanes -- round 6's first form stored five 8-byte pieces per step, eighty
  // partial writes per chunk, and took its scalar reductions (double precision) every step instead of once per chunk:
  // 5 % of the sweep (profiles/r6_peer_sweep_counters.md).  The steps behind the chunked part of a unit's work (CgMap's tail: single groups)
  // are chunks of one step.
  const int KC = 1 << map.shift;  // steps per chunk
  auto chunk_pos = [&](int64_t i) __attribute__((always_inline)) -> int { return i < (int64_t)map.head ? (int)(i & (int64_t)(KC - 1)) : 0; };
  auto chunk_cols = [&](int64_t i) __attribute__((always_inline)) -> int { return i < (int64_t)map.head ? KC * C : C; };
  T out_g = T(0), out_y = T(0), out_z = T(0), out_r = T(0), out_v = T(0);
  // the small loads of step i's epilogue, issued with the poll (before the tile loads: they return first): lane l asks for x and z_old of
  // column l of step i's chunk -- every step of the chunk asks for the same addresses again (cache hits; no state to carry and NO
  // BRANCH: a load behind a branch inside the steady loop makes the compiler give up counting the loads in flight -- tried here
  // as "once per chunk": its waits fell from vmcnt(34..51) to vmcnt(17), one tile in flight instead of two)
  auto fetch = [&](Pend& pd, int64_t i) __attribute__((always_inline)) {
#ifdef PG_TNT_EXPERIMENT
    if (a.dbg & 32) return;  // timing experiment: no small loads
#endif
    const int k = chunk_pos(i);
    const int64_t j = map.at(i - k) * C + min(lane, chunk_cols(i) - 1);
    const int64_t jc = j < a.n ? j : a.n - 1;
    pd.xs = a.x[jc];
    pd.zos = a.z_old[jc];
  };
  // totals of step i (all members have posted, or will shortly) -> epilogue -> v_j (0 for columns past the end)
  auto totals = [&](int64_t i, Pend& pd, T (&vj)[C]) __attribute__((always_inline)) {
    const unsigned tag = tag_of(i);
#ifdef PG_TNT_EXPERIMENT
    if (a.dbg & 1) dead = true;  // timing experiment: never wait (totals are then wrong)
#endif
    // The first look at the granules stays OUTSIDE the retry loop (pg_gemv_tnt.h).
    if (!dead && !arrived(pd.w, tag)) {
      long long spins = 0;
#pragma nounroll
      for (;;) {
        __builtin_amdgcn_s_sleep(1);
        pd.w = poll_word(i);
        if (arrived(pd.w, tag)) break;
        if (++spins > a.spin_limit) {
          dead = true;
          if (lane == 0) __hip_atomic_store(a.team_err, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      late_steps += 1;
      late_polls += (unsigned)(spins + 1);
    }
    if constexpr (DELAY) {
      const unsigned now = (unsigned)__builtin_amdgcn_s_memrealtime();
      const unsigned age = now - (unsigned)__builtin_amdgcn_readlane((int)(unsigned)pd.w, npoll);
      if (!dead) age_sum += age, age_cnt += 1;
    }
    // this lane's value (lanes that hold no member's value: zero), then the sum over the members
    T val;
    if constexpr (G == 1) {
      val = __builtin_bit_cast(float, (unsigned)pd.w);
    } else {
      const unsigned lo = (unsigned)pd.w;
      const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)lo, 0x101, 0xF, 0xF, true);  // the odd lane's half
      val = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    }
    val = adds ? val : T(0);
    if constexpr (PAIR) {  // the odd step's granules sit S lanes above the even step's
      const T up = row_from_above<S>(val);
      val = (i & 1) != 0 ? up : val;
    }
    T gs = val;
    row_member_sum<MS, 1>(gs, val);           // the members of this lane's row, ascending
    if (npoll > 16) gs = swap16_add(gs, gs);  // (wave-uniform) rows 0 + 1, 2 + 3
    if (npoll > 32) gs = swap32_add(gs, gs);  // (rows 0 + 1) + (rows 2 + 3): the same order on every device
    // lane c * G (c < C) of row 0 holds column c's total: hand it to every lane whose column of the chunk is column c of a step
    T g = pg_readlane(gs, 0);
#pragma unroll
    for (int c = 1; c < C; ++c) {
      const T tc = pg_readlane(gs, c * G);
      if (lane % C == c) g = tc;
    }
    const int k = chunk_pos(i), ncols = chunk_cols(i);
    const int64_t j = map.at(i - k) * C + min(lane, ncols - 1);
    const bool valid = lane < ncols && j < a.n;
    const bool mine_now = lane / C == k;  // this lane's column is one of THIS step's
    if (a.lam_ls != T(1)) g = a.lam_ls * g;
    // per-element parameters of g: loaded HERE, late and behind a (wave-uniform) branch -- a wave may have at most 63 vector-memory
    // operations in flight (the counter has six bits), and with three 16-load tiles under way every small load issued ahead with
    // the poll is one the next tile's loads wait for at ISSUE; the weighted / per-element-bound forms pay with a wait for
    // everything in flight instead (they are the rare case)
    if (has_pv) {
      const int64_t jq = j < a.n ? j : a.n - 1;
      pd.q0 = a.p0v[jq];
      if (a.p1v != nullptr) pd.q1 = a.p1v[jq];
    }
    const T xj = pd.xs, zo = pd.zos;
    const T yj = xj - a.gamma * g;  // forward_backward.jl:117 / fast_forward_backward.jl:140
    T zj;                            // :118 / :141
    if (a.g_kind == PG_G_NORML1) {
      const T th = has_pv ? pg_l1w_threshold(a.gamma, pd.q0) : a.p0;  // per-element weights lam_j
      zj = yj <= -th ? yj + th : (yj >= th ? yj - th : T(0));
    } else if (a.g_kind == PG_G_INDBOX) {
      const T lo = has_pv ? pd.q0 : a.p0, hi = has_pv ? pd.q1 : a.p1;  // per-element bounds
      zj = fmin(hi, fmax(lo, yj));
    } else
      zj = yj;
    const T rj = xj - zj;                                                   // :120 / :142
    const T vl = valid ? (a.v_is_res ? rj : zj + a.beta * (zj - zo)) : T(0);  // fast_forward_backward.jl:135 of the next iteration
    out_g = mine_now ? g : out_g, out_y = mine_now ? yj : out_y, out_z = mine_now ? zj : out_z, out_r = mine_now ? rj : out_r,
    out_v = mine_now ? vl : out_v;
#ifdef PG_TNT_EXPERIMENT
    if (!(a.dbg & 16))  // timing experiment: no output stores
#endif
    if (k == ncols / C - 1) {  // the chunk's last step: whole lines
#ifdef PG_TNT_EXPERIMENT
      if (valid && (a.dbg & 1024)) {  // timing experiment: the outputs as non-temporal stores
        __builtin_nontemporal_store(out_g, &a.g_out[j]);
        __builtin_nontemporal_store(out_y, &a.y[j]);
        __builtin_nontemporal_store(out_z, &a.z_new[j]);
        __builtin_nontemporal_store(out_r, &a.res[j]);
        if (a.v_out != nullptr) __builtin_nontemporal_store(out_v, &a.v_out[j]);
      } else
#endif
      if (valid) {
        a.g_out[j] = out_g;
        a.y[j] = out_y;
        a.z_new[j] = out_z;
        a.res[j] = out_r;
        if (a.v_out != nullptr) a.v_out[j] = out_v;
        if (a.g_kind == PG_G_NORML1) acc[0] += has_pv ? (double)pd.q0 * fabs((double)out_z) : fabs((double)out_z);
        acc[1] = pg_maxn(acc[1], fabs((double)out_r));
        acc[2] += (double)out_g * (double)out_r;
        acc[3] += (double)out_r * (double)out_r;
      }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) vj[c] = pg_readlane(vl, k * C + c);
  };
  auto park_slot = [&](int64_t i) { return park + (size_t)(LAG > 0 ? i % (LAG > 0 ? LAG : 1) : 0) * (C * U * WAVE) + lane; };

  // One step, as in gemv_tnt_kernel: [poll the totals of step i - LT, fetch its x_j / z_old_j] [start loading tile i + PF]
  // [dot + post tile i] [totals of step i - LT -> v_j ; A v accumulation from the parked tile] [park tile i - LAGR].
  auto step = [&](auto allc, Tile& cur, Tile& nxt, Tile& old, int64_t i) __attribute__((always_inline)) {
    constexpr bool ALL = decltype(allc)::value;
    if (!ALL && i >= cnt + LT) return;
    const bool has_fma = ALL || i >= LT;
    Pend pd{};
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (AHEAD) {
      pd = pn;  // what the previous step asked for
      if (ALL || (i + 1 >= LT && i + 1 < cnt + LT)) {  // the next step has totals to consume: ask now
        pn.w = poll_word(i + 1 - LT);
        fetch(pn, i + 1 - LT);
      }
    } else if (has_fma) {
      pd.w = poll_word(i - LT);  // issued BEFORE the next tile's loads: it returns first
      fetch(pd, i - LT);
    }
    // (the small loads first, pinned: loads return in issue order, and behind the tile's sixteen the epilogue would wait for them too)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LOADS_FIRST) {
      if (ALL || i + PF < cnt) load(nxt, i + PF);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ALL || i < cnt) dot_post(cur, i);
    if constexpr (!LOADS_FIRST) {
      // (experiment: the next tile's loads AFTER the dots, when tile i has retired from the 63 vector-memory operations a wave may
      // have in flight)
      __builtin_amdgcn_sched_barrier(0);
      if (ALL || i + PF < cnt) load(nxt, i + PF);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (has_fma) {
      T vj[C];
      totals(i - LT, pd, vj);
      if constexpr (LAG == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) racc[u][e] = fma(old.col[c][u][e], vj[c], racc[u][e]);
          }
        }
#ifdef PG_TNT_EXPERIMENT
      } else if (a.dbg & 2) {  // timing experiment: no LDS read-back (wrong tile)
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) racc[u][e] = fma(cur.col[c][u][e], vj[c], racc[u][e]);
          }
        }
#endif
      } else {
        // Four 16-byte reads at a time, their order pinned (pg_gemv_tnt.h: left alone the compiler issues all C * U reads up
        // front and spills the tile that is in flight).
        const V* __restrict__ src = park_slot(i - LT);
        int dep = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
          for (int u0 = 0; u0 < U; u0 += 4) {
            const V* __restrict__ sp = src + dep;
            V col[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (u0 + k < U) col[k] = sp[(c * U + u0 + k) * WAVE];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (u0 + k < U) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) racc[u0 + k][e] = fma(col[k][e], vj[c], racc[u0 + k][e]);
              }
            }
            asm volatile("v_mov_b32 %0, 0" : "=v"(dep) : "v"(racc[u0][0]));
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(racc[u]));
    if constexpr (LAG > 0) {
#ifdef PG_TNT_EXPERIMENT
      if (!(a.dbg & 4))  // timing experiment: no parking
#endif
      if (ALL || (i >= LAGR && i - LAGR < cnt)) {
        V* __restrict__ dst = park_slot(i - LAGR);
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
          for (int u = 0; u < U; ++u) dst[(c * U + u) * WAVE] = old.col[c][u];
        }
      }
    }
  };

  constexpr int NR = PF + 1 + LAGR;
  static_assert(NR >= 2 && NR <= 8, "two to eight register tiles");
  static_assert(PF == 1 || PF == 2, "one or two tiles in flight");
  Tile t0, t1, t2, t3, t4, t5, t6, t7;
  auto tile = [&](auto k) __attribute__((always_inline)) -> Tile& {
    constexpr int K = decltype(k)::value;
    if constexpr (K == 0) return t0;
    else if constexpr (K == 1) return t1;
    else if constexpr (K == 2) return t2;
    else if constexpr (K == 3) return t3;
    else if constexpr (K == 4) return t4;
    else if constexpr (K == 5) return t5;
    else if constexpr (K == 6) return t6;
    else return t7;
  };
  if (cnt > 0) load(t0, 0);
  if constexpr (PF > 1) {
    if (cnt > 1) load(t1, 1);
  }
  auto round = [&](auto allc, int64_t base) __attribute__((always_inline)) {
    auto one = [&](auto sc) __attribute__((always_inline)) {
      constexpr int SS = decltype(sc)::value;
      if constexpr (SS < NR)
        step(allc, tile(std::integral_constant<int, SS>{}), tile(std::integral_constant<int, (SS + PF) % NR>{}),
             tile(std::integral_constant<int, (SS + NR - LAGR) % NR>{}), base + SS);
    };
    one(std::integral_constant<int, 0>{});
    one(std::integral_constant<int, 1>{});
    one(std::integral_constant<int, 2>{});
    one(std::integral_constant<int, 3>{});
    one(std::integral_constant<int, 4>{});
    one(std::integral_constant<int, 5>{});
    one(std::integral_constant<int, 6>{});
    one(std::integral_constant<int, 7>{});
  };
  constexpr int64_t HEAD = (LT + NR - 1) / NR * NR;
  int64_t base = 0;
  for (; base < HEAD && base < cnt + LT; base += NR) round(std::false_type{}, base);
  if (base + NR + PF <= cnt) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the steady loop starts from a state the compiler knows exactly
    for (; base + NR + PF <= cnt; base += NR) round(std::true_type{}, base);
  }
  for (; base < cnt + LT; base += NR) round(std::false_type{}, base);
  // this device's rows of the workgroup's partial of A v
  T* part = a.partials + (int64_t)team * a.ld + lane * VEC;
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (u < a.nrg) *reinterpret_cast<V*>(part + (int64_t)u * (WAVE * VEC)) = racc[u];
  if (lane == 0 && a.wait_stats != nullptr && late_steps != 0) {
    atomicAdd(a.wait_stats, (unsigned long long)late_steps);
    atomicAdd(a.wait_stats + 1, (unsigned long long)late_polls);
  }
  if constexpr (DELAY) {
    if (lane == 0 && a.wait_stats != nullptr && age_cnt != 0) {
      atomicAdd(a.wait_stats + 2, age_sum);
      atomicAdd(a.wait_stats + 3, (unsigned long long)age_cnt);
    }
  }
  const double ps[4] = {a.gscale, 1.0, 1.0, 1.0};
  grid_reduce_finalize<4, 0x2u, 1>(acc, a.red_partials, a.red_counter, a.scal_out, ps);
}
