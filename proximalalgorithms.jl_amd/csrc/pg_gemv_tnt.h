// The team sweep (gemv_tnt_kernel): columns whose rows are split over several workgroups -- of one device (pg_gemv_tn2.hip,
// LONG columns) or, PEER, of several devices, one row block each (pg_gemv_tn4.hip, north_star's row layout at one read of A
// per iteration).  Included inside `namespace pgtn { namespace {` by both files; wave_allsum / CgMap / TNArgs come from
// pg_gemv_tn.h.
#pragma once
// ---------------------------------------------------------------------------------------------------------------
// LONG columns: teams of workgroups
// ---------------------------------------------------------------------------------------------------------------
constexpr int TEAM_MEMBER_RG = 64;              // a member holds WAVES * U = 64 row groups of every column (16384 rows in Float32)
constexpr int TEAM_MAX = 16;                    // members per team: up to 1024 row groups (262144 rows f32, 131072 f64)
constexpr int TEAM_RING = 16;                   // granule ring slots per team (>= 2 (LAG + LAGR) + 2, see the protocol note)
constexpr int PEER_RING = 16;                   // ... per workgroup index of a PEER team (members on different devices: LAG up to 7)
constexpr long long TEAM_SPIN_LIMIT = 1 << 21;  // polls before a member gives up (~ seconds): bounded, never a hang

template <typename T, int C>
struct Pending {
  unsigned long long w;  // this lane's granule of the awaited step
  T xs[C], zos[C];       // the columns' x_j and z_old_j, fetched together with the poll
};

// Protocol.  Step i of a team = column group map.at(i) (CgMap, pg_gemv_tn.h: chunks of whole output lines per team).  Member p writes its C partial dots of step i as
// granules {value bits (32 per granule; an f64 takes two), tag = i + 1} into ring slot i % TEAM_RING at offset
// (p * C + c) * G + h with ONE 8-byte agent-scope store each, so a reader that sees the tag sees the value.  Every wave
// of every member polls the team_size * C * G granules of a step (one lane per granule), then sums the values in member
// order.  A member posts step i only after it has consumed step i - LAG - 1, i.e. after ALL members have posted step
// i - LAG - 1, which each of them did after consuming step i - 2 LAG - 2: a ring of 2 LAG + 2 slots is never
// overwritten before everyone has read it.  Tags carry the launch epoch in their high byte (tag = epoch << 24 | i + 1, i + 1
// < 2^24), so the ring needs no zeroing between launches: every slot is rewritten by every launch (the step count of a
// matrix is fixed), and a granule left by the launch 256 epochs ago has been overwritten 255 times since.
//
// Where the column tiles live.  PF + 1 register tiles rotate (one being dotted, PF being loaded), as in
// gemv_tn_kernel.  A tile whose totals are still travelling (LAG > 0) is parked in LDS -- LAG slots of
// WAVES * C * U KiB, 128 KiB of the CU's 160 KiB at the default geometry -- and read back, 16 bytes per lane at a time,
// for the A v accumulation: every wave touches only its own region of a slot (no barrier), reads the slot of step
// i - LAG and then parks step i in the same slot.  (Keeping the waiting tiles in registers was tried first: the kernel
// then needs ~340 VGPRs and spills whole tiles to scratch.)
//
// What the loop body may not contain (each cost a factor on the device, all seen in the disassembly): a branch around any
// of the streaming loads, taken or not -- the compiler then no longer counts the loads in flight and waits for ALL of them
// (s_waitcnt vmcnt(0)) before every dot product; a load inside the retry loop of the poll without a first look outside
// it -- same effect at the loop header; the small loads (poll, x_j, z_old_j) issued AFTER the tile loads -- they return
// in order, so reading them waits for the tile too.  Hence: a branch-free steady-state loop (ALL = true) between a
// conditional head and tail, row groups past the end of a column clamped (masked through r = 0) instead of skipped.
//
// PEER teams (pg_gemv_tn4.hip).  The members of a team are the workgroups with the SAME blockIdx on peer_n DEVICES: device p
// holds the row block A_p (all of it: at most WAVES * U row groups) and its slice of r; everything else is as above, with
// three differences.  (1) Every device has its own ring (its inbox, a.xch) and polls only that; a member posts its granules
// into the inbox of EVERY member (a.peer_ring[q], system-scope stores over the fabric: pushes, never remote reads).
// (2) Every member writes the column's outputs -- the n-vectors are replicated, and all members form the same bits from the
// same granules in the same order.  (3) The ring is PEER_RING slots deep so that LAG can cover the fabric's latency.
//
// Lag tiles in REGISTERS (LAGR, round 5).  The slack the exchange has is parked bytes per compute unit / streaming rate, and LDS
// holds 128 KiB of them; the register file holds 512 KiB per compute unit.  A tile therefore waits its first LAGR steps where it
// was loaded (PF + 1 + LAGR named register tiles rotate instead of PF + 1; the compiler places what does not fit the 256
// architectural VGPRs of a one-wave-per-SIMD geometry in accumulation registers by itself -- on gfx950 they are one 512-entry
// file and a load may target either half), is parked in LDS for the remaining LAG steps, and is read back for the A v
// accumulation LAG + LAGR steps after its dots were posted.  LAG = 0 with LAGR > 0: no LDS at all, the tile is multiplied where
// it sits.
//
// DELAY (tests only, PEER): a latency injector for the hand-off.  Next to its granules a member posts one more, carrying the
// constant 100 MHz clock (s_memrealtime) at the time of the store; a consumer accepts the granules of a step only once every
// member's stamp is a.delay_ticks old.  All members must share a clock: ranks of ONE device -- which is where the injector is
// used (profiles/r5_row_team_latency_sweep.md): the on-chip hand-off stands in for a fabric hop of a chosen length.  The hop is
// thus max(on-chip hand-off, a.delay_ticks), not their sum, and nobody is held up who would not have been by a real hop of
// that length: the sender does not wait, the stamp travels with the data.
//
// Tried on top and not kept (profiles/r5_team_options_not_kept.md): polling a step ahead (-17 %), a barrier-free dot exchange
// through a tagged LDS ring with a rotating poster (+0.4 %).
//
// AHEAD (round 6, PEER): the poll of a step's totals and the fetch of its x_j / z_old_j are issued ONE STEP before their use.  Vector
// loads return in issue order, so a poll issued at the start of the step that consumes it sits behind the loads of the tile the NEXT
// step will dot: reading it waits for that tile as well, and only one of the two tiles "in flight" ever is.  Issued a step earlier it
// returns with a tile that is waited for anyway.  Price: the first look comes one step sooner after the post (LT - 1 steps of slack).
// (Round 5 tried this at LT = 2 -- one step of slack left, -17 % -- and dropped it; at LT = 4 it is what keeps two tiles in flight.)
template <typename T, int U, int C, int WAVES, int LAG, int PF = 1, bool PEER = false, int LAGR = 0, bool DELAY = false, bool AHEAD = false>
__global__ __launch_bounds__(WAVES * 64) void gemv_tnt_kernel(TNArgs<T> a) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::N;
  constexpr int G = (int)sizeof(T) / 4;  // granules per value
  constexpr int RING = PEER ? PEER_RING : TEAM_RING;
  constexpr int SCOPE = PEER ? __HIP_MEMORY_SCOPE_SYSTEM : __HIP_MEMORY_SCOPE_AGENT;
  constexpr int LT = LAG + LAGR;  // steps between a tile's dots and its multiply-adds
  static_assert(2 * LT + 2 <= RING, "granule ring too short for this lag");
  static_assert(PEER || TEAM_MAX * C * G <= 64, "one lane per granule of a step");  // (PEER: peer_n * C * G <= 64, checked at launch)
  static_assert(!DELAY || PEER, "the latency injector belongs to the row-team sweep");
  static_assert(!AHEAD || (PEER && LT > 0), "the poll one step ahead: row-team sweeps with a lag");
  __shared__ T sm_dot[2][C][WAVES];
  extern __shared__ __attribute__((aligned(16))) unsigned char park_raw[];
  V* const park = reinterpret_cast<V*>(park_raw);  // [LAG][WAVES][C][U][64]
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int team = PEER ? (int)blockIdx.x : (int)(blockIdx.x % (unsigned)a.nteams);
  const int member = PEER ? a.peer_rank : (int)(blockIdx.x / (unsigned)a.nteams);
  const int TM = PEER ? a.peer_n : a.team_size;
  const int npoll = TM * C * G;
  const int64_t ncg = (a.n + C - 1) / C;
  const CgMap map(ncg, C, a.line_cols, team, a.nteams);  // step i of this team = column group map.at(i)
  const int64_t cnt = map.cnt;
  // This wave's contiguous run of row groups [rg0, rg0 + nu) of every column, nu <= U.  deal_even: the nrg row groups are dealt
  // over the team's TMe * WAVES waves as evenly as whole row groups allow -- base = nrg / waves each, one more for the first
  // nrg % waves waves counted wave-index-major (wave 0 of every member, then wave 1 of every member, ...), so that the extra
  // row groups spread over the MEMBERS (a team is as fast as its fullest member: 196 row groups over 4 x 4 waves are 49 per
  // member -- 13 + 12 + 12 + 12 -- where ceil(196 / 16) = 13 for everyone made three members of 52 and one of 40).
  const int TMe = PEER ? 1 : TM, memb = PEER ? 0 : member;
  int nu = a.ueff, rg0 = (memb * WAVES + wave) * a.ueff;
  if (a.deal_even) {
    const int base = a.nrg / (TMe * WAVES), extra = a.nrg % (TMe * WAVES);
    nu = base + ((wave * TMe + memb) < extra ? 1 : 0);
    int before = 0;  // waves ahead of this one (member-major row order) that hold base + 1
    for (int mm = 0; mm < memb; ++mm)
      for (int w = 0; w < WAVES; ++w) before += (w * TMe + mm) < extra ? 1 : 0;
    for (int w = 0; w < wave; ++w) before += (w * TMe + memb) < extra ? 1 : 0;
    rg0 = (memb * WAVES + wave) * base + before;
  }
  const size_t ring_off = (size_t)team * RING * (size_t)(TEAM_MAX * C * G);
  unsigned long long* const ring = a.xch + ring_off;

  V rk[U], racc[U];
  int rgc[U];  // row groups past the end of the column are clamped to the last one; their r is zero
#pragma unroll
  for (int u = 0; u < U; ++u) {
    rgc[u] = max(0, min(rg0 + min(u, nu - 1), a.nrg - 1));  // past the run: the wave's own last row group again (a cached load)
#pragma unroll
    for (int e = 0; e < VEC; ++e) racc[u][e] = T(0);
    if (u < nu && rg0 + u < a.nrg) {
      rk[u] = *reinterpret_cast<const V*>(a.r + (int64_t)(rg0 + u) * (WAVE * VEC) + lane * VEC);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) rk[u][e] = T(0);
    }
  }
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  bool dead = false;  // wave-uniform: a poll timed out, stop waiting (the launch is reported as failed)
  unsigned late_steps = 0, late_polls = 0;  // PEER telemetry: steps whose granules were not there at the first look, polls spent
  unsigned long long age_sum = 0;            // DELAY telemetry: clock ticks between the stamp of a step's granules (member 0's) and their use (the slack that was left)
  unsigned age_cnt = 0;

  struct Tile {
    V col[C][U];
  };
  auto load = [&](Tile& t, int64_t i) __attribute__((always_inline)) {
    const int64_t j0 = map.at(i) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t j = (j0 + c < a.n) ? (j0 + c) : (a.n - 1);
      // wave-uniform base (scalar registers) + one shared per-lane offset: no 64-bit vector address per load
      const T* __restrict__ p = a.A + j * a.ld;
#pragma unroll
      for (int u = 0; u < U; ++u)
        t.col[c][u] = nt_load(reinterpret_cast<const V*>(p + (int64_t)rgc[u] * (WAVE * VEC)) + lane);
    }
  };
  // this member's partial dots of step i -> ring
  auto dot_post = [&](const Tile& t, int64_t i) __attribute__((always_inline)) {
    const int buf = (int)(i & 1);
#pragma unroll
    for (int c = 0; c < C; ++c) {
      T d = T(0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) d = fma(t.col[c][u][e], rk[u][e], d);
      }
      d = wave_allsum(d);
      if (lane == 0) sm_dot[buf][c][wave] = d;
    }
    if constexpr (WAVES > 1) __syncthreads();  // (a one-wave workgroup reads back what it wrote itself: program order)
    if (wave == 0 && lane < C * G + (DELAY ? 1 : 0)) {
      T mine = T(0);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        T s = sm_dot[buf][c][0];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) s += sm_dot[buf][c][w];
        if (lane / G == c) mine = s;
      }
      unsigned bits;
      if constexpr (G == 1) {
        bits = __builtin_bit_cast(unsigned, mine);
      } else {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, mine);
        bits = (lane % G) == 0 ? (unsigned)b : (unsigned)(b >> 32);
      }
      size_t off = (size_t)(i % RING) * (TEAM_MAX * C * G) + (size_t)member * (C * G) + lane;
      if constexpr (DELAY) {
        if (lane == C * G) {  // the stamp granule: behind the TM members' value granules, one per member
          bits = (unsigned)__builtin_amdgcn_s_memrealtime();
          off = (size_t)(i % RING) * (TEAM_MAX * C * G) + (size_t)TM * (C * G) + (size_t)member;
        }
      }
      const unsigned long long word = ((unsigned long long)(a.tag_base + (unsigned)(i + 1)) << 32) | bits;
      if constexpr (PEER) {
        for (int q = 0; q < TM; ++q)  // one 8-byte store into every member's inbox (its own included)
          __hip_atomic_store(a.peer_ring[q] + ring_off + off, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        __hip_atomic_store(ring + off, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  // every lane polls (no exec-masked load): lanes >= npoll repeat the last granule (DELAY: the TM stamp granules follow the values)
  const int npoll_all = DELAY ? npoll + TM : npoll;
  const int poll_lane = lane < npoll_all ? lane : npoll_all - 1;
  // all granules of the step carry its tag (DELAY: and every member's stamp is a.delay_ticks old)
  auto arrived = [&](unsigned long long w, unsigned tag) __attribute__((always_inline)) -> bool {
    bool ok = (unsigned)(w >> 32) == tag;
    if constexpr (DELAY) {
      const unsigned now = (unsigned)__builtin_amdgcn_s_memrealtime();
      if (lane >= npoll && lane < npoll_all) ok = ok && (now - (unsigned)w) >= a.delay_ticks;
    }
    return __builtin_amdgcn_ballot_w64(ok) == ~0ull;
  };
  auto poll_word = [&](int64_t i) __attribute__((always_inline)) -> unsigned long long {
    return __hip_atomic_load(ring + (size_t)(i % RING) * (TEAM_MAX * C * G) + poll_lane, __ATOMIC_RELAXED, SCOPE);
  };
  auto fetch_xz = [&](Pending<T, C>& pd, int64_t i) __attribute__((always_inline)) {
    const int64_t j0 = map.at(i) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const int64_t jc = (j0 + c < a.n) ? (j0 + c) : (a.n - 1);
      pd.xs[c] = a.x[jc];
      pd.zos[c] = a.z_old[jc];
    }
  };
  // totals of step i (all members have posted, or will shortly) -> epilogue -> v_j (0 for columns past the end)
  auto totals = [&](int64_t i, Pending<T, C>& pd, T (&vj)[C]) __attribute__((always_inline)) {
    const unsigned tag = a.tag_base + (unsigned)(i + 1);
#ifdef PG_TNT_EXPERIMENT
    if (a.dbg & 1) dead = true;  // timing experiment: never wait (totals are then wrong)
#endif
    // The first look at the granules stays OUTSIDE the retry loop (see the note above the kernel).
    if (!dead && !arrived(pd.w, tag)) {
      long long spins = 0;
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
      if constexpr (PEER) {  // telemetry (kept in registers, ONE atomic pair per wave at the end of the kernel)
        late_steps += 1;
        late_polls += (unsigned)(spins + 1);
      }
    }
    if constexpr (DELAY) {  // how old the step's youngest granule is when it is consumed: the slack the pipeline had left
      const unsigned now = (unsigned)__builtin_amdgcn_s_memrealtime();
      const unsigned age = now - (unsigned)__builtin_amdgcn_readlane((int)(unsigned)pd.w, npoll);  // (member 0's stamp; wave-uniform: scalar registers)
      if (!dead) age_sum += age, age_cnt += 1;
    }
    const int w_lo = (int)(unsigned)pd.w;
    const int64_t j0 = map.at(i) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      T g = T(0);
      for (int p = 0; p < TM; ++p) {  // member order: every member forms the same bits
        const int idx = (p * C + c) * G;
        if constexpr (G == 1) {
          g += __builtin_bit_cast(float, __builtin_amdgcn_readlane(w_lo, idx));
        } else {
          const unsigned lo = (unsigned)__builtin_amdgcn_readlane(w_lo, idx);
          const unsigned hi = (unsigned)__builtin_amdgcn_readlane(w_lo, idx + 1);
          g += __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        }
      }
      const int64_t j = j0 + c;
      const bool valid = j < a.n;
      if (a.lam_ls != T(1)) g = a.lam_ls * g;
      const T xj = pd.xs[c], zo = pd.zos[c];
      const T yj = xj - a.gamma * g;  // forward_backward.jl:117 / fast_forward_backward.jl:140
      T zj;                            // :118 / :141
      if (a.g_kind == PG_G_NORML1) {
        T th = a.p0;
        if (a.p0v != nullptr) th = pg_l1w_threshold(a.gamma, a.p0v[valid ? j : a.n - 1]);  // per-element weights lam_j
        zj = yj <= -th ? yj + th : (yj >= th ? yj - th : T(0));
      } else if (a.g_kind == PG_G_INDBOX) {
        T lo = a.p0, hi = a.p1;
        if (a.p0v != nullptr) lo = a.p0v[valid ? j : a.n - 1], hi = a.p1v[valid ? j : a.n - 1];  // per-element bounds
        zj = fmin(hi, fmax(lo, yj));
      } else
        zj = yj;
      const T rj = xj - zj;                             // :120 / :142
      vj[c] = valid ? (a.v_is_res ? rj : zj + a.beta * (zj - zo)) : T(0);   // fast_forward_backward.jl:135 of the next iteration
      if ((PEER || member == 0) && (int)threadIdx.x == c && valid) {
        a.g_out[j] = g;
        a.y[j] = yj;
        a.z_new[j] = zj;
        a.res[j] = rj;
        if (a.v_out != nullptr) a.v_out[j] = vj[c];
        if (a.g_kind == PG_G_NORML1) acc[0] += a.p0v != nullptr ? (double)a.p0v[j] * fabs((double)zj) : fabs((double)zj);
        acc[1] = pg_maxn(acc[1], fabs((double)rj));
        acc[2] += (double)g * (double)rj;
        acc[3] += (double)rj * (double)rj;
      }
    }
  };
  auto park_slot = [&](int64_t i) { return park + ((size_t)(LAG > 0 ? i % (LAG > 0 ? LAG : 1) : 0) * WAVES + wave) * (C * U * WAVE) + lane; };
  Pending<T, C> pn{};  // AHEAD: the poll and the small loads of the NEXT step's totals, in flight across a step

  // One step: [poll the totals of step i - LT, fetch its x_j / z_old_j] [start loading tile i + PF into `nxt`]
  // [dot + post tile i = `cur`] [totals of step i - LT -> v_j ; A v accumulation from the parked tile] [park tile i - LAGR = `old`
  // (LAGR = 0: `cur` itself)].  ALL = steady state: every part runs, no branch.
  auto step = [&](auto allc, Tile& cur, Tile& nxt, Tile& old, int64_t i) __attribute__((always_inline)) {
    constexpr bool ALL = decltype(allc)::value;
    if (!ALL && i >= cnt + LT) return;
    const bool has_fma = ALL || i >= LT;
    Pending<T, C> pd{};
    // scheduling fences around the load issue: `nxt` is the register tile the previous step read last; without them the
    // scheduler hoists these loads above that step's multiply-adds into fresh registers and the kernel spills
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (AHEAD) {
      pd = pn;  // what the previous step asked for
      if (ALL || (i + 1 >= LT && i + 1 < cnt + LT)) {  // the next step has totals to consume: ask now
        pn.w = poll_word(i + 1 - LT);
        fetch_xz(pn, i + 1 - LT);
      }
      __builtin_amdgcn_sched_barrier(0);  // (the small loads first, pinned: loads return in issue order)
    } else if (has_fma) {
      if constexpr (LT > 0) pd.w = poll_word(i - LT);  // issued BEFORE the next tile's loads: it returns first
      fetch_xz(pd, i - LT);
    }
    if (ALL || i + PF < cnt) load(nxt, i + PF);
    __builtin_amdgcn_sched_barrier(0);
    if (ALL || i < cnt) dot_post(cur, i);
    if constexpr (LT == 0) {
      if (has_fma) pd.w = poll_word(i);
    }
    if (has_fma) {
      T vj[C];
      totals(i - LT, pd, vj);
      if constexpr (LAG == 0) {  // the tile of step i - LT is still in registers (`old`; LT = 0: `cur`)
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
        // Four 16-byte reads at a time.  Left alone the compiler issues all C * U reads up front (C * U * 4 registers) and,
        // for two columns per step, makes room by spilling the tile that is in flight.  The address of every chunk depends
        // (through an opaque v_mov that always yields 0) on an accumulator of the previous chunk, which pins the order
        // read 4 -> multiply-add 4 -> read 4 ...
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
    // Pin the accumulators HERE.  Their only use is at the end of the kernel, so the compiler is free to sink this step's
    // multiply-adds into the next step (keeping this step's tile alive across it: +C * U * 4 registers, whole tiles spilled).
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(racc[u]));
    if constexpr (LAG > 0) {
#ifdef PG_TNT_EXPERIMENT
      if (!(a.dbg & 4))
#endif
      if (ALL || (i >= LAGR && i - LAGR < cnt)) {  // same slot as the tile just read ((i - LAGR) % LAG == (i - LT) % LAG): this wave's region only, program order suffices
        V* __restrict__ dst = park_slot(i - LAGR);
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
          for (int u = 0; u < U; ++u) dst[(c * U + u) * WAVE] = old.col[c][u];
        }
      }
    }
  };

  // PF + 1 + LAGR register tiles rotate: tile i (in t[i % NR]) is dotted while tiles i + 1 .. i + PF are in flight and tiles
  // i - LAGR .. i - 1 wait for their totals; the load of tile i + PF goes where tile i - LAGR - 1 sat, which the previous step
  // parked (or multiplied).  Named variables, not an array: an array of tiles handed to the step by reference ends up in
  // scratch memory.
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
      constexpr int S = decltype(sc)::value;
      if constexpr (S < NR)
        step(allc, tile(std::integral_constant<int, S>{}), tile(std::integral_constant<int, (S + PF) % NR>{}),
             tile(std::integral_constant<int, (S + NR - LAGR) % NR>{}), base + S);
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
  constexpr int64_t HEAD = (LT + NR - 1) / NR * NR;  // first round boundary from which every step has totals to consume
  int64_t base = 0;
  for (; base < HEAD && base < cnt + LT; base += NR) round(std::false_type{}, base);
  if (base + NR + PF <= cnt) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the steady loop starts from a state the compiler knows exactly
    for (; base + NR + PF <= cnt; base += NR) round(std::true_type{}, base);
  }
  for (; base < cnt + LT; base += NR) round(std::false_type{}, base);
  // this member's rows of the team's partial of A v
  T* part = a.partials + (int64_t)team * a.ld + lane * VEC;
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (u < nu && rg0 + u < a.nrg) *reinterpret_cast<V*>(part + (int64_t)(rg0 + u) * (WAVE * VEC)) = racc[u];
  if constexpr (PEER) {
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
  }
  const double ps[4] = {a.gscale, 1.0, 1.0, 1.0};
  grid_reduce_finalize<4, 0x2u, WAVES>(acc, a.red_partials, a.red_counter, a.scal_out, ps);
}

