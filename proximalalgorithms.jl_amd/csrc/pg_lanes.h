// Lane exchanges of one wavefront without an LDS round trip (gfx950): v_permlane32_swap / v_permlane16_swap for the two
// cross-row steps, DPP moves inside a row.  Shared by the sweeps (pg_gemv_tn.h: wave_allsum; pg_gemv_tn2.hip, pg_gemv_tnp1.h:
// the column-parallel reduction).  Included inside namespace pgtn.
#pragma once
// ---------------------------------------------------------------------------------------------------------------
// column-parallel wave reduction: d[0..C) per lane in, the total of column (lane / (64 / C)) in every lane out
// ---------------------------------------------------------------------------------------------------------------
// v_permlane32_swap: vdst[32:63] <-> vsrc[0:31] ; v_permlane16_swap: odd rows of vdst <-> even rows of vsrc (gfx950).
// Inline assembly, not __builtin_amdgcn_permlane{32,16}_swap: with ROCm 7.2's compiler `r[0] + r[1]` on the builtin's
// result pair compiles to `v_add_f32 v1, v1, v1` (the second output is lost) -- seen in the disassembly and as wrong Float32
// results on the device.  The s_nop covers the VALU-write -> permlane-swap-read hazard the compiler would otherwise handle.
__device__ __forceinline__ void lane_swap32(unsigned& x, unsigned& y) {
  asm("s_nop 1\n\tv_permlane32_swap_b32_e32 %0, %1" : "+v"(x), "+v"(y));
}
__device__ __forceinline__ void lane_swap16(unsigned& x, unsigned& y) {
  asm("s_nop 1\n\tv_permlane16_swap_b32_e32 %0, %1" : "+v"(x), "+v"(y));
}
// lanes < 32: a(l) + a(l + 32) ; lanes >= 32: b(l - 32) + b(l)
__device__ __forceinline__ float swap32_add(float a, float b) {
  unsigned x = __builtin_bit_cast(unsigned, a), y = __builtin_bit_cast(unsigned, b);
  lane_swap32(x, y);
  return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}
// rows 0, 2: a(row) + a(row + 1) ; rows 1, 3: b(row - 1) + b(row)
__device__ __forceinline__ float swap16_add(float a, float b) {
  unsigned x = __builtin_bit_cast(unsigned, a), y = __builtin_bit_cast(unsigned, b);
  lane_swap16(x, y);
  return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ double swap32_add(double a, double b) {
  const unsigned long long ab = __builtin_bit_cast(unsigned long long, a), bb = __builtin_bit_cast(unsigned long long, b);
  unsigned xl = (unsigned)ab, yl = (unsigned)bb, xh = (unsigned)(ab >> 32), yh = (unsigned)(bb >> 32);
  lane_swap32(xl, yl);
  lane_swap32(xh, yh);
  return __builtin_bit_cast(double, ((unsigned long long)xh << 32) | xl) +
         __builtin_bit_cast(double, ((unsigned long long)yh << 32) | yl);
}
__device__ __forceinline__ double swap16_add(double a, double b) {
  const unsigned long long ab = __builtin_bit_cast(unsigned long long, a), bb = __builtin_bit_cast(unsigned long long, b);
  unsigned xl = (unsigned)ab, yl = (unsigned)bb, xh = (unsigned)(ab >> 32), yh = (unsigned)(bb >> 32);
  lane_swap16(xl, yl);
  lane_swap16(xh, yh);
  return __builtin_bit_cast(double, ((unsigned long long)xh << 32) | xl) +
         __builtin_bit_cast(double, ((unsigned long long)yh << 32) | yl);
}

// One butterfly stage.  Stage s pairs lanes that differ in bit 5 - s (and, for the DPP mirrors, in lower bits as well):
// a lane whose bit is 0 keeps `a` and receives its partner's `a`; a lane whose bit is 1 keeps `b` and receives `b`.
// With a == b this is one step of a plain all-reduce.  Fixed pairing, commutative adds: deterministic, and all lanes
// that end up with the same column hold the same bits.
template <int STAGE, typename T>
__device__ __forceinline__ T cr_combine(T a, T b, int lane) {
  if constexpr (STAGE == 0) {
    return swap32_add(a, b);
  } else if constexpr (STAGE == 1) {
    return swap16_add(a, b);
  } else {
    constexpr int BIT = 5 - STAGE;
    const bool upper = ((lane >> BIT) & 1) != 0;
    const T send = upper ? a : b, keep = upper ? b : a;
    if constexpr (STAGE == 2) return keep + pg_dpp_mov<0x140>(send);       // row_mirror: lane ^ 15
    else if constexpr (STAGE == 3) return keep + pg_dpp_mov<0x141>(send);  // row_half_mirror: lane ^ 7
    else if constexpr (STAGE == 4) return keep + pg_dpp_mov<0x4E>(send);   // quad_perm [2,3,0,1]: lane ^ 2
    else return keep + pg_dpp_mov<0xB1>(send);                             // quad_perm [1,0,3,2]: lane ^ 1
  }
}

template <typename T, int CNT, int STAGE>
__device__ __forceinline__ void cr_stage(T* d, int lane) {
  if constexpr (STAGE < 6) {
    if constexpr (CNT > 1) {
#pragma unroll
      for (int k = 0; k < CNT / 2; ++k) d[k] = cr_combine<STAGE>(d[k], d[k + CNT / 2], lane);
      cr_stage<T, CNT / 2, STAGE + 1>(d, lane);
    } else {
      d[0] = cr_combine<STAGE>(d[0], d[0], lane);
      cr_stage<T, 1, STAGE + 1>(d, lane);
    }
  }
}

