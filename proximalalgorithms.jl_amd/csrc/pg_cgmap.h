// CgMap: which column group a unit (workgroup / wave run / team) works on in its i-th step.  Kept in a header of its own,
// free of HIP types, so that the index arithmetic can be compiled and checked on the host (tests/test_cpu_host.py compiles
// tests/c_abi/cgmap_check.cpp against it with g++: every group visited exactly once, step counts balanced).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PG_HD __host__ __device__ __forceinline__
#else
#define PG_HD inline
#endif

namespace pgtn {

// Column groups -> units (workgroups; waves in gemv_tnw; teams in gemv_tnt).  A unit produces, per column, five 4-byte
// outputs (g, y, z, res, v).  With the groups dealt round-robin (unit u takes groups u, u + N, u + 2N ...) every 128-byte
// line of those vectors is written in 4..16 pieces by as many workgroups on different XCDs: none of their L2s ever holds
// the whole line, each piece goes to memory as a masked partial write, and the sweep loses 3-8 % of its streaming rate to
// them (scripts/tile_pattern.hip, profiles/r3_mid_columns_counters.md: the same kernel without the stores, or with the stores
// going to workgroup-private lines, runs 7.15-7.3 TB/s where it ran 6.7-6.9).  So a unit takes line_cols / C CONSECUTIVE
// groups -- line_cols = 32 columns: whole lines of every output, written by one workgroup within a few steps -- and then
// jumps by N such chunks; A is still swept as one moving window (N chunks wide).  The groups left over by the last
// incomplete round are dealt one by one again, so no unit gets more than one step more than another.
struct CgMap {
  // 32-bit on purpose (a matrix has < 2^31 column groups): as 64-bit fields these were 13 more scalar registers live through
  // the team kernel's steady loop plus 64-bit multiplies three times per step -- its scalar registers spilled to vector lanes
  // (v_writelane / v_readlane 253 -> 574) and the 50000-row and Float64 team sweeps lost 6-10 %
  int head, cnt;  // steps under the chunked assignment; all steps of this unit
  int gridK, tail0, unit, nunits;
  int shift;  // log2 of the chunk length (groups)
  PG_HD CgMap(int64_t ncg64, int C, int line_cols, int64_t unit_, int64_t nunits_) : unit((int)unit_), nunits((int)nunits_) {
    const int ncg = (int)ncg64;
    int K = line_cols / C;  // C and line_cols are powers of two
    if (K < 1) K = 1;
    shift = 31 - __builtin_clz((unsigned)K);
    // chunking needs units * K <= 2^30 (it is <= 4096 * 32 in every launcher); otherwise deal one by one
    if ((int64_t)nunits << shift > (int64_t(1) << 30)) shift = 0;
    gridK = nunits << shift;
    const int rounds = ncg / gridK;
    head = rounds << shift;
    tail0 = rounds * gridK;
    const int rem = ncg - tail0;
    cnt = head + (rem > unit ? (rem - unit + nunits - 1) / nunits : 0);
  }
  PG_HD int64_t at(int64_t i64) const {
    const int i = (int)i64;
    const int chunked = (i >> shift) * gridK + (unit << shift) + (i & ((1 << shift) - 1));
    const int tail = tail0 + unit + (i - head) * nunits;
    return (int64_t)(i < head ? chunked : tail);
  }
};

}  // namespace pgtn
