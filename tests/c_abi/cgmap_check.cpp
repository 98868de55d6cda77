// Host-side check of CgMap (csrc/pg_cgmap.h), the column-group -> unit assignment of every sweep kernel: for a range of
// (column groups, columns per group, units, line length) every group must be visited exactly once, by a step index below
// the unit's count, the units' step counts must differ by at most one chunk, and with whole-line chunks a unit's chunked
// steps must cover runs of line_cols / C consecutive groups.  Built and run by tests/test_cpu_host.py (g++, no GPU).
#include <cstdio>
#include <vector>

#include "pg_cgmap.h"

int main() {
  long cases = 0;
  const int Cs[] = {1, 2, 4, 8, 16, 32};
  const int lines[] = {1, 32, 64};
  const long ncgs[] = {1, 2, 3, 7, 31, 32, 33, 255, 256, 257, 1000, 1024, 4095, 4096, 4097, 32768, 42000, 131072, 524288, 1000003};
  const int units[] = {1, 2, 3, 16, 32, 64, 255, 256, 1024, 4096};
  for (int C : Cs)
    for (int line : lines)
      for (long ncg : ncgs)
        for (int nu : units) {
          if (nu > ncg) continue;  // launchers clamp the grid to the number of groups
          std::vector<unsigned char> seen(ncg, 0);
          int K = line / C;
          if (K < 1) K = 1;
          int cmin = 1 << 30, cmax = 0;
          for (int u = 0; u < nu; ++u) {
            pgtn::CgMap map(ncg, C, line, u, nu);
            if (map.cnt < cmin) cmin = map.cnt;
            if (map.cnt > cmax) cmax = map.cnt;
            long prev = -1;
            for (long i = 0; i < map.cnt; ++i) {
              const long cg = map.at(i);
              if (cg < 0 || cg >= ncg) { printf("FAIL range: C=%d line=%d ncg=%ld units=%d unit=%d i=%ld cg=%ld\n", C, line, ncg, nu, u, i, cg); return 1; }
              if (seen[cg]++) { printf("FAIL twice: C=%d line=%d ncg=%ld units=%d unit=%d i=%ld cg=%ld\n", C, line, ncg, nu, u, i, cg); return 1; }
              if (i < map.head && (i % K) != 0 && cg != prev + 1) { printf("FAIL run: C=%d line=%d ncg=%ld units=%d unit=%d i=%ld\n", C, line, ncg, nu, u, i); return 1; }
              prev = cg;
            }
          }
          for (long g = 0; g < ncg; ++g)
            if (!seen[g]) { printf("FAIL missed: C=%d line=%d ncg=%ld units=%d cg=%ld\n", C, line, ncg, nu, g); return 1; }
          if (cmax - cmin > 1) { printf("FAIL balance: C=%d line=%d ncg=%ld units=%d counts %d..%d\n", C, line, ncg, nu, cmin, cmax); return 1; }
          ++cases;
        }
  printf("CGMAP_OK %ld cases\n", cases);
  return 0;
}
