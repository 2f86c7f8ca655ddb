// ThreadSanitizer driver for the persistent host workers of a handle (navtex_amd/csrc/nvx_pool.h): runs of changing
// width, each index exactly once, results visible to the caller when run() returns, clean shutdown with parked workers.
// Built by tests/test_sanitizers.py with -fsanitize=thread.
#include "nvx_pool.h"

#include <atomic>
#include <cstdio>

int main()
{
    long long total = 0, want = 0;
    {
        HostPool pool;
        std::vector<int> hits(16);
        unsigned x = 7;
        for (int round = 0; round < 3000; round++) {
            x = x * 1664525u + 1013904223u;
            const int n = 1 + (int)((x >> 20) % 16);
            for (int t = 0; t < 16; t++) hits[t] = 0;
            std::vector<long long> part(n, 0);
            pool.run(n, [&](int t) { hits[t]++; for (int k = 0; k < 50; k++) part[t] += (long long)t * k; });   // plain writes: run() must order them
            for (int t = 0; t < 16; t++) if (hits[t] != (t < n ? 1 : 0)) { fprintf(stderr, "round %d: index %d ran %d times\n", round, t, hits[t]); return 2; }
            for (int t = 0; t < n; t++) { total += part[t]; want += (long long)t * (49 * 50 / 2); }
        }
        if ((int)pool.workers.size() != 15) return 3;          // started once, as many as the widest run needed
    }                                                          // ~HostPool joins the parked workers
    if (total != want) return 4;
    printf("tsan pool ok\n");
    return 0;
}
