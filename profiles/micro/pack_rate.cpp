#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include <cstdlib>
#include "host.hpp"
int main(int argc, char **argv) {
    const int nt = argc > 1 ? atoi(argv[1]) : 8;
    const size_t N = 3ull << 30;
    std::vector<unsigned char> src(N), dst(N / 4 + 64);
    const char L[4] = {'A','C','G','T'};
    unsigned x = 1;
    for (size_t i = 0; i < N; ++i) { x = x * 1664525u + 1013904223u; src[i] = L[x >> 30]; }
    for (int rep = 0; rep < 4; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        const size_t share = N / nt;
        for (int t = 0; t < nt; ++t) th.emplace_back([&, t] {
            ts::PackRuns R;
            for (size_t a = t * share; a < (t + 1) * share; a += 16384) ts::pack_bases(src.data() + a, 16384, dst.data() + (a >> 2), true, (uint32_t)a, R);
        });
        for (auto &t : th) t.join();
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%d threads: %.2f GB/s\n", nt, N / s / 1e9);
    }
}
