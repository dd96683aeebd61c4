// How fast can the HOST of a GPU box assemble MOI.ScalarQuadraticTerms (24 B: coeff, row, col; row-major upper triangle) from the CSC values of
// the same upper triangle (column-major, what pmt_quad_gram_csc_deliver_f64 delivers)?  If this is faster than shipping the 201 MB of terms
// over PCIe (3.7 ms at 54 GB/s), the reference's own boundary could take 84 MB instead of 252 MB per solve.
//   g++ -O3 -march=native -pthread tools/host_assemble_probe.cpp -o tools/host_assemble_probe
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include <immintrin.h>
struct QT { double c; int64_t r, k; };
static inline void store_nt(QT *p, double c, int64_t r, int64_t k) {
    _mm_stream_si64(reinterpret_cast<long long *>(p), *reinterpret_cast<long long *>(&c));
    _mm_stream_si64(reinterpret_cast<long long *>(p) + 1, r);
    _mm_stream_si64(reinterpret_cast<long long *>(p) + 2, k);
}
// columns [k0, k1): dst[pos(j,k)] for j <= k, pos = j*n - j(j-1)/2 + (k - j); blocked 8 rows x 64 columns
template <bool NT> static void assemble(QT *dst, const double *csc, const int64_t *var, int64_t n, int64_t k0, int64_t k1, int64_t j0, int64_t j1) {
    for (int64_t jb = j0; jb < j1; jb += 8) {
        for (int64_t kb = std::max(k0, jb); kb < k1; kb += 64) {
            const int64_t ke = std::min(k1, kb + 64);
            for (int64_t j = jb; j < std::min(j1, jb + 8); ++j) {
                QT *row = dst + (j * n - j * (j - 1) / 2 - j);
                const int64_t vj = var[j];
                for (int64_t k = std::max(kb, j); k < ke; ++k) {
                    const double c = csc[k * (k + 1) / 2 + j];
                    if (NT) store_nt(row + k, c, vj, var[k]); else { row[k].c = c; row[k].r = vj; row[k].k = var[k]; }
                }
            }
        }
    }
}
int main(int argc, char **argv) {
    const int64_t n = 4096, nq = n * (n + 1) / 2;
    std::vector<double> csc(nq);
    for (int64_t i = 0; i < nq; ++i) csc[i] = (double)i;
    std::vector<int64_t> var(n);
    for (int64_t i = 0; i < n; ++i) var[i] = i + 1;
    QT *dst = static_cast<QT *>(aligned_alloc(64, nq * sizeof(QT)));
    for (int64_t i = 0; i < nq; ++i) dst[i] = QT{0, 0, 0};
    for (int nt : {1, 4, 8, 16, 32, 64, 128}) {
        for (int mode = 0; mode < 2; ++mode) {
            double best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                auto t0 = std::chrono::steady_clock::now();
                std::vector<std::thread> pool;
                // rows dealt out so that every thread gets the same number of terms: row j has n - j terms
                std::vector<int64_t> cut(nt + 1, 0);
                for (int t = 1; t <= nt; ++t) {
                    const double target = (double)nq * t / nt;
                    int64_t lo = cut[t - 1], hi = n;
                    while (lo < hi) { int64_t mid = (lo + hi) / 2; double terms = (double)mid * n - (double)mid * (mid - 1) / 2; if (terms < target) lo = mid + 1; else hi = mid; }
                    cut[t] = std::min<int64_t>(n, (lo + 7) / 8 * 8);
                }
                cut[nt] = n;
                for (int t = 0; t < nt; ++t)
                    pool.emplace_back([&, t] { if (mode) assemble<true>(dst, csc.data(), var.data(), n, 0, n, cut[t], cut[t + 1]); else assemble<false>(dst, csc.data(), var.data(), n, 0, n, cut[t], cut[t + 1]); });
                for (auto &th : pool) th.join();
                best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            }
            // check
            int64_t bad = 0;
            for (int64_t j = 0; j < n; j += 97) for (int64_t k = j; k < n; k += 61) { const QT &t = dst[j * n - j * (j - 1) / 2 + (k - j)]; if (t.c != (double)(k * (k + 1) / 2 + j) || t.r != j + 1 || t.k != k + 1) ++bad; }
            printf("%3d threads  %-14s %.3f ms  (%.1f GB/s of terms written)%s\n", nt, mode ? "nontemporal" : "plain stores", best * 1e3, nq * 24.0 / best / 1e9, bad ? "  WRONG" : "");
        }
    }
    return 0;
}
