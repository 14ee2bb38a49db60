// Sanitizer harness of the host side of the pose-graph solver (ordering, symbolic analysis, worker pool, host twin of the numeric
// phase): three threads call dsss_host_pg_solve concurrently on lawn-mower-like graphs.  Built and run by tools/sanitize/run.sh with
// -fsanitize=thread and with -fsanitize=address,undefined (CPU only: the GPU box has no sanitizer support).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <thread>
#include <vector>
extern "C" int dsss_host_pg_solve(int ns, const int32_t* edge_a, const int32_t* edge_b, int nedges, const double* cx, const double* cy,
                                  const int32_t* part, int nparts, const double* aval36, const double* rhs6, double* x6_host, int64_t* stats8);
extern "C" int dsss_host_pg_solve_local(int ns, const int32_t* edge_a, const int32_t* edge_b, int nedges, const double* cx, const double* cy,
                                        const int32_t* iface_last, int nlast, const double* aval36, const double* rhs6, double* x6_host, int64_t* stats8);
extern "C" int dsss_host_pg_solve_parts(int ns, const int32_t* edge_a, const int32_t* edge_b, int nedges, const double* cx, const double* cy,
                                        int K, const double* aval36, const double* rhs6, double* x6_host, int64_t* stats8);
static void run(unsigned seed, int ns, int nlc, int reps)
{
    unsigned long long lcg = seed * 2654435761ull + 1; auto rnd = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (int)((lcg >> 33) & 0x7fffffff); };
    std::vector<int32_t> ea, eb; std::vector<double> cx(ns), cy(ns);
    for (int k = 0; k + 1 < ns; ++k) { ea.push_back(k); eb.push_back(k + 1); }
    const int leg = 200;
    for (int k = 0; k < ns; ++k) { const int l = k / leg, p = k % leg; cx[k] = (l & 1) ? leg - p : p; cy[k] = 3.0 * l; }
    for (int e = 0; e < nlc; ++e) {
        int a = rnd() % ns, l = a / leg, p = a % leg;
        int b = (l + 1) * leg + (leg - 1 - p) + (rnd() % 5 - 2);
        if (b <= a || b >= ns) { --e; if (l + 1 >= ns / leg) { ++e; } continue; }
        bool dup = false; for (size_t q = ns - 1; q < ea.size(); ++q) if (ea[q] == a && eb[q] == b) dup = true;
        if (dup || b == a + 1) continue;
        ea.push_back(a); eb.push_back(b);
    }
    const int ne = (int)ea.size();
    std::vector<double> aval((size_t)(ns + ne) * 36, 0.0), rhs((size_t)ns * 6), x((size_t)ns * 6);
    std::vector<double> diag(ns, 1.0);
    for (int e = 0; e < ne; ++e) { for (int i = 0; i < 6; ++i) aval[(size_t)(ns + e) * 36 + i * 6 + i] = -0.3; diag[ea[e]] += 0.5; diag[eb[e]] += 0.5; }
    for (int k = 0; k < ns; ++k) for (int i = 0; i < 6; ++i) aval[(size_t)k * 36 + i * 6 + i] = diag[k] + 0.1 * i;
    for (size_t i = 0; i < rhs.size(); ++i) rhs[i] = sin(0.1 * i);
    for (int r = 0; r < reps; ++r) {
        int64_t st[8];
        const int rc = dsss_host_pg_solve(ns, ea.data(), eb.data(), ne, cx.data(), cy.data(), nullptr, 1, aval.data(), rhs.data(), x.data(), st);
        if (rc) { printf("rc %d\n", rc); exit(1); }
    }
    printf("seed %u ok: x[0] %.6f\n", seed, x[0]);
    // the analysis one rank of several runs (round 6): every 37th node a prescribed interface node, eliminated last as one dense front
    std::vector<int32_t> last; for (int k = 5; k < ns; k += 37) last.push_back(k);
    std::vector<double> x2((size_t)ns * 6);
    int64_t st2[8];
    const int rc2 = dsss_host_pg_solve_local(ns, ea.data(), eb.data(), ne, cx.data(), cy.data(), last.data(), (int)last.size(), aval.data(), rhs.data(), x2.data(), st2);
    if (rc2) { printf("local rc %d\n", rc2); exit(1); }
    double dmax = 0; for (size_t i = 0; i < x.size(); ++i) dmax = fmax(dmax, fabs(x[i] - x2[i]));
    if (!(dmax < 1e-9)) { printf("prescribed interface: solutions differ by %g\n", dmax); exit(1); }
    printf("seed %u ok with a prescribed interface of %zu nodes (max difference %.1e)\n", seed, last.size(), dmax);
    // one rank analysed by parts (round 6): the parts run at the same time on the pool, then write their shares of the joined tables
    for (int K : { 3, 8 }) {
        std::vector<double> x3((size_t)ns * 6);
        int64_t st3[8];
        const int rc3 = dsss_host_pg_solve_parts(ns, ea.data(), eb.data(), ne, cx.data(), cy.data(), K, aval.data(), rhs.data(), x3.data(), st3);
        if (rc3) { printf("parts rc %d\n", rc3); exit(1); }
        double d3 = 0; for (size_t i = 0; i < x.size(); ++i) d3 = fmax(d3, fabs(x[i] - x3[i]));
        if (!(d3 < 1e-9)) { printf("analysis by %d parts: solutions differ by %g\n", K, d3); exit(1); }
    }
    printf("seed %u ok analysed by 3 and by 8 parts\n", seed);
}
// analysis only (x = NULL) of a graph beyond 65 536 separators: the passes of the ordering that run by ranges of a large node set (round 5:
// key copies, histograms, marks, counts; the difference array of the chain-order cut with relaxed atomic adds)
static void run_large(unsigned seed, int ns, int nlc)
{
    unsigned long long lcg = seed * 2654435761ull + 1; auto rnd = [&]() { lcg = lcg * 6364136223846793005ull + 1442695040888963407ull; return (int)((lcg >> 33) & 0x7fffffff); };
    std::vector<int32_t> ea, eb; std::vector<double> cx(ns), cy(ns);
    for (int k = 0; k + 1 < ns; ++k) { ea.push_back(k); eb.push_back(k + 1); }
    const int leg = 400;
    for (int k = 0; k < ns; ++k) { const int l = k / leg, p = k % leg; cx[k] = (l & 1) ? leg - p : p; cy[k] = 3.0 * l + 1e-3 * (rnd() % 100); }
    for (int e = 0; e < nlc; ++e) { const int a = rnd() % (ns - 2 * leg), b = a + leg + (leg - 2 * (a % leg)) - 1 + (rnd() % 3); if (b > a && b < ns) { ea.push_back(a); eb.push_back(b); } }
    int64_t st[8];
    setenv("DSSS_SYM_THREADS", "4", 1);
    const int rc = dsss_host_pg_solve(ns, ea.data(), eb.data(), (int)ea.size(), cx.data(), cy.data(), nullptr, 1, nullptr, nullptr, nullptr, st);
    unsetenv("DSSS_SYM_THREADS");
    if (rc) { printf("large analysis rc %d\n", rc); exit(1); }
    printf("large analysis ok: %d separators, %lld blocks of L, %lld levels\n", ns, (long long)st[0], (long long)st[3]);
}
int main()
{
    run_large(7u, 90000, 40000);
    std::thread a(run, 1u, 9000, 1200, 2), b(run, 2u, 2400, 300, 3);     // (9000 separators: large enough for the concurrently counted cut candidates of the ordering)
    run(3u, 1800, 250, 3);
    a.join(); b.join();
    return 0;
}
