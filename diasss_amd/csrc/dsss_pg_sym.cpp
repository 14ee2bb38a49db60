// diasss_amd/csrc/dsss_pg_sym.cpp -- ordering, symbolic factorisation, bins, fronts and schedule of the reduced pose-graph
// system (see dsss_pg_sym.h), plus a host twin of the numeric phase for the CPU test-suite.  Plain C++, no HIP.
#include "dsss_pg_sym.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <thread>

namespace {

// A small pool of worker threads that lives as long as the process: the analysis is ~7 ms of host work per solve made of a
// dozen short parallel phases, and a fresh std::thread per phase (50-100 us each to create, plus scheduling jitter) cost as
// much as it saved.  Workers spin for a short while after a task before they go to sleep on the condition variable, so the
// phases of one analysis find them awake.  fork() runs a task on a worker (or inline, if no worker took it by the time the
// caller waits: the waiter steals it back), so recursion can fork at every level without ever blocking on a busy pool.
class pg_pool {
public:
    struct task { std::function<void()> fn; std::atomic<int> st{ 0 }; };      // 0 queued, 1 running, 2 done
    static pg_pool& get() { static pg_pool P; return P; }
    void fork(task* t)
    {
        if (workers.empty()) { t->st.store(1); t->fn(); t->st.store(2, std::memory_order_release); return; }
        { std::lock_guard<std::mutex> lk(mu); q.push_back(t); npending.fetch_add(1, std::memory_order_release); }
        if (nsleep.load(std::memory_order_acquire) > 0) cv.notify_one();
    }
    void join(task* t)
    {
        if (t->st.load(std::memory_order_acquire) == 0) {        // not started yet: take it out of the queue and run it here
            bool mine = false;
            { std::lock_guard<std::mutex> lk(mu);
              auto it = std::find(q.begin(), q.end(), t);
              if (it != q.end()) { q.erase(it); npending.fetch_sub(1, std::memory_order_release); mine = true; } }
            if (mine) { t->st.store(1); t->fn(); t->st.store(2, std::memory_order_release); return; }
        }
        for (int spins = 0; t->st.load(std::memory_order_acquire) != 2; ++spins) { if (spins > 2000) std::this_thread::yield(); }
    }
    ~pg_pool()
    {
        { std::lock_guard<std::mutex> lk(mu); stop.store(true); }
        cv.notify_all();
        for (auto& w : workers) w.join();
    }
private:
    pg_pool()
    {
        const unsigned hc = std::thread::hardware_concurrency();
        int n = hc > 1 ? (int)std::min(23u, hc - 1) : 0;      // (the analysis of a C3-size graph forks eight ways; graphs of several hundred thousand separators use them all: dsss_pg.hip, sym_threads)
        for (int i = 0; i < n; ++i) workers.emplace_back([this] { run(); });
    }
    void run()
    {
        for (;;) {
            task* t = nullptr;
            const auto t0 = std::chrono::steady_clock::now();
            for (int spins = 0; !t; ++spins) {
                if (stop.load(std::memory_order_acquire)) return;
                if (npending.load(std::memory_order_acquire) > 0) {
                    std::lock_guard<std::mutex> lk(mu);
                    if (!q.empty()) { t = q.front(); q.pop_front(); npending.fetch_sub(1, std::memory_order_release); t->st.store(1, std::memory_order_release); }
                } else if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) {
                    std::unique_lock<std::mutex> lk(mu);
                    nsleep.fetch_add(1, std::memory_order_release);
                    cv.wait(lk, [&] { return stop.load() || !q.empty(); });
                    nsleep.fetch_sub(1, std::memory_order_release);
                    if (stop.load()) return;
                    t = q.front(); q.pop_front(); npending.fetch_sub(1, std::memory_order_release); t->st.store(1, std::memory_order_release);
                }
            }
            t->fn();
            t->st.store(2, std::memory_order_release);
        }
    }
    std::mutex mu; std::condition_variable cv; std::deque<task*> q;
    std::atomic<int> npending{ 0 }, nsleep{ 0 }; std::atomic<bool> stop{ false };
    std::vector<std::thread> workers;
};

template <class F> void par_ranges(int n, int T, F fn)          // fn(t, lo, hi) over T contiguous ranges of [0, n)
{
    T = std::max(1, std::min(T, std::max(n, 1)));
    pg_pool& P = pg_pool::get();
    std::vector<pg_pool::task> tk(T > 1 ? T - 1 : 0);
    for (int t = 1; t < T; ++t) { tk[t - 1].fn = [&fn, n, T, t] { fn(t, (int)((long long)n * t / T), (int)((long long)n * (t + 1) / T)); }; P.fork(&tk[t - 1]); }
    fn(0, 0, (int)((long long)n / T));
    for (auto& x : tk) P.join(&x);
}

} // namespace

void dsss_pool_run(int T, const std::function<void(int)>& fn)
{
    T = std::max(1, T);
    pg_pool& P = pg_pool::get();
    std::vector<pg_pool::task> tk(T - 1);
    for (int t = 1; t < T; ++t) { tk[t - 1].fn = [&fn, t] { fn(t); }; P.fork(&tk[t - 1]); }
    fn(0);
    for (auto& x : tk) P.join(&x);
}

namespace {

// nested dissection with vertex separators taken from the lower half.  The order of a subtree is [A][B][separator]; A and B
// never touch, so the first PG_ND_PAR levels run their two halves on two host threads and the same tree of ranges later
// drives the parallel column-structure pass.  While a node set spans several ranks the cut is the rank boundary (lower
// ranks = A), so every rank's interior is one contiguous range of the order and the rank-level separators come last.
struct nd_tree { int a = -1, b = -1, size = 0; };          // children (indices into the node pool) or -1,-1 for a leaf
#define PG_ND_PAR 4
struct nd_ctx {
    const int* adj_ptr; const int* adj_idx; const double* cx; const double* cy; char* side; char* side2; int leaf; int both_axes; bool geo_first;
    const int* part; char* iface;                          // rank of every node (or null); iface[v] = 1 for rank-level separator nodes
    const char* forced;                                    // nodes with a neighbour of a HIGHER rank: they must end up in the interface (see nd_order)
    std::vector<nd_tree>* pool; std::mutex* mu;
    std::atomic<long long>* tns;                           // phase timers (nd_timer) or null
    bool index_cuts;                                       // a third candidate per set: the cheapest cut of the CHAIN ORDER near the median (see nd_order)
    int* pos;                                              // scratch of that candidate: rank of a node inside the set that holds it (-1: in a separator already)
    int threads;                                           // ranges of the passes over a LARGE node set (the top of the recursion is the serial part of the ordering)
};
// DSSS_PG_VERBOSE: thread-time per phase of nd_order over all calls of ONE analysis (1 candidates, 2 final boundary, 4 leaves); the counters
// belong to the analysis that asked for them (several solves may run at once)
struct nd_timer { std::atomic<long long>* a; std::chrono::steady_clock::time_point t0;
                  nd_timer(std::atomic<long long>* tns, int k) : a(tns ? tns + k : nullptr) { if (a) t0 = std::chrono::steady_clock::now(); }
                  ~nd_timer() { if (a) *a += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); } };
int nd_order(std::vector<int>& nodes, const nd_ctx& C, std::vector<int>& order, int depth)
{
    auto new_node = [&](int a, int b, int size) { std::lock_guard<std::mutex> g(*C.mu); C.pool->push_back({ a, b, size }); return (int)C.pool->size() - 1; };
    const int total = (int)nodes.size();
    // ranks the node set spans: while it spans several, every separator cut out of it is INTERFACE (replicated on all ranks), and the
    // recursion must go on until the pieces belong to one rank each
    int pmin = 0, pmax = 0;
    if (C.part) { pmin = 1 << 30; pmax = -1; for (int v : nodes) { pmin = std::min(pmin, C.part[v]); pmax = std::max(pmax, C.part[v]); } }
    const bool multi = pmax > pmin;
    if (!multi && total <= C.leaf) { nd_timer tm(C.tns, 4); std::sort(nodes.begin(), nodes.end()); for (int v : nodes) order.push_back(v); return depth <= PG_ND_PAR ? new_node(-1, -1, total) : -1; }
    std::vector<int> A, B, S;
    // The separator of a split (sides are marked in C.side): a lower-half node with a neighbour in the upper half -- except that of a
    // cut edge between nodes of DIFFERENT ranks it is always the lower-rank end that goes into the separator.  The numeric phase relies
    // on that: a factor belongs to the rank of its higher pose and adds to the diagonal block of the lower one, which therefore has to
    // be an interface separator (summed over the ranks).  Rank cuts satisfy it by themselves (lower ranks are the lower half).
    // b: when given, receives the upper half without its separator nodes.
    // (sd: the scratch marks of the split -- C.side, or C.side2 for the second candidate, which is evaluated on another thread)
    auto boundary = [&](const std::vector<int>& nd, size_t h, std::vector<int>* a, std::vector<int>* s, std::vector<int>* b = nullptr, char* sd = nullptr) {
        if (!sd) sd = C.side;
        for (size_t i = 0; i < nd.size(); ++i) sd[nd[i]] = i < h ? 1 : 2;
        size_t cnt = 0;
        for (size_t i = 0; i < h; ++i) {
            const int v = nd[i];
            bool cut = false;
            for (int q = C.adj_ptr[v]; q < C.adj_ptr[v + 1]; ++q) {
                const int u = C.adj_idx[q];
                if (sd[u] != 2 && sd[u] != 4) continue;
                if (multi && C.part[u] < C.part[v]) { if (sd[u] == 2) { sd[u] = 4; ++cnt; } }      // the upper-half end has the lower rank
                else cut = true;
            }
            if (cut) { ++cnt; if (s) s->push_back(v); } else if (a) a->push_back(v);
        }
        if (s || b) for (size_t i = h; i < nd.size(); ++i) { const int u = nd[i]; if (sd[u] == 4) { if (s) s->push_back(u); } else if (b) b->push_back(u); }
        return cnt;
    };
    // candidates: the rank cut (lower ranks first) while the set spans several ranks; the median cut along the longer extent, and --
    // for sets of at least both_axes nodes -- along the other axis too.  A survey is a long strip of parallel legs: the cut across the
    // longer extent is not the cheaper one once a piece holds few legs (a cut between legs costs the loop closures of two legs, a cut
    // across them one pose per leg), and rank cuts are cuts between legs.  The smallest separator wins; ties go to the rank cut, then
    // to the longer extent.  (With geo_first off a multi-rank set always takes its rank cut: every rank's interior is then ONE range
    // of the order, the layout of rounds 1-2.)
    size_t half = 0, best = (size_t)-1;
    std::vector<int> cand;
    if (multi) {
        const int pmid = (pmin + pmax + 1) / 2;
        half = std::stable_partition(nodes.begin(), nodes.end(), [&](int v) { return C.part[v] < pmid; }) - nodes.begin();
        best = boundary(nodes, half, nullptr, nullptr);
    }
    const bool geo = !multi || (C.geo_first && total > C.leaf);
    if (geo) {
        nd_timer tm(C.tns, 1);
        double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300; int i0 = 1 << 30, i1 = -1;
        for (int v : nodes) { x0 = std::min(x0, C.cx[v]); x1 = std::max(x1, C.cx[v]); y0 = std::min(y0, C.cy[v]); y1 = std::max(y1, C.cy[v]); i0 = std::min(i0, v); i1 = std::max(i1, v); }
        const bool byx = (x1 - x0) >= (y1 - y0);
        const size_t h2 = nodes.size() / 2;
        // split at the median of the (coordinate, index) total order; only the two halves matter, not their inner order
        // (round 4: the selection runs on a contiguous array of (coordinate, index) pairs -- through the index array every comparison was two
        // dependent loads into the coordinate table, and the top of the recursion, where one or two threads hold most of the nodes, is the
        // serial part of the ordering.  The two halves are determined by the total order alone: same sets, same elimination order.)
        auto split = [&](std::vector<int>& nd, bool bx) {
            const double* key = bx ? C.cx : C.cy;
            if (nd.size() < 96) {
                std::nth_element(nd.begin(), nd.begin() + h2, nd.end(), [&](int a, int b) {
                    const double ka = key[a], kb = key[b];
                    return ka != kb ? ka < kb : a < b; });
                return;
            }
            std::vector<std::pair<double, int>> kv(nd.size());
            for (size_t i = 0; i < nd.size(); ++i) kv[i] = { key[nd[i]], nd[i] };
            std::nth_element(kv.begin(), kv.begin() + h2, kv.end());       // pair order = (coordinate, index): the comparator above
            for (size_t i = 0; i < nd.size(); ++i) nd[i] = kv[i].second;
        };
        // A CANDIDATE is only counted, not carried out: the element of rank h2 of the (coordinate, index) order is found through a
        // histogram of the coordinates (one pass over a contiguous copy of the keys, then a selection inside ONE bucket), the halves are
        // marked by comparing with it, and the separator is counted from the marks -- three light passes instead of a selection that moves
        // 16-byte pairs about.  Only the winner is partitioned.  Same halves (they are determined by the total order), same counts.
        // axis 0 = y, 1 = x: median cuts of the dead-reckoned coordinates.  2 = a cut of the CHAIN ORDER (round 5, chain_cut below): a
        // survey is a lawn-mower chain, consecutive separators lie on one leg, so a cut of the chain order runs BETWEEN two legs (or through
        // one leg at ONE chain edge) and costs the loop closures that cross it.  The median of y does the same only on paper -- the median
        // node sits in the middle of some leg whose dead-reckoned y wanders by centimetres, and the cut zig-zags through that leg's chain
        // edges -- and, worse, it is taken wherever the median happens to fall.
        struct cut_cand { int bx; double pk; int pi; size_t cnt; size_t half; };
        std::vector<double> keys;                                          // (one candidate at a time per call; the second one of a large set brings its own)
        auto less_than = [](double k, int v, double pk, int pi) { return k != pk ? k < pk : v < pi; };
        auto count_cut = [&](const std::vector<int>& nd, int bx, double lo, double hi, std::vector<double>& kx, char* sd) {
            const double* key = bx == 1 ? C.cx : C.cy;
            const size_t m = nd.size();
            kx.resize(m);
            // (sets of 64 k nodes and more -- the top of the recursion at C5's 635 k separators, where one or two threads hold everything --
            // run their passes over ranges of the set; same marks, same counts)
            const int TP = (m >= 65536 && !multi) ? std::max(1, C.threads) : 1;
            par_ranges((int)m, TP, [&](int, int a0, int a1) { for (int i = a0; i < a1; ++i) kx[i] = key[nd[i]]; });
            cut_cand cc{ bx, 0.0, 0, 0, h2 };
            constexpr int NB = 1024;
            if (hi > lo && std::isfinite(hi - lo)) {
                const double scale = NB / (hi - lo);
                auto bucket = [&](double k) { const int b = (int)((k - lo) * scale); return b < 0 ? 0 : (b >= NB ? NB - 1 : b); };      // monotone in k
                unsigned hist[NB] = { 0 };
                if (TP > 1) {
                    std::vector<unsigned> ph((size_t)TP * NB, 0u);
                    par_ranges((int)m, TP, [&](int t, int a0, int a1) { unsigned* h = ph.data() + (size_t)t * NB; for (int i = a0; i < a1; ++i) h[bucket(kx[i])]++; });
                    for (int t = 0; t < TP; ++t) for (int q = 0; q < NB; ++q) hist[q] += ph[(size_t)t * NB + q];
                } else for (size_t i = 0; i < m; ++i) hist[bucket(kx[i])]++;
                size_t below = 0; int b = 0;
                while (below + hist[b] <= h2) below += hist[b++];              // the bucket that holds rank h2 (h2 < m)
                std::vector<std::pair<double, int>> in;
                in.reserve(hist[b]);
                for (size_t i = 0; i < m; ++i) if (bucket(kx[i]) == b) in.push_back({ kx[i], nd[i] });
                std::nth_element(in.begin(), in.begin() + (h2 - below), in.end());
                cc.pk = in[h2 - below].first; cc.pi = in[h2 - below].second;
            } else {                                                          // all coordinates equal (or not finite): the order is the index order
                std::vector<std::pair<double, int>> in(m);
                for (size_t i = 0; i < m; ++i) in[i] = { kx[i], nd[i] };
                std::nth_element(in.begin(), in.begin() + h2, in.end());
                cc.pk = in[h2].first; cc.pi = in[h2].second;
            }
            par_ranges((int)m, TP, [&](int, int a0, int a1) { for (int i = a0; i < a1; ++i) sd[nd[i]] = less_than(kx[i], nd[i], cc.pk, cc.pi) ? 1 : 2; });
            size_t cnt = 0;
            if (TP > 1) {                                                     // (one partition: the marks are only read here)
                std::vector<size_t> pc(TP, 0);
                par_ranges((int)m, TP, [&](int t, int a0, int a1) {
                    size_t c2 = 0;
                    for (int i = a0; i < a1; ++i) {
                        const int v = nd[i];
                        if (sd[v] != 1) continue;
                        for (int q = C.adj_ptr[v]; q < C.adj_ptr[v + 1]; ++q) if (sd[C.adj_idx[q]] == 2) { ++c2; break; }
                    }
                    pc[t] = c2;
                });
                for (size_t c2 : pc) cnt += c2;
            } else
            for (size_t i = 0; i < m; ++i) {
                const int v = nd[i];
                if (sd[v] != 1) continue;
                bool cut = false;
                for (int q = C.adj_ptr[v]; q < C.adj_ptr[v + 1]; ++q) {
                    const int u = C.adj_idx[q];
                    if (sd[u] != 2 && sd[u] != 4) continue;
                    if (multi && C.part[u] < C.part[v]) { if (sd[u] == 2) { sd[u] = 4; ++cnt; } }
                    else cut = true;
                }
                if (cut) ++cnt;
            }
            par_ranges((int)m, TP, [&](int, int a0, int a1) { for (int i = a0; i < a1; ++i) sd[nd[i]] = 0; });
            cc.cnt = cnt;
            return cc;
        };
        // The chain-order candidate.  Node sets are kept in ascending index order (every split is a stable partition), so a cut of the
        // chain order is a position p of the list: nodes [0, p) below, [p, m) above.  A node of rank r whose highest neighbour inside the
        // set has rank R > r is a separator node of every cut r < p <= R: one difference array prices ALL positions in one pass over the
        // edges, and the cheapest position of the balance window [m / 3, 2 m / 3] is the candidate (the one nearest the middle among equals).
        // On the C5 graph the cost of a cut between two legs varies from 23 to 740 separator nodes with the pair of legs it runs between
        // (few loop closures where two legs barely overlap); the median position is rarely a cheap one.  With this candidate the C3 graph
        // factorises in 12 panel levels instead of 29 (0.6 instead of 2.1 GFLOP, largest front 49 instead of 140 block rows), the C5 graph
        // in 37 instead of 169 (30 instead of 457 GFLOP, 172 instead of 1 184 block rows).
        auto chain_cut = [&](const std::vector<int>& nd) {
            const size_t m = nd.size();
            cut_cand cc{ 2, 0.0, 0, (size_t)-1, h2 };
            const int TP = m >= 65536 ? std::max(1, C.threads) : 1;
            par_ranges((int)m, TP, [&](int, int a0, int a1) { for (int r = a0; r < a1; ++r) C.pos[nd[r]] = r; });
            std::vector<int> diff(m + 2, 0);
            par_ranges((int)m, TP, [&](int, int a0, int a1) {
                for (int r = a0; r < a1; ++r) {
                    const int v = nd[r];
                    int R = -1;
                    for (int q = C.adj_ptr[v]; q < C.adj_ptr[v + 1]; ++q) { const int pu = C.pos[C.adj_idx[q]]; if (pu > R) R = pu; }      // (neighbours outside the set sit in separators: -1)
                    if (R > r) {
                        if (TP > 1) { __atomic_fetch_add(&diff[(size_t)r + 1], 1, __ATOMIC_RELAXED); __atomic_fetch_sub(&diff[(size_t)R + 1], 1, __ATOMIC_RELAXED); }
                        else { diff[(size_t)r + 1]++; diff[(size_t)R + 1]--; }
                    }
                }
            });
            const size_t lo = std::max<size_t>(1, m / 3), hi = std::min(m - 1, m - m / 3);      // (windows of +-2 / 5 / 10 / 17 / 25 % of the set: C3 17 / 14 / 12 / 12 / 13 panel levels, C5 59 / 41 / 47 / 37 / 35)
            long long run = 0;
            for (size_t p2 = 1; p2 <= hi; ++p2) {
                run += diff[p2];
                if (p2 < lo) continue;
                const size_t c = (size_t)run, dist = p2 > h2 ? p2 - h2 : h2 - p2, bdist = cc.half > h2 ? cc.half - h2 : h2 - cc.half;
                if (c < cc.cnt || (c == cc.cnt && dist < bdist)) { cc.cnt = c; cc.half = p2; }
            }
            cc.pi = nd[cc.half];
            return cc;
        };
        auto carry_out = [&](std::vector<int>& nd, const cut_cand& cc) {      // lower half first, both halves in ascending index order
            if (cc.bx == 2) return;                                           // (a position of the sorted list: nothing moves)
            const double* key = cc.bx == 1 ? C.cx : C.cy;
            std::stable_partition(nd.begin(), nd.end(), [&](int v) { return less_than(key[v], v, cc.pk, cc.pi); });
        };
        const double lo_of[3] = { y0, x0, (double)i0 }, hi_of[3] = { y1, x1, (double)i1 };           // [bx]
        const bool idx_cand = C.index_cuts && total >= C.both_axes && total >= 96;      // (only from 512 / 2 048 / 4 096 / 16 384 nodes on: C3 13 / 13 / 14 / 35 panel levels instead of 12, C5 35 / 36 / 37 / 44 instead of 35 -- the big sets matter most, the small ones still count)
        // The chain-order candidate goes first, and a very cheap one (at most 1/1024 of the set) ends the search: at the top of the recursion --
        // the serial part of the ordering -- it costs one pass over the set's edges and wins anyway (C3: 1 to 7 separator nodes where the
        // coordinate medians cost 26 to 207), the coordinate candidates are a histogram selection and a counting pass each.
        cut_cand c3{}; bool chain_done = false;
        if (idx_cand) {
            c3 = chain_cut(nodes);
            if (c3.cnt < best && c3.cnt * 1024 <= (size_t)total) { best = c3.cnt; half = c3.half; chain_done = true; }      // (1/64 took cuts of 700 nodes where the cut across 79 k nodes of C5 costs 140: twice the flops)
        }
        if (chain_done) {}
        else if (!multi && total < C.both_axes) { split(nodes, byx); half = h2; }       // the only candidate: nothing to compare
        else if (nodes.size() < 96) {
            for (int pass = 0; pass < (total >= C.both_axes ? 2 : 1); ++pass) {
                cand = nodes;
                split(cand, pass == 0 ? byx : !byx);
                const size_t c = boundary(cand, h2, nullptr, nullptr);
                if (c < best) { best = c; nodes.swap(cand); half = h2; }
            }
        }
        else if (total >= C.both_axes && total >= 4096 && depth <= PG_ND_PAR) {
            // both axes, large set: the two candidates are counted at the same time (the top of the recursion is the serial part of the
            // ordering); the marks of a node set are private to the call that holds it, the second candidate marks in side2
            cut_cand c2{};
            std::vector<double> keys2;
            pg_pool::task tk; tk.fn = [&] { c2 = count_cut(nodes, (int)!byx, lo_of[!byx], hi_of[!byx], keys2, C.side2); };
            pg_pool::get().fork(&tk);
            const cut_cand c1 = count_cut(nodes, (int)byx, lo_of[byx], hi_of[byx], keys, C.side);
            pg_pool::get().join(&tk);
            const cut_cand* win = nullptr;
            if (c1.cnt < best) { best = c1.cnt; win = &c1; }
            if (c2.cnt < best) { best = c2.cnt; win = &c2; }
            if (idx_cand && c3.cnt < best) { best = c3.cnt; win = &c3; }
            if (win) { carry_out(nodes, *win); half = win->half; }
        }
        else {
            cut_cand wc{}; bool have = false;
            for (int pass = 0; pass < (total >= C.both_axes ? 2 : 1); ++pass) {
                const int bx = pass == 0 ? (int)byx : (int)!byx;
                const cut_cand c = count_cut(nodes, bx, lo_of[bx], hi_of[bx], keys, C.side);
                if (c.cnt < best) { best = c.cnt; wc = c; have = true; }
            }
            if (idx_cand && c3.cnt < best) { best = c3.cnt; wc = c3; have = true; }
            if (have) { carry_out(nodes, wc); half = wc.half; }
        }
    }
    { nd_timer tm(C.tns, 2);
    boundary(nodes, half, &A, &S, &B);
    for (int v : nodes) C.side[v] = 0; }
    if (multi) {
        // A factor belongs to the rank of its higher pose and adds to the diagonal block of the lower one too, so a node with a
        // neighbour of a higher rank has to be INTERFACE (its entries are summed over the ranks) whether or not a cut ever runs between
        // the two -- the neighbour may have left into a separator higher up.  Such nodes are pulled into this separator when the half
        // they sit in has become one rank's (a half that still spans ranks passes them on): the lowest interface level there is.
        for (std::vector<int>* X : { &A, &B }) {
            int lo = 1 << 30, hi = -1;
            for (int v : *X) { lo = std::min(lo, C.part[v]); hi = std::max(hi, C.part[v]); }
            if (hi > lo) continue;
            size_t w = 0;
            for (size_t i = 0; i < X->size(); ++i) { const int v = (*X)[i]; if (C.forced[v]) S.push_back(v); else (*X)[w++] = v; }
            X->resize(w);
        }
    }
    std::sort(S.begin(), S.end());
    if (C.index_cuts) for (int v : S) C.pos[v] = -1;
    if (multi) for (int v : S) C.iface[v] = 1;
    if (!multi && (A.empty() || B.empty())) {            // degenerate cut: fall back to index order
        std::sort(nodes.begin(), nodes.end());
        for (int v : nodes) order.push_back(v);
        return depth <= PG_ND_PAR ? new_node(-1, -1, total) : -1;
    }
    int na = -1, nb = -1;
    if (depth < PG_ND_PAR && (total > 2048 || multi)) {
        std::vector<int> oa;
        pg_pool::task tk; tk.fn = [&] { na = nd_order(A, C, oa, depth + 1); };
        pg_pool::get().fork(&tk);
        std::vector<int> ob;
        nb = nd_order(B, C, ob, depth + 1);
        pg_pool::get().join(&tk);
        order.insert(order.end(), oa.begin(), oa.end());
        order.insert(order.end(), ob.begin(), ob.end());
    } else {
        nd_order(A, C, order, PG_ND_PAR + 1);
        nd_order(B, C, order, PG_ND_PAR + 1);
    }
    for (int v : S) order.push_back(v);
    return depth <= PG_ND_PAR ? new_node(na, nb, total) : -1;
}

// column structures of the range [lo, lo + size) of the elimination order described by tree node `t`, children merged
// into parents (elimination tree built on the fly).  A column whose parent lies outside the range hands the
// (parent, column) pair up to its caller.  No per-column allocations: the row lists of one call go into that call's pool
// (cref = pool, offset, length) and the children of a column are a linked list (kid_head / kid_next).
struct cref { int pool, off, n; };
struct cs_ctx {
    const int* adj_ptr; const int* adj_idx; const int* order; const int* perm; const std::vector<nd_tree>* pool;
    std::vector<std::vector<int>>* pools; cref* cols; int* kid_head; int* kid_next; int* parent;
};
void col_structs(const cs_ctx& C, int t, int lo, int size, std::vector<std::pair<int, int>>& up, int depth)
{
    const nd_tree nd = t >= 0 ? (*C.pool)[t] : nd_tree();
    int seq_lo = lo;
    const int hi = lo + size;
    auto add_kid = [&](int par, int j) { C.kid_next[j] = C.kid_head[par]; C.kid_head[par] = j; };
    if (t >= 0 && nd.a >= 0 && nd.b >= 0) {
        const int sa = (*C.pool)[nd.a].size, sb = (*C.pool)[nd.b].size;
        std::vector<std::pair<int, int>> ua, ub;
        pg_pool::task tk; tk.fn = [&] { col_structs(C, nd.a, lo, sa, ua, depth + 1); };
        pg_pool::get().fork(&tk);
        col_structs(C, nd.b, lo + sa, sb, ub, depth + 1);
        pg_pool::get().join(&tk);
        for (auto* u : { &ua, &ub })
            for (auto& e : *u) { if (e.first < hi) add_kid(e.first, e.second); else up.push_back(e); }
        seq_lo = lo + sa + sb;
    }
    const int my_pool = t >= 0 ? t : (int)C.pools->size() - 1;
    std::vector<int>& P = (*C.pools)[my_pool];
    P.reserve((size_t)(hi - seq_lo) * 24);
    std::vector<int> c;
    for (int j = seq_lo; j < hi; ++j) {
        c.clear();
        const int v = C.order[j];
        for (int q = C.adj_ptr[v]; q < C.adj_ptr[v + 1]; ++q) { const int pu = C.perm[C.adj_idx[q]]; if (pu > j) c.push_back(pu); }
        for (int k = C.kid_head[j]; k >= 0; k = C.kid_next[k]) {
            const cref ck = C.cols[k]; const int* d = (*C.pools)[ck.pool].data() + ck.off;
            for (int q = 1; q < ck.n; ++q) if (d[q] != j) c.push_back(d[q]);
        }
        std::sort(c.begin(), c.end()); c.erase(std::unique(c.begin(), c.end()), c.end());
        C.cols[j] = { my_pool, (int)P.size(), (int)c.size() + 1 };
        P.push_back(j); P.insert(P.end(), c.begin(), c.end());
        if (!c.empty()) {
            C.parent[j] = c[0];
            if (c[0] < hi) add_kid(c[0], j); else up.push_back({ c[0], j });
        }
    }
}

} // namespace


namespace {
// panel levels of the fronts (children have smaller indices than their parents)
void sym_levels(pg_sym& S)
{
    const int nf = (int)S.f_c0.size();
    // schedule: a front starts one level after its last child front has finished; panel steps are consecutive levels
    S.f_level0.assign(nf, 0); S.f_npan.resize(nf); S.f_pan0.resize(nf);
    int maxl = -1, np = 0;
    for (int f = 0; f < nf; ++f) {
        S.f_npan[f] = (S.f_s[f] + PG_PW - 1) / PG_PW; S.f_pan0[f] = np; np += S.f_npan[f];
        int l0 = 0;
        for (int c = S.ch_ptr[f]; c < S.ch_ptr[f + 1]; ++c) if (!S.ch_kind[c]) { const int g = S.ch_id[c]; l0 = std::max(l0, S.f_level0[g] + S.f_npan[g]); }
        S.f_level0[f] = l0; maxl = std::max(maxl, l0 + S.f_npan[f] - 1);
    }
    S.npanels = np; S.nlev = maxl + 1;
    S.lv_ptr.assign(S.nlev + 1, 0); S.asm_ptr.assign(S.nlev + 1, 0);
    for (int f = 0; f < nf; ++f) { S.asm_ptr[S.f_level0[f] + 1]++; for (int k = 0; k < S.f_npan[f]; ++k) S.lv_ptr[S.f_level0[f] + k + 1]++; }
    for (int l = 0; l < S.nlev; ++l) { S.lv_ptr[l + 1] += S.lv_ptr[l]; S.asm_ptr[l + 1] += S.asm_ptr[l]; }
    S.lv_front.resize(np); S.lv_step.resize(np); S.asm_front.resize(nf);
    {
        std::vector<int> fp(S.lv_ptr.begin(), S.lv_ptr.end() - 1), fq(S.asm_ptr.begin(), S.asm_ptr.end() - 1);
        for (int f = 0; f < nf; ++f) {
            S.asm_front[fq[S.f_level0[f]]++] = f;
            for (int k = 0; k < S.f_npan[f]; ++k) { const int at = fp[S.f_level0[f] + k]++; S.lv_front[at] = f; S.lv_step[at] = k; }
        }
    }
}

// per-row views of the fronts for the assembly kernel
void sym_row_views(pg_sym& S, int T)
{
    const int nf = (int)S.f_c0.size();
    // ---- per-row views for the assembly, work items of the assembly and of the trailing update
    {
        const int nrows_all = S.f_rowptr[nf];
        S.fa_rowptr.assign(nrows_all + 1, 0);
        S.xr_ptr.assign(nrows_all + 1, 0);
        // (a front's block rows are its own range of both views: counts and fills run by ranges of fronts, the two prefix sums between them
        // are one pass each)
        par_ranges(nf, T, [&](int, int lo, int hi) {
            for (int f = lo; f < hi; ++f) {
                for (int e = S.fa_ptr[f]; e < S.fa_ptr[f + 1]; ++e) S.fa_rowptr[S.f_rowptr[f] + S.fa_row[e] + 1]++;
                for (int c = S.ch_ptr[f]; c < S.ch_ptr[f + 1]; ++c) { const long long r0 = S.ch_relptr[c], r1 = S.ch_relptr[c + 1]; for (long long q = r0; q < r1; ++q) S.xr_ptr[S.f_rowptr[f] + S.rel[q] + 1]++; }
            }
        });
        // entries are sorted by (front, row): the CSR offsets are global positions in fa_*
        for (int i = 0; i < nrows_all; ++i) { S.fa_rowptr[i + 1] += S.fa_rowptr[i]; S.xr_ptr[i + 1] += S.xr_ptr[i]; }
        S.xr_child.resize(S.xr_ptr[nrows_all]); S.xr_row.resize(S.xr_ptr[nrows_all]);
        par_ranges(nf, T, [&](int, int lo, int hi) {
            std::vector<int> fp;
            for (int f = lo; f < hi; ++f) {
                const int r0f = S.f_rowptr[f], nr = S.f_n[f];
                fp.assign(S.xr_ptr.begin() + r0f, S.xr_ptr.begin() + r0f + nr);
                for (int c = S.ch_ptr[f]; c < S.ch_ptr[f + 1]; ++c) {                                   // children in their fixed order
                    const long long r0 = S.ch_relptr[c], r1 = S.ch_relptr[c + 1];
                    for (long long q = r0; q < r1; ++q) { const int at = fp[S.rel[q]]++; S.xr_child[at] = c; S.xr_row[at] = (int)(q - r0); }
                }
            }
        });
    }
}
} // namespace

void pg_symbolic(int ns, const std::vector<std::pair<int, int>>& edges, int nchain, const double* cx, const double* cy,
                 const int* part, int nparts, const pg_sym_opts& opt, pg_sym& S)
{
    S = pg_sym();
    S.ns = ns; S.nparts = std::max(1, nparts);
    const bool tv = getenv("DSSS_PG_VERBOSE") != nullptr && !opt.to_be_joined;      // (the parts of pg_symbolic_parts report together)
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto tms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto q0 = tnow();
    const int T = std::max(1, opt.threads);
    // adjacency in CSR form, rows sorted and deduplicated
    std::vector<int> adj_ptr(ns + 1, 0), adj_idx;
    {   // rows are tiny (two chain neighbours + the loop closures of the pose): insertion sort and duplicate removal in place, rows
        // compacted behind each other (std::sort + std::unique + a vector insert per row cost a millisecond at 23 k rows)
        for (auto& e : edges) if (e.first != e.second) { adj_ptr[e.first + 1]++; adj_ptr[e.second + 1]++; }
        for (int i = 0; i < ns; ++i) adj_ptr[i + 1] += adj_ptr[i];
        adj_idx.resize(adj_ptr[ns]);
        std::vector<int> fill(adj_ptr.begin(), adj_ptr.end() - 1);
        for (auto& e : edges) if (e.first != e.second) { adj_idx[fill[e.first]++] = e.second; adj_idx[fill[e.second]++] = e.first; }
        int w = 0;
        for (int i = 0; i < ns; ++i) {
            const int b0 = adj_ptr[i], e0 = adj_ptr[i + 1];
            adj_ptr[i] = w;
            for (int q = b0; q < e0; ++q) {                          // insert adj_idx[q] into the sorted, duplicate-free run [adj_ptr[i], w)
                const int v = adj_idx[q];
                int pos = w;
                while (pos > adj_ptr[i] && adj_idx[pos - 1] > v) --pos;
                if (pos > adj_ptr[i] && adj_idx[pos - 1] == v) continue;
                for (int k = w; k > pos; --k) adj_idx[k] = adj_idx[k - 1];
                adj_idx[pos] = v; ++w;
            }
        }
        adj_ptr[ns] = w;
        adj_idx.resize(w);
    }
    const int nlast = (int)opt.iface_last.size();
    if (nlast > 0) {
        // the prescribed interface is a clique: every interface row gets all the other interface nodes (merged into its sorted row)
        std::vector<int> ap(ns + 1, 0), ai;
        std::vector<char> is_last(ns, 0);
        for (int v : opt.iface_last) is_last[v] = 1;
        ai.reserve(adj_idx.size() + (size_t)nlast * nlast);
        for (int v = 0; v < ns; ++v) {
            ap[v] = (int)ai.size();
            if (!is_last[v]) { ai.insert(ai.end(), adj_idx.begin() + adj_ptr[v], adj_idx.begin() + adj_ptr[v + 1]); continue; }
            int q = adj_ptr[v]; const int qe = adj_ptr[v + 1]; int w2 = 0;
            while (q < qe || w2 < nlast) {
                const int a = q < qe ? adj_idx[q] : (1 << 30), b = w2 < nlast ? opt.iface_last[w2] : (1 << 30);
                if (a < b) { ai.push_back(a); ++q; }
                else { if (b != v) ai.push_back(b); ++w2; if (a == b) ++q; }
            }
        }
        ap[ns] = (int)ai.size();
        adj_ptr.swap(ap); adj_idx.swap(ai);
    }
    if (opt.before_order) opt.before_order();
    const auto q0a = tnow();
    std::vector<nd_tree> pool; std::mutex mu;
    std::vector<char> iface(ns, 0);
    int root = -1;
    {
        std::vector<int> nodes; nodes.reserve(ns);
        for (int v : opt.iface_last) iface[v] = 1;
        for (int v = 0; v < ns; ++v) if (!iface[v]) nodes.push_back(v);
        std::vector<char> side(ns, 0), side2(ns, 0);
        S.order.reserve(ns);
        std::vector<char> forced(ns, 0);
        std::vector<int> pos_scratch(ns, -1);
        if (S.nparts > 1 && part) for (int v = 0; v < ns; ++v) for (int q = adj_ptr[v]; q < adj_ptr[v + 1]; ++q) if (part[adj_idx[q]] > part[v]) { forced[v] = 1; break; }
        std::atomic<long long> nd_ns[8];
        for (auto& a : nd_ns) a = 0;
        nd_ctx C{ adj_ptr.data(), adj_idx.data(), cx, cy, side.data(), side2.data(), opt.leaf, opt.nd_both_axes, opt.nd_geo_first, S.nparts > 1 ? part : nullptr, iface.data(), forced.data(), &pool, &mu, tv ? nd_ns : nullptr, opt.nd_index_cuts != 0, pos_scratch.data(), T };
        root = nd_order(nodes, C, S.order, 0);
        if (nlast > 0) {        // the prescribed interface behind everything else; in the tree of ranges: a root whose second half is empty
            for (int v : opt.iface_last) S.order.push_back(v);
            pool.push_back({ -1, -1, 0 });
            pool.push_back({ root, (int)pool.size() - 1, ns });
            root = (int)pool.size() - 1;
        }
        if (tv) fprintf(stderr, "[dsss pg symbolic] nd_order thread-time: candidates %.2f ms, final boundary %.2f ms, leaves %.2f ms\n", nd_ns[1] / 1e6, nd_ns[2] / 1e6, nd_ns[4] / 1e6);
    }
    const auto q1 = tnow();
    S.perm.assign(ns, 0);
    for (int i = 0; i < ns; ++i) S.perm[S.order[i]] = i;
    S.col_part.assign(ns, 0);
    for (int j = 0; j < ns; ++j) { const int v = S.order[j]; S.col_part[j] = iface[v] ? -1 : (part && S.nparts > 1 ? part[v] : 0); }
    if (part && S.nparts > 1)
        for (int v = 0; v < ns; ++v) if (!iface[v]) for (int q = adj_ptr[v]; q < adj_ptr[v + 1]; ++q) if (part[adj_idx[q]] > part[v]) { S.ownership_violations++; break; }
    std::vector<cref> cols(ns);
    S.parent.assign(ns, -1);
    std::vector<int> kid_head(ns, -1), kid_next(ns, -1);
    std::vector<std::vector<int>> pools(pool.size() + 1);
    {
        cs_ctx C{ adj_ptr.data(), adj_idx.data(), S.order.data(), S.perm.data(), &pool, &pools, cols.data(), kid_head.data(), kid_next.data(), S.parent.data() };
        std::vector<std::pair<int, int>> up;
        col_structs(C, root, 0, ns, up, 0);
    }
    const auto q2 = tnow();
    S.colptr.assign(ns + 1, 0);
    for (int j = 0; j < ns; ++j) S.colptr[j + 1] = S.colptr[j] + cols[j].n;
    auto csz = [&](int j) { return S.colptr[j + 1] - S.colptr[j]; };
    S.rowidx.resize(S.colptr[ns]);
    S.nnzL = S.colptr[ns];
    par_ranges(ns, T, [&](int, int lo, int hi) {
        for (int k = lo; k < hi; ++k) { const int* d = pools[cols[k].pool].data() + cols[k].off; std::copy(d, d + cols[k].n, S.rowidx.begin() + S.colptr[k]); }
    });
    const auto f0 = tnow();
    const std::vector<int>& parent = S.parent;
    // ---- bottom subtrees -> bins (one workgroup each); only interior columns of one rank, at most 42 blocks per column
    std::vector<double> sub_cost(ns, 0);
    std::vector<char> sub_ok(ns, 0);
    std::vector<int> nsrc(ns, 0);                          // number of source columns of every column = its count of off-diagonal blocks in row j
    {   // a histogram per range of source columns, then summed (788 k scattered increments at C3: a third of this phase when serial)
        std::vector<std::vector<int>> part_h(T > 1 ? T - 1 : 0);
        par_ranges(ns, T, [&](int t, int lo, int hi) {
            int* h = nsrc.data();
            if (t > 0) { part_h[t - 1].assign(ns, 0); h = part_h[t - 1].data(); }
            for (int k = lo; k < hi; ++k) for (int q = S.colptr[k] + 1; q < S.colptr[k + 1]; ++q) h[S.rowidx[q]]++;
        });
        par_ranges(ns, T, [&](int, int lo, int hi) { for (auto& h : part_h) if (!h.empty()) for (int j = lo; j < hi; ++j) nsrc[j] += h[j]; });
    }
    for (int j = 0; j < ns; ++j) {
        const int mj = csz(j);
        double cst = nsrc[j] + 20.0; bool ok = mj <= 42 && S.col_part[j] >= 0;
        for (int k = kid_head[j]; k >= 0; k = kid_next[k]) { cst += sub_cost[k]; ok = ok && sub_ok[k]; }
        sub_cost[j] = cst; sub_ok[j] = ok && cst <= opt.bin_cost;
    }
    S.binned.assign(sub_ok.begin(), sub_ok.end());
    std::vector<int> root_of(ns, -1);                       // subtree root of every binned column
    for (int j = ns - 1; j >= 0; --j) {
        if (!sub_ok[j]) continue;
        const int par = parent[j];
        root_of[j] = (par >= 0 && sub_ok[par]) ? root_of[par] : j;
    }
    S.root_of = root_of;
    if (opt.on_lists_ready && nparts <= 1) opt.on_lists_ready();       // colptr, rowidx, binned, root_of: final
    // subtree roots in bin order, and the roots that hand an update matrix up (the fronts of the top need these; the packing itself not)
    std::vector<int> roots;
    for (int j = 0; j < ns; ++j) if (sub_ok[j] && root_of[j] == j) roots.push_back(j);
    // (a rank's interior may be several ranges of the order when geometric cuts come before rank cuts: its bins stay one range)
    std::stable_sort(roots.begin(), roots.end(), [&](int a, int b) { return S.col_part[a] < S.col_part[b]; });
    std::vector<int> idx_of_root(ns, -1);
    for (int r : roots) if (csz(r) > 1) { idx_of_root[r] = (int)S.broot.size(); S.broot.push_back(r); S.broot_b.push_back(csz(r) - 1); }
    // where the assembled blocks go: value index k < ns diagonal of separator k, then the chain couplings, then the LC edges
    const int ne = (int)edges.size() - nchain, nval = ns + nchain + ne;
    S.dest_bin.assign(nval, -1);
    std::vector<int> v_row(nval), v_col(nval), v_tr(nval);
    par_ranges(nval, T, [&](int, int lo, int hi) {
        for (int v = lo; v < hi; ++v) {
            int pa, pb;                                   // block H(a, b): rows of a, columns of b
            if (v < ns) { pa = pb = S.perm[v]; }
            else { const auto& e = edges[v - ns]; pa = S.perm[e.first]; pb = S.perm[e.second]; }
            v_row[v] = std::max(pa, pb); v_col[v] = std::min(pa, pb); v_tr[v] = (pa >= pb) ? 0 : 1;
        }
    });
    S.nval = nval;
    const auto f1 = tnow();
    // ---- THE BOTTOM TABLES (bin packing, the bins' index lists unless the caller builds them, the destinations of the values that go
    // into binned columns) and THE TOP (fronts and schedule, below) do not depend on each other: with one partition the bottom runs as a
    // task of the pool beside the top, and hands itself over (on_bottom_ready) as soon as it is complete
    std::vector<int> bin_of_root(ns, -1);
    auto bottom_tables = [&] {
        {   // greedy packing of whole subtrees into bins, subtrees taken in ascending root order, never across ranks
            const double pack = opt.pack_cost > 0 ? opt.pack_cost : opt.bin_cost;
            int nbins = 0; double fill = pack + 1; int cur_part = -2;
            std::vector<double> bin_fill;
            for (int r : roots) {
                if (fill + sub_cost[r] > pack || S.col_part[r] != cur_part) { ++nbins; fill = 0; cur_part = S.col_part[r]; S.bin_part.push_back(cur_part); bin_fill.push_back(0.0); }
                fill += sub_cost[r]; bin_of_root[r] = nbins - 1; bin_fill.back() = fill;
            }
            // launch order: descending work inside every rank's (contiguous) range of bins; ties keep the bin order
            S.bin_work = bin_fill;
            S.bin_perm.resize(nbins);
            for (int b = 0; b < nbins; ++b) S.bin_perm[b] = b;
            for (int b0 = 0; b0 < nbins;) {
                int b1 = b0; while (b1 < nbins && S.bin_part[b1] == S.bin_part[b0]) ++b1;
                std::stable_sort(S.bin_perm.begin() + b0, S.bin_perm.begin() + b1, [&](int x, int y) { return bin_fill[x] > bin_fill[y]; });
                b0 = b1;
            }
            S.binptr.assign(nbins + 1, 0);
            for (int j = 0; j < ns; ++j) if (sub_ok[j]) S.binptr[bin_of_root[root_of[j]] + 1]++;
            for (int b = 0; b < nbins; ++b) S.binptr[b + 1] += S.binptr[b];
            S.bincols.resize(S.binptr[nbins]);
            std::vector<int> fillp(S.binptr.begin(), S.binptr.end() - 1);
            for (int j = 0; j < ns; ++j) if (sub_ok[j]) S.bincols[fillp[bin_of_root[root_of[j]]]++] = j;     // ascending within a bin
            S.broot_of_col.assign(ns, -1);
            S.binroot_ptr.assign(nbins + 1, 0);
            for (size_t i = 0; i < S.broot.size(); ++i) S.binroot_ptr[bin_of_root[S.broot[i]] + 1]++;
            for (int b = 0; b < nbins; ++b) S.binroot_ptr[b + 1] += S.binroot_ptr[b];
            S.binroot_idx.resize(S.broot.size());
            { std::vector<int> fp(S.binroot_ptr.begin(), S.binroot_ptr.end() - 1); for (size_t i = 0; i < S.broot.size(); ++i) S.binroot_idx[fp[bin_of_root[S.broot[i]]]++] = (int)i; }
            for (int j = 0; j < ns; ++j) if (sub_ok[j]) S.broot_of_col[j] = idx_of_root[root_of[j]];
            S.broot_uoff.resize(S.broot.size());
            long long o = 0;
            for (size_t i = 0; i < S.broot.size(); ++i) { S.broot_uoff[i] = o; const long long n6 = 6LL * S.broot_b[i]; o += n6 * n6 + n6; o = (o + 31) & ~31LL; }
            S.ubin_doubles = o;
        }
        if (!opt.lists_on_device) {
            // rows of a binned column beyond its subtree root, as indices into the root's boundary list
            S.anc_first.assign(ns, 0);
            S.anc_rel.assign(S.nnzL, -1);
            par_ranges(ns, T, [&](int, int lo, int hi) {
                for (int k = lo; k < hi; ++k) {
                    if (!sub_ok[k]) continue;
                    const int r = root_of[k], c0 = S.colptr[k], m = csz(k);
                    int q = 0;
                    while (q < m && S.rowidx[c0 + q] <= r) ++q;
                    S.anc_first[k] = q;
                    const int* rb = S.rowidx.data() + S.colptr[r] + 1; const int nb = csz(r) - 1;
                    int w = 0;
                    for (; q < m; ++q) { const int row = S.rowidx[c0 + q]; while (w < nb && rb[w] < row) ++w; S.anc_rel[c0 + q] = (w < nb && rb[w] == row) ? w : -1; }
                }
            });
            // update lists of the binned columns (sources are binned columns of the same subtree), ascending source
            std::vector<std::vector<int>> hist(T, std::vector<int>(ns, 0));
            par_ranges(ns, T, [&](int t, int lo, int hi) {
                std::vector<int>& h = hist[t];
                for (int k = lo; k < hi; ++k) { if (!sub_ok[k]) continue; for (int q = S.colptr[k] + 1; q < S.colptr[k + 1]; ++q) { const int j = S.rowidx[q]; if (sub_ok[j]) h[j]++; } }
            });
            S.rlptr.assign(ns + 1, 0);
            for (int j = 0; j < ns; ++j) { int tot = 0; for (int t = 0; t < T; ++t) { const int c = hist[t][j]; hist[t][j] = tot; tot += c; } S.rlptr[j + 1] = S.rlptr[j] + tot; }
            S.rlcol.resize(S.rlptr[ns]); S.rlpos.resize(S.rlptr[ns]); S.rlrow.resize(S.rlptr[ns]);
            par_ranges(ns, T, [&](int t, int lo, int hi) {
                std::vector<int>& fill = hist[t];
                for (int k = lo; k < hi; ++k) {
                    if (!sub_ok[k]) continue;
                    for (int q = S.colptr[k] + 1; q < S.colptr[k + 1]; ++q) {
                        const int j = S.rowidx[q];
                        if (!sub_ok[j]) continue;
                        const int at = S.rlptr[j] + fill[j]++;
                        S.rlcol[at] = k; S.rlpos[at] = q; S.rlrow[at] = j;
                    }
                }
            });
            S.mapptr.assign(ns + 1, 0);
            for (int j = 0; j < ns; ++j) S.mapptr[j + 1] = S.mapptr[j] + (long long)(S.rlptr[j + 1] - S.rlptr[j]) * (long long)csz(j);
        }
        {   // a value whose destination column is binned gets its position in the block-sparse factor (the rest -- fronts, interface --
            // is settled with the fronts)
            auto find = [&](int row, int col) { const auto b = S.rowidx.begin() + S.colptr[col], e = S.rowidx.begin() + S.colptr[col + 1];
                                                return (int)(std::lower_bound(b, e, row) - S.rowidx.begin()); };
            par_ranges(nval, T, [&](int, int lo, int hi) { for (int v = lo; v < hi; ++v) if (sub_ok[v_col[v]]) S.dest_bin[v] = (find(v_row[v], v_col[v]) << 1) | v_tr[v]; });
        }
        if (opt.on_bottom_ready && nparts <= 1) opt.on_bottom_ready();
    };
    pg_pool::task bottom_task; bool bottom_forked = false;
    if (nparts <= 1 && T > 1) { bottom_task.fn = bottom_tables; pg_pool::get().fork(&bottom_task); bottom_forked = true; }
    else bottom_tables();
    struct bottom_joiner { pg_pool::task* t; bool on; ~bottom_joiner() { if (on) pg_pool::get().join(t); } } bottom_join{ &bottom_task, bottom_forked };      // (before anything of this function goes away)
    const auto q3 = tnow();
    if (tv) fprintf(stderr, "[dsss pg symbolic] bins: flags + roots + value coordinates %.2f ms; bottom tables %s%s\n", tms(f0, f1), bottom_forked ? "as a task beside the fronts" : "in line",
                    opt.lists_on_device ? " (lists and root indices: on the device)" : "");
    // ---- top: supernodes of the remaining columns become fronts.  Fundamental supernodes (consecutive columns with nested
    // structure) first; then RELAXED amalgamation along the column order: a front whose columns end where its parent's begin is
    // merged into the parent when that adds few explicit zero blocks -- every merge removes a level of the schedule, and the
    // levels (one dependent launch sequence each) are what the factorisation time is made of.
    S.front_of_col.assign(ns, -1);
    {
        struct fnd { int c0, s, n; };                       // fundamental supernodes, ascending
        std::vector<fnd> fund;
        for (int j = 0; j < ns; ++j) {
            if (sub_ok[j]) continue;
            const bool chain = !fund.empty() && fund.back().c0 + fund.back().s == j && parent[j - 1] == j && csz(j) + 1 == csz(j - 1) && S.col_part[j] == S.col_part[j - 1];
            if (chain) fund.back().s++;
            else fund.push_back({ j, 1, csz(j) });
        }
        // sum over the s columns of a front of n block rows of 216 (m^2 + 3 m) + 72, m = n - 1 - j: in closed form and in integers (the
        // column-by-column sum is a sum of integers below 2^53, i.e. exact: the same double, so the same merge decisions; as a loop it made
        // every merge test of a growing front linear in its width -- 0.3 of the 0.5 ms of this phase, which the GPU waits for)
        auto flops_of = [](double s_, double n_) {
            const long long s = (long long)s_, n = (long long)n_;
            auto Q = [](long long k) { return k <= 0 ? 0LL : k * (k + 1) * (2 * k + 1) / 6; };
            auto T1 = [](long long k) { return k <= 0 ? 0LL : k * (k + 1) / 2; };
            if (s <= 0) return 0.0;
            const long long hi = n - 1, lo = n - s - 1;         // m runs over lo + 1 .. hi
            return (double)(216 * ((Q(hi) - Q(lo)) + 3 * (T1(hi) - T1(lo))) + 72 * s);
        };
        // The merge decisions need sizes and ONE row of a front -- its first boundary row -- and the rows of a front are its own columns
        // (consecutive) followed by the boundary of the LAST supernode merged into it: decided here in one cheap pass (f_last = that
        // supernode's first column), the row lists are then written by all threads (at C5 they are 13 M entries: 8 ms when one thread
        // copied them while it decided).
        std::vector<int> f_last;
        for (const fnd& g : fund) {
            bool merged = false;
            if (!S.f_c0.empty()) {
                const int c = (int)S.f_c0.size() - 1;
                const int sc = S.f_s[c], nc = S.f_n[c], bc = nc - sc;
                const int lastc = f_last[c];                                  // the boundary of front c is the tail of this column's structure
                const int bfirst = bc > 0 ? S.rowidx[S.colptr[lastc] + (csz(lastc) - bc)] : -1;
                if (S.f_c0[c] + sc == g.c0 && bc > 0 && bfirst >= g.c0 && bfirst < g.c0 + g.s && S.f_part[c] == S.col_part[g.c0]) {
                    const double zeros = (double)(g.n - bc) * sc;                     // explicit zero blocks the merge puts into L
                    const double f_sep = flops_of(sc, nc) + flops_of(g.s, g.n), f_mrg = flops_of(sc + g.s, sc + g.n);
                    const bool one_panel = sc + g.s <= PG_PW;
                    // a merge that removes a panel step (the two column counts round up to fewer panels together) is worth a bounded
                    // amount of extra arithmetic whatever the ratio: a level costs about 100 us of dependent launches, i.e. more than
                    // a GFLOP of trailing update
                    const bool saves_step = (sc + g.s + PG_PW - 1) / PG_PW < (sc + PG_PW - 1) / PG_PW + (g.s + PG_PW - 1) / PG_PW;
                    if (zeros <= opt.relax_zero_blocks || f_mrg <= f_sep * (one_panel ? opt.relax_flops_small : opt.relax_flops) ||
                        (saves_step && f_mrg - f_sep <= opt.relax_abs_flops)) {
                        S.f_s[c] = sc + g.s; S.f_n[c] = sc + g.n; f_last[c] = g.c0;
                        merged = true;
                    }
                }
            }
            if (!merged) { S.f_c0.push_back(g.c0); S.f_s.push_back(g.s); S.f_n.push_back(g.n); S.f_part.push_back(S.col_part[g.c0]); f_last.push_back(g.c0); }
        }
        {
            const int nfr = (int)S.f_c0.size();
            S.f_rowptr.assign(nfr + 1, 0);
            for (int f = 0; f < nfr; ++f) S.f_rowptr[f + 1] = S.f_rowptr[f] + S.f_n[f];
            S.f_rows.resize(S.f_rowptr[nfr]);
            par_ranges(nfr, T, [&](int, int lo, int hi) {
                for (int f = lo; f < hi; ++f) {
                    int* out = S.f_rows.data() + S.f_rowptr[f];
                    const int sc = S.f_s[f], bc = S.f_n[f] - sc, lastc = f_last[f];
                    for (int q = 0; q < sc; ++q) out[q] = S.f_c0[f] + q;
                    std::copy(S.rowidx.begin() + S.colptr[lastc] + (csz(lastc) - bc), S.rowidx.begin() + S.colptr[lastc + 1], out + sc);
                }
            });
        }
        for (size_t f = 0; f < S.f_c0.size(); ++f) for (int c = 0; c < S.f_s[f]; ++c) S.front_of_col[S.f_c0[f] + c] = (int)f;
    }
    const int nf = (int)S.f_c0.size();
    S.f_ld.resize(nf); S.f_off.resize(nf); S.f_roff.resize(nf); S.f_parent.assign(nf, -1);
    {
        long long o = 0, ro = 0;
        for (int f = 0; f < nf; ++f) {
            const int ld = (6 * S.f_n[f] + 15) & ~15;
            S.f_ld[f] = ld; S.f_off[f] = o; o += (long long)ld * ld;
            S.f_roff[f] = ro; ro += ld;
            S.max_front_n = std::max(S.max_front_n, S.f_n[f]);
            if (S.f_n[f] > S.f_s[f]) S.f_parent[f] = S.front_of_col[S.f_rows[S.f_rowptr[f] + S.f_s[f]]];
            S.flops_fronts += [&] { double fl = 0; for (int j = 0; j < S.f_s[f]; ++j) { const double m = S.f_n[f] - j - 1; fl += 216.0 * (m * m + 3 * m) + 72.0; } return fl; }();      // (once per front: linear in total)
        }
        S.front_doubles = o; S.frhs_doubles = ro;
    }
    const auto fA = tnow();
    // children (bin roots first, then fronts, both ascending) and their boundary -> parent row maps
    {
        std::vector<int> cnt(nf + 1, 0);
        std::vector<int> broot_front(S.broot.size());
        for (size_t i = 0; i < S.broot.size(); ++i) { const int p = parent[S.broot[i]]; broot_front[i] = S.front_of_col[p]; cnt[broot_front[i] + 1]++; }
        for (int f = 0; f < nf; ++f) if (S.f_parent[f] >= 0) cnt[S.f_parent[f] + 1]++;
        S.ch_ptr.assign(nf + 1, 0);
        for (int f = 0; f < nf; ++f) S.ch_ptr[f + 1] = S.ch_ptr[f] + cnt[f + 1];
        const int nch = S.ch_ptr[nf];
        S.ch_kind.resize(nch); S.ch_id.resize(nch); S.ch_relptr.resize(nch + 1);
        std::vector<int> fp(S.ch_ptr.begin(), S.ch_ptr.end() - 1);
        for (size_t i = 0; i < S.broot.size(); ++i) { const int at = fp[broot_front[i]]++; S.ch_kind[at] = 1; S.ch_id[at] = (int)i; }
        for (int f = 0; f < nf; ++f) if (S.f_parent[f] >= 0) { const int at = fp[S.f_parent[f]]++; S.ch_kind[at] = 0; S.ch_id[at] = f; }
        long long o = 0;
        for (int c = 0; c < nch; ++c) { S.ch_relptr[c] = o; o += S.ch_kind[c] ? S.broot_b[S.ch_id[c]] : S.f_n[S.ch_id[c]] - S.f_s[S.ch_id[c]]; }
        S.ch_relptr[nch] = o;
        S.rel.assign(o, -1);
        par_ranges(nf, T, [&](int, int lo, int hi) {
            for (int f = lo; f < hi; ++f) {
                const int* pr = S.f_rows.data() + S.f_rowptr[f]; const int pn = S.f_n[f];
                for (int c = S.ch_ptr[f]; c < S.ch_ptr[f + 1]; ++c) {
                    const int* cr; int cn;
                    if (S.ch_kind[c]) { const int r = S.broot[S.ch_id[c]]; cr = S.rowidx.data() + S.colptr[r] + 1; cn = csz(r) - 1; }
                    else { const int g = S.ch_id[c]; cr = S.f_rows.data() + S.f_rowptr[g] + S.f_s[g]; cn = S.f_n[g] - S.f_s[g]; }
                    int w = 0;
                    for (int q = 0; q < cn; ++q) { while (w < pn && pr[w] < cr[q]) ++w; S.rel[S.ch_relptr[c] + q] = (w < pn && pr[w] == cr[q]) ? w : -1; }
                }
            }
        });
    }
    const auto fB = tnow();
    // ---- where the assembled blocks go, second half: the original entries of every front
    {
        std::vector<int> cnt(nf + 1, 0);
        for (int v = 0; v < nval; ++v) if (!sub_ok[v_col[v]]) cnt[S.front_of_col[v_col[v]] + 1]++;
        S.fa_ptr.assign(nf + 1, 0);
        for (int f = 0; f < nf; ++f) S.fa_ptr[f + 1] = S.fa_ptr[f] + cnt[f + 1];
        const int nfa = S.fa_ptr[nf];
        S.fa_src.resize(nfa); S.fa_row.resize(nfa); S.fa_col.resize(nfa); S.fa_tr.resize(nfa);
        {   // (any order inside a front: the sort below orders by (row, column, value index), and the value index is unique)
            std::vector<std::atomic<int>> fp(nf);
            for (int f = 0; f < nf; ++f) fp[f].store(S.fa_ptr[f], std::memory_order_relaxed);
            par_ranges(nval, T, [&](int, int lo, int hi) {
                for (int v = lo; v < hi; ++v) {
                    if (sub_ok[v_col[v]]) continue;
                    const int f = S.front_of_col[v_col[v]], at = fp[f].fetch_add(1, std::memory_order_relaxed);
                    const int* pr = S.f_rows.data() + S.f_rowptr[f];
                    S.fa_src[at] = v; S.fa_col[at] = v_col[v] - S.f_c0[f]; S.fa_tr[at] = v_tr[v];
                    S.fa_row[at] = (int)(std::lower_bound(pr, pr + S.f_n[f], v_row[v]) - pr);
                }
            });
        }
        par_ranges(nf, T, [&](int, int lo, int hi) {
            std::vector<int> idx, a, b, c2, d;
            for (int f = lo; f < hi; ++f) {
                const int b0 = S.fa_ptr[f], n = S.fa_ptr[f + 1] - b0;
                idx.resize(n); std::iota(idx.begin(), idx.end(), 0);
                std::sort(idx.begin(), idx.end(), [&](int x, int y) {
                    if (S.fa_row[b0 + x] != S.fa_row[b0 + y]) return S.fa_row[b0 + x] < S.fa_row[b0 + y];
                    if (S.fa_col[b0 + x] != S.fa_col[b0 + y]) return S.fa_col[b0 + x] < S.fa_col[b0 + y];
                    return S.fa_src[b0 + x] < S.fa_src[b0 + y]; });
                a.resize(n); b.resize(n); c2.resize(n); d.resize(n);
                for (int i = 0; i < n; ++i) { a[i] = S.fa_src[b0 + idx[i]]; b[i] = S.fa_row[b0 + idx[i]]; c2[i] = S.fa_col[b0 + idx[i]]; d[i] = S.fa_tr[b0 + idx[i]]; }
                for (int i = 0; i < n; ++i) { S.fa_src[b0 + i] = a[i]; S.fa_row[b0 + i] = b[i]; S.fa_col[b0 + i] = c2[i]; S.fa_tr[b0 + i] = d[i]; }
            }
        });
    }
    const auto fC = tnow();
    if (!opt.to_be_joined) sym_levels(S);
    if (tv && atoi(getenv("DSSS_PG_VERBOSE")) >= 2 && !opt.to_be_joined) {      // critical path of the schedule, root first
        int f = -1;
        for (int g = 0; g < nf; ++g) if (S.f_level0[g] + S.f_npan[g] == S.nlev) f = g;
        while (f >= 0) {
            fprintf(stderr, "[dsss pg path] front %d: cols %d rows %d panels %d levels %d..%d part %d children %d\n", f, S.f_s[f], S.f_n[f], S.f_npan[f], S.f_level0[f], S.f_level0[f] + S.f_npan[f] - 1, S.f_part[f], S.ch_ptr[f + 1] - S.ch_ptr[f]);
            int nxt = -1;
            for (int c = S.ch_ptr[f]; c < S.ch_ptr[f + 1]; ++c) if (!S.ch_kind[c]) { const int g = S.ch_id[c]; if (S.f_level0[g] + S.f_npan[g] == S.f_level0[f]) nxt = g; }
            f = nxt;
        }
    }
    const auto fD = tnow();
    if (!opt.to_be_joined) sym_row_views(S, T);
    const auto fE = tnow();
    // ---- children that cross from a rank's interior into the interface
    {
        long long o = 0;
        for (int f = 0; f < nf; ++f) {
            if (S.f_part[f] >= 0) continue;
            for (int c = S.ch_ptr[f]; c < S.ch_ptr[f + 1]; ++c) {
                const int kind = S.ch_kind[c], id = S.ch_id[c];
                const int cp = kind ? S.col_part[S.broot[id]] : S.f_part[id];
                if (cp < 0 || nlast > 0) continue;               // (rank-local analysis: the interface front is summed whole, its children stay where they are)
                const long long n6 = 6LL * (kind ? S.broot_b[id] : S.f_n[id] - S.f_s[id]);
                S.comm_kind.push_back(kind); S.comm_id.push_back(id); S.comm_part.push_back(cp); S.comm_off.push_back(o);
                o += n6 * n6 + n6; o = (o + 31) & ~31LL;
            }
        }
        S.comm_doubles = o;
        // original values of interface fronts and the interface separators
        S.nval = nval;
        if (S.nparts > 1 || nlast > 0) {
            std::vector<int> slot_of(nval, -1);
            for (int f = 0; f < nf && !opt.iface_plain; ++f) {
                if (S.f_part[f] >= 0) continue;
                for (int e = S.fa_ptr[f]; e < S.fa_ptr[f + 1]; ++e) {
                    const int v = S.fa_src[e];
                    if (slot_of[v] < 0) { slot_of[v] = (int)S.comm_vals.size(); S.comm_vals.push_back(v); S.dest_bin[v] = -2 - slot_of[v]; }
                    S.fa_src[e] = nval + slot_of[v];
                }
            }
            for (int k = 0; k < ns; ++k) if (S.col_part[S.perm[k]] < 0) S.iface_seps.push_back(k);
        }
    }
    if (bottom_join.on) { pg_pool::get().join(&bottom_task); bottom_join.on = false; }      // the bottom tables are complete from here on
    // statistics
    for (int j = 0; j < ns; ++j) { const double m = csz(j) - 1; S.flops_factor += 36.0 * 6.0 * (m * m + 3 * m) + 72.0; }
    if (tv) fprintf(stderr, "[dsss pg symbolic] fronts+schedule: supernodes %.2f, children+rel %.2f, destinations+entries %.2f, schedule %.2f, row views %.2f, rest %.2f ms\n", tms(q3, fA), tms(fA, fB), tms(fB, fC), tms(fC, fD), tms(fD, fE), tms(fE, tnow()));
    if (tv) {
        long long zeros = 0;
        int big = 0;
        for (int f = 0; f < nf; ++f) { if (S.f_n[f] > 100) ++big; zeros += 0; }
        fprintf(stderr, "[dsss pg symbolic] adjacency %.2f + ND %.2f ms, column structures %.1f ms, bins+lists %.1f ms, fronts+schedule %.1f ms | ns %d nnzL %lld bins %d (%zu cols, %zu roots, U %.1f MB) fronts %d (>100 rows: %d, max %d) panels %d levels %d front arena %.1f MB comm %.1f MB front GFLOP %.1f (column count %.1f)\n",
                tms(q0, q0a), tms(q0a, q1), tms(q1, q2), tms(q2, q3), tms(q3, tnow()), ns, S.nnzL, (int)S.binptr.size() - 1, S.bincols.size(), S.broot.size(), S.ubin_doubles * 8e-6,
                nf, big, S.max_front_n, S.npanels, S.nlev, S.front_doubles * 8e-6, S.comm_doubles * 8e-6, S.flops_fronts * 1e-9, S.flops_factor * 1e-9);
    }
}

void pg_build_schedule(const pg_sym& S, int part_lo, int part_hi, pg_sched& out)
{
    out = pg_sched();
    out.nlev = S.nlev;
    const int nl = S.nlev;
    out.lv_ptr.assign(nl + 1, 0); out.asmrow_ptr.assign(nl + 1, 0); out.tile_ptr.assign(nl + 1, 0);
    out.trsm_chunks.assign(nl, 0); out.fl_diag.assign(nl, 0); out.fl_trsm.assign(nl, 0); out.fl_syrk.assign(nl, 0); out.fl_bwd.assign(nl, 0);
    out.max_w6.assign(nl, 0); out.max_rows.assign(nl, 0);
    auto sel = [&](int f) { return S.f_part[f] >= part_lo && S.f_part[f] < part_hi; };
    {   // (sizes first: the lists are filled on the analysing thread, which the GPU waits for -- no reallocation on the way)
        size_t nrow = 0, nitem = 0, ntile = 0;
        for (size_t f = 0; f < S.f_c0.size(); ++f) {
            if (!sel((int)f)) continue;
            nrow += (size_t)S.f_n[f]; nitem += (size_t)S.f_npan[f];
            for (int k = 0; k < S.f_npan[f]; ++k) { const int n6 = 6 * S.f_n[f], w6 = std::min(96, 6 * S.f_s[f] - 96 * k), nt = (n6 - 96 * k - w6 + 63) / 64; ntile += (size_t)nt * (nt + 1) / 2; }
        }
        out.asmrow_front.reserve(nrow); out.asmrow_row.reserve(nrow); out.lv_front.reserve(nitem); out.lv_step.reserve(nitem); out.tile_item.reserve(ntile); out.tile_ij.reserve(ntile);
    }
    for (int l = 0; l < nl; ++l) {
        for (int q = S.asm_ptr[l]; q < S.asm_ptr[l + 1]; ++q) { const int f = S.asm_front[q]; if (!sel(f)) continue; for (int r = 0; r < S.f_n[f]; ++r) { out.asmrow_front.push_back(f); out.asmrow_row.push_back(r); } }
        out.asmrow_ptr[l + 1] = (int)out.asmrow_front.size();
        for (int q = S.lv_ptr[l]; q < S.lv_ptr[l + 1]; ++q) {
            const int f = S.lv_front[q], k = S.lv_step[q];
            if (!sel(f)) continue;
            const int item = (int)out.lv_front.size() - out.lv_ptr[l];
            out.lv_front.push_back(f); out.lv_step.push_back(k);
            const int n6 = 6 * S.f_n[f], w6 = std::min(96, 6 * S.f_s[f] - 96 * k), nrows = n6 - 96 * k - w6, nt = (nrows + 63) / 64;
            for (int ti = 0; ti < nt; ++ti) for (int tj = 0; tj <= ti; ++tj) { out.tile_item.push_back(item); out.tile_ij.push_back((ti << 16) | tj); }
            out.max_n6 = std::max(out.max_n6, n6);
            out.max_w6[l] = std::max(out.max_w6[l], w6); out.max_rows[l] = std::max(out.max_rows[l], nrows);
            out.trsm_chunks[l] = std::max(out.trsm_chunks[l], (nrows + 63) / 64);
            const double nn = w6, rows = nrows;
            out.fl_diag[l] += nn * nn * nn / 3.0 + nn * nn; out.fl_trsm[l] += rows * nn * nn; out.fl_bwd[l] += 2.0 * rows * nn + nn * nn; out.fl_syrk[l] += rows * (rows + 1) * nn;
        }
        out.lv_ptr[l + 1] = (int)out.lv_front.size(); out.tile_ptr[l + 1] = (int)out.tile_item.size();
    }
}

void pg_sym_opts_env(pg_sym_opts& opt)
{
    if (getenv("DSSS_PG_ND_BOTH")) opt.nd_both_axes = atoi(getenv("DSSS_PG_ND_BOTH"));
    if (getenv("DSSS_PG_LEAF")) opt.leaf = atoi(getenv("DSSS_PG_LEAF"));
    if (getenv("DSSS_PG_ND_INDEX")) opt.nd_index_cuts = atoi(getenv("DSSS_PG_ND_INDEX"));      // (tools/sym_time.py, tools/pg_sweep.sh: the chain-order cut candidate on / off)
}

// ------------------------------------------------------------------ host twin of the numeric phase (CPU tests only)
namespace {
int h_chol(double* A, int n, int ld)          // in-place lower Cholesky of the leading n x n block
{
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * ld + j];
        for (int k = 0; k < j; ++k) d -= A[(size_t)j * ld + k] * A[(size_t)j * ld + k];
        if (!(d > 0) || !std::isfinite(d)) return -1;
        d = std::sqrt(d); A[(size_t)j * ld + j] = d;
        for (int i = j + 1; i < n; ++i) { double s = A[(size_t)i * ld + j]; for (int k = 0; k < j; ++k) s -= A[(size_t)i * ld + k] * A[(size_t)j * ld + k]; A[(size_t)i * ld + j] = s / d; }
    }
    return 0;
}
} // namespace

int pg_host_solve(const pg_sym& S, int ne, const std::vector<std::pair<int, int>>& edges, const double* aval, const double* rhs, double* xout)
{
    (void)edges; (void)ne;
    const int ns = S.ns, nf = (int)S.f_c0.size();
    auto csz = [&](int j) { return S.colptr[j + 1] - S.colptr[j]; };
    std::vector<double> L((size_t)S.nnzL * 36, 0.0), x((size_t)ns * 6, 0.0), F((size_t)S.front_doubles, 0.0), R((size_t)S.frhs_doubles, 0.0), U((size_t)S.ubin_doubles, 0.0);
    for (int k = 0; k < ns; ++k) for (int a = 0; a < 6; ++a) x[(size_t)S.perm[k] * 6 + a] = rhs[(size_t)k * 6 + a];
    const int nval = (int)S.dest_bin.size();
    for (int v = 0; v < nval; ++v) {
        if (S.dest_bin[v] < 0) continue;
        const int pos = S.dest_bin[v] >> 1, tr = S.dest_bin[v] & 1;
        for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) L[(size_t)pos * 36 + (tr ? b * 6 + a : a * 6 + b)] += aval[(size_t)v * 36 + a * 6 + b];
    }
    // bins, column by column (left-looking)
    for (size_t ci = 0; ci < S.bincols.size(); ++ci) {
        const int j = S.bincols[ci], c0 = S.colptr[j], m = csz(j);
        for (int t = S.rlptr[j]; t < S.rlptr[j + 1]; ++t) {
            const int k = S.rlcol[t], pj = S.rlpos[t];
            const double* Ljk = &L[(size_t)pj * 36];
            int q = 0;
            for (int p = pj; p < S.colptr[k + 1]; ++p) {
                const int i = S.rowidx[p];
                while (q < m && S.rowidx[c0 + q] < i) ++q;
                if (q >= m || S.rowidx[c0 + q] != i) return -2;         // structure violated
                const double* Lik = &L[(size_t)p * 36]; double* Aij = &L[(size_t)(c0 + q) * 36];
                for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) { double s = 0; for (int c = 0; c < 6; ++c) s += Lik[a * 6 + c] * Ljk[b * 6 + c]; Aij[a * 6 + b] -= s; }
            }
            for (int a = 0; a < 6; ++a) { double s = 0; for (int c = 0; c < 6; ++c) s += Ljk[a * 6 + c] * x[(size_t)k * 6 + c]; x[(size_t)j * 6 + a] -= s; }
        }
        double* D = &L[(size_t)c0 * 36];
        if (h_chol(D, 6, 6)) return -1;
        for (int a = 0; a < 6; ++a) for (int b = a + 1; b < 6; ++b) D[a * 6 + b] = 0;
        for (int a = 0; a < 6; ++a) { double s = x[(size_t)j * 6 + a]; for (int b = 0; b < a; ++b) s -= D[a * 6 + b] * x[(size_t)j * 6 + b]; x[(size_t)j * 6 + a] = s / D[a * 6 + a]; }
        for (int q = 1; q < m; ++q) {
            double* B = &L[(size_t)(c0 + q) * 36];
            for (int a = 0; a < 6; ++a) for (int s6 = 0; s6 < 6; ++s6) { double v = B[a * 6 + s6]; for (int c = 0; c < s6; ++c) v -= B[a * 6 + c] * D[s6 * 6 + c]; B[a * 6 + s6] = v / D[s6 * 6 + s6]; }
        }
        // contribution to the update matrix of the subtree root
        const int ri = S.broot_of_col[j];
        if (ri >= 0) {
            const int b6 = 6 * S.broot_b[ri]; double* Ur = &U[(size_t)S.broot_uoff[ri]]; double* gr = Ur + (size_t)b6 * b6;
            for (int p = c0 + S.anc_first[j]; p < c0 + m; ++p) {
                const int ia = S.anc_rel[p]; if (ia < 0) return -3;
                const double* La = &L[(size_t)p * 36];
                for (int a = 0; a < 6; ++a) { double s = 0; for (int c = 0; c < 6; ++c) s += La[a * 6 + c] * x[(size_t)j * 6 + c]; gr[ia * 6 + a] -= s; }
                for (int p2 = c0 + S.anc_first[j]; p2 <= p; ++p2) {
                    const int ib = S.anc_rel[p2]; const double* Lb = &L[(size_t)p2 * 36];
                    for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) { double s = 0; for (int c = 0; c < 6; ++c) s += La[a * 6 + c] * Lb[b * 6 + c]; Ur[(size_t)(ia * 6 + a) * b6 + ib * 6 + b] -= s; }
                }
            }
        }
    }
    // fronts, ascending (children come first)
    for (int f = 0; f < nf; ++f) {
        const int ld = S.f_ld[f], s6 = 6 * S.f_s[f], n6 = 6 * S.f_n[f], c0 = S.f_c0[f];
        double* A = &F[(size_t)S.f_off[f]]; double* r = &R[(size_t)S.f_roff[f]];
        for (int e = S.fa_ptr[f]; e < S.fa_ptr[f + 1]; ++e) {
            const int vsrc = S.fa_src[e] >= S.nval ? S.comm_vals[S.fa_src[e] - S.nval] : S.fa_src[e];
            const double* B = aval + (size_t)vsrc * 36; const int tr = S.fa_tr[e];
            for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) A[(size_t)(S.fa_row[e] * 6 + a) * ld + S.fa_col[e] * 6 + b] += tr ? B[b * 6 + a] : B[a * 6 + b];
        }
        for (int i = 0; i < s6; ++i) r[i] = x[(size_t)c0 * 6 + i];
        for (int c = S.ch_ptr[f]; c < S.ch_ptr[f + 1]; ++c) {
            const int* rl = &S.rel[S.ch_relptr[c]];
            const double* Uc; const double* gc; int cb, cld;
            if (S.ch_kind[c]) { const int ri = S.ch_id[c]; cb = S.broot_b[ri]; cld = 6 * cb; Uc = &U[(size_t)S.broot_uoff[ri]]; gc = Uc + (size_t)cld * cld; }
            else { const int g = S.ch_id[c]; cb = S.f_n[g] - S.f_s[g]; cld = S.f_ld[g]; Uc = &F[(size_t)S.f_off[g]] + (size_t)(6 * S.f_s[g]) * cld + 6 * S.f_s[g]; gc = &R[(size_t)S.f_roff[g]] + 6 * S.f_s[g]; }
            for (int i = 0; i < cb; ++i) {
                if (rl[i] < 0) return -4;
                for (int a = 0; a < 6; ++a) {
                    r[rl[i] * 6 + a] += gc[i * 6 + a];
                    for (int j2 = 0; j2 <= i; ++j2) for (int b = 0; b < 6; ++b) A[(size_t)(rl[i] * 6 + a) * ld + rl[j2] * 6 + b] += Uc[(size_t)(i * 6 + a) * cld + j2 * 6 + b];
                }
            }
        }
        if (h_chol(A, s6, ld)) return -1;
        for (int i = 0; i < s6; ++i) { double s = r[i]; for (int k = 0; k < i; ++k) s -= A[(size_t)i * ld + k] * r[k]; r[i] = s / A[(size_t)i * ld + i]; }
        for (int i = s6; i < n6; ++i) {
            for (int j2 = 0; j2 < s6; ++j2) { double s = A[(size_t)i * ld + j2]; for (int k = 0; k < j2; ++k) s -= A[(size_t)i * ld + k] * A[(size_t)j2 * ld + k]; A[(size_t)i * ld + j2] = s / A[(size_t)j2 * ld + j2]; }
            double s = 0; for (int k = 0; k < s6; ++k) s += A[(size_t)i * ld + k] * r[k];
            r[i] -= s;
            for (int j2 = s6; j2 <= i; ++j2) { double t = 0; for (int k = 0; k < s6; ++k) t += A[(size_t)i * ld + k] * A[(size_t)j2 * ld + k]; A[(size_t)i * ld + j2] -= t; }
        }
    }
    // backward: fronts descending, then binned columns descending
    for (int f = nf - 1; f >= 0; --f) {
        const int ld = S.f_ld[f], s6 = 6 * S.f_s[f], n6 = 6 * S.f_n[f], c0 = S.f_c0[f];
        const double* A = &F[(size_t)S.f_off[f]]; const double* r = &R[(size_t)S.f_roff[f]];
        const int* rows = &S.f_rows[S.f_rowptr[f]];
        std::vector<double> z(r, r + s6);
        for (int i = s6; i < n6; ++i) { const double xi = x[(size_t)rows[i / 6] * 6 + i % 6]; for (int k = 0; k < s6; ++k) z[k] -= A[(size_t)i * ld + k] * xi; }
        for (int i = s6 - 1; i >= 0; --i) { double s = z[i]; for (int k = i + 1; k < s6; ++k) s -= A[(size_t)k * ld + i] * z[k]; z[i] = s / A[(size_t)i * ld + i]; }
        for (int i = 0; i < s6; ++i) x[(size_t)c0 * 6 + i] = z[i];
    }
    for (int ci = (int)S.bincols.size() - 1; ci >= 0; --ci) {
        const int j = S.bincols[ci], c0 = S.colptr[j], m = csz(j);
        double z[6];
        for (int a = 0; a < 6; ++a) z[a] = x[(size_t)j * 6 + a];
        for (int q = 1; q < m; ++q) { const double* B = &L[(size_t)(c0 + q) * 36]; const double* xi = &x[(size_t)S.rowidx[c0 + q] * 6]; for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) z[a] -= B[b * 6 + a] * xi[b]; }
        const double* D = &L[(size_t)c0 * 36];
        for (int a = 5; a >= 0; --a) { double s = z[a]; for (int b = a + 1; b < 6; ++b) s -= D[b * 6 + a] * z[b]; z[a] = s / D[a * 6 + a]; }
        for (int a = 0; a < 6; ++a) x[(size_t)j * 6 + a] = z[a];
    }
    for (int k = 0; k < ns; ++k) for (int a = 0; a < 6; ++a) xout[(size_t)k * 6 + a] = x[(size_t)S.perm[k] * 6 + a];
    return 0;
}

// ------------------------------------------------------------------ C ABI of the host twin (CPU test-suite; include/dsss.h)
#include "../../include/dsss.h"
extern "C" int dsss_host_pg_solve(int ns, const int32_t* edge_a, const int32_t* edge_b, int nedges, const double* cx, const double* cy,
                                  const int32_t* part, int nparts, const double* aval, const double* rhs, double* x, int64_t* stats8)
{
    if (ns < 1 || nedges < ns - 1 || !edge_a || !edge_b || !cx || !cy || (x && (!aval || !rhs))) return DSSS_E_ARG;      // x == NULL: analysis only
    std::vector<std::pair<int, int>> edges(nedges);
    for (int e = 0; e < nedges; ++e) {
        edges[e] = { edge_a[e], edge_b[e] };
        if (edge_a[e] < 0 || edge_a[e] >= ns || edge_b[e] < 0 || edge_b[e] >= ns) return DSSS_E_ARG;
        if (e < ns - 1 && (edge_a[e] != e || edge_b[e] != e + 1)) return DSSS_E_ARG;       // the chain couplings come first
    }
    pg_sym S; pg_sym_opts opt;
    if (getenv("DSSS_PG_BIN_COST")) opt.bin_cost = atof(getenv("DSSS_PG_BIN_COST"));
    if (getenv("DSSS_PG_LEAF")) opt.leaf = atoi(getenv("DSSS_PG_LEAF"));
    if (getenv("DSSS_PG_ND_BOTH")) opt.nd_both_axes = atoi(getenv("DSSS_PG_ND_BOTH"));
    pg_sym_opts_env(opt);
    if (getenv("DSSS_SYM_THREADS")) opt.threads = std::max(1, atoi(getenv("DSSS_SYM_THREADS")));
    opt.lists_on_device = x == nullptr;                     // analysis only: as the product runs it (the bins' lists are built on the device there)
    pg_symbolic(ns, edges, ns - 1, cx, cy, part, nparts, opt, S);
    if (S.ownership_violations) return DSSS_E_STATE;        // a lower-rank end of a cross-rank factor outside the interface
    if (!x) {      // (what the product builds next; timed with the analysis by tools/sym_time.py)
        const auto t0 = std::chrono::steady_clock::now();
        pg_sched so, si; pg_build_schedule(S, 0, std::max(1, nparts), so); if (nparts > 1) pg_build_schedule(S, -1, 0, si);
        if (getenv("DSSS_PG_VERBOSE")) fprintf(stderr, "[dsss pg symbolic] launch lists %.2f ms (%zu assembly rows, %zu panel steps, %zu tiles)\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                                               so.asmrow_front.size() + si.asmrow_front.size(), so.lv_front.size() + si.lv_front.size(), so.tile_item.size() + si.tile_item.size());
    }
    const int rc = x ? pg_host_solve(S, nedges - (ns - 1), edges, aval, rhs, x) : 0;
    if (stats8) {
        stats8[0] = S.nnzL; stats8[1] = (int64_t)S.f_c0.size(); stats8[2] = S.npanels; stats8[3] = S.nlev;
        stats8[4] = S.front_doubles; stats8[5] = S.comm_doubles; stats8[6] = (int64_t)S.bincols.size(); stats8[7] = S.max_front_n;
    }
    return rc == 0 ? DSSS_OK : (rc == -1 ? DSSS_E_NUMERIC : DSSS_E_STATE);
}

// the same with a PRESCRIBED interface (pg_sym_opts::iface_last): the analysis one rank of several runs on its own separators + the
// interface nodes.  Any edge list (no chain prefix); iface_last ascending.
extern "C" int dsss_host_pg_solve_local(int ns, const int32_t* edge_a, const int32_t* edge_b, int nedges, const double* cx, const double* cy,
                                        const int32_t* iface_last, int nlast, const double* aval, const double* rhs, double* x, int64_t* stats8)
{
    if (ns < 1 || nedges < 0 || (nedges > 0 && (!edge_a || !edge_b)) || !cx || !cy || nlast < 0 || nlast > ns || (nlast > 0 && !iface_last) || !x || !aval || !rhs) return DSSS_E_ARG;
    std::vector<std::pair<int, int>> edges(nedges);
    for (int e = 0; e < nedges; ++e) {
        edges[e] = { edge_a[e], edge_b[e] };
        if (edge_a[e] < 0 || edge_a[e] >= ns || edge_b[e] < 0 || edge_b[e] >= ns || edge_a[e] == edge_b[e]) return DSSS_E_ARG;
    }
    pg_sym S; pg_sym_opts opt;
    if (getenv("DSSS_PG_BIN_COST")) opt.bin_cost = atof(getenv("DSSS_PG_BIN_COST"));
    pg_sym_opts_env(opt);
    for (int q = 0; q < nlast; ++q) {
        if (iface_last[q] < 0 || iface_last[q] >= ns || (q > 0 && iface_last[q] <= iface_last[q - 1])) return DSSS_E_ARG;
        opt.iface_last.push_back(iface_last[q]);
    }
    pg_symbolic(ns, edges, 0, cx, cy, nullptr, 1, opt, S);
    // the interface is the LAST front, dense over exactly the prescribed nodes
    if (nlast > 0) {
        const int nf = (int)S.f_c0.size();
        if (nf < 1 || S.f_part[nf - 1] != -1 || S.f_s[nf - 1] != nlast || S.f_n[nf - 1] != nlast || S.f_c0[nf - 1] != ns - nlast) return DSSS_E_STATE;
        for (int f = 0; f + 1 < nf; ++f) if (S.f_part[f] < 0) return DSSS_E_STATE;
        for (int q = 0; q < nlast; ++q) if (S.order[ns - nlast + q] != iface_last[q] || (int)S.iface_seps.size() != nlast || S.iface_seps[q] != iface_last[q]) return DSSS_E_STATE;
        if (!S.comm_kind.empty()) return DSSS_E_STATE;
    }
    const int rc = pg_host_solve(S, nedges, edges, aval, rhs, x);
    if (stats8) {
        stats8[0] = S.nnzL; stats8[1] = (int64_t)S.f_c0.size(); stats8[2] = S.npanels; stats8[3] = S.nlev;
        stats8[4] = S.front_doubles; stats8[5] = (int64_t)S.comm_vals.size(); stats8[6] = (int64_t)S.bincols.size(); stats8[7] = S.max_front_n;
    }
    return rc == 0 ? DSSS_OK : (rc == -1 ? DSSS_E_NUMERIC : DSSS_E_STATE);
}

// ------------------------------------------------------------------ one rank analysed by parts (dsss_pg_sym.h)
bool pg_symbolic_parts(int ns, const std::vector<std::pair<int, int>>& edges, int nchain, const double* cx, const double* cy,
                       const int* part, int K, int max_iface, const pg_sym_opts& opt, pg_sym& G)
{
    (void)nchain;
    const bool tv = getenv("DSSS_PG_VERBOSE") != nullptr;
    const auto q0 = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    // the interface: nodes with a neighbour in a higher part (every edge between two parts has its lower end here)
    std::vector<char> isif(ns, 0);
    for (const auto& e : edges) { const int pa = part[e.first], pb = part[e.second]; if (pa < pb) isif[e.first] = 1; else if (pb < pa) isif[e.second] = 1; }
    std::vector<int> I, iidx(ns, -1);
    for (int k = 0; k < ns; ++k) if (isif[k]) { iidx[k] = (int)I.size(); I.push_back(k); }
    const int nif = (int)I.size();
    if (nif > max_iface || K < 2) return false;
    struct part_t {
        pg_sym S; std::vector<int> glob_of, ledge_g; std::vector<std::pair<int, int>> ledges; std::vector<double> cx, cy; std::vector<int> loc_of;
        int nint = 0, nfi = 0;                                     // interior columns, interior fronts
        long long c0 = 0, nnz0 = 0, f0 = 0, b0 = 0, br0 = 0, ch0 = 0, rel0 = 0, fa0 = 0, fd0 = 0, fr0 = 0, ub0 = 0, bc0 = 0, rl0 = 0, frow0 = 0;
        long long nnz = 0, nchI = 0, nrelI = 0, nfaI = 0, nrows = 0;   // interior: factor blocks, children, rel entries, original entries, front rows
    };
    std::vector<part_t> P(K);
    std::once_flag coords_once;                                    // (the coordinates may still be on their way: the parts build their graphs first, ONE of them waits, the others behind it)
    const double t_pre = ms_since(q0);
    // ---- phase 1: every part on its own
    dsss_pool_run(K, [&](int p) {
        part_t& Q = P[p];
        Q.loc_of.assign(ns, -1);
        pg_sym_opts o = opt;
        o.threads = ns >= 131072 ? std::max(1, opt.threads / K) : 1;      // (a part of a C5-size graph is itself large enough for ranges; at C3's size forks cost what they gain)
        o.before_order = nullptr; o.on_bottom_ready = nullptr; o.on_lists_ready = nullptr; o.iface_plain = true; o.to_be_joined = true; o.iface_last.clear();
        for (int k = 0; k < ns; ++k)
            if (isif[k] || part[k] == p) { Q.loc_of[k] = (int)Q.glob_of.size(); if (isif[k]) o.iface_last.push_back(Q.loc_of[k]); Q.glob_of.push_back(k); }
        for (size_t g = 0; g < edges.size(); ++g) {
            const int a = Q.loc_of[edges[g].first], b = Q.loc_of[edges[g].second];
            if (a >= 0 && b >= 0 && a != b) { Q.ledges.push_back({ a, b }); Q.ledge_g.push_back((int)g); }
        }
        const int nsl = (int)Q.glob_of.size();
        Q.cx.resize(nsl); Q.cy.resize(nsl);
        std::call_once(coords_once, [&] { if (opt.before_order) opt.before_order(); });
        for (int i = 0; i < nsl; ++i) { Q.cx[i] = cx[Q.glob_of[i]]; Q.cy[i] = cy[Q.glob_of[i]]; }
        pg_symbolic(nsl, Q.ledges, 0, Q.cx.data(), Q.cy.data(), nullptr, 1, o, Q.S);
        Q.nint = nsl - nif;
    });
    const double t_parts = ms_since(q0);
    // ---- offsets of every index space
    long long c0 = 0, nnz0 = 0, f0 = 0, b0 = 0, br0 = 0, ch0 = 0, rel0 = 0, fa0 = 0, fd0 = 0, fr0 = 0, ub0 = 0, bc0 = 0, rl0 = 0, frow0 = 0;
    for (int p = 0; p < K; ++p) {
        part_t& Q = P[p]; const pg_sym& S = Q.S;
        const int nf = (int)S.f_c0.size();
        if (nif > 0 && (nf < 1 || S.f_part[nf - 1] != -1 || S.f_s[nf - 1] != nif || S.f_n[nf - 1] != nif || S.f_c0[nf - 1] != Q.nint)) return false;      // (the interface must be the last, dense front of every part)
        Q.nfi = nif > 0 ? nf - 1 : nf;
        Q.nnz = S.colptr[Q.nint];
        Q.nchI = S.ch_ptr[Q.nfi]; Q.nrelI = S.ch_relptr[Q.nchI]; Q.nfaI = S.fa_ptr[Q.nfi]; Q.nrows = S.f_rowptr[Q.nfi];
        Q.c0 = c0; Q.nnz0 = nnz0; Q.f0 = f0; Q.b0 = b0; Q.br0 = br0; Q.ch0 = ch0; Q.rel0 = rel0; Q.fa0 = fa0; Q.fd0 = fd0; Q.fr0 = fr0; Q.ub0 = ub0; Q.bc0 = bc0; Q.rl0 = rl0; Q.frow0 = frow0;
        c0 += Q.nint; nnz0 += Q.nnz; f0 += Q.nfi; b0 += (long long)S.binptr.size() - 1; br0 += (long long)S.broot.size(); ch0 += Q.nchI; rel0 += Q.nrelI; fa0 += Q.nfaI;
        fd0 += Q.nfi < nf ? S.f_off[Q.nfi] : S.front_doubles; fr0 += Q.nfi < nf ? S.f_roff[Q.nfi] : S.frhs_doubles; ub0 += S.ubin_doubles; bc0 += (long long)S.bincols.size();
        rl0 += S.rlptr.empty() ? 0 : S.rlptr[Q.nint]; frow0 += Q.nrows;
    }
    const int cI = (int)c0, nfG = (int)f0 + (nif > 0 ? 1 : 0), nbG = (int)b0, nbrG = (int)br0;
    const long long nnzI = (long long)nif * (nif + 1) / 2, nnzG = nnz0 + nnzI;
    // children / rel / original entries of the interface front: the parts' own, behind each other
    long long chI = 0, relI = 0;
    std::vector<long long> chI0(K, 0), relI0(K, 0);
    for (int p = 0; p < K; ++p) {
        const pg_sym& S = P[p].S;
        chI0[p] = chI; relI0[p] = relI;
        if (nif > 0) { chI += S.ch_ptr[P[p].nfi + 1] - S.ch_ptr[P[p].nfi]; relI += S.ch_relptr[S.ch_ptr[P[p].nfi + 1]] - S.ch_relptr[S.ch_ptr[P[p].nfi]]; }
    }
    const long long faI = nif > 0 ? P[0].S.fa_ptr[P[0].nfi + 1] - P[0].S.fa_ptr[P[0].nfi] : 0;
    const bool host_lists = !opt.lists_on_device;
    G = pg_sym();
    G.ns = ns; G.nparts = 1;
    G.nval = ns + (int)edges.size();
    G.perm.assign(ns, -1); G.order.assign(ns, -1);
    G.colptr.assign(ns + 1, 0); G.rowidx.resize(nnzG); G.parent.assign(ns, -1); G.col_part.assign(ns, 0);
    G.binned.assign(ns, 0); G.root_of.assign(ns, -1); G.broot_of_col.assign(ns, -1); G.front_of_col.assign(ns, -1);
    G.binptr.assign(nbG + 1, 0); G.bincols.resize(bc0); G.bin_part.assign(nbG, 0); G.bin_work.resize(nbG); G.bin_perm.resize(nbG);
    G.broot.resize(nbrG); G.broot_b.resize(nbrG); G.broot_uoff.resize(nbrG); G.binroot_ptr.assign(nbG + 1, 0); G.binroot_idx.resize(nbrG);
    G.ubin_doubles = ub0;
    if (host_lists) { G.rlptr.assign(ns + 1, 0); G.rlcol.resize(rl0); G.rlpos.resize(rl0); G.rlrow.resize(rl0); G.anc_first.assign(ns, 0); G.anc_rel.assign(nnzG, -1); }
    G.f_c0.resize(nfG); G.f_s.resize(nfG); G.f_n.resize(nfG); G.f_ld.resize(nfG); G.f_off.resize(nfG); G.f_roff.resize(nfG); G.f_parent.assign(nfG, -1); G.f_part.assign(nfG, 0);
    G.f_rowptr.assign(nfG + 1, 0); G.f_rows.resize(frow0 + nif);
    G.ch_ptr.assign(nfG + 1, 0); G.ch_kind.resize(ch0 + chI); G.ch_id.resize(ch0 + chI); G.ch_relptr.assign(ch0 + chI + 1, 0); G.rel.resize(rel0 + relI);
    G.fa_ptr.assign(nfG + 1, 0); G.fa_src.resize(fa0 + faI); G.fa_row.resize(fa0 + faI); G.fa_col.resize(fa0 + faI); G.fa_tr.resize(fa0 + faI);
    G.dest_bin.assign(G.nval, -1);
    // ---- phase 2: every part writes its share of the joined tables
    dsss_pool_run(K, [&](int p) {
        const part_t& Q = P[p]; const pg_sym& S = Q.S;
        const int nint = Q.nint;
        auto colmap = [&](int j) { return j < nint ? (int)Q.c0 + j : cI + (j - nint); };
        auto valmap = [&](int v) { const int nsl = (int)Q.glob_of.size(); return v < nsl ? Q.glob_of[v] : ns + Q.ledge_g[v - nsl]; };
        for (int j = 0; j < nint; ++j) {
            const int c = (int)Q.c0 + j, v = Q.glob_of[S.order[j]];
            G.order[c] = v; G.perm[v] = c;
            const int b = S.colptr[j], m = S.colptr[j + 1] - b;
            G.colptr[c + 1] = m;                                                  // (sizes now, the prefix sum below)
            int* out = G.rowidx.data() + Q.nnz0 + b;
            for (int q = 0; q < m; ++q) out[q] = colmap(S.rowidx[b + q]);
            G.parent[c] = S.parent[j] >= 0 ? colmap(S.parent[j]) : -1;
            G.binned[c] = S.binned[j];
            G.root_of[c] = S.root_of[j] >= 0 ? (int)Q.c0 + S.root_of[j] : -1;
            G.broot_of_col[c] = S.broot_of_col.empty() || S.broot_of_col[j] < 0 ? -1 : (int)Q.br0 + S.broot_of_col[j];
            G.front_of_col[c] = S.front_of_col[j] < 0 ? -1 : (S.front_of_col[j] >= Q.nfi ? nfG - 1 : (int)Q.f0 + S.front_of_col[j]);
            if (host_lists) {
                G.rlptr[c + 1] = S.rlptr[j + 1] - S.rlptr[j];
                for (int t = S.rlptr[j]; t < S.rlptr[j + 1]; ++t) { const long long at = Q.rl0 + t; G.rlcol[at] = (int)Q.c0 + S.rlcol[t]; G.rlpos[at] = (int)(Q.nnz0 + S.rlpos[t]); G.rlrow[at] = (int)Q.c0 + S.rlrow[t]; }
                G.anc_first[c] = S.anc_first[j];
                for (int q = 0; q < m; ++q) G.anc_rel[Q.nnz0 + b + q] = S.anc_rel[b + q];
            }
        }
        // bins and their roots
        const int nb = (int)S.binptr.size() - 1;
        for (int b = 0; b < nb; ++b) {
            G.binptr[Q.b0 + b + 1] = S.binptr[b + 1] - S.binptr[b];
            G.bin_work[Q.b0 + b] = S.bin_work[b];
            G.binroot_ptr[Q.b0 + b + 1] = S.binroot_ptr[b + 1] - S.binroot_ptr[b];
        }
        for (size_t i = 0; i < S.bincols.size(); ++i) G.bincols[Q.bc0 + i] = (int)Q.c0 + S.bincols[i];
        for (size_t i = 0; i < S.broot.size(); ++i) {
            G.broot[Q.br0 + i] = (int)Q.c0 + S.broot[i]; G.broot_b[Q.br0 + i] = S.broot_b[i]; G.broot_uoff[Q.br0 + i] = Q.ub0 + S.broot_uoff[i];
            G.binroot_idx[Q.br0 + i] = (int)Q.br0 + S.binroot_idx[i];
        }
        // interior fronts
        for (int f = 0; f < Q.nfi; ++f) {
            const int g = (int)Q.f0 + f;
            G.f_c0[g] = (int)Q.c0 + S.f_c0[f]; G.f_s[g] = S.f_s[f]; G.f_n[g] = S.f_n[f]; G.f_ld[g] = S.f_ld[f];
            G.f_off[g] = Q.fd0 + S.f_off[f]; G.f_roff[g] = Q.fr0 + S.f_roff[f];
            G.f_parent[g] = S.f_parent[f] < 0 ? -1 : (S.f_parent[f] >= Q.nfi ? nfG - 1 : (int)Q.f0 + S.f_parent[f]);
            G.f_rowptr[g + 1] = S.f_n[f];
            int* out = G.f_rows.data() + Q.frow0 + S.f_rowptr[f];
            for (int q = 0; q < S.f_n[f]; ++q) out[q] = colmap(S.f_rows[S.f_rowptr[f] + q]);
            G.ch_ptr[g + 1] = S.ch_ptr[f + 1] - S.ch_ptr[f];
            G.fa_ptr[g + 1] = S.fa_ptr[f + 1] - S.fa_ptr[f];
        }
        auto put_children = [&](int c_lo, int c_hi, long long at, long long rel_at) {
            for (int c = c_lo; c < c_hi; ++c, ++at) {
                G.ch_kind[at] = S.ch_kind[c]; G.ch_id[at] = S.ch_kind[c] ? (int)Q.br0 + S.ch_id[c] : (int)Q.f0 + S.ch_id[c];
                const long long r0 = S.ch_relptr[c], r1 = S.ch_relptr[c + 1];
                G.ch_relptr[at + 1] = r1 - r0;                                    // (sizes now, the prefix sum below)
                for (long long q = r0; q < r1; ++q) G.rel[rel_at + (q - r0)] = S.rel[q];
                rel_at += r1 - r0;
            }
        };
        put_children(0, (int)Q.nchI, Q.ch0, Q.rel0);
        if (nif > 0) put_children(S.ch_ptr[Q.nfi], S.ch_ptr[Q.nfi + 1], ch0 + chI0[p], rel0 + relI0[p]);
        auto put_entries = [&](int e_lo, int e_hi, long long at) {
            for (int e = e_lo; e < e_hi; ++e, ++at) { G.fa_src[at] = valmap(S.fa_src[e]); G.fa_row[at] = S.fa_row[e]; G.fa_col[at] = S.fa_col[e]; G.fa_tr[at] = S.fa_tr[e]; }
        };
        put_entries(0, (int)Q.nfaI, Q.fa0);
        if (nif > 0 && p == 0) put_entries(S.fa_ptr[Q.nfi], S.fa_ptr[Q.nfi + 1], fa0);      // (the interface's own values: every part lists them all, one copy counts)
        // destinations of the values whose column is one of this part's interior columns (the interface's own stay -1: front)
        for (int v = 0; v < (int)S.dest_bin.size(); ++v) if (S.dest_bin[v] >= 0) G.dest_bin[valmap(v)] = (int)(((Q.nnz0 + (S.dest_bin[v] >> 1)) << 1) | (S.dest_bin[v] & 1));
    });
    // ---- the interface columns and their front; prefix sums; orders
    for (int i = 0; i < nif; ++i) { const int c = cI + i; G.order[c] = I[i]; G.perm[I[i]] = c; G.colptr[c + 1] = nif - i; G.parent[c] = i + 1 < nif ? c + 1 : -1; G.front_of_col[c] = nfG - 1; }
    for (int c = 0; c < ns; ++c) G.colptr[c + 1] += G.colptr[c];
    { long long at = nnz0; for (int i = 0; i < nif; ++i) for (int r = i; r < nif; ++r) G.rowidx[at++] = cI + r; }
    G.nnzL = nnzG;
    for (int b = 0; b < nbG; ++b) { G.binptr[b + 1] += G.binptr[b]; G.binroot_ptr[b + 1] += G.binroot_ptr[b]; }
    if (host_lists) {
        for (int c = 0; c < ns; ++c) G.rlptr[c + 1] += G.rlptr[c];
        G.mapptr.assign(ns + 1, 0);
        for (int c = 0; c < ns; ++c) G.mapptr[c + 1] = G.mapptr[c] + (long long)(G.rlptr[c + 1] - G.rlptr[c]) * (long long)(G.colptr[c + 1] - G.colptr[c]);
    }
    for (int b = 0; b < nbG; ++b) G.bin_perm[b] = b;
    std::stable_sort(G.bin_perm.begin(), G.bin_perm.end(), [&](int x, int y) { return G.bin_work[x] > G.bin_work[y]; });
    if (nif > 0) {
        const int g = nfG - 1, ld = (6 * nif + 15) & ~15;
        G.f_c0[g] = cI; G.f_s[g] = nif; G.f_n[g] = nif; G.f_ld[g] = ld; G.f_off[g] = fd0; G.f_roff[g] = fr0; G.f_parent[g] = -1;
        G.f_rowptr[g + 1] = nif;
        for (int i = 0; i < nif; ++i) G.f_rows[frow0 + i] = cI + i;
        G.ch_ptr[g + 1] = (int)chI; G.fa_ptr[g + 1] = (int)faI;
        G.front_doubles = fd0 + (long long)ld * ld; G.frhs_doubles = fr0 + ld;
    } else { G.front_doubles = fd0; G.frhs_doubles = fr0; }
    for (int f = 0; f < nfG; ++f) { G.f_rowptr[f + 1] += G.f_rowptr[f]; G.ch_ptr[f + 1] += G.ch_ptr[f]; G.fa_ptr[f + 1] += G.fa_ptr[f]; }
    for (size_t c = 0; c + 1 < G.ch_relptr.size(); ++c) G.ch_relptr[c + 1] += G.ch_relptr[c];
    // statistics: the parts counted the interface columns once each
    double if_cols = 0, if_front = 0;
    for (int i = 0; i < nif; ++i) { const double m = nif - 1 - i; if_cols += 36.0 * 6.0 * (m * m + 3 * m) + 72.0; if_front += 216.0 * (m * m + 3 * m) + 72.0; }
    for (int p = 0; p < K; ++p) { G.flops_factor += P[p].S.flops_factor - if_cols; G.flops_fronts += P[p].S.flops_fronts - (nif > 0 ? if_front : 0); }
    G.flops_factor += if_cols; G.flops_fronts += nif > 0 ? if_front : 0;
    for (int f = 0; f < nfG; ++f) G.max_front_n = std::max(G.max_front_n, G.f_n[f]);
    const double t_join = ms_since(q0);
    if (opt.on_lists_ready) opt.on_lists_ready();
    if (opt.on_bottom_ready) opt.on_bottom_ready();
    sym_levels(G);
    sym_row_views(G, std::max(1, opt.threads));
    if (tv) { fprintf(stderr, "[dsss pg symbolic] separators per part:"); for (auto& q : P) fprintf(stderr, " %d", q.nint); fprintf(stderr, "\n"); }
    if (tv) fprintf(stderr, "[dsss pg symbolic] %d parts + an interface of %d: interface %.2f ms, the parts (their graphs, the wait for the coordinates, the analyses; largest %d separators) %.2f ms, joined %.2f ms, levels and row views %.2f ms | ns %d nnzL %lld bins %d fronts %d (max %d rows) panels %d levels %d front arena %.1f MB\n",
                    K, nif, t_pre, [&] { int m = 0; for (auto& q : P) m = std::max(m, (int)q.glob_of.size()); return m; }(), t_parts - t_pre, t_join - t_parts, ms_since(q0) - t_join, ns, G.nnzL,
                    nbG, nfG, G.max_front_n, G.npanels, G.nlev, G.front_doubles * 8e-6);
    return true;
}

// host twin entry: one rank analysed by K parts of equal size in the chain order (CPU test-suite)
extern "C" int dsss_host_pg_solve_parts(int ns, const int32_t* edge_a, const int32_t* edge_b, int nedges, const double* cx, const double* cy,
                                        int K, const double* aval, const double* rhs, double* x, int64_t* stats8)
{
    if (ns < 1 || nedges < ns - 1 || !edge_a || !edge_b || !cx || !cy || K < 2 || !x || !aval || !rhs) return DSSS_E_ARG;
    std::vector<std::pair<int, int>> edges(nedges);
    for (int e = 0; e < nedges; ++e) {
        edges[e] = { edge_a[e], edge_b[e] };
        if (edge_a[e] < 0 || edge_a[e] >= ns || edge_b[e] < 0 || edge_b[e] >= ns) return DSSS_E_ARG;
        if (e < ns - 1 && (edge_a[e] != e || edge_b[e] != e + 1)) return DSSS_E_ARG;
    }
    pg_sym S; pg_sym_opts opt;
    if (getenv("DSSS_PG_BIN_COST")) opt.bin_cost = atof(getenv("DSSS_PG_BIN_COST"));
    pg_sym_opts_env(opt);
    std::vector<int> part(ns);
    for (int k = 0; k < ns; ++k) part[k] = (int)((long long)k * K / ns);
    if (!pg_symbolic_parts(ns, edges, ns - 1, cx, cy, part.data(), K, ns, opt, S)) return DSSS_E_STATE;
    const int rc = pg_host_solve(S, nedges - (ns - 1), edges, aval, rhs, x);
    if (stats8) {
        stats8[0] = S.nnzL; stats8[1] = (int64_t)S.f_c0.size(); stats8[2] = S.npanels; stats8[3] = S.nlev;
        stats8[4] = S.front_doubles; stats8[5] = 0; stats8[6] = (int64_t)S.bincols.size(); stats8[7] = S.max_front_n;
    }
    return rc == 0 ? DSSS_OK : (rc == -1 ? DSSS_E_NUMERIC : DSSS_E_STATE);
}
