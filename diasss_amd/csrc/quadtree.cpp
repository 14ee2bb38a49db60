// diasss_amd/csrc/quadtree.cpp -- HOST twin of the device quadtree (dsss_quadtree.hip): spatial culling of FAST
// candidates to the per-level quota.  Restates ORBextractor::DistributeOctTree / ExtractorNode::DivideNode
// (/root/reference/thirdparty/ORBextractor.cpp:481-763).  The hot path never calls it: extraction runs
// quadtree_kernel on the device.  It is exported (dsss_host_quadtree) only so that the CPU test-suite can pin the
// list-order semantics (children pushed to the FRONT of a list that is being walked; largest nodes split first once
// the quota is near) against the oracle without a GPU.
// Two places where the reference is undefined are pinned down exactly as in oracle/orc_orb.c:
//   - nIni = max(1, round(width/height))      (the reference divides by zero for tall levels, :543-545)
//   - equal-size nodes are split in creation order (the reference compares heap addresses, :684)
#include "dsss_internal.h"
#include <algorithm>
#include <list>

namespace {

struct Cell {
    int x0, y0, x1, y1;            // [x0,x1) x [y0,y1) in candidate coordinates
    std::vector<int> pts;          // candidate indices, original order preserved
    bool leaf = false;             // holds exactly one point: never split again
    int serial = 0;                // creation order
};

using CellList = std::list<Cell>;
struct Pending { int size; int serial; CellList::iterator it; };

struct Splitter {
    const float* xs; const float* ys;
    CellList cells;
    int next_serial = 0;

    // four children of *it in the reference's n1..n4 order: top-left, top-right, bottom-left, bottom-right
    void split(CellList::iterator it, Cell out[4]) const
    {
        const Cell& c = *it;
        const int hx = (int)std::ceil((float)(c.x1 - c.x0) / 2), hy = (int)std::ceil((float)(c.y1 - c.y0) / 2);
        const int mx = c.x0 + hx, my = c.y0 + hy;
        out[0].x0 = c.x0; out[0].y0 = c.y0; out[0].x1 = mx;   out[0].y1 = my;
        out[1].x0 = mx;   out[1].y0 = c.y0; out[1].x1 = c.x1; out[1].y1 = my;
        out[2].x0 = c.x0; out[2].y0 = my;   out[2].x1 = mx;   out[2].y1 = c.y1;
        out[3].x0 = mx;   out[3].y0 = my;   out[3].x1 = c.x1; out[3].y1 = c.y1;
        for (int k = 0; k < 4; ++k) { out[k].pts.clear(); out[k].pts.reserve(c.pts.size()); out[k].leaf = false; }
        for (int id : c.pts) {
            const bool left = xs[id] < (float)mx, top = ys[id] < (float)my;
            out[left ? (top ? 0 : 2) : (top ? 1 : 3)].pts.push_back(id);
        }
        for (int k = 0; k < 4; ++k) out[k].leaf = out[k].pts.size() == 1;
    }

    // replace *it by its non-empty children (pushed to the list front); children that can still be split are
    // recorded in `todo`
    void expand(CellList::iterator it, std::vector<Pending>& todo, int* splittable)
    {
        Cell kids[4];
        split(it, kids);
        for (int k = 0; k < 4; ++k) {
            if (kids[k].pts.empty()) continue;
            kids[k].serial = next_serial++;
            cells.push_front(std::move(kids[k]));
            if (cells.front().pts.size() > 1) {
                if (splittable) ++*splittable;
                todo.push_back({ (int)cells.front().pts.size(), cells.front().serial, cells.begin() });
            }
        }
        cells.erase(it);
    }
};

} // namespace

// returns the kept candidate indices in the reference's output order (list order, best response per node)
int dsss_quadtree_cull(const float* xs, const float* ys, const float* resp, int n,
                       int minX, int maxX, int minY, int maxY, int quota, std::vector<int>& keep)
{
    keep.clear();
    if (n <= 0) return 0;
    Splitter S; S.xs = xs; S.ys = ys;
    int nroot = (int)std::round((float)(maxX - minX) / (float)(maxY - minY));
    if (nroot < 1) nroot = 1;
    const float hX = (float)(maxX - minX) / nroot;
    std::vector<CellList::iterator> roots(nroot);
    for (int i = 0; i < nroot; ++i) {
        Cell c;
        c.x0 = (int)(hX * (float)i); c.x1 = (int)(hX * (float)(i + 1)); c.y0 = 0; c.y1 = maxY - minY;
        c.serial = S.next_serial++;
        S.cells.push_back(std::move(c));
        roots[i] = std::prev(S.cells.end());
    }
    for (int i = 0; i < n; ++i) {
        int r = (int)(xs[i] / hX);
        if (r >= nroot) r = nroot - 1;
        roots[r]->pts.push_back(i);
    }
    for (auto it = S.cells.begin(); it != S.cells.end();) {
        if (it->pts.size() == 1) { it->leaf = true; ++it; }
        else if (it->pts.empty()) it = S.cells.erase(it);
        else ++it;
    }
    std::vector<Pending> todo;
    bool done = false;
    while (!done) {
        const int before = (int)S.cells.size();
        int splittable = 0;
        todo.clear();
        for (auto it = S.cells.begin(); it != S.cells.end();) {
            if (it->leaf) { ++it; continue; }
            auto nxt = std::next(it);
            S.expand(it, todo, &splittable);
            it = nxt;
        }
        const int now = (int)S.cells.size();
        if (now >= quota || now == before) done = true;
        else if (now + splittable * 3 > quota) {
            // close to the quota: split the most populated nodes first, stop as soon as the quota is reached
            while (!done) {
                const int before2 = (int)S.cells.size();
                std::vector<Pending> cur;
                cur.swap(todo);
                std::sort(cur.begin(), cur.end(), [](const Pending& a, const Pending& b) {
                    return a.size != b.size ? a.size < b.size : a.serial < b.serial; });
                for (int j = (int)cur.size() - 1; j >= 0; --j) {
                    S.expand(cur[j].it, todo, nullptr);
                    if ((int)S.cells.size() >= quota) break;
                }
                if ((int)S.cells.size() >= quota || (int)S.cells.size() == before2) done = true;
            }
        }
    }
    for (const Cell& c : S.cells) {
        int best = c.pts[0];
        float r = resp[best];
        for (size_t k = 1; k < c.pts.size(); ++k) if (resp[c.pts[k]] > r) { best = c.pts[k]; r = resp[best]; }
        keep.push_back(best);
    }
    return (int)keep.size();
}
