/* oracle/orc_orb.c -- ORB extractor restated from /root/reference/thirdparty/ORBextractor.cpp
 * (ORB/rBRIEF configuration, see orc.h for the documented deviations).  Test infrastructure.
 * OpenCV primitives (cv::resize, cv::FAST, fastAtan2, GaussianBlur) follow SURVEY.md Appendix A.1. */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PATCH_SIZE 31
#define HALF_PATCH 15
#define EDGE_THRESHOLD 19

static const int bit_pattern_31[256 * 4] = {
#include "orb_pattern_31.inc"
};

void orc_orb_params_default(orc_orb_params* p)
{ p->nfeatures = 2000; p->scale = 1.2f; p->nlevels = 6; p->ini_th = 12; p->min_th = 7; }

/* scale tables of the ctor (ORBextractor.cpp:415-431), all float */
static void scale_tables(const orc_orb_params* p, float* sf, float* inv)
{
    sf[0] = 1.0f;
    for (int i = 1; i < p->nlevels; ++i) sf[i] = sf[i - 1] * p->scale;
    for (int i = 0; i < p->nlevels; ++i) inv[i] = 1.0f / sf[i];
}

/* ComputePyramid level sizes (ORBextractor.cpp:1119-1120) */
void orc_orb_level_sizes(int rows, int cols, const orc_orb_params* p, int* lrows, int* lcols)
{
    float sf[ORC_MAX_LEVELS], inv[ORC_MAX_LEVELS];
    scale_tables(p, sf, inv);
    for (int l = 0; l < p->nlevels; ++l) {
        lcols[l] = orc_cvroundf((float)cols * inv[l]);
        lrows[l] = orc_cvroundf((float)rows * inv[l]);
    }
}

/* mnFeaturesPerLevel (ORBextractor.cpp:435-446) */
void orc_orb_level_quota(const orc_orb_params* p, int* quota)
{
    float factor = 1.0f / p->scale;
    float nDesired = p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)p->nlevels));
    int sum = 0;
    for (int l = 0; l < p->nlevels - 1; ++l) {
        quota[l] = orc_cvroundf(nDesired);
        sum += quota[l];
        nDesired *= factor;
    }
    int last = p->nfeatures - sum;
    quota[p->nlevels - 1] = last > 0 ? last : 0;
}

/* umax (ORBextractor.cpp:454-469) */
void orc_orb_umax(int* umax)
{
    int v, v0;
    int vmax = (int)floor(HALF_PATCH * sqrt(2.f) / 2 + 1);
    int vmin = (int)ceil(HALF_PATCH * sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH * HALF_PATCH;
    for (v = 0; v <= HALF_PATCH; ++v) umax[v] = 0;
    for (v = 0; v <= vmax; ++v) umax[v] = orc_cvround(sqrt(hp2 - v * v));
    for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

/* cv::resize(src, dst, INTER_LINEAR) for CV_8UC1 (called at ORBextractor.cpp:1128):
 * 11-bit fixed-point coefficients, horizontal pass in int, vertical pass with the >>4 / >>16 / +2>>2 form. */
void orc_resize_linear_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int* xofs = (int*)malloc(sizeof(int) * dw);
    short* ialpha = (short*)malloc(sizeof(short) * 2 * dw);
    int* yofs = (int*)malloc(sizeof(int) * dh);
    short* ibeta = (short*)malloc(sizeof(short) * 2 * dh);
    int xmax = dw;
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = orc_cvfloorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) { if (dx < xmax) xmax = dx; if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
        xofs[dx] = sx;
        float c0 = 1.f - fx, c1 = fx;
        int a0 = orc_cvroundf(c0 * 2048.f), a1 = orc_cvroundf(c1 * 2048.f);
        ialpha[dx * 2] = (short)a0; ialpha[dx * 2 + 1] = (short)a1;
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = orc_cvfloorf(fy);
        fy -= sy;
        yofs[dy] = sy;
        float c0 = 1.f - fy, c1 = fy;
        ibeta[dy * 2] = (short)orc_cvroundf(c0 * 2048.f);
        ibeta[dy * 2 + 1] = (short)orc_cvroundf(c1 * 2048.f);
    }
    int* r0 = (int*)malloc(sizeof(int) * dw);
    int* r1 = (int*)malloc(sizeof(int) * dw);
    for (int dy = 0; dy < dh; ++dy) {
        int sy0 = yofs[dy];
        int ya = sy0 < 0 ? 0 : (sy0 < sh ? sy0 : sh - 1);
        int yb = sy0 + 1 < 0 ? 0 : (sy0 + 1 < sh ? sy0 + 1 : sh - 1);
        const uint8_t* S0 = src + (size_t)ya * sw;
        const uint8_t* S1 = src + (size_t)yb * sw;
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx];
            if (dx < xmax) {
                r0[dx] = S0[sx] * ialpha[dx * 2] + S0[sx + 1] * ialpha[dx * 2 + 1];
                r1[dx] = S1[sx] * ialpha[dx * 2] + S1[sx + 1] * ialpha[dx * 2 + 1];
            } else {
                r0[dx] = S0[sx] * 2048;
                r1[dx] = S1[sx] * 2048;
            }
        }
        int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        for (int dx = 0; dx < dw; ++dx)
            dst[(size_t)dy * dw + dx] = (uint8_t)((((b0 * (r0[dx] >> 4)) >> 16) + ((b1 * (r1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(yofs); free(ibeta); free(r0); free(r1);
}

/* ---- cv::FAST TYPE_9_16 -------------------------------------------------------------------
 * Ring offsets (dx,dy) in OpenCV order.  For a pixel v the arc value
 *   A = max over the 16 arcs of 9 contiguous ring pixels of max( min(v - ring), min(ring - v) )
 * The pixel is a corner at threshold t iff A > t and cornerScore() returns A - 1. */
static const int ring_dx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
static const int ring_dy[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };

static int fast_arc_value(const uint8_t* p, int stride)
{
    int d[25];
    int v = p[0];
    for (int k = 0; k < 16; ++k) d[k] = v - (int)p[ring_dy[k] * stride + ring_dx[k]];
    for (int k = 16; k < 25; ++k) d[k] = d[k - 16];
    int best = 0;
    for (int k = 0; k < 16; ++k) {
        int mn = d[k], mx = d[k];
        for (int j = 1; j < 9; ++j) { if (d[k + j] < mn) mn = d[k + j]; if (d[k + j] > mx) mx = d[k + j]; }
        if (mn > best) best = mn;       /* all darker ring: v - ring > t */
        if (-mx > best) best = -mx;     /* all brighter ring: ring - v > t */
    }
    return best;
}

void orc_fast_arc_map(const uint8_t* img, int stride, int h, int w, int* A)
{
    memset(A, 0, sizeof(int) * (size_t)h * w);
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x)
            A[(size_t)y * w + x] = fast_arc_value(img + (size_t)y * stride + x, stride);
}

/* cv::FAST(window, kps, thr, true): corners with 3x3 non-max suppression on the score (strictly greater
 * than all 8 neighbours; pixels outside [3,dim-3) and non-corners score 0); row-major output order */
int orc_fast_window(const uint8_t* img, int stride, int h, int w, int thr, int* xs, int* ys, int* sc, int cap)
{
    if (h < 7 || w < 7) return 0;
    int* A = (int*)malloc(sizeof(int) * (size_t)h * w);
    orc_fast_arc_map(img, stride, h, w, A);
    int n = 0;
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            int a = A[(size_t)y * w + x];
            if (a <= thr) continue;
            int score = a - 1, keep = 1;
            for (int dy = -1; dy <= 1 && keep; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    if (!dx && !dy) continue;
                    int an = A[(size_t)(y + dy) * w + (x + dx)];   /* 0 outside the evaluated range */
                    int sn = an > thr ? an - 1 : 0;
                    if (!(score > sn)) { keep = 0; break; }
                }
            if (keep && n < cap) { xs[n] = x; ys[n] = y; sc[n] = score; ++n; }
        }
    free(A);
    return n;
}

/* per-level cell loop of ComputeKeyPointsOctTree (ORBextractor.cpp:771-829) */
int orc_fast_level(const uint8_t* img, int rows, int cols, int ini_th, int min_th,
                   float* xs, float* ys, float* resp, int cap)
{
    const float W = 30;
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = cols - EDGE_THRESHOLD + 3, maxBorderY = rows - EDGE_THRESHOLD + 3;
    const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols <= 0 || nRows <= 0) return 0;
    const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
    int n = 0;
    int cx[1600], cy[1600], cs[1600];
    for (int i = 0; i < nRows; ++i) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; ++j) {
            const float iniX = (float)(minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = (float)maxBorderX;
            int y0 = (int)iniY, y1 = (int)maxY, x0 = (int)iniX, x1 = (int)maxX;
            const uint8_t* win = img + (size_t)y0 * cols + x0;
            int k = orc_fast_window(win, cols, y1 - y0, x1 - x0, ini_th, cx, cy, cs, 1600);
            if (k == 0) k = orc_fast_window(win, cols, y1 - y0, x1 - x0, min_th, cx, cy, cs, 1600);
            for (int q = 0; q < k && n < cap; ++q) {
                xs[n] = (float)cx[q] + (float)(j * wCell);
                ys[n] = (float)cy[q] + (float)(i * hCell);
                resp[n] = (float)cs[q];
                ++n;
            }
        }
    }
    return n;
}

/* ---- DistributeOctTree (ORBextractor.cpp:481-763) ------------------------------------------
 * std::list<ExtractorNode> restated with an index-linked list; sort ties by creation order. */
typedef struct {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    int* keys; int nkeys;
    int noMore;
    int prev, next;
    int seq;
    int alive;
} qnode;

typedef struct {
    qnode* nodes; int nnodes, cap;
    int head, tail, size;
    int seq;
    const float *xs, *ys;
} qtree;

static int qt_new(qtree* t)
{
    if (t->nnodes == t->cap) { t->cap *= 2; t->nodes = (qnode*)realloc(t->nodes, sizeof(qnode) * t->cap); }
    qnode* n = &t->nodes[t->nnodes];
    memset(n, 0, sizeof *n);
    n->prev = n->next = -1; n->alive = 0; n->seq = t->seq++;
    return t->nnodes++;
}
static void qt_push_front(qtree* t, int id)
{
    qnode* n = &t->nodes[id];
    n->prev = -1; n->next = t->head; n->alive = 1;
    if (t->head >= 0) t->nodes[t->head].prev = id; else t->tail = id;
    t->head = id; t->size++;
}
static void qt_push_back(qtree* t, int id)
{
    qnode* n = &t->nodes[id];
    n->next = -1; n->prev = t->tail; n->alive = 1;
    if (t->tail >= 0) t->nodes[t->tail].next = id; else t->head = id;
    t->tail = id; t->size++;
}
static int qt_erase(qtree* t, int id)   /* returns next */
{
    qnode* n = &t->nodes[id];
    int nx = n->next;
    if (n->prev >= 0) t->nodes[n->prev].next = n->next; else t->head = n->next;
    if (n->next >= 0) t->nodes[n->next].prev = n->prev; else t->tail = n->prev;
    n->alive = 0; t->size--;
    free(n->keys); n->keys = NULL;
    return nx;
}

/* ExtractorNode::DivideNode (ORBextractor.cpp:481-537); children ids in c[4] (n1..n4) */
static void qt_divide(qtree* t, int id, int c[4])
{
    for (int k = 0; k < 4; ++k) c[k] = qt_new(t);
    qnode* P = &t->nodes[id];
    qnode *n1 = &t->nodes[c[0]], *n2 = &t->nodes[c[1]], *n3 = &t->nodes[c[2]], *n4 = &t->nodes[c[3]];
    const int halfX = (int)ceilf((float)(P->URx - P->ULx) / 2);
    const int halfY = (int)ceilf((float)(P->BRy - P->ULy) / 2);
    n1->ULx = P->ULx; n1->ULy = P->ULy;
    n1->URx = P->ULx + halfX; n1->URy = P->ULy;
    n1->BLx = P->ULx; n1->BLy = P->ULy + halfY;
    n1->BRx = P->ULx + halfX; n1->BRy = P->ULy + halfY;
    n2->ULx = n1->URx; n2->ULy = n1->URy;
    n2->URx = P->URx; n2->URy = P->URy;
    n2->BLx = n1->BRx; n2->BLy = n1->BRy;
    n2->BRx = P->URx; n2->BRy = P->ULy + halfY;
    n3->ULx = n1->BLx; n3->ULy = n1->BLy;
    n3->URx = n1->BRx; n3->URy = n1->BRy;
    n3->BLx = P->BLx; n3->BLy = P->BLy;
    n3->BRx = n1->BRx; n3->BRy = P->BLy;
    n4->ULx = n3->URx; n4->ULy = n3->URy;
    n4->URx = n2->BRx; n4->URy = n2->BRy;
    n4->BLx = n3->BRx; n4->BLy = n3->BRy;
    n4->BRx = P->BRx; n4->BRy = P->BRy;
    for (int k = 0; k < 4; ++k) { t->nodes[c[k]].keys = (int*)malloc(sizeof(int) * (P->nkeys > 0 ? P->nkeys : 1)); t->nodes[c[k]].nkeys = 0; }
    for (int i = 0; i < P->nkeys; ++i) {
        int key = P->keys[i];
        float kx = t->xs[key], ky = t->ys[key];
        qnode* dst;
        if (kx < (float)n1->URx) dst = (ky < (float)n1->BRy) ? n1 : n3;
        else if (ky < (float)n1->BRy) dst = n2;
        else dst = n4;
        dst->keys[dst->nkeys++] = key;
    }
    for (int k = 0; k < 4; ++k) if (t->nodes[c[k]].nkeys == 1) t->nodes[c[k]].noMore = 1;
}

typedef struct { int size, seq, id; } qsz;
static int qsz_cmp(const void* a, const void* b)
{
    const qsz *x = (const qsz*)a, *y = (const qsz*)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);
}

/* push the non-empty children to the list front in n1..n4 order (ORBextractor.cpp:621-660 / :691-726) */
static void qt_add_children(qtree* t, const int c[4], qsz** vec, int* nvec, int* capvec, int* nToExpand)
{
    for (int k = 0; k < 4; ++k) {
        qnode* n = &t->nodes[c[k]];
        if (n->nkeys > 0) {
            qt_push_front(t, c[k]);
            if (n->nkeys > 1) {
                if (nToExpand) (*nToExpand)++;
                if (*nvec == *capvec) { *capvec *= 2; *vec = (qsz*)realloc(*vec, sizeof(qsz) * *capvec); }
                (*vec)[*nvec].size = n->nkeys; (*vec)[*nvec].seq = n->seq; (*vec)[*nvec].id = c[k]; (*nvec)++;
            }
        } else { free(n->keys); n->keys = NULL; }
    }
}

int orc_quadtree(const float* xs, const float* ys, const float* resp, int n,
                 int minX, int maxX, int minY, int maxY, int N, int* keep_idx)
{
    if (n == 0) return 0;
    qtree T; T.cap = 1024; T.nodes = (qnode*)malloc(sizeof(qnode) * T.cap); T.nnodes = 0;
    T.head = T.tail = -1; T.size = 0; T.seq = 0; T.xs = xs; T.ys = ys;

    int nIni = (int)roundf((float)(maxX - minX) / (float)(maxY - minY));
    if (nIni < 1) nIni = 1;                                   /* deviation: reference divides by zero here */
    const float hX = (float)(maxX - minX) / nIni;
    int* ini = (int*)malloc(sizeof(int) * nIni);
    for (int i = 0; i < nIni; ++i) {
        int id = qt_new(&T);
        qnode* q = &T.nodes[id];
        q->ULx = (int)(hX * (float)i); q->ULy = 0;
        q->URx = (int)(hX * (float)(i + 1)); q->URy = 0;
        q->BLx = q->ULx; q->BLy = maxY - minY;
        q->BRx = q->URx; q->BRy = maxY - minY;
        q->keys = (int*)malloc(sizeof(int) * n); q->nkeys = 0;
        qt_push_back(&T, id);
        ini[i] = id;
    }
    for (int i = 0; i < n; ++i) {
        int which = (int)(xs[i] / hX);
        if (which >= nIni) which = nIni - 1;
        qnode* q = &T.nodes[ini[which]];
        q->keys[q->nkeys++] = i;
    }
    free(ini);
    for (int it = T.head; it >= 0;) {
        qnode* q = &T.nodes[it];
        if (q->nkeys == 1) { q->noMore = 1; it = q->next; }
        else if (q->nkeys == 0) it = qt_erase(&T, it);
        else it = q->next;
    }

    int bFinish = 0;
    int capvec = 256, nvec = 0;
    qsz* vec = (qsz*)malloc(sizeof(qsz) * capvec);
    while (!bFinish) {
        int prevSize = T.size;
        int nToExpand = 0;
        nvec = 0;
        for (int it = T.head; it >= 0;) {
            qnode* q = &T.nodes[it];
            if (q->noMore) { it = q->next; continue; }
            int c[4];
            qt_divide(&T, it, c);
            qt_add_children(&T, c, &vec, &nvec, &capvec, &nToExpand);
            it = qt_erase(&T, it);
        }
        if (T.size >= N || T.size == prevSize) bFinish = 1;
        else if (T.size + nToExpand * 3 > N) {
            while (!bFinish) {
                prevSize = T.size;
                int nprev = nvec;
                qsz* prev = (qsz*)malloc(sizeof(qsz) * (nprev > 0 ? nprev : 1));
                memcpy(prev, vec, sizeof(qsz) * nprev);
                nvec = 0;
                qsort(prev, nprev, sizeof(qsz), qsz_cmp);
                for (int j = nprev - 1; j >= 0; --j) {
                    int c[4];
                    qt_divide(&T, prev[j].id, c);
                    qt_add_children(&T, c, &vec, &nvec, &capvec, NULL);
                    qt_erase(&T, prev[j].id);
                    if (T.size >= N) break;
                }
                free(prev);
                if (T.size >= N || T.size == prevSize) bFinish = 1;
            }
        }
    }
    free(vec);
    /* retain the best point of each node, list order (ORBextractor.cpp:741-760) */
    int nk = 0;
    for (int it = T.head; it >= 0; it = T.nodes[it].next) {
        qnode* q = &T.nodes[it];
        int best = q->keys[0];
        float maxR = resp[best];
        for (int k = 1; k < q->nkeys; ++k)
            if (resp[q->keys[k]] > maxR) { best = q->keys[k]; maxR = resp[best]; }
        keep_idx[nk++] = best;
    }
    for (int i = 0; i < T.nnodes; ++i) free(T.nodes[i].keys);
    free(T.nodes);
    return nk;
}

/* IC_Angle (ORBextractor.cpp:77-104) */
float orc_ic_angle(const uint8_t* img, int stride, int x, int y)
{
    int umax[HALF_PATCH + 2];
    orc_orb_umax(umax);
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)y * stride + x;
    for (int u = -HALF_PATCH; u <= HALF_PATCH; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* 13-tap sigma-2 Gaussian in 8.8 fixed point, by the rule OpenCV >= 4.5.2 uses for its bit-exact 8-bit GaussianBlur
 * (imgproc smooth.dispatch.cpp, getGaussianKernelFixedPoint_ED, restated from its published source: OpenCV is not in this
 * image): the normalised taps times 256 are rounded from the edge inwards, every rounding error carried into the next tap,
 * and the centre tap takes what is left of 256.  -> 1 2 7 16 31 45 52 45 31 16 7 2 1.  (Rounds 1 and 2 of this project
 * rounded every tap on its own and folded the residue into the centre: 17 and 50 where this has 16 and 52.) */
void orc_gauss13_taps(int taps[13])
{
    double g[13], s = 0;
    for (int i = 0; i < 13; ++i) { double d = i - 6; g[i] = exp(-(d * d) / 8.0); s += g[i]; }
    double err = 0;
    int tot = 0;
    for (int i = 0; i < 6; ++i) {
        const double adj = g[i] / s * 256.0 + err;
        const int v0 = (int)lrint(adj);                    /* cvRound */
        err = adj - v0;
        taps[i] = taps[12 - i] = v0; tot += v0;
    }
    taps[6] = 256 - 2 * tot;
}

static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

/* GaussianBlur(13x13, sigma 2, BORDER_REFLECT_101) on a level clone (ORBextractor.cpp:1091-1092):
 * horizontal 8.8 pass (u16), vertical pass, (v + 2^15) >> 16 */
void orc_blur13(const uint8_t* src, int rows, int cols, uint8_t* dst)
{
    int taps[13];
    orc_gauss13_taps(taps);
    uint16_t* tmp = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)rows * cols);
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            int acc = 0;
            for (int k = 0; k < 13; ++k) acc += taps[k] * src[(size_t)y * cols + reflect101(x + k - 6, cols)];
            tmp[(size_t)y * cols + x] = (uint16_t)acc;
        }
    for (int y = 0; y < rows; ++y)
        for (int x = 0; x < cols; ++x) {
            unsigned acc = 0;
            for (int k = 0; k < 13; ++k) acc += (unsigned)taps[k] * tmp[(size_t)reflect101(y + k - 6, rows) * cols + x];
            dst[(size_t)y * cols + x] = (uint8_t)((acc + 32768u) >> 16);
        }
    free(tmp);
}

/* computeOrbDescriptor (ORBextractor.cpp:108-147) */
void orc_brief(const uint8_t* img, int stride, int x, int y, float angle_deg, uint8_t desc[32])
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = angle_deg * factorPI;
    double sd, cd;
    orc_sincos((double)angle, &sd, &cd);
    float a = (float)cd, b = (float)sd;
    const uint8_t* center = img + (size_t)y * stride + x;
    const int* pat = bit_pattern_31;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int k = 0; k < 8; ++k) {
            int x0 = pat[4 * k], y0 = pat[4 * k + 1], x1 = pat[4 * k + 2], y1 = pat[4 * k + 3];
            int t0 = center[orc_cvroundf((float)x0 * b + (float)y0 * a) * stride + orc_cvroundf((float)x0 * a - (float)y0 * b)];
            int t1 = center[orc_cvroundf((float)x1 * b + (float)y1 * a) * stride + orc_cvroundf((float)x1 * a - (float)y1 * b)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ORBextractor::operator() with the ORB descriptor call (ORBextractor.cpp:1049-1113) */
static int orb_extract_impl(const uint8_t* img, int rows, int cols, const orc_orb_params* p,
                            orc_kp* kps, uint8_t* desc, uint8_t* desc128, int cap);
int orc_orb_extract(const uint8_t* img, int rows, int cols, const orc_orb_params* p,
                    orc_kp* kps, uint8_t* desc, int cap)
{
    return orb_extract_impl(img, rows, cols, p, kps, desc, NULL, cap);
}
/* the same with the SIFT call site's descriptor next to the ORB one (ORBextractor.cpp:1098 as intended, oracle/orc_sift.c):
 * desc128 = n x 128 integer-valued bytes (the 128 floats of a cv::SIFT row) */
int orc_orb_extract_sift(const uint8_t* img, int rows, int cols, const orc_orb_params* p,
                         orc_kp* kps, uint8_t* desc, uint8_t* desc128, int cap)
{
    return orb_extract_impl(img, rows, cols, p, kps, desc, desc128, cap);
}
static int orb_extract_impl(const uint8_t* img, int rows, int cols, const orc_orb_params* p,
                            orc_kp* kps, uint8_t* desc, uint8_t* desc128, int cap)
{
    int L = p->nlevels;
    int lrows[ORC_MAX_LEVELS], lcols[ORC_MAX_LEVELS], quota[ORC_MAX_LEVELS];
    float sf[ORC_MAX_LEVELS], inv[ORC_MAX_LEVELS];
    scale_tables(p, sf, inv);
    orc_orb_level_sizes(rows, cols, p, lrows, lcols);
    orc_orb_level_quota(p, quota);
    uint8_t* pyr[ORC_MAX_LEVELS];
    pyr[0] = (uint8_t*)malloc((size_t)rows * cols);
    memcpy(pyr[0], img, (size_t)rows * cols);
    for (int l = 1; l < L; ++l) {
        pyr[l] = (uint8_t*)malloc((size_t)lrows[l] * lcols[l]);
        orc_resize_linear_u8(pyr[l - 1], lrows[l - 1], lcols[l - 1], pyr[l], lrows[l], lcols[l]);
    }
    int total = 0;
    int capc = rows * cols / 4 + 1024;      /* strict 3x3 local maxima cannot exceed one pixel in four */
    float* cx = (float*)malloc(sizeof(float) * capc);
    float* cy = (float*)malloc(sizeof(float) * capc);
    float* cr = (float*)malloc(sizeof(float) * capc);
    int* keep = (int*)malloc(sizeof(int) * capc);
    for (int l = 0; l < L; ++l) {
        int r = lrows[l], c = lcols[l];
        const int minB = EDGE_THRESHOLD - 3;
        int ncand = orc_fast_level(pyr[l], r, c, p->ini_th, p->min_th, cx, cy, cr, capc);
        int nk = orc_quadtree(cx, cy, cr, ncand, minB, c - EDGE_THRESHOLD + 3, minB, r - EDGE_THRESHOLD + 3, quota[l], keep);
        if (nk == 0) continue;
        uint8_t* blur = (uint8_t*)malloc((size_t)r * c);
        orc_blur13(pyr[l], r, c, blur);
        const int scaledPatch = (int)(PATCH_SIZE * sf[l]);
        for (int k = 0; k < nk && total < cap; ++k) {
            orc_kp kp;
            kp.x = cx[keep[k]] + minB; kp.y = cy[keep[k]] + minB;
            kp.response = cr[keep[k]]; kp.octave = l; kp.size = (float)scaledPatch;
            int xi = orc_cvroundf(kp.x), yi = orc_cvroundf(kp.y);
            kp.angle = orc_ic_angle(pyr[l], c, xi, yi);
            orc_brief(blur, c, xi, yi, kp.angle, desc + (size_t)total * 32);
            if (desc128) orc_sift128(blur, r, c, xi, yi, kp.angle, desc128 + (size_t)total * 128);
            if (l != 0) { kp.x *= sf[l]; kp.y *= sf[l]; }
            kps[total++] = kp;
        }
        free(blur);
    }
    for (int l = 0; l < L; ++l) free(pyr[l]);
    free(cx); free(cy); free(cr); free(keep);
    return total;
}

/* Frame::DetectFeature tail (frame.cpp:184-195): keep kp iff mask(int(y), int(x)) != 0 */
int orc_mask_filter(orc_kp* kps, uint8_t* desc, int n, const uint8_t* mask, int cols)
{
    return orc_mask_filter2(kps, desc, NULL, n, mask, cols);
}
int orc_mask_filter2(orc_kp* kps, uint8_t* desc, uint8_t* desc128, int n, const uint8_t* mask, int cols)
{
    int m = 0;
    for (int i = 0; i < n; ++i) {
        int v = (int)kps[i].y, u = (int)kps[i].x;
        if (mask[(size_t)v * cols + u]) {
            if (m != i && desc128) memmove(desc128 + (size_t)m * 128, desc128 + (size_t)i * 128, 128);
            if (m != i) { kps[m] = kps[i]; memmove(desc + (size_t)m * 32, desc + (size_t)i * 32, 32); }
            ++m;
        }
    }
    return m;
}
