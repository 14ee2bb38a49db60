/* oracle/orc_match.c -- FEAmatcher restated from /root/reference/src/core/FEAmatcher.cpp.
 * Test infrastructure, see orc.h for the documented deviations. */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_match_params_default(orc_match_params* p)
{
    p->use_l2 = 0; p->radius = 8; p->bound_same = 88; p->bound_diff = 80; p->l2_bound = 350;
    p->ratio = 0.35; p->scc_iters = 1000; p->pix_err = 2.5; p->merge_thr = 2.5;
}

/* first stage of GeoNearNeighSearch (FEAmatcher.cpp:79-183) */
int orc_match_nn(int id, int id_ref, const orc_kp* kps, const uint8_t* desc, const double* geo, int na,
                 const orc_kp* kps_ref, const uint8_t* desc_ref, const double* geo_ref, int nb,
                 const double bbox_ref[4], const orc_match_params* p, int32_t* corres,
                 int32_t* best_d, int32_t* second_d, int32_t* ncand)
{
    (void)kps; (void)kps_ref;
    int accepted = 0;
    double bx_min = bbox_ref[0], bx_max = bbox_ref[1], by_min = bbox_ref[2], by_max = bbox_ref[3];
    for (int i = 0; i < na; ++i) {
        corres[i] = -1;
        if (best_d) best_d[i] = 1000;
        if (second_d) second_d[i] = 1000;
        if (ncand) ncand[i] = 0;
        double loc_x = geo[2 * i], loc_y = geo[2 * i + 1];
        if (loc_x < bx_min || loc_y < by_min || loc_x > bx_max || loc_y > by_max) continue;
        int nc = 0, best_id = -1;
        if (!p->use_l2) {
            /* Hamming branch (:141-176) */
            int best = 1000, second = 1000, bound = p->bound_same;
            if (id % 2 != id_ref % 2) bound = p->bound_diff;
            for (int j = 0; j < nb; ++j) {
                double rx = geo_ref[2 * j], ry = geo_ref[2 * j + 1];
                double gd = sqrt((loc_x - rx) * (loc_x - rx) + (loc_y - ry) * (loc_y - ry));
                if (!(gd < p->radius)) continue;
                ++nc;
                int d = orc_hamming256(desc + (size_t)i * 32, desc_ref + (size_t)j * 32);
                if (d < best) { second = best; best = d; best_id = j; }
                else if (d < second) second = d;
            }
            if (best_d) best_d[i] = best;
            if (second_d) second_d[i] = second;
            if (ncand) ncand[i] = nc;
            if (!nc) continue;
            double ratio = (double)best / second;
            if (best_id != -1 && best <= bound && ratio <= p->ratio && second != 1000) { corres[i] = best_id; ++accepted; }
            else if (nc == 1 && best <= bound) { corres[i] = best_id; ++accepted; }
        } else {
            /* L2 branch (:106-139); rows of integer-valued elements: distances are sqrt of integers */
            const int dlen = p->use_l2 == 2 ? 128 : 32;
            double best = 1000, second = 1000;
            int bi = 1000000, si = 1000000;
            for (int j = 0; j < nb; ++j) {
                double rx = geo_ref[2 * j], ry = geo_ref[2 * j + 1];
                double gd = sqrt((loc_x - rx) * (loc_x - rx) + (loc_y - ry) * (loc_y - ry));
                if (!(gd < p->radius)) continue;
                ++nc;
                int sq = orc_l2sq(desc + (size_t)i * dlen, desc_ref + (size_t)j * dlen, dlen);
                double d = sqrt((double)sq);
                if (d < best) { second = best; si = bi; best = d; bi = sq; best_id = j; }
                else if (d < second) { second = d; si = sq; }
            }
            if (best_d) best_d[i] = bi;
            if (second_d) second_d[i] = si;
            if (ncand) ncand[i] = nc;
            if (!nc) continue;
            double ratio = best / second;
            if (best_id != -1 && best < p->l2_bound && ratio <= p->ratio) { corres[i] = best_id; ++accepted; }
            else if (nc == 1 && best < p->l2_bound) { corres[i] = best_id; ++accepted; }
        }
    }
    return accepted;
}

/* SCC_x (FEAmatcher.cpp:186-248).  The local cv::RNG (:59) is default constructed on every call. */
int orc_match_scc(int id, int id_ref, int rows_ref, const orc_kp* kps, int na, const orc_kp* kps_ref,
                  const orc_match_params* p, int32_t* corres, int* scc_count, double* scc_model)
{
    *scc_count = 0; *scc_model = 0;
    int nloc = 0;
    int* id_loc = (int*)malloc(sizeof(int) * (na > 0 ? na : 1));
    for (int i = 0; i < na; ++i) if (corres[i] != -1) id_loc[nloc++] = i;
    if (nloc == 0) { free(id_loc); return 0; }          /* deviation: reference indexes ID_loc[0] of an empty vector */
    uint64_t rng = 0xffffffffu;
    int flip = (id % 2 != id_ref % 2);
    int final_inl = 0, hist = 0;
    int32_t* fin = (int32_t*)malloc(sizeof(int32_t) * na);
    int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * na);
    for (int i = 0; i < na; ++i) fin[i] = -1;
    for (int it = 0; it < p->scc_iters; ++it) {
        int s[2];
        for (int k = 0; k < 2; ++k) s[k] = id_loc[orc_cvrng_uniform(&rng, 0, nloc)];
        double ModelX = 0;
        for (int k = 0; k < 2; ++k) {
            float ya = kps[s[k]].y, yb = kps_ref[corres[s[k]]].y;
            if (flip) ModelX = ModelX + fabsf(ya - ((float)rows_ref - yb + 1));
            else ModelX = ModelX + fabsf(ya - yb);
        }
        ModelX = ModelX / 2;
        int cnt = 0;
        for (int j = 0; j < na; ++j) {
            cur[j] = -1;
            if (corres[j] == -1) continue;
            float ya = kps[j].y, yb = kps_ref[corres[j]].y;
            double X_tmp = flip ? (double)fabsf(ya - ((float)rows_ref - yb + 1)) : (double)fabsf(ya - yb);
            if (fabs(ModelX - X_tmp) <= p->pix_err) { cur[j] = corres[j]; ++cnt; }
        }
        if (final_inl < cnt) {
            memcpy(fin, cur, sizeof(int32_t) * na);
            final_inl = cnt;
            *scc_count = cnt; *scc_model = ModelX;   /* == scc.rbegin-sorted [0]: counts strictly increase */
            ++hist;
        }
    }
    memcpy(corres, fin, sizeof(int32_t) * na);
    free(fin); free(cur); free(id_loc);
    return hist;
}

int orc_match_dir(int id, int id_ref, int rows_ref, const orc_kp* kps, const uint8_t* desc, const double* geo, int na,
                  const orc_kp* kps_ref, const uint8_t* desc_ref, const double* geo_ref, int nb,
                  const double bbox_ref[4], const orc_match_params* p, int32_t* corres,
                  int* scc_count, double* scc_model)
{
    orc_match_nn(id, id_ref, kps, desc, geo, na, kps_ref, desc_ref, geo_ref, nb, bbox_ref, p, corres, NULL, NULL, NULL);
    return orc_match_scc(id, id_ref, rows_ref, kps, na, kps_ref, p, corres, scc_count, scc_model);
}

/* ConsistentCheck (FEAmatcher.cpp:323-405) on index pairs */
int orc_consistent_check(int id_s, int id_t, int rows_s, int rows_t,
                         const int32_t* c1, int n1, const int32_t* c2, int n2,
                         int hist1, int cnt1, double model1, int hist2, int cnt2, double model2,
                         const orc_match_params* p, int32_t* src_idx, int32_t* tgt_idx)
{
    (void)cnt1; (void)cnt2;
    int count = 0;
    double img_diff = 0;
    if (id_s % 2 != id_t % 2) img_diff = abs(rows_s - rows_t);
    int merge = 0;
    if (hist1 > 0 && hist2 > 0) {          /* deviation: scc_x[0] of an empty vector is UB in the reference */
        double kp_diff = fabs(fabs(model1 - model2) - img_diff);
        merge = kp_diff <= p->merge_thr;
    }
    if (merge) {
        for (int i = 0; i < n1; ++i) {
            if (c1[i] == -1) continue;
            if (c2[c1[i]] == i) continue;
            src_idx[count] = i; tgt_idx[count] = c1[i]; ++count;
        }
        for (int i = 0; i < n2; ++i) {
            if (c2[i] == -1) continue;
            src_idx[count] = c2[i]; tgt_idx[count] = i; ++count;
        }
    } else {
        int inl1 = 0, inl2 = 0;
        for (int i = 0; i < n1; ++i) inl1 += c1[i] != -1;
        for (int i = 0; i < n2; ++i) inl2 += c2[i] != -1;
        if (inl1 > inl2) {
            for (int i = 0; i < n1; ++i) { if (c1[i] == -1) continue; src_idx[count] = i; tgt_idx[count] = c1[i]; ++count; }
        } else {
            for (int i = 0; i < n2; ++i) { if (c2[i] == -1) continue; src_idx[count] = c2[i]; tgt_idx[count] = i; ++count; }
        }
    }
    return count;
}

/* RobustMatching (FEAmatcher.cpp:13-50): returns the rows appended to SourceFrame.corres_kps */
int orc_robust_matching(int id_s, int id_t, int rows_s, int rows_t,
                        const orc_kp* kps_s, const uint8_t* desc_s, const double* geo_s, int ns, const double bbox_s[4],
                        const orc_kp* kps_t, const uint8_t* desc_t, const double* geo_t, int nt, const double bbox_t[4],
                        const orc_match_params* p, double* rows6, int cap)
{
    int32_t* c1 = (int32_t*)malloc(sizeof(int32_t) * (ns > 0 ? ns : 1));
    int32_t* c2 = (int32_t*)malloc(sizeof(int32_t) * (nt > 0 ? nt : 1));
    int cnt1, cnt2; double m1, m2;
    int h1 = orc_match_dir(id_s, id_t, rows_t, kps_s, desc_s, geo_s, ns, kps_t, desc_t, geo_t, nt, bbox_t, p, c1, &cnt1, &m1);
    int h2 = orc_match_dir(id_t, id_s, rows_s, kps_t, desc_t, geo_t, nt, kps_s, desc_s, geo_s, ns, bbox_s, p, c2, &cnt2, &m2);
    int32_t* si = (int32_t*)malloc(sizeof(int32_t) * (ns + nt + 1));
    int32_t* ti = (int32_t*)malloc(sizeof(int32_t) * (ns + nt + 1));
    int n = orc_consistent_check(id_s, id_t, rows_s, rows_t, c1, ns, c2, nt, h1, cnt1, m1, h2, cnt2, m2, p, si, ti);
    int out = 0;
    for (int i = 0; i < n && out < cap; ++i, ++out) {
        double* r = rows6 + (size_t)out * 6;
        r[0] = id_s; r[1] = id_t;
        r[2] = kps_s[si[i]].y; r[3] = kps_s[si[i]].x;
        r[4] = kps_t[ti[i]].y; r[5] = kps_t[ti[i]].x;
    }
    free(c1); free(c2); free(si); free(ti);
    return out;
}
