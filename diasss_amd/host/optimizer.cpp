// diasss_amd/host/optimizer.cpp -- Diasss::Optimizer over the C ABI; control flow of
// /root/reference/src/core/optimizer.cpp:21-317 (TrajOptimizationAll), :575-639 (GetKpsPairs), :641-982
// (LoopClosingTFs -> dsss_lc_solve), :1164-1214 (SaveTrajactoryAll).  The numerics run in libdsss.so.
#include "optimizer.h"
#include "dsss_device.h"
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <stdexcept>
#include <algorithm>

namespace Diasss
{

bool Optimizer::USE_ANNO = 0;
bool Optimizer::ADD_LC = 1;
bool Optimizer::ONLINE = 0;
int Optimizer::ONLINE_WINDOW = 0;
bool Optimizer::EVAL_1 = 0;
bool Optimizer::EVAL_2 = 0;

std::vector<Vector7> Optimizer::GetKpsPairs(const bool &use_anno, const cv::Mat &kps, const int &id_s, const int &id_t,
                                            const std::vector<double> &alts_s, const std::vector<double> &gras_s,
                                            const std::vector<double> &alts_t, const std::vector<double> &gras_t)
{
    (void)id_s;
    std::vector<Vector7> kps_pairs;
    for (int i = 0; i < kps.rows; ++i) {
        int id_check, kp_s[2], kp_t[2];
        if (use_anno) {
            id_check = kps.at<int>(i, 1);
            kp_s[0] = kps.at<int>(i, 2); kp_s[1] = kps.at<int>(i, 3); kp_t[0] = kps.at<int>(i, 4); kp_t[1] = kps.at<int>(i, 5);
        } else {
            id_check = (int)kps.at<double>(i, 1);
            kp_s[0] = (int)kps.at<double>(i, 2); kp_s[1] = (int)kps.at<double>(i, 3);
            kp_t[0] = (int)kps.at<double>(i, 4); kp_t[1] = (int)kps.at<double>(i, 5);
        }
        const int nd_thres = 20;                                                     // optimizer.cpp:602
        const int ds = kp_s[1] - (int)gras_s.size(), dt = kp_t[1] - (int)gras_t.size();
        if (std::abs(ds) < nd_thres || std::abs(dt) < nd_thres) continue;
        if (id_check != id_t) continue;
        const double a_s = alts_s[kp_s[0]], g_s = gras_s[std::abs(ds)], a_t = alts_t[kp_t[0]], g_t = gras_t[std::abs(dt)];
        double depth = 0;
        if (use_anno) depth = double(kps.at<int>(i, 6)) / 100000.0;
        kps_pairs.push_back(Vector7{ { (double)kp_s[0], (double)kp_s[1], std::sqrt(a_s * a_s + g_s * g_s),
                                       (double)kp_t[0], (double)kp_t[1], std::sqrt(a_t * a_t + g_t * g_t), depth } });
    }
    return kps_pairs;
}

std::vector<std::tuple<Pose3,Vector6,double>> Optimizer::LoopClosingTFs(const std::vector<Vector7> &kps_pairs,
                                                const std::vector<double> &, const std::vector<double> &,
                                                const int &img_id_s, const int &img_id_t,
                                                const std::vector<cv::Mat> &, const std::vector<cv::Mat> &,
                                                const std::vector<double> &, const std::vector<double> &,
                                                const std::vector<double> &, const std::vector<double> &,
                                                const cv::Mat &, const cv::Mat &)
{
    // geometry (DR poses, altitudes, ground ranges) of both frames is already resident on the device (Frame ctor)
    std::vector<std::tuple<Pose3,Vector6,double>> out;
    const int n = (int)kps_pairs.size();
    if (!n) return out;
    std::vector<dsss_lc> lc(n);
    Device::check(dsss_lc_solve(Device::ctx(), img_id_s, img_id_t, kps_pairs[0].data(), n, lc.data()), "dsss_lc_solve");
    for (int i = 0; i < n; ++i) {
        Pose3 T; Vector6 v;
        for (int k = 0; k < 9; ++k) T.R[k] = lc[i].rel[k];
        for (int k = 0; k < 3; ++k) T.t[k] = lc[i].rel[9 + k];
        for (int k = 0; k < 6; ++k) v[k] = lc[i].var[k];
        out.push_back(std::make_tuple(T, v, lc[i].score));
    }
    return out;
}

Optimizer::Point3 Optimizer::TriangulateOneLandmark(const Vector7 &kps_pair, const Pose3 &, const Pose3 &,
                                                    const Pose3 &Tp_s, const Pose3 &Tp_t, const Point3 &lm_ini)
{
    double in27[27], out7[7];
    for (int k = 0; k < 9; ++k) { in27[k] = Tp_s.R[k]; in27[12 + k] = Tp_t.R[k]; }
    for (int k = 0; k < 3; ++k) { in27[9 + k] = Tp_s.t[k]; in27[21 + k] = Tp_t.t[k]; in27[24 + k] = lm_ini[k]; }
    Device::check(dsss_triangulate_poses(Device::ctx(), kps_pair.data(), in27, 1, out7), "dsss_triangulate_poses");
    return Point3{ { out7[0], out7[1], out7[2] } };
}

void Optimizer::TrajOptimizationAll(std::vector<Frame> &AllFrames)
{
    const size_t F = AllFrames.size();
    dsss_ctx* c = Device::ctx();
    // --- keypoint pairs of every frame pair (optimizer.cpp:35-97): GetKpsPairs on the host from corres_kps or, with
    //     USE_ANNO, from anno_kps; the LoopClosingTFs calls of ALL pairs then run as ONE device launch
    std::vector<std::vector<Vector7>> kps_pairs_all;
    std::vector<int> ps, pt, pair_off(1, 0);
    std::vector<double> kp7;
    for (size_t i = 0; i < F; i++)
        for (size_t j = i + 1; j < F; j++) {
            const cv::Mat &src = USE_ANNO ? AllFrames[i].anno_kps : AllFrames[i].corres_kps;
            kps_pairs_all.push_back(GetKpsPairs(USE_ANNO, src, AllFrames[i].img_id, AllFrames[j].img_id, AllFrames[i].altitudes,
                                                AllFrames[i].ground_ranges, AllFrames[j].altitudes, AllFrames[j].ground_ranges));
            ps.push_back(AllFrames[i].img_id); pt.push_back(AllFrames[j].img_id);
            for (const Vector7 &v : kps_pairs_all.back()) kp7.insert(kp7.end(), v.begin(), v.end());
            pair_off.push_back((int)(kp7.size() / 7));
        }
    // the device works on frame ids 0..F-1 in AllFrames order (diasss2.cpp:84 numbers them that way)
    for (size_t i = 0; i < F; i++)
        if (AllFrames[i].img_id != (int)i) throw std::runtime_error("TrajOptimizationAll: AllFrames[i].img_id must equal i (diasss2.cpp:84)");
    if (!(ADD_LC && ONLINE))
        Device::check(dsss_lc_solve_pairs(c, ps.data(), pt.data(), (int)ps.size(), kp7.data(), pair_off.data()), "dsss_lc_solve_pairs");
    // --- unique pose ids (optimizer.cpp:101-114)
    int id_sum = 0;
    std::vector<std::vector<int>> unique_id;
    for (size_t i = 0; i < F; i++) {
        std::vector<int> id_tmp(AllFrames[i].dr_poses.rows);
        for (int j = 0; j < AllFrames[i].dr_poses.rows; j++) id_tmp[j] = id_sum++;
        unique_id.push_back(id_tmp);
    }
    // --- LC factor per target ping (last pair wins, first kp in it, score > 0: optimizer.cpp:203-258) and the batch LM over
    //     every ping (replaces the iSAM2 loop, :134-279), both on the device
    std::vector<double> poses12((size_t)id_sum * 12);
    double stats[4] = { 0, 0, 0, 0 };
    int n_edges = 0;
    if (ADD_LC && ONLINE) {
        // the reference's own order of events (optimizer.cpp:145-272): frame by frame, the loop closures that END in the new
        // frame join the graph together with its pings, and the estimate is brought up to date before the next frame
        Device::check(dsss_posegraph_reset(c), "dsss_posegraph_reset");
        int updates = 0, trials = 0;
        for (size_t j = 0; j < F; j++) {
            std::vector<int> s_j, t_j, off_j(1, 0);
            std::vector<double> k_j;
            for (size_t p = 0; p < ps.size(); p++)
                if (pt[p] == (int)j && pair_off[p + 1] > pair_off[p]) {
                    s_j.push_back(ps[p]); t_j.push_back(pt[p]);
                    k_j.insert(k_j.end(), kp7.begin() + (size_t)7 * pair_off[p], kp7.begin() + (size_t)7 * pair_off[p + 1]);
                    off_j.push_back((int)(k_j.size() / 7));
                }
            if (!s_j.empty()) Device::check(dsss_lc_solve_pairs(c, s_j.data(), t_j.data(), (int)s_j.size(), k_j.data(), off_j.data()), "dsss_lc_solve_pairs");
            double st_j[4] = { 0, 0, 0, 0 };
            // ONLINE_WINDOW > 0: the incremental form -- an update solves the last ONLINE_WINDOW frames conditioned on the frozen rest (cost per
            // update independent of the survey's length); the LAST update is the global one, warm-started from the windowed estimates: the
            // batch optimum is what the reference reads after its loop (calculateEstimate, optimizer.cpp:279)
            if (ONLINE_WINDOW > 0 && j + 1 < F) Device::check(dsss_posegraph_update_window(c, (int)j + 1, ONLINE_WINDOW, nullptr, nullptr, st_j), "dsss_posegraph_update_window");
            else Device::check(dsss_posegraph_update(c, (int)j + 1, j + 1 == F ? poses12.data() : nullptr, nullptr, st_j), "dsss_posegraph_update");
            ++updates; trials += (int)st_j[0];
            if (j == 0) stats[1] = st_j[1];
            stats[0] = st_j[0]; stats[2] = st_j[2]; stats[3] = st_j[3];
        }
        n_edges = dsss_posegraph_online_edges(c);
        std::cout << "online: " << updates << " updates, " << trials << " accepted LM steps in all" << std::endl;
    } else if (ADD_LC) {
        Device::check(dsss_posegraph_solve(c, (int)F, poses12.data(), nullptr, stats), "dsss_posegraph_solve");
        std::vector<dsss_lc_edge> tmp((size_t)std::max<size_t>(kp7.size() / 7, 1));
        Device::check(dsss_posegraph_select(c, (int)F, tmp.data(), (int)tmp.size(), &n_edges), "dsss_posegraph_select");
    } else {
        std::vector<double> dr;
        for (size_t i = 0; i < F; i++) dr.insert(dr.end(), AllFrames[i].dr_poses.ptr<double>(), AllFrames[i].dr_poses.ptr<double>() + (size_t)AllFrames[i].dr_poses.rows * 6);
        Device::check(dsss_posegraph_solve_edges(c, dr.data(), id_sum, nullptr, 0, poses12.data(), stats), "dsss_posegraph_solve_edges");
    }
    std::cout << "pose graph: " << id_sum << " poses, " << n_edges << " loop closures, " << (int)stats[0] << " LM iterations, error "
              << stats[1] << " -> " << stats[2] << std::endl;
    size_t o = 0;
    for (size_t i = 0; i < F; i++) {
        AllFrames[i].est_poses = cv::Mat(AllFrames[i].dr_poses.rows, 12, CV_64F);
        std::copy(poses12.begin() + o, poses12.begin() + o + (size_t)AllFrames[i].dr_poses.rows * 12, AllFrames[i].est_poses.ptr<double>());
        o += (size_t)AllFrames[i].dr_poses.rows * 12;
    }
    std::vector<cv::Mat> dr_poses_all;
    for (size_t i = 0; i < F; i++) dr_poses_all.push_back(AllFrames[i].dr_poses);
    SaveTrajactoryAll(poses12, unique_id, dr_poses_all);
    // --- evaluation with the annotated keypoints (optimizer.cpp:288-312)
    if (EVAL_1 || EVAL_2) {
        std::vector<std::vector<Vector7>> anno_pairs;
        std::vector<std::pair<int,int>> ids;
        for (size_t i = 0; i < F; i++)
            for (size_t j = i + 1; j < F; j++) {
                anno_pairs.push_back(GetKpsPairs(true, AllFrames[i].anno_kps, AllFrames[i].img_id, AllFrames[j].img_id, AllFrames[i].altitudes,
                                                 AllFrames[i].ground_ranges, AllFrames[j].altitudes, AllFrames[j].ground_ranges));
                ids.push_back(std::make_pair((int)i, (int)j));
            }
        EvaluateByAnnosAll(poses12, unique_id, AllFrames, anno_pairs, ids);
    }
}

// Frame::GetGeoImg for one bin (frame.cpp:126-165, tf = 0): starboard = columns >= M/2 at yaw + PI/2, port = columns < M/2 at yaw - PI/2
static void geo_of(const cv::Mat &dr, const std::vector<double> &gr, int M, int row, int col, double &x, double &y)
{
    const double PI = 3.14159265358979323846;
    const double* P = dr.ptr<double>(row);
    const int half = M / 2;
    int idx; double ang;
    if (col >= half) { idx = col - half; ang = P[2] + PI / 2; }
    else { idx = half - col; if (idx > half - 1) idx = half - 1; ang = P[2] - PI / 2; }
    x = P[3] + gr[idx] * std::cos(ang); y = P[4] + gr[idx] * std::sin(ang);
}
static void rodrigues(const double* w, double* R);
static void rpy_of(const double* R, double* rpy);

std::vector<Optimizer::AnnoStats> Optimizer::EvaluateByAnnosAll(const std::vector<double> &poses12, const std::vector<std::vector<int>> &unique_id,
                                                                const std::vector<Frame> &AllFrames,
                                                                const std::vector<std::vector<Vector7>> &kps_pairs_all,
                                                                const std::vector<std::pair<int,int>> &img_pairs_ids)
{
    const double PI = 3.14159265358979323846;
    const bool save_result = true, show_stats = true;                       // optimizer.cpp:1578
    const char* dir = std::getenv("DSSS_OUT_DIR");
    const std::string base = dir ? std::string(dir) + "/" : std::string("../");
    auto open_out = [&](std::ofstream &f, const std::string &rel) { if (save_result) f.open((base + rel).c_str(), std::ios::trunc); };
    std::vector<AnnoStats> out(kps_pairs_all.size());
    for (size_t i = 0; i < kps_pairs_all.size(); i++) { out[i].img_s = img_pairs_ids[i].first; out[i].img_t = img_pairs_ids[i].second; out[i].n = (int)kps_pairs_all[i].size(); }
    if (EVAL_2) {
        // every landmark is triangulated twice (dead-reckoning poses, estimated poses): two batched device calls for all pairs
        std::vector<double> kp7, in_dr, in_est;
        for (size_t i = 0; i < kps_pairs_all.size(); i++) {
            const Frame &S = AllFrames[img_pairs_ids[i].first], &T = AllFrames[img_pairs_ids[i].second];
            for (const Vector7 &v : kps_pairs_all[i]) {
                const int id_s = (int)v[0], id_ss = (int)v[1], id_t = (int)v[3], id_tt = (int)v[4];
                double xs, ys, xt, yt;
                geo_of(S.dr_poses, S.ground_ranges, S.raw_img.cols, id_s, id_ss, xs, ys);
                geo_of(T.dr_poses, T.ground_ranges, T.raw_img.cols, id_t, id_tt, xt, yt);
                const double ini[3] = { (xs + xt) / 2, (ys + yt) / 2,
                                        ((S.dr_poses.at<double>(id_s, 5) - S.altitudes[id_s]) + (T.dr_poses.at<double>(id_t, 5) - T.altitudes[id_t])) / 2 };
                kp7.insert(kp7.end(), v.begin(), v.end());
                double r27[27];
                rodrigues(S.dr_poses.ptr<double>(id_s), r27); rodrigues(T.dr_poses.ptr<double>(id_t), r27 + 12);
                for (int k = 0; k < 3; ++k) { r27[9 + k] = S.dr_poses.at<double>(id_s, 3 + k); r27[21 + k] = T.dr_poses.at<double>(id_t, 3 + k); r27[24 + k] = ini[k]; }
                in_dr.insert(in_dr.end(), r27, r27 + 27);
                const double* Es = poses12.data() + (size_t)unique_id[img_pairs_ids[i].first][id_s] * 12;
                const double* Et = poses12.data() + (size_t)unique_id[img_pairs_ids[i].second][id_t] * 12;
                std::copy(Es, Es + 12, r27); std::copy(Et, Et + 12, r27 + 12);
                in_est.insert(in_est.end(), r27, r27 + 27);
            }
        }
        const int n = (int)(kp7.size() / 7);
        std::vector<double> o_dr((size_t)std::max(n, 1) * 7), o_est((size_t)std::max(n, 1) * 7);
        if (n > 0) {
            Device::check(dsss_triangulate_poses(Device::ctx(), kp7.data(), in_dr.data(), n, o_dr.data()), "dsss_triangulate_poses");
            Device::check(dsss_triangulate_poses(Device::ctx(), kp7.data(), in_est.data(), n, o_est.data()), "dsss_triangulate_poses");
        }
        // out7 = landmark (3) and the four consistency figures of the call: |range_s|, |plane_s|, |range_t|, |plane_t| (dsss.h)
        std::ofstream a1, a2, a3, a4;
        open_out(a1, "result/pr_errors/dr_range_e_avg.txt"); open_out(a2, "result/pr_errors/dr_plane_e_avg.txt");
        open_out(a3, "result/pr_errors/est_range_e_avg.txt"); open_out(a4, "result/pr_errors/est_plane_e_avg.txt");
        size_t q = 0;
        for (size_t i = 0; i < kps_pairs_all.size(); i++) {
            AnnoStats &st = out[i];
            std::ofstream f1, f2, f3, f4;
            open_out(f1, "result/pr_errors/dr_range_e_" + std::to_string(i) + ".txt"); open_out(f2, "result/pr_errors/dr_plane_e_" + std::to_string(i) + ".txt");
            open_out(f3, "result/pr_errors/est_range_e_" + std::to_string(i) + ".txt"); open_out(f4, "result/pr_errors/est_plane_e_" + std::to_string(i) + ".txt");
            int g1 = 0, g2 = 0;
            for (size_t j = 0; j < kps_pairs_all[i].size(); j++, q++) {
                const double* d = o_dr.data() + q * 7; const double* e = o_est.data() + q * 7;
                const double range_dr = (d[3] + d[5]) / 2, plane_dr = (d[4] + d[6]) / 2, range_est = (e[3] + e[5]) / 2, plane_est = (e[6] + e[4]) / 2;
                if (f1.is_open()) { f1 << range_dr << std::endl; f2 << plane_dr << std::endl; f3 << range_est << std::endl; f4 << plane_est << std::endl; }
                g1 += range_dr > range_est; g2 += plane_dr > plane_est;
                st.range_dr += range_dr; st.plane_dr += plane_dr; st.range_est += range_est; st.plane_est += plane_est;
            }
            const double m = (double)kps_pairs_all[i].size();
            st.good_range = g1 / m * 100; st.good_plane = g2 / m * 100;
            st.range_dr /= m; st.plane_dr /= m; st.range_est /= m; st.plane_est /= m;
            if (a1.is_open()) { a1 << st.range_dr << std::endl; a2 << st.plane_dr << std::endl; a3 << st.range_est << std::endl; a4 << st.plane_est << std::endl; }
            if (show_stats) {
                std::cout << "Metric Statics: " << st.good_range << " " << st.good_plane << " " << kps_pairs_all[i].size() << " " << st.img_s << " " << st.img_t << std::endl;
                std::cout << "Avg R and P (DR/EST): " << st.range_dr << "/" << st.range_est << " " << st.plane_dr << "/" << st.plane_est << std::endl << std::endl;
            }
        }
    }
    if (EVAL_1) {
        for (size_t i = 0; i < kps_pairs_all.size(); i++) {
            AnnoStats &st = out[i];
            const Frame &S = AllFrames[img_pairs_ids[i].first], &T = AllFrames[img_pairs_ids[i].second];
            std::ofstream f1, f2, f3;
            open_out(f1, "result/anno_errors/dr_lm_dist_" + std::to_string(i) + ".txt"); open_out(f2, "result/anno_errors/est_lm_dist_" + std::to_string(i) + ".txt");
            open_out(f3, "result/anno_errors/lm_dist_compare_" + std::to_string(i) + ".txt");
            int good = 0;
            for (const Vector7 &v : kps_pairs_all[i]) {
                const int id_s = (int)v[0], id_ss = (int)v[1], id_t = (int)v[3], id_tt = (int)v[4];
                double xs, ys, xt, yt;
                geo_of(S.dr_poses, S.ground_ranges, S.raw_img.cols, id_s, id_ss, xs, ys);
                geo_of(T.dr_poses, T.ground_ranges, T.raw_img.cols, id_t, id_tt, xt, yt);
                const double ix = xs - xt, iy = ys - yt, ini = std::sqrt(ix * ix + iy * iy);
                auto est_geo = [&](const Frame &Fm, int uid, int bin, double &gx, double &gy) {          // optimizer.cpp:1806-1832
                    const double* E = poses12.data() + (size_t)uid * 12;
                    double rpy[3]; rpy_of(E, rpy);
                    const int half = Fm.raw_img.cols / 2;
                    if (bin < half) { const int g = half - bin; gx = E[9] + Fm.ground_ranges[g] * std::cos(rpy[2] + PI / 2 - PI); gy = E[10] + Fm.ground_ranges[g] * std::sin(rpy[2] + PI / 2 - PI); }
                    else { const int g = bin - half; gx = E[9] + Fm.ground_ranges[g] * std::cos(rpy[2] - PI / 2 - PI); gy = E[10] + Fm.ground_ranges[g] * std::sin(rpy[2] - PI / 2 - PI); }
                };
                double sx, sy, tx, ty;
                est_geo(S, unique_id[img_pairs_ids[i].first][id_s], id_ss, sx, sy);
                est_geo(T, unique_id[img_pairs_ids[i].second][id_t], id_tt, tx, ty);
                const double fx = sx - tx, fy = sy - ty, fin = std::sqrt(fx * fx + fy * fy);
                st.x_dr += std::fabs(ix); st.y_dr += std::fabs(iy); st.all_dr += std::fabs(ini);
                st.x_est += std::fabs(fx); st.y_est += std::fabs(fy); st.all_est += std::fabs(fin);
                good += ini > fin;
                if (f1.is_open()) { f1 << ini << std::endl; f2 << fin << std::endl; f3 << ini - fin << std::endl; }
            }
            const double m = (double)kps_pairs_all[i].size();
            st.good_dist = good / m * 100;
            st.x_dr /= m; st.y_dr /= m; st.all_dr /= m; st.x_est /= m; st.y_est /= m; st.all_est /= m;
            if (show_stats) {
                std::cout << "LM Metric Statics: " << st.good_dist << " " << kps_pairs_all[i].size() << " " << st.img_s << " " << st.img_t << std::endl;
                std::cout << "Avg X,Y,NORM (DR/EST): " << st.x_dr << "/" << st.x_est << " " << st.y_dr << "/" << st.y_est << " " << st.all_dr << "/" << st.all_est << std::endl << std::endl;
            }
        }
    }
    return out;
}

static void rpy_of(const double* R, double* rpy)
{
    rpy[0] = std::atan2(R[7], R[8]); rpy[1] = std::atan2(-R[6], std::sqrt(R[7] * R[7] + R[8] * R[8])); rpy[2] = std::atan2(R[3], R[0]);
}
static void rodrigues(const double* w, double* R)
{
    const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    const double W[9] = { 0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0 };
    double WW[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) WW[3 * i + j] = W[3 * i] * W[j] + W[3 * i + 1] * W[3 + j] + W[3 * i + 2] * W[6 + j];
    const double th = std::sqrt(t2);
    const double a = t2 > 2.220446049250313e-16 ? std::sin(th) / th : 1.0, b = t2 > 2.220446049250313e-16 ? (1 - std::cos(th)) / t2 : 0.0;
    for (int i = 0; i < 9; ++i) R[i] = a * W[i] + b * WW[i];
    R[0] += 1; R[4] += 1; R[8] += 1;
}

// "r p y x y z", fixed, 9 decimals (optimizer.cpp:1164-1214)
void Optimizer::SaveTrajactoryAll(const std::vector<double> &poses12, const std::vector<std::vector<int>> &unique_id,
                                  const std::vector<cv::Mat> &dr_poses_all)
{
    const char* dir = std::getenv("DSSS_OUT_DIR");
    const std::string base = dir ? std::string(dir) + "/" : std::string("../");
    std::ofstream f1((base + "dr_poses_all.txt").c_str(), std::ios::trunc);
    for (size_t i = 0; i < dr_poses_all.size(); i++)
        for (int j = 0; j < dr_poses_all[i].rows; j++) {
            const double* p = dr_poses_all[i].ptr<double>(j);
            double R[9], rpy[3];
            rodrigues(p, R); rpy_of(R, rpy);
            f1 << std::fixed << std::setprecision(9) << rpy[0] << " " << rpy[1] << " " << rpy[2] << " " << p[3] << " " << p[4] << " " << p[5] << std::endl;
        }
    f1.close();
    std::ofstream f2((base + "est_poses_all.txt").c_str(), std::ios::trunc);
    for (size_t i = 0; i < unique_id.size(); i++)
        for (size_t j = 0; j < unique_id[i].size(); j++) {
            const double* T = poses12.data() + (size_t)unique_id[i][j] * 12;
            double rpy[3];
            rpy_of(T, rpy);
            f2 << std::fixed << std::setprecision(9) << rpy[0] << " " << rpy[1] << " " << rpy[2] << " " << T[9] << " " << T[10] << " " << T[11] << std::endl;
        }
    f2.close();
}

} // namespace Diasss
