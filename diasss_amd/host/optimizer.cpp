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
    if (ADD_LC) {
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
