// diasss_amd/host/FEAmatcher.cpp -- FEAmatcher over the C ABI (mirrors /root/reference/src/core/FEAmatcher.cpp:13-50)
#include "FEAmatcher.h"
#include "dsss_device.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace Diasss
{

int FEAmatcher::USE_SIFT = 0;

static void apply_mode(dsss_ctx* c)
{
    dsss_match_params mt; dsss_match_params_default(&mt);
    mt.use_l2 = FEAmatcher::USE_SIFT;
    Device::check(dsss_set_params(c, nullptr, nullptr, &mt, nullptr), "dsss_set_params");
}

static void append_rows(Frame &S, Frame &T, const std::vector<double> &rows, int n)
{
    for (int i = 0; i < n; ++i) {
        const double* r = rows.data() + (size_t)i * 6;
        cv::Mat s(1, 6, CV_64F), t(1, 6, CV_64F);
        for (int k = 0; k < 6; ++k) s.at<double>(0, k) = r[k];
        t.at<double>(0, 0) = r[1]; t.at<double>(0, 1) = r[0]; t.at<double>(0, 2) = r[4]; t.at<double>(0, 3) = r[5];
        t.at<double>(0, 4) = r[2]; t.at<double>(0, 5) = r[3];                        // mirrored row (FEAmatcher.cpp:41-44)
        S.corres_kps.push_back(s);
        T.corres_kps.push_back(t);
    }
}

void FEAmatcher::RobustMatchingAll(std::vector<Frame> &Frames, const std::vector<std::pair<int,int>> &Pairs)
{
    dsss_ctx* c = Device::ctx();
    std::vector<int> s, t;
    for (auto &p : Pairs) { s.push_back(Frames[p.first].img_id); t.push_back(Frames[p.second].img_id); }
    apply_mode(c);
    Device::check(dsss_match_pairs(c, s.data(), t.data(), (int)Pairs.size()), "dsss_match_pairs");
    for (size_t p = 0; p < Pairs.size(); ++p) {
        int n = 0;
        Device::check(dsss_match_get_rows(c, (int)p, nullptr, 0, &n), "dsss_match_get_rows");
        if (!n) continue;
        std::vector<double> rows((size_t)n * 6);
        Device::check(dsss_match_get_rows(c, (int)p, rows.data(), n, &n), "dsss_match_get_rows");
        append_rows(Frames[Pairs[p].first], Frames[Pairs[p].second], rows, n);
    }
}

void FEAmatcher::RobustMatching(Frame &SourceFrame, Frame &TargetFrame)
{
    dsss_ctx* c = Device::ctx();
    const int s = SourceFrame.img_id, t = TargetFrame.img_id;
    apply_mode(c);
    Device::check(dsss_match_pairs(c, &s, &t, 1), "dsss_match_pairs");
    int n = 0;
    Device::check(dsss_match_get_rows(c, 0, nullptr, 0, &n), "dsss_match_get_rows");
    if (!n) return;
    std::vector<double> rows((size_t)n * 6);
    Device::check(dsss_match_get_rows(c, 0, rows.data(), n, &n), "dsss_match_get_rows");
    append_rows(SourceFrame, TargetFrame, rows, n);
}

std::vector<int> FEAmatcher::GeoNearNeighSearch(const int &img_id, const int &img_id_ref, const cv::Mat &, const cv::Mat &,
                                                const std::vector<cv::KeyPoint> &kps, const cv::Mat &, const std::vector<cv::Mat> &,
                                                const std::vector<cv::KeyPoint> &, const cv::Mat &, const std::vector<cv::Mat> &,
                                                std::vector<std::pair<int,double>> &scc)
{
    dsss_ctx* c = Device::ctx();
    const int s = img_id, t = img_id_ref;
    apply_mode(c);
    Device::check(dsss_match_pairs(c, &s, &t, 1), "dsss_match_pairs");
    std::vector<int> CorresID(kps.size(), -1);
    int hist = 0, count = 0; double model = 0;
    std::vector<int32_t> buf(std::max<size_t>(kps.size(), 1));
    Device::check(dsss_match_get_dir(c, 0, 0, nullptr, buf.data(), (int)buf.size(), &hist, &count, &model), "dsss_match_get_dir");
    for (size_t i = 0; i < kps.size(); ++i) CorresID[i] = buf[i];
    scc.clear();
    if (hist > 0) scc.push_back(std::make_pair(count, model));         // FEAmatcher.cpp:229-231: the last improvement is the maximum
    return CorresID;
}

void FEAmatcher::ConsistentCheck(const Frame &SourceFrame, const Frame &TargetFrame,
                                 const std::vector<int> &CorresID_1, const std::vector<int> &CorresID_2,
                                 std::vector<std::pair<int,double>> &scc_1, std::vector<std::pair<int,double>> &scc_2,
                                 std::vector<cv::KeyPoint> &SourceKeys, std::vector<cv::KeyPoint> &TargetKeys)
{
    const double kp_diff_thres = 2.5;                                  // FEAmatcher.cpp:329
    std::sort(scc_1.rbegin(), scc_1.rend());
    std::sort(scc_2.rbegin(), scc_2.rend());
    double img_diff = 0;
    if (SourceFrame.img_id % 2 != TargetFrame.img_id % 2) img_diff = std::abs(SourceFrame.raw_img.rows - TargetFrame.raw_img.rows);
    // scc_x[0] of an empty vector is undefined in the reference (:344): no model on either side means no merge
    const bool merge = !scc_1.empty() && !scc_2.empty() && std::abs(std::abs(scc_1[0].second - scc_2[0].second) - img_diff) <= kp_diff_thres;
    auto take1 = [&](bool skip_mutual) {
        for (size_t i = 0; i < CorresID_1.size(); i++) {
            if (CorresID_1[i] == -1) continue;
            if (skip_mutual && CorresID_2[CorresID_1[i]] == (int)i) continue;
            SourceKeys.push_back(SourceFrame.kps[i]); TargetKeys.push_back(TargetFrame.kps[CorresID_1[i]]);
        }
    };
    auto take2 = [&]() {
        for (size_t i = 0; i < CorresID_2.size(); i++) {
            if (CorresID_2[i] == -1) continue;
            SourceKeys.push_back(SourceFrame.kps[CorresID_2[i]]); TargetKeys.push_back(TargetFrame.kps[i]);
        }
    };
    if (merge) { take1(true); take2(); }
    else {
        const long inl_1 = (long)CorresID_1.size() - std::count(CorresID_1.begin(), CorresID_1.end(), -1);
        const long inl_2 = (long)CorresID_2.size() - std::count(CorresID_2.begin(), CorresID_2.end(), -1);
        if (inl_1 > inl_2) take1(false); else take2();
    }
}

int FEAmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
{
    const uint32_t* pa = a.ptr<uint32_t>(); const uint32_t* pb = b.ptr<uint32_t>();
    int d = 0;
    for (int i = 0; i < 8; ++i) d += __builtin_popcount(pa[i] ^ pb[i]);
    return d;
}

} // namespace Diasss
