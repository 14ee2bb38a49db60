// diasss_amd/host/FEAmatcher.cpp -- FEAmatcher over the C ABI (mirrors /root/reference/src/core/FEAmatcher.cpp:13-50)
#include "FEAmatcher.h"
#include "dsss_device.h"

namespace Diasss
{

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
    Device::check(dsss_match_pairs(c, &s, &t, 1), "dsss_match_pairs");
    int n = 0;
    Device::check(dsss_match_get_rows(c, 0, nullptr, 0, &n), "dsss_match_get_rows");
    if (!n) return;
    std::vector<double> rows((size_t)n * 6);
    Device::check(dsss_match_get_rows(c, 0, rows.data(), n, &n), "dsss_match_get_rows");
    append_rows(SourceFrame, TargetFrame, rows, n);
}

int FEAmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
{
    const uint32_t* pa = a.ptr<uint32_t>(); const uint32_t* pb = b.ptr<uint32_t>();
    int d = 0;
    for (int i = 0; i < 8; ++i) d += __builtin_popcount(pa[i] ^ pb[i]);
    return d;
}

} // namespace Diasss
