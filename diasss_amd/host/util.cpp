// diasss_amd/host/util.cpp -- Util::ComputeIntersection (mirrors /root/reference/src/util/util.cpp:13-43) on the
// compact geo_img (the four minMaxLoc results were computed on the device by geo_bbox_kernel)
#include "util.h"
#include <algorithm>
#include <cmath>

namespace Diasss
{

static void min_max(const cv::Mat &m, double &mn, double &mx)
{
    mn = mx = m.at<double>(0, 0);
    for (int i = 0; i < m.rows; ++i) for (int j = 0; j < m.cols; ++j) { const double v = m.at<double>(i, j); mn = std::min(mn, v); mx = std::max(mx, v); }
}

float Util::ComputeIntersection(const std::vector<cv::Mat> &geo_img_s, const std::vector<cv::Mat> &geo_img_t)
{
    float output = 0.0;
    double sx_min, sy_min, sx_max, sy_max, tx_min, tx_max, ty_min, ty_max;
    min_max(geo_img_s[0], sx_min, sx_max); min_max(geo_img_s[1], sy_min, sy_max);
    min_max(geo_img_t[0], tx_min, tx_max); min_max(geo_img_t[1], ty_min, ty_max);
    float x_dist_ol = std::min(sx_max, tx_max) - std::max(sx_min, tx_min);
    float y_dist_ol = std::min(ty_max, sy_max) - std::max(sy_min, ty_min);
    if (x_dist_ol > 0 && y_dist_ol > 0) {
        float area_ol = x_dist_ol * y_dist_ol;
        float area_s = std::abs(sx_max - sx_min) * std::abs(sy_max - sy_min);
        float area_t = std::abs(tx_max - tx_min) * std::abs(ty_max - ty_min);
        output = area_ol / (area_s + area_t - area_ol);
    }
    return output;
}

} // namespace Diasss
