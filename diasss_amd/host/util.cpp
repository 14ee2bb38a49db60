// diasss_amd/host/util.cpp -- Util::ComputeIntersection (mirrors /root/reference/src/util/util.cpp:13-43) on the
// compact geo_img (the four minMaxLoc results were computed on the device by geo_bbox_kernel)
#include "util.h"
#include "filestorage.h"
#include <algorithm>
#include <cmath>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <sstream>

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

// files of a folder, ordered by name (util.cpp:48-81 sorts boost::filesystem paths the same way)
static std::vector<std::string> sorted_files(const std::string &folder)
{
    std::vector<std::string> out;
    if (folder.empty()) return out;
    std::error_code ec;
    for (const auto &e : std::filesystem::directory_iterator(folder, ec)) out.push_back(e.path().string());
    if (ec) std::cout << "cannot list " << folder << ": " << ec.message() << std::endl;
    std::sort(out.begin(), out.end());
    return out;
}

// one number per non-empty line (util.cpp:129-148): the first token of the line
static std::vector<double> read_column(const std::string &path)
{
    std::vector<double> v;
    std::ifstream f(path.c_str());
    std::string s;
    while (std::getline(f, s)) {
        if (s.empty()) continue;
        std::stringstream ss; ss << s;
        double x;
        if (ss >> x) v.push_back(x);
    }
    return v;
}

void Util::LoadInputData(const std::string &strImageFolder, const std::string &strPoseFolder, const std::string &strAltitudeFolder,
                         const std::string &strGroundRangeFolder, const std::string &strAnnotationFolder,
                         std::vector<cv::Mat> &vmImgs, std::vector<cv::Mat> &vmPoses, std::vector<std::vector<double>> &vvAltts,
                         std::vector<std::vector<double>> &vvGranges, std::vector<cv::Mat> &vmAnnos)
{
    auto load_mats = [](const std::string &folder, const char *node, const char *what, std::vector<cv::Mat> &dst) {
        for (const std::string &path : sorted_files(folder)) {
            cv::Mat m; std::string err;
            if (!ReadStorageMatrix(path, node, m, &err)) std::cout << "skipping " << path << ": " << err << std::endl;   // the reference pushes an empty Mat and fails later
            dst.push_back(m);
            std::cout << what << " size: " << m.rows << " " << m.cols << std::endl;
        }
    };
    load_mats(strImageFolder, "ct_img", "image", vmImgs);                      // util.cpp:84-101
    load_mats(strPoseFolder, "auv_pose", "pose", vmPoses);                     // :104-124
    for (const std::string &path : sorted_files(strAltitudeFolder)) {          // :127-153
        vvAltts.push_back(read_column(path));
        std::cout << "alttitude size: " << vvAltts.back().size() << std::endl;
    }
    for (const std::string &path : sorted_files(strGroundRangeFolder)) {       // :156-182
        vvGranges.push_back(read_column(path));
        std::cout << "ground range size: " << vvGranges.back().size() << std::endl;
    }
    load_mats(strAnnotationFolder, "anno_kps", "annotation", vmAnnos);         // :185-208
}

} // namespace Diasss
