// diasss_amd/host/util.h -- drop-in for the in-scope part of /root/reference/src/util/util.h:25-28
#ifndef UTIL_H
#define UTIL_H

#include <string>
#include <vector>
#include "cvlite.h"

namespace Diasss
{

    class Util
    {
    public:
        // util.h:25 / util.cpp:13-43: IoU of the geo bounding boxes (float arithmetic on double extrema)
        static float ComputeIntersection(const std::vector<cv::Mat> &geo_img_s, const std::vector<cv::Mat> &geo_img_t);
        // util.h:27-28 / util.cpp:45-213: every file of the five folders in name order -- images
        // ("ct_img"), DR poses ("auv_pose") and annotations ("anno_kps") as OpenCV FileStorage matrices (XML or YAML,
        // read without OpenCV by filestorage.cpp), altitudes and ground ranges as text, first number of every non-empty line.
        // An empty annotation folder name is allowed (USE_ANNO = 0 never reads the annotations).
        static void LoadInputData(const std::string &strImageFolder, const std::string &strPoseFolder, const std::string &strAltitudeFolder,
                                  const std::string &strGroundRangeFolder, const std::string &strAnnotationFolder,
                                  std::vector<cv::Mat> &vmImgs, std::vector<cv::Mat> &vmPoses, std::vector<std::vector<double>> &vvAltts,
                                  std::vector<std::vector<double>> &vvGranges, std::vector<cv::Mat> &vmAnnos);
    };

}

#endif
