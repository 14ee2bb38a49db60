// diasss_amd/host/util.h -- drop-in for the in-scope part of /root/reference/src/util/util.h:25-28
#ifndef UTIL_H
#define UTIL_H

#include <vector>
#include "cvlite.h"

namespace Diasss
{

    class Util
    {
    public:
        // util.h:25 / util.cpp:13-43: IoU of the geo bounding boxes (float arithmetic on double extrema)
        static float ComputeIntersection(const std::vector<cv::Mat> &geo_img_s, const std::vector<cv::Mat> &geo_img_t);
        // Util::LoadInputData (OpenCV FileStorage XML + txt) is out of scope
    };

}

#endif
