// diasss_amd/host/FEAmatcher.h -- drop-in for /root/reference/src/core/FEAmatcher.h:16-36
#ifndef FEAMATCHER_H
#define FEAMATCHER_H

#include <utility>
#include <vector>
#include "frame.h"

namespace Diasss
{

    class FEAmatcher
    {
    public:

        // FEAmatcher.h:20 -- appends [id_s,id_t,y_s,x_s,y_t,x_t] rows to both frames' corres_kps (FEAmatcher.cpp:35-45)
        static void RobustMatching(Frame &SourceFrame, Frame &TargetFrame);

        // batched extension: every listed (source index, target index) pair in ONE device call
        static void RobustMatchingAll(std::vector<Frame> &Frames, const std::vector<std::pair<int,int>> &Pairs);

        // FEAmatcher.h:33 on descriptor rows (host utility; the device matcher uses v_bcnt)
        static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);
    };

}

#endif
