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

        // FEAmatcher.h:22-26 -- one direction of the geo-gated nearest-neighbour search + sliding compatibility check.
        // The features of both frames are the device-resident ones of img_id / img_id_ref (what RobustMatching passes:
        // the frames' own kps / dst / geo_img), so the image, keypoint and descriptor arguments only fix the sizes.
        // scc receives the best (inlier count, model) pair -- the only entry ConsistentCheck reads after its sort.
        static std::vector<int> GeoNearNeighSearch(const int &img_id, const int &img_id_ref,
                                                   const cv::Mat &img, const cv::Mat &img_ref,
                                                   const std::vector<cv::KeyPoint> &kps, const cv::Mat &dst, const std::vector<cv::Mat> &geo_img,
                                                   const std::vector<cv::KeyPoint> &kps_ref, const cv::Mat &dst_ref, const std::vector<cv::Mat> &geo_img_ref,
                                                   std::vector<std::pair<int,double>> &scc);

        // FEAmatcher.h:28-31 -- bidirectional merge (FEAmatcher.cpp:323-405), host logic on the two CorresID vectors
        static void ConsistentCheck(const Frame &SourceFrame, const Frame &TargetFrame,
                                    const std::vector<int> &CorresID_1, const std::vector<int> &CorresID_2,
                                    std::vector<std::pair<int,double>> &scc_1, std::vector<std::pair<int,double>> &scc_2,
                                    std::vector<cv::KeyPoint> &SourceKeys, std::vector<cv::KeyPoint> &TargetKeys);

        // FEAmatcher.cpp:63 hard-codes USE_SIFT = 1 (the L2 branch :106-139 on whatever Frame::dst holds).  0 (default): the Hamming branch
        // on the ORB rows; 1: the L2 branch on the 32 ORB bytes (the shipped behaviour, minus the uninitialised memory); 2: the L2 branch on
        // the 128-float rows (needs Frame::USE_SIFT).  Applied to the device context by every matching call.
        static int USE_SIFT;

        // FEAmatcher.h:33 on descriptor rows (host utility; the device matcher uses v_bcnt)
        static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);
    };

}

#endif
