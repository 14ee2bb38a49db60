// diasss_amd/host/frame.h -- drop-in for /root/reference/src/core/frame.h:13-55 (same class, constructor signature and
// public data members); the work happens in libdsss.so (dsss_frame_set + dsss_extract).
#ifndef FRAME_H
#define FRAME_H

#include <vector>
#include "cvlite.h"

namespace Diasss
{

    class Frame
    {

    public:

        // Constructor (frame.h:19-20): id, CV_64F NxM waterfall, CV_64F Nx6 DR poses, altitudes[N], ground ranges[M/2], annotations
        Frame(const int &id, const cv::Mat &mImg, const cv::Mat &mPose, const std::vector<double> &vAltt,
              const std::vector<double> &vGrange, const cv::Mat &mAnno);

        // the reference's helper members (frame.h:23-27); here they read back what the device computed
        cv::Mat GetNormalizeSSS(const cv::Mat &sss_raw_img);
        cv::Mat GetFilteredMask(const cv::Mat &sss_raw_img);
        void DetectFeature(const cv::Mat &img, const cv::Mat &mask, std::vector<cv::KeyPoint> &kps, cv::Mat &dst);
        std::vector<cv::Mat> GetGeoImg(const int &row, const int &col, const cv::Mat &pose, const std::vector<double> &g_range,
                                       const std::vector<double> &tf_stb, const std::vector<double> &tf_port);

        // Initialization items
        int img_id;
        cv::Mat anno_kps;
        cv::Mat raw_img;
        cv::Mat dr_poses;
        std::vector<double> altitudes;
        std::vector<double> ground_ranges;
        std::vector<double> tf_stb;
        std::vector<double> tf_port;

        // Produced items
        cv::Mat norm_img;              // fetched lazily: FetchImages()
        cv::Mat flt_mask;              // fetched lazily: FetchImages()
        std::vector<cv::Mat> geo_img;  // COMPACT: two 1x2 CV_64F mats [min,max] of x and y -- every consumer of the full
                                       // N x M geo image in the reference only takes its extremes or samples it at keypoints
        std::vector<cv::KeyPoint> kps;
        cv::Mat dst;                   // K x 32 CV_8U (ORB rows), or K x 128 CV_32F when Frame::USE_SIFT: what the SIFT call site of
                                       // ORBextractor.cpp:1098 was meant to leave here

        // The reference hard-codes its descriptor: the live call is the SIFT one (ORBextractor.cpp:1097-1098), whose output is lost.
        // false (default): the ORB configuration; true: ORB keypoints + the 128-float rows (DSSS_DESC_SIFT128).  Set before constructing frames.
        static bool USE_SIFT;
        cv::Mat corres_kps;            // rows: frame_id, ref_frame_id, kp_y, kp_x, kp_ref_y, kp_ref_x (CV_64F)
        cv::Mat est_poses;

        void FetchImages();            // norm_img + flt_mask from the device (2 x N*M bytes over PCIe, off the hot path)
        void FetchGeoImg();            // geo_img as the reference holds it: the full N x M pair (frame.cpp:126-165), 16 N M bytes over PCIe;
                                       // Util::ComputeIntersection gives the same overlap on either form (it takes the extremes)
    };

}

#endif
