// diasss_amd/host/test_demo.cpp -- the driver loop of /root/reference/src/diasss2.cpp:83-101 against the drop-in
// classes.  Input parsing differs: Util::LoadInputData (OpenCV FileStorage XML, util.cpp:45-213) is out of scope, so
// frames come from the flat binary dumps written by tools/export_survey.py:
//     <dir>/frame_%03d.bin = int32 N, int32 M, f64 raw[N*M], f64 pose[N*6], f64 alt[N], f64 gr[M/2]
#include <cstdio>
#include <iostream>
#include <string>
#include "frame.h"
#include "FEAmatcher.h"
#include "optimizer.h"
#include "util.h"

using namespace Diasss;

int main(int argc, char** argv)
{
    if (argc < 2) { std::cout << "usage: test_demo <dir with frame_%03d.bin> [min_overlap]" << std::endl; return 0; }
    float MIN_OVERLAP = argc > 2 ? (float)atof(argv[2]) : 0.4f;                  // diasss2.cpp:28
    std::vector<Frame> test_frames;
    for (int i = 0;; ++i) {
        char path[512];
        snprintf(path, sizeof path, "%s/frame_%03d.bin", argv[1], i);
        FILE* f = fopen(path, "rb");
        if (!f) break;
        int N, M;
        if (fread(&N, 4, 1, f) != 1 || fread(&M, 4, 1, f) != 1) { fclose(f); break; }
        cv::Mat img(N, M, CV_64F), pose(N, 6, CV_64F), anno;
        std::vector<double> alt(N), gr(M / 2);
        size_t ok = fread(img.data(), 8, (size_t)N * M, f) + fread(pose.data(), 8, (size_t)N * 6, f) + fread(alt.data(), 8, N, f) + fread(gr.data(), 8, M / 2, f);
        fclose(f);
        if (ok != (size_t)N * M + (size_t)N * 6 + N + M / 2) { std::cout << "short read: " << path << std::endl; return 1; }
        test_frames.push_back(Frame(i, img, pose, alt, gr, anno));
        std::cout << "frame " << i << ": " << N << " x " << M << ", " << test_frames.back().kps.size() << " keypoints" << std::endl;
    }
    for (size_t i = 0; i < test_frames.size(); i++)
        for (size_t j = i + 1; j < test_frames.size(); j++) {
            float overlap_percentage = Util::ComputeIntersection(test_frames[i].geo_img, test_frames[j].geo_img);
            std::cout << "The OVERLAPPING RATE Between image " << i << " and " << j << " : " << overlap_percentage << " ..." << std::endl;
            if (overlap_percentage > MIN_OVERLAP) {
                FEAmatcher::RobustMatching(test_frames[i], test_frames[j]);
                std::cout << "  matches so far in frame " << i << ": " << test_frames[i].corres_kps.rows << std::endl;
            }
        }
    Optimizer::TrajOptimizationAll(test_frames);
    return 0;
}
