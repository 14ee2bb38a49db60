// diasss_amd/host/test_demo.cpp -- the driver loop of /root/reference/src/diasss2.cpp:83-101 against the drop-in
// classes.  Two input forms:
//   test_demo --image DIR --pose DIR --altitude DIR --groundrange DIR [--annotation DIR] [--min-overlap X] [--descriptor orb|orb-l2|sift] [--use-anno 0|1] [--add-lc 0|1] [--eval 0..3]
//       the reference's own layout (diasss2.cpp:33-66) through Util::LoadInputData: OpenCV FileStorage XML / YAML + txt
//   test_demo <dir with frame_%03d.bin> [min_overlap]
//       flat binary dumps written by tools/export_survey.py:
//       frame_%03d.bin = int32 N, int32 M, f64 raw[N*M], f64 pose[N*6], f64 alt[N], f64 gr[M/2]
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
    if (argc < 2) { std::cout << "usage: test_demo <dir with frame_%03d.bin> [min_overlap]  |  test_demo --image D --pose D --altitude D --groundrange D [--annotation D]" << std::endl; return 0; }
    float MIN_OVERLAP = 0.4f;                                                    // diasss2.cpp:28
    std::vector<Frame> test_frames;
    const bool folders = std::string(argv[1]).rfind("--", 0) == 0;
    if (folders) {
        std::string dI, dP, dA, dG, dN;
        for (int a = 1; a + 1 < argc; a += 2) {
            const std::string k = argv[a], v = argv[a + 1];
            if (k == "--image") dI = v; else if (k == "--pose") dP = v; else if (k == "--altitude") dA = v;
            else if (k == "--groundrange") dG = v; else if (k == "--annotation") dN = v; else if (k == "--min-overlap") MIN_OVERLAP = (float)atof(v.c_str());
            else if (k == "--descriptor") { const bool sift = v == "sift"; Frame::USE_SIFT = sift; FEAmatcher::USE_SIFT = sift ? 2 : (v == "orb-l2" ? 1 : 0); }      // orb (default) | orb-l2 (the shipped matcher branch on the ORB bytes) | sift (N4: 128-float rows + L2)
            else if (k == "--use-anno") Optimizer::USE_ANNO = atoi(v.c_str()) != 0;      // optimizer.cpp:26 hard-codes 1 (hand annotations); default here 0
            else if (k == "--add-lc") Optimizer::ADD_LC = atoi(v.c_str()) != 0;
            else if (k == "--online-window") Optimizer::ONLINE_WINDOW = atoi(v.c_str());      // with --online 1: updates solve the last N frames only (incremental), the last one is global
            else if (k == "--online") Optimizer::ONLINE = atoi(v.c_str()) != 0;      // frame-by-frame updates as the reference's iSAM2 loop does (default: one batch solve)
            else if (k == "--eval") { Optimizer::EVAL_1 = (atoi(v.c_str()) & 1) != 0; Optimizer::EVAL_2 = (atoi(v.c_str()) & 2) != 0; }      // optimizer.cpp:1579 hard-codes both off
        }
        if (dI.empty() || dP.empty() || dA.empty() || dG.empty()) { std::cout << "Please provide the image, pose, altitude and groundrange folders..." << std::endl; return 0; }
        std::vector<cv::Mat> vmImgs, vmPoses, vmAnnos; std::vector<std::vector<double>> vvAltts, vvGranges;
        Util::LoadInputData(dI, dP, dA, dG, dN, vmImgs, vmPoses, vvAltts, vvGranges, vmAnnos);
        if (vmPoses.size() != vmImgs.size() || vvAltts.size() != vmImgs.size() || vvGranges.size() != vmImgs.size()) { std::cout << "folder sizes differ" << std::endl; return 1; }
        for (size_t i = 0; i < vmImgs.size(); ++i) {                             // diasss2.cpp:83-86
            cv::Mat anno = i < vmAnnos.size() ? vmAnnos[i] : cv::Mat();
            test_frames.push_back(Frame((int)i, vmImgs[i], vmPoses[i], vvAltts[i], vvGranges[i], anno));
            std::cout << "frame " << i << ": " << vmImgs[i].rows << " x " << vmImgs[i].cols << ", " << test_frames.back().kps.size() << " keypoints" << std::endl;
        }
    } else if (argc > 2) MIN_OVERLAP = (float)atof(argv[2]);
    for (int i = 0; !folders; ++i) {
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
