// diasss_amd/host/optimizer.h -- drop-in for the in-scope part of /root/reference/src/core/optimizer.h:31-67.
// GTSAM's Vector7 / Vector6 / Pose3 appear in the reference's public signatures (optimizer.h:12-25); GTSAM is not a
// dependency here, so POD equivalents carry the same names inside namespace Diasss.
#ifndef OPTIMIZER_H
#define OPTIMIZER_H

#include <array>
#include <tuple>
#include <vector>
#include "frame.h"

namespace Diasss
{

    typedef std::array<double, 7> Vector7;     // [ping_s, bin_s, slant_s, ping_t, bin_t, slant_t, depth]
    typedef std::array<double, 6> Vector6;     // tangent order [rot(3), trans(3)]
    struct Pose3 { double R[9]; double t[3]; double x() const { return t[0]; } double y() const { return t[1]; } double z() const { return t[2]; } };

    class Optimizer
    {

    public:

        // the reference hard-codes these as locals (optimizer.cpp:26); USE_ANNO = 1 there consumes hand annotations,
        // the synthetic configs need the matcher's output, so the default here is 0
        static bool USE_ANNO;
        static bool ADD_LC;
        static bool ONLINE;      // frame-by-frame updates (the reference's iSAM2 loop, optimizer.cpp:134-272) instead of one batch solve
        static int ONLINE_WINDOW; // > 0 with ONLINE: an update solves only the last ONLINE_WINDOW frames, conditioned on the frozen rest (dsss_posegraph_update_window); the last update is global
        // the two annotation evaluators of EvaluateByAnnosAll; the reference hard-codes both to 0 (optimizer.cpp:1579)
        static bool EVAL_1;
        static bool EVAL_2;

        void static TrajOptimizationAll(std::vector<Frame> &AllFrames);                                   // optimizer.h:43

        std::vector<Vector7> static GetKpsPairs(const bool &USE_ANNO, const cv::Mat &kps, const int &id_s, const int &id_t,
                                         const std::vector<double> &alts_s, const std::vector<double> &gras_s,
                                         const std::vector<double> &alts_t, const std::vector<double> &gras_t);   // optimizer.h:45-47

        std::vector<std::tuple<Pose3,Vector6,double>> static LoopClosingTFs(const std::vector<Vector7> &kps_pairs,
                                                        const std::vector<double> &tf_stb, const std::vector<double> &tf_port,
                                                        const int &img_id_s, const int &img_id_t,
                                                        const std::vector<cv::Mat> &geo_s, const std::vector<cv::Mat> &geo_t,
                                                        const std::vector<double> &alts_s, const std::vector<double> &alts_t,
                                                        const std::vector<double> &gras_s, const std::vector<double> &gras_t,
                                                        const cv::Mat &dr_poses_s, const cv::Mat &dr_poses_t);      // optimizer.h:61-67

        // optimizer.h:56-59 (3-DoF LM on one landmark, both ping poses fixed; Ts_s / Ts_t must be the identity, frame.cpp:38-39)
        typedef std::array<double, 3> Point3;
        Point3 static TriangulateOneLandmark(const Vector7 &kps_pair, const Pose3 &Ts_s, const Pose3 &Ts_t,
                                             const Pose3 &Tp_s, const Pose3 &Tp_t, const Point3 &lm_ini);

        // optimizer.h:84-91 / optimizer.cpp:1568-1886: consistency of the annotated keypoint pairs under the dead-reckoning and the
        // estimated poses.  eval_2 (:1581-1762): landmark triangulated from both pings (TriangulateOneLandmark), slant-range and
        // plane error per pair; eval_1 (:1764-1883): distance of the two geo-referenced observations of the landmark.  One entry per
        // frame pair; the reference prints these figures ("Metric Statics", "Avg R and P", "LM Metric Statics", "Avg X,Y,NORM") and
        // writes the per-match values under ../result/{pr_errors,anno_errors}/ -- here under $DSSS_OUT_DIR or ../ when the folder exists.
        struct AnnoStats {
            int img_s = 0, img_t = 0, n = 0;
            double good_range = 0, good_plane = 0, range_dr = 0, range_est = 0, plane_dr = 0, plane_est = 0;          // eval_2 (percent, averages)
            double good_dist = 0, x_dr = 0, x_est = 0, y_dr = 0, y_est = 0, all_dr = 0, all_est = 0;                   // eval_1
        };
        std::vector<AnnoStats> static EvaluateByAnnosAll(const std::vector<double> &poses12, const std::vector<std::vector<int>> &unique_id,
                                                         const std::vector<Frame> &AllFrames,
                                                         const std::vector<std::vector<Vector7>> &kps_pairs_all,
                                                         const std::vector<std::pair<int,int>> &img_pairs_ids);

        // optimizer.h:73-74; poses12 = total x 12 (R row-major, t) instead of gtsam::Values
        void static SaveTrajactoryAll(const std::vector<double> &poses12, const std::vector<std::vector<int>> &unique_id,
                                      const std::vector<cv::Mat> &dr_poses_all);
    };

}

#endif // OPTIMIZER_H
