// diasss_amd/host/load_check.cpp -- host-only check of Util::LoadInputData (no GPU, no libdsss): prints, for every item
// the loader returned, its shape and exact f64 sum / first / last value in hex floats, one line each, so that the CPU
// test (tests/test_host_logic.py) can compare against the arrays the files were written from.
#include <cstdio>
#include "util.h"

using namespace Diasss;

static void line(const char* kind, size_t i, int rows, int cols, const double* v, size_t n)
{
    long double s = 0;
    for (size_t k = 0; k < n; ++k) s += v[k];
    std::printf("%s %zu %d %d %a %a %a\n", kind, i, rows, cols, (double)s, n ? v[0] : 0.0, n ? v[n - 1] : 0.0);
}

int main(int argc, char** argv)
{
    if (argc < 6) { std::printf("usage: load_check IMAGE_DIR POSE_DIR ALT_DIR GRANGE_DIR ANNO_DIR\n"); return 2; }
    std::vector<cv::Mat> I, P, A; std::vector<std::vector<double>> al, gr;
    Util::LoadInputData(argv[1], argv[2], argv[3], argv[4], argv[5], I, P, al, gr, A);
    for (size_t i = 0; i < I.size(); ++i) line("@img", i, I[i].rows, I[i].cols, I[i].ptr<double>(), (size_t)I[i].rows * I[i].cols);
    for (size_t i = 0; i < P.size(); ++i) line("@pose", i, P[i].rows, P[i].cols, P[i].ptr<double>(), (size_t)P[i].rows * P[i].cols);
    for (size_t i = 0; i < al.size(); ++i) line("@alt", i, (int)al[i].size(), 1, al[i].data(), al[i].size());
    for (size_t i = 0; i < gr.size(); ++i) line("@gr", i, (int)gr[i].size(), 1, gr[i].data(), gr[i].size());
    for (size_t i = 0; i < A.size(); ++i) {
        long long s = 0;
        for (int k = 0; k < A[i].rows * A[i].cols; ++k) s += A[i].ptr<int32_t>()[k];
        std::printf("@anno %zu %d %d %lld type %d\n", i, A[i].rows, A[i].cols, s, A[i].type());
    }
    return 0;
}
