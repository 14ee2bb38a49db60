// diasss_amd/host/cvlite.h -- the handful of cv:: members the drop-in API touches
// (/root/reference/src/core/frame.h:19-46, FEAmatcher.h:20-33, util.h:25-28).  OpenCV is not in this image; define
// DSSS_USE_OPENCV to compile the host mirror against the real headers instead.
#pragma once
#ifdef DSSS_USE_OPENCV
#include <opencv2/opencv.hpp>
#else
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32S 4
#define CV_32F 5
#define CV_64F 6

namespace cv {

struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };

struct KeyPoint {
    Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0;
    KeyPoint() {}
    KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0) : pt(x, y), size(s), angle(a), response(r), octave(o) {}
};

// row-major, ref-counted, single channel
class Mat {
public:
    int rows = 0, cols = 0;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    static Mat zeros(int r, int c, int type) { Mat m(r, c, type); std::memset(m.data(), 0, m.bytes()); return m; }
    void create(int r, int c, int type) { rows = r; cols = c; type_ = type; buf_ = std::make_shared<std::vector<uint8_t>>((size_t)r * c * esz()); }
    int type() const { return type_; }
    bool empty() const { return rows == 0 || cols == 0; }
    size_t esz() const { return type_ == CV_8U ? 1 : ((type_ == CV_32S || type_ == CV_32F) ? 4 : 8); }
    size_t bytes() const { return (size_t)rows * cols * esz(); }
    uint8_t* data() { return buf_ ? buf_->data() : nullptr; }
    const uint8_t* data() const { return buf_ ? buf_->data() : nullptr; }
    template <typename T> T& at(int i, int j) { return reinterpret_cast<T*>(data())[(size_t)i * cols + j]; }
    template <typename T> const T& at(int i, int j) const { return reinterpret_cast<const T*>(data())[(size_t)i * cols + j]; }
    template <typename T> T* ptr(int i = 0) { return reinterpret_cast<T*>(data()) + (size_t)i * cols; }
    template <typename T> const T* ptr(int i = 0) const { return reinterpret_cast<const T*>(data()) + (size_t)i * cols; }
    Mat row(int i) const { Mat m(1, cols, type_); std::memcpy(m.data(), data() + (size_t)i * cols * esz(), (size_t)cols * esz()); return m; }
    Mat clone() const { Mat m(rows, cols, type_); if (bytes()) std::memcpy(m.data(), data(), bytes()); return m; }
    // append rows (cv::Mat::push_back); an empty Mat adopts the type and width of the first row pushed
    void push_back(const Mat& r) {
        if (empty()) { type_ = r.type_; cols = r.cols; rows = 0; buf_ = std::make_shared<std::vector<uint8_t>>(); }
        else if (buf_.use_count() > 1) buf_ = std::make_shared<std::vector<uint8_t>>(*buf_);
        buf_->insert(buf_->end(), r.data(), r.data() + r.bytes());
        rows += r.rows;
    }
private:
    int type_ = CV_8U;
    std::shared_ptr<std::vector<uint8_t>> buf_;
};

} // namespace cv
#endif
