// diasss_amd/host/filestorage.cpp -- see filestorage.h
#include "filestorage.h"
#include <cctype>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace Diasss {
namespace {

bool fail(std::string* err, const std::string& m) { if (err) *err = m; return false; }

// text between the first `open` and the following `close` at or after `from`; npos-safe
bool between(const std::string& s, size_t from, const std::string& open, const std::string& close, std::string& out, size_t* end = nullptr)
{
    const size_t a = s.find(open, from);
    if (a == std::string::npos) return false;
    const size_t b = s.find(close, a + open.size());
    if (b == std::string::npos) return false;
    out = s.substr(a + open.size(), b - a - open.size());
    if (end) *end = b + close.size();
    return true;
}

std::string trim(const std::string& s)
{
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

// numbers separated by white space and / or commas
bool parse_values(const std::string& text, char dt, cv::Mat& out, std::string* err)
{
    const size_t need = (size_t)out.rows * out.cols;
    const char* p = text.c_str();
    size_t n = 0;
    while (*p) {
        while (*p && (std::isspace((unsigned char)*p) || *p == ',')) ++p;
        if (!*p) break;
        char* e = nullptr;
        if (n >= need) return fail(err, "more values than rows*cols");
        if (dt == 'd' || dt == 'f') {
            double v;
            if (p[0] == '.' && (p[1] == 'N' || p[1] == 'n')) { v = std::strtod("nan", nullptr); e = const_cast<char*>(p) + 4; }       // .Nan
            else if ((p[0] == '.' || ((p[0] == '-' || p[0] == '+') && p[1] == '.')) && (p[1] == 'I' || p[2] == 'I' || p[1] == 'i' || p[2] == 'i')) {
                v = p[0] == '-' ? -std::strtod("inf", nullptr) : std::strtod("inf", nullptr); e = const_cast<char*>(p) + (p[0] == '.' ? 4 : 5);
            } else v = std::strtod(p, &e);
            if (e == p) return fail(err, "bad number in <data>");
            reinterpret_cast<double*>(out.data())[n++] = v;
        } else {
            const long v = std::strtol(p, &e, 10);
            if (e == p) return fail(err, "bad integer in <data>");
            if (dt == 'i') reinterpret_cast<int32_t*>(out.data())[n++] = (int32_t)v; else out.data()[n++] = (uint8_t)v;
        }
        p = e;
    }
    if (n != need) return fail(err, "fewer values than rows*cols");
    return true;
}

bool make(int rows, int cols, const std::string& dts, const std::string& data, cv::Mat& out, std::string* err)
{
    std::string d = trim(dts);
    if (!d.empty() && d[0] == '"') d = trim(d.substr(1, d.size() > 1 ? d.size() - 2 : 0));
    if (d.size() != 1) return fail(err, "only single-channel matrices are supported (dt '" + d + "')");
    const char dt = d[0];
    if (rows < 0 || cols < 0) return fail(err, "negative size");
    if (dt == 'd' || dt == 'f') out.create(rows, cols, CV_64F);
    else if (dt == 'i') out.create(rows, cols, CV_32S);
    else if (dt == 'u') out.create(rows, cols, CV_8U);
    else return fail(err, std::string("unsupported dt '") + dt + "'");
    return parse_values(data, dt, out, err);
}

bool read_xml(const std::string& s, const std::string& node, cv::Mat& out, std::string* err)
{
    const size_t a = s.find("<" + node);
    if (a == std::string::npos) return fail(err, "node <" + node + "> not found");
    std::string rows, cols, dt, data;
    if (!between(s, a, "<rows>", "</rows>", rows) || !between(s, a, "<cols>", "</cols>", cols) || !between(s, a, "<dt>", "</dt>", dt) ||
        !between(s, a, "<data>", "</data>", data)) return fail(err, "node <" + node + "> is not an opencv-matrix");
    return make(std::atoi(rows.c_str()), std::atoi(cols.c_str()), dt, data, out, err);
}

bool yaml_field(const std::string& s, size_t from, const std::string& key, std::string& out)
{
    const size_t a = s.find(key + ":", from);
    if (a == std::string::npos) return false;
    const size_t b = s.find('\n', a);
    out = trim(s.substr(a + key.size() + 1, (b == std::string::npos ? s.size() : b) - a - key.size() - 1));
    return true;
}

bool read_yaml(const std::string& s, const std::string& node, cv::Mat& out, std::string* err)
{
    size_t a = s.find("\n" + node + ":");
    if (a == std::string::npos) { if (s.compare(0, node.size() + 1, node + ":") == 0) a = 0; else return fail(err, "node " + node + " not found"); }
    std::string rows, cols, dt, data;
    if (!yaml_field(s, a, "rows", rows) || !yaml_field(s, a, "cols", cols) || !yaml_field(s, a, "dt", dt)) return fail(err, "node " + node + " is not an opencv-matrix");
    const size_t d0 = s.find("data:", a);
    if (d0 == std::string::npos || !between(s, d0, "[", "]", data)) return fail(err, "node " + node + " has no data list");
    return make(std::atoi(rows.c_str()), std::atoi(cols.c_str()), dt, data, out, err);
}

} // namespace

bool ReadStorageMatrix(const std::string& path, const std::string& node, cv::Mat& out, std::string* err)
{
    std::ifstream f(path.c_str(), std::ios::binary);
    if (!f) return fail(err, "cannot open " + path);
    std::stringstream ss; ss << f.rdbuf();
    const std::string s = ss.str();
    size_t i = 0;
    while (i < s.size() && std::isspace((unsigned char)s[i])) ++i;
    if (s.compare(i, 5, "<?xml") == 0 || s.compare(i, 15, "<opencv_storage") == 0) return read_xml(s, node, out, err);
    if (s.compare(i, 5, "%YAML") == 0 || s.find(node + ":") != std::string::npos) return read_yaml(s, node, out, err);
    return fail(err, path + ": neither OpenCV XML nor YAML storage");
}

} // namespace Diasss
