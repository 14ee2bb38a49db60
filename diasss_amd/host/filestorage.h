// diasss_amd/host/filestorage.h -- reads ONE matrix node out of an OpenCV FileStorage file (XML or YAML), which is all
// Util::LoadInputData needs (/root/reference/src/util/util.cpp:84-86,107-109,186-188: `fs["ct_img"] >> mat`).
// OpenCV itself is not in this image.  Supported element types: d (f64), f (f32, widened to f64), i (int32), u (uint8);
// single channel.  Format as written by cv::FileStorage:
//   XML :  <name type_id="opencv-matrix"><rows>R</rows><cols>C</cols><dt>d</dt><data> v v v ... </data></name>
//   YAML:  name: !!opencv-matrix\n   rows: R\n   cols: C\n   dt: d\n   data: [ v, v, ... ]
#ifndef DSSS_FILESTORAGE_H
#define DSSS_FILESTORAGE_H
#include <string>
#include "cvlite.h"

namespace Diasss {
// returns true and fills `out` (CV_64F / CV_32S / CV_8U); false with a message in `err` otherwise
bool ReadStorageMatrix(const std::string& path, const std::string& node, cv::Mat& out, std::string* err = nullptr);
}
#endif
