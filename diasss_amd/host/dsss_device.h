// diasss_amd/host/dsss_device.h -- process-wide dsss context shared by the value-type Frame objects
#pragma once
#include <stdexcept>
#include <string>
#include "../../include/dsss.h"

namespace Diasss {

struct Device {
    static dsss_ctx* ctx() {
        static Device d;
        return d.c_;
    }
    static void check(int rc, const char* what) {
        if (rc != DSSS_OK) throw std::runtime_error(std::string(what) + ": " + dsss_strerror(rc) + " (" + dsss_last_error(ctx()) + ")");
    }
    static int& max_frames() { static int n = 1024; return n; }
private:
    Device() {
        int rc = dsss_create(0, max_frames(), &c_);
        if (rc != DSSS_OK) throw std::runtime_error(std::string("dsss_create: ") + dsss_strerror(rc));   // no CPU fallback
    }
    ~Device() { dsss_destroy(c_); }
    dsss_ctx* c_ = nullptr;
};

} // namespace Diasss
