// common.hpp -- internal types shared by the host side of libookiedokie_amd.
#pragma once

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/ookiedokie_amd.h"

namespace ookd {

// Experiment / test hooks read from the environment only exist for a process that asks for
// them with OOKD_DEVELOPER=1: a normal run of the library never changes behaviour because of a stray variable.
// (OOKD_DEBUG* diagnostics print, they do not change results, and stay plain getenv.)
inline const char *dev_getenv(const char *name) {
    const char *d = std::getenv("OOKD_DEVELOPER");
    return (d && d[0] && d[0] != '0') ? std::getenv(name) : nullptr;
}

// Thread-local "last error" text (the reference prints through log_error).
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
void clear_error();

struct FilterStage {
    uint32_t decimation = 1;
    std::vector<float> taps;
};

struct FieldEnum {
    std::string name;
    uint64_t value = 0;
};

// One entry of the device's "fields" array (src/device.c:229-424).
struct FieldDesc {
    std::string name;
    std::string default_value;
    int start_bit = 0;
    int end_bit = 0;
    int endianness = 0;   // 1 big, 2 little
    int format = 0;       // 1 hex, 2 unsigned dec, 3 sign-mag, 4 two's c, 5 float, 6 enum
    float offset = 0.0f;
    float scaling = 0.0f;
    std::vector<FieldEnum> enums;
};

}  // namespace ookd

struct ookd_filter {
    std::vector<ookd::FilterStage> stages;
    uint32_t total_decimation = 1;
};

struct ookd_device {
    std::string name;
    std::string description;
    uint32_t num_bits = 0;
    uint32_t sample_rate = 0;
    std::vector<std::string> state_names;
    std::vector<uint64_t> state_duration_us, state_timeout_us;
    std::vector<uint32_t> trig_begin;
    std::vector<uint8_t> trig_cond, trig_action;
    std::vector<uint32_t> trig_next;
    std::vector<uint64_t> trig_duration_us;
    // integer sample-count tables
    std::vector<uint64_t> state_kmin, state_kmax, state_kto;
    std::vector<uint64_t> trig_kmin, trig_kmax;
    // formatter description (host side, "next" row f2)
    std::vector<ookd::FieldDesc> fields;
    int ts_mode = 0;      // 0 none, 1 unix, 2 unix-frac, 3 datetime-24, 4 datetime-ampm
};

namespace ookd {
// Fills the integer tables from the *_us fields; false + error on failure.
bool build_count_tables(ookd_device &d);
}
