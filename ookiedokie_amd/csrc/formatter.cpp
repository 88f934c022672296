// formatter.cpp -- payload bits <-> per-field text, and the rx printer.
//
// Host-side half of the receive path: what the
// reference does with a payload once the state machine has reported
// OUTPUT_READY -- formatter_data_to_keyval (src/formatter.c:715-739) and
// rx_print (src/ookiedokie.c:181-220) -- plus the inverse used on the tx side
// (formatter_default_data / formatter_keyval_to_data, formatter.c:793-846).
// Per message, a few fields: this is not GPU work and stays on the host.
//
// The reference's conversions lean on C behaviour that is formally undefined
// (`1 << n` for n >= 31, float -> integer casts out of range).  The text it
// prints is therefore "what gcc on x86-64 makes of it"; the helpers below
// spell that behaviour out so the result does not depend on our compiler.
#include <algorithm>
#include <cerrno>
#include <cinttypes>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <strings.h>
#include <sys/time.h>

#include "common.hpp"

namespace ookd {
namespace {

enum Fmt { kHex = 1, kUnsigned = 2, kSignMag = 3, kTwos = 4, kFloat = 5, kEnum = 6 };
enum Endian { kBig = 1, kLittle = 2 };
enum TsMode { kTsNone = 0, kTsUnix = 1, kTsUnixFrac = 2, kTsDate24 = 3, kTsDateAmPm = 4 };

constexpr size_t kValueChars = 80;      // char buf[80] in formatter_data_to_keyval
const char kTsKey[] = "Decode Timestamp";

struct EnumDef {
    std::string str;
    int64_t value;
};

struct Field {
    std::string name;
    unsigned start = 0, end = 0;
    int format = 0, endianness = 0;
    float scaling = 1.0f, offset = 0.0f;
    int64_t default_value = 0;
    std::vector<EnumDef> enums;
    unsigned width() const { return end - start + 1; }
    uint64_t mask() const { return width() < 64 ? (1ull << width()) - 1 : ~0ull; }
};

// `1 << n` with an int 1 and a run-time n, widened to 64 bits: x86 masks the
// shift count to 5 bits and the int result sign-extends.
int64_t int_one_shl(unsigned n) { return (int64_t)(int32_t)(1u << (n & 31u)); }
// `(1 << n) - 1` in int arithmetic (wraps at n = 31), widened.
int64_t int_one_shl_minus1(unsigned n) { return (int64_t)(int32_t)((1u << (n & 31u)) - 1u); }

// (int64_t) x for a float: cvttss2si, "integer indefinite" when out of range.
int64_t float_to_i64(float x) {
    if (!(x >= -9223372036854775808.0f && x < 9223372036854775808.0f)) return INT64_MIN;
    return (int64_t)x;
}
// (uint64_t) x for a float, gcc's x86-64 sequence.
uint64_t float_to_u64(float x) {
    if (x < 9223372036854775808.0f) return (uint64_t)float_to_i64(x);
    return (uint64_t)float_to_i64(x - 9223372036854775808.0f) ^ 0x8000000000000000ull;
}

// conversions.c:93-142: whole string, base auto-detected, errno clean.
bool parse_u64(const char *s, uint64_t &out) {
    errno = 0;
    char *end = nullptr;
    const unsigned long long v = strtoull(s, &end, 0);
    if (errno != 0 || end == s || *end != '\0') return false;
    out = v;
    return true;
}
bool parse_i64(const char *s, int64_t &out) {
    errno = 0;
    char *end = nullptr;
    const long v = strtol(s, &end, 0);
    if (errno != 0 || end == s || *end != '\0') return false;
    out = v;
    return true;
}
bool parse_double(const char *s, double &out) {
    errno = 0;
    char *end = nullptr;
    const double v = strtod(s, &end);
    if (errno != 0 || !(v >= -1.7976931348623157e308) || !(v <= 1.7976931348623157e308) || end == s ||
        *end != '\0') {
        return false;
    }
    out = v;
    return true;
}

// formatter.c:425-452: payload bit i is bit (i % 8) of byte (i / 8); a big
// endian field receives its first bit in its most significant position.
int64_t extract(const Field &f, const uint8_t *data) {
    uint64_t v = 0;
    const unsigned w = f.width();
    for (unsigned n = 0; n < w; n++) {
        const unsigned i = f.start + n;
        const uint64_t bit = (data[i >> 3] >> (i & 7u)) & 1u;
        v |= bit << (f.endianness == kBig ? w - 1 - n : n);
    }
    return (int64_t)v;
}

// formatter.c:454-573, one case per format.
void value_to_text(const Field &f, int64_t value, char *str) {
    const unsigned w = f.width();
    str[0] = '\0';
    switch (f.format) {
        case kHex: {
            const uint64_t t = float_to_u64((float)(uint64_t)value * f.scaling + f.offset);
            // :471-487 -- up to 32 bits real hex digits, beyond that the
            // reference prints DECIMAL digits after "0x" (PRIu64)
            if (w <= 8) snprintf(str, kValueChars, "0x%02x", (unsigned)(uint8_t)t);
            else if (w <= 16) snprintf(str, kValueChars, "0x%02x", (unsigned)(uint16_t)t);
            else if (w <= 24) snprintf(str, kValueChars, "0x%06x", (uint32_t)t);
            else if (w <= 32) snprintf(str, kValueChars, "0x%08x", (uint32_t)t);
            else if (w <= 40) snprintf(str, kValueChars, "0x%010" PRIu64, t);
            else if (w <= 48) snprintf(str, kValueChars, "0x%012" PRIu64, t);
            else if (w <= 56) snprintf(str, kValueChars, "0x%014" PRIu64, t);
            else snprintf(str, kValueChars, "0x%016" PRIu64, t);
            break;
        }
        case kUnsigned: {
            const uint64_t t = float_to_u64((float)(uint64_t)value * f.scaling + f.offset);
            snprintf(str, kValueChars, "%" PRIu64, t);
            break;
        }
        case kTwos: {
            const bool neg = (value & int_one_shl(w - 1)) != 0;
            if (neg) value = (int64_t)(((uint64_t)~value + 1u) & f.mask());
            int64_t t = neg ? (int64_t)(0u - (uint64_t)value) : value;
            t = float_to_i64((float)t * f.scaling + f.offset);
            snprintf(str, kValueChars, "%" PRIi64, t);
            break;
        }
        case kSignMag: {
            const bool neg = ((uint64_t)value & (uint64_t)int_one_shl(w - 1)) != 0;
            int64_t t = (int64_t)((uint64_t)value & (uint64_t)int_one_shl_minus1(w - 1));
            if (neg) t = (int64_t)(0u - (uint64_t)t);
            t = float_to_i64((float)t * f.scaling + f.offset);
            snprintf(str, kValueChars, "%" PRIi64, t);
            break;
        }
        case kFloat: {
            float scaling = f.scaling;
            if ((value & int_one_shl(w - 1)) != 0) {
                value = (int64_t)(((uint64_t)~value + 1u) & f.mask());
                scaling = -scaling;
            }
            const float t = (float)value * scaling + f.offset;      // spt_to_float, spt.h:81-84
            snprintf(str, kValueChars, "%1.3f", (double)t);
            break;
        }
        case kEnum: {
            for (const EnumDef &e : f.enums) {
                if (e.value == value) {
                    snprintf(str, kValueChars, "%s", e.str.c_str());
                    return;
                }
            }
            snprintf(str, kValueChars, "0x%" PRIx64, (uint64_t)value);
            break;
        }
        default:
            break;
    }
}

// formatter.c:140-257: the text of a parameter / default -> field bits.
bool text_to_value(const Field &f, const char *str, int64_t &out) {
    const unsigned w = f.width();
    int64_t value = 0;
    switch (f.format) {
        case kHex:
        case kUnsigned: {
            uint64_t t;
            if (!parse_u64(str, t)) goto invalid;
            value = (int64_t)float_to_u64(((float)t - f.offset) / f.scaling);
            break;
        }
        case kTwos: {
            int64_t t;
            if (!parse_i64(str, t)) goto invalid;
            t = float_to_i64(((float)t - f.offset) / f.scaling);
            value = (int64_t)((uint64_t)t & f.mask());
            break;
        }
        case kSignMag: {
            int64_t t;
            if (!parse_i64(str, t)) goto invalid;
            const bool negative = t < 0;
            t = float_to_i64(((float)t - f.offset) / f.scaling);
            t &= int_one_shl_minus1(w - 1);
            if (negative) t |= int_one_shl(w - 1);
            value = t;
            break;
        }
        case kFloat: {
            double d;
            if (!parse_double(str, d)) goto invalid;
            const float t = (float)d;
            value = float_to_i64((t - f.offset) / f.scaling);       // spt_from_float, spt.h:58-62
            value = (int64_t)((uint64_t)value & f.mask());
            break;
        }
        case kEnum: {
            bool found = false;
            for (const EnumDef &e : f.enums) {
                if (!strcasecmp(str, e.str.c_str())) {
                    value = e.value;
                    found = true;
                    break;
                }
            }
            if (!found) {
                uint64_t t;
                if (!parse_u64(str, t)) goto invalid;
                value = (int64_t)t;
            }
            break;
        }
        default:
            set_error("Bug: Invalid field format: %d", f.format);
            return false;
    }
    if (((uint64_t)value & f.mask()) != (uint64_t)value) {
        set_error("Value is too large for field \"%s\": %s", f.name.c_str(), str);
        return false;
    }
    out = value;
    return true;
invalid:
    set_error("Invalid value for field \"%s\": %s", f.name.c_str(), str);
    return false;
}

// formatter.c:755-786.  `input_bits & (1 << src_bit)` is int arithmetic in the
// reference: from source bit 31 upwards the test is not the bit one expects.
void deposit(const Field &f, uint64_t bits, uint8_t *data) {
    const unsigned w = f.width();
    for (unsigned n = 0; n < w; n++) {
        const unsigned i = f.start + n;
        const unsigned src = f.endianness == kBig ? w - 1 - n : n;
        if (bits & (uint64_t)int_one_shl(src)) {
            data[i >> 3] |= (uint8_t)(1u << (i & 7u));
        } else {
            data[i >> 3] &= (uint8_t)~(1u << (i & 7u));
        }
    }
}

struct TextOut {            // snprintf-like sink: counts everything, stores what fits
    char *buf;
    size_t cap;
    size_t len = 0;
    void put(const char *s, size_t n) {
        if (buf && len < cap) memcpy(buf + len, s, std::min(n, cap - len));
        len += n;
    }
    void put(const char *s) { put(s, strlen(s)); }
    void put(char c) { put(&c, 1); }
    void finish() {
        if (buf && cap) buf[std::min(len, cap - 1)] = '\0';
    }
};

struct KeyVal {
    std::string key, value;
};

// formatter.c:602-713.  unix: the reference formats nothing into its buffer
// for the integer mode (it only adds 0.5 to a local); the evident intent --
// rounded integer seconds -- is what is produced here.
bool timestamp_text(int mode, std::string &out) {
    char buf[80];
    if (mode == kTsUnix || mode == kTsUnixFrac) {
        struct timeval tv;
        if (gettimeofday(&tv, nullptr) != 0) return false;
        const double ts = (double)tv.tv_sec + (double)tv.tv_usec / 1000000.0;
        if (mode == kTsUnixFrac) snprintf(buf, sizeof(buf), "%f", ts);
        else snprintf(buf, sizeof(buf), "%" PRIu64, (uint64_t)(ts + 0.5));
        out = buf;
        return true;
    }
    const time_t t = time(nullptr);
    struct tm tmv;
    if (!localtime_r(&t, &tmv)) return false;
    const size_t n = strftime(buf, sizeof(buf),
                              mode == kTsDateAmPm ? "%Y-%m-%d %I:%M:%S %p" : "%Y-%m-%d %H:%M:%S", &tmv);
    if (n == 0) return false;
    out = buf;
    return true;
}

}  // namespace
}  // namespace ookd

using namespace ookd;

struct ookd_formatter {
    std::vector<Field> fields;
    uint32_t num_bits = 0;
    int ts_mode = 0;

    const Field *find(const char *name) const {     // first case-insensitive match
        for (const Field &f : fields) {
            if (!strcasecmp(f.name.c_str(), name)) return &f;
        }
        return nullptr;
    }

    // formatter_data_to_keyval: [timestamp,] one pair per field, in file order
    void append_keyvals(const uint8_t *payload, std::vector<KeyVal> &kv) const {
        if (ts_mode != kTsNone) {
            std::string ts;
            if (timestamp_text(ts_mode, ts)) kv.push_back({kTsKey, ts});
        }
        char buf[kValueChars];
        for (const Field &f : fields) {
            value_to_text(f, extract(f, payload), buf);
            kv.push_back({f.name, buf});
        }
    }
};

namespace {

// rx_print (ookiedokie.c:181-220) over one keyval list = everything decoded
// from one buffer.
void print_record(int rx_fmt, int *first_print, const std::vector<KeyVal> &kv, TextOut &o) {
    const size_t n = kv.size();
    if (n == 0) return;
    if (rx_fmt == OOKD_RX_FMT_CSV) {
        if (first_print && *first_print) {
            for (size_t i = 0; i < n; i++) {
                o.put(kv[i].key.c_str());
                o.put(i + 1 < n ? ',' : '\n');
            }
            *first_print = 0;
        }
        for (size_t i = 0; i < n; i++) {
            o.put(kv[i].value.c_str());
            o.put(i + 1 < n ? ',' : '\n');
        }
    } else {
        char line[256];
        for (size_t i = 0; i < n; i++) {
            const int m = snprintf(line, sizeof(line), "%20s : ", kv[i].key.c_str());
            if (m < (int)sizeof(line)) {
                o.put(line, (size_t)m);
            } else {            // key longer than the scratch line: emit unpadded
                o.put(kv[i].key.c_str());
                o.put(" : ");
            }
            o.put(kv[i].value.c_str());
            o.put('\n');
        }
        o.put('\n');
    }
}

}  // namespace

extern "C" {

ookd_formatter *ookd_formatter_create(const ookd_device *device) {
    clear_error();
    if (!device) {
        set_error("ookd_formatter_create: null device");
        return nullptr;
    }
    // formatter_init (formatter.c:75-138)
    if (device->fields.empty()) {
        set_error("Formatter must be initialized for one or more fields.");
        return nullptr;
    }
    if (device->num_bits == 0) {
        set_error("Formatter cannot be initialized for 0 bits.");
        return nullptr;
    }
    std::unique_ptr<ookd_formatter> f(new ookd_formatter());
    f->num_bits = device->num_bits;
    f->ts_mode = device->ts_mode;
    const unsigned data_bits = ((device->num_bits + 7) / 8) * 8;    // device.c:551
    for (const FieldDesc &d : device->fields) {
        Field fld;
        fld.name = d.name;
        if (d.start_bit < 0 || d.end_bit < 0) {
            set_error("Invalid %s bit: %d", d.start_bit < 0 ? "start" : "end",
                      d.start_bit < 0 ? d.start_bit : d.end_bit);
            return nullptr;
        }
        fld.start = (unsigned)d.start_bit;
        fld.end = (unsigned)d.end_bit;
        if (fld.end < fld.start) {
            set_error("End bit must be >= start bit");
            return nullptr;
        }
        if (fld.end - fld.start + 1 > 64) {
            set_error("Fields larger than 64-bits are not currently supported.");
            return nullptr;
        }
        if (fld.end >= data_bits) {
            // the reference would read past its (num_bits + 7) / 8 byte buffer here
            set_error("Field \"%s\" ends at bit %u, beyond the device's %u-bit message", d.name.c_str(),
                      fld.end, device->num_bits);
            return nullptr;
        }
        fld.format = d.format;
        fld.endianness = d.endianness;
        fld.scaling = d.scaling == 0.0f ? 1.0f : d.scaling;        // formatter.c:283
        fld.offset = d.offset;
        f->fields.push_back(std::move(fld));
        // enums are attached to the FIRST field of that name (formatter_add_field_enum
        // looks the field up by name, formatter.c:352-358), duplicates refused (:380-383)
        Field *target = nullptr;
        for (Field &g : f->fields) {
            if (!strcasecmp(g.name.c_str(), d.name.c_str())) {
                target = &g;
                break;
            }
        }
        // ... so a later field that shares a name has no slots of its own to fill
        size_t room = target == &f->fields.back() ? d.enums.size() : 0;
        for (const FieldEnum &e : d.enums) {
            for (const EnumDef &have : target->enums) {
                if (!strcasecmp(have.str.c_str(), e.name.c_str())) {
                    set_error("Error: Duplicate enumeration name (%s)", e.name.c_str());
                    return nullptr;
                }
            }
            if (room == 0) {
                set_error("Error: Enum list size exceeded.");
                return nullptr;
            }
            room--;
            target->enums.push_back({e.name, (int64_t)e.value});
        }
        // formatter_set_field_default: again the first field of that name
        int64_t dv = 0;
        if (!text_to_value(*target, d.default_value.c_str(), dv)) {
            const std::string why = ookd_last_error();
            set_error("Invalid default value for field \"%s\": %s (%s)", target->name.c_str(),
                      d.default_value.c_str(), why.c_str());
            return nullptr;
        }
        target->default_value = dv;
    }
    return f.release();
}

void ookd_formatter_free(ookd_formatter *f) { delete f; }

uint32_t ookd_formatter_num_fields(const ookd_formatter *f) { return f ? (uint32_t)f->fields.size() : 0; }

const char *ookd_formatter_field_name(const ookd_formatter *f, uint32_t field) {
    return (f && field < f->fields.size()) ? f->fields[field].name.c_str() : nullptr;
}

int ookd_formatter_ts_mode(const ookd_formatter *f) { return f ? f->ts_mode : 0; }

int ookd_formatter_field_to_str(const ookd_formatter *f, uint32_t field, const uint8_t *payload, char *out,
                                size_t capacity) {
    clear_error();
    if (!f || field >= f->fields.size() || !payload || !out || capacity == 0) {
        set_error("ookd_formatter_field_to_str: bad argument");
        return OOKD_ERR_ARG;
    }
    char buf[kValueChars];
    value_to_text(f->fields[field], extract(f->fields[field], payload), buf);
    snprintf(out, capacity, "%s", buf);
    return OOKD_OK;
}

int ookd_formatter_default_data(const ookd_formatter *f, uint8_t *payload, size_t len) {
    clear_error();
    if (!f || !payload || len * 8 < ((f->num_bits + 7) / 8) * 8) {
        set_error("ookd_formatter_default_data: bad argument");
        return OOKD_ERR_ARG;
    }
    for (const Field &fld : f->fields) deposit(fld, (uint64_t)fld.default_value, payload);
    return OOKD_OK;
}

int ookd_formatter_set_field(const ookd_formatter *f, const char *name, const char *value, uint8_t *payload,
                             size_t len) {
    clear_error();
    if (!f || !name || !value || !payload || len * 8 < ((f->num_bits + 7) / 8) * 8) {
        set_error("ookd_formatter_set_field: bad argument");
        return OOKD_ERR_ARG;
    }
    const Field *fld = f->find(name);
    if (!fld) {
        set_error("Invalid parameter name: %s", name);
        return OOKD_ERR_ARG;
    }
    int64_t v = 0;
    if (!text_to_value(*fld, value, v)) return OOKD_ERR_ARG;
    deposit(*fld, (uint64_t)v, payload);
    return OOKD_OK;
}

size_t ookd_print_record(const ookd_formatter *f, int rx_fmt, int *first_print, const uint8_t *const *payloads,
                         size_t count, char *out, size_t capacity) {
    clear_error();
    TextOut o{out, capacity};
    if (!f || (count && !payloads) || (rx_fmt != OOKD_RX_FMT_PRETTY && rx_fmt != OOKD_RX_FMT_CSV)) {
        set_error("ookd_print_record: bad argument");
        o.finish();
        return 0;
    }
    std::vector<KeyVal> kv;
    for (size_t i = 0; i < count; i++) f->append_keyvals(payloads[i], kv);
    print_record(rx_fmt, first_print, kv, o);
    o.finish();
    return o.len;
}

size_t ookd_print_messages(const ookd_formatter *f, int rx_fmt, int *first_print, const ookd_message *msgs,
                           uint64_t num_messages, uint32_t samples_per_buffer, uint32_t total_decimation,
                           char *out, size_t capacity) {
    clear_error();
    TextOut o{out, capacity};
    if (!f || (num_messages && !msgs) || samples_per_buffer == 0 || total_decimation == 0 ||
        (rx_fmt != OOKD_RX_FMT_PRETTY && rx_fmt != OOKD_RX_FMT_CSV)) {
        set_error("ookd_print_messages: bad argument");
        o.finish();
        return 0;
    }
    // device_process (device.c:634-658) collects every message of one sdr_rx
    // buffer in one keyval list, and the rx loop prints that list once
    // (ookiedokie.c:279-286).  Decimated sample j leaves the filter while
    // input D*(j+1)-1 is consumed, i.e. in buffer ceil((j+1)*D / spb) - 1.
    std::vector<KeyVal> kv;
    uint64_t group_cap = 0, group_buf = 0;
    bool open = false;
    for (uint64_t i = 0; i < num_messages; i++) {
        const unsigned __int128 in_end = (unsigned __int128)(msgs[i].sample + 1) * total_decimation;
        const uint64_t buf = (uint64_t)((in_end + samples_per_buffer - 1) / samples_per_buffer) - 1;
        if (open && (msgs[i].capture != group_cap || buf != group_buf)) {
            print_record(rx_fmt, first_print, kv, o);
            kv.clear();
        }
        open = true;
        group_cap = msgs[i].capture;
        group_buf = buf;
        f->append_keyvals(msgs[i].payload, kv);
    }
    if (open) print_record(rx_fmt, first_print, kv, o);
    o.finish();
    return o.len;
}

}  // extern "C"
