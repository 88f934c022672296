// loaders.cpp -- host-side construction of filters and devices.
//
// Mirrors, for the rx direction, what fir_init (src/fir.c:68-249) and
// device_init (src/device.c:574-632, with add_state :76-193,
// create_state_machine :195-227, add_field :229-424) build from the JSON
// files, which are consumed unchanged.  File lookup (src/find.c) is the
// host's business: these take explicit paths.
#include <strings.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <fstream>
#include <sstream>

#include "common.hpp"
#include "json_min.hpp"

namespace ookd {

static thread_local std::string g_error;

void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
}

void clear_error() { g_error.clear(); }

static bool read_file(const char *path, std::string &out) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    std::ostringstream ss;
    ss << in.rdbuf();
    out = ss.str();
    return true;
}

static bool parse_file(const char *path, json::Value &root) {
    std::string text;
    if (!path || !read_file(path, text)) {
        set_error("Unable to open file: %s", path ? path : "(null)");
        return false;
    }
    json::ParseError pe;
    json::Parser p(text.data(), text.size());
    if (!p.parse(root, pe)) {
        // same shape as the reference message, src/fir.c:89-90
        set_error("Error in %s (line %d, column %d):\n  %s", path, pe.line,
                  pe.column, pe.text.c_str());
        return false;
    }
    if (!root.is_object() && !root.is_array()) {
        set_error("Error in %s: '[' or '{' expected", path);
        return false;
    }
    return true;
}

// ---------------------------------------------------------------------------
// sample-count tables
// ---------------------------------------------------------------------------

static const double kTolerance = 0.15;      // src/state_machine.c:55
static const uint64_t kReplayLimit = 1ull << 33;

struct Query {
    double bound;       // compare E(k) against this (a float or an integer, widened)
    bool strict;        // false: first k with E >= bound; true: first k with E > bound
    uint64_t *out;
};

// Replays E(0)=0, E(k)=E(k-1)+(1.0/rate)*1e6 once for all queries.
static bool replay(uint32_t rate, std::vector<Query> &qs) {
    std::sort(qs.begin(), qs.end(), [](const Query &a, const Query &b) {
        if (a.bound != b.bound) return a.bound < b.bound;
        return (int)a.strict < (int)b.strict;
    });
    const double delta = ((double)1u / (double)rate) * 1e6;   // state_machine.c:78-82
    double e = 0.0;
    uint64_t k = 0;
    size_t qi = 0;
    while (qi < qs.size()) {
        while (qi < qs.size() &&
               (qs[qi].strict ? e > qs[qi].bound : e >= qs[qi].bound)) {
            *qs[qi].out = k;
            qi++;
        }
        if (qi == qs.size()) break;
        if (k >= kReplayLimit) {
            set_error("duration/timeout too long for sample rate %u "
                      "(more than 2^33 samples)", rate);
            return false;
        }
        e += delta;
        k++;
    }
    return true;
}

bool build_count_tables(ookd_device &d) {
    const size_t ns = d.state_duration_us.size();
    const size_t nt = d.trig_duration_us.size();
    if (d.sample_rate == 0) {
        set_error("sample rate must be non-zero");
        return false;
    }
    d.state_kmin.assign(ns, 0);
    d.state_kmax.assign(ns, UINT64_MAX);
    d.state_kto.assign(ns, UINT64_MAX);
    d.trig_kmin.assign(nt, 0);
    d.trig_kmax.assign(nt, UINT64_MAX);

    std::vector<uint64_t> first_gt_hi_state(ns, 0), first_gt_hi_trig(nt, 0);
    std::vector<Query> qs;
    auto window = [&](uint64_t dur, uint64_t *kmin, uint64_t *gt_hi) {
        // const float min = d - TOL*d, max = d + TOL*d (state_machine.c:109-110)
        const float lo = (float)((double)dur - (kTolerance * (double)dur));
        const float hi = (float)((double)dur + (kTolerance * (double)dur));
        qs.push_back({(double)lo, false, kmin});
        qs.push_back({(double)hi, true, gt_hi});
    };
    for (size_t s = 0; s < ns; s++) {
        if (d.state_duration_us[s] != 0) {
            window(d.state_duration_us[s], &d.state_kmin[s], &first_gt_hi_state[s]);
        }
        if (d.state_timeout_us[s] != 0) {
            qs.push_back({(double)d.state_timeout_us[s], false, &d.state_kto[s]});
        }
    }
    for (size_t t = 0; t < nt; t++) {
        if (d.trig_duration_us[t] != 0) {
            window(d.trig_duration_us[t], &d.trig_kmin[t], &first_gt_hi_trig[t]);
        }
    }
    if (!replay(d.sample_rate, qs)) return false;
    // kmax = (first k with E > hi) - 1; empty window encoded kmin > kmax
    auto finish = [](uint64_t dur, uint64_t &kmin, uint64_t &kmax, uint64_t gt_hi) {
        if (dur == 0) return;
        if (gt_hi == 0 || gt_hi - 1 < kmin) {
            kmin = 1;
            kmax = 0;
        } else {
            kmax = gt_hi - 1;
        }
    };
    for (size_t s = 0; s < ns; s++) {
        finish(d.state_duration_us[s], d.state_kmin[s], d.state_kmax[s], first_gt_hi_state[s]);
    }
    for (size_t t = 0; t < nt; t++) {
        finish(d.trig_duration_us[t], d.trig_kmin[t], d.trig_kmax[t], first_gt_hi_trig[t]);
    }
    return true;
}

// ---------------------------------------------------------------------------
// filter
// ---------------------------------------------------------------------------

static ookd_filter *finish_filter(ookd_filter *f) {
    uint32_t total = 1;
    for (const auto &st : f->stages) total *= st.decimation;     // fir.c:159
    f->total_decimation = total;
    return f;
}

static ookd_filter *filter_from_json(const json::Value &root) {
    const json::Value *jf = root.get("filter");
    if (!jf) {
        set_error("Error: Failed to find \"filter\" entry in filter file.");
        return nullptr;
    }
    const json::Value *stages = jf->get("stages");
    if (!stages) {
        set_error("Error: Failed to find \"stages\" entry in filter file.");
        return nullptr;
    }
    if (!stages->is_array()) {
        set_error("Error: \"filter\" entry in filter file is not an array.");
        return nullptr;
    }
    if (stages->arr.empty()) {
        set_error("Error: Filter must have 1 or more stages.");     // fir.c:118-121
        return nullptr;
    }
    std::unique_ptr<ookd_filter> f(new ookd_filter());
    for (size_t i = 0; i < stages->arr.size(); i++) {
        const json::Value &st = stages->arr[i];
        FilterStage out;
        if (const json::Value *dec = st.get("decimation")) {
            if (!dec->is_integer()) {
                set_error("Error: Decimation must be an integer.");
                return nullptr;
            }
            if (dec->i <= 0 || dec->i >= (long long)UINT_MAX) {      // fir.c:148-152
                set_error("Error: Decimation value is outside of allowed range.");
                return nullptr;
            }
            out.decimation = (uint32_t)dec->i;
        }
        const json::Value *taps = st.get("taps");
        if (!taps) {
            set_error("Error: Filter stage is missing \"taps\" entry.");
            return nullptr;
        }
        if (!taps->is_array()) {
            set_error("Error: Filter \"taps\" must be an array.");
            return nullptr;
        }
        if (taps->arr.empty()) {
            set_error("Error: Filter stage %zu must have 1 or more taps.", i + 1);
            return nullptr;
        }
        for (size_t t = 0; t < taps->arr.size(); t++) {
            if (!taps->arr[t].is_number()) {
                set_error("Error: tap %zu in stage %zu is an invalid value.", t + 1, i + 1);
                return nullptr;
            }
            out.taps.push_back((float)taps->arr[t].number());        // fir.c:224
        }
        f->stages.push_back(std::move(out));
    }
    return finish_filter(f.release());
}

// ---------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------

static int cond_value(const std::string &s) {       // state_machine.c:331-346
    static const char *names[] = {"always", "pulse_start", "pulse_end", "timeout",
                                  "msg_complete"};
    for (int i = 0; i < 5; i++) {
        if (!strcasecmp(s.c_str(), names[i])) return i + 1;
    }
    return 0;
}

static int action_value(const std::string &s) {     // state_machine.c:348-362
    static const char *names[] = {"none", "append_0", "append_1", "output_data"};
    for (int i = 0; i < 4; i++) {
        if (!strcasecmp(s.c_str(), names[i])) return i + 1;
    }
    return 0;
}

// Slot assignment of get_or_reserve_state (state_machine.c:208-247).
struct StateSlots {
    std::vector<std::string> names;
    std::vector<bool> used;
    explicit StateSlots(size_t n) : names(n), used(n, false) {}
    int get(const std::string &name) {
        if (!strcasecmp("reset", name.c_str()) && !used[0]) {
            used[0] = true;
            names[0] = name;
            return 0;
        }
        for (size_t i = 0; i < names.size(); i++) {
            if (!used[i]) {
                used[i] = true;
                names[i] = name;
                return (int)i;
            }
            if (names[i] == name) return (int)i;
        }
        return -1;
    }
};

struct TrigTmp {
    uint8_t cond, action;
    uint32_t next;
    uint64_t duration;
};

static bool load_states(const json::Value &dev, ookd_device &d) {
    const json::Value *states = dev.get("states");
    if (!states || !states->is_array()) {
        set_error("Failed to get states array.");
        return false;
    }
    const size_t n = states->arr.size();
    if (n == 0) {
        set_error("States array is empty.");
        return false;
    }
    StateSlots slots(n);
    std::vector<std::vector<TrigTmp>> trigs(n);
    d.state_duration_us.assign(n, 0);
    d.state_timeout_us.assign(n, 0);

    for (const json::Value &st : states->arr) {
        const json::Value *tmp = st.get("name");
        if (!tmp || !tmp->is_string()) {
            set_error("Failed to get state name.");
            return false;
        }
        const std::string name = tmp->s;
        uint64_t timeout_us = 0, duration_us = 0;

        tmp = st.get("timeout_us");
        if (tmp && tmp->is_integer()) {
            int v = (int)tmp->i;                    // device.c:97 (int tmpval)
            if (v < 0) {
                set_error("Invalid timeout value: %d", v);
                return false;
            }
            timeout_us = (uint64_t)v;
        }
        tmp = st.get("duration_us");
        if (tmp && tmp->is_integer()) {
            int v = (int)tmp->i;                    // device.c:109
            if (v >= 0) duration_us = (uint64_t)v;  // negative: logged, stays 0
        }
        const json::Value *jt = st.get("triggers");
        if (!jt || !jt->is_array()) {
            set_error("Failed to get triggers for state \"%s\"", name.c_str());
            return false;
        }
        if (jt->arr.empty()) {
            set_error("Triggers array is empty for state \"%s\"", name.c_str());
            return false;
        }
        int idx = slots.get(name);                  // sm_add_state
        if (idx < 0) {
            set_error("Failed to add \"%s\" to state machine.", name.c_str());
            return false;
        }
        d.state_duration_us[idx] = duration_us;
        d.state_timeout_us[idx] = timeout_us;
        trigs[idx].clear();                         // re-definition replaces (state_machine.c:261-266)

        for (const json::Value &tr : jt->arr) {
            TrigTmp t{};
            tmp = tr.get("condition");
            if (!tmp || !tmp->is_string()) {
                set_error("Failed to get trigger condition.");
                return false;
            }
            t.cond = (uint8_t)cond_value(tmp->s);
            if (t.cond == 0) {
                set_error("Got invalid trigger condition: %s", tmp->s.c_str());
                return false;
            }
            tmp = tr.get("duration_us");
            if (tmp && tmp->is_integer()) {         // device.c:157-162
                t.duration = tmp->i < 0 ? 0 : (uint64_t)tmp->i;
            }
            tmp = tr.get("state");
            if (!tmp || !tmp->is_string()) {
                set_error("Failed to get trigger's next state.");
                return false;
            }
            const std::string next = tmp->s;
            tmp = tr.get("action");
            if (tmp && tmp->is_string()) {
                t.action = (uint8_t)action_value(tmp->s);
                if (t.action == 0) {
                    set_error("Got invalid trigger action: %s", tmp->s.c_str());
                    return false;
                }
            } else {
                t.action = 1;                       // none
            }
            int ni = slots.get(next);
            if (ni < 0) {
                // The reference ignores this failure and later dereferences a
                // NULL next_state; refuse the file instead.
                set_error("No room left to add state \"%s\"", next.c_str());
                return false;
            }
            t.next = (uint32_t)ni;
            trigs[idx].push_back(t);
        }
    }
    for (size_t i = 0; i < n; i++) {
        if (!slots.used[i]) {                       // sm_initialized (state_machine.c:181-206)
            set_error("State machine is missing states or triggers.");
            return false;
        }
    }
    d.state_names = slots.names;
    d.trig_begin.assign(1, 0);
    for (size_t i = 0; i < n; i++) {
        for (const TrigTmp &t : trigs[i]) {
            d.trig_cond.push_back(t.cond);
            d.trig_action.push_back(t.action);
            d.trig_next.push_back(t.next);
            d.trig_duration_us.push_back(t.duration);
        }
        d.trig_begin.push_back((uint32_t)d.trig_cond.size());
    }
    return true;
}

static int endianness_value(const std::string &s) {     // formatter.c:848-857
    if (!strcasecmp("big", s.c_str())) return 1;
    if (!strcasecmp("little", s.c_str())) return 2;
    return 0;
}

static int format_value(const std::string &s) {         // formatter.c:859-876
    static const char *names[] = {"hex", "unsigned decimal", "sign-magnitude",
                                  "two's complement", "float", "enumeration"};
    for (int i = 0; i < 6; i++) {
        if (!strcasecmp(s.c_str(), names[i])) return i + 1;
    }
    return 0;
}

static int ts_mode_value(const std::string &s) {        // formatter.c:878-893
    static const char *names[] = {"none", "unix", "unix-frac", "datetime-24",
                                  "datetime-ampm"};
    for (int i = 0; i < 5; i++) {
        if (!strcasecmp(s.c_str(), names[i])) return i;
    }
    return -1;
}

static bool load_fields(const json::Value &dev, ookd_device &d) {
    const json::Value *fields = dev.get("fields");
    if (!fields || !fields->is_array()) {
        set_error("Failed to get fields array.");
        return false;
    }
    if (fields->arr.empty()) {
        set_error("Fields array is empty.");
        return false;
    }
    if (const json::Value *ts = dev.get("ts_mode")) {
        if (!ts->is_string()) {
            set_error("'ts_mode' must be a string.");
            return false;
        }
        int m = ts_mode_value(ts->s);
        if (m < 0) {
            set_error("Invalid 'ts_mode' value: %s", ts->s.c_str());
            return false;
        }
        d.ts_mode = m;
    }
    for (const json::Value &jf : fields->arr) {
        FieldDesc f;
        const json::Value *tmp = jf.get("name");
        if (!tmp || !tmp->is_string()) {
            set_error("Failed to get field name.");
            return false;
        }
        f.name = tmp->s;
        tmp = jf.get("default");
        if (!tmp || !tmp->is_string()) {
            set_error("Failed to get default for \"%s\" field.", f.name.c_str());
            return false;
        }
        f.default_value = tmp->s;
        tmp = jf.get("start_bit");
        if (!tmp || !tmp->is_integer()) {
            set_error("Failed to get start bit for \"%s\" field.", f.name.c_str());
            return false;
        }
        f.start_bit = (int)tmp->i;
        tmp = jf.get("end_bit");
        if (!tmp || !tmp->is_integer()) {
            set_error("Failed to get end bit for \"%s\" field.", f.name.c_str());
            return false;
        }
        f.end_bit = (int)tmp->i;
        tmp = jf.get("endianness");
        if (!tmp || !tmp->is_string()) {
            set_error("Failed to get endianness for \"%s\" field.", f.name.c_str());
            return false;
        }
        f.endianness = endianness_value(tmp->s);
        if (!f.endianness) {
            set_error("Invalid endianness specified: %s", tmp->s.c_str());
            return false;
        }
        tmp = jf.get("format");
        if (!tmp || !tmp->is_string()) {
            set_error("Failed to get format for \"%s\" field.", f.name.c_str());
            return false;
        }
        f.format = format_value(tmp->s);
        if (!f.format) {
            set_error("Invalid format: %s", tmp->s.c_str());
            return false;
        }
        if (f.format == 6) {
            const json::Value *enums = jf.get("enum_values");
            if (!enums || !enums->is_array()) {
                set_error("No \"enum_values\" array found for enumeration: %s", f.name.c_str());
                return false;
            }
            for (size_t i = 0; i < enums->arr.size(); i++) {
                const json::Value &e = enums->arr[i];
                const json::Value *es = e.get("string");
                if (!es || !es->is_string()) {
                    set_error("Enumeration value %zu is missing \"string.\"", i);
                    return false;
                }
                const json::Value *ev = e.get("value");
                if (!ev || !ev->is_string()) {
                    set_error("Enumeration item \"%s\" is missing \"value.\"", es->s.c_str());
                    return false;
                }
                // str2uint64 (conversions.c): strtoull base 0, whole string
                errno = 0;
                char *endp = nullptr;
                unsigned long long v = strtoull(ev->s.c_str(), &endp, 0);
                if (errno != 0 || endp == ev->s.c_str() || *endp != '\0') {
                    set_error("Invalid enumeration value: %s", ev->s.c_str());
                    return false;
                }
                f.enums.push_back({es->s, (uint64_t)v});
            }
        }
        tmp = jf.get("offset");
        if (tmp && tmp->is_number()) f.offset = (float)tmp->number();
        tmp = jf.get("scaling");
        if (tmp && tmp->is_number()) f.scaling = (float)tmp->number();
        // formatter_add_field checks (formatter.c:262-270, :300-304)
        if ((unsigned)f.end_bit < (unsigned)f.start_bit) {
            set_error("End bit must be >= start bit");
            return false;
        }
        if (((unsigned)f.end_bit - (unsigned)f.start_bit + 1) > 64) {
            set_error("Fields larger than 64-bits are not currently supported.");
            return false;
        }
        if (f.format == 6 && f.enums.empty()) {
            set_error("Enumeration format requires 1 or more values to be defined");
            return false;
        }
        d.fields.push_back(std::move(f));
    }
    return true;
}

static ookd_device *device_from_json(const json::Value &root, uint32_t sample_rate) {
    const json::Value *dev = root.get("device");
    if (!dev) {
        set_error("Failed to find \"device\" entry in device file");
        return nullptr;
    }
    std::unique_ptr<ookd_device> d(new ookd_device());
    d->sample_rate = sample_rate;
    const json::Value *tmp = dev->get("name");
    if (!tmp || !tmp->is_string()) {
        set_error("Failed to read device name string.");
        return nullptr;
    }
    d->name = tmp->s;
    tmp = dev->get("description");
    if (!tmp || !tmp->is_string()) {
        set_error("Failed to read device description string.");
        return nullptr;
    }
    d->description = tmp->s;
    tmp = dev->get("num_bits");
    if (!tmp || !tmp->is_integer()) {
        set_error("Failed to read \"num_bits\" property.");
        return nullptr;
    }
    if ((int)tmp->i <= 0) {
        set_error("Invalid \"num_bits\" value: %d", (int)tmp->i);
        return nullptr;
    }
    d->num_bits = (uint32_t)(int)tmp->i;
    if (d->num_bits > OOKD_MAX_PAYLOAD_BYTES * 8) {
        set_error("num_bits %u exceeds the %d bits this build carries per message",
                  d->num_bits, OOKD_MAX_PAYLOAD_BYTES * 8);
        return nullptr;
    }
    if (!load_states(*dev, *d)) return nullptr;
    if (!load_fields(*dev, *d)) return nullptr;
    // device_init fails when create_formatter does (device.c:563-566): defaults
    // are parsed against their field's format there
    ookd_formatter *fmt = ookd_formatter_create(d.get());
    if (!fmt) return nullptr;
    ookd_formatter_free(fmt);
    if (!build_count_tables(*d)) return nullptr;
    return d.release();
}

}  // namespace ookd

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
using namespace ookd;

extern "C" {

const char *ookd_last_error(void) { return g_error.c_str(); }
int ookd_api_version(void) { return OOKD_API_VERSION; }

ookd_filter *ookd_filter_load(const char *path) {
    clear_error();
    json::Value root;
    if (!parse_file(path, root)) return nullptr;
    return filter_from_json(root);
}

ookd_filter *ookd_filter_create(uint32_t num_stages, const uint32_t *decimation,
                                const uint32_t *num_taps, const float *taps) {
    clear_error();
    if (num_stages == 0 || !decimation || !num_taps || !taps) {
        set_error("Error: Filter must have 1 or more stages.");
        return nullptr;
    }
    std::unique_ptr<ookd_filter> f(new ookd_filter());
    size_t off = 0;
    for (uint32_t s = 0; s < num_stages; s++) {
        if (decimation[s] == 0 || num_taps[s] == 0) {
            set_error("Error: stage %u needs decimation > 0 and 1 or more taps.", s + 1);
            return nullptr;
        }
        FilterStage st;
        st.decimation = decimation[s];
        st.taps.assign(taps + off, taps + off + num_taps[s]);
        off += num_taps[s];
        f->stages.push_back(std::move(st));
    }
    return finish_filter(f.release());
}

void ookd_filter_free(ookd_filter *f) { delete f; }

uint32_t ookd_filter_total_decimation(const ookd_filter *f) {
    return f ? f->total_decimation : 1;
}

uint32_t ookd_filter_num_stages(const ookd_filter *f) {
    return f ? (uint32_t)f->stages.size() : 0;
}

int ookd_filter_stage(const ookd_filter *f, uint32_t stage, uint32_t *decimation,
                      uint32_t *num_taps, const float **taps) {
    if (!f || stage >= f->stages.size()) return OOKD_ERR_ARG;
    if (decimation) *decimation = f->stages[stage].decimation;
    if (num_taps) *num_taps = (uint32_t)f->stages[stage].taps.size();
    if (taps) *taps = f->stages[stage].taps.data();
    return OOKD_OK;
}

ookd_device *ookd_device_load(const char *path, uint32_t sample_rate) {
    clear_error();
    json::Value root;
    if (!parse_file(path, root)) return nullptr;
    return device_from_json(root, sample_rate);
}

ookd_device *ookd_device_create(const ookd_fsm_tables *t) {
    clear_error();
    if (!t || t->num_states == 0 || t->max_bits == 0 ||
        t->max_bits > OOKD_MAX_PAYLOAD_BYTES * 8 || !t->trig_begin) {
        set_error("ookd_device_create: bad tables");
        return nullptr;
    }
    std::unique_ptr<ookd_device> d(new ookd_device());
    d->name = "custom";
    d->num_bits = t->max_bits;
    d->sample_rate = t->sample_rate;
    const uint32_t ns = t->num_states, nt = t->trig_begin[ns];
    if (nt != t->num_triggers) {
        set_error("ookd_device_create: trig_begin[num_states] != num_triggers");
        return nullptr;
    }
    d->state_duration_us.assign(t->state_duration_us, t->state_duration_us + ns);
    d->state_timeout_us.assign(t->state_timeout_us, t->state_timeout_us + ns);
    d->trig_begin.assign(t->trig_begin, t->trig_begin + ns + 1);
    d->trig_cond.assign(t->trig_cond, t->trig_cond + nt);
    d->trig_action.assign(t->trig_action, t->trig_action + nt);
    d->trig_next.assign(t->trig_next, t->trig_next + nt);
    d->trig_duration_us.assign(t->trig_duration_us, t->trig_duration_us + nt);
    for (uint32_t s = 0; s < ns; s++) {
        d->state_names.push_back(s == 0 ? "reset" : "s" + std::to_string(s));
        if (d->trig_begin[s] > d->trig_begin[s + 1]) {
            set_error("ookd_device_create: trig_begin not monotonic");
            return nullptr;
        }
    }
    for (uint32_t i = 0; i < nt; i++) {
        if (d->trig_next[i] >= ns || d->trig_cond[i] < 1 || d->trig_cond[i] > 5 ||
            d->trig_action[i] < 1 || d->trig_action[i] > 4) {
            set_error("ookd_device_create: trigger %u is invalid", i);
            return nullptr;
        }
    }
    if (!build_count_tables(*d)) return nullptr;
    return d.release();
}

void ookd_device_free(ookd_device *d) { delete d; }

uint32_t ookd_device_num_bits(const ookd_device *d) { return d ? d->num_bits : 0; }

const char *ookd_device_name(const ookd_device *d) { return d ? d->name.c_str() : ""; }

const char *ookd_device_state_name(const ookd_device *d, uint32_t state) {
    if (!d || state >= d->state_names.size()) return "";
    return d->state_names[state].c_str();
}

int ookd_device_tables(const ookd_device *d, ookd_fsm_tables *out) {
    if (!d || !out) return OOKD_ERR_ARG;
    out->num_states = (uint32_t)d->state_duration_us.size();
    out->max_bits = d->num_bits;
    out->sample_rate = d->sample_rate;
    out->num_triggers = (uint32_t)d->trig_cond.size();
    out->state_duration_us = d->state_duration_us.data();
    out->state_timeout_us = d->state_timeout_us.data();
    out->trig_begin = d->trig_begin.data();
    out->trig_cond = d->trig_cond.data();
    out->trig_action = d->trig_action.data();
    out->trig_next = d->trig_next.data();
    out->trig_duration_us = d->trig_duration_us.data();
    out->state_kmin = d->state_kmin.data();
    out->state_kmax = d->state_kmax.data();
    out->state_kto = d->state_kto.data();
    out->trig_kmin = d->trig_kmin.data();
    out->trig_kmax = d->trig_kmax.data();
    return OOKD_OK;
}

}  // extern "C"
