// synth.cpp -- synthetic SC16Q11 capture generator.
//
// The envelope of each message is the walk the reference's tx half makes
// through the device state machine (sm_generate / generate /
// handle_tx_triggers / get_tx_trigger / append_samples,
// src/state_machine.c:572-873), restated here on the flat tables; durations
// become sample counts with the reference's rounding
// (unsigned)(us * (rate / 1e6) + 0.5), state_machine.c:88-92.  The "on"
// level is 1945 = (int16_t)(0.95f * 2048.0f) (device.c:675, complexf.h:93).
// Messages are separated by pseudo-random gaps, may carry a random carrier
// phase and integer noise, and 1 in N is preceded by a 100 us glitch pulse
// 1 ms before its start pulse (exercises ERROR + drop-rest-of-buffer).
// Everything is integer and counter based, so the device kernel and the
// host fill produce identical samples.
#include <cmath>
#include <cstring>
#include <memory>

#include "common.hpp"
#include "kernels.hpp"

using namespace ookd;

struct ookd_synth {
    ookd_synth_config cfg{};
    uint64_t num_samples = 0;
    std::vector<SynthRun> runs;
    std::vector<uint64_t> msg_start;
    std::vector<std::vector<uint8_t>> payloads;
    // device copy (per HIP device ordinal, created lazily)
    int dev = -1;
    SynthRun *d_runs = nullptr;

    ~ookd_synth() {
        if (d_runs) {
            (void)hipSetDevice(dev);
            (void)hipFree(d_runs);
        }
    }
};

namespace {

enum { kAlways = 1, kPulseStart, kPulseEnd, kTimeout, kMsgComplete };
enum { kNone = 1, kAppend0, kAppend1, kOutput };

struct TxWalk {
    const ookd_device &d;
    uint32_t rate;
    uint32_t cur = 0;
    uint32_t nbits = 0;
    bool level = false;
    std::vector<std::pair<bool, uint32_t>> runs;    // (level, samples)
    std::string err;

    TxWalk(const ookd_device &dev, uint32_t r) : d(dev), rate(r) {}

    // state_machine.c:88-92
    uint32_t to_samples(uint64_t us) const {
        return (unsigned int)((double)us * ((double)rate / 1e6) + 0.5);
    }

    void append(uint64_t us) {                      // :572-620
        const uint32_t n = to_samples(us);
        if (n == 0) return;
        if (!runs.empty() && runs.back().first == level) runs.back().second += n;
        else runs.emplace_back(level, n);
    }

    // get_tx_trigger, :622-699.  Returns false on the reference's error path.
    bool pick(bool bit, bool check_action, int &active) {
        active = -1;
        for (uint32_t t = d.trig_begin[cur]; t < d.trig_begin[cur + 1] && active < 0; ++t) {
            if (check_action) {
                bool have;
                switch (d.trig_action[t]) {
                case kAppend0: have = !bit; break;
                case kAppend1: have = bit; break;
                case kOutput: have = true; break;
                default: have = false;
                }
                if (!have) continue;
            }
            switch (d.trig_cond[t]) {
            case kMsgComplete:
                if (nbits == d.num_bits) active = (int)t;
                break;
            case kAlways:
            case kPulseStart:
            case kPulseEnd:
                active = (int)t;
                break;
            case kTimeout:
                err = "encountered a timeout trigger while generating samples";
                return false;
            default:
                err = "unhandled trigger condition";
                return false;
            }
        }
        return true;
    }

    // handle_tx_triggers, :701-806
    bool handle(bool bit, bool &done) {
        done = false;
        int a;
        if (!pick(bit, true, a)) return false;
        if (a < 0 && !pick(bit, false, a)) return false;
        if (a < 0) {
            err = "no usable trigger in state " + d.state_names[cur];
            return false;
        }
        if (d.state_duration_us[cur] == 0 && d.trig_duration_us[a] != 0) append(d.trig_duration_us[a]);
        if (d.trig_cond[a] == kPulseStart) {
            if (level) {
                err = "pulse_start while already on";
                return false;
            }
            level = true;
        } else if (d.trig_cond[a] == kPulseEnd) {
            if (!level) {
                err = "pulse_end while already off";
                return false;
            }
            level = false;
        }
        switch (d.trig_action[a]) {
        case kAppend0:
        case kAppend1:
            if (nbits < d.num_bits) {
                nbits++;
                done = true;
            } else if (nbits > d.num_bits) {
                err = "bit count exceeded max";
                return false;
            }
            break;
        case kOutput:
            done = true;
            break;
        default:
            break;
        }
        cur = d.trig_next[a];
        if (d.state_duration_us[cur] != 0) append(d.state_duration_us[cur]);
        return true;
    }

    bool generate(bool bit) {                       // :809-822
        bool done = false;
        // a well-formed device reaches `done`; bound the walk so a malformed
        // one fails instead of spinning
        for (int guard = 0; guard < 100000 && !done; ++guard) {
            if (!handle(bit, done)) return false;
        }
        if (!done) {
            err = "tx walk does not terminate";
            return false;
        }
        return true;
    }

    bool message(const uint8_t *payload) {          // sm_generate, :825-873
        cur = 0;
        nbits = 0;
        level = false;
        runs.clear();
        for (uint32_t i = 0; i < d.num_bits; ++i) {
            const bool bit = (payload[i / 8] >> (i % 8)) & 1u;
            if (!generate(bit)) return false;
        }
        return generate(false);
    }
};

}  // namespace

extern "C" {

ookd_synth *ookd_synth_create(const ookd_device *device, const ookd_synth_config *cfg,
                              uint64_t num_samples) {
    clear_error();
    if (!device || !cfg || cfg->sample_rate == 0) {
        set_error("ookd_synth_create: bad argument");
        return nullptr;
    }
    std::unique_ptr<ookd_synth> s(new ookd_synth());
    s->cfg = *cfg;
    if (s->cfg.amplitude == 0) s->cfg.amplitude = 1945;
    if (s->cfg.gap_min_us == 0 && s->cfg.gap_max_us == 0) {
        s->cfg.gap_min_us = 4000;
        s->cfg.gap_max_us = 20000;
    }
    if (s->cfg.gap_max_us < s->cfg.gap_min_us) s->cfg.gap_max_us = s->cfg.gap_min_us;
    s->num_samples = num_samples;

    TxWalk walk(*device, cfg->sample_rate);
    const uint32_t nbytes = (device->num_bits + 7) / 8;
    const double rate_us = (double)cfg->sample_rate / 1e6;
    const uint64_t glitch_len = (uint64_t)(100.0 * rate_us + 0.5);
    const uint64_t glitch_lead = (uint64_t)(1000.0 * rate_us + 0.5);
    uint64_t pos = 0;
    s->runs.push_back({0, 0, 0, 0});
    for (uint64_t i = 0; pos < num_samples; ++i) {
        const uint64_t h = synth_mix(s->cfg.seed + 0x1000 + i);
        const uint64_t h2 = synth_mix(h);
        const uint64_t h3 = synth_mix(h2);
        std::vector<uint8_t> payload(OOKD_MAX_PAYLOAD_BYTES, 0);
        for (uint32_t b = 0; b < device->num_bits; ++b) {
            const uint64_t word = synth_mix(h3 + (b >> 6));
            if ((word >> (b & 63)) & 1ull) payload[b / 8] |= (uint8_t)(1u << (b % 8));
        }
        const uint64_t span = (uint64_t)s->cfg.gap_max_us - s->cfg.gap_min_us + 1;
        const uint64_t gap_us = s->cfg.gap_min_us + (h % span);
        const uint64_t gap = (uint64_t)((double)gap_us * rate_us + 0.5);
        int16_t li = (int16_t)s->cfg.amplitude, lq = 0;
        if (s->cfg.random_phase) {
            const double phi = 2.0 * M_PI * (double)(h2 >> 11) / 9007199254740992.0;
            li = (int16_t)std::lround((double)s->cfg.amplitude * std::cos(phi));
            lq = (int16_t)std::lround((double)s->cfg.amplitude * std::sin(phi));
        }
        const bool glitch = s->cfg.glitch_every && (i % s->cfg.glitch_every) == s->cfg.glitch_every - 1 &&
                            gap > glitch_lead + glitch_len && glitch_len > 0;
        // gap (off), optional glitch inside it
        if (glitch) {
            const uint64_t g0 = pos + gap - glitch_lead;
            s->runs.push_back({g0, li, lq, 0});
            s->runs.push_back({g0 + glitch_len, 0, 0, 0});
        }
        pos += gap;
        if (pos >= num_samples) break;
        if (!walk.message(payload.data())) {
            set_error("device cannot be transmitted: %s", walk.err.c_str());
            return nullptr;
        }
        s->msg_start.push_back(pos);
        payload.resize(nbytes);
        s->payloads.push_back(payload);
        for (const auto &r : walk.runs) {
            s->runs.push_back({pos, r.first ? li : (int16_t)0, r.first ? lq : (int16_t)0, 0});
            pos += r.second;
        }
        // the waveform ends "on"; the following gap starts with the off level
        s->runs.push_back({pos, 0, 0, 0});
    }
    return s.release();
}

void ookd_synth_free(ookd_synth *s) { delete s; }

uint64_t ookd_synth_num_messages(const ookd_synth *s) { return s ? s->msg_start.size() : 0; }

int ookd_synth_message(const ookd_synth *s, uint64_t i, uint64_t *start_sample, uint8_t *payload) {
    if (!s || i >= s->msg_start.size()) return OOKD_ERR_ARG;
    if (start_sample) *start_sample = s->msg_start[i];
    if (payload) {
        memset(payload, 0, OOKD_MAX_PAYLOAD_BYTES);
        memcpy(payload, s->payloads[i].data(), s->payloads[i].size());
    }
    return OOKD_OK;
}

int ookd_synth_fill_host(const ookd_synth *s, uint64_t first, uint64_t count, int16_t *iq) {
    if (!s || (!iq && count)) return OOKD_ERR_ARG;
    // last run with start <= first
    size_t lo = 0, hi = s->runs.size();
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) / 2;
        if (s->runs[mid].start <= first) lo = mid;
        else hi = mid;
    }
    size_t ri = lo;
    for (uint64_t k = 0; k < count; ++k) {
        const uint64_t n = first + k;
        while (ri + 1 < s->runs.size() && s->runs[ri + 1].start <= n) ri++;
        int ni, nq;
        synth_noise(s->cfg.seed, n, s->cfg.noise, ni, nq);
        int vi = s->runs[ri].i_level + ni;
        int vq = s->runs[ri].q_level + nq;
        vi = vi < -32768 ? -32768 : (vi > 32767 ? 32767 : vi);
        vq = vq < -32768 ? -32768 : (vq > 32767 ? 32767 : vq);
        iq[2 * k] = (int16_t)vi;
        iq[2 * k + 1] = (int16_t)vq;
    }
    return OOKD_OK;
}

int ookd_synth_fill_device(const ookd_synth *cs, int32_t hip_device, uint64_t first, uint64_t count,
                           void *d_iq, void *stream) {
    clear_error();
    ookd_synth *s = const_cast<ookd_synth *>(cs);
    if (!s || (!d_iq && count)) return OOKD_ERR_ARG;
    if (hipSetDevice(hip_device) != hipSuccess) {
        set_error("hipSetDevice(%d) failed: no HIP device, and there is no CPU fallback", hip_device);
        return OOKD_ERR_HIP;
    }
    if (s->d_runs && s->dev != hip_device) {
        (void)hipSetDevice(s->dev);
        (void)hipFree(s->d_runs);
        s->d_runs = nullptr;
        (void)hipSetDevice(hip_device);
    }
    if (!s->d_runs) {
        if (hipMalloc(reinterpret_cast<void **>(&s->d_runs), s->runs.size() * sizeof(SynthRun)) != hipSuccess ||
            hipMemcpy(s->d_runs, s->runs.data(), s->runs.size() * sizeof(SynthRun), hipMemcpyHostToDevice) !=
                hipSuccess) {
            set_error("run table upload failed");
            return OOKD_ERR_HIP;
        }
        s->dev = hip_device;
    }
    hipError_t e = launch_synth(s->d_runs, s->runs.size(), s->cfg.seed, s->cfg.noise, first, count,
                                static_cast<int16_t *>(d_iq), static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        set_error("synth kernel launch failed: %s", hipGetErrorString(e));
        return OOKD_ERR_HIP;
    }
    return OOKD_OK;
}

}  // extern "C"
