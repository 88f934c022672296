// stream_fir.cpp -- GPU-backed equivalent of the reference's fine-grained FIR
// API: fir_filter_and_decimate / fir_reset (src/fir.h:68-81,
// src/fir.c:272-295, :355-395).  Host complexf in, host complexf out, state
// carried across calls so results do not depend on chunking.  The carried
// state is the raw input history (the filter is feed-forward, so every
// stage's history is a function of it) plus the count of samples consumed,
// which fixes each stage's decimation phase (fir.c:47, :290).
#include <cstring>
#include <memory>

#include "common.hpp"
#include "kernels.hpp"

using namespace ookd;

struct ookd_fir {
    int dev = 0;
    hipStream_t stream = nullptr;
    uint32_t num_stages = 0;
    FirStageDev stage[kMaxStages]{};
    uint32_t total_decim = 1;
    uint64_t halo_needed = 0;
    size_t max_input = 0;
    float *d_taps = nullptr;
    float *d_in = nullptr, *d_out = nullptr, *d_halo = nullptr;
    std::vector<float> history;     // last halo_needed input samples (float2), oldest first
    uint64_t consumed = 0;          // samples seen since reset

    ~ookd_fir() {
        (void)hipSetDevice(dev);
        if (d_taps) (void)hipFree(d_taps);
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        if (d_halo) (void)hipFree(d_halo);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

extern "C" {

ookd_fir *ookd_fir_create(int32_t hip_device, const ookd_filter *filter, size_t max_input, uint32_t) {
    clear_error();
    if (!filter || max_input == 0 || filter->stages.size() > (size_t)kMaxStages) {
        set_error("ookd_fir_create: bad argument");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || hip_device < 0 || hip_device >= ndev) {
        set_error("no HIP device %d available: libookiedokie_amd has no CPU fallback", hip_device);
        return nullptr;
    }
    std::unique_ptr<ookd_fir> f(new ookd_fir());
    f->dev = hip_device;
    f->max_input = max_input;
    (void)hipSetDevice(hip_device);
    std::vector<float> taps;
    uint64_t mult = 1;
    f->num_stages = (uint32_t)filter->stages.size();
    f->total_decim = filter->total_decimation;
    for (uint32_t s = 0; s < f->num_stages; ++s) {
        const auto &st = filter->stages[s];
        FirStageDev d{};
        d.decim = st.decimation;
        d.ntaps = (uint32_t)st.taps.size();
        d.ntaps_pad = d.ntaps;
        d.tap_off = (uint32_t)taps.size();
        taps.insert(taps.end(), st.taps.begin(), st.taps.end());
        f->stage[s] = d;
        // one extra decimation period of slack covers any phase
        f->halo_needed += (uint64_t)(d.ntaps - 1 + d.decim) * mult;
        mult *= d.decim;
    }
    const size_t out_max = max_input / 1 + 2;
    if (hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&f->d_taps), taps.size() * sizeof(float)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&f->d_in), max_input * 2 * sizeof(float)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&f->d_out), out_max * 2 * sizeof(float)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&f->d_halo), (f->halo_needed + 1) * 2 * sizeof(float)) != hipSuccess ||
        hipMemcpy(f->d_taps, taps.data(), taps.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        set_error("ookd_fir_create: device allocation failed");
        return nullptr;
    }
    FrontParams probe{};
    probe.num_stages = f->num_stages;
    for (uint32_t s = 0; s < f->num_stages; ++s) probe.stage[s] = f->stage[s];
    if (generic_lds_bytes(probe) > 160 * 1024) {
        set_error("filter needs more than 160 KiB of LDS per tile");
        return nullptr;
    }
    ookd_fir_reset(f.get());
    return f.release();
}

void ookd_fir_reset(ookd_fir *f) {
    if (!f) return;
    f->history.assign(2 * f->halo_needed, 0.0f);    // fir.c:279-281: zero state
    f->consumed = 0;                                // fir.c:290: count = decimation
}

void ookd_fir_destroy(ookd_fir *f) { delete f; }

size_t ookd_fir_filter_and_decimate(ookd_fir *f, const ookd_complexf *input, size_t count,
                                    ookd_complexf *output) {
    clear_error();
    if (!f || (!input && count) || count > f->max_input) {
        set_error("ookd_fir_filter_and_decimate: bad argument (count %zu, max_input %zu)", count,
                  f ? f->max_input : 0);
        return 0;
    }
    if (count == 0) return 0;
    (void)hipSetDevice(f->dev);
    const uint64_t g0 = f->consumed;
    const uint64_t n_out = (g0 + count) / f->total_decim - g0 / f->total_decim;
    const size_t H = (size_t)f->halo_needed;
    bool ok = true;
    ok = ok && hipMemcpyAsync(f->d_in, input, count * 8, hipMemcpyHostToDevice, f->stream) == hipSuccess;
    if (H) ok = ok && hipMemcpyAsync(f->d_halo, f->history.data(), H * 8, hipMemcpyHostToDevice, f->stream) == hipSuccess;
    if (ok && n_out) {
        FrontParams p{};
        p.iq_f32 = f->d_in;
        p.cap_stride = count;
        p.n_valid = count;
        p.n_in = count;
        p.n_out = n_out;
        p.origin = g0;
        p.halo_f32 = f->d_halo;
        p.halo_len = (uint32_t)H;
        p.num_stages = f->num_stages;
        for (uint32_t s = 0; s < f->num_stages; ++s) p.stage[s] = f->stage[s];
        p.taps = f->d_taps;
        p.fir_out = f->d_out;
        p.p_star = 0.0f;
        ok = launch_front_generic(p, 1, f->stream) == hipSuccess;
        ok = ok && hipMemcpyAsync(output, f->d_out, n_out * 8, hipMemcpyDeviceToHost, f->stream) == hipSuccess;
    }
    ok = ok && hipStreamSynchronize(f->stream) == hipSuccess;
    if (!ok) {
        set_error("ookd_fir_filter_and_decimate: HIP failure: %s", hipGetErrorString(hipGetLastError()));
        return 0;
    }
    // slide the history: keep the newest H samples of (history ++ input)
    if (H) {
        const float *in = reinterpret_cast<const float *>(input);
        if (count >= H) {
            memcpy(f->history.data(), in + 2 * (count - H), H * 8);
        } else {
            memmove(f->history.data(), f->history.data() + 2 * count, (H - count) * 8);
            memcpy(f->history.data() + 2 * (H - count), in, count * 8);
        }
    }
    f->consumed += count;
    return (size_t)n_out;
}

}  // extern "C"
