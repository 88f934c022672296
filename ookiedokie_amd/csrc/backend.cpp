// backend.cpp -- the SDR backend entry points the reference's table binds
// (SDR_PROTOTYPES / SDR_INTERFACE, src/sdr/supported_devices.h:32-48;
// vtable src/sdr/sdr.c:50-122), for a file handler named "hip_file".
//
// Behaviour follows src/sdr/bladeRF_file.c: rx reads up to buf_len samples
// per fread (:107-110), a read of 0 items is SDR_FILE_EOF (:111-112), a
// short read is zero padded and reported as success (:113-117), then the
// block is unpacked (:119) -- here by the unpack kernel on the GPU.  tx is
// the reference's truncating pack (complexf.h:87-96) + fwrite; it is not on
// the hot path and stays on the host.
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <memory>

#include "common.hpp"
#include "ingest.hpp"
#include "kernels.hpp"

using namespace ookd;

namespace {

struct HipFile {
    FILE *file = nullptr;
    bool rx = true;
    unsigned buf_len = 0;
    int dev = 0;
    hipStream_t stream = nullptr;
    int16_t *h_raw = nullptr;       // pinned, buf_len samples
    float *h_out = nullptr;         // pinned
    int16_t *d_raw = nullptr;
    float *d_out = nullptr;
    // whole capture in HBM for the fused path
    int16_t *d_capture = nullptr;
    uint64_t capture_samples = 0;
    std::string path;
    Ingest ingest;

    ~HipFile() {
        if (file) fclose(file);
        if (h_raw) (void)hipHostFree(h_raw);
        if (h_out) (void)hipHostFree(h_out);
        if (d_raw) (void)hipFree(d_raw);
        if (d_out) (void)hipFree(d_out);
        if (d_capture) (void)hipFree(d_capture);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

}  // namespace

extern "C" {

void sdr_hip_file_deinit(void *handle) { delete static_cast<HipFile *>(handle); }

void *sdr_hip_file_init(const struct ookiedokie_cfg *cfg_v) {
    clear_error();
    // same layout as the reference's struct (tests/test_boundary.py asserts the offsets against its header)
    const ookd_host_cfg *cfg = reinterpret_cast<const ookd_host_cfg *>(cfg_v);
    if (!cfg || !cfg->sdr_args || cfg->samples_per_buffer == 0) {
        set_error("A filename must be provided as \"SDR args\" when using hip_file.");
        return nullptr;
    }
    std::unique_ptr<HipFile> h(new HipFile());
    h->rx = (cfg->direction == 0);
    h->buf_len = cfg->samples_per_buffer;
    h->path = cfg->sdr_args;
    if (h->rx) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
            set_error("no HIP device available: the hip_file backend has no CPU fallback");
            return nullptr;
        }
        const size_t n = h->buf_len;
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void **>(&h->h_raw), n * 4) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void **>(&h->h_out), n * 8) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&h->d_raw), n * 4) != hipSuccess ||
            hipMalloc(reinterpret_cast<void **>(&h->d_out), n * 8) != hipSuccess) {
            set_error("hip_file: buffer allocation failed");
            return nullptr;
        }
    }
    h->file = fopen(cfg->sdr_args, h->rx ? "rb" : "wb");
    if (!h->file) {
        set_error("Failed to open %s: %s", cfg->sdr_args, strerror(errno));
        return nullptr;
    }
    return h.release();
}

int sdr_hip_file_rx(void *handle, struct complexf *samples_v, unsigned int count) {
    ookd_complexf *samples = reinterpret_cast<ookd_complexf *>(samples_v);
    HipFile *h = static_cast<HipFile *>(handle);
    if (!h || !h->rx || !samples) return -1;
    int status = 0;
    unsigned total = 0;
    while (status == 0 && total < count) {
        const unsigned to_read = std::min(h->buf_len, count - total);
        const size_t n = fread(h->h_raw, 4, to_read, h->file);
        if (n == 0) {
            status = OOKD_FILE_EOF;
        } else if (n < to_read) {
            memset(h->h_raw + 2 * n, 0, 4 * (size_t)(to_read - n));
        }
        bool ok = hipMemcpyAsync(h->d_raw, h->h_raw, (size_t)to_read * 4, hipMemcpyHostToDevice, h->stream) == hipSuccess;
        ok = ok && launch_unpack(h->d_raw, h->d_out, to_read, h->stream) == hipSuccess;
        ok = ok && hipMemcpyAsync(samples, h->d_out, (size_t)to_read * 8, hipMemcpyDeviceToHost, h->stream) == hipSuccess;
        ok = ok && hipStreamSynchronize(h->stream) == hipSuccess;
        if (!ok) {
            set_error("hip_file rx: HIP failure: %s", hipGetErrorString(hipGetLastError()));
            return OOKD_ERR_HIP;
        }
        samples += to_read;
        total += to_read;
    }
    return status;
}

int sdr_hip_file_tx(void *handle, const struct complexf *samples_v, unsigned int count) {
    const ookd_complexf *samples = reinterpret_cast<const ookd_complexf *>(samples_v);
    HipFile *h = static_cast<HipFile *>(handle);
    if (!h || h->rx || !samples) return -1;
    std::vector<int16_t> buf(2 * (size_t)h->buf_len);
    unsigned total = 0;
    while (total < count) {
        const unsigned to_write = std::min(h->buf_len, count - total);
        for (unsigned i = 0; i < to_write; ++i) {       // complexf.h:87-96
            buf[2 * i] = (int16_t)(samples[i].real * 2048.0f);
            buf[2 * i + 1] = (int16_t)(samples[i].imag * 2048.0f);
        }
        if (fwrite(buf.data(), 4, to_write, h->file) != to_write) return -1;
        samples += to_write;
        total += to_write;
    }
    return 0;
}

int sdr_hip_file_flush(void *) { return 0; }    // bladeRF_file.c:157-161

int sdr_hip_file_capture(void *handle, const void **d_iq, uint64_t *num_samples) {
    clear_error();
    HipFile *h = static_cast<HipFile *>(handle);
    if (!h || !h->rx || !d_iq || !num_samples) return OOKD_ERR_ARG;
    if (!h->d_capture) {
        FILE *f = fopen(h->path.c_str(), "rb");
        if (!f) {
            set_error("Failed to open %s: %s", h->path.c_str(), strerror(errno));
            return OOKD_ERR_IO;
        }
        fseek(f, 0, SEEK_END);
        const long long bytes = ftell(f);
        fseek(f, 0, SEEK_SET);
        const uint64_t n = bytes > 0 ? (uint64_t)bytes / 4 : 0;     // whole samples only
        if (hipMalloc(reinterpret_cast<void **>(&h->d_capture), (n + 4) * 4) != hipSuccess) {
            fclose(f);
            set_error("hip_file: capture allocation of %llu bytes failed", (unsigned long long)n * 4);
            return OOKD_ERR_NOMEM;
        }
        // pinned double-buffered ingest: fread of block k+1 overlaps the DMA of block k
        const long long got = h->ingest.from_file(f, h->d_capture, n * 4);
        if (got < 0) {
            fclose(f);
            return OOKD_ERR_HIP;
        }
        const uint64_t done = (uint64_t)got / 4;
        fclose(f);
        h->capture_samples = done;
    }
    *d_iq = h->d_capture;
    *num_samples = h->capture_samples;
    return OOKD_OK;
}

}  // extern "C"
