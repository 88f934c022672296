/*
 * ookd_rx.c -- a C99 host on top of libookiedokie_amd.so: the shape of
 *     ookiedokie --rx hip_file -A <capture> -d <device> -F <filter> -s <rate> -f csv|pretty
 * reduced to what the library's boundary covers.  Shows the calls a maintainer
 * of the reference would make from ookiedokie_rx():
 * SDR backend handle -> capture resident in HBM -> fused demodulation ->
 * formatter / rx_print text on stdout, optional --rx-rec-dig file.
 *
 * Build (see tests/test_gpu_parity.py::test_c_host_example):
 *   gcc -std=c99 -Wall -Iinclude examples/ookd_rx.c -o ookd_rx \
 *       -Lookiedokie_amd/lib -lookiedokie_amd -Wl,-rpath,$PWD/ookiedokie_amd/lib
 *
 * ookd_rx <capture.sc16q11> <device.json> <filter.json|none> <samplerate> [csv|pretty] [dig.csv]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ookiedokie_amd.h"

static int fail(const char *what)
{
    fprintf(stderr, "%s: %s\n", what, ookd_last_error());
    return EXIT_FAILURE;
}

int main(int argc, char **argv)
{
    if (argc < 5) {
        fprintf(stderr, "usage: %s <capture.sc16q11> <device.json> <filter.json|none> <samplerate> "
                        "[csv|pretty] [dig.csv]\n", argv[0]);
        return EXIT_FAILURE;
    }
    const int fmt = (argc > 5 && !strcmp(argv[5], "csv")) ? OOKD_RX_FMT_CSV : OOKD_RX_FMT_PRETTY;
    const unsigned rate = (unsigned) strtoul(argv[4], NULL, 0);
    int status = EXIT_FAILURE;

    ookd_host_cfg cfg;                      /* struct ookiedokie_cfg, field for field */
    memset(&cfg, 0, sizeof(cfg));
    cfg.sdr_type = "hip_file";
    cfg.direction = 0;
    cfg.sdr_args = argv[1];
    cfg.samplerate = rate;
    cfg.rx_threshold = 0.1f;                /* ookiedokie_cfg.h defaults */
    cfg.samples_per_buffer = 8192;

    ookd_filter *filter = NULL;
    ookd_device *device = NULL;
    ookd_formatter *formatter = NULL;
    ookd_rx *rx = NULL;
    char *text = NULL;

    void *sdr = sdr_hip_file_init((const struct ookiedokie_cfg *)&cfg);   /* same layout: see ookd_host_cfg */
    if (!sdr) return fail("sdr_hip_file_init");

    const void *d_iq = NULL;
    uint64_t n = 0;
    if (sdr_hip_file_capture(sdr, &d_iq, &n) != 0) { fail("sdr_hip_file_capture"); goto out; }

    if (strcmp(argv[3], "none")) {
        filter = ookd_filter_load(argv[3]);
        if (!filter) { fail("ookd_filter_load"); goto out; }
    }
    const unsigned decimation = filter ? ookd_filter_total_decimation(filter) : 1;
    device = ookd_device_load(argv[2], rate / decimation);      /* main.c:683 */
    if (!device) { fail("ookd_device_load"); goto out; }
    formatter = ookd_formatter_create(device);
    if (!formatter) { fail("ookd_formatter_create"); goto out; }

    ookd_rx_config rc;
    memset(&rc, 0, sizeof(rc));
    rc.threshold = cfg.rx_threshold;
    rc.samples_per_buffer = cfg.samples_per_buffer;
    rc.max_samples = n ? n : 1;
    rx = ookd_rx_create(&rc, filter, device);
    if (!rx) { fail("ookd_rx_create"); goto out; }

    if (ookd_rx_process_device(rx, d_iq, 1, n, n) != 0) { fail("ookd_rx_process_device"); goto out; }

    int first_print = 1;
    const uint64_t nmsg = ookd_rx_num_messages(rx);
    const ookd_message *msgs = ookd_rx_messages(rx);
    const size_t len = ookd_print_messages(formatter, fmt, &first_print, msgs, nmsg,
                                           cfg.samples_per_buffer, decimation, NULL, 0);
    text = malloc(len + 1);
    if (!text) goto out;
    first_print = 1;
    ookd_print_messages(formatter, fmt, &first_print, msgs, nmsg, cfg.samples_per_buffer, decimation,
                        text, len + 1);
    fputs(text, stdout);

    if (argc > 6 && ookd_rx_record_dig(rx, 0, argv[6]) != 0) { fail("ookd_rx_record_dig"); goto out; }

    ookd_rx_stats st;
    if (ookd_rx_get_stats(rx, &st) == 0) {
        fprintf(stderr, "%llu samples, %llu edges, %llu messages, front end %.3f ms\n",
                (unsigned long long) st.input_samples, (unsigned long long) st.num_edges,
                (unsigned long long) st.num_messages, st.fir_kernel_ms);
    }
    status = EXIT_SUCCESS;

out:
    free(text);
    ookd_rx_destroy(rx);
    ookd_formatter_free(formatter);
    ookd_device_free(device);
    ookd_filter_free(filter);
    sdr_hip_file_deinit(sdr);
    return status;
}
