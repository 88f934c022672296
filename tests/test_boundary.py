"""The drop-in boundary against the reference's REAL headers (runs only where /root/reference exists:
the reference cannot travel to the GPU box).  A C99 translation unit includes the reference's
src/sdr/sdr.h and src/sdr/supported_devices.h (with the file backend enabled, as CMake does), our
include/ookiedokie_amd.h, says SDR_PROTOTYPES(hip_file) exactly as src/sdr/sdr.c would after
INTEGRATION.md section 1, builds the vtable entry with the reference's own SDR_INTERFACE macro into a
struct of the member types of src/sdr/sdr.c:50-59, asserts that ookd_host_cfg mirrors
struct ookiedokie_cfg (src/ookiedokie_cfg.h:50-91) member by member, and -- linked against
libookiedokie_amd.so and the reference's own ookiedokie_cfg.c / keyval_list.c -- drives the backend
through that vtable with a configuration initialised by the reference's ookiedokie_cfg_init
(src/ookiedokie_cfg.c:40): the tx direction needs no GPU, so the file the backend writes is checked
against complexf_to_sc16q11 (src/complexf.h:87-96)."""
import os
import struct
import subprocess

import pytest

import ookiedokie_amd as ok
from ookiedokie_amd import build as okbuild

REF = "/root/reference/src"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")

TU = r'''
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sdr/sdr.h"                    /* reference: ookiedokie_cfg.h, complexf.h, SDR_FILE_EOF */
#include "sdr/supported_devices.h"      /* reference: SDR_PROTOTYPES, SDR_INTERFACE */
#include "ookiedokie_amd.h"             /* ours, in the same translation unit */

SDR_PROTOTYPES(hip_file);               /* what sdr.c gains (INTEGRATION.md section 1) */

/* member types of the reference's backend table, src/sdr/sdr.c:50-59 */
struct sdr_interface {
    const char *name;
    const char *file_handler;
    const char *default_filter;
    void * (*init)(const struct ookiedokie_cfg *config);
    void (*deinit)(void *dev);
    int (*rx)(void *dev, struct complexf *samples, unsigned int count);
    int (*tx)(void *dev, const struct complexf *samples, unsigned int count);
    int (*flush)(void *dev);
};
static const struct sdr_interface hip = SDR_INTERFACE(hip_file, hip_file, "fs128_fs16_dec4");

#define SAME(m) _Static_assert(offsetof(struct ookiedokie_cfg, m) == offsetof(ookd_host_cfg, m), #m " moved")
SAME(sdr_type); SAME(direction); SAME(sdr_args); SAME(frequency); SAME(bandwidth); SAME(samplerate); SAME(gain);
SAME(device); SAME(tx_count); SAME(tx_delay_us); SAME(device_params); SAME(rx_fmt); SAME(rx_threshold);
SAME(rx_rec_filename); SAME(rx_rec_type); SAME(rx_filter); SAME(rx_rec_dig); SAME(rx_rec_input);
SAME(samples_per_buffer); SAME(num_buffers); SAME(num_transfers); SAME(stream_timeout_ms); SAME(sync_timeout_ms);
SAME(verbosity);
_Static_assert(sizeof(struct ookiedokie_cfg) == sizeof(ookd_host_cfg), "struct ookiedokie_cfg changed size");
_Static_assert(sizeof(struct complexf) == sizeof(ookd_complexf) && offsetof(struct complexf, imag) == offsetof(ookd_complexf, imag),
               "struct complexf is not ookd_complexf");
_Static_assert(SDR_FILE_EOF == OOKD_FILE_EOF, "EOF code");

int main(int argc, char **argv) {
    struct ookiedokie_cfg cfg;
    if (argc < 2 || ookiedokie_cfg_init(&cfg) != 0) return 2;
    if (strcmp(hip.name, "hip_file") || strcmp(hip.file_handler, hip.name)) return 3;   /* a file handler: sdr.c:131-134 */
    cfg.sdr_type = strdup(hip.name);            /* ookiedokie_cfg_deinit frees its strings (ookiedokie_cfg.c:89-99) */
    cfg.direction = DIRECTION_TX;
    /* no file name: the reference's file backend refuses (bladeRF_file.c:63-67), so does ours */
    if (hip.init(&cfg) != NULL) return 4;
    cfg.sdr_args = strdup(argv[1]);
    void *dev = hip.init(&cfg);
    if (!dev) { fprintf(stderr, "%s\n", ookd_last_error()); return 5; }
    struct complexf s[5] = {{0.95f, 0.0f}, {-0.5f, 0.25f}, {0.0f, -1.0f}, {0.999f, 0.001f}, {0.0f, 0.0f}};
    int16_t want[10];
    complexf_to_sc16q11(s, want, 5);            /* the reference's own conversion */
    fwrite(want, 2, 10, stdout);
    if (hip.rx(dev, s, 5) == 0) return 6;       /* a tx handle does not receive */
    if (hip.tx(dev, s, 5) != 0 || hip.flush(dev) != 0) return 7;
    hip.deinit(dev);
    hip.deinit(NULL);                           /* NULL-safe like bladeRF_file.c:50 */
    ookiedokie_cfg_deinit(&cfg);
    return 0;
}
'''


def test_backend_binds_through_the_reference_headers(tmp_path):
    lib = okbuild.build()
    src = tmp_path / "bind.c"
    src.write_text(TU)
    exe = tmp_path / "bind"
    # the reference's own sources (GNU C, its own warnings are its business) ...
    objs = []
    for name in ("ookiedokie_cfg.c", "keyval_list.c", "log.c"):
        obj = tmp_path / (name + ".o")
        r = subprocess.run(["gcc", "-std=gnu99", "-w", '-DSHORT_FILE_="%s"' % name, "-I", REF, "-c", os.path.join(REF, name),
                            "-o", str(obj)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        objs.append(str(obj))
    # ... and the binding translation unit, warnings as errors: a conflicting declaration would stop here
    cmd = ["gcc", "-std=gnu11", "-Wall", "-Werror", "-Wno-unused-function", "-Wno-unused-variable",
           "-DENABLE_BLADERF_SC16Q11_FILE=1", '-DSHORT_FILE_="bind.c"',
           "-I", REF, "-I", os.path.dirname(ok.HEADER_PATH), str(src)] + objs + [
           "-L", os.path.dirname(lib), "-lookiedokie_amd", "-Wl,-rpath," + os.path.dirname(lib), "-lm", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = tmp_path / "tx.sc16q11"
    r = subprocess.run([str(exe), str(out)], capture_output=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    assert out.read_bytes() == r.stdout and len(r.stdout) == 20
    assert struct.unpack("<10h", r.stdout)[:2] == (1945, 0)         # 0.95 * 2048 truncated: device.c:675
