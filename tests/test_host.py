"""CPU-side tests of the product's host layer (no GPU needed):
the C-ABI library loads and exports every symbol the header declares, the
C++ JSON/filter/device loaders agree with an independent Python loader, the
integer sample-count tables equal the oracle's replay of the reference
accumulation, and compute entry points fail loudly without a GPU."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from tests.helpers import GOLDEN, golden_path, stream_from_runs

import ookiedokie_amd as ok
from ookiedokie_amd import build as okbuild


@pytest.fixture(scope="session", autouse=True)
def built_lib():
    okbuild.build()
    return ok.lib()


def test_library_exports_every_declared_symbol(built_lib):
    text = open(ok.HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b((?:ookd|sdr_hip_file)_[a-z0-9_]+)\s*\(", text))
    assert len(names) >= 45
    missing = [n for n in sorted(names) if not hasattr(built_lib, n)]
    assert not missing, missing
    # and the Python binding table covers the same set
    assert set(ok._PROTOTYPES) == names
    assert built_lib.ookd_api_version() == 1


def test_header_is_valid_c99(tmp_path):
    import subprocess
    src = tmp_path / "use.c"
    src.write_text('#include "ookiedokie_amd.h"\n'
                   'int f(void) { ookd_rx_config c = {0}; ookd_message m; (void)m;\n'
                   '  return (int)sizeof(ookd_fsm_state) + (int)c.samples_per_buffer + OOKD_FILE_EOF; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only",
                        "-I", os.path.dirname(ok.HEADER_PATH), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_struct_layouts_match_header(tmp_path):
    """The ctypes mirrors against what a C compiler makes of the header (sizes and the offsets of the
    last members)."""
    import subprocess
    assert C.sizeof(ok.Message) == 48
    assert C.sizeof(ok.FsmState) == 64
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ookiedokie_amd.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(ookd_rx_config), sizeof(ookd_rx_stats),\n'
                   '  sizeof(ookd_message), sizeof(ookd_fsm_state), offsetof(ookd_rx_config, pipeline_chunk_samples),\n'
                   '  offsetof(ookd_rx_stats, pipeline_chunks)); return 0; }\n')
    exe = tmp_path / "sizes"
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.dirname(ok.HEADER_PATH), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True).stdout.split()]
    assert got == [C.sizeof(ok.RxConfig), C.sizeof(ok.RxStats), C.sizeof(ok.Message), C.sizeof(ok.FsmState),
                   ok.RxConfig.pipeline_chunk_samples.offset, ok.RxStats.pipeline_chunks.offset]


@pytest.mark.parametrize("name", ["fs32_fs4", "fs128_fs16_dec4", "unity1", "unity16"])
def test_filter_loader_matches_python_loader(oracle, name):
    f = ok.Filter.load(golden_path("filters", name))
    want = oracle.load_filter_json(golden_path("filters", name))
    assert f.num_stages == want.num_stages
    assert f.total_decimation == want.total_decimation
    for s in range(f.num_stages):
        d, taps = f.stage(s)
        assert d == int(want.decimation[s])
        assert (taps.view(np.uint32) == want.stage_taps(s).view(np.uint32)).all()


@pytest.mark.parametrize("name", ["p3l-nexa2012", "unknown-remote1"])
@pytest.mark.parametrize("rate", [3000000, 2000000, 750000])
def test_device_loader_matches_python_loader_and_oracle_windows(oracle, name, rate):
    d = ok.Device.load(golden_path("devices", name), rate)
    want, _ = oracle.load_device_json(golden_path("devices", name), rate)
    t = d.tables()
    assert t["state_names"] == want.state_names
    assert t["max_bits"] == want.max_bits and t["sample_rate"] == rate
    for key in ("state_duration_us", "state_timeout_us", "trig_begin", "trig_cond",
                "trig_action", "trig_next", "trig_duration_us"):
        assert (t[key] == getattr(want, key)).all(), key
    NONE = np.uint64(2 ** 64 - 1)
    for s in range(t["num_states"]):
        dur, to = int(t["state_duration_us"][s]), int(t["state_timeout_us"][s])
        if dur:
            assert (int(t["state_kmin"][s]), int(t["state_kmax"][s])) == oracle.duration_window(rate, dur)
        else:
            assert t["state_kmin"][s] == 0 and t["state_kmax"][s] == NONE
        assert int(t["state_kto"][s]) == (oracle.timeout_count(rate, to) if to else int(NONE))
    for i in range(t["num_triggers"]):
        dur = int(t["trig_duration_us"][i])
        if dur:
            assert (int(t["trig_kmin"][i]), int(t["trig_kmax"][i])) == oracle.duration_window(rate, dur)


def test_survey_count_table_at_3msps(vectors):
    d = ok.Device.load(golden_path("devices", "p3l-nexa2012"), 3000000)
    t = d.tables()
    g = vectors["G5"]
    for s in range(t["num_states"]):
        dur = str(int(t["state_duration_us"][s]))
        if dur in g["survey_windows"]:
            assert [int(t["state_kmin"][s]), int(t["state_kmax"][s])] == g["survey_windows"][dur]
        to = str(int(t["state_timeout_us"][s]))
        if to in g["survey_timeouts"]:
            assert int(t["state_kto"][s]) == g["survey_timeouts"][to]


def _write(tmp_path, name, obj_or_text):
    p = tmp_path / name
    p.write_text(obj_or_text if isinstance(obj_or_text, str) else json.dumps(obj_or_text))
    return str(p)


def test_filter_loader_rejects_what_the_reference_rejects(tmp_path):
    bad = [
        {"filter": {"stages": []}},                                 # fir.c:118-121
        {"filter": {"stages": [{"taps": []}]}},                     # fir.c:172-176
        {"filter": {"stages": [{"decimation": 0, "taps": [1]}]}},   # fir.c:148-152
        {"filter": {"stages": [{"decimation": 1.0, "taps": [1]}]}},  # must be integer
        {"filter": {"stages": [{"taps": [1, "x"]}]}},               # fir.c:217-222
        {"filter": {}},
        {"nofilter": 1},
        '{"filter": {"stages": [{"taps": [1]}], "stages": []}}',    # duplicate key
        '{"filter": {"stages": [{"taps": [1,]}]}}',                 # syntax
    ]
    for i, b in enumerate(bad):
        with pytest.raises(ok.OokdError):
            ok.Filter.load(_write(tmp_path, "f%d.json" % i, b))
    with pytest.raises(ok.OokdError):
        ok.Filter.load(str(tmp_path / "does-not-exist.json"))
    # defaults: decimation 1, integers accepted as taps
    f = ok.Filter.load(_write(tmp_path, "ok.json", {"filter": {"stages": [{"taps": [1, 2.5, -3e-1]}]}}))
    d, taps = f.stage(0)
    assert d == 1 and list(taps) == [np.float32(1), np.float32(2.5), np.float32(-0.3)]


def test_device_loader_state_slot_rules(tmp_path, oracle):
    base = json.load(open(golden_path("devices", "p3l-nexa2012")))
    # (1) states listed in another order.  get_or_reserve_state only forces
    #     "reset" into slot 0 while slot 0 is still free (state_machine.c:216-228):
    #     with "idle" listed first, idle takes slot 0 and IS the reset state.
    dev = json.loads(json.dumps(base))
    dev["device"]["states"] = dev["device"]["states"][1:] + dev["device"]["states"][:1]
    p = _write(tmp_path, "reordered.json", dev)
    t = ok.Device.load(p, 3000000).tables()
    want, _ = oracle.load_device_json(p, 3000000)
    assert t["state_names"] == want.state_names and t["state_names"][0] == "idle"
    assert (t["trig_next"] == want.trig_next).all()
    # (2) missing required members
    for key in ("description", "num_bits", "states", "fields", "name"):
        dev = json.loads(json.dumps(base))
        del dev["device"][key]
        with pytest.raises(ok.OokdError):
            ok.Device.load(_write(tmp_path, "no_%s.json" % key, dev), 3000000)
    # (3) invalid trigger condition / action, negative timeout
    dev = json.loads(json.dumps(base))
    dev["device"]["states"][1]["triggers"][0]["condition"] = "sometimes"
    with pytest.raises(ok.OokdError):
        ok.Device.load(_write(tmp_path, "badcond.json", dev), 3000000)
    dev = json.loads(json.dumps(base))
    dev["device"]["states"][2]["timeout_us"] = -5
    with pytest.raises(ok.OokdError):
        ok.Device.load(_write(tmp_path, "negto.json", dev), 3000000)
    # (4) condition names are case-insensitive (strcasecmp)
    dev = json.loads(json.dumps(base))
    dev["device"]["states"][0]["triggers"][0]["condition"] = "ALWAYS"
    assert ok.Device.load(_write(tmp_path, "case.json", dev), 3000000).num_bits == 36


def test_synth_tx_walk_matches_reference_sm_generate(oracle):
    """The product's tx walk (synth.cpp) gives the run lengths the
    reference's sm_generate produces for the same payload."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    for name, rate in (("p3l-nexa2012", 3000000), ("unknown-remote1", 3000000),
                       ("p3l-nexa2012", 1000000)):
        dev = ok.Device.load(golden_path("devices", name), rate)
        odev, _ = oracle.load_device_json(golden_path("devices", name), rate)
        syn = ok.Synth(dev, 3_000_000, seed=7, sample_rate=rate, noise=0, glitch_every=0,
                       random_phase=False, gap_us=(4000, 4000))
        assert syn.num_messages >= 2
        iq = syn.fill_host()
        assert not iq[1::2].any()
        level = (iq[0::2] != 0).astype(np.int8)
        for i in range(min(syn.num_messages, 3)):
            start, payload = syn.message(i)
            wave = oracle.RefSm(odev).generate(payload, 0.95)
            ref_i = oracle.ref_pack(wave)[0::2]
            n = min(ref_i.size, level.size - start)
            assert n > 1000
            assert (iq[0::2][start:start + n] == ref_i[:n]).all(), (name, i)
        # gap before the first message is the requested 4000 us
        assert syn.message(0)[0] == int(4000 * rate / 1e6 + 0.5)


@pytest.mark.parametrize("name,stuck", [("p3l-nexa2012", 36), ("unknown-remote1", 32)])
@pytest.mark.parametrize("rate", [3000000, 750000])
def test_scan_domain_of_the_shipped_devices(name, stuck, rate):
    """Host logic of the scan form (no GPU): span tables build (and their merged form -- one search per
    leaf -- says at every breakpoint, next to it and far beyond what the per-row searches say: a
    mismatch reports `built` = 0), every interval is a
    looked-up result, the level-aware closure finds the codes a glitch inside a bit gap
    leaves "stuck" (bit_off_time with fewer than max_bits bits, one table row), and the
    domain with their twins stays within 512 codes."""
    d = ok.Device.load(golden_path("devices", name), rate)
    out = (C.c_uint32 * 8)()
    assert ok.lib().ookd_scan_domain_info(d._h, 8192, 1, out) == 0
    built, intervals, need_sim, reach, nstuck, stuck_rows, domain, states = list(out)
    assert built == 1 and need_sim == 0 and states == 6
    assert intervals == 74
    assert nstuck == stuck and stuck_rows == 1
    base = states * (d.num_bits + 2) + 3
    depth = (domain - base) // nstuck
    assert domain == base + depth * nstuck and 3 <= depth <= 8 and domain <= 512
    assert base // 3 < reach <= domain          # far fewer than all codes are ever entered


def test_synth_is_deterministic_and_windowed():
    dev = ok.Device.load(golden_path("devices", "unknown-remote1"), 3000000)
    a = ok.Synth(dev, 500_000, seed=3).fill_host()
    b = ok.Synth(dev, 500_000, seed=3)
    assert (a == b.fill_host()).all()
    assert (a[2 * 123457:2 * 223457] == b.fill_host(123457, 100000)).all()
    assert not (a == ok.Synth(dev, 500_000, seed=4).fill_host()).all()
    assert np.abs(a).max() <= 1945 + 40


def test_compute_entry_points_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    f = ok.Filter.load(golden_path("filters", "fs32_fs4"))
    d = ok.Device.load(golden_path("devices", "p3l-nexa2012"), 3000000)
    with pytest.raises(ok.OokdError) as e:
        ok.Receiver(f, d, max_samples=1 << 20)
    assert "no CPU fallback" in str(e.value)
    with pytest.raises(ok.OokdError):
        ok.StreamFir(f, 4096)


def _build_c_example(tmp_path):
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "ookiedokie_amd", "lib")
    exe = str(tmp_path / "ookd_rx")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "ookd_rx.c"), "-o", exe, "-L" + libdir, "-lookiedokie_amd",
                    "-Wl,-rpath," + libdir], check=True)
    return exe


def test_c_host_example_builds_and_fails_loudly_without_gpu(tmp_path):
    """examples/ookd_rx.c is the C99 host INTEGRATION.md describes: it must build
    against the header with gcc alone and, like every compute entry, refuse to
    run without a HIP device."""
    import subprocess
    ok.lib()
    exe = _build_c_example(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([exe, "/nonexistent.sc16q11", golden_path("devices", "p3l-nexa2012"), "none", "3000000"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "no HIP device" in r.stderr
