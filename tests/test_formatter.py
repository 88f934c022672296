"""Host side of a decoded message: payload -> field
text -> printed text, and the inverse used by the tx side.

Pinned three ways:
  * tests/golden/formatter_vectors.json -- outputs of the reference's own
    formatter.c (compiled by oracle/Makefile) on the shipped devices and on
    random field layouts, incl. the widths where its C is formally undefined;
  * the same comparison live against oracle/_ref when it is present;
  * the CSV lines the real binary printed for G1 / G2.
rx_print has no reference build (static function of a file that needs the
SDR/FIR/jansson parts): its expectations below are derived by hand from its
format strings, ookiedokie.c:181-220.
"""
import copy
import json
import os

import numpy as np
import pytest

import ookiedokie_amd as ok
from tests.helpers import GOLDEN, golden_path

RATE = 3_000_000


@pytest.fixture(scope="module")
def vectors():
    with open(os.path.join(GOLDEN, "formatter_vectors.json")) as f:
        return json.load(f)["cases"]


@pytest.fixture(scope="module")
def template():
    with open(golden_path("devices", "p3l-nexa2012")) as f:
        return json.load(f)


def _device_file(tmp_path, template, fields, num_bits, ts_mode=None, name="case.json"):
    dev = copy.deepcopy(template)
    dev["device"]["fields"] = fields
    dev["device"]["num_bits"] = num_bits
    dev["device"].pop("ts_mode", None)
    if ts_mode is not None:
        dev["device"]["ts_mode"] = ts_mode
    p = tmp_path / name
    p.write_text(json.dumps(dev))
    return str(p)


def _formatter(tmp_path, template, fields, num_bits, **kw):
    d = ok.Device.load(_device_file(tmp_path, template, fields, num_bits, **kw), RATE)
    return ok.Formatter(d)


def _check_case(f, case):
    assert f.default_data().tobytes().hex() == case["default_data"]
    for p in case["payloads"]:
        got = f.data_to_keyval(bytes.fromhex(p["data"]))
        assert got == [tuple(kv) for kv in p["keyval"]], p["data"]
    for s in case["sets"]:
        base = np.frombuffer(bytes.fromhex(s["base"]), dtype=np.uint8)
        try:
            # names are matched case-insensitively (formatter.c:809)
            got = f.keyval_to_data([(s["field"].upper(), s["value"])], base).tobytes().hex()
        except ok.OokdError:
            got = None
        assert got == s["data"], s


def test_golden_vectors(vectors, template, tmp_path):
    ran = 0
    for i, case in enumerate(vectors):
        if not case["accepted"]:
            # the reference's create_formatter fails (bad default for the field's
            # format / range): the device must not load here either
            with pytest.raises(ok.OokdError):
                _formatter(tmp_path, template, case["fields"], case["num_bits"], name="c%d.json" % i)
            continue
        _check_case(_formatter(tmp_path, template, case["fields"], case["num_bits"], name="c%d.json" % i), case)
        ran += 1
    assert ran >= 40


def test_shipped_devices_keep_their_timestamp_mode(vectors):
    for case in vectors:
        if "device" not in case:
            continue
        d = ok.Device.load(golden_path("devices", case["device"]), RATE)
        f = ok.Formatter(d)
        _check_case(f, case)
        with open(golden_path("devices", case["device"])) as fh:
            want = json.load(fh)["device"].get("ts_mode", "none")
        modes = {"none": 0, "unix": 1, "unix-frac": 2, "datetime-24": 3, "datetime-ampm": 4}
        assert f.ts_mode == modes[want]


def test_live_reference_differential(template, tmp_path):
    import oracle as O
    if not os.path.exists(os.path.join(os.path.dirname(O.__file__), "_ref", "libookref.so")):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    import sys
    sys.path.insert(0, GOLDEN)
    import make_formatter_golden as G
    rng = np.random.default_rng(7)
    compared = 0
    for i in range(120):
        nb = int(rng.integers(1, 257))
        fields = G.random_fields(rng, (nb + 7) // 8)
        try:
            rf = O.RefFormatter(fields, nb)
        except ValueError:
            with pytest.raises(ok.OokdError):
                _formatter(tmp_path, template, fields, nb, name="l%d.json" % i)
            continue
        f = _formatter(tmp_path, template, fields, nb, name="l%d.json" % i)
        for _ in range(10):
            pay = rng.integers(0, 256, size=(nb + 7) // 8, dtype=np.uint8)
            assert f.data_to_keyval(pay) == rf.data_to_keyval(pay.tobytes())
            compared += 1
    assert compared > 300


# ---- what the real binary printed -----------------------------------

def test_g1_tx_parameters_and_csv_line(template, tmp_path):
    # G1: --tx ... -d p3l-nexa2012 -p "Channel=2" -p "Temperature (C)=21.5" gave payload
    # 1001 1111 0101 0101 0000 1101 0111 0000 0000 (first bit first) and the rx side
    # printed 0x27,0xd5,2,21.500,70.700,0x00
    with open(golden_path("devices", "p3l-nexa2012")) as fh:
        dev = json.load(fh)["device"]
    f = _formatter(tmp_path, template, dev["fields"], dev["num_bits"])      # ts_mode dropped
    data = f.keyval_to_data([("Channel", "2"), ("Temperature (C)", "21.5")])
    bits = "".join(str(int(b)) for b in np.unpackbits(data, bitorder="little")[:36])
    assert bits == "100111110101010100001101011100000000"
    assert f.print_record([data], ok.RX_FMT_CSV) == (
        "Preamble,Unknown-1,Channel,Temperature (C),Temperature (F),Unknown-2\n"
        "0x27,0xd5,2,21.500,70.700,0x00\n")
    # heading only once (ookiedokie.c:190-198)
    assert f.print_record([data], ok.RX_FMT_CSV) == "0x27,0xd5,2,21.500,70.700,0x00\n"


def test_g2_tx_parameters_and_csv_line():
    d = ok.Device.load(golden_path("devices", "unknown-remote1"), RATE)
    f = ok.Formatter(d)
    data = f.keyval_to_data([("Button", "P2"), ("ID", "0x42")])
    bits = "".join(str(int(b)) for b in np.unpackbits(data, bitorder="little")[:32])
    assert bits == "01011101010000100110000010011111"
    f.first_print = False
    assert f.print_record([data], ok.RX_FMT_CSV) == "0x5d,0x42,P2\n"


# ---- rx_print layout ------------------------------------------------------------------

FIELDS = [
    {"name": "A", "start_bit": 0, "end_bit": 7, "format": "hex", "endianness": "big", "default": "0"},
    {"name": "A rather long field name, longer than twenty", "start_bit": 8, "end_bit": 15,
     "format": "unsigned decimal", "endianness": "little", "default": "0"},
]


def test_pretty_layout(template, tmp_path):
    f = _formatter(tmp_path, template, FIELDS, 16)
    text = f.print_record([bytes([0x80, 7])], ok.RX_FMT_PRETTY)
    # "%20s : %s\n" per pair, one blank line per record (ookiedokie.c:209-215)
    assert text == ("                   A : 0x01\n"
                    "A rather long field name, longer than twenty : 7\n"
                    "\n")


def test_records_group_the_messages_of_one_buffer(template, tmp_path):
    f = _formatter(tmp_path, template, FIELDS[:1], 8)
    # device_process appends every message of a buffer to ONE keyval list
    # (device.c:634-658), so two messages in a buffer share a CSV line -- and a
    # heading, if it is the first print
    assert f.print_record([b"\x80", b"\x40"], ok.RX_FMT_CSV) == "A,A\n0x01,0x02\n"
    assert f.print_record([b"\xc0"], ok.RX_FMT_CSV) == "0x03\n"
    assert f.print_record([], ok.RX_FMT_CSV) == ""          # nothing decoded: nothing printed (:284)
    assert f.print_record([b"\x80", b"\x40"], ok.RX_FMT_PRETTY) == (
        "                   A : 0x01\n                   A : 0x02\n\n")


def test_print_messages_groups_by_buffer(template, tmp_path):
    f = _formatter(tmp_path, template, FIELDS[:1], 8)
    pay = np.zeros((4, 1), dtype=np.uint8)
    pay[:, 0] = [0x80, 0x40, 0xC0, 0x20]
    # spb 1000, decimation 4: decimated sample j leaves the filter in buffer
    # ceil(4 (j+1) / 1000) - 1 -> 249 | 250, 499 | 500
    res = ok.RxResult(np.zeros(4, np.uint32), np.array([249, 250, 499, 500], np.uint64), pay, {})
    assert f.print_messages(res, 1000, 4, ok.RX_FMT_CSV) == "A\n0x01\n0x02,0x03\n0x04\n"
    # a new capture always starts a new record
    res = ok.RxResult(np.array([0, 1], np.uint32), np.array([10, 10], np.uint64), pay[:2], {})
    assert f.print_messages(res, 1000, 4, ok.RX_FMT_CSV) == "0x01\n0x02\n"


def test_timestamp_pair_leads_each_message(template, tmp_path):
    for mode, pattern in (("unix", r"^\d{9,}$"), ("unix-frac", r"^\d+\.\d{6}$"),
                          ("datetime-24", r"^\d{4}-\d\d-\d\d \d\d:\d\d:\d\d$"),
                          ("datetime-ampm", r"^\d{4}-\d\d-\d\d \d\d:\d\d:\d\d [AP]M$")):
        import re
        f = _formatter(tmp_path, template, FIELDS[:1], 8, ts_mode=mode, name="ts_%s.json" % mode)
        lines = f.print_record([b"\x80", b"\x80"], ok.RX_FMT_CSV).split("\n")
        assert lines[0] == "Decode Timestamp,A,Decode Timestamp,A"
        vals = lines[1].split(",")
        assert re.match(pattern, vals[0]) and re.match(pattern, vals[2]), vals
        assert vals[1] == vals[3] == "0x01"


# ---- loader refusals (device.c:255-499, formatter.c:259-344) -------------------------------

@pytest.mark.parametrize("mutate, text", [
    (lambda f: f.pop("default"), "Failed to get default"),
    (lambda f: f.update(default="zz"), "Invalid default value"),
    (lambda f: f.update(default="256"), "Invalid default value"),          # too large for 8 bits
    (lambda f: f.update(start_bit=20, end_bit=24), "beyond the device"),
    (lambda f: f.update(start_bit=5, end_bit=4), "End bit must be >= start bit"),
    (lambda f: f.update(format="octal"), "Invalid format"),
    (lambda f: f.update(endianness="middle"), "Invalid endianness"),
    (lambda f: f.update(format="enumeration"), "enum_values"),
    (lambda f: f.update(format="enumeration", enum_values=[]), "1 or more values"),
    (lambda f: f.update(format="enumeration", enum_values=[{"string": "x", "value": "1"},
                                                           {"string": "X", "value": "2"}]),
     "Duplicate enumeration name"),
])
def test_bad_field_descriptions_fail_like_the_reference(template, tmp_path, mutate, text):
    fld = copy.deepcopy(FIELDS[0])
    mutate(fld)
    with pytest.raises(ok.OokdError) as e:
        _formatter(tmp_path, template, [fld], 16)
    assert text in str(e.value)
