#!/usr/bin/env python3
"""Generates tests/golden/formatter_vectors.json from the REFERENCE's own
formatter (oracle/_ref/libookref.so = /root/reference/src/formatter.c compiled
where it lies, see oracle/Makefile).  Run in the build container only:

    make -C oracle && python tests/golden/make_formatter_golden.py

The file holds data only: field descriptions (in the device-file schema),
payload bytes, and the strings / bytes the reference produced for them.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import oracle as O  # noqa: E402

FORMATS = ["hex", "unsigned decimal", "sign-magnitude", "two's complement", "float", "enumeration"]
VALUES = ["0", "1", "-1", "5", "-5", "0x7f", "255", "1e3", "3.5", "-2.25", "E1", "e2", "zz", "70000",
          "-70000", "4294967296", "18446744073709551615", "0x8000000000000000"]


def random_fields(rng, nbytes):
    fields = []
    for i in range(int(rng.integers(1, 6))):
        w = int(rng.integers(1, min(64, nbytes * 8) + 1))
        s = int(rng.integers(0, nbytes * 8 - w + 1))
        fmt = FORMATS[int(rng.integers(0, 6))]
        f = {"name": "f%d" % i, "start_bit": s, "end_bit": s + w - 1, "format": fmt,
             "endianness": ["big", "little"][int(rng.integers(0, 2))], "default": "0"}
        if rng.random() < 0.6:
            f["scaling"] = float(rng.choice([1, 0.1, 0.5, 2, -1, 10, 0.01]))
        if rng.random() < 0.6:
            f["offset"] = float(rng.choice([0, 32, -40, 0.5, 1000]))
        if fmt == "enumeration":
            f["enum_values"] = [{"string": "E%d" % k, "value": hex(int(rng.integers(0, min(2 ** w, 2 ** 63))))}
                                for k in range(3)]
        fields.append(f)
    return fields


def one_case(rng, fields, num_bits, npay):
    nbytes = (num_bits + 7) // 8
    try:
        rf = O.RefFormatter(fields, num_bits)
    except ValueError:
        return {"fields": fields, "num_bits": num_bits, "accepted": False}
    case = {"fields": fields, "num_bits": num_bits, "accepted": True,
            "default_data": rf.default_data().tobytes().hex(), "payloads": [], "sets": []}
    for j in range(npay):
        pay = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
        if j % 5 == 4:
            pay[:] = 0xFF
        case["payloads"].append({"data": pay.tobytes().hex(), "keyval": rf.data_to_keyval(pay.tobytes())})
    for _ in range(12):
        fd = fields[int(rng.integers(0, len(fields)))]
        v = VALUES[int(rng.integers(0, len(VALUES)))]
        base = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
        try:
            out = rf.keyval_to_data([(fd["name"], v)], base).tobytes().hex()
        except ValueError:
            out = None
        case["sets"].append({"field": fd["name"], "value": v, "base": base.tobytes().hex(), "data": out})
    return case


def main():
    O.build()
    rng = np.random.default_rng(20261004)
    cases = []
    for name in ("p3l-nexa2012", "unknown-remote1"):
        with open(os.path.join(HERE, "devices", name + ".json")) as f:
            dev = json.load(f)["device"]
        c = one_case(rng, dev["fields"], dev["num_bits"], 40)
        c["device"] = name
        cases.append(c)
    while len(cases) < 90:
        nb = int(rng.integers(1, 257))
        cases.append(one_case(rng, random_fields(rng, (nb + 7) // 8), nb, 6))
    with open(os.path.join(HERE, "formatter_vectors.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_formatter_golden.py",
                   "source": "reference src/formatter.c via oracle/_ref (gcc, x86-64)",
                   "cases": cases}, f, separators=(",", ":"))
    print("wrote %d cases (%d accepted)" % (len(cases), sum(c["accepted"] for c in cases)))


if __name__ == "__main__":
    main()
