"""Pins the CPU oracle (oracle/ook_oracle.c) before anything trusts it.

Sources of truth, strongest first:
  1. the reference's own state_machine.c / complexf.h compiled into
     oracle/_ref (differential, sample for sample);
  2. numbers SURVEY.md 8(c) recorded from the real ookiedokie binary
     (tests/golden/vectors.json "survey" entries);
  3. known-answer FIR vectors of the reference's manual harness
     (src/test/fir_test.c + src/matlab/gen_samples.m + unity filters) and an
     independent numpy float32 restatement of src/fir.c:313-318.
"""
import numpy as np
import pytest

from tests.helpers import harness_tone, steady_state_response, edges_of, golden_path, iq_from_rle, stream_from_runs

RATE = 3000000


def _dev(O, name, rate=RATE):
    return O.load_device_json(golden_path("devices", name), rate)[0]


def _fir(O, name):
    return O.load_filter_json(golden_path("filters", name))


def np_fir_sequential(x, taps, decim):
    """Independent restatement of one stage (src/fir.c:302-353): float32
    products and sums rounded separately, tap 0 first, zero history, outputs
    at input indices D-1, 2D-1, ..."""
    x = np.asarray(x, dtype=np.float32)
    taps = np.asarray(taps, dtype=np.float32)
    n = x.shape[0]
    idx = np.arange(decim - 1, n, decim)
    acc = np.zeros((idx.size, 2), dtype=np.float32)
    xp = np.concatenate([np.zeros((taps.size, 2), np.float32), x])
    for k in range(taps.size):
        prod = (taps[k] * xp[idx + taps.size - k]).astype(np.float32)
        acc = (acc + prod).astype(np.float32)
    return acc


# ---------------------------------------------------------------- unpack ----

def test_unpack_all_int16_values_match_reference(oracle):
    v = np.arange(-32768, 32768, dtype=np.int16)
    iq = np.empty(2 * v.size, dtype=np.int16)
    iq[0::2] = v
    iq[1::2] = v[::-1]
    got = oracle.unpack(iq)
    assert got.dtype == np.float32
    assert (got[:, 0] == v.astype(np.float32) / np.float32(2048)).all()
    if oracle.have_ref():
        assert (got.view(np.uint32) == oracle.ref_unpack(iq).view(np.uint32)).all()


def test_threshold_matches_reference_near_boundary(oracle):
    # default threshold 0.1f: the smallest power whose sqrtf is >= thr is one
    # ulp BELOW thr*thr (SURVEY.md hard part 3).
    thr = np.float32(0.1)
    base = np.float32(thr * thr).view(np.uint32)
    p = (base + np.arange(-200, 200, dtype=np.int64)).astype(np.uint32).view(np.float32)
    x = np.zeros((p.size, 2), dtype=np.float32)
    x[:, 0] = np.sqrt(p.astype(np.float64)).astype(np.float32)
    rng = np.random.default_rng(3)
    y = rng.uniform(-0.2, 0.2, size=(50000, 2)).astype(np.float32)
    allx = np.concatenate([x, y])
    got = oracle.threshold(allx, 0.1)
    want = (np.sqrt((allx[:, 0] * allx[:, 0] + allx[:, 1] * allx[:, 1])
                    .astype(np.float32)) >= thr)
    assert (got.astype(bool) == want).all()
    if oracle.have_ref():
        assert (got == oracle.ref_threshold(allx, 0.1)).all()


# ------------------------------------------------------------------- FIR ----

def _impulse(n=100, at=49, q=False):
    x = np.zeros((n, 2), dtype=np.float32)
    x[at, 1 if q else 0] = 1.0
    return x


@pytest.mark.parametrize("chunk", [None, 1, 7, 32, 33])
def test_fir_unity16_impulse_is_boxcar(oracle, chunk):
    # gen_samples.m:5-8 impulse at sample 50 (index 49) of 100; unity16 -> ones at 49..64
    y = oracle.fir_run(_fir(oracle, "unity16"), _impulse(), chunk)
    want = np.zeros(100, dtype=np.float32)
    want[49:65] = 1
    assert (y[:, 0] == want).all() and not y[:, 1].any()


@pytest.mark.parametrize("chunk", [None, 7, 32, 33])
def test_fir_fs32_impulse_returns_taps_exactly(oracle, chunk):
    f = _fir(oracle, "fs32_fs4")
    y = oracle.fir_run(f, _impulse(), chunk)
    assert (y[49:81, 0].view(np.uint32) == f.taps.view(np.uint32)).all()
    assert not y[:49, 0].any() and not y[81:, 0].any() and not y[:, 1].any()
    yq = oracle.fir_run(f, _impulse(q=True), chunk)
    assert (yq[49:81, 1] == f.taps).all() and not yq[:, 0].any()


def test_fir_unity1_is_identity(oracle):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1000, 2)).astype(np.float32)
    y = oracle.fir_run(_fir(oracle, "unity1"), x, 17)
    assert (y.view(np.uint32) == x.view(np.uint32)).all()


@pytest.mark.parametrize("chunk", [None, 33, 8192])
def test_fir_dec4_matches_numpy_restatement(oracle, chunk):
    f = _fir(oracle, "fs128_fs16_dec4")
    y = oracle.fir_run(f, _impulse(), chunk)
    assert y.shape[0] == 25
    s1 = np_fir_sequential(_impulse(), f.stage_taps(0), 2)
    s2 = np_fir_sequential(s1, f.stage_taps(1), 2)
    assert (y.view(np.uint32) == s2.view(np.uint32)).all()


def test_fir_long_impulse_returns_taps_exactly(oracle):
    # gen_samples.m:13-16: 10^6 samples, impulse at sample 1000 (index 999)
    f = _fir(oracle, "fs32_fs4")
    y = oracle.fir_run(f, _impulse(n=1000000, at=999), 8192)
    assert (y[999:1031, 0].view(np.uint32) == f.taps.view(np.uint32)).all()
    assert not y[:999].any() and not y[1031:].any()


@pytest.mark.parametrize("name", ["fs32_fs4", "fs128_fs16_dec4", "unity16"])
@pytest.mark.parametrize("period", [4.0, 32.0])
def test_fir_tone_steady_state_is_the_frequency_response(oracle, name, period):
    """gen_samples.m:19-34 (tones at Fs/4 and Fs/32, 10^6 samples): behind the start-up the output is
    H(w) * tone -- a known answer that does not come from any restatement of the filter loop.
    Tolerance: float32 accumulation over T taps, 1e-5 of sum|h| (the north_star's FIR tolerance)."""
    f = _fir(oracle, name)
    x = harness_tone(1000000, period)
    y = oracle.fir_run(f, x, 8192)
    taps = [f.stage_taps(s) for s in range(f.num_stages)]
    decs = [int(d) for d in f.decimation]
    want, first = steady_state_response(taps, decs, x.shape[0], period)
    assert y.shape[0] == want.size
    tol = 1e-5 * float(np.prod([np.abs(t).sum() for t in taps]))
    got = y[:, 0].astype(np.float64) + 1j * y[:, 1].astype(np.float64)
    assert np.abs(got[first + 64:] - want[first + 64:]).max() <= tol
    # fs32_fs4 passes Fs/32 and stops Fs/4 (filters/README.md: cutoff Fs/4... the name says it)
    if name == "fs32_fs4":
        mag = np.abs(got[2000:]).mean()
        assert (mag > 0.9) if period == 32.0 else (mag < 0.05)


def test_fir_two_tone_is_the_sum_of_the_tones(oracle):
    # gen_samples.m:36-38: (tone_fs4 + tone_fs32) / 2
    f = _fir(oracle, "fs32_fs4")
    x = (harness_tone(200000, 4.0).astype(np.float64) + harness_tone(200000, 32.0)) / 2.0
    y = oracle.fir_run(f, x.astype(np.float32), 4096)
    w4, first = steady_state_response([f.taps], [1], 200000, 4.0)
    w32, _ = steady_state_response([f.taps], [1], 200000, 32.0)
    got = y[:, 0].astype(np.float64) + 1j * y[:, 1].astype(np.float64)
    assert np.abs(got[first + 64:] - (w4 + w32)[first + 64:] / 2.0).max() <= 1e-5 * float(np.abs(f.taps).sum())


def test_fir_noisy_signal_bit_identical_to_numpy_restatement(oracle, vectors):
    # G6 shape: first 200k samples of G1 + uniform +-40 LSB noise on I and Q.
    g1 = vectors["G1"]
    iq = iq_from_rle(g1["i_rle"], g1["num_samples"])[:400000].copy()
    rng = np.random.default_rng(1)
    iq += rng.integers(-40, 41, size=iq.size).astype(np.int16)
    x = oracle.unpack(iq)
    for name in ("fs32_fs4", "fs128_fs16_dec4"):
        f = _fir(oracle, name)
        y = oracle.fir_run(f, x, 8192)
        want = x
        for s in range(f.num_stages):
            want = np_fir_sequential(want, f.stage_taps(s), int(f.decimation[s]))
        assert (y.view(np.uint32) == want.view(np.uint32)).all(), name


# ------------------------------------------------------------ whole path ----

def test_g1_survey_numbers(oracle, vectors):
    g = vectors["G1"]
    iq = iq_from_rle(g["i_rle"], g["num_samples"])
    dev = _dev(oracle, g["device"])
    fir = _fir(oracle, g["filter"])
    r = oracle.rx(iq, fir, 0.1, dev, g["spb"], want_bits=True)
    e = edges_of(r.bits)
    s = g["survey"]
    assert len(e) == s["num_edges"]
    assert list(e[:6]) == s["first_edges"]
    assert list(e) == g["oracle"]["edges"]
    assert list(r.msg_samples) == s["msg_samples"]
    assert all(r.payload_bits(i, dev.max_bits) == s["payload_bits"]
               for i in range(len(r.msg_samples)))
    # final partial buffer is zero padded and processed (bladeRF_file.c:113-117)
    assert r.decimated == -(-g["num_samples"] // g["spb"]) * g["spb"]
    for spb in s["same_msgs_at_spb"]:
        assert list(oracle.rx(iq, fir, 0.1, dev, spb).msg_samples) == s["msg_samples"]


def test_g1_default_dec4_filter(oracle, vectors):
    g = vectors["G1"]
    iq = iq_from_rle(g["i_rle"], g["num_samples"])
    fir = _fir(oracle, "fs128_fs16_dec4")
    dev = _dev(oracle, g["device"], RATE // fir.total_decimation)
    r = oracle.rx(iq, fir, 0.1, dev, 8192)
    assert list(r.msg_samples) == g["oracle"]["dec4_msg_samples"]
    assert len(r.msg_samples) == 3


def test_g2_survey_numbers(oracle, vectors):
    g = vectors["G2"]
    iq = iq_from_rle(g["i_rle"], g["num_samples"])
    dev = _dev(oracle, g["device"])
    r = oracle.rx(iq, _fir(oracle, g["filter"]), 0.1, dev, g["spb"])
    assert list(r.msg_samples) == g["survey"]["msg_samples"]
    assert all(r.payload_bits(i, 32) == g["survey"]["payload_bits"] for i in range(2))


def test_g3_error_drops_rest_of_buffer(oracle, vectors):
    g1, g3 = vectors["G1"], vectors["G3"]
    iq = iq_from_rle(g1["i_rle"], g1["num_samples"])
    a, b, v = g3["glitch"]
    iq[2 * a:2 * b:2] = v
    dev = _dev(oracle, g1["device"])
    fir = _fir(oracle, g1["filter"])
    for spb, n in g3["survey_num_msgs"].items():
        r = oracle.rx(iq, fir, 0.1, dev, int(spb))
        assert len(r.msg_samples) == n, spb
        assert list(r.err_samples) == g3["oracle"][spb]["err_samples"]


def test_no_filter_path_and_short_capture(oracle, vectors):
    g = vectors["G1"]
    iq = iq_from_rle(g["i_rle"], g["num_samples"])
    dev = _dev(oracle, g["device"])
    r = oracle.rx(iq, None, 0.1, dev, 8192)
    assert len(r.msg_samples) == 3
    # empty capture: first read returns 0 items => EOF, nothing processed
    r0 = oracle.rx(np.zeros(0, np.int16), None, 0.1, dev, 8192)
    assert r0.decimated == 0 and len(r0.msg_samples) == 0
    # capture shorter than one buffer is padded to one buffer
    r1 = oracle.rx(iq[:2 * 100], None, 0.1, dev, 8192, want_bits=True)
    assert r1.decimated == 8192


# ------------------------------------------------------- state machine ----

def test_g5_integer_windows(oracle, vectors):
    g = vectors["G5"]
    for d, (lo, hi) in g["survey_windows"].items():
        assert oracle.duration_window(g["rate"], int(d)) == (lo, hi)
    for t, k in g["survey_timeouts"].items():
        assert oracle.timeout_count(g["rate"], int(t)) == k


def test_g7_tolerance_boundaries(oracle, vectors):
    g = vectors["G7"]
    dev = _dev(oracle, g["device"])
    from tests.golden import make_golden as mg
    for c in g["cases"]:
        runs = mg.p3l_runs(g["payload_bits"], **{c["param"]: c["value"]})
        ms, pay, es = oracle.sm_stream(dev, stream_from_runs(runs), g["spb"])
        assert len(ms) == c["num_msgs"], c
        assert list(ms) == c["msg_samples"]


def test_random_streams_match_reference_fsm_fixtures(oracle, vectors):
    for case in vectors["random_streams"]:
        dev = _dev(oracle, case["device"], case["rate"])
        stream = stream_from_runs(case["runs"])
        for buf, want in case["ref_fsm"].items():
            ms, pay, es = oracle.sm_stream(dev, stream, int(buf))
            assert list(ms) == want["msg_samples"]
            assert [bytes(p).hex() for p in pay] == want["payloads"]
            assert list(es) == want["err_samples"]


def _random_fsm(O, rng, rate, ns=None, max_triggers=4):
    """A random (mostly nonsensical) device: exercises every trigger kind,
    reset without 'always', states with zero-duration windows, etc."""
    import numpy as np
    ns = int(rng.integers(2, 6)) if ns is None else ns
    durs = [0, 0, 30, 100, 250]
    sdur = rng.choice(durs, ns)
    sto = rng.choice([0, 0, 50, 400, 1000], ns)
    tbeg, cond, act, nxt, tdur = [0], [], [], [], []
    for s in range(ns):
        nt = int(rng.integers(1, max_triggers))
        for _ in range(nt):
            if s == 0 and rng.random() < 0.6:
                c = 1
            else:
                c = int(rng.choice([1, 2, 2, 3, 3, 4, 5]))
            cond.append(c)
            act.append(int(rng.choice([1, 1, 2, 3, 4])))
            nxt.append(int(rng.integers(0, ns)))
            tdur.append(int(rng.choice([0, 0, 0, 60, 200])))
        tbeg.append(len(cond))
    return O.FsmDesc(
        state_names=["reset"] + ["s%d" % i for i in range(1, ns)],
        max_bits=int(rng.integers(1, 41)), sample_rate=rate,
        state_duration_us=np.array(sdur, dtype=np.uint64),
        state_timeout_us=np.array(sto, dtype=np.uint64),
        trig_begin=np.array(tbeg, dtype=np.uint32),
        trig_cond=np.array(cond, dtype=np.uint8),
        trig_action=np.array(act, dtype=np.uint8),
        trig_next=np.array(nxt, dtype=np.uint32),
        trig_duration_us=np.array(tdur, dtype=np.uint64))


def test_random_devices_differential_vs_reference_fsm(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(77)
    for it in range(60):
        rate = int(rng.choice([3000000, 1000000, 750000, 48000]))
        dev = _random_fsm(oracle, rng, rate)
        scale = rate / 1e6
        runs = [max(1, int(rng.choice([30, 60, 100, 200, 250, 400, 1000])
                           * scale * rng.uniform(0.8, 1.2)))
                for _ in range(int(rng.integers(20, 200)))]
        stream = stream_from_runs(runs)
        for buf in (97, 4096):
            ms, pay, es = oracle.RefSm(dev).stream(stream, buf)
            oms, opay, oes = oracle.sm_stream(dev, stream, buf)
            assert list(ms) == list(oms), it
            assert (pay == opay).all(), it
            assert list(es) == list(oes), it


def test_oracle_under_sanitizers(oracle, vectors):
    """ASan/UBSan build of the restatement runs the G3 case cleanly."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(oracle.__file__)
    out = subprocess.run(["make", "-C", here, os.path.join(here, "libook_oracle_asan.so")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    code = (
        "import ctypes,sys,os,json,numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import oracle as O\n"
        "O._lib=None\n"
        "import oracle\n"
        "orig=os.path.join(os.path.dirname(O.__file__),'libook_oracle.so')\n"
        "O_asan=os.path.join(os.path.dirname(O.__file__),'libook_oracle_asan.so')\n"
        "import ctypes as C\n"
        "real=C.CDLL\n"
        "C.CDLL=lambda p,*a,**k: real(O_asan if p==orig else p,*a,**k)\n"
        "from tests.helpers import iq_from_rle, golden_path\n"
        "v=json.load(open(os.path.join(%r,'vectors.json')))\n"
        "g=v['G1']; iq=iq_from_rle(g['i_rle'],g['num_samples'])[:2*300000]\n"
        "iq[2*9000:2*9300:2]=1945\n"
        "dev=O.load_device_json(golden_path('devices','p3l-nexa2012'),3000000)[0]\n"
        "fir=O.load_filter_json(golden_path('filters','fs128_fs16_dec4'))\n"
        "r=O.rx(iq,fir,0.1,dev.with_rate(750000),1000,want_bits=True,want_fir=True)\n"
        "print('ok',r.decimated,len(r.err_samples))\n"
    ) % (os.path.dirname(here), os.path.join(os.path.dirname(here), "tests"),
         os.path.join(os.path.dirname(here), "tests", "golden"))
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True,
                          text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert res.returncode == 0 and res.stdout.startswith("ok"), res.stdout + res.stderr
