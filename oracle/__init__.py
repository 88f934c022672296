"""CPU oracle for the OOKiedokie rx hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package, and only as the checker.  The
product (``ookiedokie_amd``) never imports it.

Contents
--------
* ``libook_oracle.so``  -- our plain-C restatement (``ook_oracle.c``).
* ``_ref/libookref.so`` -- the reference's own ``state_machine.c`` + ``log.c``
  + ``complexf.h`` compiled where they lie with our driver (``ref_driver.c``);
  built only in the container that has ``/root/reference``, shipped prebuilt
  to the GPU box.
* Python loaders that turn the device / filter JSON into flat tables using
  Python's own ``json`` module, i.e. independently of the product's C++ JSON
  reader (they follow src/device.c:76-193, :206-258 and
  src/state_machine.c:208-335 for naming, src/fir.c:118-225 for filters).
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_PAYLOAD = 64

COND = {"always": 1, "pulse_start": 2, "pulse_end": 3, "timeout": 4,
        "msg_complete": 5}
ACTION = {"none": 1, "append_0": 2, "append_1": 3, "output_data": 4}


def build(verbose: bool = False) -> None:
    """Compile the oracle (and oracle/_ref when the reference tree exists)."""
    out = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if verbose:
        print(out.stdout)


# --------------------------------------------------------------------------
# descriptors
# --------------------------------------------------------------------------

class _FsmDescC(C.Structure):
    _fields_ = [
        ("num_states", C.c_uint32), ("max_bits", C.c_uint32),
        ("sample_rate", C.c_uint32), ("num_triggers", C.c_uint32),
        ("state_duration_us", C.c_void_p), ("state_timeout_us", C.c_void_p),
        ("trig_begin", C.c_void_p), ("trig_cond", C.c_void_p),
        ("trig_action", C.c_void_p), ("trig_next", C.c_void_p),
        ("trig_duration_us", C.c_void_p),
    ]


class _FirDescC(C.Structure):
    _fields_ = [
        ("num_stages", C.c_uint32), ("decimation", C.c_void_p),
        ("num_taps", C.c_void_p), ("taps", C.c_void_p),
    ]


class _MsgC(C.Structure):
    _fields_ = [("sample", C.c_uint64), ("payload", C.c_uint8 * MAX_PAYLOAD)]


@dataclass
class FsmDesc:
    """Flat state machine tables; state 0 is the reset state."""
    state_names: List[str]
    max_bits: int
    sample_rate: int
    state_duration_us: np.ndarray
    state_timeout_us: np.ndarray
    trig_begin: np.ndarray
    trig_cond: np.ndarray
    trig_action: np.ndarray
    trig_next: np.ndarray
    trig_duration_us: np.ndarray

    @property
    def num_states(self) -> int:
        return len(self.state_duration_us)

    @property
    def payload_bytes(self) -> int:
        return (self.max_bits + 7) // 8

    def with_rate(self, rate: int) -> "FsmDesc":
        d = FsmDesc(**self.__dict__)
        d.sample_rate = int(rate)
        return d

    def c_struct(self) -> _FsmDescC:
        s = _FsmDescC()
        s.num_states = self.num_states
        s.max_bits = self.max_bits
        s.sample_rate = self.sample_rate
        s.num_triggers = len(self.trig_cond)
        for name in ("state_duration_us", "state_timeout_us", "trig_begin",
                     "trig_cond", "trig_action", "trig_next",
                     "trig_duration_us"):
            setattr(s, name, getattr(self, name).ctypes.data)
        return s


@dataclass
class FirDesc:
    decimation: np.ndarray
    num_taps: np.ndarray
    taps: np.ndarray            # float32, concatenated

    @property
    def num_stages(self) -> int:
        return len(self.decimation)

    @property
    def total_decimation(self) -> int:
        return int(np.prod(self.decimation.astype(np.int64)))

    def stage_taps(self, s: int) -> np.ndarray:
        off = int(self.num_taps[:s].sum())
        return self.taps[off:off + int(self.num_taps[s])]

    def c_struct(self) -> _FirDescC:
        s = _FirDescC()
        s.num_stages = self.num_stages
        s.decimation = self.decimation.ctypes.data
        s.num_taps = self.num_taps.ctypes.data
        s.taps = self.taps.ctypes.data
        return s


def make_fir(stages: List[Tuple[int, np.ndarray]]) -> FirDesc:
    dec = np.array([d for d, _ in stages], dtype=np.uint32)
    nt = np.array([len(t) for _, t in stages], dtype=np.uint32)
    taps = np.concatenate([np.asarray(t, dtype=np.float64).astype(np.float32)
                           for _, t in stages]).astype(np.float32)
    return FirDesc(dec, nt, np.ascontiguousarray(taps))


def load_filter_json(path: str) -> FirDesc:
    """src/fir.c:87-225: {"filter":{"stages":[{"decimation":int?,"taps":[..]}]}}.

    decimation defaults to 1 (:155-157) and must be > 0 (:149); taps are JSON
    numbers cast double -> float (:224)."""
    with open(path) as f:
        root = json.load(f)
    stages = root["filter"]["stages"]
    if not isinstance(stages, list) or not stages:
        raise ValueError("filter must have 1 or more stages")
    out = []
    for st in stages:
        dec = st.get("decimation", 1)
        if not isinstance(dec, int) or isinstance(dec, bool) or dec <= 0:
            raise ValueError("bad decimation")
        taps = st["taps"]
        if not isinstance(taps, list) or not taps:
            raise ValueError("stage must have 1 or more taps")
        out.append((dec, np.array(taps, dtype=np.float64)))
    return make_fir(out)


def _is_int(v) -> bool:
    return isinstance(v, int) and not isinstance(v, bool)


def load_device_json(path: str, sample_rate: int) -> Tuple[FsmDesc, dict]:
    """Device JSON -> flat tables, following the reference loader.

    State slots are handed out the way get_or_reserve_state does
    (src/state_machine.c:208-247): a name matching "reset" case-insensitively
    takes slot 0 while slot 0 is free, anything else takes the first free
    slot in order of first mention (as a state or as a trigger target)."""
    with open(path) as f:
        dev = json.load(f)["device"]
    num_bits = dev["num_bits"]
    states = dev["states"]
    n = len(states)
    names: List[Optional[str]] = [None] * n

    def slot(name: str) -> int:
        if name.lower() == "reset" and names[0] is None:
            names[0] = name
            return 0
        for i in range(n):
            if names[i] is None:
                names[i] = name
                return i
            if names[i] == name:
                return i
        raise ValueError("no room left to add state %r" % name)

    sdur = [0] * n
    sto = [0] * n
    trigs: List[list] = [[] for _ in range(n)]
    for st in states:
        idx = slot(st["name"])
        to = st.get("timeout_us")
        sto[idx] = to if _is_int(to) and to >= 0 else 0       # device.c:94-105
        du = st.get("duration_us")
        sdur[idx] = du if _is_int(du) and du >= 0 else 0      # device.c:107-115
        tl = []
        for tr in st["triggers"]:
            cond = COND[tr["condition"].lower()]
            d = tr.get("duration_us")
            d = d if _is_int(d) else 0                        # device.c:157-162
            act = tr.get("action")
            act = ACTION[act.lower()] if isinstance(act, str) else ACTION["none"]
            tl.append((cond, d, slot(tr["state"]), act))
        trigs[idx] = tl
    tbeg = [0]
    for tl in trigs:
        tbeg.append(tbeg[-1] + len(tl))
    flat = [t for tl in trigs for t in tl]
    desc = FsmDesc(
        state_names=[nm if nm is not None else "" for nm in names],
        max_bits=int(num_bits), sample_rate=int(sample_rate),
        state_duration_us=np.array(sdur, dtype=np.uint64),
        state_timeout_us=np.array(sto, dtype=np.uint64),
        trig_begin=np.array(tbeg, dtype=np.uint32),
        trig_cond=np.array([t[0] for t in flat], dtype=np.uint8),
        trig_action=np.array([t[3] for t in flat], dtype=np.uint8),
        trig_next=np.array([t[2] for t in flat], dtype=np.uint32),
        trig_duration_us=np.array([t[1] for t in flat], dtype=np.uint64),
    )
    return desc, dev


# --------------------------------------------------------------------------
# library handles
# --------------------------------------------------------------------------

_lib = None
_ref = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libook_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.ook_unpack.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ook_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ook_threshold.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_size_t]
        L.ook_fir_new.restype = C.c_void_p
        L.ook_fir_new.argtypes = [C.POINTER(_FirDescC)]
        L.ook_fir_free.argtypes = [C.c_void_p]
        L.ook_fir_reset.argtypes = [C.c_void_p]
        L.ook_fir_run.restype = C.c_size_t
        L.ook_fir_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.ook_sm_new.restype = C.c_void_p
        L.ook_sm_new.argtypes = [C.POINTER(_FsmDescC)]
        L.ook_sm_free.argtypes = [C.c_void_p]
        L.ook_sm_process.restype = C.c_int
        L.ook_sm_process.argtypes = [C.c_void_p, C.c_void_p, C.c_uint,
                                     C.POINTER(C.c_uint)]
        L.ook_sm_data.restype = C.POINTER(C.c_uint8)
        L.ook_sm_data.argtypes = [C.c_void_p]
        L.ook_sm_peek.argtypes = [C.c_void_p, C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_double),
                                  C.POINTER(C.c_int)]
        L.ook_oracle_rx.restype = C.c_uint64
        L.ook_oracle_rx.argtypes = [
            C.c_void_p, C.c_uint64, C.POINTER(_FirDescC), C.c_float,
            C.POINTER(_FsmDescC), C.c_uint32,
            C.POINTER(_MsgC), C.c_uint64, C.POINTER(C.c_uint64),
            C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
            C.c_void_p, C.c_void_p]
        L.ook_duration_window.restype = C.c_int
        L.ook_duration_window.argtypes = [C.c_uint32, C.c_uint64,
                                          C.POINTER(C.c_uint64),
                                          C.POINTER(C.c_uint64)]
        L.ook_timeout_count.restype = C.c_int
        L.ook_timeout_count.argtypes = [C.c_uint32, C.c_uint64,
                                        C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def have_ref() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "libookref.so"))


def ref() -> C.CDLL:
    """The reference's own code (state_machine.c, complexf.h) via our driver."""
    global _ref
    if _ref is None:
        R = C.CDLL(os.path.join(_HERE, "_ref", "libookref.so"))
        R.ref_sm_new.restype = C.c_void_p
        R.ref_sm_new.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32] + [C.c_void_p] * 7
        R.ref_sm_free.argtypes = [C.c_void_p]
        R.ref_sm_process.restype = C.c_int
        R.ref_sm_process.argtypes = [C.c_void_p, C.c_void_p, C.c_uint,
                                     C.POINTER(C.c_uint)]
        R.ref_sm_data.restype = C.POINTER(C.c_uint8)
        R.ref_sm_data.argtypes = [C.c_void_p]
        R.ref_device_stream.restype = C.c_uint64
        R.ref_device_stream.argtypes = [
            C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
            C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64,
            C.POINTER(C.c_uint64)]
        R.ref_sm_generate.restype = C.POINTER(C.c_float)
        R.ref_sm_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_uint,
                                      C.c_float, C.POINTER(C.c_uint)]
        R.ref_free.argtypes = [C.c_void_p]
        R.ref_unpack.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        R.ref_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        R.ref_threshold.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_uint]
        R.ref_fmt_new.restype = C.c_void_p
        R.ref_fmt_new.argtypes = [C.c_uint, C.c_uint]
        R.ref_fmt_free.argtypes = [C.c_void_p]
        R.ref_fmt_add_field.argtypes = [C.c_void_p, C.c_char_p, C.c_uint, C.c_uint, C.c_int, C.c_size_t,
                                        C.c_int, C.c_float, C.c_float]
        R.ref_fmt_add_enum.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint64]
        R.ref_fmt_set_default.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        R.ref_fmt_initialized.argtypes = [C.c_void_p]
        R.ref_fmt_format.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
        R.ref_fmt_default_data.argtypes = [C.c_void_p, C.c_void_p]
        R.ref_fmt_set.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p]
        _ref = R
    return _ref


# --------------------------------------------------------------------------
# oracle wrappers
# --------------------------------------------------------------------------

@dataclass
class RxResult:
    decimated: int
    msg_samples: np.ndarray             # uint64
    payloads: np.ndarray                # uint8 [n, payload_bytes]
    err_samples: np.ndarray             # uint64
    bits: Optional[np.ndarray] = None   # uint8 per decimated sample
    fir: Optional[np.ndarray] = None    # float32 [n, 2]

    def payload_bits(self, i: int, nbits: int) -> str:
        """Payload i as a string of bits, first received bit first."""
        b = np.unpackbits(self.payloads[i], bitorder="little")[:nbits]
        return "".join(str(int(x)) for x in b)


def unpack(iq: np.ndarray) -> np.ndarray:
    iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1)
    n = iq.size // 2
    out = np.empty((n, 2), dtype=np.float32)
    lib().ook_unpack(iq.ctypes.data, out.ctypes.data, n)
    return out


def pack(x: np.ndarray) -> np.ndarray:
    """complexf_to_sc16q11 (complexf.h:87-96)."""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 2)
    out = np.empty(2 * x.shape[0], dtype=np.int16)
    lib().ook_pack(x.ctypes.data, out.ctypes.data, x.shape[0])
    return out


def dig_text(bits: np.ndarray, buffer_len: int = 8192) -> str:
    """record_dig (ookiedokie.c:146-169) as written: per buffer, per sample,
    with its `prev` / `sample_no` carried across buffers.  Small inputs only."""
    bits = np.asarray(bits, dtype=np.uint8)
    out = []
    sample_no = 0
    prev = 0
    for start in range(0, bits.size, buffer_len):
        buf = bits[start:start + buffer_len]
        if sample_no == 0:                                   # :150-153
            prev = int(buf[0])
            out.append("0, %c\n" % ("1" if buf[0] else "0"))
        idx = np.nonzero(np.diff(np.concatenate(([prev], buf))))[0]
        for i in idx:                                        # :155-166
            cur = int(buf[i])
            out.append("%d, %c\n%d, %c\n" % (sample_no + i - 1, "1" if prev else "0",
                                              sample_no + i, "1" if cur else "0"))
            prev = cur
        sample_no += buf.size
    return "".join(out)


def threshold(x: np.ndarray, thr: float) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 2)
    out = np.empty(x.shape[0], dtype=np.uint8)
    lib().ook_threshold(x.ctypes.data, np.float32(thr), out.ctypes.data, x.shape[0])
    return out


def fir_run(fir: FirDesc, x: np.ndarray, chunk: Optional[int] = None) -> np.ndarray:
    """Streaming FIR over complex float input [n,2]; optional chunking."""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 2)
    cs = fir.c_struct()
    h = lib().ook_fir_new(C.byref(cs))
    if not h:
        raise ValueError("bad filter")
    n = x.shape[0]
    out = np.empty((n // 1 + 2, 2), dtype=np.float32)
    produced = 0
    step = chunk or max(n, 1)
    for off in range(0, n, step):
        seg = x[off:off + step]
        produced += lib().ook_fir_run(h, seg.ctypes.data, seg.shape[0],
                                      out[produced:].ctypes.data)
    lib().ook_fir_free(h)
    return out[:produced].copy()


def rx(iq: np.ndarray, fir: Optional[FirDesc], thr: float,
       fsm: Optional[FsmDesc], spb: int = 8192, want_bits: bool = False,
       want_fir: bool = False, msg_cap: int = 1 << 16,
       err_cap: int = 1 << 20) -> RxResult:
    """Whole reference rx path over an in-memory SC16Q11 capture."""
    iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1)
    n = iq.size // 2
    nbuf = (n + spb - 1) // spb
    dec = fir.total_decimation if fir is not None else 1
    max_dec = nbuf * spb // dec + nbuf + 1
    bits = np.zeros(max_dec, dtype=np.uint8) if want_bits else None
    firo = np.zeros((max_dec, 2), dtype=np.float32) if want_fir else None
    msgs = (_MsgC * msg_cap)()
    errs = np.zeros(err_cap, dtype=np.uint64)
    nm = C.c_uint64(0)
    ne = C.c_uint64(0)
    fs = fir.c_struct() if fir is not None else None
    ss = fsm.c_struct() if fsm is not None else None
    total = lib().ook_oracle_rx(
        iq.ctypes.data, n, C.byref(fs) if fs is not None else None,
        np.float32(thr), C.byref(ss) if ss is not None else None, spb,
        msgs, msg_cap, C.byref(nm), errs.ctypes.data, err_cap, C.byref(ne),
        bits.ctypes.data if bits is not None else None,
        firo.ctypes.data if firo is not None else None)
    k = min(nm.value, msg_cap)
    pb = fsm.payload_bytes if fsm is not None else 0
    samples = np.array([msgs[i].sample for i in range(k)], dtype=np.uint64)
    pay = np.zeros((k, pb), dtype=np.uint8)
    for i in range(k):
        pay[i] = np.frombuffer(bytes(msgs[i].payload), dtype=np.uint8)[:pb]
    return RxResult(int(total), samples, pay,
                    errs[:min(ne.value, err_cap)].copy(),
                    bits[:total] if bits is not None else None,
                    firo[:total] if firo is not None else None)


def sm_stream(fsm: FsmDesc, bits: np.ndarray, buf_len: int):
    """device_process loop (oracle restatement) over a raw 0/1 stream."""
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    cs = fsm.c_struct()
    h = lib().ook_sm_new(C.byref(cs))
    msgs, pays, errs = [], [], []
    nproc = C.c_uint(0)
    pb = fsm.payload_bytes
    for base in range(0, bits.size, buf_len):
        count = min(buf_len, bits.size - base)
        total, r = 0, 0
        while total < count and r != -1:
            r = lib().ook_sm_process(h, bits[base + total:].ctypes.data,
                                     count - total, C.byref(nproc))
            total += nproc.value
            if r == 1:
                msgs.append(base + total - 1)
                d = lib().ook_sm_data(h)
                pays.append(bytes(d[i] for i in range(pb)))
            elif r == -1:
                errs.append(base + total - 1)
    lib().ook_sm_free(h)
    return (np.array(msgs, dtype=np.uint64),
            np.frombuffer(b"".join(pays), dtype=np.uint8).reshape(len(pays), pb)
            if pays else np.zeros((0, pb), dtype=np.uint8),
            np.array(errs, dtype=np.uint64))


def duration_window(rate: int, dur: int) -> Tuple[int, int]:
    a, b = C.c_uint64(0), C.c_uint64(0)
    if lib().ook_duration_window(rate, dur, C.byref(a), C.byref(b)) != 0:
        raise RuntimeError("replay limit")
    return a.value, b.value


def timeout_count(rate: int, t: int) -> int:
    a = C.c_uint64(0)
    if lib().ook_timeout_count(rate, t, C.byref(a)) != 0:
        raise RuntimeError("replay limit")
    return a.value


# --------------------------------------------------------------------------
# reference wrappers (oracle/_ref)
# --------------------------------------------------------------------------

class RefSm:
    """The reference's real state machine built through its public API."""

    def __init__(self, fsm: FsmDesc):
        self.fsm = fsm
        self.h = ref().ref_sm_new(
            fsm.num_states, fsm.max_bits, fsm.sample_rate,
            fsm.state_duration_us.ctypes.data, fsm.state_timeout_us.ctypes.data,
            fsm.trig_begin.ctypes.data, fsm.trig_cond.ctypes.data,
            fsm.trig_action.ctypes.data, fsm.trig_next.ctypes.data,
            fsm.trig_duration_us.ctypes.data)
        if not self.h:
            raise RuntimeError("reference sm_init failed")

    def close(self):
        if self.h:
            ref().ref_sm_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def stream(self, bits: np.ndarray, buf_len: int, cap: int = 1 << 16):
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        pb = self.fsm.payload_bytes
        ms = np.zeros(cap, dtype=np.uint64)
        pay = np.zeros((cap, pb), dtype=np.uint8)
        es = np.zeros(1 << 20, dtype=np.uint64)
        ne = C.c_uint64(0)
        n = ref().ref_device_stream(self.h, bits.ctypes.data, bits.size,
                                    buf_len, ms.ctypes.data, pay.ctypes.data,
                                    pb, cap, es.ctypes.data, es.size,
                                    C.byref(ne))
        n = min(n, cap)
        return ms[:n].copy(), pay[:n].copy(), es[:min(ne.value, es.size)].copy()

    def generate(self, payload: bytes, on_val: float = 0.95) -> np.ndarray:
        """sm_generate -> float32 [n,2]; use a fresh RefSm per call
        (the reference never rewinds num_bits)."""
        n = C.c_uint(0)
        buf = (C.c_uint8 * 80)(*payload)
        p = ref().ref_sm_generate(self.h, buf, self.fsm.max_bits,
                                  np.float32(on_val), C.byref(n))
        if not p:
            raise RuntimeError("sm_generate failed")
        out = np.ctypeslib.as_array(p, shape=(n.value, 2)).copy()
        ref().ref_free(p)
        return out


def ref_unpack(iq: np.ndarray) -> np.ndarray:
    iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1)
    out = np.empty((iq.size // 2, 2), dtype=np.float32)
    ref().ref_unpack(iq.ctypes.data, out.ctypes.data, iq.size // 2)
    return out


def ref_pack(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 2)
    out = np.empty(x.shape[0] * 2, dtype=np.int16)
    ref().ref_pack(x.ctypes.data, out.ctypes.data, x.shape[0])
    return out


def ref_threshold(x: np.ndarray, thr: float) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 2)
    out = np.empty(x.shape[0], dtype=np.uint8)
    ref().ref_threshold(x.ctypes.data, np.float32(thr), out.ctypes.data, x.shape[0])
    return out


FMT_CODES = {"hex": 1, "unsigned decimal": 2, "sign-magnitude": 3, "two's complement": 4,
             "float": 5, "enumeration": 6}       # formatter_fmt_value, formatter.c:859-876
ENDIAN_CODES = {"big": 1, "little": 2}           # formatter_endianess_value, formatter.c:848-857


class RefFormatter:
    """The reference's real formatter (formatter.c) fed the way device.c's
    add_field (:255-422) feeds it from a device file's "fields" entries
    (dicts with the JSON keys).  ts_mode is always "none"."""

    def __init__(self, fields, num_bits: int):
        self.h = ref().ref_fmt_new(len(fields), num_bits)
        if not self.h:
            raise RuntimeError("reference formatter_init failed")
        self.nbytes = (num_bits + 7) // 8
        for f in fields:
            fmt = FMT_CODES[f["format"].lower()]
            enums = f.get("enum_values", []) if fmt == 6 else []
            if ref().ref_fmt_add_field(self.h, f["name"].encode(), int(f["start_bit"]), int(f["end_bit"]),
                                       fmt, len(enums), ENDIAN_CODES[f["endianness"].lower()],
                                       float(f.get("scaling", 0.0)), float(f.get("offset", 0.0))) != 0:
                raise ValueError("reference formatter_add_field refused %r" % f["name"])
            for e in enums:
                if ref().ref_fmt_add_enum(self.h, f["name"].encode(), e["string"].encode(),
                                          int(e["value"], 0)) != 0:
                    raise ValueError("reference formatter_add_field_enum refused %r" % e["string"])
            if ref().ref_fmt_set_default(self.h, f["name"].encode(), f["default"].encode()) != 0:
                raise ValueError("reference formatter_set_field_default refused %r" % f["default"])
        if not ref().ref_fmt_initialized(self.h):
            raise ValueError("reference formatter_initialized says no")

    def data_to_keyval(self, payload):
        a = np.zeros(max(self.nbytes, 1) + 8, dtype=np.uint8)
        p = np.frombuffer(bytes(payload), dtype=np.uint8)
        a[:min(p.size, self.nbytes)] = p[:self.nbytes]
        buf = C.create_string_buffer(1 << 16)
        if ref().ref_fmt_format(self.h, a.ctypes.data, buf, len(buf)) != 0:
            raise RuntimeError("reference formatter_data_to_keyval failed")
        out = []
        for line in buf.value.decode("utf-8", "replace").split("\n"):
            if line:
                k, v = line.split("\t", 1)
                out.append((k, v))
        return out

    def default_data(self) -> np.ndarray:
        a = np.zeros(self.nbytes + 8, dtype=np.uint8)
        ref().ref_fmt_default_data(self.h, a.ctypes.data)
        return a[:self.nbytes].copy()

    def keyval_to_data(self, params, data=None) -> np.ndarray:
        a = np.zeros(self.nbytes + 8, dtype=np.uint8)
        base = self.default_data() if data is None else np.asarray(data, dtype=np.uint8)
        a[:base.size] = base
        for k, v in params:
            if ref().ref_fmt_set(self.h, k.encode(), v.encode(), a.ctypes.data) != 0:
                raise ValueError("reference formatter_keyval_to_data refused %s=%s" % (k, v))
        return a[:self.nbytes].copy()

    def close(self):
        if self.h:
            ref().ref_fmt_free(self.h)
            self.h = None

    def __del__(self):
        self.close()
