"""ookiedokie_amd -- MI355X-native OOK receive / demodulation path.

Thin Python mirror of the reference's operator interface for the rx hot
path, bound over the C ABI of ``libookiedokie_amd.so`` with ctypes
(``include/ookiedokie_amd.h``).  Names follow the reference:

================================  ==========================================
here                              reference (OOKiedokie ``src/``)
================================  ==========================================
``Filter.load`` / ``fir_init``    ``fir_init`` (fir.h:43-55)
``Device.load`` / ``device_init`` ``device_init`` (device.h:47-55)
``Receiver.rx``                   loop body of ``ookiedokie_rx``
                                  (ookiedokie.c:243-288)
``StreamFir.filter_and_decimate`` ``fir_filter_and_decimate`` (fir.h:68-81)
``HipFileBackend``                ``sdr_<name>_{init,deinit,rx,tx,flush}``
                                  (sdr/supported_devices.h:32-48)
================================  ==========================================

There is no CPU fallback: the library is hand-written HIP for gfx950 and
every compute call fails loudly without a GPU (``OokdError``).  PyTorch is
only used by callers for device memory / streams / ``torch.distributed``.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libookiedokie_amd.so")
HEADER_PATH = os.path.join(_HERE, "..", "include", "ookiedokie_amd.h")

MAX_PAYLOAD_BYTES = 32
FILE_EOF = -(2 ** 31)                   # SDR_FILE_EOF, sdr.h:36
RX_EXACT_FIR = 1
RX_KEEP_FIR = 2
RX_FSM_ROUNDS = 4
RX_NO_QUIET_SKIP = 8
RX_COUNT_QUIET = 16
RX_SCAN_SIMS = 32
RX_FRONT_GRID = 64
RX_NO_PIPELINE = 128
RX_FIR_VALU = 256
RX_SCAN_TABLES = 512
DEFAULT_THRESHOLD = 0.1                 # ookiedokie_cfg.c:27
DEFAULT_RATE = 3000000                  # ookiedokie_cfg.c:32
DEFAULT_SAMPLES_PER_BUF = 8192          # ookiedokie_cfg.c:34


class OokdError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__("ookiedokie_amd error %d: %s" % (code, text))
        self.code = code


# --------------------------------------------------------------------------
# C structures
# --------------------------------------------------------------------------

class FsmTables(C.Structure):
    _fields_ = [
        ("num_states", C.c_uint32), ("max_bits", C.c_uint32),
        ("sample_rate", C.c_uint32), ("num_triggers", C.c_uint32),
        ("state_duration_us", C.POINTER(C.c_uint64)),
        ("state_timeout_us", C.POINTER(C.c_uint64)),
        ("trig_begin", C.POINTER(C.c_uint32)),
        ("trig_cond", C.POINTER(C.c_uint8)),
        ("trig_action", C.POINTER(C.c_uint8)),
        ("trig_next", C.POINTER(C.c_uint32)),
        ("trig_duration_us", C.POINTER(C.c_uint64)),
        ("state_kmin", C.POINTER(C.c_uint64)), ("state_kmax", C.POINTER(C.c_uint64)),
        ("state_kto", C.POINTER(C.c_uint64)),
        ("trig_kmin", C.POINTER(C.c_uint64)), ("trig_kmax", C.POINTER(C.c_uint64)),
    ]


class RxConfig(C.Structure):
    _fields_ = [
        ("hip_device", C.c_int32), ("flags", C.c_uint32), ("threshold", C.c_float),
        ("samples_per_buffer", C.c_uint32), ("max_samples", C.c_uint64),
        ("max_captures", C.c_uint32), ("edge_capacity", C.c_uint64),
        ("segment_buffers", C.c_uint32), ("message_slots", C.c_uint32),
        ("message_capacity", C.c_uint64), ("stream", C.c_void_p),
        ("pipeline_chunk_samples", C.c_uint64), ("front_gate", C.c_void_p),
    ]


class Message(C.Structure):
    _fields_ = [("capture", C.c_uint32), ("reserved", C.c_uint32),
                ("sample", C.c_uint64), ("payload", C.c_uint8 * MAX_PAYLOAD_BYTES)]


class FsmState(C.Structure):
    _fields_ = [("state", C.c_uint32), ("num_bits", C.c_uint32), ("k", C.c_uint64),
                ("prev_bit", C.c_uint32), ("reserved", C.c_uint32),
                ("payload", C.c_uint8 * (MAX_PAYLOAD_BYTES + 8))]

    def key(self) -> bytes:
        return bytes(self)


class RxStats(C.Structure):
    _fields_ = [
        ("input_samples", C.c_uint64), ("decimated_samples", C.c_uint64),
        ("num_edges", C.c_uint64), ("num_messages", C.c_uint64),
        ("num_errors", C.c_uint64), ("guard_recomputes", C.c_uint64),
        ("fsm_iterations", C.c_uint32), ("num_segments", C.c_uint32),
        ("fsm_path", C.c_uint32), ("fsm_fallback_reason", C.c_uint32),
        ("fir_kernel_ms", C.c_float), ("total_device_ms", C.c_float),
        ("quiet_waves", C.c_uint64), ("total_waves", C.c_uint64),
        ("pipeline_chunks", C.c_uint32), ("front_launches", C.c_uint32),
        ("scan_entry_form", C.c_uint32), ("reserved", C.c_uint32),
    ]


class SynthConfig(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("sample_rate", C.c_uint32), ("amplitude", C.c_uint32),
        ("noise", C.c_uint32), ("gap_min_us", C.c_uint32), ("gap_max_us", C.c_uint32),
        ("glitch_every", C.c_uint32), ("random_phase", C.c_uint32),
    ]


class HostCfg(C.Structure):
    """struct ookiedokie_cfg (ookiedokie_cfg.h:50-91), field for field."""
    _fields_ = [
        ("sdr_type", C.c_char_p), ("direction", C.c_int), ("sdr_args", C.c_char_p),
        ("frequency", C.c_uint), ("bandwidth", C.c_uint), ("samplerate", C.c_uint),
        ("gain", C.c_int), ("device", C.c_char_p), ("tx_count", C.c_uint),
        ("tx_delay_us", C.c_uint), ("device_params", C.c_void_p), ("rx_fmt", C.c_int),
        ("rx_threshold", C.c_float), ("rx_rec_filename", C.c_char_p),
        ("rx_rec_type", C.c_char_p), ("rx_filter", C.c_char_p),
        ("rx_rec_dig", C.c_char_p), ("rx_rec_input", C.c_ubyte),
        ("samples_per_buffer", C.c_uint), ("num_buffers", C.c_uint),
        ("num_transfers", C.c_uint), ("stream_timeout_ms", C.c_uint),
        ("sync_timeout_ms", C.c_uint), ("verbosity", C.c_int),
    ]


# --------------------------------------------------------------------------
# library
# --------------------------------------------------------------------------

_lib: Optional[C.CDLL] = None

_PROTOTYPES = {
    "ookd_last_error": (C.c_char_p, []),
    "ookd_api_version": (C.c_int, []),
    "ookd_filter_load": (C.c_void_p, [C.c_char_p]),
    "ookd_filter_create": (C.c_void_p, [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ookd_filter_free": (None, [C.c_void_p]),
    "ookd_filter_total_decimation": (C.c_uint32, [C.c_void_p]),
    "ookd_filter_num_stages": (C.c_uint32, [C.c_void_p]),
    "ookd_filter_stage": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_float))]),
    "ookd_device_load": (C.c_void_p, [C.c_char_p, C.c_uint32]),
    "ookd_device_create": (C.c_void_p, [C.POINTER(FsmTables)]),
    "ookd_device_free": (None, [C.c_void_p]),
    "ookd_device_num_bits": (C.c_uint32, [C.c_void_p]),
    "ookd_device_name": (C.c_char_p, [C.c_void_p]),
    "ookd_device_state_name": (C.c_char_p, [C.c_void_p, C.c_uint32]),
    "ookd_device_tables": (C.c_int, [C.c_void_p, C.POINTER(FsmTables)]),
    "ookd_rx_create": (C.c_void_p, [C.POINTER(RxConfig), C.c_void_p, C.c_void_p]),
    "ookd_rx_destroy": (None, [C.c_void_p]),
    "ookd_rx_process_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64]),
    "ookd_rx_submit_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64]),
    "ookd_scan_domain_info": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "ookd_rx_wait": (C.c_int, [C.c_void_p]),
    "ookd_rx_process_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "ookd_rx_shard_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                      C.c_int, C.POINTER(FsmState), C.POINTER(FsmState)]),
    "ookd_rx_shard_refine": (C.c_int, [C.c_void_p, C.POINTER(FsmState), C.POINTER(FsmState)]),
    "ookd_rx_halo_samples": (C.c_uint64, [C.c_void_p]),
    "ookd_rx_num_messages": (C.c_uint64, [C.c_void_p]),
    "ookd_rx_messages": (C.POINTER(Message), [C.c_void_p]),
    "ookd_rx_get_stats": (C.c_int, [C.c_void_p, C.POINTER(RxStats)]),
    "ookd_rx_bit_words": (C.c_uint64, [C.c_void_p]),
    "ookd_rx_get_bits": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]),
    "ookd_rx_get_edges": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64,
                                    C.POINTER(C.c_uint64)]),
    "ookd_rx_get_fir": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]),
    "ookd_rx_get_errors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "ookd_fir_create": (C.c_void_p, [C.c_int32, C.c_void_p, C.c_size_t, C.c_uint32]),
    "ookd_fir_reset": (None, [C.c_void_p]),
    "ookd_fir_destroy": (None, [C.c_void_p]),
    "ookd_fir_filter_and_decimate": (C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ookd_synth_create": (C.c_void_p, [C.c_void_p, C.POINTER(SynthConfig), C.c_uint64]),
    "ookd_synth_free": (None, [C.c_void_p]),
    "ookd_synth_num_messages": (C.c_uint64, [C.c_void_p]),
    "ookd_synth_message": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p]),
    "ookd_synth_fill_host": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "ookd_synth_fill_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64,
                                         C.c_void_p, C.c_void_p]),
    "ookd_rx_dig_text": (C.c_size_t, [C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t]),
    "ookd_rx_record_dig": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p]),
    "ookd_rx_get_fir_sc16q11": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64]),
    "ookd_rx_record_fir": (C.c_int, [C.c_void_p, C.c_uint32, C.c_char_p]),
    "ookd_formatter_create": (C.c_void_p, [C.c_void_p]),
    "ookd_formatter_free": (None, [C.c_void_p]),
    "ookd_formatter_num_fields": (C.c_uint32, [C.c_void_p]),
    "ookd_formatter_field_name": (C.c_char_p, [C.c_void_p, C.c_uint32]),
    "ookd_formatter_ts_mode": (C.c_int, [C.c_void_p]),
    "ookd_formatter_field_to_str": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_char_p, C.c_size_t]),
    "ookd_formatter_default_data": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ookd_formatter_set_field": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_size_t]),
    "ookd_print_record": (C.c_size_t, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_size_t,
                                       C.c_char_p, C.c_size_t]),
    "ookd_print_messages": (C.c_size_t, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_uint64,
                                         C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]),
    "ookd_rx_gate_create": (C.c_void_p, []),
    "ookd_rx_gate_destroy": (None, [C.c_void_p]),
    "sdr_hip_file_init": (C.c_void_p, [C.c_void_p]),
    "sdr_hip_file_deinit": (None, [C.c_void_p]),
    "sdr_hip_file_rx": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint]),
    "sdr_hip_file_tx": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint]),
    "sdr_hip_file_flush": (C.c_int, [C.c_void_p]),
    "sdr_hip_file_capture": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]),
}


def lib() -> C.CDLL:
    """Loads libookiedokie_amd.so.  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OokdError(-4, "%s is missing: run `python -m ookiedokie_amd.build` "
                                "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error() -> str:
    return (lib().ookd_last_error() or b"").decode("utf-8", "replace")


def _check(rc: int) -> None:
    if rc != 0:
        raise OokdError(rc, last_error())


# --------------------------------------------------------------------------
# filter / device
# --------------------------------------------------------------------------

class Filter:
    """Multi-stage decimating FIR description (struct fir_filter, fir.c:60-66)."""

    def __init__(self, handle: int):
        if not handle:
            raise OokdError(-3, last_error())
        self._h = handle

    @classmethod
    def load(cls, path: str) -> "Filter":
        return cls(lib().ookd_filter_load(os.fsencode(path)))

    @classmethod
    def from_stages(cls, stages: Sequence[Tuple[int, Sequence[float]]]) -> "Filter":
        dec = np.array([d for d, _ in stages], dtype=np.uint32)
        nt = np.array([len(t) for _, t in stages], dtype=np.uint32)
        taps = np.concatenate([np.asarray(t, dtype=np.float32) for _, t in stages]) \
            if stages else np.zeros(0, np.float32)
        taps = np.ascontiguousarray(taps, dtype=np.float32)
        return cls(lib().ookd_filter_create(len(stages), dec.ctypes.data, nt.ctypes.data,
                                            taps.ctypes.data))

    @property
    def total_decimation(self) -> int:
        return int(lib().ookd_filter_total_decimation(self._h))

    @property
    def num_stages(self) -> int:
        return int(lib().ookd_filter_num_stages(self._h))

    def stage(self, s: int) -> Tuple[int, np.ndarray]:
        d, n = C.c_uint32(0), C.c_uint32(0)
        p = C.POINTER(C.c_float)()
        _check(lib().ookd_filter_stage(self._h, s, C.byref(d), C.byref(n), C.byref(p)))
        return int(d.value), np.ctypeslib.as_array(p, shape=(n.value,)).copy()

    def close(self) -> None:
        if self._h:
            lib().ookd_filter_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fir_init(path: str) -> Filter:
    return Filter.load(path)


class Device:
    """OOK device description: state machine tables + field layout (device.c:60-74)."""

    def __init__(self, handle: int):
        if not handle:
            raise OokdError(-3, last_error())
        self._h = handle

    @classmethod
    def load(cls, path: str, sample_rate: int) -> "Device":
        return cls(lib().ookd_device_load(os.fsencode(path), int(sample_rate)))

    @classmethod
    def from_tables(cls, *, max_bits: int, sample_rate: int, state_duration_us, state_timeout_us,
                    trig_begin, trig_cond, trig_action, trig_next, trig_duration_us) -> "Device":
        keep = [np.ascontiguousarray(state_duration_us, dtype=np.uint64),
                np.ascontiguousarray(state_timeout_us, dtype=np.uint64),
                np.ascontiguousarray(trig_begin, dtype=np.uint32),
                np.ascontiguousarray(trig_cond, dtype=np.uint8),
                np.ascontiguousarray(trig_action, dtype=np.uint8),
                np.ascontiguousarray(trig_next, dtype=np.uint32),
                np.ascontiguousarray(trig_duration_us, dtype=np.uint64)]
        t = FsmTables()
        t.num_states = len(keep[0])
        t.max_bits = max_bits
        t.sample_rate = sample_rate
        t.num_triggers = len(keep[3])
        t.state_duration_us = keep[0].ctypes.data_as(C.POINTER(C.c_uint64))
        t.state_timeout_us = keep[1].ctypes.data_as(C.POINTER(C.c_uint64))
        t.trig_begin = keep[2].ctypes.data_as(C.POINTER(C.c_uint32))
        t.trig_cond = keep[3].ctypes.data_as(C.POINTER(C.c_uint8))
        t.trig_action = keep[4].ctypes.data_as(C.POINTER(C.c_uint8))
        t.trig_next = keep[5].ctypes.data_as(C.POINTER(C.c_uint32))
        t.trig_duration_us = keep[6].ctypes.data_as(C.POINTER(C.c_uint64))
        return cls(lib().ookd_device_create(C.byref(t)))

    @property
    def num_bits(self) -> int:
        return int(lib().ookd_device_num_bits(self._h))

    @property
    def payload_bytes(self) -> int:
        return (self.num_bits + 7) // 8

    @property
    def name(self) -> str:
        return lib().ookd_device_name(self._h).decode()

    def tables(self) -> dict:
        t = FsmTables()
        _check(lib().ookd_device_tables(self._h, C.byref(t)))
        ns, nt = t.num_states, t.num_triggers

        def arr(p, n):
            return np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0)

        out = {"num_states": ns, "max_bits": t.max_bits, "sample_rate": t.sample_rate,
               "num_triggers": nt,
               "state_names": [lib().ookd_device_state_name(self._h, s).decode() for s in range(ns)]}
        for name in ("state_duration_us", "state_timeout_us", "state_kmin", "state_kmax", "state_kto"):
            out[name] = arr(getattr(t, name), ns)
        out["trig_begin"] = arr(t.trig_begin, ns + 1)
        for name in ("trig_cond", "trig_action", "trig_next", "trig_duration_us", "trig_kmin", "trig_kmax"):
            out[name] = arr(getattr(t, name), nt)
        return out

    def close(self) -> None:
        if self._h:
            lib().ookd_device_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_init(path: str, sample_rate: int) -> Device:
    return Device.load(path, sample_rate)


# --------------------------------------------------------------------------
# formatter / printer (host side of a decoded message)
# --------------------------------------------------------------------------

RX_FMT_PRETTY = 0       # enum ookiedokie_rx_fmt, ookiedokie_cfg.h:41-45
RX_FMT_CSV = 1


class Formatter:
    """struct formatter (src/formatter.h) built from a device's "fields":
    payload bits -> per-field text -> the text rx_print writes."""

    def __init__(self, device: Device):
        h = lib().ookd_formatter_create(device._h)
        if not h:
            raise OokdError(-1, last_error())
        self._h = h
        self._payload_bytes = max(device.payload_bytes, 1)
        self.first_print = True         # CSV heading still to be printed

    @property
    def field_names(self) -> List[str]:
        n = lib().ookd_formatter_num_fields(self._h)
        return [lib().ookd_formatter_field_name(self._h, i).decode() for i in range(n)]

    @property
    def ts_mode(self) -> int:
        return int(lib().ookd_formatter_ts_mode(self._h))

    def _payload(self, payload) -> np.ndarray:
        a = np.zeros(MAX_PAYLOAD_BYTES, dtype=np.uint8)
        p = np.frombuffer(bytes(payload), dtype=np.uint8) if not isinstance(payload, np.ndarray) else payload
        a[:min(p.size, a.size)] = p[:a.size]
        return a

    def data_to_keyval(self, payload) -> List[Tuple[str, str]]:
        """formatter_data_to_keyval (formatter.c:715-739), without the timestamp pair."""
        a = self._payload(payload)
        out = []
        buf = C.create_string_buffer(96)
        for i, name in enumerate(self.field_names):
            _check(lib().ookd_formatter_field_to_str(self._h, i, a.ctypes.data, buf, len(buf)))
            out.append((name, buf.value.decode("utf-8", "replace")))
        return out

    def default_data(self) -> np.ndarray:
        """formatter_default_data (formatter.c:835-846)."""
        a = np.zeros(MAX_PAYLOAD_BYTES, dtype=np.uint8)
        _check(lib().ookd_formatter_default_data(self._h, a.ctypes.data, a.size))
        return a[:self._payload_bytes].copy()

    def keyval_to_data(self, params: Sequence[Tuple[str, str]], data: Optional[np.ndarray] = None) -> np.ndarray:
        """formatter_keyval_to_data (formatter.c:793-832) on top of `data`
        (default: the defaults, as device_generate does, device.c:660-676)."""
        a = np.zeros(MAX_PAYLOAD_BYTES, dtype=np.uint8)
        base = self.default_data() if data is None else np.asarray(data, dtype=np.uint8)
        a[:base.size] = base
        for k, v in params:
            _check(lib().ookd_formatter_set_field(self._h, k.encode(), v.encode(), a.ctypes.data, a.size))
        return a[:self._payload_bytes].copy()

    def print_record(self, payloads: Sequence, fmt: int = RX_FMT_PRETTY) -> str:
        """rx_print for the messages of ONE buffer."""
        arrs = [self._payload(p) for p in payloads]
        ptrs = (C.c_void_p * max(len(arrs), 1))(*[a.ctypes.data for a in arrs])
        fp = C.c_int(1 if self.first_print else 0)
        need = lib().ookd_print_record(self._h, fmt, C.byref(fp), ptrs, len(arrs), None, 0)
        fp = C.c_int(1 if self.first_print else 0)
        buf = C.create_string_buffer(need + 1)
        lib().ookd_print_record(self._h, fmt, C.byref(fp), ptrs, len(arrs), buf, len(buf))
        self.first_print = bool(fp.value)
        return buf.value.decode("utf-8", "replace")

    def print_messages(self, result: "RxResult", samples_per_buffer: int, total_decimation: int = 1,
                       fmt: int = RX_FMT_PRETTY) -> str:
        """Everything the reference's rx loop prints for a run's messages."""
        n = len(result.msg_samples)
        msgs = (Message * max(n, 1))()
        for i in range(n):
            msgs[i].capture = int(result.captures[i])
            msgs[i].sample = int(result.msg_samples[i])
            p = self._payload(result.payloads[i])
            C.memmove(msgs[i].payload, p.ctypes.data, MAX_PAYLOAD_BYTES)
        fp = C.c_int(1 if self.first_print else 0)
        need = lib().ookd_print_messages(self._h, fmt, C.byref(fp), msgs, n, samples_per_buffer,
                                         total_decimation, None, 0)
        fp = C.c_int(1 if self.first_print else 0)
        buf = C.create_string_buffer(need + 1)
        lib().ookd_print_messages(self._h, fmt, C.byref(fp), msgs, n, samples_per_buffer, total_decimation,
                                  buf, len(buf))
        self.first_print = bool(fp.value)
        return buf.value.decode("utf-8", "replace")

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().ookd_formatter_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# --------------------------------------------------------------------------
# rx
# --------------------------------------------------------------------------

@dataclass
class RxResult:
    captures: np.ndarray        # uint32 [n]
    msg_samples: np.ndarray     # uint64 [n] decimated index of OUTPUT_READY
    payloads: np.ndarray        # uint8 [n, payload_bytes]
    stats: dict

    def payload_bits(self, i: int, nbits: int) -> str:
        b = np.unpackbits(self.payloads[i], bitorder="little")[:nbits]
        return "".join(str(int(x)) for x in b)

    def for_capture(self, c: int) -> "RxResult":
        m = self.captures == c
        return RxResult(self.captures[m], self.msg_samples[m], self.payloads[m], self.stats)


class FrontGate:
    """Contexts created with the same gate take turns for their (HBM-bound) front-end kernels
    (ookd_rx_gate_create); pass it to every Receiver that should."""

    def __init__(self):
        self._h = lib().ookd_rx_gate_create()
        if not self._h:
            raise OokdError(-3, "ookd_rx_gate_create failed")

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().ookd_rx_gate_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Receiver:
    """Fused replacement of the reference rx loop body for whole captures in HBM."""

    def __init__(self, filt: Optional[Filter], device: Optional[Device], *,
                 max_samples: int, threshold: float = DEFAULT_THRESHOLD,
                 samples_per_buffer: int = DEFAULT_SAMPLES_PER_BUF, hip_device: int = 0,
                 max_captures: int = 1, exact_fir: bool = False, keep_fir: bool = False,
                 edge_capacity: int = 0, segment_buffers: int = 0, message_slots: int = 0,
                 message_capacity: int = 0, stream: int = 0, fsm_rounds: bool = False,
                 quiet_skip: bool = True, count_quiet: bool = False, scan_sims: bool = False,
                 front_grid: bool = False, pipeline: bool = True, pipeline_chunk_samples: int = 0,
                 front_gate: Optional["FrontGate"] = None, fir_valu: bool = False,
                 scan_tables: bool = False):
        cfg = RxConfig()
        cfg.hip_device = hip_device
        cfg.flags = ((RX_EXACT_FIR if exact_fir else 0) | (RX_KEEP_FIR if keep_fir else 0)
                     | (RX_FSM_ROUNDS if fsm_rounds else 0) | (0 if quiet_skip else RX_NO_QUIET_SKIP)
                     | (RX_COUNT_QUIET if count_quiet else 0) | (RX_SCAN_SIMS if scan_sims else 0)
                     | (RX_FRONT_GRID if front_grid else 0) | (0 if pipeline else RX_NO_PIPELINE)
                     | (RX_FIR_VALU if fir_valu else 0) | (RX_SCAN_TABLES if scan_tables else 0))
        cfg.threshold = threshold
        cfg.samples_per_buffer = samples_per_buffer
        cfg.max_samples = max_samples
        cfg.max_captures = max_captures
        cfg.edge_capacity = edge_capacity
        cfg.segment_buffers = segment_buffers
        cfg.message_slots = message_slots
        cfg.message_capacity = message_capacity
        cfg.stream = stream
        cfg.pipeline_chunk_samples = pipeline_chunk_samples
        cfg.front_gate = front_gate._h if front_gate is not None else None
        self._gate = front_gate         # keeps it alive as long as the context
        self._filter, self._device = filt, device
        self.payload_bytes = device.payload_bytes if device else 0
        self.total_decimation = filt.total_decimation if filt else 1
        self._h = lib().ookd_rx_create(C.byref(cfg), filt._h if filt else None,
                                       device._h if device else None)
        if not self._h:
            raise OokdError(-4, last_error())

    # -- runs ---------------------------------------------------------------
    def rx_device(self, d_iq_ptr: int, samples_per_capture: int, num_captures: int = 1,
                  stride: Optional[int] = None) -> RxResult:
        """Captures already resident in HBM (int16 I,Q interleaved)."""
        _check(lib().ookd_rx_process_device(self._h, d_iq_ptr, num_captures, samples_per_capture,
                                            stride if stride is not None else samples_per_capture))
        return self._result()

    def process_device(self, d_iq_ptr: int, samples_per_capture: int, num_captures: int = 1,
                       stride: Optional[int] = None) -> None:
        """ookd_rx_process_device and nothing else: when it returns the messages
        are in host memory behind ookd_rx_messages(); `result()` / `raw_stats()`
        turn them into Python objects when (and if) the caller wants them."""
        _check(lib().ookd_rx_process_device(self._h, d_iq_ptr, num_captures, samples_per_capture,
                                            stride if stride is not None else samples_per_capture))

    def submit_device(self, d_iq_ptr: int, samples_per_capture: int, num_captures: int = 1,
                      stride: Optional[int] = None) -> None:
        """Queue a run on this context's stream and return (ookd_rx_submit_device)."""
        _check(lib().ookd_rx_submit_device(self._h, d_iq_ptr, num_captures, samples_per_capture,
                                           stride if stride is not None else samples_per_capture))

    def wait(self) -> None:
        """Block until the submitted run's results are in host memory (ookd_rx_wait)."""
        _check(lib().ookd_rx_wait(self._h))

    def result(self) -> RxResult:
        return self._result()

    def raw_stats(self) -> RxStats:
        s = RxStats()
        _check(lib().ookd_rx_get_stats(self._h, C.byref(s)))
        return s

    def rx(self, iq: np.ndarray) -> RxResult:
        """Host capture (staged over PCIe first)."""
        iq = np.ascontiguousarray(iq, dtype=np.int16).reshape(-1)
        _check(lib().ookd_rx_process_host(self._h, iq.ctypes.data, iq.size // 2))
        return self._result()

    def shard_begin(self, d_iq_ptr: int, num_samples: int, halo: Optional[np.ndarray],
                    last_shard: bool, state_in: Optional[FsmState]) -> Tuple[RxResult, FsmState]:
        out = FsmState()
        hp, hn = None, 0
        if halo is not None and hasattr(halo, "data_ptr"):
            # a device (or host) tensor an RCCL recv landed in: handed over as it is, never staged through numpy
            hp, hn = halo.data_ptr(), halo.numel() // 2
        elif halo is not None:
            halo = np.ascontiguousarray(halo, dtype=np.int16).reshape(-1)
            hp, hn = halo.ctypes.data, halo.size // 2
        _check(lib().ookd_rx_shard_begin(self._h, d_iq_ptr, num_samples, hp, hn, int(last_shard),
                                         C.byref(state_in) if state_in is not None else None,
                                         C.byref(out)))
        return self._result(), out

    def shard_refine(self, state_in: FsmState) -> Tuple[RxResult, FsmState]:
        out = FsmState()
        _check(lib().ookd_rx_shard_refine(self._h, C.byref(state_in), C.byref(out)))
        return self._result(), out

    @property
    def halo_samples(self) -> int:
        return int(lib().ookd_rx_halo_samples(self._h))

    @staticmethod
    def state_from_bytes(raw: bytes) -> FsmState:
        return FsmState.from_buffer_copy(raw)

    # -- results --------------------------------------------------------------
    def stats(self) -> dict:
        s = RxStats()
        _check(lib().ookd_rx_get_stats(self._h, C.byref(s)))
        return {name: getattr(s, name) for name, _ in RxStats._fields_}

    def _result(self) -> RxResult:
        n = int(lib().ookd_rx_num_messages(self._h))
        pb = self.payload_bytes
        caps = np.zeros(n, dtype=np.uint32)
        samples = np.zeros(n, dtype=np.uint64)
        pay = np.zeros((n, pb), dtype=np.uint8)
        if n:
            raw = np.ctypeslib.as_array(
                C.cast(lib().ookd_rx_messages(self._h), C.POINTER(C.c_uint8)),
                shape=(n, C.sizeof(Message))).copy()
            caps = raw[:, 0:4].copy().view(np.uint32).reshape(-1)
            samples = raw[:, 8:16].copy().view(np.uint64).reshape(-1)
            pay = raw[:, 16:16 + pb].copy()
        return RxResult(caps, samples, pay, self.stats())

    def bits(self, capture: int = 0) -> np.ndarray:
        """Thresholded stream as one uint8 per decimated sample."""
        nw = int(lib().ookd_rx_bit_words(self._h))
        words = np.zeros(max(nw, 1), dtype=np.uint64)
        _check(lib().ookd_rx_get_bits(self._h, capture, words.ctypes.data, nw))
        n = self.stats()["decimated_samples"]
        return np.unpackbits(words[:nw].view(np.uint8), bitorder="little")[:n]

    def edges(self, capture: int = 0) -> np.ndarray:
        n = C.c_uint64(0)
        _check(lib().ookd_rx_get_edges(self._h, capture, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.uint64)
        _check(lib().ookd_rx_get_edges(self._h, capture, out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value]

    def fir_output(self, capture: int = 0) -> np.ndarray:
        n = self.stats()["decimated_samples"]
        out = np.zeros((max(n, 1), 2), dtype=np.float32)
        _check(lib().ookd_rx_get_fir(self._h, capture, out.ctypes.data, n))
        return out[:n]

    # -- recorders (ookiedokie.c:146-169, :265-270) -------------------------------
    def dig_text(self, capture: int = 0) -> str:
        """The text --rx-rec-dig would hold for this capture."""
        need = lib().ookd_rx_dig_text(self._h, capture, None, 0)
        if need == 0 and last_error():
            raise OokdError(-1, last_error())
        buf = C.create_string_buffer(need + 1)
        lib().ookd_rx_dig_text(self._h, capture, buf, len(buf))
        return buf.value.decode("ascii")

    def record_dig(self, path: str, capture: int = 0) -> None:
        _check(lib().ookd_rx_record_dig(self._h, capture, os.fsencode(path)))

    def fir_sc16q11(self, capture: int = 0) -> np.ndarray:
        """Post-filter samples as SC16Q11 (needs keep_fir): what --rx-rec records."""
        n = self.stats()["decimated_samples"]
        out = np.zeros(2 * max(n, 1), dtype=np.int16)
        _check(lib().ookd_rx_get_fir_sc16q11(self._h, capture, out.ctypes.data, n))
        return out[:2 * n]

    def record_fir(self, path: str, capture: int = 0) -> None:
        _check(lib().ookd_rx_record_fir(self._h, capture, os.fsencode(path)))

    def errors(self, cap: int = 1 << 16) -> Tuple[np.ndarray, int]:
        n = C.c_uint64(0)
        out = np.zeros(cap, dtype=np.uint64)
        _check(lib().ookd_rx_get_errors(self._h, out.ctypes.data, cap, C.byref(n)))
        return out[:min(cap, n.value)], int(n.value)

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().ookd_rx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StreamFir:
    """fir_filter_and_decimate with carried state (fir.h:68-81), GPU backed."""

    def __init__(self, filt: Filter, max_input: int, hip_device: int = 0):
        self._filter = filt
        self.max_input = max_input
        self._h = lib().ookd_fir_create(hip_device, filt._h, max_input, 0)
        if not self._h:
            raise OokdError(-4, last_error())

    def reset(self) -> None:
        lib().ookd_fir_reset(self._h)

    def filter_and_decimate(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 2)
        out = np.zeros((x.shape[0] + 2, 2), dtype=np.float32)
        n = lib().ookd_fir_filter_and_decimate(self._h, x.ctypes.data, x.shape[0], out.ctypes.data)
        err = last_error()
        if n == 0 and err:
            raise OokdError(-4, err)
        return out[:n].copy()

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().ookd_fir_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Synth:
    """Deterministic synthetic capture (same samples on host and device)."""

    def __init__(self, device: Device, num_samples: int, *, seed: int = 0x00C0FFEE,
                 sample_rate: int = DEFAULT_RATE, amplitude: int = 1945, noise: int = 40,
                 gap_us: Tuple[int, int] = (4000, 20000), glitch_every: int = 64,
                 random_phase: bool = True):
        cfg = SynthConfig(seed, sample_rate, amplitude, noise, gap_us[0], gap_us[1],
                          glitch_every, int(random_phase))
        self._device = device
        self.num_samples = num_samples
        self._h = lib().ookd_synth_create(device._h, C.byref(cfg), num_samples)
        if not self._h:
            raise OokdError(-1, last_error())

    @property
    def num_messages(self) -> int:
        return int(lib().ookd_synth_num_messages(self._h))

    def message(self, i: int) -> Tuple[int, bytes]:
        start = C.c_uint64(0)
        buf = (C.c_uint8 * MAX_PAYLOAD_BYTES)()
        _check(lib().ookd_synth_message(self._h, i, C.byref(start), buf))
        return int(start.value), bytes(buf)[:self._device.payload_bytes]

    def fill_host(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        count = self.num_samples - first if count is None else count
        out = np.zeros(2 * count, dtype=np.int16)
        _check(lib().ookd_synth_fill_host(self._h, first, count, out.ctypes.data))
        return out

    def fill_device(self, d_ptr: int, first: int = 0, count: Optional[int] = None,
                    hip_device: int = 0, stream: int = 0) -> None:
        count = self.num_samples - first if count is None else count
        _check(lib().ookd_synth_fill_device(self._h, hip_device, first, count, d_ptr, stream))

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().ookd_synth_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipFileBackend:
    """The five backend functions of SDR_INTERFACE(hip_file, ...) as an object."""

    def __init__(self, path: str, *, rx: bool = True,
                 samples_per_buffer: int = DEFAULT_SAMPLES_PER_BUF):
        cfg = HostCfg()
        cfg.sdr_type = b"hip_file"
        cfg.direction = 0 if rx else 1
        self._path = os.fsencode(path)
        cfg.sdr_args = self._path
        cfg.samples_per_buffer = samples_per_buffer
        cfg.rx_threshold = DEFAULT_THRESHOLD
        cfg.samplerate = DEFAULT_RATE
        self._cfg = cfg
        self._h = lib().sdr_hip_file_init(C.byref(cfg))
        if not self._h:
            raise OokdError(-2, last_error())

    def rx(self, count: int) -> Tuple[int, np.ndarray]:
        out = np.zeros((count, 2), dtype=np.float32)
        status = lib().sdr_hip_file_rx(self._h, out.ctypes.data, count)
        return status, out

    def tx(self, samples: np.ndarray) -> int:
        samples = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1, 2)
        return lib().sdr_hip_file_tx(self._h, samples.ctypes.data, samples.shape[0])

    def flush(self) -> int:
        return lib().sdr_hip_file_flush(self._h)

    def capture(self) -> Tuple[int, int]:
        p, n = C.c_void_p(0), C.c_uint64(0)
        _check(lib().sdr_hip_file_capture(self._h, C.byref(p), C.byref(n)))
        return int(p.value or 0), int(n.value)

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().sdr_hip_file_deinit(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
