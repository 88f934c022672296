/*
 * ookiedokie_amd.h -- C ABI of the MI355X-native OOK receive / demodulation
 * path (libookiedokie_amd.so).
 *
 * This is the drop-in boundary for OOKiedokie's rx hot loop
 *     SC16Q11 -> complexf -> FIR -> |.| >= thr -> symbol state machine
 * (reference: src/ookiedokie.c:238-290).  Plain C types only: pointers,
 * sizes, opaque handles.  Every entry point names the reference interface
 * it replaces (file:line under the OOKiedokie source tree).  How a
 * maintainer binds it from the existing C host is shown in INTEGRATION.md.
 *
 * Conventions kept from the reference:
 *   - constructors return a heap handle or NULL; `*_free(NULL)` is a no-op;
 *   - functions returning int use 0 = success, non-zero = failure;
 *     OOKD_FILE_EOF (= INT_MIN, src/sdr/sdr.h:36) means clean end of input;
 *   - nothing here calls exit(); the text of the last failure on the
 *     calling thread is available from ookd_last_error() (the reference
 *     prints the same kind of text through log_error, src/log.h);
 *   - handles are not thread safe; one rx context per host thread / GPU.
 *
 * There is NO CPU fallback: every compute entry point needs a HIP device
 * and fails with OOKD_ERR_HIP otherwise.
 */
#ifndef OOKIEDOKIE_AMD_H
#define OOKIEDOKIE_AMD_H

#include <limits.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OOKD_API_VERSION 1

/* src/sdr/sdr.h:36 */
#define OOKD_FILE_EOF INT_MIN

enum {
    OOKD_OK = 0,
    OOKD_ERR_ARG = -1,          /* bad argument / unsupported configuration  */
    OOKD_ERR_IO = -2,           /* file could not be opened / read            */
    OOKD_ERR_PARSE = -3,        /* JSON syntax or schema error                */
    OOKD_ERR_HIP = -4,          /* HIP runtime failure (message has details)  */
    OOKD_ERR_CAPACITY = -5,     /* a device-side list overflowed; see message */
    OOKD_ERR_NOMEM = -6
};

/* Payload bytes carried per decoded message: supports num_bits <= 256. */
#define OOKD_MAX_PAYLOAD_BYTES 32

/* src/complexf.h:31-34 */
typedef struct ookd_complexf {
    float real;
    float imag;
} ookd_complexf;

/* Text of the last error raised on this thread ("" if none). */
const char *ookd_last_error(void);
int ookd_api_version(void);

/* ------------------------------------------------------------------------
 * Filter: replaces fir_init / fir_deinit / fir_get_total_decimation
 * (src/fir.h:43-66, loader src/fir.c:68-249).  The JSON schema is the
 * reference's, unchanged (filters/README.md:31-63): decimation optional,
 * default 1, must be > 0; taps are numbers cast double -> float.
 * `path` is an explicit file name (the reference's search path, src/find.c,
 * stays with the host).
 * ---------------------------------------------------------------------- */
typedef struct ookd_filter ookd_filter;

ookd_filter *ookd_filter_load(const char *path);
ookd_filter *ookd_filter_create(uint32_t num_stages,
                                const uint32_t *decimation,
                                const uint32_t *num_taps,
                                const float *taps /* all stages, in order */);
void ookd_filter_free(ookd_filter *f);
uint32_t ookd_filter_total_decimation(const ookd_filter *f);  /* fir.h:66 */
uint32_t ookd_filter_num_stages(const ookd_filter *f);
/* Stage geometry and a pointer to its float taps (owned by the filter). */
int ookd_filter_stage(const ookd_filter *f, uint32_t stage,
                      uint32_t *decimation, uint32_t *num_taps,
                      const float **taps);

/* ------------------------------------------------------------------------
 * Device: replaces device_init / device_deinit (src/device.h:47-91, loader
 * src/device.c:76-632) for the rx direction.  `sample_rate` is the rate the
 * state machine sees, i.e. samplerate / total decimation (src/main.c:683).
 * The device JSON is consumed unchanged (devices/README.md).
 * ---------------------------------------------------------------------- */
typedef struct ookd_device ookd_device;

/* Flat view of the state machine tables (what sm_add_state /
 * sm_add_state_trigger were fed, src/state_machine.c:250-335); state 0 is
 * the reset state.  Pointers are owned by the device handle. */
typedef struct ookd_fsm_tables {
    uint32_t num_states;
    uint32_t max_bits;
    uint32_t sample_rate;
    uint32_t num_triggers;
    const uint64_t *state_duration_us;
    const uint64_t *state_timeout_us;
    const uint32_t *trig_begin;         /* num_states + 1 */
    const uint8_t *trig_cond;           /* enum sm_trigger_cond values   */
    const uint8_t *trig_action;         /* enum sm_trigger_action values */
    const uint32_t *trig_next;
    const uint64_t *trig_duration_us;
    /* Integer sample-count form of the above, obtained by replaying the
     * reference's double accumulation of elapsed_us (state_machine.c:78-82,
     * :514) against its float windows (:100-133) -- what the GPU uses.
     * A window is empty when kmin > kmax; "none" is encoded as kmin = 0,
     * kmax = UINT64_MAX; no timeout as UINT64_MAX. */
    const uint64_t *state_kmin, *state_kmax, *state_kto;
    const uint64_t *trig_kmin, *trig_kmax;
} ookd_fsm_tables;

ookd_device *ookd_device_load(const char *path, uint32_t sample_rate);
ookd_device *ookd_device_create(const ookd_fsm_tables *t /* *_us fields + counts */);
void ookd_device_free(ookd_device *d);
uint32_t ookd_device_num_bits(const ookd_device *d);
const char *ookd_device_name(const ookd_device *d);
const char *ookd_device_state_name(const ookd_device *d, uint32_t state);
int ookd_device_tables(const ookd_device *d, ookd_fsm_tables *out);

/* ------------------------------------------------------------------------
 * Rx context: the fused replacement for one or more iterations of the
 * reference loop body, src/ookiedokie.c:243-288
 *     sdr_rx -> fir_filter_and_decimate -> threshold -> device_process
 * over a whole SC16Q11 capture resident in HBM.  Semantics kept:
 *   - the capture is consumed in buffers of `samples_per_buffer` input
 *     samples; a short final buffer is zero padded and fully processed
 *     (src/sdr/bladeRF_file.c:107-119), so ceil(n/spb)*spb samples are
 *     filtered and floor(that / total_decimation) are decoded;
 *   - FIR history starts at zero (fir_reset, src/fir.c:272-295), outputs
 *     at input indices D-1, 2D-1, ...;
 *   - after a state machine ERROR the rest of THAT buffer is not fed to the
 *     state machine (src/device.c:646) -- results depend on
 *     samples_per_buffer exactly as the reference's do.
 * filter may be NULL ("-F none", ookiedokie.c:260-263); device may be NULL
 * (threshold / bit stream only).
 * ---------------------------------------------------------------------- */
typedef struct ookd_rx ookd_rx;

enum {
    /* FIR arithmetic. Default (0): fused multiply-add accumulation plus a
     * guard band around the threshold inside which the sample is recomputed
     * in the reference's exact order -- bits are identical to the
     * reference, floats within 1e-5.  EXACT: unfused mul/add in reference
     * order everywhere -- floats bit-identical too, ~half the speed. */
    OOKD_RX_EXACT_FIR = 1u << 0,
    /* Keep the post-filter complexf stream in HBM (parity / --rx-rec). */
    OOKD_RX_KEEP_FIR = 1u << 1,
    /* State machine: always use the segment/round path.  By default the
     * state machine runs as a scan of per-edge transition functions and
     * falls back to this path only when a capture leaves the scan's model
     * (results are identical either way). */
    OOKD_RX_FSM_ROUNDS = 1u << 2,
    /* Front end: never take the "quiet" shortcut.  By default a wavefront whose
     * whole input window is provably too small to reach the threshold
     * (sqrt(2) * sum|taps| * max|sample| < threshold) emits its 1024 zero bits
     * without running the filter -- the bits are identical, captures that are
     * mostly silence run at memory speed.  Set this to time the worst case. */
    OOKD_RX_NO_QUIET_SKIP = 1u << 3,
    /* Diagnostics: count the windows that took the shortcut (one atomic per
     * quiet window -- slows the front end, keep out of timed runs). */
    OOKD_RX_COUNT_QUIET = 1u << 4,
    /* State machine scan: simulate every span instead of looking its result
     * up in the per-device span tables built at create time (identical
     * results; exists so the tests can run both). */
    OOKD_RX_SCAN_SIMS = 1u << 5,
    /* Front end, packed-VALU form: the hardware-dispatched grid of one-tile
     * workgroups.  This IS the default; the flag only matters to a developer
     * build that selected the experimental persistent streaming form
     * (OOKD_DEVELOPER=1 OOKD_FRONT_STREAM=1), where it forces
     * the grid form back (identical bits). */
    OOKD_RX_FRONT_GRID = 1u << 6,
    /* Never pipeline a long capture in chunks (see pipeline_chunk_samples). */
    OOKD_RX_NO_PIPELINE = 1u << 7,
    /* Front end, single-stage filters of up to 256 taps without decimation: by
     * default the filter runs on the matrix cores (taps and samples split into
     * fp16 pieces whose products are exact, fp32 accumulation, guard band +
     * exact recompute as in the fused form: identical bits, floats within 1e-5).
     * Set this for the packed-VALU loop instead (exists so the tests run both). */
    OOKD_RX_FIR_VALU = 1u << 8,
    /* State machine scan: always compose the per-block transition tables.  By
     * default an edge list of 20 000 edges and more is first searched for
     * SYNCHRONISING spans -- stretches of constant level long enough that the
     * machine can only end them in one of a few states whatever state it
     * entered them in (the silence between two messages) --, every stretch
     * between two of them is walked once per such state, and a scan over the
     * resulting few-entry maps picks the true one; the composing kernels run
     * for short edge lists and for captures without such spans (no one within
     * 512 edges).  Identical results; the flag exists so the tests can run
     * both (stats.scan_entry_form). */
    OOKD_RX_SCAN_TABLES = 1u << 9
};

/* Contexts created with the same gate (and on the same device) queue their front-end kernels one
 * after the other instead of side by side: the front end is HBM bound, two at once only slow each
 * other down, while the latency-bound edges / state machine of one capture do overlap the next
 * context's front end.  For hosts that keep several captures in flight (bench.py does).  The gate
 * must outlive the contexts that use it; contexts without one never wait for anybody. */
typedef struct ookd_rx_gate ookd_rx_gate;
ookd_rx_gate *ookd_rx_gate_create(void);
void ookd_rx_gate_destroy(ookd_rx_gate *gate);

typedef struct ookd_rx_config {
    int32_t hip_device;             /* ordinal, e.g. LOCAL_RANK               */
    uint32_t flags;                 /* OOKD_RX_*                              */
    float threshold;                /* cfg->rx_threshold, default 0.1f        */
    uint32_t samples_per_buffer;    /* cfg->samples_per_buffer, default 8192  */
    uint64_t max_samples;           /* largest capture (input samples) / run  */
    uint32_t max_captures;          /* captures per batched run (>= 1)        */
    uint64_t edge_capacity;         /* 0 = default (max_samples/32 + 1M)      */
    uint32_t segment_buffers;       /* buffers per FSM segment, 0 = default   */
    uint32_t message_slots;         /* per segment, 0 = default               */
    uint64_t message_capacity;      /* messages per run, 0 = default (65536)  */
    void *stream;                   /* hipStream_t to launch on, NULL = own   */
    uint64_t pipeline_chunk_samples;/* non-zero: single-capture runs at least twice this long are
                                       pipelined in chunks of about this many input samples: the front
                                       end of chunk c+1 is queued beside the edges / state machine of
                                       chunk c, the state machine's state carried from chunk to chunk in
                                       device memory (results identical).  0 = never (the default: on
                                       this runtime the chunked run is slower)          */
    ookd_rx_gate *front_gate;       /* shared front-end gate (see above), NULL = none */
} ookd_rx_config;

typedef struct ookd_message {
    uint32_t capture;               /* index within a batched run             */
    uint32_t reserved;
    uint64_t sample;                /* decimated sample index on which the
                                       state machine returned OUTPUT_READY
                                       (sm_process, state_machine.c:541-556)  */
    uint8_t payload[OOKD_MAX_PAYLOAD_BYTES]; /* first received bit = bit 0 of
                                       byte 0 (state_machine.c:365-385)       */
} ookd_message;

/* Carried state of the symbol state machine between shards of one capture
 * (what struct state_machine holds across sm_process calls,
 * src/state_machine.c:57-75): current state, increments of elapsed_us since
 * it was last zeroed, previous bit, bits collected so far and the payload. */
typedef struct ookd_fsm_state {
    uint32_t state;
    uint32_t num_bits;
    uint64_t k;
    uint32_t prev_bit;
    uint32_t reserved;
    uint8_t payload[OOKD_MAX_PAYLOAD_BYTES + 8];
} ookd_fsm_state;

typedef struct ookd_rx_stats {
    uint64_t input_samples;         /* per capture, after zero padding        */
    uint64_t decimated_samples;     /* per capture                            */
    uint64_t num_edges;             /* whole batch                            */
    uint64_t num_messages;
    uint64_t num_errors;            /* state machine ERROR events             */
    uint64_t guard_recomputes;      /* samples redone in exact order          */
    uint32_t fsm_iterations;        /* segment-parallel fix-point rounds      */
    uint32_t num_segments;
    uint32_t fsm_path;              /* 1 scan, 2 rounds, 3 scan fell back to rounds */
    uint32_t fsm_fallback_reason;   /* scan's refusal bits (0 = none)         */
    float fir_kernel_ms;            /* HIP-event time of the dominant kernel  */
    float total_device_ms;          /* first kernel start -> last kernel end  */
    uint64_t quiet_waves;           /* 1024-output windows that took the quiet
                                       shortcut (only with OOKD_RX_COUNT_QUIET) */
    uint64_t total_waves;           /* 1024-output windows of the run (1-stage
                                       decimation-1 front end; else 0)        */
    uint32_t pipeline_chunks;       /* chunks the run was pipelined in (0 = not pipelined) */
    uint32_t front_launches;        /* grid launches the front end went out as; fir_kernel_ms spans
                                       first start -> last end                */
    uint32_t scan_entry_form;       /* how the scan found the leaves' entry states: 1 walk from
                                       synchronising spans, 2 composed block tables, 0 no scan */
    uint32_t reserved;
} ookd_rx_stats;

ookd_rx *ookd_rx_create(const ookd_rx_config *cfg, const ookd_filter *filter,
                        const ookd_device *device);
void ookd_rx_destroy(ookd_rx *rx);

/* Demodulate `num_captures` independent captures of `samples_per_capture`
 * SC16Q11 samples each, already resident in HBM at d_iq (int16 I,Q
 * interleaved, capture c at d_iq + 2*c*capture_stride_samples).  Each
 * capture starts from zero FIR history and a reset state machine.  Blocks
 * until results are in host memory. */
int ookd_rx_process_device(ookd_rx *rx, const void *d_iq,
                           uint32_t num_captures,
                           uint64_t samples_per_capture,
                           uint64_t capture_stride_samples);

/* The same split in two, for hosts that overlap captures: submit queues the
 * whole run on the context's stream and returns; wait blocks until its
 * results are in host memory.  One run in flight per context -- use two
 * contexts (each has its own stream) to let the memory-bound front end of
 * one capture overlap the latency-bound state machine of the previous one.
 * ookd_rx_process_device == submit + wait. */
int ookd_rx_submit_device(ookd_rx *rx, const void *d_iq,
                          uint32_t num_captures,
                          uint64_t samples_per_capture,
                          uint64_t capture_stride_samples);
int ookd_rx_wait(ookd_rx *rx);

/* Same over a host buffer: stages it to HBM first (PCIe-bound; never the
 * figure bench.py reports). */
int ookd_rx_process_host(ookd_rx *rx, const int16_t *iq,
                         uint64_t num_samples);

/* One shard of a larger capture (multi-GPU split).
 *   halo / halo_samples : the last input samples of the previous shard
 *       (host pointer, at least ookd_rx_halo_samples(rx) of them), or NULL
 *       for the first shard (zero history);
 *   last_shard : non-zero => zero pad to a whole buffer; otherwise
 *       num_samples must be a multiple of lcm(spb, total decimation);
 *   runs the front end (FIR, threshold, edges) and ONE speculative state
 *   machine pass from `state_in` (NULL = reset state), leaving the shard's
 *   outgoing state in *state_out.  Call ookd_rx_shard_refine with the true
 *   incoming state (the previous shard's state_out) until no rank's
 *   state_out changes; messages are valid after the last refine. */
int ookd_rx_shard_begin(ookd_rx *rx, const void *d_iq, uint64_t num_samples,
                        const int16_t *halo, uint64_t halo_samples,
                        int last_shard, const ookd_fsm_state *state_in,
                        ookd_fsm_state *state_out);
int ookd_rx_shard_refine(ookd_rx *rx, const ookd_fsm_state *state_in,
                         ookd_fsm_state *state_out);
uint64_t ookd_rx_halo_samples(const ookd_rx *rx);

/* Diagnostic, host only (no GPU): the abstract domain the scan form of the
 * state machine would use for this device at this rate / buffer size.
 * out = { span tables built (0/1), table intervals, intervals that need a
 * simulation, reachable codes (0 = unknown), normal codes that can get
 * "stuck" (an edge on which no trigger fires), table rows holding such a
 * result, domain size, machine states }. */
int ookd_scan_domain_info(const ookd_device *device,
                          uint32_t samples_per_buffer,
                          uint32_t total_decimation, uint32_t out[8]);

/* Results of the last run (host copies, valid until the next run). */
uint64_t ookd_rx_num_messages(const ookd_rx *rx);
const ookd_message *ookd_rx_messages(const ookd_rx *rx);
int ookd_rx_get_stats(const ookd_rx *rx, ookd_rx_stats *out);

/* Parity / recorder taps (device -> host copies of intermediate data):
 *   bits  : 1 bit per decimated sample, LSB-first in 64-bit words, per
 *           capture `ookd_rx_bit_words()` words (ookiedokie.c:171-179);
 *   edges : decimated indices whose bit differs from the previous sample's
 *           (sample -1 counts as 0) -- the content of --rx-rec-dig
 *           (ookiedokie.c:146-169);
 *   fir   : post-filter complexf (needs OOKD_RX_KEEP_FIR). */
uint64_t ookd_rx_bit_words(const ookd_rx *rx);
int ookd_rx_get_bits(const ookd_rx *rx, uint32_t capture, uint64_t *words,
                     uint64_t capacity_words);
int ookd_rx_get_edges(const ookd_rx *rx, uint32_t capture, uint64_t *edges,
                      uint64_t capacity, uint64_t *num_edges);
int ookd_rx_get_fir(const ookd_rx *rx, uint32_t capture, ookd_complexf *out,
                    uint64_t capacity);
/* errors: capture-local decimated indices of the samples on which the state
 *         machine reported an error (device.c:646), in increasing order per
 *         capture.  In a batched run the list is capture-major, EXCEPT when the
 *         scan form refused some captures and those were redone by the round
 *         form (stats.fsm_path 3): then the refused captures' errors follow
 *         those of all the others.  *num is always the total. */
int ookd_rx_get_errors(const ookd_rx *rx, uint64_t *samples, uint64_t capacity,
                       uint64_t *num);

/* Recorders, by-products of a finished run:
 *   dig : the text of `--rx-rec-dig` (record_dig, ookiedokie.c:146-169) --
 *         "0, <level of sample 0>", then per level change at i the pair
 *         "i-1, <old>" / "i, <new>"; snprintf convention for the in-memory
 *         form (returns the full length);
 *   fir : the post-filter stream as SC16Q11, what `--rx-rec` writes when
 *         rx_rec_input is false (ookiedokie.c:265-270 through
 *         complexf_to_sc16q11, complexf.h:87-96); needs OOKD_RX_KEEP_FIR.
 *         (With rx_rec_input the recording is the input capture itself.) */
size_t ookd_rx_dig_text(const ookd_rx *rx, uint32_t capture, char *out, size_t capacity);
int ookd_rx_record_dig(const ookd_rx *rx, uint32_t capture, const char *path);
int ookd_rx_get_fir_sc16q11(const ookd_rx *rx, uint32_t capture, int16_t *out,
                            uint64_t capacity_samples);
int ookd_rx_record_fir(const ookd_rx *rx, uint32_t capture, const char *path);

/* ------------------------------------------------------------------------
 * Host side of a decoded message: payload bits -> per-field text -> stdout
 * text.  Replaces formatter_data_to_keyval
 * (src/formatter.c:715-739, field rules :425-573), rx_print
 * (src/ookiedokie.c:181-220) and, for the tx side, formatter_default_data /
 * formatter_keyval_to_data (formatter.c:793-846).  Pure host code: a few
 * fields per message.  The text is byte-identical to the reference's except
 * for the optional "Decode Timestamp" value (wall clock).
 * ---------------------------------------------------------------------- */
typedef struct ookd_formatter ookd_formatter;

enum {                              /* enum ookiedokie_rx_fmt, ookiedokie_cfg.h:41-45 */
    OOKD_RX_FMT_PRETTY = 0,
    OOKD_RX_FMT_CSV = 1
};

/* create_formatter (device.c:424-499) from the device's "fields" / "ts_mode". */
ookd_formatter *ookd_formatter_create(const ookd_device *device);
void ookd_formatter_free(ookd_formatter *f);
uint32_t ookd_formatter_num_fields(const ookd_formatter *f);
const char *ookd_formatter_field_name(const ookd_formatter *f, uint32_t field);
int ookd_formatter_ts_mode(const ookd_formatter *f); /* 0 none, 1 unix, 2 unix-frac,
                                                        3 datetime-24, 4 datetime-ampm */
/* The value text of one field of a payload (at most 79 characters, the
 * reference's char buf[80]). */
int ookd_formatter_field_to_str(const ookd_formatter *f, uint32_t field,
                                const uint8_t *payload, char *out, size_t capacity);
/* formatter_default_data: every field's "default" deposited into payload;
 * formatter_keyval_to_data for one (name, value) pair.  `len` = bytes at
 * payload, at least (num_bits + 7) / 8. */
int ookd_formatter_default_data(const ookd_formatter *f, uint8_t *payload, size_t len);
int ookd_formatter_set_field(const ookd_formatter *f, const char *name, const char *value,
                             uint8_t *payload, size_t len);

/* What rx_print writes for ONE keyval list = the `count` messages decoded
 * from one sdr_rx buffer (device_process appends them to the same list,
 * device.c:634-658).  snprintf convention: returns the number of characters
 * the full text has, stores at most capacity - 1 of them plus a NUL.
 * *first_print (CSV heading pending) is read and cleared like the
 * reference's flag. */
size_t ookd_print_record(const ookd_formatter *f, int rx_fmt, int *first_print,
                         const uint8_t *const *payloads, size_t count,
                         char *out, size_t capacity);
/* The whole stdout text of a run: messages grouped into records by the
 * buffer their OUTPUT_READY sample fell into, exactly as the reference's
 * per-buffer loop prints them (ookiedokie.c:279-286). */
size_t ookd_print_messages(const ookd_formatter *f, int rx_fmt, int *first_print,
                           const ookd_message *msgs, uint64_t num_messages,
                           uint32_t samples_per_buffer, uint32_t total_decimation,
                           char *out, size_t capacity);

/* ------------------------------------------------------------------------
 * Streaming FIR with the reference's call shape: replaces
 * fir_filter_and_decimate / fir_reset (src/fir.h:68-81): history carried
 * across calls, result independent of chunking.  Host pointers in and out
 * (PCIe-bound); exists so the fine-grained reference API has a GPU-backed
 * equivalent and for the FIR float parity tests.
 * ---------------------------------------------------------------------- */
typedef struct ookd_fir ookd_fir;
ookd_fir *ookd_fir_create(int32_t hip_device, const ookd_filter *filter,
                          size_t max_input, uint32_t flags);
void ookd_fir_reset(ookd_fir *f);
void ookd_fir_destroy(ookd_fir *f);
size_t ookd_fir_filter_and_decimate(ookd_fir *f, const ookd_complexf *input,
                                    size_t count, ookd_complexf *output);

/* ------------------------------------------------------------------------
 * Synthetic capture generator: envelope = the device's tx
 * state machine walk (sm_generate, src/state_machine.c:825-873) for
 * pseudo-random payloads, on-amplitude 1945 (device.c:675 0.95f,
 * complexf.h:93 truncation), per-message carrier phase, uniform integer
 * noise, occasional glitch pulses; counter-based PRNG so the same capture
 * can be produced on the device (fills HBM directly) and on the host.
 * ---------------------------------------------------------------------- */
typedef struct ookd_synth ookd_synth;

typedef struct ookd_synth_config {
    uint64_t seed;
    uint32_t sample_rate;           /* rate of the generated capture (Hz)    */
    uint32_t amplitude;             /* on level, default 1945                */
    uint32_t noise;                 /* +-noise LSB uniform, default 40       */
    uint32_t gap_min_us, gap_max_us;/* inter-message gap, default 4000..20000*/
    uint32_t glitch_every;          /* 1 message in N gets a glitch, 0 = off */
    uint32_t random_phase;          /* 0: I only (reference tx); 1: random   */
} ookd_synth_config;

ookd_synth *ookd_synth_create(const ookd_device *device,
                              const ookd_synth_config *cfg,
                              uint64_t num_samples);
void ookd_synth_free(ookd_synth *s);
uint64_t ookd_synth_num_messages(const ookd_synth *s);
/* Expected payloads in transmit order and the input-sample index at which
 * each message's waveform starts. */
int ookd_synth_message(const ookd_synth *s, uint64_t i, uint64_t *start_sample,
                       uint8_t *payload /* OOKD_MAX_PAYLOAD_BYTES */);
/* Fill samples [first, first+count) of the capture into host memory. */
int ookd_synth_fill_host(const ookd_synth *s, uint64_t first, uint64_t count,
                         int16_t *iq);
/* Same into device memory d_iq (which receives sample `first` at offset 0). */
int ookd_synth_fill_device(const ookd_synth *s, int32_t hip_device,
                           uint64_t first, uint64_t count, void *d_iq,
                           void *stream);

/* ------------------------------------------------------------------------
 * SDR backend: the five functions the reference's backend table binds
 * (SDR_PROTOTYPES / SDR_INTERFACE, src/sdr/supported_devices.h:32-48;
 * vtable src/sdr/sdr.c:50-122).  Registered as
 *     SDR_INTERFACE(hip_file, hip_file, "fs128_fs16_dec4")
 * it is a file handler for SC16Q11 captures like bladerf_file
 * (src/sdr/bladeRF_file.c) whose unpack runs on the GPU and whose handle
 * also keeps the raw capture resident in HBM for ookd_rx_process_device.
 * `cfg` is the reference's `const struct ookiedokie_cfg *`
 * (src/ookiedokie_cfg.h:50-91); only direction, sdr_args and
 * samples_per_buffer are read (as bladeRF_file.c:63,71,79 does).  The
 * struct layout is mirrored in ookd_host_cfg below for hosts that do not
 * include the reference header.
 * ---------------------------------------------------------------------- */
typedef struct ookd_host_cfg {
    const char *sdr_type;
    int direction;                  /* 0 = rx, 1 = tx (ookiedokie_cfg.h:32-36) */
    const char *sdr_args;           /* file name                              */
    unsigned int frequency, bandwidth, samplerate;
    int gain;
    const char *device;
    unsigned int tx_count, tx_delay_us;
    void *device_params;
    int rx_fmt;
    float rx_threshold;
    const char *rx_rec_filename, *rx_rec_type, *rx_filter, *rx_rec_dig;
    unsigned char rx_rec_input;     /* bool in the reference struct */
    unsigned int samples_per_buffer, num_buffers, num_transfers;
    unsigned int stream_timeout_ms, sync_timeout_ms;
    int verbosity;
} ookd_host_cfg;

/* Declared with the reference's own type names (forward declarations only), so that a translation unit
 * which also includes the reference's headers and says SDR_PROTOTYPES(hip_file) -- as sdr.c does
 * after INTEGRATION.md section 1 -- sees the very same declarations (tests/test_boundary.py compiles
 * exactly that).  struct complexf (src/complexf.h:31-34) and ookd_complexf are layout-identical;
 * a host without the reference headers passes (const struct ookiedokie_cfg *)&its_ookd_host_cfg. */
struct ookiedokie_cfg;
struct complexf;
void *sdr_hip_file_init(const struct ookiedokie_cfg *cfg);
void sdr_hip_file_deinit(void *handle);
int sdr_hip_file_rx(void *handle, struct complexf *samples, unsigned int count);
int sdr_hip_file_tx(void *handle, const struct complexf *samples,
                    unsigned int count);
int sdr_hip_file_flush(void *handle);
/* Extra, beyond the vtable: the whole capture as a device pointer
 * (int16 I,Q interleaved) for the fused path; loads the file to HBM on
 * first use. */
int sdr_hip_file_capture(void *handle, const void **d_iq,
                         uint64_t *num_samples);

#ifdef __cplusplus
}
#endif

#endif /* OOKIEDOKIE_AMD_H */
