/*
 * ook_oracle.h -- CPU restatement of the OOKiedokie rx hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product path
 * (ookiedokie_amd/) never links or calls it.
 *
 * Each function cites the reference file:line (under /root/reference) whose
 * behaviour it restates.  The restatement is plain scalar C, compiled with
 * -ffp-contract=off and without -march=native so float arithmetic keeps the
 * reference's unfused mul-then-add rounding (the reference
 * Release build is scalar mulss/addss).
 *
 * Parity pinning (see oracle/README.md): the state machine restatement is
 * checked sample-for-sample against the reference's own state_machine.c
 * compiled into oracle/_ref; unpack / magnitude against complexf.h compiled
 * into oracle/_ref; the FIR restatement against the known-answer vectors of
 * SURVEY.md section 8(c) (fir.c itself needs libjansson, absent here).
 */
#ifndef OOK_ORACLE_H
#define OOK_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/complexf.h:31-34 */
typedef struct ook_cf {
    float re;
    float im;
} ook_cf;

/* src/state_machine.h:33-52 -- same numeric values as the reference enums */
enum {
    OOK_COND_INVALID = 0,
    OOK_COND_ALWAYS,
    OOK_COND_PULSE_START,
    OOK_COND_PULSE_END,
    OOK_COND_TIMEOUT,
    OOK_COND_MSG_COMPLETE
};

enum {
    OOK_ACT_INVALID = 0,
    OOK_ACT_NONE,
    OOK_ACT_APPEND_0,
    OOK_ACT_APPEND_1,
    OOK_ACT_OUTPUT_DATA
};

/* src/state_machine.h:56-60 */
enum {
    OOK_RESULT_ERROR = -1,
    OOK_RESULT_NO_OUTPUT = 0,
    OOK_RESULT_OUTPUT_READY = 1
};

#define OOK_MAX_PAYLOAD_BYTES 64

/*
 * Flat description of a device state machine: what sm_init / sm_add_state /
 * sm_add_state_trigger (src/state_machine.c:135-335) build from a device
 * JSON.  State 0 is the reset state (state_machine.c:51-52).
 * Triggers of state s are trig_*[trig_begin[s] .. trig_begin[s+1]).
 */
typedef struct ook_fsm_desc {
    uint32_t num_states;
    uint32_t max_bits;
    uint32_t sample_rate;       /* already divided by total decimation (main.c:683) */
    uint32_t num_triggers;
    const uint64_t *state_duration_us;
    const uint64_t *state_timeout_us;
    const uint32_t *trig_begin;     /* num_states + 1 entries */
    const uint8_t  *trig_cond;
    const uint8_t  *trig_action;
    const uint32_t *trig_next;
    const uint64_t *trig_duration_us;
} ook_fsm_desc;

/* Multi-stage FIR description: what fir_init builds (src/fir.c:68-249). */
typedef struct ook_fir_desc {
    uint32_t num_stages;
    const uint32_t *decimation;     /* per stage */
    const uint32_t *num_taps;       /* per stage */
    const float *taps;              /* all stages concatenated */
} ook_fir_desc;

typedef struct ook_msg {
    uint64_t sample;                /* decimated-domain index of the sample on
                                       which sm_process returned OUTPUT_READY */
    uint8_t payload[OOK_MAX_PAYLOAD_BYTES];
} ook_msg;

/* ---- stage functions ------------------------------------------------- */

/* src/complexf.h:68-77 */
void ook_unpack(const int16_t *in, ook_cf *out, size_t n);

/* src/complexf.h:87-96 (tx direction; used to build fixtures) */
void ook_pack(const ook_cf *in, int16_t *out, size_t n);

/* src/ookiedokie.c:171-179 with src/complexf.h:43-58 */
void ook_threshold(const ook_cf *in, float thr, uint8_t *bits, size_t n);

/* Streaming FIR: src/fir.c:39-66 state, :272-295 reset, :302-395 filter. */
typedef struct ook_fir ook_fir;
ook_fir *ook_fir_new(const ook_fir_desc *d);
void ook_fir_reset(ook_fir *f);
void ook_fir_free(ook_fir *f);
unsigned ook_fir_total_decimation(const ook_fir *f);
size_t ook_fir_run(ook_fir *f, const ook_cf *in, size_t n, ook_cf *out);

/* Per-sample state machine: src/state_machine.c:100-133, :365-556. */
typedef struct ook_sm ook_sm;
ook_sm *ook_sm_new(const ook_fsm_desc *d);
void ook_sm_free(ook_sm *sm);
int ook_sm_process(ook_sm *sm, const uint8_t *bits, unsigned count,
                   unsigned *num_proc);
const uint8_t *ook_sm_data(const ook_sm *sm);
/* Introspection for differential tests against oracle/_ref. */
void ook_sm_peek(const ook_sm *sm, uint32_t *state, uint32_t *num_bits,
                 double *elapsed_us, int *prev_bit);

/* ---- whole-path driver ------------------------------------------------
 * Restates ookiedokie_rx (src/ookiedokie.c:238-290) fed by the file
 * backend (src/sdr/bladeRF_file.c:97-126) over an in-memory SC16Q11
 * capture: buffers of samples_per_buffer input samples, zero padded final
 * buffer, optional FIR, threshold, device_process with the
 * drop-rest-of-buffer-on-error rule (src/device.c:634-658).
 *
 * fir may be NULL (reference "-F none", ookiedokie.c:260-263).
 * Optional outputs (any may be NULL):
 *   msgs / msg_cap / *num_msgs   decoded messages (count keeps going past cap)
 *   err_samples / err_cap / *num_errs   decimated index of each FSM ERROR
 *   bits_out    one byte per decimated sample actually thresholded
 *   fir_out     post-filter complexf per decimated sample
 * Returns the number of decimated samples produced.
 */
uint64_t ook_oracle_rx(const int16_t *iq, uint64_t num_samples,
                       const ook_fir_desc *fir, float threshold,
                       const ook_fsm_desc *fsm, uint32_t samples_per_buffer,
                       ook_msg *msgs, uint64_t msg_cap, uint64_t *num_msgs,
                       uint64_t *err_samples, uint64_t err_cap,
                       uint64_t *num_errs,
                       uint8_t *bits_out, ook_cf *fir_out);

/*
 * Replay of the reference's elapsed_us accumulation
 * (state_machine.c:78-82, :514) to obtain integer sample-count windows:
 * for a duration d (state or trigger, :100-133) the smallest / largest k
 * with E(k) inside the float window, and for a timeout t (:460-461) the
 * smallest k with E(k) >= t.  E(0)=0, E(k)=E(k-1)+(1.0/rate)*1e6.
 * Returns 0 on success, -1 if the replay limit was hit.
 */
int ook_duration_window(uint32_t rate, uint64_t duration_us,
                        uint64_t *kmin, uint64_t *kmax);
int ook_timeout_count(uint32_t rate, uint64_t timeout_us, uint64_t *kto);

#ifdef __cplusplus
}
#endif

#endif
