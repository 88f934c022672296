/*
 * ref_driver.c -- thin driver around the REFERENCE's own sources.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; it is compiled together
 * with /root/reference/src/state_machine.c and /root/reference/src/log.c
 * (where they lie, never copied) into oracle/_ref/libookref.so by
 * oracle/Makefile, and includes the reference's complexf.h for the
 * header-only unpack / pack / magnitude.  It exposes them through flat
 * C entry points so tests can run the restatement (ook_oracle.c) and the
 * real reference code on the same inputs.
 *
 * Not reachable from the reference without libjansson (absent in this
 * image): fir.c and device.c.  The ten-line device_process loop
 * (src/device.c:634-658) is therefore restated below on top of the real
 * sm_process; the FIR has no reference build at all (see oracle/README.md).
 */
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "complexf.h"
#include "state_machine.h"
#include "log.h"

struct ref_sm {
    struct state_machine *sm;
    uint8_t *data;
    unsigned max_bits;
};

static void state_name(unsigned idx, char *buf, size_t len)
{
    if (idx == 0) {
        snprintf(buf, len, "reset");
    } else {
        snprintf(buf, len, "s%u", idx);
    }
}

/* Builds a reference state machine through its public API
 * (state_machine.h:63-137) from the same flat tables ook_fsm_desc holds. */
struct ref_sm *ref_sm_new(uint32_t num_states, uint32_t max_bits,
                          uint32_t sample_rate,
                          const uint64_t *state_duration_us,
                          const uint64_t *state_timeout_us,
                          const uint32_t *trig_begin,
                          const uint8_t *trig_cond,
                          const uint8_t *trig_action,
                          const uint32_t *trig_next,
                          const uint64_t *trig_duration_us)
{
    char name[32], next[32];
    struct ref_sm *r = calloc(1, sizeof(*r));

    log_set_verbosity(LOG_LEVEL_SILENT);
    r->max_bits = max_bits;
    /* one spare byte: the reference may store bit index == max_bits
     * (state_machine.c:370) */
    r->data = calloc((max_bits + 7) / 8 + 1, 1);
    r->sm = sm_init(num_states, r->data, max_bits, sample_rate);
    if (!r->sm) {
        free(r->data);
        free(r);
        return NULL;
    }
    for (uint32_t s = 0; s < num_states; s++) {
        state_name(s, name, sizeof(name));
        uint32_t nt = trig_begin[s + 1] - trig_begin[s];
        if (!sm_add_state(r->sm, name, state_duration_us[s],
                          state_timeout_us[s], nt)) {
            return NULL;
        }
        for (uint32_t t = trig_begin[s]; t < trig_begin[s + 1]; t++) {
            state_name(trig_next[t], next, sizeof(next));
            if (!sm_add_state_trigger(r->sm, name,
                                      (enum sm_trigger_cond)trig_cond[t],
                                      trig_duration_us[t], next,
                                      (enum sm_trigger_action)trig_action[t])) {
                return NULL;
            }
        }
    }
    if (!sm_initialized(r->sm)) {
        return NULL;
    }
    return r;
}

void ref_sm_free(struct ref_sm *r)
{
    if (r) {
        sm_deinit(r->sm);
        free(r->data);
        free(r);
    }
}

int ref_sm_process(struct ref_sm *r, const uint8_t *bits, unsigned count,
                   unsigned *num_proc)
{
    /* bool is one byte holding 0/1 on this ABI; callers pass 0/1 only. */
    return (int)sm_process(r->sm, (const bool *)bits, count, num_proc);
}

const uint8_t *ref_sm_data(const struct ref_sm *r)
{
    return r->data;
}

/*
 * The device_process loop (src/device.c:634-658) over a whole bit stream
 * cut into buffers of buf_len samples, on top of the reference sm_process.
 * msg_samples / payloads (payload_stride bytes each) and err_samples are
 * optional.  Returns number of messages.
 */
uint64_t ref_device_stream(struct ref_sm *r, const uint8_t *bits,
                           uint64_t total_bits, uint32_t buf_len,
                           uint64_t *msg_samples, uint8_t *payloads,
                           uint32_t payload_stride, uint64_t msg_cap,
                           uint64_t *err_samples, uint64_t err_cap,
                           uint64_t *num_errs)
{
    uint64_t n_msgs = 0, n_errs = 0;
    const unsigned nbytes = (r->max_bits + 7) / 8;

    for (uint64_t base = 0; base < total_bits; base += buf_len) {
        unsigned count = (total_bits - base) < buf_len
                             ? (unsigned)(total_bits - base) : buf_len;
        unsigned total = 0, nproc = 0;
        enum sm_process_result proc = SM_PROCESS_RESULT_NO_OUTPUT;

        while (total < count && proc != SM_PROCESS_RESULT_ERROR) {
            proc = sm_process(r->sm, (const bool *)(bits + base + total),
                              count - total, &nproc);
            total += nproc;
            if (proc == SM_PROCESS_RESULT_OUTPUT_READY) {
                if (n_msgs < msg_cap) {
                    if (msg_samples) {
                        msg_samples[n_msgs] = base + total - 1;
                    }
                    if (payloads) {
                        memset(payloads + n_msgs * payload_stride, 0,
                               payload_stride);
                        memcpy(payloads + n_msgs * payload_stride, r->data,
                               nbytes < payload_stride ? nbytes
                                                       : payload_stride);
                    }
                }
                n_msgs++;
            } else if (proc == SM_PROCESS_RESULT_ERROR) {
                if (err_samples && n_errs < err_cap) {
                    err_samples[n_errs] = base + total - 1;
                }
                n_errs++;
            }
        }
    }
    if (num_errs) {
        *num_errs = n_errs;
    }
    return n_msgs;
}

/* sm_generate (state_machine.c:825-873): returns a malloc'd float pair
 * array the caller releases with ref_free. */
float *ref_sm_generate(struct ref_sm *r, const uint8_t *payload,
                       unsigned num_bits, float on_val, unsigned *num_samples)
{
    uint8_t tmp[80] = {0};
    memcpy(tmp, payload, (num_bits + 7) / 8);
    struct complexf *s = sm_generate(r->sm, tmp, num_bits, on_val,
                                     num_samples);
    return (float *)s;
}

void ref_free(void *p)
{
    free(p);
}

/* complexf.h:68-77 */
void ref_unpack(const int16_t *in, float *out, unsigned n)
{
    sc16q11_to_complexf(in, (struct complexf *)out, n);
}

/* complexf.h:87-96 */
void ref_pack(const float *in, int16_t *out, unsigned n)
{
    complexf_to_sc16q11((const struct complexf *)in, out, n);
}

/* ookiedokie.c:171-179 uses complexf_magnitude (complexf.h:55-58); the
 * function itself is static in ookiedokie.c, so its one line is repeated
 * here around the reference inline. */
void ref_threshold(const float *in, float thr, uint8_t *bits, unsigned n)
{
    const struct complexf *x = (const struct complexf *)in;
    for (unsigned i = 0; i < n; i++) {
        bits[i] = complexf_magnitude(&x[i]) >= thr;
    }
}
