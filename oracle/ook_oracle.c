/*
 * ook_oracle.c -- CPU restatement of the OOKiedokie rx hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see ook_oracle.h).  Citations are file:line
 * under /root/reference.  Build with -ffp-contract=off, no -ffast-math, no
 * -march=native: the arithmetic below must round exactly like the
 * reference's scalar Release build.
 */
#include "ook_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* SC16Q11 <-> complexf                                                */
/* ------------------------------------------------------------------ */

/* src/complexf.h:68-77: each component is (float)v * (1.0f/2048.0f). */
void ook_unpack(const int16_t *in, ook_cf *out, size_t n)
{
    const float scale = 1.0f / 2048.0f;
    for (size_t j = 0; j < n; j++) {
        out[j].re = (float)in[2 * j] * scale;
        out[j].im = (float)in[2 * j + 1] * scale;
    }
}

/* src/complexf.h:87-96: truncating cast of v * 2048.0f. */
void ook_pack(const ook_cf *in, int16_t *out, size_t n)
{
    for (size_t j = 0; j < n; j++) {
        out[2 * j] = (int16_t)(in[j].re * 2048.0f);
        out[2 * j + 1] = (int16_t)(in[j].im * 2048.0f);
    }
}

/* src/complexf.h:43-58 + src/ookiedokie.c:171-179:
 * power = re*re + im*im (two products, one add, each rounded),
 * magnitude = sqrtf(power), bit = magnitude >= threshold. */
void ook_threshold(const ook_cf *in, float thr, uint8_t *bits, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        float rr = in[i].re * in[i].re;
        float ii = in[i].im * in[i].im;
        float p = rr + ii;
        bits[i] = sqrtf(p) >= thr;
    }
}

/* ------------------------------------------------------------------ */
/* FIR                                                                 */
/* ------------------------------------------------------------------ */

/*
 * One stage keeps the last num_taps inputs in a ring (the reference keeps
 * them in a double-length buffer, src/fir.c:49-54, which is only a way to
 * avoid the wrap; the values read are the same) and a countdown to the
 * next output (src/fir.c:47, :290).
 */
struct stage {
    unsigned decim;
    size_t ntaps;
    const float *taps;
    ook_cf *hist;       /* ring of ntaps samples */
    size_t head;        /* index of the newest sample */
    size_t countdown;
    ook_cf *out;        /* inter-stage buffer, grown on demand */
    size_t out_cap;
};

struct ook_fir {
    size_t nstages;
    struct stage *st;
    float *taps;
    unsigned total_decim;
};

ook_fir *ook_fir_new(const ook_fir_desc *d)
{
    if (!d || d->num_stages == 0) {
        return NULL;                /* src/fir.c:118-121 */
    }
    ook_fir *f = calloc(1, sizeof(*f));
    if (!f) {
        return NULL;
    }
    f->nstages = d->num_stages;
    f->st = calloc(f->nstages, sizeof(f->st[0]));
    size_t total = 0;
    for (size_t s = 0; s < f->nstages; s++) {
        total += d->num_taps[s];
    }
    f->taps = malloc(sizeof(float) * (total ? total : 1));
    memcpy(f->taps, d->taps, sizeof(float) * total);
    f->total_decim = 1;
    size_t off = 0;
    for (size_t s = 0; s < f->nstages; s++) {
        struct stage *g = &f->st[s];
        if (d->decimation[s] == 0 || d->num_taps[s] == 0) {
            ook_fir_free(f);        /* src/fir.c:149, :173 */
            return NULL;
        }
        g->decim = d->decimation[s];
        g->ntaps = d->num_taps[s];
        g->taps = f->taps + off;
        off += g->ntaps;
        g->hist = calloc(g->ntaps, sizeof(ook_cf));
        f->total_decim *= g->decim; /* src/fir.c:159 */
    }
    ook_fir_reset(f);
    return f;
}

/* src/fir.c:272-295: zero history, countdown = decimation. */
void ook_fir_reset(ook_fir *f)
{
    for (size_t s = 0; s < f->nstages; s++) {
        struct stage *g = &f->st[s];
        memset(g->hist, 0, g->ntaps * sizeof(ook_cf));
        g->head = 0;
        g->countdown = g->decim;
    }
}

void ook_fir_free(ook_fir *f)
{
    if (!f) {
        return;
    }
    if (f->st) {
        for (size_t s = 0; s < f->nstages; s++) {
            free(f->st[s].hist);
            free(f->st[s].out);
        }
    }
    free(f->st);
    free(f->taps);
    free(f);
}

unsigned ook_fir_total_decimation(const ook_fir *f)
{
    return f->total_decim;          /* src/fir.c:297-300 */
}

/*
 * src/fir.c:336-353 (perform_stage) with :302-334 (update): push the
 * sample, decrement the countdown, and when it reaches zero emit
 *   out = 0; for i in 0..T-1: out += taps[i] * x[newest - i]
 * with the product and the sum rounded separately, tap 0 (newest sample)
 * first, real and imaginary parts independently (:313-318).
 */
static size_t stage_run(struct stage *g, const ook_cf *in, size_t n,
                        ook_cf *out)
{
    size_t produced = 0;
    for (size_t i = 0; i < n; i++) {
        g->head = (g->head + 1 == g->ntaps) ? 0 : g->head + 1;
        g->hist[g->head] = in[i];
        if (--g->countdown == 0) {
            float acc_re = 0.0f;
            float acc_im = 0.0f;
            size_t pos = g->head;
            for (size_t k = 0; k < g->ntaps; k++) {
                float pr = g->taps[k] * g->hist[pos].re;
                float pi = g->taps[k] * g->hist[pos].im;
                acc_re = acc_re + pr;
                acc_im = acc_im + pi;
                pos = (pos == 0) ? g->ntaps - 1 : pos - 1;
            }
            out[produced].re = acc_re;
            out[produced].im = acc_im;
            produced++;
            g->countdown = g->decim;
        }
    }
    return produced;
}

/* src/fir.c:355-395: stages chained through per-stage output buffers. */
size_t ook_fir_run(ook_fir *f, const ook_cf *in, size_t n, ook_cf *out)
{
    const ook_cf *src = in;
    size_t count = n;
    for (size_t s = 0; s < f->nstages; s++) {
        struct stage *g = &f->st[s];
        ook_cf *dst;
        if (s + 1 == f->nstages) {
            dst = out;
        } else {
            size_t need = count / g->decim + 2;
            if (g->out_cap < need) {
                free(g->out);
                g->out = malloc(need * sizeof(ook_cf));
                g->out_cap = need;
            }
            dst = g->out;
        }
        count = stage_run(g, src, count, dst);
        src = dst;
    }
    return count;
}

/* ------------------------------------------------------------------ */
/* State machine (rx half)                                             */
/* ------------------------------------------------------------------ */

#define TOL 0.15    /* src/state_machine.c:55 */

struct ook_sm {
    ook_fsm_desc d;
    /* owned copies of the tables */
    uint64_t *sdur, *sto, *tdur;
    uint32_t *tbeg, *tnext;
    uint8_t *tcond, *tact;

    uint32_t cur;
    uint32_t num_bits;
    int prev_bit;
    double elapsed_us;
    uint8_t data[OOK_MAX_PAYLOAD_BYTES + 1];
};

static void *dup_mem(const void *p, size_t bytes)
{
    void *q = malloc(bytes ? bytes : 1);
    if (q && bytes) {
        memcpy(q, p, bytes);
    }
    return q;
}

ook_sm *ook_sm_new(const ook_fsm_desc *d)
{
    if (!d || d->max_bits == 0 || d->num_states == 0 ||
        (d->max_bits + 7) / 8 > OOK_MAX_PAYLOAD_BYTES) {
        return NULL;                /* src/state_machine.c:143-145 */
    }
    ook_sm *sm = calloc(1, sizeof(*sm));
    sm->d = *d;
    sm->sdur = dup_mem(d->state_duration_us, 8u * d->num_states);
    sm->sto = dup_mem(d->state_timeout_us, 8u * d->num_states);
    sm->tbeg = dup_mem(d->trig_begin, 4u * (d->num_states + 1));
    sm->tcond = dup_mem(d->trig_cond, d->num_triggers);
    sm->tact = dup_mem(d->trig_action, d->num_triggers);
    sm->tnext = dup_mem(d->trig_next, 4u * d->num_triggers);
    sm->tdur = dup_mem(d->trig_duration_us, 8u * d->num_triggers);
    sm->cur = 0;                    /* src/state_machine.c:165 */
    return sm;
}

void ook_sm_free(ook_sm *sm)
{
    if (sm) {
        free(sm->sdur);
        free(sm->sto);
        free(sm->tbeg);
        free(sm->tcond);
        free(sm->tact);
        free(sm->tnext);
        free(sm->tdur);
        free(sm);
    }
}

const uint8_t *ook_sm_data(const ook_sm *sm)
{
    return sm->data;
}

void ook_sm_peek(const ook_sm *sm, uint32_t *state, uint32_t *num_bits,
                 double *elapsed_us, int *prev_bit)
{
    if (state) *state = sm->cur;
    if (num_bits) *num_bits = sm->num_bits;
    if (elapsed_us) *elapsed_us = sm->elapsed_us;
    if (prev_bit) *prev_bit = sm->prev_bit;
}

/*
 * src/state_machine.c:100-133.  A non-zero duration d gives the window
 * [ (float)(d - 0.15*d), (float)(d + 0.15*d) ]: the arithmetic is done in
 * double (TOLERANCE is a double literal), the bounds are then stored in
 * `const float`, and the double elapsed_us is compared against them.
 */
static int in_window(double elapsed, uint64_t dur)
{
    const float lo = (double)dur - (TOL * (double)dur);
    const float hi = (double)dur + (TOL * (double)dur);
    return elapsed >= lo && elapsed <= hi;
}

/* src/state_machine.c:365-385 (append) + :388-419 (actions). */
static int run_action(ook_sm *sm, uint8_t action)
{
    switch (action) {
    case OOK_ACT_NONE:
        return OOK_RESULT_NO_OUTPUT;
    case OOK_ACT_APPEND_0:
    case OOK_ACT_APPEND_1:
        /* The reference stores the bit while num_bits <= max_bits (sic) and
         * increments num_bits regardless. */
        if (sm->num_bits <= sm->d.max_bits) {
            unsigned byte = sm->num_bits / 8;
            unsigned bit = sm->num_bits % 8;
            if (byte <= OOK_MAX_PAYLOAD_BYTES) {
                if (action == OOK_ACT_APPEND_1) {
                    sm->data[byte] |= (uint8_t)(1u << bit);
                } else {
                    sm->data[byte] &= (uint8_t)~(1u << bit);
                }
            }
        }
        sm->num_bits++;
        return OOK_RESULT_NO_OUTPUT;
    case OOK_ACT_OUTPUT_DATA:
        return OOK_RESULT_OUTPUT_READY;
    default:
        return OOK_RESULT_ERROR;
    }
}

/* src/state_machine.c:421-519: one evaluation of the current state's
 * triggers against sample b. */
static int eval_triggers(ook_sm *sm, int b)
{
    const uint32_t s = sm->cur;
    int fired = -1;
    int edge_trigger = 0;

    for (uint32_t t = sm->tbeg[s]; t < sm->tbeg[s + 1] && fired < 0; t++) {
        if (sm->tdur[t] != 0 && !in_window(sm->elapsed_us, sm->tdur[t])) {
            continue;               /* :119-133, :433 */
        }
        switch (sm->tcond[t]) {
        case OOK_COND_ALWAYS:
            fired = (int)t;
            break;
        case OOK_COND_PULSE_START:
            if (!sm->prev_bit && b) {
                fired = (int)t;
                edge_trigger = 1;
            }
            break;
        case OOK_COND_PULSE_END:
            if (sm->prev_bit && !b) {
                fired = (int)t;
                edge_trigger = 1;
            }
            break;
        case OOK_COND_TIMEOUT:
            if (sm->sto[s] != 0 && sm->elapsed_us >= (double)sm->sto[s]) {
                fired = (int)t;
            }
            break;
        case OOK_COND_MSG_COMPLETE:
            if (sm->num_bits >= sm->d.max_bits) {
                fired = (int)t;
            }
            break;
        default:
            return OOK_RESULT_ERROR;    /* :476-479, leaves elapsed alone */
        }
    }

    if (fired < 0) {
        /* :513-515 with :78-82: to_duration_us(sm, 1) */
        sm->elapsed_us += ((double)1u / (double)sm->d.sample_rate) * 1e6;
        return OOK_RESULT_NO_OUTPUT;
    }

    int result;
    int dur_ok = 1;
    if (edge_trigger && sm->sdur[s] != 0) {
        dur_ok = in_window(sm->elapsed_us, sm->sdur[s]);    /* :100-117 */
    }
    if (dur_ok) {
        result = run_action(sm, sm->tact[fired]);
        if (result != OOK_RESULT_ERROR) {
            sm->cur = sm->tnext[fired];
        }
    } else {
        result = OOK_RESULT_ERROR;
    }
    if (result == OOK_RESULT_ERROR) {
        sm->cur = 0;                /* :505-509 */
    }
    sm->elapsed_us = 0;             /* :511 */
    return result;
}

/* src/state_machine.c:521-539: reset is passed through on the same sample. */
static int step(ook_sm *sm, int b)
{
    if (sm->cur == 0) {
        sm->num_bits = 0;
        memset(sm->data, 0, (sm->d.max_bits + 7) / 8);
        int r = eval_triggers(sm, b);
        if (r != 0) {
            return r;
        }
    }
    return eval_triggers(sm, b);
}

/* src/state_machine.c:541-556 */
int ook_sm_process(ook_sm *sm, const uint8_t *bits, unsigned count,
                   unsigned *num_proc)
{
    int result = OOK_RESULT_NO_OUTPUT;
    unsigned i;
    for (i = 0; i < count && result == OOK_RESULT_NO_OUTPUT; i++) {
        result = step(sm, bits[i] != 0);
        sm->prev_bit = bits[i] != 0;
    }
    *num_proc = i;
    return result;
}

/* ------------------------------------------------------------------ */
/* Whole-path driver                                                   */
/* ------------------------------------------------------------------ */

uint64_t ook_oracle_rx(const int16_t *iq, uint64_t num_samples,
                       const ook_fir_desc *fir, float threshold,
                       const ook_fsm_desc *fsm, uint32_t spb,
                       ook_msg *msgs, uint64_t msg_cap, uint64_t *num_msgs,
                       uint64_t *err_samples, uint64_t err_cap,
                       uint64_t *num_errs,
                       uint8_t *bits_out, ook_cf *fir_out)
{
    uint64_t n_msgs = 0, n_errs = 0, dec_total = 0;
    ook_fir *f = fir ? ook_fir_new(fir) : NULL;
    ook_sm *sm = fsm ? ook_sm_new(fsm) : NULL;
    int16_t *raw = malloc(sizeof(int16_t) * 2 * (size_t)spb);
    ook_cf *samples = malloc(sizeof(ook_cf) * spb);
    ook_cf *post = malloc(sizeof(ook_cf) * spb);
    uint8_t *bits = malloc(spb);
    const unsigned payload_bytes = fsm ? (fsm->max_bits + 7) / 8 : 0;

    uint64_t consumed = 0;
    /* src/ookiedokie.c:238: one iteration per buffer until the backend
     * reports EOF. */
    while (consumed < num_samples) {
        /* src/sdr/bladeRF_file.c:107-119: short final read is zero padded
         * and still processed as a full buffer; a read of 0 items is EOF and
         * the buffer is discarded (ookiedokie.c:243-246). */
        uint64_t avail = num_samples - consumed;
        size_t got = avail < spb ? (size_t)avail : spb;
        memcpy(raw, iq + 2 * consumed, got * 2 * sizeof(int16_t));
        if (got < spb) {
            memset(raw + 2 * got, 0, (spb - got) * 2 * sizeof(int16_t));
        }
        consumed += got;
        ook_unpack(raw, samples, spb);

        const ook_cf *to_thr;
        size_t count;
        if (f) {                    /* ookiedokie.c:255-263 */
            count = ook_fir_run(f, samples, spb, post);
            to_thr = post;
        } else {
            count = spb;
            to_thr = samples;
        }
        if (fir_out) {
            memcpy(fir_out + dec_total, to_thr, count * sizeof(ook_cf));
        }

        ook_threshold(to_thr, threshold, bits, count);  /* :272-274 */
        if (bits_out) {
            memcpy(bits_out + dec_total, bits, count);
        }

        if (sm) {
            /* src/device.c:634-658: keep calling sm_process until the
             * buffer is consumed, but stop for the rest of THIS buffer after
             * an ERROR. */
            unsigned total = 0, nproc = 0;
            int r = OOK_RESULT_NO_OUTPUT;
            while (total < count && r != OOK_RESULT_ERROR) {
                r = ook_sm_process(sm, bits + total,
                                   (unsigned)count - total, &nproc);
                total += nproc;
                if (r == OOK_RESULT_OUTPUT_READY) {
                    if (msgs && n_msgs < msg_cap) {
                        msgs[n_msgs].sample = dec_total + total - 1;
                        memset(msgs[n_msgs].payload, 0,
                               OOK_MAX_PAYLOAD_BYTES);
                        memcpy(msgs[n_msgs].payload, sm->data,
                               payload_bytes);
                    }
                    n_msgs++;
                } else if (r == OOK_RESULT_ERROR) {
                    if (err_samples && n_errs < err_cap) {
                        err_samples[n_errs] = dec_total + total - 1;
                    }
                    n_errs++;
                }
            }
        }
        dec_total += count;
    }

    if (num_msgs) *num_msgs = n_msgs;
    if (num_errs) *num_errs = n_errs;
    free(raw);
    free(samples);
    free(post);
    free(bits);
    ook_fir_free(f);
    ook_sm_free(sm);
    return dec_total;
}

/* ------------------------------------------------------------------ */
/* Integer sample-count windows                                        */
/* ------------------------------------------------------------------ */

#define REPLAY_LIMIT (1ull << 33)

int ook_duration_window(uint32_t rate, uint64_t dur, uint64_t *kmin,
                        uint64_t *kmax)
{
    const float lo = (double)dur - (TOL * (double)dur);
    const float hi = (double)dur + (TOL * (double)dur);
    const double delta = ((double)1u / (double)rate) * 1e6;
    double e = 0.0;
    uint64_t k = 0;
    int have_min = 0;
    *kmin = 1;
    *kmax = 0;                      /* empty window unless found */
    while (k < REPLAY_LIMIT) {
        if (e > hi) {
            return 0;
        }
        if (e >= lo) {
            if (!have_min) {
                *kmin = k;
                have_min = 1;
            }
            *kmax = k;
        }
        e += delta;
        k++;
    }
    return -1;
}

int ook_timeout_count(uint32_t rate, uint64_t timeout_us, uint64_t *kto)
{
    const double delta = ((double)1u / (double)rate) * 1e6;
    const double t = (double)timeout_us;
    double e = 0.0;
    uint64_t k = 0;
    while (k < REPLAY_LIMIT) {
        if (e >= t) {
            *kto = k;
            return 0;
        }
        e += delta;
        k++;
    }
    return -1;
}
