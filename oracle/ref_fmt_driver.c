/*
 * ref_fmt_driver.c -- thin driver around the REFERENCE's own formatter.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is ours; oracle/Makefile compiles it
 * together with /root/reference/src/{formatter,conversions,keyval_list,log}.c
 * (where they lie, never copied) into oracle/_ref/libookref.so.  It lets the
 * tests put the same field descriptions and payloads through the real
 * formatter_data_to_keyval / formatter_keyval_to_data and through
 * libookiedokie_amd's restatement (csrc/formatter.cpp).
 *
 * rx_print (src/ookiedokie.c:181-220) is a static function of a file that
 * does not link without the SDR / FIR / jansson parts, so the printer has no
 * reference build; its format strings are pinned by hand-derived expectations
 * in tests/test_formatter.py.
 */
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "formatter.h"
#include "keyval_list.h"

void *ref_fmt_new(unsigned num_fields, unsigned max_bit)
{
    return formatter_init(num_fields, max_bit, FORMATTER_TS_NONE);
}

void ref_fmt_free(void *f) { formatter_deinit((struct formatter *) f); }

/* format: 1 hex .. 6 enum; endianness: 1 big, 2 little (formatter.h enums) */
int ref_fmt_add_field(void *f, const char *name, unsigned start_bit, unsigned end_bit,
                      int format, size_t enum_count, int endianness, float scaling, float offset)
{
    return formatter_add_field((struct formatter *) f, name, start_bit, end_bit,
                               (enum formatter_fmt) format, enum_count,
                               (enum formatter_endianness) endianness, scaling, offset) ? 0 : -1;
}

int ref_fmt_add_enum(void *f, const char *field, const char *name, uint64_t value)
{
    return formatter_add_field_enum((struct formatter *) f, field, name, spt_from_uint64(value)) ? 0 : -1;
}

int ref_fmt_set_default(void *f, const char *field, const char *value)
{
    return formatter_set_field_default((struct formatter *) f, field, value) ? 0 : -1;
}

int ref_fmt_initialized(void *f) { return formatter_initialized((struct formatter *) f) ? 1 : 0; }

/* "key\tvalue\n" for every pair formatter_data_to_keyval produces */
int ref_fmt_format(void *f, const uint8_t *data, char *out, size_t cap)
{
    struct keyval_list *kv = keyval_list_init();
    size_t i, used = 0;
    int rc = 0;

    if (!kv) return -1;
    if (!formatter_data_to_keyval((struct formatter *) f, data, kv)) rc = -1;
    for (i = 0; rc == 0 && i < keyval_list_size(kv); i++) {
        const struct keyval *p = keyval_list_at(kv, i);
        int n = snprintf(out + used, cap - used, "%s\t%s\n", p->key, p->value);
        if (n < 0 || (size_t) n >= cap - used) { rc = -2; break; }
        used += (size_t) n;
    }
    keyval_list_deinit(kv);
    return rc;
}

void ref_fmt_default_data(void *f, uint8_t *data)
{
    formatter_default_data((struct formatter *) f, data);
}

int ref_fmt_set(void *f, const char *key, const char *value, uint8_t *data)
{
    struct keyval_list *kv = keyval_list_init();
    struct keyval p;
    int rc;

    if (!kv) return -1;
    p.key = key;
    p.value = value;
    keyval_list_append(kv, &p);
    rc = formatter_keyval_to_data((struct formatter *) f, kv, data) ? 0 : -1;
    keyval_list_deinit(kv);
    return rc;
}
