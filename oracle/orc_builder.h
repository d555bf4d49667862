/*
 * orc_builder.h -- growable columnar batch shared by the two CPU checkers.
 * TEST INFRASTRUCTURE ONLY (see mdc_oracle.h).
 */
#ifndef ORC_BUILDER_H
#define ORC_BUILDER_H
#include "mdc_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ batch building */

typedef struct {
    orc_batch b;
    int64_t cap_rows;
    uint64_t *bytes_cap;
} builder;

static inline builder *builder_new(int ncols, const int *typed_on) {
    builder *B = (builder *)calloc(1, sizeof(builder));
    orc_batch *b = &B->b;
    b->ncols = ncols;
    b->words_per_row = (ncols + 31) / 32;
    if (b->words_per_row == 0) b->words_per_row = 1;
    b->off = (uint32_t **)calloc((size_t)ncols + 1, sizeof(uint32_t *));
    b->bytes = (uint8_t **)calloc((size_t)ncols + 1, sizeof(uint8_t *));
    b->nbytes = (uint64_t *)calloc((size_t)ncols + 1, sizeof(uint64_t));
    b->val = (uint64_t **)calloc((size_t)ncols + 1, sizeof(uint64_t *));
    B->bytes_cap = (uint64_t *)calloc((size_t)ncols + 1, sizeof(uint64_t));
    B->cap_rows = 1024;
    b->rec_off = (uint64_t *)calloc((size_t)B->cap_rows + 1, 8);
    b->valid = (uint32_t *)calloc((size_t)B->cap_rows * b->words_per_row, 4);
    b->found = (uint32_t *)calloc((size_t)B->cap_rows * b->words_per_row, 4);
    for (int c = 0; c < ncols; ++c) {
        b->off[c] = (uint32_t *)calloc((size_t)B->cap_rows + 1, 4);
        if (typed_on[c] != ORC_T_NONE) b->val[c] = (uint64_t *)calloc((size_t)B->cap_rows, 8);
    }
    return B;
}

static inline void builder_grow_rows(builder *B) {
    orc_batch *b = &B->b;
    int64_t oc = B->cap_rows, nc = oc * 2;
    b->rec_off = (uint64_t *)realloc(b->rec_off, ((size_t)nc + 1) * 8);
    b->valid = (uint32_t *)realloc(b->valid, (size_t)nc * b->words_per_row * 4);
    b->found = (uint32_t *)realloc(b->found, (size_t)nc * b->words_per_row * 4);
    memset(b->valid + (size_t)oc * b->words_per_row, 0, (size_t)(nc - oc) * b->words_per_row * 4);
    memset(b->found + (size_t)oc * b->words_per_row, 0, (size_t)(nc - oc) * b->words_per_row * 4);
    for (int c = 0; c < b->ncols; ++c) {
        b->off[c] = (uint32_t *)realloc(b->off[c], ((size_t)nc + 1) * 4);
        if (b->val[c]) {
            b->val[c] = (uint64_t *)realloc(b->val[c], (size_t)nc * 8);
            memset(b->val[c] + oc, 0, (size_t)(nc - oc) * 8);
        }
    }
    B->cap_rows = nc;
}

static inline void builder_put_cell(builder *B, int64_t row, int c, const uint8_t *s, size_t n) {
    orc_batch *b = &B->b;
    if (b->nbytes[c] + n > B->bytes_cap[c]) {
        uint64_t nc = B->bytes_cap[c] ? B->bytes_cap[c] * 2 : 4096;
        while (nc < b->nbytes[c] + n) nc *= 2;
        b->bytes[c] = (uint8_t *)realloc(b->bytes[c], nc);
        B->bytes_cap[c] = nc;
    }
    if (n) memcpy(b->bytes[c] + b->nbytes[c], s, n);
    b->nbytes[c] += n;
    b->off[c][row + 1] = (uint32_t)b->nbytes[c];
}

static inline void default_typed(const orc_col *cols, int ncols, const int *typed, int *out) {
    for (int c = 0; c < ncols; ++c) {
        if (typed)
            out[c] = typed[c];
        else if (strcmp(cols[c].datatype, "number") == 0)
            out[c] = ORC_T_F64;
        else if (strcmp(cols[c].datatype, "date") == 0)
            out[c] = ORC_T_I64;
        else
            out[c] = ORC_T_NONE;
    }
}


static inline void builder_free(orc_batch *b) {
    if (!b) return;
    builder *B = (builder *)b; /* orc_batch is the first member */
    for (int c = 0; c < b->ncols; ++c) {
        free(b->off[c]);
        free(b->bytes[c]);
        free(b->val[c]);
    }
    free(b->off);
    free(b->bytes);
    free(b->nbytes);
    free(b->val);
    free(b->rec_off);
    free(b->valid);
    free(b->found);
    free(B->bytes_cap);
    free(B);
}
#endif
