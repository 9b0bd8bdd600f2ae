/* cabac.c - tables and set-up of the CABAC decoding engine (cabac.h), and the flat entry point the known-answer test
 * drives (include/p264parse.h: p264cabac_decode_ops). */
#include <string.h>
#include "cabac.h"
#include "p264parse.h"
#include "host_cpu.h"

void p264cabac_init_contexts(p264cabac_t *c, int is_i_slice, int cabac_init_idc, int slice_qp)
{
    const int t = is_i_slice ? 0 : 1 + (cabac_init_idc < 0 ? 0 : cabac_init_idc > 2 ? 2 : cabac_init_idc);
    const int qp = slice_qp < 0 ? 0 : slice_qp > 51 ? 51 : slice_qp;
    for (int i = 0; i < P264_CABAC_CONTEXTS; i++) {
        int pre = ((cabac_mn[i][t][0] * qp) >> 4) + cabac_mn[i][t][1];           /* preCtxState, 9.3.1.1 */
        pre = pre < 1 ? 1 : pre > 126 ? 126 : pre;
        c->state[i] = pre <= 63 ? (uint16_t)((63 - pre) << 1) : (uint16_t)((pre - 64) << 1 | 1);
    }
    /* state transitions (9.3.3.2.1.1) on the packed state byte: MPS -> transIdxMPS = min(pStateIdx + 1, 62); LPS -> table 9-45,
     * and at pStateIdx 0 the MPS flips */
    for (int s = 0; s < 128; s++) {
        const int idx = s >> 1, mps = s & 1;
        c->next[s][0] = (uint8_t)((idx < 62 ? idx + 1 : 62) << 1 | mps);
        c->next[s][1] = (uint8_t)(cabac_trans_lps[idx] << 1 | (idx == 0 ? mps ^ 1 : mps));
    }
}

void p264cabac_start(p264cabac_t *c, const uint8_t *data, size_t bytes)
{
    c->data = data; c->size = bytes; c->pos = 0;
    c->low = 0; c->n = -9;                                 /* the first refill leaves codIOffset = the first 9 bits, 23 bits behind it */
    c->range = 510;
    p264cabac_refill(c);
}

int p264cabac_decode_ops(const uint8_t *data, size_t bytes, int is_i_slice, int cabac_init_idc, int slice_qp,
                         const int16_t *ops, int n_ops, uint8_t *bins)
{
    if (p264amd_cpu_refuse("p264cabac_decode_ops")) return -1;
    if (!data || !ops || !bins || n_ops < 0) return -1;
    p264cabac_t c;
    p264cabac_init_contexts(&c, is_i_slice, cabac_init_idc, slice_qp);
    p264cabac_start(&c, data, bytes);
    for (int i = 0; i < n_ops; i++) {
        const int op = ops[i];
        if (op >= P264_CABAC_CONTEXTS || op < -2) return -1;
        bins[i] = (uint8_t)(op >= 0 ? p264cabac_decision(&c, op) : op == -1 ? p264cabac_bypass(&c) : p264cabac_terminate(&c));
    }
    return p264cabac_bits_left(&c) < 0 ? 1 : 0;
}
