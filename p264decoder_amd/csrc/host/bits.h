/* bits.h - MSB-first bit reader over an RBSP buffer, with Exp-Golomb helpers.
 *
 * Replaces core/bs.h:54-170 of the reference.  Built around a 64-bit window that is
 * refilled four bytes at a time, so peeks of up to 32 bits are branch-light and reads past the end
 * return zeros (the reference's bs_read stops at p_end; its bs_show over-reads 3 bytes,
 * core/bs.h:119-126 - we pad instead).
 */
#ifndef P264_BITS_H
#define P264_BITS_H
#include <stdint.h>
#include <stddef.h>
#include <string.h>

typedef struct {
    const uint8_t *buf;
    size_t   size;      /* bytes */
    size_t   pos;       /* next byte to load into the window */
    uint64_t win;       /* bits are left-aligned: next bit is bit 63 */
    int      avail;     /* valid bits in win */
} bitrd_t;
/* total bits consumed: every loaded byte adds 8 to both terms (not counted per skip: one add less in the hottest function) */
static inline size_t br_consumed(const bitrd_t *b) { return b->pos * 8 - (size_t)b->avail; }

static inline void br_refill(bitrd_t *b)
{
    /* keep at least 33 valid bits in the window (peeks take up to 32): four bytes at a time while the buffer lasts,
     * bytewise (zero-padded) at its end */
    if (b->avail > 32) return;
    if (b->pos + 4 <= b->size) {
        uint32_t v;
        memcpy(&v, b->buf + b->pos, 4);
        b->win |= (uint64_t)__builtin_bswap32(v) << (32 - b->avail);
        b->pos += 4; b->avail += 32;
        return;
    }
    while (b->avail <= 56) {
        uint64_t byte = b->pos < b->size ? b->buf[b->pos] : 0;
        b->pos++;
        b->win |= byte << (56 - b->avail);
        b->avail += 8;
    }
}

static inline void br_init(bitrd_t *b, const uint8_t *buf, size_t size)
{
    b->buf = buf; b->size = size; b->pos = 0; b->win = 0; b->avail = 0;
    br_refill(b);
}

/* 1 <= n <= 32 */
static inline uint32_t br_peek(bitrd_t *b, int n) { return (uint32_t)(b->win >> (64 - n)); }

static inline void br_skip(bitrd_t *b, int n)
{
    b->win <<= n; b->avail -= n;
    br_refill(b);
}

static inline uint32_t br_u(bitrd_t *b, int n)
{
    if (n == 0) return 0;
    uint32_t v = br_peek(b, n);
    br_skip(b, n);
    return v;
}

static inline uint32_t br_u1(bitrd_t *b) { return br_u(b, 1); }

/* bits left in the buffer (may go negative after an over-read) */
static inline long br_bits_left(const bitrd_t *b) { return (long)(b->size * 8) - (long)br_consumed(b); }
static inline int  br_overrun(const bitrd_t *b)   { return br_consumed(b) > b->size * 8; }
/* mirrors bs_eof (core/bs.h:63-66): true once the byte cursor reached the end */
static inline int  br_eof(const bitrd_t *b)       { return (br_consumed(b) >> 3) >= b->size; }

static inline uint32_t br_ue(bitrd_t *b)
{
    /* leading zeros of the next 32 bits in one step (the window always holds at least 33 valid bits; past the end of the
     * buffer it is zero-padded, so a truncated code reads as "32 zeros" and is rejected by the callers' range checks) */
    const uint32_t w = br_peek(b, 32);
    const int zeros = w ? __builtin_clz(w) : 32;
    if (zeros <= 15) {                   /* the whole code (2 * zeros + 1 bits) is inside the peeked word: one step */
        br_skip(b, 2 * zeros + 1);
        return (w >> (31 - 2 * zeros)) - 1u;
    }
    br_skip(b, zeros);
    br_skip(b, 1);                       /* the terminating 1 */
    if (zeros >= 32) return 0xffffffffu;
    return ((1u << zeros) - 1u) + br_u(b, zeros);
}

static inline int32_t br_se(bitrd_t *b)
{
    uint32_t k = br_ue(b);
    return (k & 1) ? (int32_t)((k + 1) >> 1) : -(int32_t)(k >> 1);
}

/* truncated Exp-Golomb, range [0, max] */
static inline uint32_t br_te(bitrd_t *b, int max)
{
    if (max == 1) return br_u1(b) ^ 1u;
    if (max > 1)  return br_ue(b);
    return 0;
}
#endif
