/* Minimal SHA-256 (FIPS 180-4) for test infrastructure: per-frame digests of
 * decoded YUV so that whole-stream parity can be pinned by small fixtures.
 * TEST INFRASTRUCTURE ONLY - not part of the product path. */
#ifndef ORACLE_SHA256_H
#define ORACLE_SHA256_H
#include <stdint.h>
#include <string.h>
#include <stddef.h>

typedef struct {
    uint32_t st[8];
    uint64_t nbytes;
    uint8_t  buf[64];
    unsigned fill;
} sha256_t;

static const uint32_t sha256_k[64] = {
    0x428a2f98,0x71374491,0xb5c0fbcf,0xe9b5dba5,0x3956c25b,0x59f111f1,0x923f82a4,0xab1c5ed5,
    0xd807aa98,0x12835b01,0x243185be,0x550c7dc3,0x72be5d74,0x80deb1fe,0x9bdc06a7,0xc19bf174,
    0xe49b69c1,0xefbe4786,0x0fc19dc6,0x240ca1cc,0x2de92c6f,0x4a7484aa,0x5cb0a9dc,0x76f988da,
    0x983e5152,0xa831c66d,0xb00327c8,0xbf597fc7,0xc6e00bf3,0xd5a79147,0x06ca6351,0x14292967,
    0x27b70a85,0x2e1b2138,0x4d2c6dfc,0x53380d13,0x650a7354,0x766a0abb,0x81c2c92e,0x92722c85,
    0xa2bfe8a1,0xa81a664b,0xc24b8b70,0xc76c51a3,0xd192e819,0xd6990624,0xf40e3585,0x106aa070,
    0x19a4c116,0x1e376c08,0x2748774c,0x34b0bcb5,0x391c0cb3,0x4ed8aa4a,0x5b9cca4f,0x682e6ff3,
    0x748f82ee,0x78a5636f,0x84c87814,0x8cc70208,0x90befffa,0xa4506ceb,0xbef9a3f7,0xc67178f2 };

static inline uint32_t sha256_ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void sha256_block(sha256_t *c, const uint8_t *p)
{
    uint32_t w[64], a, b, d, e, f, g, h, cc;
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)p[4*i] << 24) | ((uint32_t)p[4*i+1] << 16) | ((uint32_t)p[4*i+2] << 8) | p[4*i+3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = sha256_ror(w[i-15], 7) ^ sha256_ror(w[i-15], 18) ^ (w[i-15] >> 3);
        uint32_t s1 = sha256_ror(w[i-2], 17) ^ sha256_ror(w[i-2], 19) ^ (w[i-2] >> 10);
        w[i] = w[i-16] + s0 + w[i-7] + s1;
    }
    a = c->st[0]; b = c->st[1]; cc = c->st[2]; d = c->st[3];
    e = c->st[4]; f = c->st[5]; g = c->st[6]; h = c->st[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = sha256_ror(e, 6) ^ sha256_ror(e, 11) ^ sha256_ror(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = h + S1 + ch + sha256_k[i] + w[i];
        uint32_t S0 = sha256_ror(a, 2) ^ sha256_ror(a, 13) ^ sha256_ror(a, 22);
        uint32_t mj = (a & b) ^ (a & cc) ^ (b & cc);
        uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = cc; cc = b; b = a; a = t1 + t2;
    }
    c->st[0] += a; c->st[1] += b; c->st[2] += cc; c->st[3] += d;
    c->st[4] += e; c->st[5] += f; c->st[6] += g; c->st[7] += h;
}

static void sha256_init(sha256_t *c)
{
    static const uint32_t iv[8] = { 0x6a09e667,0xbb67ae85,0x3c6ef372,0xa54ff53a,
                                    0x510e527f,0x9b05688c,0x1f83d9ab,0x5be0cd19 };
    memcpy(c->st, iv, sizeof iv);
    c->nbytes = 0; c->fill = 0;
}

static void sha256_update(sha256_t *c, const void *data, size_t n)
{
    const uint8_t *p = (const uint8_t *)data;
    c->nbytes += n;
    while (n) {
        if (c->fill == 0 && n >= 64) { sha256_block(c, p); p += 64; n -= 64; continue; }
        size_t take = 64 - c->fill; if (take > n) take = n;
        memcpy(c->buf + c->fill, p, take);
        c->fill += (unsigned)take; p += take; n -= take;
        if (c->fill == 64) { sha256_block(c, c->buf); c->fill = 0; }
    }
}

static void sha256_final(sha256_t *c, uint8_t out[32])
{
    uint64_t bits = c->nbytes * 8;
    uint8_t pad = 0x80;
    sha256_update(c, &pad, 1);
    pad = 0;
    while (c->fill != 56) sha256_update(c, &pad, 1);
    uint8_t len[8];
    for (int i = 0; i < 8; i++) len[i] = (uint8_t)(bits >> (56 - 8*i));
    sha256_update(c, len, 8);
    for (int i = 0; i < 8; i++) {
        out[4*i] = (uint8_t)(c->st[i] >> 24); out[4*i+1] = (uint8_t)(c->st[i] >> 16);
        out[4*i+2] = (uint8_t)(c->st[i] >> 8); out[4*i+3] = (uint8_t)c->st[i];
    }
}

static void sha256_hex(const uint8_t d[32], char hex[65])
{
    static const char *x = "0123456789abcdef";
    for (int i = 0; i < 32; i++) { hex[2*i] = x[d[i] >> 4]; hex[2*i+1] = x[d[i] & 15]; }
    hex[64] = 0;
}
#endif
