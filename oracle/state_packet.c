/*
 * oracle/state_packet.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * StatePacket wire layout: DataPacket<T>::encode/decode (data_packet.h:313-333),
 * StateBuffer::encode/decode (data_packet.cpp:143-174), BinaryBuffer append/read
 * of PODs, strings and vectors (memory_util.h:307-333,362-388):
 *
 *   u64 n_keys
 *   repeat n_keys:
 *     u64 key_len ; key bytes ; NUL
 *     u8  flags   (1 reals | 2 pixels | 4 id | 8 str)
 *     [u64 n ; f32 x n]   if reals
 *     [u64 n ; u8  x n]   if pixels
 *     [u64 n ; i32 x n]   if id
 *     [u64 len ; bytes ; NUL] if str
 *
 * little-endian, size_t = 8 bytes.  Key order on the wire is the reference's
 * unordered_map iteration order (unspecified); this encoder writes fields in
 * the order given, the decoder accepts any order.
 * Pinned by the round trip of tests/test_statepacket.cpp:77-104.
 */
#include "oracle.h"
#include <string.h>

typedef struct { uint8_t *p; size_t cap, n; } wr;

static void put(wr *w, const void *d, size_t len) {
    if (w->p && w->n + len <= w->cap) memcpy(w->p + w->n, d, len);
    w->n += len;
}
static void put_u64(wr *w, uint64_t v) { put(w, &v, 8); }
static void put_str(wr *w, const char *s) {
    size_t len = strlen(s);
    put_u64(w, (uint64_t)len);
    put(w, s, len + 1);
}

size_t orc_packet_encode(const orc_packet_field *f, int n_fields, uint8_t *out, size_t cap) {
    wr w = {out, cap, 0};
    put_u64(&w, (uint64_t)n_fields);
    for (int i = 0; i < n_fields; ++i) {
        put_str(&w, f[i].key);
        uint8_t flags = (uint8_t)((f[i].has_reals ? 1 : 0) | (f[i].has_pixels ? 2 : 0) |
                                  (f[i].has_id ? 4 : 0) | (f[i].has_str ? 8 : 0));
        put(&w, &flags, 1);
        if (f[i].has_reals) { put_u64(&w, f[i].n_reals); put(&w, f[i].reals, 4 * f[i].n_reals); }
        if (f[i].has_pixels) { put_u64(&w, f[i].n_pixels); put(&w, f[i].pixels, f[i].n_pixels); }
        if (f[i].has_id) { put_u64(&w, f[i].n_id); put(&w, f[i].id, 4 * f[i].n_id); }
        if (f[i].has_str) put_str(&w, f[i].str);
    }
    return w.n;
}

typedef struct { const uint8_t *p; size_t len, at; int bad; } rd;

static const void *take(rd *r, size_t n) {
    if (r->at + n > r->len) { r->bad = 1; return r->p; }
    const void *q = r->p + r->at;
    r->at += n;
    return q;
}
static uint64_t take_u64(rd *r) { uint64_t v = 0; memcpy(&v, take(r, 8), 8); return r->bad ? 0 : v; }
static const char *take_str(rd *r) {
    uint64_t len = take_u64(r);
    return (const char *)take(r, (size_t)len + 1);
}

int orc_packet_decode(const uint8_t *buf, size_t len, orc_packet_field *f, int max_fields) {
    rd r = {buf, len, 0, 0};
    uint64_t n = take_u64(&r);
    if (r.bad || n > (uint64_t)max_fields) return -1;
    for (uint64_t i = 0; i < n; ++i) {
        memset(&f[i], 0, sizeof f[i]);
        f[i].key = take_str(&r);
        uint8_t flags = *(const uint8_t *)take(&r, 1);
        if (r.bad) return -1;
        if (flags & 1) { f[i].has_reals = 1; f[i].n_reals = (size_t)take_u64(&r); f[i].reals = (const float *)take(&r, 4 * f[i].n_reals); }
        if (flags & 2) { f[i].has_pixels = 1; f[i].n_pixels = (size_t)take_u64(&r); f[i].pixels = (const uint8_t *)take(&r, f[i].n_pixels); }
        if (flags & 4) { f[i].has_id = 1; f[i].n_id = (size_t)take_u64(&r); f[i].id = (const int32_t *)take(&r, 4 * f[i].n_id); }
        if (flags & 8) { f[i].has_str = 1; f[i].str = take_str(&r); }
        if (r.bad) return -1;
    }
    return (r.at == len) ? (int)n : -1;
}
